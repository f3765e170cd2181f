// Internal declarations shared by the C-ABI shim (mdc_api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "mdc.h"

namespace mdc {

void set_error(const char* fmt, ...);

// Nothing throws across the boundary: the library is built WITH exceptions so that an allocation failure inside
// (std::vector growth in the packers, event lists) surfaces as MDC_ENOMEM instead of aborting the host process.
template <class Fn>
inline int guarded(const char* what, Fn&& fn) noexcept {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", what);
        return MDC_ENOMEM;
    } catch (const std::exception& e) {
        set_error("%s: %s", what, e.what());
        return MDC_EIO;
    } catch (...) {
        set_error("%s: unexpected exception", what);
        return MDC_EIO;
    }
}

#define MDC_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            mdc::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MDC_EIO;                                                                \
        }                                                                                  \
    } while (0)

// frame geometry (reference: in_shp = [2, 128], cnn.py:89 / CNN.ipynb cell 5)
constexpr int kRows = 2;
constexpr int kSamples = 128;
constexpr int kFrameFloats = kRows * kSamples;

// canonical VT-CNN2 dimensions (RML2016.10a_VTCNN2_example.ipynb:190-210)
constexpr int kC1 = 256;     // conv1 filters
constexpr int kW1 = 130;     // conv1 output width
constexpr int kC2 = 80;      // conv2 filters
constexpr int kW2 = 132;     // conv2 output width
constexpr int kFeat = kC2 * kW2;   // 10560
constexpr int kHid = 256;    // dense1 units

// Feature row of the 16-bit modes (bf16 features; round 3): output positions in PAIRS of 160 elements --
//   [64 channels of the even position][64 channels of the odd position][channels 64..79, each as (even, odd)]
// so that the conv kernels' finishing lanes, which own ONE channel of the fifth output tile per position, store one
// dword per pair of positions instead of one 2-byte store per position (a global_store_short costs a 16-cycle MFMA gap
// 16 cycles, a dword store 4: tools/microbench/mfma_gap.hip).  The order of K is dense1's to choose: its weight rows
// are permuted with the same function at pack time; the conv / flat taps un-permute.  (f32 features: [w][o], unchanged.)
__host__ __device__ constexpr int feat16_index(int w, int o) {
    return (w >> 1) * (2 * kC2) + (o < 64 ? (w & 1) * 64 + o : 128 + 2 * (o - 64) + (w & 1));
}

// Feature row of the fp8 mode's E4M3 features (round 4; one byte per element, one power-of-two scale per tensor): output
// positions in GROUPS OF FOUR of 320 bytes --
//   [pair (w0, w1): sixteen 8-byte slots {4 channels of w0, the same 4 channels of w1}]  128 B, channels 0..63
//   [pair (w2, w3): the same]                                                            128 B
//   [channels 64..79, each as the bytes (w0, w1, w2, w3)]                                 64 B
// so that the fp8 conv kernel's finishing lanes store one dwordx2 per position PAIR (their four channels of both
// positions) and one dword per GROUP (their fifth-tile channel of all four) -- no store at all on even steps.  As with
// feat16_index the order of K is dense1's to choose: its weight rows are permuted with this function at pack time.
__host__ __device__ constexpr int feat8_index(int w, int o) {
    return (w >> 2) * (4 * kC2) + (o < 64 ? ((w >> 1) & 1) * 128 + 8 * (o >> 2) + 4 * (w & 1) + (o & 3) : 256 + 4 * (o - 64) + (w & 3));
}

// 8 raw bytes from an address that is only 2-byte aligned (a window of a uint8 I/Q capture at an arbitrary hop):
// gfx950 global loads take unaligned addresses, and hipcc emits ONE global_load_dwordx2 for this
__device__ __forceinline__ uint2 load8_unaligned(const unsigned char* p) {
    uint2 r;
    __builtin_memcpy(&r, p, 8);
    return r;
}

// alternates (test build only, see mdc_model::alt)
enum { kAltConvHipcc = 1,      // MDC_CONV_SCHED=0: the hipcc-scheduled statement of the bf16 conv kernel (vtcnn2_bf16.hip)
       kAltDense1Simple = 2,   // MDC_DENSE1_PHASED=0: one barrier per K-tile (the race screen of the phased kernel)
       kAltDepF32Mfma = 4,     // MDC_DEP_F32_MFMA=1: deployed nets' f32 dense layer on the f32 matrix pipe (deployed_f32m.hip)
       kAltSeparateHead = 8 }; // MDC_D1_FUSED_HEAD=0: dense2 + softmax as their own launch at every batch size

struct ProfSlot {
    const char* name;
    std::vector<hipEvent_t> ev;   // start/stop pairs
    double total_ms = 0;
    int64_t launches = 0;
};

}  // namespace mdc

struct mdc_model {
    mdc_topology topo{};
    int device = 0;
    int dtype = MDC_F32;
    bool finalized = false;
    int nlayers = 0;
    size_t nk[4]{}, nb[4]{};
    std::vector<float> hk[4], hb[4];   // host copies until finalize
    bool have[4]{};

    // device-resident packed weights (meaning depends on kind/dtype)
    void* d_pack[8]{};
    size_t pack_bytes[8]{};

    // fp8 mode (vtcnn2): largest |sample| the caller expects (sets the activation scale); fp8_feat_scale_log2 holds the
    // E8M0 block-scale byte the fp8 conv kernel hands its MFMAs (vtcnn2_fp8_conv.hip)
    // Alternate kernels exist only in the -DMDC_ALTERNATES test build (libmdc_alt.so); there mdc_create reads the
    // selecting environment variables ONCE into this field (mdc::kAlt* bits).  Always 0 in the product library: no entry
    // point under mdc_forward* reads the environment.
    int alt = 0;
    int alt_ring = -1;           // alternates build, MDC_DEP_RING=N: ring depth of the 3-filter f32 kernel (0 = direct loads); -1 = the product's choice
    float fp8_input_absmax = 0.02f;
    float fp8_feature_absmax = 0.f;   // vtcnn2 at MDC_FP8, E4M3 features: the caller's measured largest conv2 feature (0 = estimate it from the weights)
    int fp8_feat_scale_log2 = 0;
    bool fp8_e4m3_features = false;   // vtcnn2 at MDC_FP8 without MDC_OPT_FP8_BF16_FEATURES: the workspace features are E4M3 bytes (x 2^feat_scale_log2)
    float fp8_feat_divisor = 1.f;     // ... and this is what the conv kernel's finish divides its sums (true x 2^-32) by before rounding them
    int feat_scale_log2 = 0;     // 16-bit modes: the workspace features are the true ones times 2^feat_scale_log2 (taps undo it)

    // profiling is the one piece of state mdc_forward touches on a finalized model: the event lists are guarded, so
    // forwards of one model from several host threads stay safe with profiling on
    bool profiling = false;
    std::vector<mdc::ProfSlot> slots;
    std::mutex prof_mu;

    // host-buffer driver (host_path.hip): streams, pinned ring and device slots, created on first use; calls are serialised
    void* host_ctx = nullptr;
    std::mutex host_mu;
};

namespace mdc {

// Bracket a launch with events when profiling is on.  The start event lives in the scope and the (start, stop) pair
// is appended under the model's mutex, so concurrent forwards of one model never interleave their pairs.
struct ProfScope {
    mdc_model* m;
    int slot;
    hipStream_t s;
    hipEvent_t start = nullptr;
    ProfScope(const mdc_model* mm, int slot_, hipStream_t s_);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};

int upload(mdc_model* m, int idx, const void* host, size_t bytes);

// Device guard: the entry points run on the model's device and leave the caller's current device as they found it.
struct DeviceScope {
    int prev = -1, want;
    bool ok = true;
    explicit DeviceScope(int device) : want(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want && hipSetDevice(want) != hipSuccess) ok = false;
    }
    ~DeviceScope() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// ---- host-buffer driver: host_path.hip ----------------------------------------------------
int predict_host(mdc_model* m, const float* x_host, int64_t n, float* probs_host, int32_t* labels_host, int64_t chunk_frames);
int predict_host_iq_u8(mdc_model* m, const uint8_t* iq_host, int64_t n, int64_t hop, float scale, float* probs_host, int32_t* labels_host,
                       int64_t chunk_frames);
void host_ctx_free(mdc_model* m);

// ---- deployed (T1/T2): deployed.hip -------------------------------------------------
int deployed_pack(mdc_model* m);
// f32 with the dense layer on the f32 matrix pipe (measured slower: alternates build only): deployed_f32m.hip
int deployed_f32m_pack(mdc_model* m);
int deployed_f32m_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels, float* tap_dense, hipStream_t s);
int deployed_f32m_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, long hop2, float scale, float* probs, int32_t* labels, hipStream_t s);
// bf16 mode (dense layer on the matrix cores, lane = frame): deployed_bf16.hip
int deployed_bf16_pack(mdc_model* m);
int deployed_bf16_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels, hipStream_t s);
int deployed_bf16_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, int64_t hop, float scale, float* probs, int32_t* labels, hipStream_t s);
// Q6.12 integer path of the deployed nets: deployed_q612.hip
int deployed_q612_pack(mdc_model* m);
int deployed_q612_forward(const mdc_model* m, const void* x, int x_is_q, int64_t n, int32_t* dense, int32_t* labels, hipStream_t s);
// fp8 mode of the canonical VT-CNN2: vtcnn2_fp8_conv.hip
int vtcnn2_fp8_pack(mdc_model* m);
int vtcnn2_fp8_conv(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s, long hop2 = 0, float scale = 0.f);
// eval_ops.hip
int confusion_launch(const int32_t* truth, const int32_t* pred, const int32_t* bin, int64_t n, int classes, int bins, int64_t* counts,
                     int64_t* bad, hipStream_t s);
int crossentropy_launch(const float* probs, const int32_t* truth, int64_t n, int classes, double* loss_sum, int64_t* bad, hipStream_t s);
int iq_u8_launch(const uint8_t* iq, int64_t n, int64_t hop, float scale, float* x, hipStream_t s);
int deployed_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                     float* tap, int tap_kind, hipStream_t s);
int deployed_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, int64_t hop, float scale, float* probs, int32_t* labels, hipStream_t s);

// ---- cnn.py literal model (T4): cnnpy.hip ----------------------------------------------
int cnnpy_pack(mdc_model* m);
int cnnpy_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                  float* tap, int tap_kind, hipStream_t s);

// ---- canonical VT-CNN2 (T3): vtcnn2_*.hip ------------------------------------------------
int vtcnn2_pack(mdc_model* m);
size_t vtcnn2_workspace_bytes(const mdc_model* m, int64_t n);
int vtcnn2_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                   float* tap, int tap_kind, void* ws, size_t ws_bytes, hipStream_t s);
// raw uint8 I/Q windows (window i = the 128 (I,Q) pairs from pair i*hop on) straight into the conv kernels' staging
int vtcnn2_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, int64_t hop, float scale, float* probs, int32_t* labels,
                         void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace mdc
