// C-ABI shim of libmdc.so (see include/mdc.h for the contract and the reference call
// each entry point replaces).  No torch types, no exceptions across the boundary.
#include "mdc_internal.h"

#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>

namespace mdc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int upload(mdc_model* m, int idx, const void* host, size_t bytes) {
    if (m->d_pack[idx]) {
        (void)hipFree(m->d_pack[idx]);
        m->d_pack[idx] = nullptr;
    }
    MDC_HIP(hipMalloc(&m->d_pack[idx], bytes));
    MDC_HIP(hipMemcpy(m->d_pack[idx], host, bytes, hipMemcpyHostToDevice));
    m->pack_bytes[idx] = bytes;
    return MDC_OK;
}

ProfScope::ProfScope(const mdc_model* mm, int slot_, hipStream_t s_) : m(const_cast<mdc_model*>(mm)), slot(slot_), s(s_) {
    if (!m->profiling) return;
    if (hipEventCreate(&start) != hipSuccess) { start = nullptr; return; }
    (void)hipEventRecord(start, s);
}
ProfScope::~ProfScope() {
    if (!start) return;
    hipEvent_t stop = nullptr;
    if (hipEventCreate(&stop) != hipSuccess) { (void)hipEventDestroy(start); return; }
    (void)hipEventRecord(stop, s);
    try {
        std::lock_guard<std::mutex> g(m->prof_mu);
        std::vector<hipEvent_t>& ev = m->slots[slot].ev;
        ev.reserve(ev.size() + 2);      // both or neither
        ev.push_back(start);
        ev.push_back(stop);
    } catch (...) {                     // out of host memory: drop the sample, never the process
        (void)hipEventDestroy(start);
        (void)hipEventDestroy(stop);
    }
}

}  // namespace mdc

using namespace mdc;

static int layer_layout(mdc_model* m) {
    const mdc_topology& t = m->topo;
    switch (t.kind) {
        case MDC_KIND_DEPLOYED:
            if (t.filters != 3 && t.filters != 10) { set_error("deployed: filters must be 3 or 10 (got %d)", t.filters); return MDC_ENOTSUP; }
            if (t.classes != 3) { set_error("deployed: classes must be 3 (got %d)", t.classes); return MDC_ENOTSUP; }
            m->nlayers = 2;
            m->nk[0] = 2 * (size_t)t.filters;            m->nb[0] = t.filters;
            m->nk[1] = 258 * (size_t)t.filters * t.classes; m->nb[1] = t.classes;
            m->slots = {{"mdc_deployed_fwd"}};
            return MDC_OK;
        case MDC_KIND_VTCNN2:
            if (t.classes < 2 || t.classes > 16) { set_error("vtcnn2: classes must be in 2..16 (got %d)", t.classes); return MDC_ENOTSUP; }
            m->nlayers = 4;
            m->nk[0] = (size_t)kC1 * 3;               m->nb[0] = kC1;
            m->nk[1] = (size_t)kC2 * kC1 * 2 * 3;     m->nb[1] = kC2;
            m->nk[2] = (size_t)kFeat * kHid;          m->nb[2] = kHid;
            m->nk[3] = (size_t)kHid * t.classes;      m->nb[3] = t.classes;
            m->slots = {{"mdc_vt_conv"}, {"mdc_vt_dense1"}, {"mdc_vt_head"}};
            return MDC_OK;
        case MDC_KIND_CNNPY:
            if (t.filters < 1 || t.filters > 10 || t.hidden < 1 || t.hidden > 16 || t.classes < 2 || t.classes > 16) {
                set_error("cnnpy: need filters 1..10, hidden 1..16, classes 2..16 (got %d,%d,%d)", t.filters, t.hidden, t.classes);
                return MDC_ENOTSUP;
            }
            m->nlayers = 3;
            m->nk[0] = 2 * 128 * (size_t)t.filters;        m->nb[0] = t.filters;
            m->nk[1] = 3 * (size_t)t.filters * t.hidden;    m->nb[1] = t.hidden;
            m->nk[2] = (size_t)t.hidden * t.classes;        m->nb[2] = t.classes;
            m->slots = {{"mdc_dense_chain"}};
            return MDC_OK;
        default:
            set_error("unknown topology kind %d", t.kind);
            return MDC_EINVAL;
    }
}

extern "C" {

int mdc_abi_version(void) { return MDC_ABI_VERSION; }

const char* mdc_last_error(void) { return g_err; }

int mdc_create(const mdc_topology* topo, int device, mdc_model** out) {
    return guarded("mdc_create", [&]() -> int {
        if (!topo || !out) { set_error("mdc_create: null argument"); return MDC_EINVAL; }
        *out = nullptr;
        if ((topo->reserved[0] & ~MDC_OPT_ALL) != 0) { set_error("mdc_create: unknown option bits 0x%x in reserved[0]", topo->reserved[0]); return MDC_EINVAL; }
        for (int i = 1; i < 4; ++i) if (topo->reserved[i] != 0) { set_error("mdc_create: reserved[1..3] must be 0"); return MDC_EINVAL; }
        if ((topo->reserved[0] & MDC_OPT_FP8_BF16_FEATURES) && topo->kind != MDC_KIND_VTCNN2) {
            set_error("mdc_create: MDC_OPT_FP8_BF16_FEATURES applies to MDC_KIND_VTCNN2 only");
            return MDC_EINVAL;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available"); return MDC_ENODEV; }
        if (device < 0 || device >= ndev) { set_error("device %d out of range (have %d)", device, ndev); return MDC_ENODEV; }
        hipDeviceProp_t prop;
        MDC_HIP(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_error("device %d is %s; libmdc.so carries gfx950 (MI355X) code only", device, prop.gcnArchName);
            return MDC_ENODEV;
        }
        mdc_model* m = new (std::nothrow) mdc_model();
        if (!m) { set_error("out of host memory"); return MDC_ENOMEM; }
        m->topo = *topo;
        m->device = device;
#ifdef MDC_ALTERNATES      // the test build: which alternate kernels this model runs, read once, here
        {
            auto flag = [](const char* name, int when) { const char* e = getenv(name); return e && atoi(e) == when; };
            if (flag("MDC_CONV_SCHED", 0)) m->alt |= kAltConvHipcc;
            if (flag("MDC_DENSE1_PHASED", 0)) m->alt |= kAltDense1Simple;
            if (flag("MDC_DEP_F32_MFMA", 1)) m->alt |= kAltDepF32Mfma;
            if (flag("MDC_D1_FUSED_HEAD", 0)) m->alt |= kAltSeparateHead;
            if (const char* e = getenv("MDC_DEP_RING")) m->alt_ring = atoi(e);
        }
#endif
        int rc = MDC_ENOMEM;
        try { rc = layer_layout(m); } catch (...) { delete m; throw; }
        if (rc != MDC_OK) { delete m; return rc; }
        *out = m;
        return MDC_OK;
    });
}

int mdc_num_layers(const mdc_model* m) {
    if (!m) { set_error("null model"); return MDC_EINVAL; }
    return m->nlayers;
}

int mdc_layer_sizes(const mdc_model* m, int layer, size_t* kernel_elems, size_t* bias_elems) {
    if (!m || layer < 0 || layer >= m->nlayers) { set_error("mdc_layer_sizes: bad layer %d", layer); return MDC_EINVAL; }
    if (kernel_elems) *kernel_elems = m->nk[layer];
    if (bias_elems) *bias_elems = m->nb[layer];
    return MDC_OK;
}

int mdc_set_weights(mdc_model* m, int layer, const float* kernel_host, size_t kernel_elems,
                    const float* bias_host, size_t bias_elems) {
    return guarded("mdc_set_weights", [&]() -> int {
        if (!m || !kernel_host || !bias_host) { set_error("mdc_set_weights: null argument"); return MDC_EINVAL; }
        if (m->finalized) { set_error("mdc_set_weights: model is finalized (immutable)"); return MDC_ESTATE; }
        if (layer < 0 || layer >= m->nlayers) { set_error("mdc_set_weights: layer %d out of range 0..%d", layer, m->nlayers - 1); return MDC_EINVAL; }
        if (kernel_elems != m->nk[layer] || bias_elems != m->nb[layer]) {
            set_error("mdc_set_weights: layer %d expects kernel %zu / bias %zu elements, got %zu / %zu",
                      layer, m->nk[layer], m->nb[layer], kernel_elems, bias_elems);
            return MDC_EINVAL;
        }
        m->have[layer] = false;
        m->hk[layer].assign(kernel_host, kernel_host + kernel_elems);      // may throw bad_alloc -> MDC_ENOMEM
        m->hb[layer].assign(bias_host, bias_host + bias_elems);
        m->have[layer] = true;
        return MDC_OK;
    });
}

int mdc_finalize(mdc_model* m, int dtype) {
    return guarded("mdc_finalize", [&]() -> int {
        if (!m) { set_error("null model"); return MDC_EINVAL; }
        if (m->finalized) { set_error("mdc_finalize: already finalized"); return MDC_ESTATE; }
        for (int l = 0; l < m->nlayers; ++l)
            if (!m->have[l]) { set_error("mdc_finalize: layer %d has no weights", l); return MDC_ESTATE; }
        if (dtype != MDC_F32 && dtype != MDC_BF16 && dtype != MDC_FP8 && dtype != MDC_F16) { set_error("unknown dtype %d", dtype); return MDC_EINVAL; }
        if (dtype == MDC_F16 && m->topo.kind != MDC_KIND_DEPLOYED) { set_error("f16 is implemented for the deployed nets only"); return MDC_ENOTSUP; }
        if (dtype == MDC_FP8 && m->topo.kind == MDC_KIND_CNNPY) { set_error("fp8 is implemented for the vtcnn2 and deployed families only"); return MDC_ENOTSUP; }
        if ((m->topo.reserved[0] & MDC_OPT_FP8_BF16_FEATURES) && dtype != MDC_FP8) { set_error("mdc_finalize: MDC_OPT_FP8_BF16_FEATURES is an option of the MDC_FP8 mode"); return MDC_EINVAL; }
        if (dtype == MDC_BF16 && m->topo.kind == MDC_KIND_CNNPY) { set_error("bf16 is implemented for the vtcnn2 and deployed families only"); return MDC_ENOTSUP; }
        m->dtype = dtype;
        DeviceScope dev(m->device);      // uploads go to the model's device; the caller's current device is restored
        if (!dev.ok) { set_error("mdc_finalize: cannot select device %d", m->device); return MDC_EIO; }
        int rc;
        switch (m->topo.kind) {
            case MDC_KIND_DEPLOYED:
                rc = deployed_pack(m);
                if (rc == MDC_OK) rc = deployed_q612_pack(m);
                if (rc == MDC_OK && dtype != MDC_F32) rc = deployed_bf16_pack(m);
                break;
            case MDC_KIND_VTCNN2:   rc = vtcnn2_pack(m); break;
            case MDC_KIND_CNNPY:    rc = cnnpy_pack(m); break;
            default: rc = MDC_EINVAL;
        }
        if (rc != MDC_OK) return rc;
        for (int l = 0; l < m->nlayers; ++l) {   // host copies no longer needed
            std::vector<float>().swap(m->hk[l]);
            std::vector<float>().swap(m->hb[l]);
        }
        m->finalized = true;
        return MDC_OK;
    });
}

size_t mdc_workspace_bytes(const mdc_model* m, int64_t n) {
    if (!m || n <= 0) return 0;
    if (m->topo.kind == MDC_KIND_VTCNN2) return vtcnn2_workspace_bytes(m, n);
    return 0;
}

int mdc_forward(const mdc_model* m, const void* x_dev, int64_t n, float* probs_dev, int32_t* labels_dev,
                float* tap_dev, int tap, void* workspace_dev, size_t workspace_bytes, void* hip_stream) {
    return guarded("mdc_forward", [&]() -> int {
        if (!m) { set_error("null model"); return MDC_EINVAL; }
        if (!m->finalized) { set_error("mdc_forward: model not finalized"); return MDC_ESTATE; }
        if (n < 0) { set_error("mdc_forward: negative frame count"); return MDC_EINVAL; }
        if (n == 0) return MDC_OK;
        if (!x_dev) { set_error("mdc_forward: null input"); return MDC_EINVAL; }
        if ((reinterpret_cast<uintptr_t>(x_dev) & 15) != 0) { set_error("mdc_forward: input must be 16-byte aligned"); return MDC_EINVAL; }
        if (tap < MDC_TAP_NONE || tap > MDC_TAP_HIDDEN) { set_error("mdc_forward: bad tap %d", tap); return MDC_EINVAL; }
        if ((tap != MDC_TAP_NONE) != (tap_dev != nullptr)) { set_error("mdc_forward: tap and tap_dev must be given together"); return MDC_EINVAL; }
        DeviceScope dev(m->device);
        if (!dev.ok) { set_error("mdc_forward: cannot select device %d", m->device); return MDC_EIO; }
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        const float* x = static_cast<const float*>(x_dev);
        switch (m->topo.kind) {
            case MDC_KIND_DEPLOYED: return deployed_forward(m, x, n, probs_dev, labels_dev, tap_dev, tap, s);
            case MDC_KIND_VTCNN2:   return vtcnn2_forward(m, x, n, probs_dev, labels_dev, tap_dev, tap, workspace_dev, workspace_bytes, s);
            case MDC_KIND_CNNPY:    return cnnpy_forward(m, x, n, probs_dev, labels_dev, tap_dev, tap, s);
            default: return MDC_EINVAL;
        }
    });
}

int mdc_forward_iq_u8(const mdc_model* m, const uint8_t* iq_dev, int64_t n, int64_t hop, float scale,
                      float* probs_dev, int32_t* labels_dev, void* workspace_dev, size_t workspace_bytes, void* hip_stream) {
    return guarded("mdc_forward_iq_u8", [&]() -> int {
        if (!m) { set_error("null model"); return MDC_EINVAL; }
        if (!m->finalized) { set_error("mdc_forward_iq_u8: model not finalized"); return MDC_ESTATE; }
        if (m->topo.kind == MDC_KIND_CNNPY) {
            set_error("mdc_forward_iq_u8: raw-IQ input exists for the deployed and vtcnn2 families; use mdc_iq_u8_to_frames + mdc_forward");
            return MDC_ENOTSUP;
        }
        if (n < 0) { set_error("mdc_forward_iq_u8: negative window count"); return MDC_EINVAL; }
        if (hop < 1 || hop > (int64_t)1 << 24) { set_error("mdc_forward_iq_u8: hop must be in 1..2^24 sample pairs (got %lld)", (long long)hop); return MDC_EINVAL; }
        if (n == 0) return MDC_OK;
        if (!iq_dev) { set_error("mdc_forward_iq_u8: null input"); return MDC_EINVAL; }
        if ((reinterpret_cast<uintptr_t>(iq_dev) & 1) != 0) { set_error("mdc_forward_iq_u8: input must start on a whole (I,Q) pair (2-byte aligned)"); return MDC_EINVAL; }
        DeviceScope dev(m->device);
        if (!dev.ok) { set_error("mdc_forward_iq_u8: cannot select device %d", m->device); return MDC_EIO; }
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        if (m->topo.kind == MDC_KIND_VTCNN2)
            return vtcnn2_forward_iq_u8(m, iq_dev, n, hop, scale, probs_dev, labels_dev, workspace_dev, workspace_bytes, s);
        return (m->dtype == MDC_F32) ? deployed_forward_iq_u8(m, iq_dev, n, hop, scale, probs_dev, labels_dev, s)
                                     : deployed_bf16_forward_iq_u8(m, iq_dev, n, hop, scale, probs_dev, labels_dev, s);
    });
}

int mdc_set_fp8_input_absmax(mdc_model* m, float absmax) {
    if (!m) { set_error("null model"); return MDC_EINVAL; }
    if (m->finalized) { set_error("mdc_set_fp8_input_absmax: call it before mdc_finalize"); return MDC_ESTATE; }
    if (!(absmax > 0.f) || !(absmax < 1e30f)) { set_error("mdc_set_fp8_input_absmax: need a positive finite value"); return MDC_EINVAL; }
    m->fp8_input_absmax = absmax;
    return MDC_OK;
}

int mdc_set_fp8_feature_absmax(mdc_model* m, float absmax) {
    if (!m) { set_error("null model"); return MDC_EINVAL; }
    if (m->finalized) { set_error("mdc_set_fp8_feature_absmax: call it before mdc_finalize"); return MDC_ESTATE; }
    if (m->topo.kind != MDC_KIND_VTCNN2) { set_error("mdc_set_fp8_feature_absmax: the E4M3 features exist for MDC_KIND_VTCNN2 only"); return MDC_EINVAL; }
    if (!(absmax > 0.f) || !(absmax < 1e30f)) { set_error("mdc_set_fp8_feature_absmax: need a positive finite value"); return MDC_EINVAL; }
    m->fp8_feature_absmax = absmax;
    return MDC_OK;
}

int mdc_forward_q612(const mdc_model* m, const void* x_dev, int x_is_q612, int64_t n, int32_t* dense_dev, int32_t* labels_dev,
                     void* hip_stream) {
    return guarded("mdc_forward_q612", [&]() -> int {
        if (!m) { set_error("null model"); return MDC_EINVAL; }
        if (!m->finalized) { set_error("mdc_forward_q612: model not finalized"); return MDC_ESTATE; }
        if (m->topo.kind != MDC_KIND_DEPLOYED) { set_error("mdc_forward_q612: the FPGA datapath exists for the deployed nets only"); return MDC_ENOTSUP; }
        if (n < 0) { set_error("mdc_forward_q612: negative frame count"); return MDC_EINVAL; }
        if (n == 0) return MDC_OK;
        if (!x_dev) { set_error("mdc_forward_q612: null input"); return MDC_EINVAL; }
        if ((reinterpret_cast<uintptr_t>(x_dev) & 15) != 0) { set_error("mdc_forward_q612: input must be 16-byte aligned"); return MDC_EINVAL; }
        DeviceScope dev(m->device);
        if (!dev.ok) { set_error("mdc_forward_q612: cannot select device %d", m->device); return MDC_EIO; }
        return deployed_q612_forward(m, x_dev, x_is_q612 != 0, n, dense_dev, labels_dev, static_cast<hipStream_t>(hip_stream));
    });
}

int mdc_confusion(const int32_t* truth_dev, const int32_t* pred_dev, int64_t n, int classes, int64_t* counts_dev, int64_t* bad_dev,
                  void* hip_stream) {
    if (n < 0) { set_error("mdc_confusion: negative count"); return MDC_EINVAL; }
    if (n > 0 && (!truth_dev || !pred_dev)) { set_error("mdc_confusion: null labels"); return MDC_EINVAL; }
    if (!counts_dev) { set_error("mdc_confusion: null counts"); return MDC_EINVAL; }
    return guarded("mdc_confusion", [&]() -> int {
        return confusion_launch(truth_dev, pred_dev, nullptr, n, classes, 1, counts_dev, bad_dev, static_cast<hipStream_t>(hip_stream)); });
}

int mdc_confusion_binned(const int32_t* truth_dev, const int32_t* pred_dev, const int32_t* bin_dev, int64_t n, int classes, int bins,
                         int64_t* counts_dev, int64_t* bad_dev, void* hip_stream) {
    if (n < 0) { set_error("mdc_confusion_binned: negative count"); return MDC_EINVAL; }
    if (n > 0 && (!truth_dev || !pred_dev || !bin_dev)) { set_error("mdc_confusion_binned: null labels"); return MDC_EINVAL; }
    if (!counts_dev) { set_error("mdc_confusion_binned: null counts"); return MDC_EINVAL; }
    if (bins < 1) { set_error("mdc_confusion_binned: bins must be >= 1 (got %d)", bins); return MDC_EINVAL; }
    return guarded("mdc_confusion_binned", [&]() -> int {
        return confusion_launch(truth_dev, pred_dev, bin_dev, n, classes, bins, counts_dev, bad_dev, static_cast<hipStream_t>(hip_stream)); });
}

int mdc_crossentropy(const float* probs_dev, const int32_t* truth_dev, int64_t n, int classes, double* loss_sum_dev, int64_t* bad_dev,
                     void* hip_stream) {
    if (n < 0) { set_error("mdc_crossentropy: negative count"); return MDC_EINVAL; }
    if (n > 0 && (!probs_dev || !truth_dev)) { set_error("mdc_crossentropy: null probabilities or labels"); return MDC_EINVAL; }
    if (!loss_sum_dev) { set_error("mdc_crossentropy: null loss accumulator"); return MDC_EINVAL; }
    return guarded("mdc_crossentropy", [&]() -> int {
        return crossentropy_launch(probs_dev, truth_dev, n, classes, loss_sum_dev, bad_dev, static_cast<hipStream_t>(hip_stream)); });
}

int mdc_iq_u8_to_frames(const uint8_t* iq_dev, int64_t n, float scale, float* x_dev, void* hip_stream) {
    if (n < 0) { set_error("mdc_iq_u8_to_frames: negative frame count"); return MDC_EINVAL; }
    if (n == 0) return MDC_OK;
    if (!iq_dev || !x_dev) { set_error("mdc_iq_u8_to_frames: null buffer"); return MDC_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(iq_dev) & 3) != 0 || (reinterpret_cast<uintptr_t>(x_dev) & 7) != 0) {
        set_error("mdc_iq_u8_to_frames: iq must be 4-byte and frames 8-byte aligned");
        return MDC_EINVAL;
    }
    return guarded("mdc_iq_u8_to_frames", [&]() -> int { return iq_u8_launch(iq_dev, n, 128, scale, x_dev, static_cast<hipStream_t>(hip_stream)); });
}

int mdc_iq_u8_windows(const uint8_t* iq_dev, int64_t n, int64_t hop, float scale, float* x_dev, void* hip_stream) {
    if (n < 0) { set_error("mdc_iq_u8_windows: negative window count"); return MDC_EINVAL; }
    if (hop < 1 || hop > (int64_t)1 << 24) { set_error("mdc_iq_u8_windows: hop must be in 1..2^24 sample pairs (got %lld)", (long long)hop); return MDC_EINVAL; }
    if (n == 0) return MDC_OK;
    if (!iq_dev || !x_dev) { set_error("mdc_iq_u8_windows: null buffer"); return MDC_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(x_dev) & 7) != 0) { set_error("mdc_iq_u8_windows: frames must be 8-byte aligned"); return MDC_EINVAL; }
    // as mdc_forward_iq_u8 (whose result this call + mdc_forward must equal): whole (I,Q) pairs, i.e. 2-byte alignment
    if ((reinterpret_cast<uintptr_t>(iq_dev) & 1) != 0) { set_error("mdc_iq_u8_windows: input must start on a whole (I,Q) pair (2-byte aligned)"); return MDC_EINVAL; }
    return guarded("mdc_iq_u8_windows", [&]() -> int { return iq_u8_launch(iq_dev, n, hop, scale, x_dev, static_cast<hipStream_t>(hip_stream)); });
}

int mdc_set_profiling(mdc_model* m, int on) {
    if (!m) { set_error("null model"); return MDC_EINVAL; }
    m->profiling = on != 0;
    return MDC_OK;
}

int mdc_profile_slots(const mdc_model* m) { return m ? (int)m->slots.size() : MDC_EINVAL; }

const char* mdc_profile_name(const mdc_model* m, int slot) {
    if (!m || slot < 0 || slot >= (int)m->slots.size()) return "";
    return m->slots[slot].name;
}

int mdc_profile_read(mdc_model* m, int slot, double* total_ms, int64_t* launches) {
    return guarded("mdc_profile_read", [&]() -> int {
        if (!m || slot < 0 || slot >= (int)m->slots.size()) { set_error("mdc_profile_read: bad slot"); return MDC_EINVAL; }
        ProfSlot& ps = m->slots[slot];
        std::vector<hipEvent_t> ev;
        {
            std::lock_guard<std::mutex> g(m->prof_mu);
            ev.swap(ps.ev);
        }
        double add_ms = 0;
        int64_t add_n = 0;
        int rc = MDC_OK;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float ms = 0.f;
            if (hipEventSynchronize(ev[i + 1]) != hipSuccess || hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) {
                set_error("mdc_profile_read: event query failed");
                rc = MDC_EIO;
                break;
            }
            add_ms += ms;
            add_n += 1;
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        std::lock_guard<std::mutex> g(m->prof_mu);
        ps.total_ms += add_ms;
        ps.launches += add_n;
        if (total_ms) *total_ms = ps.total_ms;
        if (launches) *launches = ps.launches;
        return rc;
    });
}

int mdc_profile_reset(mdc_model* m) {
    if (!m) { set_error("null model"); return MDC_EINVAL; }
    std::lock_guard<std::mutex> g(m->prof_mu);
    for (ProfSlot& ps : m->slots) {
        for (hipEvent_t e : ps.ev) (void)hipEventDestroy(e);
        ps.ev.clear();
        ps.total_ms = 0;
        ps.launches = 0;
    }
    return MDC_OK;
}

int mdc_predict_host(mdc_model* m, const float* x_host, int64_t n, float* probs_host, int32_t* labels_host, int64_t chunk_frames) {
    return guarded("mdc_predict_host", [&]() -> int { return predict_host(m, x_host, n, probs_host, labels_host, chunk_frames); });
}

int mdc_predict_host_iq_u8(mdc_model* m, const uint8_t* iq_host, int64_t n, int64_t hop, float scale, float* probs_host,
                           int32_t* labels_host, int64_t chunk_frames) {
    return guarded("mdc_predict_host_iq_u8", [&]() -> int { return predict_host_iq_u8(m, iq_host, n, hop, scale, probs_host, labels_host, chunk_frames); });
}

void mdc_destroy(mdc_model* m) {
    if (!m) return;
    {
        DeviceScope dev(m->device);
        host_ctx_free(m);
    }
    for (void*& p : m->d_pack) if (p) { (void)hipFree(p); p = nullptr; }
    for (ProfSlot& ps : m->slots) for (hipEvent_t e : ps.ev) (void)hipEventDestroy(e);
    delete m;
}

}  // extern "C"
