// Canonical VT-CNN2 (T3), bf16 path: dense1 (10560 -> 256, bias + ReLU) as a tiled MFMA GEMM.
//
// vt_dense1_bf16_kernel -- 256x256x64-tile bf16 GEMM (M = frames, N = 256 hidden units,
// K = 10560), LDS-DMA staging with an XOR-swizzled source (so ds_read_b128 fragments spread
// over the banks), two LDS buffers, 8 waves (2 x 4), fused bias + ReLU epilogue.  48 % MFMA-busy;
// SQ_WAIT_ANY is 44 % of its wave cycles.  (Tried and dropped: a ring of four 32-deep stages with a
// counted vmcnt(8) -- 6 % slower: the waits are the per-k-step LDS fragment reads and the barrier, not HBM
// latency; the next step for this kernel is fragment prefetch into registers / the 8-phase schedule.)
#include "vtcnn2_bf16_common.h"

#include <cstdlib>

namespace mdc {

namespace {

// ------------------------------------------------------------------------------------
// dense1 bf16 GEMM
// ------------------------------------------------------------------------------------
constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kTileBytes = kBM * kBK * 2;                     // 32 KiB per operand tile
constexpr size_t kDenseBf16Lds = (size_t)4 * kTileBytes;      // A,B x 2 buffers = 128 KiB
constexpr int kNT = kFeat / kBK;                              // 165 K-tiles


template <int ABL>   // 0 = product; 1..3 = timing-only probes (tools/ablate_dense1.py, -DMDC_ABLATIONS; results wrong)
__global__ __launch_bounds__(512) void vt_dense1_bf16_kernel(const unsigned short* __restrict__ feat, long n,
                                                             const unsigned short* __restrict__ w1t,   // [165 k-tiles][256][64] bf16
                                                             const float* __restrict__ c1,
                                                             float* __restrict__ hid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const long row0 = (long)blockIdx.x * kBM;

    // staging: each wave moves 4 pieces (8 rows x 128 B) of A and 4 of B per K-tile.  LDS is linear
    // (row*128 + pos*16); the SOURCE chunk is pos ^ (row & 7), and readers apply the same XOR.
    const int srow = lane >> 3, spos = lane & 7;
    const unsigned short* asrc[4];
    const unsigned short* bsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = (wv * 4 + p) * 8 + srow;              // tile row 0..255
        long gr = (ABL == 1 ? 0 : row0) + r;                // probe 1: every work-group streams the same rows (L2 hits)
        if (gr >= n) gr = n - 1;                            // clamp: rows past the end are computed, not stored
        asrc[p] = feat + gr * (long)kFeat + ((spos ^ (r & 7)) * 8);
        bsrc[p] = w1t + r * kBK + ((spos ^ (r & 7)) * 8);
    }
    auto stage = [&](int t, int b) {
        if (ABL == 3 && t > 1) return;                      // probe 3: no staging traffic (compute on stale tiles)
        unsigned char* A = smem + (size_t)b * 2 * kTileBytes;
        unsigned char* B = A + kTileBytes;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            glds16(asrc[p] + t * kBK, A + (wv * 4 + p) * 1024);
            glds16(bsrc[p] + (long)t * (kBN * kBK), B + (wv * 4 + p) * 1024);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int b) {
        const unsigned char* A = smem + (size_t)b * 2 * kTileBytes;
        const unsigned char* B = A + kTileBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = wr * 128 + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(A + r * 128 + (((ks * 4 + fg) ^ (r & 7)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wc * 64 + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(B + r * 128 + (((ks * 4 + fg) ^ (r & 7)) * 16));
            }
            if (ABL == 2) {                                  // probe 2: staging + fragment reads, no MFMA
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(bfr[j]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    stage(0, 0);
    __syncthreads();                 // drains the LDS-DMA (vmcnt(0)) and orders it for every wave
    int cur = 0;
    for (int t = 0; t < kNT - 1; ++t) {
        stage(t + 1, cur ^ 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wc * 64 + j * 16 + fr;
        const float bias = c1[col];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + wr * 128 + i * 16 + fg * 4 + r;
                if (row < n) hid[row * kHid + col] = fmaxf(acc[i][j][r] + bias, 0.f);
            }
    }
}

}  // namespace

int vtcnn2_bf16_dense1(const mdc_model* m, const void* feat, int64_t n, float* hid, hipStream_t s) {
#define MDC_LAUNCH_D1(A) do { \
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseBf16Lds)); \
    hipLaunchKernelGGL(vt_dense1_bf16_kernel<A>, dim3((unsigned)((n + kBM - 1) / kBM)), dim3(512), kDenseBf16Lds, s, \
                       static_cast<const unsigned short*>(feat), (long)n, static_cast<const unsigned short*>(m->d_pack[3]), \
                       static_cast<const float*>(m->d_pack[4]), hid); } while (0)
#ifdef MDC_ABLATIONS
    static const int abl = getenv("MDC_ABLATE_D1") ? atoi(getenv("MDC_ABLATE_D1")) : 0;
    switch (abl) { case 1: MDC_LAUNCH_D1(1); break; case 2: MDC_LAUNCH_D1(2); break; case 3: MDC_LAUNCH_D1(3); break; default: MDC_LAUNCH_D1(0); }
#else
    MDC_LAUNCH_D1(0);
#endif
#undef MDC_LAUNCH_D1
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
