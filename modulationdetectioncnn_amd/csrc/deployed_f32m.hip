// Deployed single-conv nets (T1: F=3, T2: F=10) at the reference's own precision (f32 in, f32 arithmetic, f32 out)
// with the Dense(3) layer on the f32 matrix pipe.
//
// Math restated from CNN.ipynb cell 6 / the model_config inside the bundled .h5:
//   y[h,w,f] = relu(b[f] + K0[f]*x[h,w-1] + K1[f]*x[h,w]),  w = 0..128,  x[h,-1] = x[h,128] = 0
//   z[c]     = relu(bd[c] + sum_{h,w,f} Wd[h*129F + w*F + f][c] * y[h,w,f]);  p = softmax(z);  label = first argmax
//
// Why a second f32 kernel: in deployed.hip every lane multiplies ITS conv outputs by ITS dense weights on the vector
// ALU -- 3 FMAs per conv output on top of the 2 FMAs + 1 max that produce it, which for F = 10 (2,580 outputs x 3
// classes) made the kernel VALU-bound at 0.29 of the HBM roofline.  v_mfma_f32_4x4x1_16B_f32 does those 3 FMAs for 64
// (frame, output) pairs in one 8-cycle instruction on the OTHER pipe, with f32 operands and an exact f32 fma chain per
// accumulator (MI355X_MICROARCH.md, Matrix cores: "exact f32 (= fmaf chain, bitwise)"), so nothing is narrower than
// the reference's arithmetic.  The instruction is 16 independent 4x4x1 outer products: block B of lanes 4B..4B+3 has
//   D_B[i][j] += A_B[i] * B_B[j],   A from lane 4B+i, B from lane 4B+j, D_B[i][j] in register i of lane 4B+j.
// Mapping: a wave takes FOUR frames at a time; lane (B, j) = (position block, frame).  Block B = 8h + g owns positions
// w = 16g .. 16g+15 of row h (and, for g = 7, the 129th position w = 128): j indexes the frame, i the class (3 of the 4
// rows; the 4th row's weights are zero).  For each of its 17*F (position, filter) slots the lane computes ONE conv
// output of its frame with the f32 kernel's fma chain -- that register is the MFMA's B operand -- and its A register
// holds the dense weight W[h][w][f][class = lane&3] of the slot: 17F weight registers per lane (51 / 170), loaded once
// per wave.  After the slots, register i of lane (B, j) holds class i's partial sum of frame j over block B's
// positions; the 16 blocks are summed with two DPP adds inside each 16-lane row and two permlane swaps across the
// rows (same operand order in every lane that keeps a total: results do not depend on where a frame sits in a batch).
// Frames reach the lanes through a wave-private LDS ring filled by LDS-DMA: one global_load_lds_dwordx4 = one whole
// frame (1 KiB contiguous), four per group, three groups in flight while one is computed; a lane reads the 64 bytes of
// its 16 samples (+ the sample before them) with ds_read_b128 (frame stride 1,040 B: conflict-free).
// Results of 16 consecutive groups are parked in a 1 KiB LDS table per wave, then all 64 lanes finish one frame each
// (bias, ReLU, softmax, first-max argmax) and the wave writes 768 B of probabilities + 256 B of labels coalesced.
// HBM traffic = the algorithmic 1,036 B/frame (1,024 in + 12 out) + 4 B label.
//
// U8 = true (mdc_forward_iq_u8): x points at raw interleaved unsigned 8-bit (I,Q) pairs, window f at byte f*hop2; a
// frame is then 256 B (16 lanes of one DMA instruction), a lane reads the 32 bytes that hold its 16 samples of BOTH rows
// and converts the row it owns with the arithmetic of iq_u8_kernel (eval_ops.hip): bit-identical to convert-then-forward.
#ifdef MDC_ALTERNATES      // measured slower than the all-VALU kernel (HISTORY.md 4.1c): test build only, not in libmdc.so
#include "vtcnn2_bf16_common.h"

#include <cstdlib>
#include <vector>

namespace mdc {

namespace {

constexpr int kC = 3;

template <int F, bool U8, int RING = 0>
struct F32mGeom {
    static constexpr int kSlots = 17 * F;                                   // (position 0..16, filter) per lane
    static constexpr int kFrameStride = U8 ? 256 + 16 : 1024 + 16;          // LDS bytes per staged frame
    static constexpr int kGroupBytes = 4 * kFrameStride;
    static constexpr int kRing = RING ? RING : (U8 ? 8 : 4);                // groups in the ring (kRing-1 in flight)
    static constexpr int kWaves = 8;
    static constexpr int kTabBytes = 1024;                                  // 64 frames x (3 sums + pad) f32
    static constexpr size_t kLds = (size_t)kWaves * (kRing * kGroupBytes + kTabBytes);
};

// sum over the four 16-lane rows; every row ends up with the same bits (both swaps pair the SAME operands in the
// same order in both halves)
__device__ __forceinline__ float rows_total(float m) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add_zero(float v) {      // v + (v moved by CTRL; lanes without a source add 0)
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}

// ABL (timing probes only, -DMDC_ABLATIONS; results wrong): 1 no MFMA (conv only), 2 no conv VALU (MFMA on raw samples),
// 4 no block reduction, 8 no DMA (ring never refilled)
template <int F, int TAP, bool U8, int ABL = 0, int RING = 0>
__global__ __launch_bounds__(512, 2) void deployed_f32m_kernel(const float* __restrict__ x, long n,
                                                               const float* __restrict__ wp, const float* __restrict__ atab,
                                                               float* __restrict__ probs, int* __restrict__ labels,
                                                               float* __restrict__ tap_dense, float scale, long hop2,
                                                               int run) {      // groups per run (1..16): see the launcher
    using G = F32mGeom<F, U8, RING>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, B = lane >> 2, g = B & 7, h = B >> 3;
    unsigned char* stage = smem + wv * (G::kRing * G::kGroupBytes + G::kTabBytes);
    float* ztab = reinterpret_cast<float*>(stage + G::kRing * G::kGroupBytes);

    float k0[F], k1[F], cb[F], bd[kC];
#pragma unroll
    for (int f = 0; f < F; ++f) { k0[f] = wp[3 * f + 0]; k1[f] = wp[3 * f + 1]; cb[f] = wp[3 * f + 2]; }
#pragma unroll
    for (int c = 0; c < kC; ++c) bd[c] = wp[3 * F + c];
    float A[G::kSlots];
#pragma unroll
    for (int s = 0; s < G::kSlots; ++s) A[s] = atab[s * 64 + lane];

    // Work split: a RUN is `run` consecutive groups (4*run consecutive frames) owned by one wave; runs are dealt
    // round-robin to the waves of the grid.  run = 16 for a large batch (64 frames finished at a time, fully
    // coalesced stores); small batches use shorter runs so that the frames spread over the chip.
    const long ngroups = (n + 3) >> 2;
    const long nruns = (ngroups + run - 1) / run;
    const long wave_id = (long)blockIdx.x * G::kWaves + wv, nwaves = (long)gridDim.x * G::kWaves;
    const long my_runs = wave_id < nruns ? (nruns - wave_id + nwaves - 1) / nwaves : 0;
    const long my_groups = my_runs * run;                 // group slots this wave walks (slots past the batch are clamped)
    auto group_of = [&](long t) -> long { return (wave_id + (t / run) * nwaves) * run + (t % run); };

    auto stage_group = [&](long t) {      // group slot t of this wave -> ring slot t % kRing (4 DMA instructions)
        long grp = group_of(t < my_groups ? t : my_groups - 1);
        unsigned char* dst = stage + (int)(t % G::kRing) * G::kGroupBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long fr = grp * 4 + i;
            fr = fr < n ? fr : n - 1;      // frames past the end re-read the last one (never stored)
            if constexpr (U8) {
                if (lane < 16) glds16_async(reinterpret_cast<const unsigned char*>(x) + fr * hop2 + lane * 16, dst + i * G::kFrameStride);
            } else {
                glds16_async(x + fr * kFrameFloats + lane * 4, dst + i * G::kFrameStride);
            }
        }
    };
    if (my_groups > 0 && !(ABL & 8)) {
#pragma unroll
        for (int t = 0; t < G::kRing - 1; ++t) stage_group(t);
    }
    const unsigned row_shift = 8 * h;      // raw bytes: row 0 = I = even bytes, row 1 = Q = odd bytes

    for (long t = 0; t < my_groups; ++t) {
        // the group kRing-1 steps ahead goes into the slot whose last reader finished a step ago (lgkmcnt(0) below);
        // then at most those groups' 4*(kRing-1) DMA instructions may be outstanding: this group has landed
        if (!(ABL & 8)) {
            stage_group(t + G::kRing - 1);
            if constexpr (G::kRing == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (G::kRing == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (G::kRing == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (G::kRing == 6) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        }
        const unsigned char* src = stage + (int)(t % G::kRing) * G::kGroupBytes + j * G::kFrameStride;
        float s[18];                       // s[0] = x[16g-1] (0 at the row start), s[1..16] = x[16g .. 16g+15], s[17] = 0
        if constexpr (U8) {
            const uint4 r0 = *reinterpret_cast<const uint4*>(src + g * 32), r1 = *reinterpret_cast<const uint4*>(src + g * 32 + 16);
            const unsigned pv = *reinterpret_cast<const unsigned short*>(src + (g ? g * 32 - 2 : 0));
            const unsigned w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned d = w[i] >> row_shift;
                s[1 + 2 * i] = ((float)(d & 0xFFu) - 127.5f) * scale;
                s[2 + 2 * i] = ((float)((d >> 16) & 0xFFu) - 127.5f) * scale;
            }
            s[0] = g ? ((float)((pv >> row_shift) & 0xFFu) - 127.5f) * scale : 0.f;
        } else {
            const float4* q = reinterpret_cast<const float4*>(src + B * 64);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const float pv = *reinterpret_cast<const float*>(src + B * 64 - (g ? 4 : 0));
            s[0] = g ? pv : 0.f;
            s[1] = q0.x; s[2] = q0.y; s[3] = q0.z; s[4] = q0.w; s[5] = q1.x; s[6] = q1.y; s[7] = q1.z; s[8] = q1.w;
            s[9] = q2.x; s[10] = q2.y; s[11] = q2.z; s[12] = q2.w; s[13] = q3.x; s[14] = q3.y; s[15] = q3.z; s[16] = q3.w;
        }
        s[17] = 0.f;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int p = 0; p < 17; ++p)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int slot = p * F + f;
                float y;
                if (ABL & 2) y = s[p + 1];
                else y = fmaxf(fmaf(k1[f], s[p + 1], fmaf(k0[f], s[p], cb[f])), 0.f);
                if (ABL & 1) acc[slot & 1][f & 3] += y * A[slot];
                else acc[slot & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[slot], y, acc[slot & 1], 0, 0, 0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of this ring slot is done
        // ---- sum over the 16 position blocks: inside a row (blocks B&3) towards lanes 12..15, then across the rows
        const int tr = (int)(t % run);
        float tot[kC];
#pragma unroll
        for (int c = 0; c < kC; ++c) {
            float v = acc[0][c] + acc[1][c];
            if (!(ABL & 4)) {
                v = dpp_add_zero<0x114>(v);      // row_shr:4
                v = dpp_add_zero<0x118>(v);      // row_shr:8  -> lanes 12..15 of each row hold the row's sum for frame j
                v = rows_total(v);
            }
            tot[c] = v;
        }
        if (lane >= 12 && lane < 16) *reinterpret_cast<float4*>(ztab + (tr * 4 + j) * 4) = make_float4(tot[0], tot[1], tot[2], 0.f);
        if (tr == run - 1) {
            // ---- all lanes: finish frame `lane` of the run (bias, ReLU, softmax, first-max argmax)
            const float4 zz = *reinterpret_cast<const float4*>(ztab + lane * 4);
            const long o = group_of(t - tr) * 4 + lane;
            if (lane < 4 * run && o < n) {
                const float z0 = fmaxf(zz.x + bd[0], 0.f);   // Dense(3, activation='relu')
                const float z1 = fmaxf(zz.y + bd[1], 0.f);
                const float z2 = fmaxf(zz.z + bd[2], 0.f);
                const float mx = fmaxf(z0, fmaxf(z1, z2));
                const float e0 = expf(z0 - mx), e1 = expf(z1 - mx), e2 = expf(z2 - mx);
                const float inv = 1.0f / (e0 + e1 + e2);
                const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
                if (probs) {
                    probs[o * 3 + 0] = p0;
                    probs[o * 3 + 1] = p1;
                    probs[o * 3 + 2] = p2;
                }
                // int(np.argmax(test_Y_hat[i,:])) (cnn.py:209): FIRST maximum of the probabilities as returned
                if (labels) labels[o] = (p0 >= p1 && p0 >= p2) ? 0 : ((p1 >= p2) ? 1 : 2);
                if (TAP == 2) {
                    tap_dense[o * 3 + 0] = z0;
                    tap_dense[o * 3 + 1] = z1;
                    tap_dense[o * 3 + 2] = z2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the table is free for the next run
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the wave's LDS allocation
}

}  // namespace

// d_pack slot 3 of a deployed model: the dense layer as per-lane MFMA A operands, atab[slot = p*F + f][lane]:
// lane (B = lane>>2, i = lane&3), B = 8h + g -> W[h][w = 16g + p][f][class i] (0 for i = 3, and for p = 16 unless g = 7)
int deployed_f32m_pack(mdc_model* m) {
    const int F = m->topo.filters;
    std::vector<float> tab((size_t)17 * F * 64, 0.f);
    const float* dk = m->hk[1].data();      // (258F, 3), rows h*129F + w*F + f
    for (int p = 0; p < 17; ++p)
        for (int f = 0; f < F; ++f)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 3, B = lane >> 2, g = B & 7, h = B >> 3;
                const int w = 16 * g + p;
                if (i >= kC || (p == 16 && g != 7)) continue;
                tab[((size_t)p * F + f) * 64 + lane] = dk[((size_t)h * 129 * F + (size_t)w * F + f) * kC + i];
            }
    return upload(m, 3, tab.data(), tab.size() * sizeof(float));
}

template <int F, int TAP, bool U8>
static int launch_f32m(const mdc_model* m, const void* x, int64_t n, float scale, long hop2, float* probs, int32_t* labels, float* tap_dense,
                       hipStream_t s) {
    using G = F32mGeom<F, U8>;
    const float* wp = static_cast<const float*>(m->d_pack[0]);
    const float* atab = static_cast<const float*>(m->d_pack[3]);
    const long ngroups = (n + 3) / 4;
    // one work-group (8 waves, 2 per SIMD) per CU; runs of up to 16 groups per wave, shorter when the batch is small
    const long total_waves = 256L * G::kWaves;
    long run = (ngroups + total_waves - 1) / total_waves;
    run = run < 1 ? 1 : (run > 16 ? 16 : run);
#ifdef MDC_ABLATIONS
    if (getenv("MDC_F32M_RUN")) { const long r = atol(getenv("MDC_F32M_RUN")); if (r >= 1 && r <= 16 && r < run) run = r; }
#endif
    const long nruns = (ngroups + run - 1) / run;
    long grid = (nruns + G::kWaves - 1) / G::kWaves;
    if (grid > 256) grid = 256;
#ifdef MDC_ABLATIONS
    static const int abl = getenv("MDC_ABLATE_F32M") ? atoi(getenv("MDC_ABLATE_F32M")) : 0;
#define MDC_F32M_ABL(A) if (abl == A) { \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deployed_f32m_kernel<F, TAP, U8, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::kLds)); \
        hipLaunchKernelGGL((deployed_f32m_kernel<F, TAP, U8, A>), dim3((unsigned)grid), dim3(64 * G::kWaves), G::kLds, s, static_cast<const float*>(x), (long)n, wp, atab, probs, labels, tap_dense, scale, hop2, (int)run); \
        MDC_HIP(hipGetLastError()); return MDC_OK; }
    MDC_F32M_ABL(1) MDC_F32M_ABL(2) MDC_F32M_ABL(3) MDC_F32M_ABL(4) MDC_F32M_ABL(8) MDC_F32M_ABL(9) MDC_F32M_ABL(11)
#undef MDC_F32M_ABL
    // ring-depth probes of the streaming rate (no conv, no MFMA): MDC_ABLATE_F32M = 100 + ring
#define MDC_F32M_RINGP(R) if (abl == 100 + R && !U8) { using GR = F32mGeom<F, U8, R>; \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deployed_f32m_kernel<F, TAP, U8, 3, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GR::kLds)); \
        hipLaunchKernelGGL((deployed_f32m_kernel<F, TAP, U8, 3, R>), dim3((unsigned)grid), dim3(64 * GR::kWaves), GR::kLds, s, static_cast<const float*>(x), (long)n, wp, atab, probs, labels, tap_dense, scale, hop2, (int)run); \
        MDC_HIP(hipGetLastError()); return MDC_OK; }
    MDC_F32M_RINGP(2) MDC_F32M_RINGP(3)
#undef MDC_F32M_RINGP
#endif
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deployed_f32m_kernel<F, TAP, U8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::kLds));
    hipLaunchKernelGGL((deployed_f32m_kernel<F, TAP, U8>), dim3((unsigned)grid), dim3(64 * G::kWaves), G::kLds, s, static_cast<const float*>(x), (long)n, wp, atab,
                       probs, labels, tap_dense, scale, hop2, (int)run);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

// f32 frames (tap_dense: the Dense+ReLU tap of CNN.ipynb cell 17, or NULL)
int deployed_f32m_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels, float* tap_dense, hipStream_t s) {
    const int F = m->topo.filters;
    if (tap_dense) return F == 3 ? launch_f32m<3, 2, false>(m, x, n, 0.f, 0, probs, labels, tap_dense, s) : launch_f32m<10, 2, false>(m, x, n, 0.f, 0, probs, labels, tap_dense, s);
    return F == 3 ? launch_f32m<3, 0, false>(m, x, n, 0.f, 0, probs, labels, nullptr, s) : launch_f32m<10, 0, false>(m, x, n, 0.f, 0, probs, labels, nullptr, s);
}

// raw uint8 I/Q windows, hop2 bytes apart
int deployed_f32m_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, long hop2, float scale, float* probs, int32_t* labels, hipStream_t s) {
    return m->topo.filters == 3 ? launch_f32m<3, 0, true>(m, iq, n, scale, hop2, probs, labels, nullptr, s)
                                : launch_f32m<10, 0, true>(m, iq, n, scale, hop2, probs, labels, nullptr, s);
}

}  // namespace mdc

#endif  // MDC_ALTERNATES
