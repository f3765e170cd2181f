// Deployed single-conv nets (T1: F=3, T2: F=10) with the dense layer on the matrix cores: MDC_BF16 and MDC_F16.
//
// Why: the f32 kernel (deployed.hip) is one wave = one frame, each lane multiplying ITS conv outputs by ITS dense
// weights on the vector ALU -- 3 FMAs per conv output, which for F = 10 (2,580 outputs x 3 classes per frame) makes
// the kernel VALU-bound at a third of the HBM roofline.  Here the dense layer is a GEMM on v_mfma_f32_16x16x32_bf16
// (or _f16):
//     D[class][frame] += A[class][k] * B[k][frame],   k = one conv output of the frame,
// which needs "lane = frame": the MFMA's B operand of lane (n = lane & 15, kg = lane >> 4) is 8 k-values of column
// (frame) n.  So a wave takes 16 frames at a time and lane (f, g) owns a QUARTER of frame f: the sixteen 4-sample
// pieces p = 4j + g (j = 0..15) of the frame's 64 pieces (row I = pieces 0..31, row Q = 32..63), read from a
// wave-private staging area in LDS that LDS-DMA fills (every byte of the batch crosses HBM exactly once).  For each
// piece the lane computes its 4 conv positions x F filters, applies ReLU on the packed halves, and the values ARE
// the B operand (k order = order of production; the host lays the dense weights out to match; only classes 0..2
// of the MFMA's 16 rows are real -- the other rows are never read back).  The sample x[w] a piece lacks for its last
// position is the first sample of the NEXT piece: one extra ds_read_b32, zero at the row end.  Position w = 0 of
// each row (x[-1] = 0) is an extra 2F values that only the g = 0 lane of a frame owns (the other lanes' weights for
// those slots are zero).  The dense part is 8F + 1..3 MFMAs per 16 frames.
//
// HALF = false (MDC_BF16): conv in f32 with the f32 kernel's fma chain (fma(K1, x[w], fma(K0, x[w-1], b)), two
// filters per v_pk_fma_f32), outputs rounded to bf16 AND rectified by one v_cvt_pk_bf16_f32 with the clamp bit (round 3,
// pack2clamp below; round 2: v_cvt_pk_bf16_f32 + v_pk_max_i16); dense weights bf16.
// 2 + 1 VALU per two outputs for F = 10 -- but v_pk_fma_f32 is a slow instruction here.
// HALF = true (MDC_F16): IEEE f16 operands, and then the conv itself runs in packed f16 (v_pk_fma_f16 on 32-bit
// registers: two filters per instruction at the plain VALU rate, ReLU = one v_pk_max_f16, no conversion before the
// MFMA), which is what lifts the F = 10 net off the v_pk_fma_f32 bound.  f16 keeps 11 significant bits (bf16: 8) but
// only ~5 decades of range: conv outputs must stay below 65,504.
// FP8 (MDC_FP8, round 2: BASELINE configs[4] read literally -- "5convmodrecnets_CNN2_0.5.wts.h5, fp8 MFMA path"): the
// bf16 mode's f32 conv, its outputs and the dense weights as OCP e4m3 on v_mfma_f32_16x16x32_fp8_fp8 (8 one-byte k values
// per lane).  e4m3 spans 2^-9 .. 448, so the conv's taps and bias are multiplied by 2^sa (sa from the largest |sample| the
// caller states, mdc_set_fp8_input_absmax, default 0.02) and the dense weights by 2^sw on the host -- powers of two,
// exact -- and the class sums are multiplied by 2^-(sa+sw) before the dense bias; the ReLU rides in the conv's second fma
// (clamp bit) and the scaled e4m3 conversion saturates under MODE.FP16_OVFL (round 3, pack4_fp8 below; round 2: a
// v_med3_f32(y, 0, 448) per value), so an input beyond the stated range saturates instead of becoming NaN.
// All: products accumulated in f32 by the MFMA; bias, ReLU, softmax, first-max argmax in f32 as in deployed.hip.
#include "vtcnn2_bf16_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace mdc {

namespace {

constexpr int kC = 3;

template <int F>
struct Bf16Geom {
    static constexpr int kUnitPieces = (4 * F) % 8 == 0 ? 1 : 2;      // pieces whose values fill whole MFMAs
    static constexpr int kUnitVals = kUnitPieces * 4 * F;
    static constexpr int kUnitMfma = kUnitVals / 8;
    static constexpr int kUnits = 16 / kUnitPieces;
    static constexpr int kMainMfma = kUnits * kUnitMfma;              // = 8F
    static constexpr int kExtraMfma = (2 * F + 7) / 8;                // position w = 0 of both rows
    static constexpr int kMfma = kMainMfma + kExtraMfma;
    static constexpr int kATabBytes = kMfma * 4 * kC * 16;            // A table, rows 0..2 only: [mfma][kg][class][8 bf16]
    static constexpr int kPairStride = 1024 + 32;                     // one DMA instruction: row r of frames i and i+8 (2 x 512 B) + pad.
                                                                      // Pairing (f, f+8) and a stride of 264 words puts the 16 lanes one
                                                                      // ds_read_b128 cycle serves -- f in {0..3, 12..15} of k-group g and
                                                                      // f in {4..11} of g+1 -- on 16 different 4-bank groups (pairing
                                                                      // (f, f+1) at 260 words ran every piece read 2-way conflicted:
                                                                      // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.40-0.62, profiles/r02)
    static constexpr int kStageBytes = 8 * kPairStride;               // ONE ROW of a 16-frame group
    static constexpr int kWaves = 8;                                  // 2 per SIMD, each with TWO row buffers (8, 12, 16 waves with
                                                                      // one buffer: 3.45 / 3.40 / 3.06e9 frames/s for F = 10)
    static constexpr size_t kLds = (size_t)kATabBytes + (size_t)kWaves * 2 * kStageBytes;
};

// Frames reach the lanes through a wave-private LDS staging area filled by LDS-DMA, one ROW of the 16 frames at a
// time: a global_load_lds_dwordx4 moves row r of two frames (2 x 512 contiguous bytes), eight of them a row of the
// group; lane (f, g) then reads its pieces of frame f with ds_read_b128 (+ the next piece's first sample with a
// ds_read_b32).  Two row buffers per wave: while row I (pieces j = 0..7) is computed row Q is in flight, while row Q
// is computed row I of the wave's next group is -- without the DMA the F = 10 kernel runs at 4.25e9 frames/s, with a
// single buffer (DMA latency exposed twice per group) at 3.4e9.
// Loading the pieces straight from global memory -- 64 contiguous bytes per frame per instruction -- ran at 3.5e9
// frames/s for F = 3 and F = 10 alike: bound by that access pattern, not by arithmetic.
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;

// bf16 mode: ReLU + pack of two conv outputs in ONE v_cvt_pk_bf16_f32 with the VOP3 clamp bit (round 3; the conversion then
// clamps to [0, 1]: tools/microbench/cvt_bf16_clamp_probe.hip).  The conv taps and bias carry 2^-kDepShift and the dense
// weights 2^+kDepShift -- exact powers of two, every product and f32 sum the same bits as with v_cvt_pk_bf16_f32 +
// v_pk_max_i16 -- so the values stay below 1 unless a conv output exceeds 2^32.
constexpr int kDepShift = 32;
__device__ __forceinline__ unsigned pack2clamp(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// fp8 mode (round 3): four conv outputs -> four e4m3 in one dword (byte i = value i) by two v_cvt_scalef32_pk_fp8_f32.  The
// values arrive already rectified -- the conv's second fma carries the clamp bit: taps and bias are scaled by 2^(sa-9), so
// [0, 1] is [0, 512) in e4m3 units -- and the conversion divides by its scale operand 2^-9; (448, 512) saturates to 448
// because the kernel sets MODE.FP16_OVFL (without it the OCP conversion returns NaN there;
// tools/microbench/cvt_fp8_bf16_probe.hip).  Round 2 spent a v_med3_f32(x, 0, 448) per value on the same thing.
constexpr int kFp8ClampShift = 9;
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    const float sc = 0x1p-9f;
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, a, b, sc, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, c, d, sc, true);
    return __builtin_bit_cast(unsigned, w);
}
__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 1.f); }      // after an fma: folds into its clamp bit

// U8 = true (mdc_forward_iq_u8): x points at raw interleaved unsigned 8-bit (I,Q) samples, 256 B per frame.  A whole
// group is then 4 KiB (one DMA instruction = four frames), the wave's staging area holds a ring of FOUR groups
// (three in flight while one is computed), and a lane reads the 8 bytes that hold its four samples of both rows and
// converts the row at hand with the arithmetic of mdc_iq_u8_to_frames -- the f32 samples, and so every result, are
// bit-identical to convert-then-forward.
template <int F, int MODE, bool U8>      // MODE: 0 bf16, 1 f16, 2 fp8 (e4m3)
__global__ __launch_bounds__(512, 1) void deployed_bf16_kernel(const float* __restrict__ x, long n,
                                                                const float* __restrict__ wp, const uint4* __restrict__ atab,
                                                                float* __restrict__ probs, int* __restrict__ labels, float scale, long hop2) {
    using G = Bf16Geom<F>;
    constexpr bool HALF = MODE == 1, FP8 = MODE == 2;
    constexpr int kPhaseUnits = G::kUnits / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* a_lds = reinterpret_cast<uint4*>(smem);
    if constexpr (FP8) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");      // FP16_OVFL: the e4m3 conversions saturate (pack4_fp8)
    for (int i = threadIdx.x; i < G::kATabBytes / 16; i += blockDim.x) a_lds[i] = atab[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int f = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* stage = smem + G::kATabBytes + wv * 2 * G::kStageBytes;      // row r of a group goes to buffer r
    // piece 4jj + g (jj = 0..7) of the staged row r of frame f at mine0 + r * kStageBytes + 64 jj
    const unsigned char* mine0 = stage + (f & 7) * G::kPairStride + (f >> 3) * 512 + g * 16;
    // this lane's A row: class c = lane & 15, k-group g.  Rows 3..15 of the MFMA are never read back, so their
    // lanes simply load class 0's weights again (no zero rows in LDS, no masking)
    const uint4* a_mine = a_lds + g * kC + (f < kC ? f : 0);

    float k0[F], k1[F], cb[F], bd[kC];
#pragma unroll
    for (int i = 0; i < F; ++i) { k0[i] = wp[3 * i + 0]; k1[i] = wp[3 * i + 1]; cb[i] = wp[3 * i + 2]; }
#pragma unroll
    for (int c = 0; c < kC; ++c) bd[c] = wp[3 * F + c];
    const float unscale = FP8 ? wp[3 * F + kC] : 1.f;      // fp8: 2^-(sa+sw), exact
    // f16 mode: taps and bias as f16 pairs (filters f, f+1); an odd F gets a zero partner
    h16x2 k0h[(F + 1) / 2], k1h[(F + 1) / 2], cbh[(F + 1) / 2];
#pragma unroll
    for (int i = 0; i < (F + 1) / 2; ++i) {
        const bool two = 2 * i + 1 < F;
        k0h[i] = h16x2{(_Float16)k0[2 * i], two ? (_Float16)k0[2 * i + (two ? 1 : 0)] : (_Float16)0.f};
        k1h[i] = h16x2{(_Float16)k1[2 * i], two ? (_Float16)k1[2 * i + (two ? 1 : 0)] : (_Float16)0.f};
        cbh[i] = h16x2{(_Float16)cb[2 * i], two ? (_Float16)cb[2 * i + (two ? 1 : 0)] : (_Float16)0.f};
    }

    const long ngroups = (n + 15) >> 4;
    const long gstep = (long)gridDim.x * G::kWaves;
    auto stage_row = [&](long grp, int r) {       // 8 x (2 x 512 B); frames past the end of the batch re-read the last one
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long fr = grp * 16 + i + 8 * (lane >> 5);
            glds16_async(x + (fr < n ? fr : n - 1) * kFrameFloats + r * kSamples + (lane & 31) * 4, stage + r * G::kStageBytes + i * G::kPairStride);
        }
    };
    // raw bytes: ring slot of a group = 4 DMA instructions (4 frames x 256 B each) at stage + slot * kRawGroup
    constexpr int kRawGroup = 4 * G::kPairStride;
    static_assert(4 * kRawGroup <= 2 * G::kStageBytes, "the ring of four raw groups fits the two row buffers");
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);
    auto stage_raw = [&](long grp, int slot) {    // groups past the end re-read the last one (constant DMA count per step)
        const long gc = grp < ngroups ? grp : ngroups - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long fr = gc * 16 + 4 * i + (lane >> 4);
            glds16_async(xb + (fr < n ? fr : n - 1) * hop2 + (lane & 15) * 16, stage + slot * kRawGroup + i * G::kPairStride);
        }
    };
    // raw bytes: the 8 bytes of piece 4jj + g (samples of BOTH rows) of frame f at rawmine + slot * kRawGroup + 32 jj
    const unsigned char* rawmine = stage + (f >> 2) * G::kPairStride + (f & 3) * 256 + g * 8;
    long grp = (long)blockIdx.x * G::kWaves + wv;
    if constexpr (U8) {
        if (grp < ngroups) { stage_raw(grp, 0); stage_raw(grp + gstep, 1); stage_raw(grp + 2 * gstep, 2); }
    } else {
        if (grp < ngroups) stage_row(grp, 0);
    }
    for (int it = 0; grp < ngroups; grp += gstep, ++it) {
        const long frame = grp * 16 + f;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float x0[2] = {0.f, 0.f};      // first sample of each row (x[0], for the g = 0 lane's position w = 0)
        const unsigned char* raw = rawmine + (it & 3) * kRawGroup;
        if constexpr (U8) {
            // the group three steps ahead goes into the slot whose last reader finished a step ago (lgkmcnt(0) below);
            // then at most those three groups' twelve DMA instructions may be outstanding: this group has landed
            stage_raw(grp + 3 * gstep, (it + 3) & 3);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned char* mine = mine0 + r * G::kStageBytes;
            if constexpr (!U8) {
                // the row after this one goes into the other buffer now (its last reader finished a row ago, lgkmcnt(0)
                // below); then wait until only those eight DMA instructions are outstanding: this row has landed
                const bool more = r == 0 || grp + gstep < ngroups;
                if (r == 0) stage_row(grp, 1);
                else if (more) stage_row(grp + gstep, 0);
                if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // LDS reads run one unit ahead of the arithmetic: pieces, next-piece samples and the A rows of the next
            // unit are requested before this one is computed (the sched_barrier that bounds the register use would
            // otherwise put every unit's LDS latency in front of its own arithmetic)
            float4 c4n[G::kUnitPieces];
            float nbn[G::kUnitPieces];
            uint2 r8n[G::kUnitPieces];       // raw bytes: the piece (I0 Q0 I1 Q1 | I2 Q2 I3 Q3) ...
            unsigned nr8n[G::kUnitPieces];   // ... and the first (I, Q) pairs of the next piece
            uint4 an[G::kUnitMfma];
            auto read_unit = [&](int up) {          // up = unit within the phase
#pragma unroll
                for (int q = 0; q < G::kUnitPieces; ++q) {
                    const int jj = up * G::kUnitPieces + q;
                    if constexpr (U8) {
                        r8n[q] = *reinterpret_cast<const uint2*>(raw + 32 * jj);          // both rows live in the same 8 bytes
                        nr8n[q] = *reinterpret_cast<const unsigned*>(raw + 32 * jj + 8);
                    } else {
                        c4n[q] = *reinterpret_cast<const float4*>(mine + 64 * jj);
                        nbn[q] = *reinterpret_cast<const float*>(mine + 64 * jj + 16);
                    }
                }
#pragma unroll
                for (int mm = 0; mm < G::kUnitMfma; ++mm) an[mm] = a_mine[((r * kPhaseUnits + up) * G::kUnitMfma + mm) * 4 * kC];
            };
            read_unit(0);
#pragma unroll
            for (int up = 0; up < kPhaseUnits; ++up) {
                const int u = r * kPhaseUnits + up;
                float4 c4[G::kUnitPieces];
                float nbv[G::kUnitPieces];
                uint4 a[G::kUnitMfma];
#pragma unroll
                for (int q = 0; q < G::kUnitPieces; ++q) {
                    if constexpr (U8) {
                        // row r = bytes r, r+2 of each dword; ((float)byte - 127.5) * scale as in iq_u8_kernel (eval_ops.hip)
                        const unsigned d0 = r8n[q].x >> (8 * r), d1 = r8n[q].y >> (8 * r), d2 = nr8n[q] >> (8 * r);
                        c4[q] = make_float4(((float)(d0 & 0xFFu) - 127.5f) * scale, ((float)((d0 >> 16) & 0xFFu) - 127.5f) * scale,
                                            ((float)(d1 & 0xFFu) - 127.5f) * scale, ((float)((d1 >> 16) & 0xFFu) - 127.5f) * scale);
                        nbv[q] = ((float)(d2 & 0xFFu) - 127.5f) * scale;
                    } else {
                        c4[q] = c4n[q]; nbv[q] = nbn[q];
                    }
                }
#pragma unroll
                for (int mm = 0; mm < G::kUnitMfma; ++mm) a[mm] = an[mm];
                if (up + 1 < kPhaseUnits) read_unit(up + 1);
                // the unit's conv outputs, packed two per word in production order (position, filter): bf16 after an f32
                // conv, or f16 straight out of a packed-f16 conv
                unsigned pk[G::kUnitVals / 2];
                if constexpr (!HALF) {
                    float vals[G::kUnitVals];
#pragma unroll
                    for (int q = 0; q < G::kUnitPieces; ++q) {
                        const int jj = up * G::kUnitPieces + q;
                        // the last piece of a row (jj == 7, g == 3) has no next sample: x[128] = 0
                        if (jj == 7 && g == 3) nbv[q] = 0.f;
                        if (jj == 0) x0[r] = c4[q].x;
                        const float xs[5] = {c4[q].x, c4[q].y, c4[q].z, c4[q].w, nbv[q]};
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            if constexpr (FP8) {      // ReLU in the second fma's clamp bit (pack4_fp8)
                                if constexpr (F % 2 == 0) {
#pragma unroll
                                    for (int ff = 0; ff < F; ff += 2) {
                                        const f32x2 t = __builtin_elementwise_fma(f32x2{k0[ff], k0[ff + 1]}, f32x2{xs[s], xs[s]}, f32x2{cb[ff], cb[ff + 1]});
                                        vals[(q * 4 + s) * F + ff] = clamp01(fmaf(k1[ff], xs[s + 1], t.x));
                                        vals[(q * 4 + s) * F + ff + 1] = clamp01(fmaf(k1[ff + 1], xs[s + 1], t.y));
                                    }
                                } else {
#pragma unroll
                                    for (int ff = 0; ff < F; ++ff) vals[(q * 4 + s) * F + ff] = clamp01(fmaf(k1[ff], xs[s + 1], fmaf(k0[ff], xs[s], cb[ff])));
                                }
                            } else if constexpr (F % 2 == 0) {
#pragma unroll
                                for (int ff = 0; ff < F; ff += 2) {
                                    const f32x2 y = __builtin_elementwise_fma(f32x2{k1[ff], k1[ff + 1]}, f32x2{xs[s + 1], xs[s + 1]},
                                                                              __builtin_elementwise_fma(f32x2{k0[ff], k0[ff + 1]}, f32x2{xs[s], xs[s]},
                                                                                                        f32x2{cb[ff], cb[ff + 1]}));
                                    vals[(q * 4 + s) * F + ff] = y.x;
                                    vals[(q * 4 + s) * F + ff + 1] = y.y;
                                }
                            } else {
#pragma unroll
                                for (int ff = 0; ff < F; ++ff) vals[(q * 4 + s) * F + ff] = fmaf(k1[ff], xs[s + 1], fmaf(k0[ff], xs[s], cb[ff]));
                            }
                        }
                    }
                    if constexpr (FP8) {
#pragma unroll
                        for (int i = 0; i < G::kUnitVals / 4; ++i) pk[i] = pack4_fp8(vals[4 * i], vals[4 * i + 1], vals[4 * i + 2], vals[4 * i + 3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < G::kUnitVals / 2; ++i) pk[i] = pack2clamp(vals[2 * i], vals[2 * i + 1]);
                    }
                } else {
                    _Float16 hv[G::kUnitVals];
#pragma unroll
                    for (int q = 0; q < G::kUnitPieces; ++q) {
                        const int jj = up * G::kUnitPieces + q;
                        if (jj == 7 && g == 3) nbv[q] = 0.f;
                        if (jj == 0) x0[r] = c4[q].x;
                        const _Float16 xh[5] = {(_Float16)c4[q].x, (_Float16)c4[q].y, (_Float16)c4[q].z, (_Float16)c4[q].w, (_Float16)nbv[q]};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int ff = 0; ff < F; ff += 2) {
                                h16x2 y = __builtin_elementwise_fma(k1h[ff / 2], h16x2{xh[s + 1], xh[s + 1]},
                                                                    __builtin_elementwise_fma(k0h[ff / 2], h16x2{xh[s], xh[s]}, cbh[ff / 2]));
                                y = __builtin_elementwise_max(y, h16x2{(_Float16)0.f, (_Float16)0.f});
                                hv[(q * 4 + s) * F + ff] = y.x;
                                if (ff + 1 < F) hv[(q * 4 + s) * F + ff + 1] = y.y;
                            }
                    }
#pragma unroll
                    for (int i = 0; i < G::kUnitVals / 2; ++i) pk[i] = __builtin_bit_cast(unsigned, h16x2{hv[2 * i], hv[2 * i + 1]});
                }
#pragma unroll
                for (int mm = 0; mm < G::kUnitMfma; ++mm) {
                    const int m = u * G::kUnitMfma + mm;
                    if constexpr (FP8) {
                        const long b8 = (long)(((unsigned long)pk[2 * mm + 1] << 32) | pk[2 * mm]);
                        const long a8 = (long)(((unsigned long)a[mm].y << 32) | a[mm].x);
                        acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, b8, acc[m & 1], 0, 0, 0);
                    } else {
                        const u32x4 b = u32x4{pk[4 * mm + 0], pk[4 * mm + 1], pk[4 * mm + 2], pk[4 * mm + 3]};
                        if constexpr (HALF) acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a[mm]), __builtin_bit_cast(h16x8, b), acc[m & 1], 0, 0, 0);
                        else acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[mm]), __builtin_bit_cast(bf16x8, b), acc[m & 1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep each unit's values inside the unit (register budget)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of this row's buffer is done
        }
        {   // position w = 0 of row I and row Q, owned by the g = 0 lane (x[-1] = 0): y = relu(b + K1 x[0]);
            // lanes g != 0 compute the same expression on finite samples of their own and meet zero weights
            float ev[8 * G::kExtraMfma];
#pragma unroll
            for (int i = 0; i < 8 * G::kExtraMfma; ++i) ev[i] = 0.f;
#pragma unroll
            for (int ff = 0; ff < F; ++ff) {
                ev[ff] = FP8 ? clamp01(fmaf(k1[ff], x0[0], cb[ff])) : fmaf(k1[ff], x0[0], cb[ff]);
                ev[F + ff] = FP8 ? clamp01(fmaf(k1[ff], x0[1], cb[ff])) : fmaf(k1[ff], x0[1], cb[ff]);
            }
#pragma unroll
            for (int mm = 0; mm < G::kExtraMfma; ++mm) {
                const int m = G::kMainMfma + mm;
                const uint4 a = a_mine[m * 4 * kC];
                if constexpr (HALF) {
                    // f16 mode: the same values through f16 arithmetic (x and taps rounded to f16 as in the main part)
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        _Float16 e[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int idx = 8 * mm + 2 * i + k;      // slot: row idx / F, filter idx % F
                            _Float16 v = (_Float16)0.f;
                            if (idx < 2 * F) {
                                const int ff = idx % F;
                                const _Float16 kk = (ff & 1) ? k1h[ff / 2].y : k1h[ff / 2].x, bb = (ff & 1) ? cbh[ff / 2].y : cbh[ff / 2].x;
                                v = __builtin_fmaf16(kk, (_Float16)x0[idx / F], bb);
                                v = v > (_Float16)0.f ? v : (_Float16)0.f;
                            }
                            e[k] = v;
                        }
                        w[i] = __builtin_bit_cast(unsigned, h16x2{e[0], e[1]});
                    }
                    acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, u32x4{w[0], w[1], w[2], w[3]}), acc[m & 1], 0, 0, 0);
                } else if constexpr (FP8) {
                    const unsigned lo = pack4_fp8(ev[8 * mm + 0], ev[8 * mm + 1], ev[8 * mm + 2], ev[8 * mm + 3]);
                    const unsigned hi = pack4_fp8(ev[8 * mm + 4], ev[8 * mm + 5], ev[8 * mm + 6], ev[8 * mm + 7]);
                    acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)(((unsigned long)a.y << 32) | a.x), (long)(((unsigned long)hi << 32) | lo), acc[m & 1], 0, 0, 0);
                } else {
                    const u32x4 b = u32x4{pack2clamp(ev[8 * mm + 0], ev[8 * mm + 1]), pack2clamp(ev[8 * mm + 2], ev[8 * mm + 3]),
                                          pack2clamp(ev[8 * mm + 4], ev[8 * mm + 5]), pack2clamp(ev[8 * mm + 6], ev[8 * mm + 7])};
                    acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m & 1], 0, 0, 0);
                }
            }
        }
        // D rows 0..2 (the classes) of column f live in lanes 0..15 (kg = 0), registers 0..2
        if (g == 0 && frame < n) {
            // Dense(3, activation='relu'); fp8: the sums carry 2^(sa+sw), removed exactly before the bias
            const float z0 = fmaxf(FP8 ? fmaf(acc[0][0] + acc[1][0], unscale, bd[0]) : acc[0][0] + acc[1][0] + bd[0], 0.f);
            const float z1 = fmaxf(FP8 ? fmaf(acc[0][1] + acc[1][1], unscale, bd[1]) : acc[0][1] + acc[1][1] + bd[1], 0.f);
            const float z2 = fmaxf(FP8 ? fmaf(acc[0][2] + acc[1][2], unscale, bd[2]) : acc[0][2] + acc[1][2] + bd[2], 0.f);
            const float mx = fmaxf(z0, fmaxf(z1, z2));
            const float e0 = expf(z0 - mx), e1 = expf(z1 - mx), e2 = expf(z2 - mx);
            const float inv = 1.0f / (e0 + e1 + e2);
            const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
            if (probs) {
                probs[frame * 3 + 0] = p0;
                probs[frame * 3 + 1] = p1;
                probs[frame * 3 + 2] = p2;
            }
            // int(np.argmax(test_Y_hat[i,:])) (cnn.py:209): FIRST maximum of the probabilities as returned
            if (labels) labels[frame] = (p0 >= p1 && p0 >= p2) ? 0 : ((p1 >= p2) ? 1 : 2);
        }
    }
}

// host f32 -> IEEE f16 (RNE, saturating to +-inf like the hardware conversion)
inline unsigned short f2h(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 0xFF) - 127 + 15;
    unsigned man = u & 0x7FFFFFu;
    if (((u >> 23) & 0xFF) == 0xFF) return (unsigned short)(sign | 0x7C00u | (man ? 0x200u : 0u));
    if (e >= 31) return (unsigned short)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (unsigned short)sign;
        man |= 0x800000u;
        const int shift = 14 - e;
        unsigned h = man >> shift;
        const unsigned rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (unsigned short)(sign | h);
    }
    unsigned h = ((unsigned)e << 10) | (man >> 13);
    const unsigned rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (unsigned short)(sign | h);
}

// e4m3 (OCP "fn": no infinities, max 448, NaN = 0x7F), round-to-nearest-even, saturating
inline unsigned char f2e4m3_sat(float f) {
    if (std::isnan(f)) return 0x7F;
    const unsigned char sign = std::signbit(f) ? 0x80 : 0x00;
    const float a = std::fabs(f);
    if (a >= 448.f) return sign | 0x7E;
    if (a < std::ldexp(1.f, -10)) return sign;                       // below half the smallest subnormal (2^-9)
    int e;
    (void)std::frexp(a, &e);                                         // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;
    if (E < -6) E = -6;                                              // subnormal: fixed exponent
    int mnt = (int)std::nearbyint(std::ldexp(a, 3 - E));             // 3 fractional bits (RNE by default)
    if (mnt == 16) { mnt = 8; ++E; }
    if (E > 8 || (E == 8 && mnt > 14)) return sign | 0x7E;
    if (mnt < 8) return sign | (unsigned char)mnt;                   // subnormal: exponent field 0
    return sign | (unsigned char)(((E + 7) << 3) | (mnt - 8));
}

// The dense layer as MFMA A operands, [mfma][k-group][class][16 bytes]: 8 bf16 / f16, or 8 e4m3 (x wscale) in the
// first 8 bytes of the entry
template <int F>
void pack_atab(const mdc_model* m, std::vector<unsigned short>& tab, float wscale) {
    using G = Bf16Geom<F>;
    tab.assign((size_t)G::kATabBytes / 2, 0);
    const float* dk = m->hk[1].data();      // (258F, 3), rows h*129F + w*F + f
    for (int mi = 0; mi < G::kMfma; ++mi)
        for (int kg = 0; kg < 4; ++kg)
            for (int c = 0; c < kC; ++c)
                for (int i = 0; i < 8; ++i) {
                    float w = 0.f;
                    if (mi < G::kMainMfma) {
                        const int local = 8 * mi + i;
                        const int j = local / (4 * F), rem = local % (4 * F), s = rem / F, ff = rem % F;
                        const int p = 4 * j + kg, h = p >> 5, pos = 4 * (p & 31) + 1 + s;
                        w = dk[((size_t)h * 129 * F + (size_t)pos * F + ff) * kC + c];
                    } else {
                        const int local = 8 * (mi - G::kMainMfma) + i;
                        if (local < 2 * F && kg == 0) {
                            const int h = local / F, ff = local % F;
                            w = dk[((size_t)h * 129 * F + ff) * kC + c];      // position 0
                        }
                    }
                    const size_t entry = (((size_t)mi * 4 + kg) * kC + c) * 8;      // in 16-bit units
                    if (m->dtype == MDC_FP8) reinterpret_cast<unsigned char*>(&tab[entry])[i] = f2e4m3_sat(w * wscale);
                    else tab[entry + i] = m->dtype == MDC_F16 ? f2h(w) : f2bf(w * wscale);      // bf16: wscale = 2^kDepShift (exact)
                }
}

}  // namespace

// d_pack slot 2: the dense layer as MFMA A operands (bf16 / f16 / e4m3).  fp8 mode also builds its own head (slot 4):
// conv taps and bias x 2^(sa-9) (the e4m3 values carry 2^sa), dense bias, 2^-(sa+sw) -- the f32 head of slot 0 stays as it is for the Q6.12 tables' sake.
int deployed_bf16_pack(mdc_model* m) {
    const int F = m->topo.filters;
    float wscale = 1.f;
    if (m->dtype == MDC_FP8) {
        const float* ck = m->hk[0].data();   // HWIO (1,2,1,F): [kw][f]
        float cbound = 0.f, wmax = 0.f;
        for (int f = 0; f < F; ++f) cbound = std::fmax(cbound, m->fp8_input_absmax * (std::fabs(ck[f]) + std::fabs(ck[F + f])) + std::fabs(m->hb[0][f]));
        for (float w : m->hk[1]) wmax = std::fmax(wmax, std::fabs(w));
        if (!(cbound > 0.f) || !(wmax > 0.f)) { set_error("fp8: degenerate weights (all zero)"); return MDC_EINVAL; }
        const int sa = (int)std::floor(std::log2(224.f / cbound)), sw = (int)std::floor(std::log2(224.f / wmax));      // a factor 2 of head-room
        const float fsa = std::ldexp(1.f, sa - kFp8ClampShift);      // the kernel's conversion multiplies by 2^9 again (pack4_fp8)
        wscale = std::ldexp(1.f, sw);
        std::vector<float> head(64, 0.f);
        for (int f = 0; f < F; ++f) {
            head[3 * f + 0] = ck[f] * fsa;
            head[3 * f + 1] = ck[F + f] * fsa;
            head[3 * f + 2] = m->hb[0][f] * fsa;
        }
        for (int c = 0; c < kC; ++c) head[3 * F + c] = m->hb[1][c];
        head[3 * F + kC] = std::ldexp(1.f, -(sa + sw));
        const int rc = upload(m, 4, head.data(), head.size() * sizeof(float));
        if (rc != MDC_OK) return rc;
    }
    if (m->dtype == MDC_BF16) {      // conv taps and bias x 2^-kDepShift, dense weights x 2^+kDepShift (pack2clamp); slot 4 like the fp8 head
        const float* ck = m->hk[0].data();
        const float sc = std::ldexp(1.f, -kDepShift);
        wscale = std::ldexp(1.f, kDepShift);
        std::vector<float> head(64, 0.f);
        for (int f = 0; f < F; ++f) {
            head[3 * f + 0] = ck[f] * sc;
            head[3 * f + 1] = ck[F + f] * sc;
            head[3 * f + 2] = m->hb[0][f] * sc;
        }
        for (int c = 0; c < kC; ++c) head[3 * F + c] = m->hb[1][c];
        head[3 * F + kC] = 1.f;
        const int rc = upload(m, 4, head.data(), head.size() * sizeof(float));
        if (rc != MDC_OK) return rc;
    }
    std::vector<unsigned short> tab;
    if (F == 3) pack_atab<3>(m, tab, wscale);
    else pack_atab<10>(m, tab, wscale);
    return upload(m, 2, tab.data(), tab.size() * sizeof(unsigned short));
}

template <int F, bool U8>
static int launch_bf16(const mdc_model* m, const void* x, int64_t n, float scale, float* probs, int32_t* labels, hipStream_t s, long hop2 = 256) {
    using G = Bf16Geom<F>;
    const float* wp = static_cast<const float*>(m->d_pack[(m->dtype == MDC_FP8 || m->dtype == MDC_BF16) ? 4 : 0]);      // scaled heads (slot 4)
    const uint4* atab = static_cast<const uint4*>(m->d_pack[2]);
    const float* xf = static_cast<const float*>(x);
    const long ngroups = (n + 15) / 16;
    long grid = (ngroups + G::kWaves - 1) / G::kWaves;
    if (grid > 256) grid = 256;      // one work-group per CU (LDS: A table + 8 x 2 row buffers)
#define MDC_LAUNCH_DEP16(MODE) do { \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deployed_bf16_kernel<F, MODE, U8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::kLds)); \
        hipLaunchKernelGGL((deployed_bf16_kernel<F, MODE, U8>), dim3((unsigned)grid), dim3(64 * G::kWaves), G::kLds, s, xf, (long)n, wp, atab, probs, labels, scale, hop2); } while (0)
    if (m->dtype == MDC_F16) MDC_LAUNCH_DEP16(1);
    else if (m->dtype == MDC_FP8) MDC_LAUNCH_DEP16(2);
    else MDC_LAUNCH_DEP16(0);
#undef MDC_LAUNCH_DEP16
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

int deployed_bf16_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels, hipStream_t s) {
    ProfScope ps(m, 0, s);
    return m->topo.filters == 3 ? launch_bf16<3, false>(m, x, n, 0.f, probs, labels, s) : launch_bf16<10, false>(m, x, n, 0.f, probs, labels, s);
}

// raw uint8 I/Q (256 B per frame) straight into the 16-bit kernels
int deployed_bf16_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, int64_t hop, float scale, float* probs, int32_t* labels, hipStream_t s) {
    ProfScope ps(m, 0, s);
    const long hop2 = 2 * (long)hop;
    return m->topo.filters == 3 ? launch_bf16<3, true>(m, iq, n, scale, probs, labels, s, hop2) : launch_bf16<10, true>(m, iq, n, scale, probs, labels, s, hop2);
}

}  // namespace mdc
