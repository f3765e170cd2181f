// Q6.12 integer forward of the deployed nets (T1/T2): the arithmetic of the reference's FPGA datapath
// (cnn_test_latest1.sv) on the GPU, so that ROM tables and test vectors written with the `float2fix` exporter can
// be validated against the float path and against each other at batch scale.
//
// Rules followed (citations into /root/reference/cnn_test_latest1.sv):
//   operands: 18-bit two's-complement Q6.12;  quantisation of floats = float2fix (CNN.ipynb cell 23): trunc(v*4096)
//   conv neuron  signed_mult1 (sv:642-658): m = a*b + c*d on a 36-bit wire; out = {m[35], m[28:12]}; out+bias wraps to
//                18 bit; ReLU = 0 when bit 17 is set
//   dense term   signed_mult  (sv:664-675): the same bit selection of I_act*W_i + Q_act*W_q, sign-extended into a
//                32-bit accumulator that starts at the sign-extended bias (sv:293-343); ReLU on bit 31 (sv:171-176)
// Activation/weight ORDER is Keras' (the RTL's clocking and index reversal are not modelled).
//
// Mapping (round 2; the first version ran one wave per frame with every weight fetched from global memory per use:
// 4.9e8 / 2.0e8 frames/s for F = 3 / 10): persistent waves, each walking 64-frame blocks.  Lane l owns conv positions
// w = 2l+1 and 2l+2 of both rows (samples x[2l], x[2l+1] by one 8-byte load per row, x[2l+2] from lane l+1 by DPP;
// lane 63's second position is w = 128 with x[128] = 0) and keeps their integer dense weights -- 2 positions x F
// filters x C classes x {I, Q} tables -- in registers for the whole kernel.  A product pair a*b + c*d of 18-bit
// operands is formed exactly in 64 bits by two v_mad_i64_i32; {m[35], m[28:12]} comes out of its two halves.  The 32-bit
// wrap-around sums are associative, so the per-lane partial sums are combined with a butterfly and the result is
// bit-identical to the FPGA's sequential accumulation; lane i of the wave parks the totals (and the quantised x[h][0])
// of the block's frame i, and after 64 frames every lane finishes one frame: position w = 0 (x[-1] = 0; uniform
// weights), bias, ReLU, first-max label, 768 + 256 B of coalesced stores.
// HBM: 1 KiB in, 4*C + 4 B out per frame.
//
// Round 3, measured and NOT kept -- both forms bit-exact on the whole Q6.12 suite:
//  * the pair from 24-bit multiplies: with b = bh*4096 + bl, floor(m / 4096) = a*bh + c*dh + ((a*bl + c*dl) >> 12), every
//    product < 2^30 (v_mul_i32_i24 / v_mad_i32_i24), and {m[35], m[28:12]} = {r[23], r[16:0]} of that r.  T1 1.21e9
//    against 1.59e9 frames/s, T2 4.4e8 against 4.8e8: v_mad_i64_i32 is not a quarter-rate instruction on gfx950 -- it
//    issues at HALF the plain VALU rate (9.5 against 4.9 cycles per wave-instruction, tools/microbench/imul_rate.hip,
//    profiles/r03_imul_rate.log), so two of them (19) beat two v_mul_i32_i24 + two v_mad_i32_i24 + shift + add (31);
//  * four frames at a time with the f32 kernel's bank-masked DPP reduce-scatter instead of the shuffle butterfly
//    (10 VALU per frame instead of 18 ds_bpermute + 18 adds) on top of the 64-bit pairs: T1 1.47e9 (118 VGPRs), and
//    F = 10 does not fit its 120 weight registers + four frames' operands in 256 VGPRs (241 spills, 2.8e8).
// The kernel below is round 2's.
#include "mdc_internal.h"

#include <cmath>

namespace mdc {

namespace {

constexpr int kQC = 3;          // classes of the deployed nets (mdc_create admits no other)

__device__ __forceinline__ int wrap18(int v) { return (v << 14) >> 14; }
// {m[35], m[28:12]} of the 36-bit wire m = a*b + c*d, as a signed 18-bit value.  The sign is BIT 35 of the wrapped sum,
// not the sign of the unwrapped value: they differ exactly when a*b + c*d = +2^35 (all four operands -2^17).
__device__ __forceinline__ int select18(long long m) {
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    return (int)((lo >> 12) & 0x1FFFFu) - (int)(((hi >> 3) & 1u) << 17);
}
__device__ __forceinline__ int pair18(int a, int b, int c, int d) {      // select18(a*b + c*d), exact
    return select18((long long)a * b + (long long)c * d);
}
__device__ __forceinline__ int quant(float v) {                  // float2fix: truncate toward zero, wrap to 18 bits
    return wrap18((int)truncf(v * 4096.f));
}

// x: (n,2,128) f32 (quantised on load) or int32 Q6.12 words; tab: [conv k0[F], k1[F], b[F]] [dense bias C] pad to 64, then
// the position-0 weights [h][f][c], then the per-lane tables [slot][lane]; dense (n,C) / labels (n): each may be NULL
struct QParams { const void* x; int x_is_q; long n; const int* tab; int* dense; int* labels; };

// Registers: the 3-filter kernel fits two waves per SIMD with room to spare (104 VGPRs).  The 10-filter one kept 120 weight
// registers per lane and, under the 256-register budget of two waves per SIMD, spilled 22 VGPRs to scratch (84 B per
// lane, round 4).  Round 5: one wave per SIMD removes the spill but costs the latency hiding (2.92 ms per 2^20 frames
// against 2.2 ms at best, profiles/r05_q612_occupancy1_ab.log), so instead the second position slot's 60 weights live in
// LDS ([entry][lane] per wave: conflict-free ds_read_b32, 15 KiB per wave) and the kernel keeps two waves per SIMD.
template <int F>
__global__ __launch_bounds__(256, 2) void deployed_q612_kernel(const void* __restrict__ px, int x_is_q, long pn, const int* __restrict__ ptab,
                                                               int* __restrict__ pdense, int* __restrict__ plabels) {
    const QParams p{px, x_is_q, pn, ptab, pdense, plabels};
    constexpr int kW0 = 64;                          // position-0 weights [h][f][c]
    constexpr int kLaneTab = kW0 + 2 * F * kQC;      // then [slot = ((s*F + f)*C + c)*2 + h][64 lanes]
    const int lane = threadIdx.x & 63;
    const int* k0 = p.tab;
    const int* k1 = p.tab + F;
    const int* cb = p.tab + 2 * F;
    const int* db = p.tab + 3 * F;
    constexpr bool WLDS = F == 10;                   // slot 1's weights in LDS (see above)
    constexpr int kRegSlots = WLDS ? 1 : 2;
    __shared__ int sw[WLDS ? 4 * F * kQC * 2 * 64 : 1];      // [wave][(f*C + c)*2 + h][lane]
    int* swv = sw + (WLDS ? (threadIdx.x >> 6) * (F * kQC * 2 * 64) : 0);
    int wd[kRegSlots][F][kQC][2];                    // [position slot][filter][class][row table I/Q]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int c = 0; c < kQC; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int v = p.tab[kLaneTab + ((((s * F + f) * kQC + c) * 2 + h) << 6) + lane];
                    if (s < kRegSlots) wd[s][f][c][h] = v;
                    else swv[(((f * kQC + c) * 2 + h) << 6) + lane] = v;      // (read back by this lane only: no barrier)
                }
    auto wq = [&](int s, int f, int c, int h) -> int { return s < kRegSlots ? wd[s < kRegSlots ? s : 0][f][c][h] : swv[(((f * kQC + c) * 2 + h) << 6) + lane]; };

    const long nblk = (p.n + 63) >> 6;
    const long nwaves = (long)gridDim.x * 4;
    auto load = [&](long fr, int h) -> int2 {        // quantised x[h][2l], x[h][2l+1] of frame fr (clamped: never stored past n)
        const long f = fr < p.n ? fr : p.n - 1;
        const long idx = f * kFrameFloats + h * kSamples + 2 * lane;
        if (p.x_is_q) {
            const int2 v = *reinterpret_cast<const int2*>(static_cast<const int*>(p.x) + idx);
            return make_int2(wrap18(v.x), wrap18(v.y));
        }
        const float2 v = *reinterpret_cast<const float2*>(static_cast<const float*>(p.x) + idx);
        return make_int2(quant(v.x), quant(v.y));
    };
    for (long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += nwaves) {
        const long base = blk << 6;
        const int cnt = (int)((p.n - base) < 64 ? (p.n - base) : 64);
        int keep[kQC] = {0, 0, 0};                   // lane i: class sums of frame base + i over positions 1..128
        int x00 = 0, x10 = 0;                        // lane i: quantised x[0][0], x[1][0] of frame base + i
        int2 nI = load(base, 0), nQ = load(base, 1);
        for (int it = 0; it < cnt; ++it) {
            if (WLDS) asm volatile("" ::: "memory");      // keep the LDS weight reads inside the loop (hoisted, they are 60 registers again)
            const int2 xi = nI, xq = nQ;
            nI = load(base + it + 1, 0);             // one frame ahead
            nQ = load(base + it + 1, 1);
            // x[2l+2] = first sample of the next lane; lane 63 gets 0 = the right zero pad x[128]
            const int xi2 = __builtin_amdgcn_update_dpp(0, xi.x, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
            const int xq2 = __builtin_amdgcn_update_dpp(0, xq.x, 0x130, 0xf, 0xf, true);
            const int xs[2][3] = {{xi.x, xi.y, xi2}, {xq.x, xq.y, xq2}};
            unsigned acc[kQC] = {0u, 0u, 0u};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    int ai = wrap18(pair18(xs[0][s], k0[f], xs[0][s + 1], k1[f]) + cb[f]);
                    int aq = wrap18(pair18(xs[1][s], k0[f], xs[1][s + 1], k1[f]) + cb[f]);
                    ai = ai < 0 ? 0 : ai;
                    aq = aq < 0 ? 0 : aq;
#pragma unroll
                    for (int c = 0; c < kQC; ++c)
                        acc[c] += (unsigned)pair18(ai, wq(s, f, c, 0), aq, wq(s, f, c, 1));      // sign-extended 18-bit term, 32-bit wrap
                }
            // wave sum (wrap-around adds commute and associate: any order gives the same bits): a DPP butterfly over each 16-lane
            // row, then the four row sums through scalar registers -- __shfl_xor would be six dependent ds_bpermute_b32 per class
#pragma unroll
            for (int c = 0; c < kQC; ++c) {
                int v = (int)acc[c];
                v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x4E /*quad_perm [2,3,0,1]*/, 0xf, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x141 /*row_half_mirror*/, 0xf, 0xf, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x140 /*row_mirror*/, 0xf, 0xf, true);
                const unsigned tot = (unsigned)__builtin_amdgcn_readlane(v, 0) + (unsigned)__builtin_amdgcn_readlane(v, 16) +
                                     (unsigned)__builtin_amdgcn_readlane(v, 32) + (unsigned)__builtin_amdgcn_readlane(v, 48);
                keep[c] = (lane == it) ? (int)tot : keep[c];
            }
            const int s0 = __builtin_amdgcn_readfirstlane(xi.x), s1 = __builtin_amdgcn_readfirstlane(xq.x);
            x00 = (lane == it) ? s0 : x00;
            x10 = (lane == it) ? s1 : x10;
        }
        // ---- all lanes: frame base + lane.  Position w = 0 of both rows (x[-1] = 0): uniform weights
        unsigned tot[kQC] = {(unsigned)keep[0], (unsigned)keep[1], (unsigned)keep[2]};
#pragma unroll
        for (int f = 0; f < F; ++f) {
            int ai = wrap18(pair18(0, k0[f], x00, k1[f]) + cb[f]);
            int aq = wrap18(pair18(0, k0[f], x10, k1[f]) + cb[f]);
            ai = ai < 0 ? 0 : ai;
            aq = aq < 0 ? 0 : aq;
#pragma unroll
            for (int c = 0; c < kQC; ++c)
                tot[c] += (unsigned)pair18(ai, p.tab[kW0 + (0 * F + f) * kQC + c], aq, p.tab[kW0 + (1 * F + f) * kQC + c]);
        }
        if (lane < cnt) {
            const long o = base + lane;
            int best = 0, bestv = 0;
#pragma unroll
            for (int c = 0; c < kQC; ++c) {
                int s = (int)(tot[c] + (unsigned)db[c]);
                s = s < 0 ? 0 : s;
                if (p.dense) p.dense[o * kQC + c] = s;
                if (c == 0 || s > bestv) { bestv = s; best = c; }      // strict >: the first maximum wins (cnn.py:209)
            }
            if (p.labels) p.labels[o] = best;
        }
    }
}

inline int host_quant(float v) {
    long long q = (long long)std::trunc((double)v * 4096.0);
    q = ((q + (1 << 17)) & ((1 << 18) - 1)) - (1 << 17);
    return (int)q;
}

}  // namespace

// d_pack slot 1 of a deployed model: the integer tables (built at finalize from the float weights, which are exact
// multiples of 2^-12 when they came from a .txt table)
int deployed_q612_pack(mdc_model* m) {
    const int F = m->topo.filters, C = m->topo.classes;
    if (C != kQC) return MDC_OK;      // no integer path for such a net; mdc_forward_q612 reports it
    const int kW0 = 64, kLaneTab = kW0 + 2 * F * C;
    std::vector<int> tab((size_t)kLaneTab + (size_t)2 * F * C * 2 * 64, 0);
    const float* ck = m->hk[0].data();   // HWIO (1,2,1,F)
    for (int f = 0; f < F; ++f) {
        tab[f] = host_quant(ck[f]);
        tab[F + f] = host_quant(ck[F + f]);
        tab[2 * F + f] = host_quant(m->hb[0][f]);
    }
    for (int c = 0; c < C; ++c) tab[3 * F + c] = host_quant(m->hb[1][c]);
    const float* dk = m->hk[1].data();   // (258F, C), rows h*129F + w*F + f
    auto wq = [&](int h, int w, int f, int c) { return host_quant(dk[((size_t)h * 129 * F + (size_t)w * F + f) * C + c]); };
    for (int h = 0; h < 2; ++h)
        for (int f = 0; f < F; ++f)
            for (int c = 0; c < C; ++c) tab[kW0 + (h * F + f) * C + c] = wq(h, 0, f, c);
    for (int s = 0; s < 2; ++s)
        for (int f = 0; f < F; ++f)
            for (int c = 0; c < C; ++c)
                for (int h = 0; h < 2; ++h)
                    for (int lane = 0; lane < 64; ++lane)
                        tab[kLaneTab + (size_t)((((s * F + f) * C + c) * 2 + h) << 6) + lane] = wq(h, 2 * lane + 1 + s, f, c);
    return upload(m, 1, tab.data(), tab.size() * sizeof(int));
}

int deployed_q612_forward(const mdc_model* m, const void* x, int x_is_q, int64_t n, int32_t* dense, int32_t* labels, hipStream_t s) {
    const int F = m->topo.filters, C = m->topo.classes;
    if (C != kQC || 3 * F + C > 64) { set_error("Q6.12 path is built for %d classes (got %d)", kQC, C); return MDC_ENOTSUP; }
    if (!m->d_pack[1]) { set_error("Q6.12 tables missing"); return MDC_ESTATE; }
    if (n == 0) return MDC_OK;
    const int* tab = static_cast<const int*>(m->d_pack[1]);
    long grid = ((n + 63) / 64 + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (F == 3) hipLaunchKernelGGL(deployed_q612_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, x, x_is_q, (long)n, tab, dense, labels);
    else if (F == 10) hipLaunchKernelGGL(deployed_q612_kernel<10>, dim3((unsigned)grid), dim3(256), 0, s, x, x_is_q, (long)n, tab, dense, labels);
    else { set_error("Q6.12 path is built for F = 3 and F = 10 (got %d)", F); return MDC_ENOTSUP; }
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
