// Pieces shared by the asm-sequenced VT-CNN2 conv kernels (vtcnn2_bf16_sched.hip, vtcnn2_fp8_conv.hip): the
// per-lane image of a 16-frame group and its staging, the conv1 operand reads and MFMA, the partial-sum exchange
// and the one-VALU-at-a-time finish.  Each helper is ONE asm instruction (or a fixed short sequence) so that the
// kernels' step functions can place it in a chosen MFMA gap.  S is the kernel's state struct; the members used
// here are: A1, X, rp, rc, L0, L1, cb, wr_addr, rd_addr, rc_addr, im_addr, tprev.
#pragma once
#include "vtcnn2_bf16_common.h"

#include <type_traits>
#include <utility>

namespace mdc {

namespace {

constexpr int kS = 140;                   // image words per lane: entries 0..67 (66, 67 zero) + 4 pad
constexpr int kSImgWords = 64 * kS;       // [lane][kS] per buffer
// Batches up to this size run the conv kernels' position-range form: 11 work-groups per 16-frame group, each walking 14 of
// the 130 steps (a group is otherwise ONE work-group on one CU for ~100 us while the chip idles).  Whole forward, bf16,
// range vs batch form: 60 vs 119 us at n = 1, 163 vs 189 us at 1,024, 284 vs 260 us at 2,048 (tools/latency.py)
constexpr long kConvRangeFrames = 1024;
constexpr size_t kSchedLds = (size_t)2 * kSImgWords * 4 + (size_t)2 * kPartFloats * 4;      // 112,640 B

using f32x16 = __attribute__((ext_vector_type(16))) float;

// conv1 operand words: entries (e, e+1) of the image (16-B aligned when e is even) / entries (e, e+1) at 8-B alignment
template <int SLOT, class S>
__device__ __forceinline__ void sch_load_even(S& st, unsigned entry_addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(st.L0[SLOT]) : "v"(entry_addr) : "memory");
}
template <class S>
__device__ __forceinline__ void sch_load_odd(S& st, unsigned entry_addr) {
    asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(st.L1) : "v"(entry_addr) : "memory");
}
// B operand dword I (0..3) of an odd position p (R = p&3; chunk c = p>>2 in slot S0, chunk c+1 in slot SN): the words
// of entries i, i+1, i+2 (i = p>>1) shifted by one sample.  Even positions use L0[S0] (R=0) or L1 (R=2) as loaded.
template <int R, int S0, int SN, int I, class S>
__device__ __forceinline__ void sch_prep(S& st) {
    if constexpr (R == 1 || R == 3) {
        // entries i, i+1, i+2 as (A, B) word pairs: e0 = {w[0], w[1]}, e1 = {w[2], w[3]}, e2 = {w[4], w[5]}
        const unsigned w0 = R == 1 ? st.L0[S0][0] : st.L1[0], w1 = R == 1 ? st.L0[S0][1] : st.L1[1];
        const unsigned w2 = R == 1 ? st.L0[S0][2] : st.L1[2], w3 = R == 1 ? st.L0[S0][3] : st.L1[3];
        const unsigned w4 = R == 1 ? st.L1[2] : st.L0[SN][2], w5 = R == 1 ? st.L1[3] : st.L0[SN][3];
        unsigned& d = st.cb[I];
        if constexpr (I == 0) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w2), "v"(w0));
        else if constexpr (I == 1) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w3), "v"(w1));
        else if constexpr (I == 2) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w4), "v"(w2));
        else asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w5), "v"(w3));
    }
}
template <int R, int S0, int CT, class S>
__device__ __forceinline__ void sch_conv1_mfma(S& st) {
    const u32x4 b = R == 0 ? st.L0[S0] : R == 2 ? st.L1 : u32x4{st.cb[0], st.cb[1], st.cb[2], st.cb[3]};
    // "=&v": the result must not share registers with an operand
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(st.X[CT]) : "v"(st.A1[CT]), "v"(b));
}
template <int PB, int OT, class S>
__device__ __forceinline__ void sch_part_write(S& st, const f32x4& a) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(st.wr_addr), "a"(a), "i"(PB * kPartFloats * 4 + OT * 1024) : "memory");
}
// read R (0..7) of the owner's share of partial(v): even = float4 of wave R/2's partial of tile q, odd = its
// word of tile 4
template <int PB, int R, class S>
__device__ __forceinline__ void sch_red_load1(S& st) {
    constexpr int K = R >> 1;
    if constexpr ((R & 1) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(st.rp[K]) : "v"(st.rd_addr), "i"(PB * kPartFloats * 4 + K * 5120) : "memory");
    else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(st.rc[K]) : "v"(st.rc_addr), "i"(PB * kPartFloats * 4 + K * 5120) : "memory");
}
template <int PB, class S>
__device__ __forceinline__ void sch_red_load(S& st) {
    sch_red_load1<PB, 0>(st); sch_red_load1<PB, 1>(st); sch_red_load1<PB, 2>(st); sch_red_load1<PB, 3>(st);
    sch_red_load1<PB, 4>(st); sch_red_load1<PB, 5>(st); sch_red_load1<PB, 6>(st); sch_red_load1<PB, 7>(st);
}
// wait for every LDS operation of this wave issued so far; names the values the waited reads produce so that
// no consumer can be scheduled above it
template <class S>
__device__ __forceinline__ void sch_wait_lds(S& st) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(st.rp[0]), "+v"(st.rp[1]), "+v"(st.rp[2]), "+v"(st.rp[3]), "+v"(st.rc[0]), "+v"(st.rc[1]), "+v"(st.rc[2]),
                   "+v"(st.rc[3]), "+v"(st.L0[0]), "+v"(st.L0[1]), "+v"(st.L0[2]), "+v"(st.L1)
                 :: "memory");
}
// finish of one output position, one VALU instruction per call: sum of the 4 partials (the bias is already in wave
// 0's), ReLU, bf16.  Plain v_add_f32: v_pk_add_f32 costs a whole MFMA gap.
struct FinOut { unsigned o0, o1, tt; };      // tt: channel 64 + 4q + gs of the EVEN position (low half) and the odd one (high half)
struct FinTmp { float s[4], u[4], a, b, t; };
// kFeatShift: the bf16 mode keeps its conv1 activations and conv2 features multiplied by 2^-kFeatShift (round 3).  A power
// of two costs nothing -- conv1's taps and bias and conv2's bias carry 2^-kFeatShift, dense1's weights 2^+kFeatShift, all
// exact in bf16 and f32, every product and sum the same bits as before -- and buys the ReLU: v_cvt_pk_bf16_f32 with the
// VOP3 clamp bit converts AND clamps to [0, 1] (tools/microbench/cvt_bf16_clamp_probe.hip: identical to v_cvt_pk_bf16_f32
// + v_pk_max_i16 for every input below 1, -x -> +0), so ReLU + pack is ONE instruction per pair as long as the values
// stay below 1, i.e. the true activations below 2^32 -- 4.3e9, eleven decades above I/Q samples of order 1e-2; beyond it
// they saturate (a NaN becomes 0).  19 fewer VALU per position step and wave (16 in conv1's pack, 3 in the finish).
constexpr int kFeatShift = 32;

// the clamp form of the finish (K = 0..17; values below 1, see kFeatShift) drops the three v_pk_max_i16.  ODD = parity of
// the output position: the fifth tile's value of an even position waits in st.tprev (f32) for its odd neighbour and the
// two are packed, clamped and stored together (feat16_index: one dword per pair) -- an even position has 17 instructions.
template <int K, int ODD, class S>
__device__ __forceinline__ void sch_fin_clamp(S& st, FinTmp& f, FinOut& out) {
    if constexpr (K < 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.s[K]) : "v"(st.rp[0][K]), "v"(st.rp[1][K]));
    else if constexpr (K < 8) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.u[K - 4]) : "v"(st.rp[2][K - 4]), "v"(st.rp[3][K - 4]));
    else if constexpr (K < 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f.s[K - 8]) : "v"(f.u[K - 8]));
    else if constexpr (K == 12) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(out.o0) : "v"(f.s[0]), "v"(f.s[1]));
    else if constexpr (K == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(out.o1) : "v"(f.s[2]), "v"(f.s[3]));
    else if constexpr (K == 14) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.a) : "v"(st.rc[0]), "v"(st.rc[1]));
    else if constexpr (K == 15) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.b) : "v"(st.rc[2]), "v"(st.rc[3]));
    else if constexpr (K == 16) {
        if constexpr (ODD) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.t) : "v"(f.a), "v"(f.b));
        else asm volatile("v_add_f32 %0, %1, %2" : "=v"(st.tprev) : "v"(f.a), "v"(f.b));
    } else if constexpr (ODD) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(out.tt) : "v"(st.tprev), "v"(f.t));
}
template <int ODD, class S>
__device__ __forceinline__ void sch_fin_all_clamp(S& st, FinOut& out) {
    FinTmp f;
    [&]<int... K>(std::integer_sequence<int, K...>) { (sch_fin_clamp<K, ODD>(st, f, out), ...); }(std::make_integer_sequence<int, 18>{});
}

// The finishing lane layout is TRANSPOSED with respect to the MFMA layout: lane L finishes frame L>>2, channel
// chunk gs = L&3, so the four lanes of a quad write 32 (and 16) contiguous bytes of one frame's row.
// Wave q stores channels [16q+4gs, +4) of output position w (WHICH = 0: 8 bytes) and, after an ODD position only,
// channel 64+4q+gs of positions w-1 and w (WHICH = 1: one dword) of its lane's frame (row frow); feat16_index layout.
template <int WHICH>
__device__ __forceinline__ void sch_store(const FinOut& fo, unsigned short* frow, int w, int q, int gs) {
    unsigned short* pair = frow + (long)(w >> 1) * (2 * kC2);
    if constexpr (WHICH == 0) *reinterpret_cast<u32x2*>(pair + (w & 1) * 64 + 16 * q + 4 * gs) = u32x2{fo.o0, fo.o1};
    else *reinterpret_cast<unsigned*>(pair + 128 + 2 * (4 * q + gs)) = fo.tt;
}

// staging of a quarter (k) of a 16-frame group into this kernel's image layout; see stage_load in the common header
__device__ __forceinline__ void sch_stage_write(int k, float4 v, long n, long frame0, unsigned* __restrict__ im, int tid) {
    const int idx = tid + 256 * k;
    const int i = idx >> 6, l = idx & 63;
    if (frame0 + i >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);      // frames past the end of the batch are zeros
    const int h = l >> 5, m = l & 31;
    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float a = xs[2 * e], b = xs[2 * e + 1];
        const unsigned hi = pack2(a, b);
        const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xFFFF0000u);
        const unsigned lo = pack2(a - ah, b - bh);
        unsigned* d = im + (i + 16 * h) * kS + 2 * (2 * m + 1 + e);      // samples 4m+2e, +1 -> padded pair 2m+1+e
        *reinterpret_cast<u32x2*>(d) = u32x2{hi, lo};      // lower k-half: (x_hi, x_lo)
        d[32 * kS] = hi;                                   // upper k-half: x_hi (its second word stays (1,1))
    }
}


}  // namespace

}  // namespace mdc
