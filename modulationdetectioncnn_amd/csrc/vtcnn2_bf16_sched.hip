// Canonical VT-CNN2 (T3), bf16 path: the production conv1+conv2 kernel (see vtcnn2_bf16.hip for the algorithm).
#include "vtcnn2_bf16_common.h"

#include <cstdlib>
#include <type_traits>

namespace mdc {

namespace {

// ------------------------------------------------------------------------------------
// vt_conv_bf16_sched_kernel: the same algorithm and data layout as vt_conv_bf16_kernel, but every
// instruction of the position step is an `asm volatile` statement, so the ORDER is the one written
// here (hipcc only allocates registers).  One wave per SIMD issues in order: each of the ~75 non-MFMA
// instructions of a step must sit in the shadow of one of its 68 MFMAs or it is exposed.
// Order of a step v (accumulators: a0 = output v+2, fresh; a1 = v+1; a2 = v, completes):
//   A  tap 2 (rows 1 then 0)  + pack of conv1 row 0 (-> Bf[0], used by the second half of A)
//                              + LDS reads of the conv1 operands of step v+1
//   B  lgkmcnt(0); conv1(v+1) (8 MFMAs); tap 1 + the 5 ds_writes of a2 + finish of output v-1
//   C  lgkmcnt(0); s_barrier; 8 ds_reads of partial(v); tap 0 (rows 1 then 0) + pack of conv1 row 1
// Hazards hipcc would not see inside asm, and how the order guarantees them:
//   VALU write -> MFMA read of Bf (2 wait states): a pack half is always >= 1 MFMA before its first reader;
//   MFMA write -> VALU/DS read (<= 11 wait states for these shapes): every reader is >= 4 MFMAs later;
//   ds_write source vs later MFMA overwrite: a2 is kept alive until after the barrier's lgkmcnt(0).
// ------------------------------------------------------------------------------------
constexpr int kNV = 36;                   // conv2 fragments kept in VGPRs; the other 24 live in AGPRs

struct SchedState {
    u32x4 Wv[kNV];
    u32x4 Wa[kWFrags - kNV];
    u32x2 A1[4];
    unsigned Bf[2][2][4];     // B operands of conv2 as scalars (asm outputs cannot name vector elements)
    f32x4 X[4][2];
    f32x4 rp[4];
    float rc[4];
    unsigned bw[2][3];
    unsigned cb[2][2];
    f32x2 bq01, bq23;
    float b4q;
    unsigned wr_addr, rd_addr, rc_addr, im_addr;    // LDS byte addresses (lane part)
};

template <int IDX>
__device__ __forceinline__ void sch_mfma(SchedState& st, f32x4& acc, const u32x4& b) {
    if constexpr (IDX < kNV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(st.Wv[IDX]), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(st.Wa[IDX - kNV]), "v"(b));
}
template <int IDX>
__device__ __forceinline__ void sch_mfma_fresh(SchedState& st, f32x4& acc, const u32x4& b) {
    if constexpr (IDX < kNV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "v"(st.Wv[IDX]), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(st.Wa[IDX - kNV]), "v"(b));
}
template <int J, int H, int CP, int OT, bool FRESH = false>
__device__ __forceinline__ void sch_tap(SchedState& st, f32x4 (&acc)[5]) {
    constexpr int IDX = ((H * 3 + J) * 2 + CP) * 5 + OT;
    const u32x4 b = u32x4{st.Bf[H][CP][0], st.Bf[H][CP][1], st.Bf[H][CP][2], st.Bf[H][CP][3]};
    if constexpr (FRESH) sch_mfma_fresh<IDX>(st, acc[OT], b);
    else sch_mfma<IDX>(st, acc[OT], b);
}
// half a pack unit: two conv1 values -> ReLU -> one packed bf16 pair of the B operand (2 VALU)
template <int H, int CP, int T, int HALF>
__device__ __forceinline__ void sch_pack(SchedState& st) {
    unsigned& d = st.Bf[H][CP][2 * T + HALF];
    const float lo = st.X[2 * CP + T][H][2 * HALF], hi = st.X[2 * CP + T][H][2 * HALF + 1];
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(d) : "v"(lo), "v"(hi));
}
template <int H, int K>
__device__ __forceinline__ void sch_oper_load(SchedState& st, int pair_off) {      // one conv1 operand word
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(st.bw[H][K]) : "v"(st.im_addr + pair_off), "i"((H * kPairs + K) * 256));
}
// conv1 B operand of row H (pairs i, i+1; odd positions start one sample later: v_alignbit)
template <int PAR, int H>
__device__ __forceinline__ void sch_conv1_operand(SchedState& st) {
    if constexpr (PAR == 0) { st.cb[H][0] = st.bw[H][0]; st.cb[H][1] = st.bw[H][1]; }
    else {
        asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(st.cb[H][0]) : "v"(st.bw[H][1]), "v"(st.bw[H][0]));
        asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(st.cb[H][1]) : "v"(st.bw[H][2]), "v"(st.bw[H][1]));
    }
}
template <int H, int CT>
__device__ __forceinline__ void sch_conv1_mfma(SchedState& st) {
    const u32x2 b = u32x2{st.cb[H][0], st.cb[H][1]};
    // "=&v": the result must not share registers with an operand
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=&v"(st.X[CT][H]) : "v"(st.A1[CT]), "v"(b));
}
template <int PAR, int H>
__device__ __forceinline__ void sch_conv1(SchedState& st) {      // un-interleaved form (prologue only)
    sch_conv1_operand<PAR, H>(st);
    asm volatile("s_nop 1");
    sch_conv1_mfma<H, 0>(st); sch_conv1_mfma<H, 1>(st); sch_conv1_mfma<H, 2>(st); sch_conv1_mfma<H, 3>(st);
}
template <int PB, int OT>
__device__ __forceinline__ void sch_part_write(SchedState& st, const f32x4& a) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(st.wr_addr), "a"(a), "i"(PB * kPartFloats * 4 + OT * 1024) : "memory");
}
template <int PB, int K>
__device__ __forceinline__ void sch_red_load1(SchedState& st) {      // partial K of this wave's tile + its tile-4 component
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(st.rp[K]) : "v"(st.rd_addr), "i"(PB * kPartFloats * 4 + K * 5120) : "memory");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(st.rc[K]) : "v"(st.rc_addr), "i"(PB * kPartFloats * 4 + K * 5120) : "memory");
}
template <int PB>
__device__ __forceinline__ void sch_red_load(SchedState& st) {
    sch_red_load1<PB, 0>(st); sch_red_load1<PB, 1>(st); sch_red_load1<PB, 2>(st); sch_red_load1<PB, 3>(st);
}
// wait for every LDS operation of this wave issued so far; names the values the waited reads produce so that
// no consumer can be scheduled above it
__device__ __forceinline__ void sch_wait_lds(SchedState& st) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(st.rp[0]), "+v"(st.rp[1]), "+v"(st.rp[2]), "+v"(st.rp[3]), "+v"(st.rc[0]), "+v"(st.rc[1]), "+v"(st.rc[2]),
                   "+v"(st.rc[3]), "+v"(st.bw[0][0]), "+v"(st.bw[0][1]), "+v"(st.bw[0][2]), "+v"(st.bw[1][0]), "+v"(st.bw[1][1]), "+v"(st.bw[1][2])
                 :: "memory");
}
// finish of one output position: sum of the 4 partials, bias, ReLU, bf16 (values only; the stores are C++)
struct FinOut { unsigned o0, o1; unsigned short t16; };
struct FinTmp { f32x2 s01, s23, u0, u1; float t; };
template <int PART>   // six parts of 2-4 VALU each, to be spread between MFMAs
__device__ __forceinline__ void sch_finish(SchedState& st, FinTmp& f, FinOut& out) {
#define LO(v) __builtin_shufflevector(v, v, 0, 1)
#define HI(v) __builtin_shufflevector(v, v, 2, 3)
    if constexpr (PART == 0) {
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.s01) : "v"(LO(st.rp[0])), "v"(LO(st.rp[1])));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.s23) : "v"(HI(st.rp[0])), "v"(HI(st.rp[1])));
    } else if constexpr (PART == 1) {
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.u0) : "v"(LO(st.rp[2])), "v"(LO(st.rp[3])));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.u1) : "v"(HI(st.rp[2])), "v"(HI(st.rp[3])));
    } else if constexpr (PART == 2) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f.s01) : "v"(f.u0));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f.s23) : "v"(f.u1));
    } else if constexpr (PART == 3) {
        float a, b;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(st.rc[0]), "v"(st.rc[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(st.rc[2]), "v"(st.rc[3]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.t) : "v"(a), "v"(b));
    } else if constexpr (PART == 4) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f.s01) : "v"(st.bq01));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f.s23) : "v"(st.bq23));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(f.t) : "v"(st.b4q));
    } else {
        const float s0 = f.s01[0], s1 = f.s01[1], s2 = f.s23[0], s3 = f.s23[1];
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(out.o0) : "v"(s0), "v"(s1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(out.o1) : "v"(s2), "v"(s3));
        unsigned tt;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1\n\tv_pk_max_i16 %0, %0, 0" : "=v"(tt) : "v"(f.t));
        out.t16 = (unsigned short)tt;
    }
#undef LO
#undef HI
}

// ABL: 0 = product; timing-only probes (tools/ablate_sched.py, -DMDC_ABLATIONS; results wrong; every MFMA stays):
//   1 no s_barrier   2 no exchange (ds_writes, barrier, reads)   3 no finish VALU / feature stores
//   4 no pack        5 no conv1 (operand reads + 8 MFMAs)        6 all of 2..5
template <int PAR, bool FIRST, bool LAST, int ABL = 0>   // PAR = v & 1
__device__ __forceinline__ void sch_step(SchedState& st, int v, int q, unsigned short* fbase,
                                         f32x4 (&a0)[5], f32x4 (&a1)[5], f32x4 (&a2)[5]) {
    constexpr int PN = 1 - PAR;            // parity of v + 1
    constexpr bool kExch = ABL != 2 && ABL != 6, kFin = ABL != 3 && ABL != 6, kPack = ABL != 4 && ABL != 6, kC1 = ABL != 5 && ABL != 6;
    const int pair_next = ((v + 1) >> 1) * 256;      // byte offset of pair (v+1)>>1 in the image row
    // ---------------- A: tap 2.  Row 1 first (its Bf was packed in C of the previous step), pack row 0 rides along
    sch_tap<2, 1, 0, 0>(st, a2); if (kPack) sch_pack<0, 0, 0, 0>(st);
    sch_tap<2, 1, 0, 1>(st, a2); if (kPack) sch_pack<0, 0, 0, 1>(st);
    sch_tap<2, 1, 0, 2>(st, a2); if (kPack) sch_pack<0, 0, 1, 0>(st);
    sch_tap<2, 1, 0, 3>(st, a2); if (kPack) sch_pack<0, 0, 1, 1>(st);
    sch_tap<2, 1, 0, 4>(st, a2); if (kPack) sch_pack<0, 1, 0, 0>(st);
    sch_tap<2, 1, 1, 0>(st, a2); if (kPack) sch_pack<0, 1, 0, 1>(st);
    sch_tap<2, 1, 1, 1>(st, a2); if (kPack) sch_pack<0, 1, 1, 0>(st);
    sch_tap<2, 1, 1, 2>(st, a2); if (kPack) sch_pack<0, 1, 1, 1>(st);
    sch_tap<2, 1, 1, 3>(st, a2); if (!LAST && kC1) { sch_oper_load<0, 0>(st, pair_next); sch_oper_load<0, 1>(st, pair_next); }
    sch_tap<2, 1, 1, 4>(st, a2); if (!LAST && kC1) { sch_oper_load<1, 0>(st, pair_next); sch_oper_load<1, 1>(st, pair_next); }
    sch_tap<2, 0, 0, 0>(st, a2); if (!LAST && kC1 && PN == 1) { sch_oper_load<0, 2>(st, pair_next); sch_oper_load<1, 2>(st, pair_next); }
    sch_tap<2, 0, 0, 1>(st, a2);
    sch_tap<2, 0, 0, 2>(st, a2);
    sch_tap<2, 0, 0, 3>(st, a2);
    sch_tap<2, 0, 0, 4>(st, a2);
    sch_tap<2, 0, 1, 0>(st, a2);
    sch_tap<2, 0, 1, 1>(st, a2);
    sch_tap<2, 0, 1, 2>(st, a2);
    sch_tap<2, 0, 1, 3>(st, a2);
    sch_tap<2, 0, 1, 4>(st, a2);
    // ---------------- B: conv1(v+1) with the finish of output v-1 in its shadows; tap 1 with the ds_writes of a2
    sch_wait_lds(st);                               // rp/rc of output v-1, conv1 operands of v+1: issued long ago
    FinTmp ft; FinOut fo;
    if (!LAST && !kC1) {
        if (!FIRST && kFin) { sch_finish<0>(st, ft, fo); sch_finish<1>(st, ft, fo); sch_finish<2>(st, ft, fo);
                              sch_finish<3>(st, ft, fo); sch_finish<4>(st, ft, fo); sch_finish<5>(st, ft, fo); }
    } else if (!LAST) {
        sch_conv1_operand<PN, 0>(st); sch_conv1_operand<PN, 1>(st);
        if (!FIRST && kFin) sch_finish<0>(st, ft, fo); else asm volatile("s_nop 1");
        sch_conv1_mfma<0, 0>(st); if (!FIRST && kFin) sch_finish<1>(st, ft, fo);
        sch_conv1_mfma<0, 1>(st); if (!FIRST && kFin) sch_finish<2>(st, ft, fo);
        sch_conv1_mfma<0, 2>(st); if (!FIRST && kFin) sch_finish<3>(st, ft, fo);
        sch_conv1_mfma<0, 3>(st); if (!FIRST && kFin) sch_finish<4>(st, ft, fo);
        sch_conv1_mfma<1, 0>(st); if (!FIRST && kFin) sch_finish<5>(st, ft, fo);
        sch_conv1_mfma<1, 1>(st);
        sch_conv1_mfma<1, 2>(st);
        sch_conv1_mfma<1, 3>(st);
    } else {
        sch_finish<0>(st, ft, fo); sch_finish<1>(st, ft, fo); sch_finish<2>(st, ft, fo);
        sch_finish<3>(st, ft, fo); sch_finish<4>(st, ft, fo); sch_finish<5>(st, ft, fo);
    }
    if (!FIRST && kFin) {
        unsigned short* dst = fbase + (long)(v - 1) * kC2;
        *reinterpret_cast<u32x2*>(dst + 16 * q) = u32x2{fo.o0, fo.o1};
        dst[64 + q] = fo.t16;
    }
    sch_tap<1, 1, 0, 0>(st, a1); if (kExch) sch_part_write<PAR, 0>(st, a2[0]);
    sch_tap<1, 1, 0, 1>(st, a1);
    sch_tap<1, 1, 0, 2>(st, a1); if (kExch) sch_part_write<PAR, 1>(st, a2[1]);
    sch_tap<1, 1, 0, 3>(st, a1);
    sch_tap<1, 1, 0, 4>(st, a1); if (kExch) sch_part_write<PAR, 2>(st, a2[2]);
    sch_tap<1, 1, 1, 0>(st, a1);
    sch_tap<1, 1, 1, 1>(st, a1); if (kExch) sch_part_write<PAR, 3>(st, a2[3]);
    sch_tap<1, 1, 1, 2>(st, a1);
    sch_tap<1, 1, 1, 3>(st, a1); if (kExch) sch_part_write<PAR, 4>(st, a2[4]);
    sch_tap<1, 1, 1, 4>(st, a1);
    sch_tap<1, 0, 0, 0>(st, a1);
    sch_tap<1, 0, 0, 1>(st, a1);
    sch_tap<1, 0, 0, 2>(st, a1);
    sch_tap<1, 0, 0, 3>(st, a1);
    sch_tap<1, 0, 0, 4>(st, a1);
    sch_tap<1, 0, 1, 0>(st, a1);
    sch_tap<1, 0, 1, 1>(st, a1);
    sch_tap<1, 0, 1, 2>(st, a1);
    sch_tap<1, 0, 1, 3>(st, a1);
    sch_tap<1, 0, 1, 4>(st, a1);
    // ---------------- C: tap 0 (fresh accumulators); the exchange hand-off (barrier) a few MFMAs in, so that the
    //                  ds_writes above have long completed when lgkmcnt(0) is asked for; then the reads of
    //                  partial(v) in the following shadows; pack of conv1 row 1
    sch_tap<0, 1, 0, 0, true>(st, a0);
    sch_tap<0, 1, 0, 1, true>(st, a0);
    sch_tap<0, 1, 0, 2, true>(st, a0);
    if (ABL == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (kExch) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // keep a2 allocated until here (see the ds_write / XDL hazard note above): its ds_writes have completed, and
    // no MFMA issued before this point can have been given its registers
    asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4]));
    sch_tap<0, 1, 0, 3, true>(st, a0); if (kExch) sch_red_load1<PAR, 0>(st);
    sch_tap<0, 1, 0, 4, true>(st, a0); if (kExch) sch_red_load1<PAR, 1>(st);
    sch_tap<0, 1, 1, 0>(st, a0); if (kExch) sch_red_load1<PAR, 2>(st);
    sch_tap<0, 1, 1, 1>(st, a0); if (kExch) sch_red_load1<PAR, 3>(st);
    sch_tap<0, 1, 1, 2>(st, a0);
    sch_tap<0, 1, 1, 3>(st, a0);
    sch_tap<0, 1, 1, 4>(st, a0);
    sch_tap<0, 0, 0, 0>(st, a0); if (!LAST && kPack) sch_pack<1, 0, 0, 0>(st);
    sch_tap<0, 0, 0, 1>(st, a0); if (!LAST && kPack) sch_pack<1, 0, 0, 1>(st);
    sch_tap<0, 0, 0, 2>(st, a0); if (!LAST && kPack) sch_pack<1, 0, 1, 0>(st);
    sch_tap<0, 0, 0, 3>(st, a0); if (!LAST && kPack) sch_pack<1, 0, 1, 1>(st);
    sch_tap<0, 0, 0, 4>(st, a0); if (!LAST && kPack) sch_pack<1, 1, 0, 0>(st);
    sch_tap<0, 0, 1, 0>(st, a0); if (!LAST && kPack) sch_pack<1, 1, 0, 1>(st);
    sch_tap<0, 0, 1, 1>(st, a0); if (!LAST && kPack) sch_pack<1, 1, 1, 0>(st);
    sch_tap<0, 0, 1, 2>(st, a0); if (!LAST && kPack) sch_pack<1, 1, 1, 1>(st);
    sch_tap<0, 0, 1, 3>(st, a0);
    sch_tap<0, 0, 1, 4>(st, a0);
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void vt_conv_bf16_sched_kernel(const float* __restrict__ x, long n,
                                                                    const u32x4* __restrict__ wq, const u32x2* __restrict__ a1q,
                                                                    const float* __restrict__ b2, unsigned short* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* img = reinterpret_cast<unsigned*>(smem);
    float* part = reinterpret_cast<float*>(smem + (size_t)2 * kImgWords * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, g = lane >> 4;

    SchedState st;
#pragma unroll
    for (int i = 0; i < kWFrags; ++i) {
        const u32x4 w = wq[(q * kWFrags + i) * 64 + lane];
        if (i < kNV) { st.Wv[i] = w; asm volatile("" : "+v"(st.Wv[i])); }
        else { st.Wa[i - kNV] = w; asm volatile("" : "+a"(st.Wa[i - kNV])); }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) st.A1[ct] = a1q[(q * 4 + ct) * 64 + lane];
    st.bq01 = *reinterpret_cast<const f32x2*>(b2 + 16 * q + 4 * g);
    st.bq23 = *reinterpret_cast<const f32x2*>(b2 + 16 * q + 4 * g + 2);
    st.b4q = b2[64 + 4 * g + q];
    asm volatile("" : "+v"(st.bq01), "+v"(st.bq23), "+v"(st.b4q));
    const unsigned part_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (size_t)2 * kImgWords * 4);
    const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    st.wr_addr = part_lds + (q * 5 * 64 + lane) * 16;
    st.rd_addr = part_lds + (q * 64 + lane) * 16;
    st.rc_addr = part_lds + (4 * 64 + lane) * 16 + q * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { st.rp[k] = f32x4{0.f, 0.f, 0.f, 0.f}; st.rc[k] = 0.f; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 3; ++k) st.bw[h][k] = 0u;

    for (int i = tid; i < 2 * kImgWords; i += 256) img[i] = ((i & 63) >= 48) ? 0x3F803F80u : 0u;
    __syncthreads();
    const long ngroups = (n + 15) >> 4;
    long grp = blockIdx.x;
    if (grp < ngroups)
        for (int k = 0; k < 4; ++k) stage_quarter(k, x, n, grp * 16, img, tid);
    __syncthreads();

    int buf = 0;
    for (; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        st.im_addr = img_lds + (buf * kImgWords + lane) * 4;
        const long fme = grp * 16 + nl;
        unsigned short* fbase = feat + fme * (long)(kW2 * kC2) + 4 * g;
        const long gnext = grp + gridDim.x;
        f32x4 acc[3][5];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        // prologue: conv1 of position 0 (both rows), pack of row 1 (row 0 is packed by step 0's phase A)
        sch_oper_load<0, 0>(st, 0); sch_oper_load<0, 1>(st, 0); sch_oper_load<1, 0>(st, 0); sch_oper_load<1, 1>(st, 0);
        sch_wait_lds(st);
        sch_conv1<0, 0>(st); sch_conv1<0, 1>(st);
        {
            f32x4 &x0 = st.X[0][1], &x1 = st.X[1][1], &x2 = st.X[2][1], &x3 = st.X[3][1];
            f32x4 &y0 = st.X[0][0], &y1 = st.X[1][0], &y2 = st.X[2][0], &y3 = st.X[3][0];
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
        }
        sch_pack<1, 0, 0, 0>(st); sch_pack<1, 0, 0, 1>(st); sch_pack<1, 0, 1, 0>(st); sch_pack<1, 0, 1, 1>(st);
        sch_pack<1, 1, 0, 0>(st); sch_pack<1, 1, 0, 1>(st); sch_pack<1, 1, 1, 0>(st); sch_pack<1, 1, 1, 1>(st);
        asm volatile("s_nop 1");

        sch_step<0, true, false, ABL>(st, 0, q, fbase, acc[2], acc[1], acc[0]);
        int v = 1;
        for (int it = 0; it < 21; ++it, v += 6) {     // v = 1 .. 126
            if (it >= 12 && it < 16 && gnext < ngroups)
                stage_quarter(it - 12, x, n, gnext * 16, img + (buf ^ 1) * kImgWords, tid);
            sch_step<1, false, false, ABL>(st, v + 0, q, fbase, acc[0], acc[2], acc[1]);
            sch_step<0, false, false, ABL>(st, v + 1, q, fbase, acc[1], acc[0], acc[2]);
            sch_step<1, false, false, ABL>(st, v + 2, q, fbase, acc[2], acc[1], acc[0]);
            sch_step<0, false, false, ABL>(st, v + 3, q, fbase, acc[0], acc[2], acc[1]);
            sch_step<1, false, false, ABL>(st, v + 4, q, fbase, acc[1], acc[0], acc[2]);
            sch_step<0, false, false, ABL>(st, v + 5, q, fbase, acc[2], acc[1], acc[0]);
        }
        sch_step<1, false, false, ABL>(st, 127, q, fbase, acc[0], acc[2], acc[1]);
        sch_step<0, false, false, ABL>(st, 128, q, fbase, acc[1], acc[0], acc[2]);
        sch_step<1, false, true, ABL>(st, 129, q, fbase, acc[2], acc[1], acc[0]);
        // tail: finish 129, then outputs 130 and 131 (complete as they are: only zero padding beyond)
        auto finish_store = [&](int w) {
            FinTmp ft; FinOut fo;
            sch_wait_lds(st);
            sch_finish<0>(st, ft, fo); sch_finish<1>(st, ft, fo); sch_finish<2>(st, ft, fo);
            sch_finish<3>(st, ft, fo); sch_finish<4>(st, ft, fo); sch_finish<5>(st, ft, fo);
            unsigned short* dst = fbase + (long)w * kC2;
            *reinterpret_cast<u32x2*>(dst + 16 * q) = u32x2{fo.o0, fo.o1};
            dst[64 + q] = fo.t16;
        };
        finish_store(129);
        asm volatile("s_nop 7\n\ts_nop 7");       // last tap-1/tap-0 MFMAs -> ds_write of their accumulators
        sch_part_write<0, 0>(st, acc[1][0]); sch_part_write<0, 1>(st, acc[1][1]); sch_part_write<0, 2>(st, acc[1][2]);
        sch_part_write<0, 3>(st, acc[1][3]); sch_part_write<0, 4>(st, acc[1][4]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        sch_red_load<0>(st);
        asm volatile("" ::"a"(acc[1][0]), "a"(acc[1][1]), "a"(acc[1][2]), "a"(acc[1][3]), "a"(acc[1][4]));
        finish_store(130);
        sch_part_write<1, 0>(st, acc[2][0]); sch_part_write<1, 1>(st, acc[2][1]); sch_part_write<1, 2>(st, acc[2][2]);
        sch_part_write<1, 3>(st, acc[2][3]); sch_part_write<1, 4>(st, acc[2][4]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        sch_red_load<1>(st);
        asm volatile("" ::"a"(acc[2][0]), "a"(acc[2][1]), "a"(acc[2][2]), "a"(acc[2][3]), "a"(acc[2][4]));
        finish_store(131);
        __syncthreads();      // next group's image is complete; partial buffers are free again
    }
}

}  // namespace

int vtcnn2_bf16_conv_sched(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s) {
    const long ngroups = (n + 15) / 16;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
#define MDC_LAUNCH_SCHED(A) do { \
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_bf16_sched_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvBf16Lds)); \
    hipLaunchKernelGGL(vt_conv_bf16_sched_kernel<A>, dim3(grid), dim3(256), kConvBf16Lds, s, x, (long)n, \
                       static_cast<const u32x4*>(m->d_pack[0]), static_cast<const u32x2*>(m->d_pack[1]), \
                       static_cast<const float*>(m->d_pack[2]), static_cast<unsigned short*>(feat)); } while (0)
#ifdef MDC_ABLATIONS   // timing-only variants for tools/ablate_sched.py (build with -DMDC_ABLATIONS); results are wrong
    static const int abl = getenv("MDC_ABLATE_S") ? atoi(getenv("MDC_ABLATE_S")) : 0;
    switch (abl) { case 1: MDC_LAUNCH_SCHED(1); break; case 2: MDC_LAUNCH_SCHED(2); break; case 3: MDC_LAUNCH_SCHED(3); break;
                   case 5: MDC_LAUNCH_SCHED(5); break; case 6: MDC_LAUNCH_SCHED(6); break; default: MDC_LAUNCH_SCHED(0); }
#else
    MDC_LAUNCH_SCHED(0);
#endif
#undef MDC_LAUNCH_SCHED
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
