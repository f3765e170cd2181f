// Canonical VT-CNN2 (T3), bf16 path: the production conv1+conv2 kernel (see vtcnn2_bf16.hip for the algorithm,
// the operand layouts and the hipcc-scheduled statement of the same computation).
#include "vtcnn2_bf16_common.h"

#include <cstdlib>
#include <type_traits>

namespace mdc {

namespace {

// ------------------------------------------------------------------------------------
// vt_conv_bf16_sched_kernel: every instruction of the position step is an `asm volatile` statement, so the
// ORDER is the one written here (hipcc only allocates registers).  One wave per SIMD issues in order, and a
// 16x16x32 MFMA keeps the SIMD's issue port for 8 of its 16 cycles: ONE 4-cycle VALU (or one LDS/VMEM issue)
// per MFMA gap is free, a second one starts to stretch the gap.  Measured on the first version of this
// kernel (2-4 VALU bunched in some gaps, none in others): every non-MFMA instruction cost its full issue time
// (tools/ablate_sched.py).  Hence this schedule: at most one VALU and one LDS instruction per gap, nothing in
// the short gaps of the K=16 conv1 MFMAs.
//
// Step v (accumulators: a0 = output v+2, fresh; a1 = v+1; a2 = v, completes), 68 MFMAs:
//   T2  tap 2, 20 MFMAs -> a2 complete.   gaps: finish of output v-1 (15 VALU: sum of the 4 partials, ReLU, bf16),
//                                          its 2 stores, v_alignbit of the conv1 operands (odd v+1)
//   C1  conv1(v+1), 8 K=16 MFMAs          gaps: nothing
//   T1  tap 1, 20 MFMAs                   gaps: 5 ds_write_b128 of a2 (partial(v)); 15 pack VALU of conv1(v+1)
//   T0  tap 0, 20 MFMAs (5 fresh, C = conv2 bias on wave 0)
//                                          gaps: 17 pack VALU; lgkmcnt(0)+s_barrier after the 3rd MFMA; 8 ds_reads of
//                                          partial(v); 4-6 ds_reads of the conv1 operands of v+2
// Bf (the packed ReLU'd conv1 output = B operand of conv2) is double-buffered by step parity, so the pack of
// step v+1 can trail the conv1 MFMAs anywhere in T1/T0 of step v.
// Hazards hipcc would not see inside asm, and how the order guarantees them:
//   VALU write -> MFMA read (2 wait states): Bf is written a phase before its first reader; cb (v_alignbit) one
//     MFMA before conv1;
//   MFMA write -> VALU/DS read (<= 11 wait states for these shapes): every reader is >= 4 MFMAs later;
//   an asm MFMA's result lands long after the statement: its destination must stay live until a reader
//     (a dead destination gets reallocated and clobbered: the "no pack" timing probe faulted that way);
//   ds_write source vs later MFMA overwrite (not interlocked for XDL writes): a2 is kept alive until after the
//     barrier's lgkmcnt(0).
// ------------------------------------------------------------------------------------
constexpr int kNV = 28;                   // conv2 fragments kept in VGPRs; the other 32 live in AGPRs

struct SchedState {
    u32x4 Wv[kNV];
    u32x4 Wa[kWFrags - kNV];
    f32x4 bias[5];            // conv2 bias tiles as the C operand of the fresh MFMAs (wave 0; zeros on waves 1-3)
    u32x2 A1[4];
    unsigned Bf[2][2][2][4];  // [step parity][row][channel pair][word]: B operands of conv2, as scalars
    f32x4 X[4][2];
    f32x4 rp[4];
    float rc[4];
    unsigned bw[2][3];
    unsigned cb[2][2];
    unsigned wr_addr, rd_addr, rc_addr, im_addr;    // LDS byte addresses (lane part)
    int gs;                                         // finishing role of this lane: channel chunk (lane & 3)
};

// conv2 MFMA number I (0..19) of tap J: row H = 1 - I/10, channel pair CP = (I/5)%2, output tile OT = I%5
template <int SP, int J, int I>
__device__ __forceinline__ void sch_tap(SchedState& st, f32x4 (&acc)[5]) {
    constexpr int H = 1 - I / 10, CP = (I / 5) % 2, OT = I % 5;
    constexpr int IDX = ((H * 3 + J) * 2 + CP) * 5 + OT;
    const u32x4 b = u32x4{st.Bf[SP][H][CP][0], st.Bf[SP][H][CP][1], st.Bf[SP][H][CP][2], st.Bf[SP][H][CP][3]};
    if constexpr (J == 0 && I < 5) {      // first MFMA of output v+2: C = bias (early-clobber: D must not alias an input)
        if constexpr (IDX < kNV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b), "a"(st.bias[OT]));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&a"(acc[OT]) : "a"(st.Wa[IDX - kNV]), "v"(b), "a"(st.bias[OT]));
    } else {
        if constexpr (IDX < kNV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[OT]) : "a"(st.Wa[IDX - kNV]), "v"(b));
    }
}
// pack instruction N (0..31) of conv1's X into Bf[SP]: unit k = N>>1 = (row, channel pair, tile, half);
// even N = v_cvt_pk_bf16_f32 of two channels, odd N = ReLU on the packed pair (negative bf16 <=> negative int16)
template <int SP, int N>
__device__ __forceinline__ void sch_packop(SchedState& st) {
    constexpr int k = N >> 1, H = k >> 3, CP = (k >> 2) & 1, T = (k >> 1) & 1, HALF = k & 1;
    unsigned& d = st.Bf[SP][H][CP][2 * T + HALF];
    if constexpr ((N & 1) == 0) {
        const float lo = st.X[2 * CP + T][H][2 * HALF], hi = st.X[2 * CP + T][H][2 * HALF + 1];
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
    } else {
        asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(d));
    }
}
template <int H, int K>
__device__ __forceinline__ void sch_oper_load(SchedState& st, int pair_off) {      // one conv1 operand word
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(st.bw[H][K]) : "v"(st.im_addr + pair_off), "i"((H * kPairs + K) * 256));
}
// conv1 B operand word W of row H at an odd position: start one sample later
template <int H, int W>
__device__ __forceinline__ void sch_align(SchedState& st) {
    asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(st.cb[H][W]) : "v"(st.bw[H][W + 1]), "v"(st.bw[H][W]));
}
template <int H, int CT>
__device__ __forceinline__ void sch_conv1_mfma(SchedState& st) {
    const u32x2 b = u32x2{st.cb[H][0], st.cb[H][1]};
    // "=&v": the result must not share registers with an operand
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=&v"(st.X[CT][H]) : "v"(st.A1[CT]), "v"(b));
}
template <int PB, int OT>
__device__ __forceinline__ void sch_part_write(SchedState& st, const f32x4& a) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(st.wr_addr), "a"(a), "i"(PB * kPartFloats * 4 + OT * 1024) : "memory");
}
// read R (0..7) of the owner's share of partial(v): even = float4 of wave R/2's partial of tile q, odd = its
// component q of tile 4
template <int PB, int R>
__device__ __forceinline__ void sch_red_load1(SchedState& st) {
    constexpr int K = R >> 1;
    if constexpr ((R & 1) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(st.rp[K]) : "v"(st.rd_addr), "i"(PB * kPartFloats * 4 + K * 5120) : "memory");
    else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(st.rc[K]) : "v"(st.rc_addr), "i"(PB * kPartFloats * 4 + K * 5120) : "memory");
}
template <int PB>
__device__ __forceinline__ void sch_red_load(SchedState& st) {
    sch_red_load1<PB, 0>(st); sch_red_load1<PB, 1>(st); sch_red_load1<PB, 2>(st); sch_red_load1<PB, 3>(st);
    sch_red_load1<PB, 4>(st); sch_red_load1<PB, 5>(st); sch_red_load1<PB, 6>(st); sch_red_load1<PB, 7>(st);
}
// wait for every LDS operation of this wave issued so far; names the values the waited reads produce so that
// no consumer can be scheduled above it
__device__ __forceinline__ void sch_wait_lds(SchedState& st) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(st.rp[0]), "+v"(st.rp[1]), "+v"(st.rp[2]), "+v"(st.rp[3]), "+v"(st.rc[0]), "+v"(st.rc[1]), "+v"(st.rc[2]),
                   "+v"(st.rc[3]), "+v"(st.bw[0][0]), "+v"(st.bw[0][1]), "+v"(st.bw[0][2]), "+v"(st.bw[1][0]), "+v"(st.bw[1][1]), "+v"(st.bw[1][2])
                 :: "memory");
}
// finish of one output position, one VALU instruction per call (K = 0..14): sum of the 4 partials (the bias is
// already in wave 0's), ReLU, bf16
struct FinOut { unsigned o0, o1, tt; };
struct FinTmp { f32x2 s01, s23, u0, u1; float a, b, t; };
template <int K>
__device__ __forceinline__ void sch_fin(SchedState& st, FinTmp& f, FinOut& out) {
#define LO(v) __builtin_shufflevector(v, v, 0, 1)
#define HI(v) __builtin_shufflevector(v, v, 2, 3)
    if constexpr (K == 0) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.s01) : "v"(LO(st.rp[0])), "v"(LO(st.rp[1])));
    else if constexpr (K == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.s23) : "v"(HI(st.rp[0])), "v"(HI(st.rp[1])));
    else if constexpr (K == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.u0) : "v"(LO(st.rp[2])), "v"(LO(st.rp[3])));
    else if constexpr (K == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(f.u1) : "v"(HI(st.rp[2])), "v"(HI(st.rp[3])));
    else if constexpr (K == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f.s01) : "v"(f.u0));
    else if constexpr (K == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f.s23) : "v"(f.u1));
    else if constexpr (K == 6) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.a) : "v"(st.rc[0]), "v"(st.rc[1]));
    else if constexpr (K == 7) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.b) : "v"(st.rc[2]), "v"(st.rc[3]));
    else if constexpr (K == 8) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.t) : "v"(f.a), "v"(f.b));
    else if constexpr (K == 9) { const float s0 = f.s01[0], s1 = f.s01[1]; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(out.o0) : "v"(s0), "v"(s1)); }
    else if constexpr (K == 10) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(out.o0));
    else if constexpr (K == 11) { const float s2 = f.s23[0], s3 = f.s23[1]; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(out.o1) : "v"(s2), "v"(s3)); }
    else if constexpr (K == 12) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(out.o1));
    else if constexpr (K == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(out.tt) : "v"(f.t));
    else asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(out.tt));
#undef LO
#undef HI
}
__device__ __forceinline__ void sch_fin_all(SchedState& st, FinOut& out) {
    FinTmp f;
    sch_fin<0>(st, f, out); sch_fin<1>(st, f, out); sch_fin<2>(st, f, out); sch_fin<3>(st, f, out); sch_fin<4>(st, f, out);
    sch_fin<5>(st, f, out); sch_fin<6>(st, f, out); sch_fin<7>(st, f, out); sch_fin<8>(st, f, out); sch_fin<9>(st, f, out);
    sch_fin<10>(st, f, out); sch_fin<11>(st, f, out); sch_fin<12>(st, f, out); sch_fin<13>(st, f, out); sch_fin<14>(st, f, out);
}
// The finishing lane layout is TRANSPOSED with respect to the MFMA layout: lane L finishes frame L>>2, channel
// chunk gs = L&3, so the four lanes of a quad write 32 (and 8) contiguous bytes of one frame's row and the
// address unit sees 16 transactions per store instruction instead of 64 (measured on the first version, where
// lane = frame + 16*chunk left adjacent lanes 21 KB apart: the two stores of a step cost 10 % of the kernel).
// Wave q stores channels [16q+4gs, +4) and channel 64+4q+gs of its lane's frame (row frow) at output position w.
__device__ __forceinline__ void sch_store(const FinOut& fo, unsigned short* frow, int w, int q, int gs) {
    unsigned short* dst = frow + (long)w * kC2;
    *reinterpret_cast<u32x2*>(dst + 16 * q + 4 * gs) = u32x2{fo.o0, fo.o1};
    dst[64 + 4 * q + gs] = (unsigned short)fo.tt;
}

// ABL: 0 = product; timing-only probes (tools/ablate_sched.py, -DMDC_ABLATIONS; results wrong; every conv2 MFMA stays):
//   1 no s_barrier   2 no exchange (ds_writes, barrier, reads)   3 no finish VALU / feature stores
//   5 no conv1 (operand reads, 8 MFMAs, pack)                     6 all of 2, 3, 5
//   7 no feature stores (finish VALU kept)   8 no ds_writes   9 no reads of the partials
//   10 no pack VALU (conv1 kept)             11 no conv1 operand reads / v_alignbit
template <int PAR, bool FIRST, bool LAST, bool LOADS, int ABL>   // PAR = v & 1; LOADS: conv1(v+2) exists
__device__ __forceinline__ void sch_step(SchedState& st, int v, int q, unsigned short* fbase,
                                         f32x4 (&a0)[5], f32x4 (&a1)[5], f32x4 (&a2)[5]) {
    constexpr int PN = 1 - PAR;            // parity of v + 1
    constexpr bool kExch = ABL != 2 && ABL != 6, kFin = ABL != 3 && ABL != 6, kC1 = ABL != 5 && ABL != 6;
    const int pair_off = ((v + 2) >> 1) * 256;      // byte offset of pair (v+2)>>1 in the image row
    FinTmp ft;
    FinOut fo;
#define FIN(K) do { if (!FIRST && kFin) sch_fin<K>(st, ft, fo); } while (0)
#define ALN(H, W) do { if (!LAST && kC1 && ABL != 11 && PN == 1) sch_align<H, W>(st); } while (0)
#define STORE_PREV() do { if (!FIRST && kFin) { if (ABL == 7) asm volatile("" ::"v"(fo.o0), "v"(fo.o1), "v"(fo.tt)); else sch_store(fo, fbase, v - 1, q, st.gs); } } while (0)
#define WR(OT) do { if (kExch && ABL != 8) sch_part_write<PAR, OT>(st, a2[OT]); } while (0)
#define PK(N) do { if (!LAST && kC1 && ABL != 10) sch_packop<PN, N>(st); } while (0)
#define RD(R) do { if (kExch && ABL != 9) sch_red_load1<PAR, R>(st); } while (0)
#define LD(H, K) do { if (LOADS && kC1 && ABL != 11 && (K < 2 || PAR == 1)) sch_oper_load<H, K>(st, pair_off); } while (0)
    // the barrier's lgkmcnt(0) covers the ds_writes of a2, issued >= 13 MFMAs earlier.  a2 stays allocated until
    // here (ds_write / XDL hazard above): no MFMA issued before this point can have been given its registers
#define HANDOFF() do { \
        if (ABL == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        else if (kExch) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4])); } while (0)
    sch_wait_lds(st);      // partial(v-1) and the conv1 operands of v+1: read during T0 of the previous step
    // ---- T2
    sch_tap<PAR, 2, 0>(st, a2); FIN(0);
    sch_tap<PAR, 2, 1>(st, a2); FIN(1);
    sch_tap<PAR, 2, 2>(st, a2); FIN(2);
    sch_tap<PAR, 2, 3>(st, a2); FIN(3);
    sch_tap<PAR, 2, 4>(st, a2); FIN(4);
    sch_tap<PAR, 2, 5>(st, a2); FIN(5);
    sch_tap<PAR, 2, 6>(st, a2); FIN(6);
    sch_tap<PAR, 2, 7>(st, a2); FIN(7);
    sch_tap<PAR, 2, 8>(st, a2); FIN(8);
    sch_tap<PAR, 2, 9>(st, a2); FIN(9);
    sch_tap<PAR, 2, 10>(st, a2); FIN(10);
    sch_tap<PAR, 2, 11>(st, a2); FIN(11);
    sch_tap<PAR, 2, 12>(st, a2); FIN(12);
    sch_tap<PAR, 2, 13>(st, a2); FIN(13);
    sch_tap<PAR, 2, 14>(st, a2); FIN(14);
    sch_tap<PAR, 2, 15>(st, a2); ALN(0, 0);
    sch_tap<PAR, 2, 16>(st, a2); ALN(0, 1);
    STORE_PREV();
    sch_tap<PAR, 2, 17>(st, a2); ALN(1, 0);
    sch_tap<PAR, 2, 18>(st, a2); ALN(1, 1);
    sch_tap<PAR, 2, 19>(st, a2);
    // ---- C1
    if (!LAST && kC1) {
        if (PN == 0) { st.cb[0][0] = st.bw[0][0]; st.cb[0][1] = st.bw[0][1]; st.cb[1][0] = st.bw[1][0]; st.cb[1][1] = st.bw[1][1]; }
        sch_conv1_mfma<0, 0>(st); sch_conv1_mfma<0, 1>(st); sch_conv1_mfma<0, 2>(st); sch_conv1_mfma<0, 3>(st);
        sch_conv1_mfma<1, 0>(st); sch_conv1_mfma<1, 1>(st); sch_conv1_mfma<1, 2>(st); sch_conv1_mfma<1, 3>(st);
    }
    // ---- T1
    sch_tap<PAR, 1, 0>(st, a1); WR(0);
    sch_tap<PAR, 1, 1>(st, a1); PK(0);
    sch_tap<PAR, 1, 2>(st, a1); WR(1);
    sch_tap<PAR, 1, 3>(st, a1); PK(1);
    sch_tap<PAR, 1, 4>(st, a1); WR(2);
    sch_tap<PAR, 1, 5>(st, a1); PK(2);
    sch_tap<PAR, 1, 6>(st, a1); WR(3);
    sch_tap<PAR, 1, 7>(st, a1); PK(3);
    sch_tap<PAR, 1, 8>(st, a1); WR(4);
    sch_tap<PAR, 1, 9>(st, a1); PK(4);
    sch_tap<PAR, 1, 10>(st, a1); PK(5);
    sch_tap<PAR, 1, 11>(st, a1); PK(6);
    sch_tap<PAR, 1, 12>(st, a1); PK(7);
    sch_tap<PAR, 1, 13>(st, a1); PK(8);
    sch_tap<PAR, 1, 14>(st, a1); PK(9);
    sch_tap<PAR, 1, 15>(st, a1); PK(10);
    sch_tap<PAR, 1, 16>(st, a1); PK(11);
    sch_tap<PAR, 1, 17>(st, a1); PK(12);
    sch_tap<PAR, 1, 18>(st, a1); PK(13);
    sch_tap<PAR, 1, 19>(st, a1); PK(14);
    // ---- T0
    sch_tap<PAR, 0, 0>(st, a0); PK(15);
    sch_tap<PAR, 0, 1>(st, a0); PK(16);
    sch_tap<PAR, 0, 2>(st, a0); HANDOFF();
    sch_tap<PAR, 0, 3>(st, a0); PK(17); RD(0);
    sch_tap<PAR, 0, 4>(st, a0); PK(18); RD(1);
    sch_tap<PAR, 0, 5>(st, a0); PK(19); RD(2);
    sch_tap<PAR, 0, 6>(st, a0); PK(20); RD(3);
    sch_tap<PAR, 0, 7>(st, a0); PK(21); RD(4);
    sch_tap<PAR, 0, 8>(st, a0); PK(22); RD(5);
    sch_tap<PAR, 0, 9>(st, a0); PK(23); RD(6);
    sch_tap<PAR, 0, 10>(st, a0); PK(24); RD(7);
    sch_tap<PAR, 0, 11>(st, a0); PK(25); LD(0, 0);
    sch_tap<PAR, 0, 12>(st, a0); PK(26); LD(0, 1);
    sch_tap<PAR, 0, 13>(st, a0); PK(27); LD(1, 0);
    sch_tap<PAR, 0, 14>(st, a0); PK(28); LD(1, 1);
    sch_tap<PAR, 0, 15>(st, a0); PK(29); LD(0, 2);
    sch_tap<PAR, 0, 16>(st, a0); PK(30); LD(1, 2);
    sch_tap<PAR, 0, 17>(st, a0); PK(31);
    sch_tap<PAR, 0, 18>(st, a0);
    sch_tap<PAR, 0, 19>(st, a0);
    if (ABL == 10 && !LAST) {      // probe 10: the conv1 results must stay live up to here (async MFMA write)
        f32x4 &x0 = st.X[0][1], &x1 = st.X[1][1], &x2 = st.X[2][1], &x3 = st.X[3][1];
        f32x4 &y0 = st.X[0][0], &y1 = st.X[1][0], &y2 = st.X[2][0], &y3 = st.X[3][0];
        asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
    }
#undef FIN
#undef ALN
#undef STORE_PREV
#undef WR
#undef PK
#undef RD
#undef LD
#undef HANDOFF
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void vt_conv_bf16_sched_kernel(const float* __restrict__ x, long n,
                                                                    const u32x4* __restrict__ wq, const u32x2* __restrict__ a1q,
                                                                    const float* __restrict__ b2, unsigned short* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* img = reinterpret_cast<unsigned*>(smem);
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
    // accumulator rows of tile ot on this lane = output channels 16*ot + 4g .. +3
#pragma unroll
    for (int ot = 0; ot < 5; ++ot) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + 16 * ot + 4 * g);
        st.bias[ot] = q == 0 ? b : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+a"(st.bias[ot]));
    }
    const unsigned part_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (size_t)2 * kImgWords * 4);
    const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // partial tile entry of (frame f, chunk g): 16*(f>>2) + 4*(f&3) + ((g + (f>>2)) & 3).  The MFMA-layout writer
    // (lane = f + 16g, ds_write_b128 in groups of 8 lanes) and the transposed reader (lane = 4f + g, ds_read_b128
    // in the 16-lane groups of MI355X_MICROARCH.md) both touch 16 distinct 16-B columns per group: conflict-free.
    auto entry = [](int f, int gg) { return 16 * (f >> 2) + 4 * (f & 3) + ((gg + (f >> 2)) & 3); };
    const int fs = lane >> 2, gs = lane & 3;        // finishing role of this lane
    st.gs = gs;
    st.wr_addr = part_lds + (q * 5 * 64 + entry(nl, g)) * 16;
    st.rd_addr = part_lds + (q * 64 + entry(fs, gs)) * 16;
    st.rc_addr = part_lds + (4 * 64 + entry(fs, q)) * 16 + gs * 4;     // tile 4: wave q takes chunk q, lane its word gs
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
        unsigned short* fbase = feat + (grp * 16 + fs) * (long)(kW2 * kC2);
        const long gnext = grp + gridDim.x;
        // outputs 0 and 1 never see a fresh MFMA: they start from the bias; acc[2] is step 0's fresh accumulator
        f32x4 acc[3][5];
#pragma unroll
        for (int b = 0; b < 5; ++b) { acc[0][b] = st.bias[b]; acc[1][b] = st.bias[b]; acc[2][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        // prologue: conv1 of position 0, packed into Bf[0]; conv1 operands of position 1
        sch_oper_load<0, 0>(st, 0); sch_oper_load<0, 1>(st, 0); sch_oper_load<1, 0>(st, 0); sch_oper_load<1, 1>(st, 0);
        sch_wait_lds(st);
        st.cb[0][0] = st.bw[0][0]; st.cb[0][1] = st.bw[0][1]; st.cb[1][0] = st.bw[1][0]; st.cb[1][1] = st.bw[1][1];
        sch_conv1_mfma<0, 0>(st); sch_conv1_mfma<0, 1>(st); sch_conv1_mfma<0, 2>(st); sch_conv1_mfma<0, 3>(st);
        sch_conv1_mfma<1, 0>(st); sch_conv1_mfma<1, 1>(st); sch_conv1_mfma<1, 2>(st); sch_conv1_mfma<1, 3>(st);
        sch_oper_load<0, 0>(st, 0); sch_oper_load<0, 1>(st, 0); sch_oper_load<0, 2>(st, 0);
        sch_oper_load<1, 0>(st, 0); sch_oper_load<1, 1>(st, 0); sch_oper_load<1, 2>(st, 0);
        {
            f32x4 &x0 = st.X[0][1], &x1 = st.X[1][1], &x2 = st.X[2][1], &x3 = st.X[3][1];
            f32x4 &y0 = st.X[0][0], &y1 = st.X[1][0], &y2 = st.X[2][0], &y3 = st.X[3][0];
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
        }
        [&]<int... N>(std::integer_sequence<int, N...>) { (sch_packop<0, N>(st), ...); }(std::make_integer_sequence<int, 32>{});
        asm volatile("s_nop 1");

        sch_step<0, true, false, true, ABL>(st, 0, q, fbase, acc[2], acc[1], acc[0]);
        int v = 1;
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int it = 0; it < 21; ++it, v += 6) {     // v = 1 .. 126
            // next group's frames -> the other image buffer, a quarter per iteration; each quarter's global load is
            // issued one iteration (6 steps) before its conversion and LDS writes
            if (it >= 12 && it < 16 && gnext < ngroups) stage_write(it - 12, sv, n, gnext * 16, img + (buf ^ 1) * kImgWords, tid);
            if (it >= 11 && it < 15 && gnext < ngroups) sv = stage_load(it - 11, x, n, gnext * 16, tid);
            sch_step<1, false, false, true, ABL>(st, v + 0, q, fbase, acc[0], acc[2], acc[1]);
            sch_step<0, false, false, true, ABL>(st, v + 1, q, fbase, acc[1], acc[0], acc[2]);
            sch_step<1, false, false, true, ABL>(st, v + 2, q, fbase, acc[2], acc[1], acc[0]);
            sch_step<0, false, false, true, ABL>(st, v + 3, q, fbase, acc[0], acc[2], acc[1]);
            sch_step<1, false, false, true, ABL>(st, v + 4, q, fbase, acc[1], acc[0], acc[2]);
            sch_step<0, false, false, true, ABL>(st, v + 5, q, fbase, acc[2], acc[1], acc[0]);
        }
        sch_step<1, false, false, true, ABL>(st, 127, q, fbase, acc[0], acc[2], acc[1]);
        sch_step<0, false, false, false, ABL>(st, 128, q, fbase, acc[1], acc[0], acc[2]);
        sch_step<1, false, true, false, ABL>(st, 129, q, fbase, acc[2], acc[1], acc[0]);
        // tail: finish 129, then outputs 130 and 131 (complete as they are: only zero padding beyond)
        auto finish_store = [&](int w) {
            FinOut fo;
            sch_wait_lds(st);
            sch_fin_all(st, fo);
            sch_store(fo, fbase, w, q, st.gs);
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
                   case 5: MDC_LAUNCH_SCHED(5); break; case 6: MDC_LAUNCH_SCHED(6); break; case 7: MDC_LAUNCH_SCHED(7); break;
                   case 8: MDC_LAUNCH_SCHED(8); break; case 9: MDC_LAUNCH_SCHED(9); break; case 10: MDC_LAUNCH_SCHED(10); break;
                   case 11: MDC_LAUNCH_SCHED(11); break; default: MDC_LAUNCH_SCHED(0); }
#else
    MDC_LAUNCH_SCHED(0);
#endif
#undef MDC_LAUNCH_SCHED
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
