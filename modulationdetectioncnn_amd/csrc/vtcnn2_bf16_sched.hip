// Canonical VT-CNN2 (T3), bf16 path: the production conv1+conv2 kernel (see vtcnn2_bf16.hip for the algorithm,
// the MFMA operand layouts and the hipcc-scheduled statement of the same computation).
#include "vtcnn2_bf16_common.h"

#include <cstdlib>
#include <type_traits>

namespace mdc {

namespace {

// ------------------------------------------------------------------------------------
// vt_conv_bf16_sched_kernel: every instruction of the position step is an `asm volatile` statement, so the
// ORDER is the one written here (hipcc only allocates registers).  One wave per SIMD issues in order; what a
// filler instruction placed between two back-to-back MFMAs costs was measured with tools/microbench/mfma_gap.hip
// (cycles added to the 16.4-cycle MFMA gap, one wave per SIMD, AGPR accumulators):
//     1 VALU (v_add_f32, v_cvt_pk_bf16_f32, DPP mov)  +0.4      v_alignbit / v_pk_max_i16 / v_lshl_add_u64  +1.2
//     2 VALU  +4.4        3 VALU  +12        v_pk_add_f32  +16.6 (packed f32 is NOT cheap beside an MFMA)
//     1 ds_read_b32/b64/b128  +3             VALU + ds_read in ONE gap  +12        two ds_reads in one gap  +11.5
//     ds_write_b128  +16 (b64: +8)           global_store_dwordx2 +4, _short +16, both in one gap +44
//     the K=16 MFMA (v_mfma_f32_16x16x16_bf16) takes the same 16.3 cycles as the K=32 one
// Hence the rule of this schedule: ONE non-MFMA instruction per gap, all 68 gaps of a step used, the few items
// beyond 68 doubled up as VALU+VALU.  (The first version bunched 2-4 VALU and LDS instructions in some gaps and
// used v_pk_add_f32: its non-MFMA work cost its full issue time, 600 cycles on a 1100-cycle MFMA floor.)
//
// Step v (accumulators: a0 = output v+2, fresh; a1 = v+1; a2 = v, completes), 68 MFMAs:
//   T2  tap 2 (20)      gaps: finish of output v-1 (21 VALU: sum of the 4 partials, ReLU, bf16), conv1 operand words
//   C1  conv1(v+1) (8)  gaps: the 5 ds_write_b128 of a2 = partial(v), the 2 feature stores of output v-1
//   T1  tap 1 (20)      gaps: 20 of the 32 pack VALU of conv1(v+1) (v_cvt_pk_bf16_f32, v_pk_max_i16 = ReLU)
//   T0  tap 0 (20, 5 fresh with C = conv2 bias on wave 0)
//                       gaps: lgkmcnt(0)+s_barrier after the 3rd MFMA, 8 ds_reads of partial(v), 12 pack VALU,
//                       every 4th step the two ds_read_b64 of the next operand chunk
// Bf (packed ReLU'd conv1 output = B operand of conv2) is double-buffered by step parity.
// Image layout of THIS kernel: [buffer][row][lane][70 words], words = bf16 pairs of the padded row (lane = frame +
// 16*k-group as in vtcnn2_bf16.hip).  A lane's pairs are contiguous, so the conv1 operands of four positions are
// one ds_read_b64 per row (pairs 2c+2, 2c+3 of chunk c = v>>2; the other half is the previous chunk's), held in
// three rotating register pairs; stride 70 = 2 mod 4 keeps ds_read_b64 conflict-free.
// Hazards hipcc would not see inside asm, and how the order guarantees them:
//   VALU write -> MFMA read (2 wait states): Bf is written a phase before its first reader; the operand words
//     at least one MFMA before conv1;
//   MFMA write -> VALU/DS read (<= 18 wait states for these shapes): every reader is >= 2 MFMAs later;
//   an asm MFMA's result lands long after the statement: its destination must stay live until a reader
//     (a dead destination gets reallocated and clobbered: a "no pack" timing probe faulted that way);
//   ds_write source vs later MFMA overwrite (not interlocked for XDL writes): a2 is kept alive until after the
//     barrier's lgkmcnt(0);
//   the same for a global_store: hipcc gave the data and address registers of a feature store, dead after it, to
//     the conv1 MFMA that followed it (seen in the ISA; wrong features now and then).  Stores and anything else whose
//     VGPR operands die at the instruction sit where only AGPR-writing MFMAs follow (T1), never in T2's tail or C1.
// ------------------------------------------------------------------------------------
constexpr int kNV = 28;                   // conv2 fragments kept in VGPRs; the other 32 live in AGPRs
constexpr int kS = 70;                    // image words per lane row: pairs 0..67 (66, 67 zero) + 2 pad
constexpr int kSImgWords = 2 * 64 * kS;   // [row][lane][kS] per buffer
constexpr size_t kSchedLds = (size_t)2 * kSImgWords * 4 + (size_t)2 * kPartFloats * 4;      // 112,640 B

struct SchedState {
    u32x4 Wv[kNV];
    u32x4 Wa[kWFrags - kNV];
    f32x4 bias[5];            // conv2 bias tiles as the C operand of the fresh MFMAs (wave 0; zeros on waves 1-3)
    u32x2 A1[4];
    unsigned Bf[2][2][2][4];  // [step parity][row][channel pair][word]: B operands of conv2, as scalars
    f32x4 X[4][2];
    f32x4 rp[4];
    float rc[4];
    u32x2 P[3][2];            // operand chunks [slot][row]: chunk c has pairs (2c, 2c+1) in slot c%3, (2c+2, 2c+3) in (c+1)%3
    unsigned cb[2][2];        // conv1 B operand words [row][word] when they are not a chunk half as it is
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
// even N = v_cvt_pk_bf16_f32 of two channels, odd N = ReLU on the packed pair (negative bf16 <=> negative int16).
// Unit k reads the result of conv1 MFMA number k>>1.
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
// operand chunk half: pairs (pair0, pair0+1) of row H into slot SLOT; chunk_addr = im_addr + 4*pair0
template <int SLOT, int H>
__device__ __forceinline__ void sch_chunk_load(SchedState& st, unsigned chunk_addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(st.P[SLOT][H]) : "v"(chunk_addr), "i"(H * 64 * kS * 4) : "memory");
}
// conv1 B operand of position p (R = p&3, chunk words W0..W3 = slot LO .x .y, slot (LO+1)%3 .x .y), prepared one word
// per call (I = 2*row + word): R=0 (W0,W1) and R=2 (W1,W2) start on a pair, R=1 / R=3 one sample later (v_alignbit)
template <int R, int LO, int I>
__device__ __forceinline__ void sch_prep(SchedState& st) {
    constexpr int H = I >> 1, W = I & 1, HI = (LO + 1) % 3;
    const unsigned w0 = st.P[LO][H][0], w1 = st.P[LO][H][1], w2 = st.P[HI][H][0], w3 = st.P[HI][H][1];
    unsigned& d = st.cb[H][W];
    if constexpr (R == 1) {
        if constexpr (W == 0) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w1), "v"(w0));
        else asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w2), "v"(w1));
    } else if constexpr (R == 3) {
        if constexpr (W == 0) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w2), "v"(w1));
        else asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(w3), "v"(w2));
    } else if constexpr (R == 2) {
        if constexpr (W == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(w1));
        else asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(w2));
    }
}
template <int R, int LO, int H, int CT>
__device__ __forceinline__ void sch_conv1_mfma(SchedState& st) {
    const u32x2 b = R == 0 ? st.P[LO][H] : u32x2{st.cb[H][0], st.cb[H][1]};
    // "=&v": the result must not share registers with an operand
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=&v"(st.X[CT][H]) : "v"(st.A1[CT]), "v"(b));
}
template <int PB, int OT>
__device__ __forceinline__ void sch_part_write(SchedState& st, const f32x4& a) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(st.wr_addr), "a"(a), "i"(PB * kPartFloats * 4 + OT * 1024) : "memory");
}
// read R (0..7) of the owner's share of partial(v): even = float4 of wave R/2's partial of tile q, odd = its
// word of tile 4
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
                   "+v"(st.rc[3]), "+v"(st.P[0][0]), "+v"(st.P[0][1]), "+v"(st.P[1][0]), "+v"(st.P[1][1]), "+v"(st.P[2][0]), "+v"(st.P[2][1])
                 :: "memory");
}
// finish of one output position, one VALU instruction per call (K = 0..20): sum of the 4 partials (the bias is
// already in wave 0's), ReLU, bf16.  Plain v_add_f32: v_pk_add_f32 costs a whole MFMA gap.
struct FinOut { unsigned o0, o1, tt; };
struct FinTmp { float s[4], u[4], a, b, t; };
template <int K>
__device__ __forceinline__ void sch_fin(SchedState& st, FinTmp& f, FinOut& out) {
    if constexpr (K < 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.s[K]) : "v"(st.rp[0][K]), "v"(st.rp[1][K]));
    else if constexpr (K < 8) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.u[K - 4]) : "v"(st.rp[2][K - 4]), "v"(st.rp[3][K - 4]));
    else if constexpr (K < 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f.s[K - 8]) : "v"(f.u[K - 8]));
    else if constexpr (K == 12) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(out.o0) : "v"(f.s[0]), "v"(f.s[1]));
    else if constexpr (K == 13) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(out.o0));
    else if constexpr (K == 14) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(out.o1) : "v"(f.s[2]), "v"(f.s[3]));
    else if constexpr (K == 15) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(out.o1));
    else if constexpr (K == 16) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.a) : "v"(st.rc[0]), "v"(st.rc[1]));
    else if constexpr (K == 17) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.b) : "v"(st.rc[2]), "v"(st.rc[3]));
    else if constexpr (K == 18) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.t) : "v"(f.a), "v"(f.b));
    else if constexpr (K == 19) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(out.tt) : "v"(f.t));
    else asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(out.tt));
}
__device__ __forceinline__ void sch_fin_all(SchedState& st, FinOut& out) {
    FinTmp f;
    [&]<int... K>(std::integer_sequence<int, K...>) { (sch_fin<K>(st, f, out), ...); }(std::make_integer_sequence<int, 21>{});
}
// The finishing lane layout is TRANSPOSED with respect to the MFMA layout: lane L finishes frame L>>2, channel
// chunk gs = L&3, so the four lanes of a quad write 32 (and 8) contiguous bytes of one frame's row.
// Wave q stores channels [16q+4gs, +4) (WHICH = 0) and channel 64+4q+gs (WHICH = 1) of its lane's frame (row frow)
// at output position w.
template <int WHICH>
__device__ __forceinline__ void sch_store(const FinOut& fo, unsigned short* frow, int w, int q, int gs) {
    unsigned short* dst = frow + (long)w * kC2;
    if constexpr (WHICH == 0) *reinterpret_cast<u32x2*>(dst + 16 * q + 4 * gs) = u32x2{fo.o0, fo.o1};
    else dst[64 + 4 * q + gs] = (unsigned short)fo.tt;
}

// One position step.  V12 = v mod 12 fixes every register choice: Bf/partial-buffer parity (v&1), the accumulator
// roles (v%3), the role of position v+1 in its operand chunk ((v+1)&3) and the chunk slots (((v+1)>>2)%3).
// ABL: 0 = product; timing-only probes (tools/ablate_sched.py, -DMDC_ABLATIONS; results wrong; every conv2 MFMA stays):
//   1 no s_barrier   2 no exchange (ds_writes, barrier, reads)   3 no finish VALU / feature stores
//   5 no conv1 (operand prep, 8 MFMAs, pack)                      6 all of 2, 3, 5
//   7 no feature stores (finish VALU kept)   8 no ds_writes   9 no reads of the partials
//   10 no pack VALU (conv1 kept)             11 no operand chunk loads / prep
template <int V12, bool FIRST, bool LAST, int ABL>
__device__ __forceinline__ void sch_step(SchedState& st, int v, int q, unsigned short* fbase, f32x4 (&acc)[3][5]) {
    constexpr int PAR = V12 & 1, PN = 1 - PAR;
    constexpr int R1 = (V12 + 1) & 3, LO1 = ((V12 + 1) >> 2) % 3;             // position v+1 in its chunk; slot of its W0,W1
    constexpr bool kLoad = (V12 & 3) == 1 && !LAST;                           // fetch pairs (2c+4, 2c+5), c = v>>2
    constexpr int LSLOT = ((V12 >> 2) + 2) % 3;
    constexpr bool kExch = ABL != 2 && ABL != 6, kFin = ABL != 3 && ABL != 6, kC1 = ABL != 5 && ABL != 6;
    f32x4 (&a2)[5] = acc[V12 % 3];
    f32x4 (&a1)[5] = acc[(V12 + 1) % 3];
    f32x4 (&a0)[5] = acc[(V12 + 2) % 3];
    const unsigned chunk_addr = st.im_addr + ((v >> 2) * 8 + 16);
    FinTmp ft;
    FinOut fo;
#define FIN(K) do { if (!FIRST && kFin) sch_fin<K>(st, ft, fo); } while (0)
#define PREP(I) do { if (!LAST && kC1 && ABL != 11) sch_prep<R1, LO1, I>(st); } while (0)
#define C1M(H, CT) do { if (!LAST && kC1) sch_conv1_mfma<R1, LO1, H, CT>(st); } while (0)
#define ST(W) do { if (!FIRST && kFin) { if (ABL == 7) asm volatile("" ::"v"(fo.o0), "v"(fo.o1), "v"(fo.tt)); else sch_store<W>(fo, fbase, v - 1, q, st.gs); } } while (0)
#define WR(OT) do { if (kExch && ABL != 8) sch_part_write<PAR, OT>(st, a2[OT]); } while (0)
#define PK(N) do { if (!LAST && kC1 && ABL != 10) sch_packop<PN, N>(st); } while (0)
#define RD(R) do { if (kExch && ABL != 9) sch_red_load1<PAR, R>(st); } while (0)
#define LD(H) do { if (kLoad && kC1 && ABL != 11) sch_chunk_load<LSLOT, H>(st, chunk_addr); } while (0)
    // the barrier's lgkmcnt(0) covers the ds_writes of a2, issued >= 30 MFMAs earlier.  a2 stays allocated until
    // here (ds_write / XDL hazard above): no MFMA issued before this point can have been given its registers
#define HANDOFF() do { \
        if (ABL == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        else if (kExch) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4])); } while (0)
    sch_wait_lds(st);      // partial(v-1) and any operand chunk: read during T0 of the previous step
    // ---- T2: tap 2 -> a2 complete.  gaps: finish of output v-1 (F0..F18); conv1 operand words of v+1 ride in the last four
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
    sch_tap<PAR, 2, 15>(st, a2); FIN(15);
    sch_tap<PAR, 2, 16>(st, a2); FIN(16); PREP(0);
    sch_tap<PAR, 2, 17>(st, a2); FIN(17); PREP(1);
    sch_tap<PAR, 2, 18>(st, a2); FIN(18); PREP(2);
    sch_tap<PAR, 2, 19>(st, a2); PREP(3);
    // ---- C1: conv1(v+1), the only MFMAs that write VGPRs.  gaps: end of the finish, first ds_writes of partial(v)
    //      (the four waves share the CU's LDS store path, 13 cycles per ds_write_b128: one every third gap keeps it
    //      unsaturated), first pack ops.  NO feature store here: see the hazard list
    C1M(0, 0); FIN(19);
    C1M(0, 1); WR(0);
    C1M(0, 2); FIN(20);
    C1M(0, 3); PK(0);
    C1M(1, 0); WR(1);
    C1M(1, 1); PK(1);
    C1M(1, 2); PK(2);
    C1M(1, 3); WR(2);
    // ---- T1: tap 1.  gaps: the two feature stores, the last two ds_writes, pack of conv1(v+1) one VALU each
    sch_tap<PAR, 1, 0>(st, a1); ST(0);
    sch_tap<PAR, 1, 1>(st, a1); PK(3);
    sch_tap<PAR, 1, 2>(st, a1); WR(3);
    sch_tap<PAR, 1, 3>(st, a1); ST(1);
    sch_tap<PAR, 1, 4>(st, a1); PK(4);
    sch_tap<PAR, 1, 5>(st, a1); WR(4);
    sch_tap<PAR, 1, 6>(st, a1); PK(5);
    sch_tap<PAR, 1, 7>(st, a1); PK(6);
    sch_tap<PAR, 1, 8>(st, a1); PK(7);
    sch_tap<PAR, 1, 9>(st, a1); PK(8);
    sch_tap<PAR, 1, 10>(st, a1); PK(9);
    sch_tap<PAR, 1, 11>(st, a1); PK(10);
    sch_tap<PAR, 1, 12>(st, a1); PK(11);
    sch_tap<PAR, 1, 13>(st, a1); PK(12);
    sch_tap<PAR, 1, 14>(st, a1); PK(13);
    sch_tap<PAR, 1, 15>(st, a1); PK(14);
    sch_tap<PAR, 1, 16>(st, a1); PK(15);
    sch_tap<PAR, 1, 17>(st, a1); PK(16);
    sch_tap<PAR, 1, 18>(st, a1); PK(17);
    sch_tap<PAR, 1, 19>(st, a1); PK(18);
    // ---- T0: tap 0 (5 fresh).  gaps: chunk loads (every 4th step), hand-off, the 8 reads of partial(v), rest of the pack
    sch_tap<PAR, 0, 0>(st, a0); PK(19); LD(0);
    sch_tap<PAR, 0, 1>(st, a0); PK(20); LD(1);
    sch_tap<PAR, 0, 2>(st, a0); PK(21); HANDOFF();
    sch_tap<PAR, 0, 3>(st, a0); RD(0);
    sch_tap<PAR, 0, 4>(st, a0); RD(1);
    sch_tap<PAR, 0, 5>(st, a0); RD(2);
    sch_tap<PAR, 0, 6>(st, a0); RD(3);
    sch_tap<PAR, 0, 7>(st, a0); RD(4);
    sch_tap<PAR, 0, 8>(st, a0); RD(5);
    sch_tap<PAR, 0, 9>(st, a0); RD(6);
    sch_tap<PAR, 0, 10>(st, a0); RD(7);
    sch_tap<PAR, 0, 11>(st, a0); PK(22);
    sch_tap<PAR, 0, 12>(st, a0); PK(23);
    sch_tap<PAR, 0, 13>(st, a0); PK(24);
    sch_tap<PAR, 0, 14>(st, a0); PK(25);
    sch_tap<PAR, 0, 15>(st, a0); PK(26);
    sch_tap<PAR, 0, 16>(st, a0); PK(27);
    sch_tap<PAR, 0, 17>(st, a0); PK(28);
    sch_tap<PAR, 0, 18>(st, a0); PK(29);
    sch_tap<PAR, 0, 19>(st, a0); PK(30); PK(31);
    if (ABL == 10 && !LAST) {      // probe 10: the conv1 results must stay live up to here (async MFMA write)
        f32x4 &x0 = st.X[0][1], &x1 = st.X[1][1], &x2 = st.X[2][1], &x3 = st.X[3][1];
        f32x4 &y0 = st.X[0][0], &y1 = st.X[1][0], &y2 = st.X[2][0], &y3 = st.X[3][0];
        asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
    }
#undef FIN
#undef PREP
#undef C1M
#undef ST
#undef WR
#undef PK
#undef RD
#undef LD
#undef HANDOFF
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
        unsigned* d = im + (h * 64 + i) * kS + 2 * m + 1 + e;      // samples 4m+2e, +1 -> padded 4m+2e+2, +3
        d[0] = hi;                // k-group 0: x hi
        d[16 * kS] = lo;          // k-group 1: x lo
        d[32 * kS] = hi;          // k-group 2: x hi again (meets the low halves of the taps)
    }
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
    const unsigned part_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (size_t)2 * kSImgWords * 4);
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
    for (int s = 0; s < 3; ++s) { st.P[s][0] = u32x2{0u, 0u}; st.P[s][1] = u32x2{0u, 0u}; }
    st.cb[0][0] = st.cb[0][1] = st.cb[1][0] = st.cb[1][1] = 0u;

    // LDS init: zero padding pairs; the k-group-3 lanes (48..63) hold the constant (1,1) of the bias slots
    for (int i = tid; i < 2 * kSImgWords; i += 256) img[i] = (((i / kS) & 63) >= 48) ? 0x3F803F80u : 0u;
    __syncthreads();
    const long ngroups = (n + 15) >> 4;
    long grp = blockIdx.x;
    if (grp < ngroups)
        for (int k = 0; k < 4; ++k) sch_stage_write(k, stage_load(k, x, n, grp * 16, tid), n, grp * 16, img, tid);
    __syncthreads();

    int buf = 0;
    for (; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        st.im_addr = img_lds + (buf * kSImgWords + lane * kS) * 4;
        unsigned short* fbase = feat + (grp * 16 + fs) * (long)(kW2 * kC2);
        const long gnext = grp + gridDim.x;
        // outputs 0 and 1 never see a fresh MFMA: they start from the bias; acc[2] is step 0's fresh accumulator
        f32x4 acc[3][5];
#pragma unroll
        for (int b = 0; b < 5; ++b) { acc[0][b] = st.bias[b]; acc[1][b] = st.bias[b]; acc[2][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        // prologue: chunk 0 (pairs 0..3), conv1 of position 0 packed into Bf[0]
        sch_chunk_load<0, 0>(st, st.im_addr); sch_chunk_load<0, 1>(st, st.im_addr);
        sch_chunk_load<1, 0>(st, st.im_addr + 8); sch_chunk_load<1, 1>(st, st.im_addr + 8);
        sch_wait_lds(st);
        sch_conv1_mfma<0, 0, 0, 0>(st); sch_conv1_mfma<0, 0, 0, 1>(st); sch_conv1_mfma<0, 0, 0, 2>(st); sch_conv1_mfma<0, 0, 0, 3>(st);
        sch_conv1_mfma<0, 0, 1, 0>(st); sch_conv1_mfma<0, 0, 1, 1>(st); sch_conv1_mfma<0, 0, 1, 2>(st); sch_conv1_mfma<0, 0, 1, 3>(st);
        {
            f32x4 &x0 = st.X[0][1], &x1 = st.X[1][1], &x2 = st.X[2][1], &x3 = st.X[3][1];
            f32x4 &y0 = st.X[0][0], &y1 = st.X[1][0], &y2 = st.X[2][0], &y3 = st.X[3][0];
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
        }
        [&]<int... N>(std::integer_sequence<int, N...>) { (sch_packop<0, N>(st), ...); }(std::make_integer_sequence<int, 32>{});
        asm volatile("s_nop 1");

        sch_step<0, true, false, ABL>(st, 0, q, fbase, acc);
        int v = 1;
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int it = 0; it < 10; ++it, v += 12) {     // v = 1 .. 120
            // next group's frames -> the other image buffer, a quarter per iteration; each quarter's global load is
            // issued one iteration (12 steps) before its conversion and LDS writes
            if (it >= 6 && gnext < ngroups) sch_stage_write(it - 6, sv, n, gnext * 16, img + (buf ^ 1) * kSImgWords, tid);
            if (it >= 5 && it < 9 && gnext < ngroups) sv = stage_load(it - 5, x, n, gnext * 16, tid);
            sch_step<1, false, false, ABL>(st, v + 0, q, fbase, acc);
            sch_step<2, false, false, ABL>(st, v + 1, q, fbase, acc);
            sch_step<3, false, false, ABL>(st, v + 2, q, fbase, acc);
            sch_step<4, false, false, ABL>(st, v + 3, q, fbase, acc);
            sch_step<5, false, false, ABL>(st, v + 4, q, fbase, acc);
            sch_step<6, false, false, ABL>(st, v + 5, q, fbase, acc);
            sch_step<7, false, false, ABL>(st, v + 6, q, fbase, acc);
            sch_step<8, false, false, ABL>(st, v + 7, q, fbase, acc);
            sch_step<9, false, false, ABL>(st, v + 8, q, fbase, acc);
            sch_step<10, false, false, ABL>(st, v + 9, q, fbase, acc);
            sch_step<11, false, false, ABL>(st, v + 10, q, fbase, acc);
            sch_step<0, false, false, ABL>(st, v + 11, q, fbase, acc);
        }
        sch_step<1, false, false, ABL>(st, 121, q, fbase, acc);
        sch_step<2, false, false, ABL>(st, 122, q, fbase, acc);
        sch_step<3, false, false, ABL>(st, 123, q, fbase, acc);
        sch_step<4, false, false, ABL>(st, 124, q, fbase, acc);
        sch_step<5, false, false, ABL>(st, 125, q, fbase, acc);
        sch_step<6, false, false, ABL>(st, 126, q, fbase, acc);
        sch_step<7, false, false, ABL>(st, 127, q, fbase, acc);
        sch_step<8, false, false, ABL>(st, 128, q, fbase, acc);
        sch_step<9, false, true, ABL>(st, 129, q, fbase, acc);
        // tail: finish 129, then outputs 130 and 131 (complete as they are: only zero padding beyond).
        // step 129 (v%3 == 0) left output 130 in acc[1] and output 131 in acc[2].
        auto finish_store = [&](int w) {
            FinOut fo;
            sch_wait_lds(st);
            sch_fin_all(st, fo);
            sch_store<0>(fo, fbase, w, q, st.gs);
            sch_store<1>(fo, fbase, w, q, st.gs);
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
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_bf16_sched_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSchedLds)); \
    hipLaunchKernelGGL(vt_conv_bf16_sched_kernel<A>, dim3(grid), dim3(256), kSchedLds, s, x, (long)n, \
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
