// Canonical VT-CNN2 (T3), bf16 path: the production conv1+conv2 kernel (see vtcnn2_bf16.hip for the algorithm
// and the hipcc-scheduled statement of the same computation; operand layouts of THIS kernel are described here).
#include "vtcnn2_bf16_common.h"
#include "vtcnn2_sched_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mdc {

namespace {

// ------------------------------------------------------------------------------------
// vt_conv_bf16_sched_kernel: every instruction of the position step is an `asm volatile` statement, so the
// ORDER is the one written here (hipcc only allocates registers).  One wave per SIMD issues in order; what a
// filler instruction placed between two back-to-back MFMAs costs was measured with tools/microbench/mfma_gap.hip
// (cycles added to the 16.4-cycle gap of v_mfma_f32_16x16x32_bf16, one wave per SIMD, AGPR accumulators):
//     1 VALU (v_add_f32, v_cvt_pk_bf16_f32, DPP mov)  +0.4      v_alignbit / v_pk_max_i16 / v_lshl_add_u64  +1.2
//     2 VALU  +4.4        3 VALU  +12        v_pk_add_f32  +16.6 (packed f32 is NOT cheap beside an MFMA)
//     1 ds_read_b32/b64/b128  +3             VALU + ds_read in ONE gap  +12        two ds_reads in one gap  +11.5
//     ds_write_b128  +16 (b64: +8)           global_store_dwordx2 +4, _short +16, both in one gap +44
//     v_mfma_f32_16x16x16_bf16 takes the same 16.3 cycles as the K=32 shape; v_mfma_f32_32x32x16_bf16 takes 32.0 and
//     its gap holds two VALU for free, VALU + ds_read for +3
// Hence the rule of this schedule: ONE non-MFMA instruction per 16-cycle gap, every gap of a step used, the few
// items beyond that doubled up as VALU+VALU.  (The first version bunched 2-4 VALU and LDS instructions in some
// gaps and used v_pk_add_f32: its non-MFMA work cost its full issue time, 600 cycles on a 1100-cycle MFMA floor.)
//
// Step v (accumulators: a0 = output v+2, fresh; a1 = v+1; a2 = v, completes), 62 MFMAs:
//   T2  tap 2 (20)      gaps: finish of output v-1 (18 VALU: sum of the 4 partials, ReLU + bf16 in the conversion's clamp
//                       bit), conv1 operand dwords
//   C1  conv1(v+1): 2 x v_mfma_f32_32x32x16_bf16 (M = 32 channels, N = 16 frames x 2 I/Q rows, K = 16 slots)
//   T1  tap 1 (20)      gaps: the 2 feature stores of output v-1, ds_writes of a2 = partial(v), pack of conv1(v+1)
//   T0  tap 0 (20, 5 fresh with C = conv2 bias on wave 0)
//                       gaps: last pack VALU, lgkmcnt(0)+s_barrier after the 3rd MFMA, 8 ds_reads of partial(v),
//                       every 2nd step one LDS read of conv1 operands; nine gaps stay empty
// Round 3: ReLU moved into the clamp bit of the bf16 conversions (19 VALU fewer per step: no gap holds two VALU any more).
// conv1 on the 32x32 shape: the MFMA's column is (frame, row), so its result holds BOTH rows of a frame in lanes
// f+16h (+32 for the upper k-half): packed to bf16 it is a B operand of conv2's 16x16x32 MFMA whose K index mixes
// rows and channels -- conv2 sums over both, so only the weight fragments' K order changes (packed to match on
// the host, vtcnn2_bf16_pack_sched).  Bf (that operand) is double-buffered by step parity.
// K slots of conv1 (16): lanes 0-31 carry k 0..7 = (x_hi s0,s1 | x_lo s0,s1 | x_hi s2,s3 | x_lo s2,s3) against the
// taps' high halves, lanes 32-63 carry k 8..15 = (x_hi s0,s1 | 1,1 | x_hi s2,s3 | 1,1) against the taps' low halves
// and the bias (hi, lo): conv1 is accurate to ~2^-16 although every operand is bf16.
// Image of a 16-frame group: [lane = f + 16h + 32*khalf][140 words]; entry e (2 words) = (bf16 pair e of the padded
// row: x_hi, then x_lo or (1,1)).  The B operand of an even position is 4 consecutive words (one ds_read_b128 or
// ds_read2_b64, used as loaded); odd positions start one sample later (4 v_alignbit).  Stride 140 = 4 * odd keeps
// ds_read_b128 conflict-free.
// Hazards hipcc would not see inside asm, and how the order guarantees them:
//   VALU write -> MFMA read (2 wait states): Bf is written a phase before its first reader; the operand dwords
//     at least one MFMA before conv1;
//   MFMA write -> VALU/DS read: every reader is >= 2 MFMAs (16-cycle shapes) / >= 5 MFMAs (the 32x32 results) later;
//   an asm MFMA's result lands long after the statement: its destination must stay live until a reader
//     (a dead destination gets reallocated and clobbered: a "no pack" timing probe faulted that way);
//   ds_write source vs later MFMA overwrite (not interlocked for XDL writes): a2 is kept alive until after the
//     barrier's lgkmcnt(0);
//   compiler-made AGPR copies next to an asm MFMA are not padded (see the accumulator init in the kernel; this one
//     was PROVEN: v_accvgpr_mov directly before the asm MFMA reading it, garbage in exactly that register): every
//     AGPR this kernel's MFMAs read is written by asm only;
//   precaution, not proven: hipcc once gave the data and address registers of a feature store, dead after it, to
//     the conv1 MFMA that followed it (seen in the ISA while chasing wrong features that the AGPR-copy hazard above
//     may equally have caused).  If a store reads its operands late, as a ds_write does, that is a wild write, so
//     stores and anything else whose VGPR operands die at the instruction sit where only AGPR-writing MFMAs follow
//     (T1), never in T2's tail or C1, and tools/lint_async_hazards.py checks the ISA for the pattern.
// ------------------------------------------------------------------------------------
constexpr int kNV = 24;                   // conv2 fragments kept in VGPRs; the other 36 live in AGPRs

struct SchedState {
    u32x4 Wv[kNV];
    u32x4 Wa[kWFrags - kNV];
    f32x4 bias[5];            // conv2 bias tiles as the C operand of the fresh MFMAs (wave 0; zeros on waves 1-3)
    u32x4 A1[2];              // conv1 A operands: 32 channels x 16 k-slots each
    unsigned Bf[2][2][2][4];  // [step parity][channel block ct][half bb][word]: B operands of conv2, as scalars
    f32x16 X[2];              // conv1 results [channel block]
    f32x4 rp[4];
    float rc[4];
    float tprev;              // fifth-tile value of the last EVEN output position (waits for its odd neighbour: one dword store per pair)
    u32x4 L0[3];              // conv1 operand words, entries (2c, 2c+1) of chunk c = position>>2, slot c%3
    u32x4 L1;                 // entries (2c+1, 2c+2)
    unsigned cb[4];           // B operand dwords of an odd position (v_alignbit results)
    unsigned wr_addr, rd_addr, rc_addr, im_addr;    // LDS byte addresses (lane part)
    int gs;                                         // finishing role of this lane: channel chunk (lane & 3)
};

// conv2 MFMA number I (0..19) of tap J: channel block CT = 1 - I/10, half BB = (I/5)%2, output tile OT = I%5
// PAD: the MFMA opens a block at a control-flow merge or split (range form: loop header, loop exit, the early exit of
// the ranges 0 .. 9 after step S+13), where hipcc may
// place AGPR copies (v_accvgpr_mov) of the accumulator right in front of it: two wait states inside the asm block
template <int SP, int J, int I, bool PAD = false>
__device__ __forceinline__ void sch_tap(SchedState& st, f32x4 (&acc)[5]) {
    constexpr int CT = 1 - I / 10, BB = (I / 5) % 2, OT = I % 5;
    constexpr int IDX = ((CT * 3 + J) * 2 + BB) * 5 + OT;
    const u32x4 b = u32x4{st.Bf[SP][CT][BB][0], st.Bf[SP][CT][BB][1], st.Bf[SP][CT][BB][2], st.Bf[SP][CT][BB][3]};
    if constexpr (J == 0 && I < 5) {      // first MFMA of output v+2: C = bias (early-clobber: D must not alias an input)
        if constexpr (IDX < kNV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b), "a"(st.bias[OT]));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&a"(acc[OT]) : "a"(st.Wa[IDX - kNV]), "v"(b), "a"(st.bias[OT]));
    } else if constexpr (PAD) {
        static_assert(IDX >= kNV, "padded form written for the AGPR-resident fragments only");
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[OT]) : "a"(st.Wa[IDX - kNV]), "v"(b));
    } else {
        if constexpr (IDX < kNV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[OT]) : "a"(st.Wa[IDX - kNV]), "v"(b));
    }
}
// pack instruction U (0..15) of conv1's X into Bf[SP]: unit U = (channel block ct, result register pair j): ReLU and bf16
// pack of two channels in ONE v_cvt_pk_bf16_f32 with the clamp bit (values below 1: kFeatShift, vtcnn2_sched_common.h)
template <int SP, int U>
__device__ __forceinline__ void sch_packop(SchedState& st) {
    constexpr int CT = U >> 3, j = U & 7;
    unsigned& d = st.Bf[SP][CT][j >> 2][j & 3];
    const float lo = st.X[CT][2 * j], hi = st.X[CT][2 * j + 1];
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(lo), "v"(hi));
}
// One position step.  V12 = v mod 12 fixes every register choice: Bf/partial-buffer parity (v&1), the accumulator
// roles (v%3), the role of position v+1 in its operand chunk ((v+1)&3) and the chunk slots (((v+1)>>2)%3).
// ABL: 0 = product; timing-only probes (tools/ablate_sched.py, -DMDC_ABLATIONS; results wrong; every conv2 MFMA stays):
//   1 no s_barrier   2 no exchange (ds_writes, barrier, reads)   3 no finish VALU / feature stores
//   5 no conv1 (operand prep, 2 MFMAs, pack)                      6 all of 2, 3, 5
//   7 no feature stores (finish VALU kept)   8 no ds_writes   9 no reads of the partials
//   10 no pack VALU (conv1 kept)             11 no operand loads / prep
// RANGE (small batches, see vt_conv_bf16_sched_kernel): the step belongs to a work-group that owns only the output
// positions >= wlo of its frame group; outputs below wlo are computed from incomplete taps and must not be stored.
template <int V12, bool FIRST, bool LAST, int ABL, bool RANGE = false>
__device__ __forceinline__ void sch_step(SchedState& st, int v, int q, unsigned short* fbase, f32x4 (&acc)[3][5], int wlo = 0) {
    constexpr int PAR = V12 & 1, PN = 1 - PAR;
    constexpr int R1 = (V12 + 1) & 3, S0 = ((V12 + 1) >> 2) % 3, SN = (S0 + 1) % 3;      // position v+1 in its chunk; slots
    // operand loads: v = 4c+1 fetches entries (2c+2, 2c+3) = L0 of chunk c+1; v = 4c+3 entries (2c+3, 2c+4) = L1 of chunk c+1
    constexpr bool kLoadEven = (V12 & 3) == 1 && !LAST, kLoadOdd = (V12 & 3) == 3 && !LAST;
    constexpr int LSLOT = ((V12 >> 2) + 1) % 3;
    constexpr bool kExch = ABL != 2 && ABL != 6, kFin = ABL != 3 && ABL != 6, kC1 = ABL != 5 && ABL != 6;
    f32x4 (&a2)[5] = acc[V12 % 3];
    f32x4 (&a1)[5] = acc[(V12 + 1) % 3];
    f32x4 (&a0)[5] = acc[(V12 + 2) % 3];
    const unsigned load_addr = st.im_addr + ((v >> 2) * 16 + (kLoadEven ? 16 : 24));      // 8 bytes per entry
    FinTmp ft;
    FinOut fo;
    constexpr int OP = (V12 + 1) & 1;      // parity of the output position v - 1 this step finishes (12 is even)
#define FIN(K) do { if (!FIRST && kFin) sch_fin_clamp<K, OP>(st, ft, fo); } while (0)
#define PREP(I) do { if (!LAST && kC1 && ABL != 11) sch_prep<R1, S0, SN, I>(st); } while (0)
#define C1M(CT) do { if (!LAST && kC1) sch_conv1_mfma<R1, S0, CT>(st); } while (0)
#define ST(W) do { if (!FIRST && kFin && (W == 0 || OP == 1)) { if (ABL == 7) { if (W == 0) asm volatile("" ::"v"(fo.o0), "v"(fo.o1)); else asm volatile("" ::"v"(fo.tt)); } else if (!RANGE || v - 1 >= wlo) sch_store<W>(fo, fbase, v - 1, q, st.gs); } } while (0)
#define WR(OT) do { if (kExch && ABL != 8) sch_part_write<PAR, OT>(st, a2[OT]); } while (0)
#define PK(N) do { if (!LAST && kC1 && ABL != 10) sch_packop<PN, N>(st); } while (0)
#define RD(R) do { if (kExch && ABL != 9) sch_red_load1<PAR, R>(st); } while (0)
#define LD() do { if (kC1 && ABL != 11) { if (kLoadEven) sch_load_even<LSLOT>(st, load_addr); else if (kLoadOdd) sch_load_odd(st, load_addr); } } while (0)
    // the barrier's lgkmcnt(0) covers the ds_writes of a2, issued >= 14 MFMAs earlier.  a2 stays allocated until
    // here (ds_write / XDL hazard above): no MFMA issued before this point can have been given its registers
    // (MDC_F8_PROBE_HALF_BARRIER, round 5: timing-only probe of a two-step rendezvous -- the barrier on even steps only; results wrong)
#ifdef MDC_F8_PROBE_HALF_BARRIER
    constexpr bool kHalfBarrierSkip = (V12 & 1) == 1;
#else
    constexpr bool kHalfBarrierSkip = false;
#endif
#define HANDOFF() do { \
        if (ABL == 1 || kHalfBarrierSkip) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        else if (kExch) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4])); } while (0)
    sch_wait_lds(st);      // partial(v-1) and any operand words: read during T0 of the previous step
    // ---- T2: tap 2 -> a2 complete.  gaps: finish of output v-1 (F0..F15); the conv1 operand dwords of v+1 in the last four
    sch_tap<PAR, 2, 0, RANGE && (V12 == 1 || V12 == 2)>(st, a2); FIN(0);
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
    sch_tap<PAR, 2, 16>(st, a2); PREP(0);
    sch_tap<PAR, 2, 17>(st, a2); PREP(1);
    sch_tap<PAR, 2, 18>(st, a2); PREP(2);
    sch_tap<PAR, 2, 19>(st, a2); PREP(3);
    // ---- C1: conv1(v+1): two 32x32x16 MFMAs, the only ones that write VGPRs.  Their 32-cycle gaps take two VALU and
    //      one LDS instruction each: end of the finish, first two ds_writes of partial(v).  NO feature store here (hazards)
    C1M(0); WR(0); FIN(16); FIN(17);
    C1M(1); WR(1);
    // ---- T1: tap 1.  gaps: the two feature stores, three more ds_writes (the four waves share the CU's LDS store path,
    //      13 cycles per ds_write_b128: one every third gap keeps it unsaturated), pack of conv1(v+1) one VALU each
    //      (unit u reads conv1 result block u >> 3: the first one sits five MFMAs behind C1M(0), unit 8 twelve behind C1M(1))
    sch_tap<PAR, 1, 0>(st, a1); ST(0);
    sch_tap<PAR, 1, 1>(st, a1); WR(2);
    sch_tap<PAR, 1, 2>(st, a1); ST(1);
    sch_tap<PAR, 1, 3>(st, a1); PK(0);
    sch_tap<PAR, 1, 4>(st, a1); WR(3);
    sch_tap<PAR, 1, 5>(st, a1); PK(1);
    sch_tap<PAR, 1, 6>(st, a1); PK(2);
    sch_tap<PAR, 1, 7>(st, a1); WR(4);
    sch_tap<PAR, 1, 8>(st, a1); PK(3);
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
    // ---- T0: tap 0 (5 fresh).  gaps: last pack, hand-off, the 8 reads of partial(v), operand load (every 2nd step)
    sch_tap<PAR, 0, 0>(st, a0); PK(15);
    sch_tap<PAR, 0, 1>(st, a0);
    sch_tap<PAR, 0, 2>(st, a0);
    sch_tap<PAR, 0, 3>(st, a0);
    sch_tap<PAR, 0, 4>(st, a0);
    sch_tap<PAR, 0, 5>(st, a0);
    sch_tap<PAR, 0, 6>(st, a0); HANDOFF();
    sch_tap<PAR, 0, 7>(st, a0); RD(0);
    sch_tap<PAR, 0, 8>(st, a0); RD(1);
    sch_tap<PAR, 0, 9>(st, a0); RD(2);
    sch_tap<PAR, 0, 10>(st, a0); RD(3);
    sch_tap<PAR, 0, 11>(st, a0); RD(4);
    sch_tap<PAR, 0, 12>(st, a0); RD(5);
    sch_tap<PAR, 0, 13>(st, a0); RD(6);
    sch_tap<PAR, 0, 14>(st, a0); RD(7);
    sch_tap<PAR, 0, 15>(st, a0); LD();
    sch_tap<PAR, 0, 16>(st, a0);
    sch_tap<PAR, 0, 17>(st, a0);
    sch_tap<PAR, 0, 18>(st, a0);
    sch_tap<PAR, 0, 19>(st, a0);
    if (ABL == 10 && !LAST) asm volatile("" ::"v"(st.X[0]), "v"(st.X[1]));      // probe 10: conv1 results stay live (async MFMA write)
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

// RANGE = true (small batches: a single window is ONE 16-frame group, i.e. one work-group walking 130 dependent steps
// on one CU while 255 idle): the group's 132 output positions are cut into 11 ranges on blockIdx.y.  Work-group r runs
// steps 12r .. 12r+13 (the last one 120 .. 129 and the tail) with the unchanged step code -- a range starts on a multiple
// of 12, where every register role (v mod 12) is that of step 0 -- and stores outputs 12r+2 .. 12r+13 (r = 0: from 0;
// r = 10: 122 .. 131): the first two outputs a range completes lack the taps of the positions before it and belong to its
// neighbour, which computes them with exactly the instruction sequence of the batch form -- the results are bit-identical.
template <int ABL, bool U8 = false, bool RANGE = false>
__global__ __launch_bounds__(256, 1) void vt_conv_bf16_sched_kernel(const float* __restrict__ x, long n,
                                                                    const u32x4* __restrict__ wq, const u32x4* __restrict__ a1q,
                                                                    const float* __restrict__ b2, unsigned short* __restrict__ feat,
                                                                    long hop2 = 256, float scale = 0.f) {
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
    st.A1[0] = a1q[(q * 2 + 0) * 64 + lane];
    st.A1[1] = a1q[(q * 2 + 1) * 64 + lane];
    // accumulator rows of tile ot on this lane = output channels 16*ot + 4g .. +3
#pragma unroll
    for (int ot = 0; ot < 5; ++ot) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + 16 * ot + 4 * g);
        st.bias[ot] = q == 0 ? b : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+a"(st.bias[ot]));
    }
    const unsigned part_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (size_t)2 * kSImgWords * 4);
    const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // partial tile entry of (frame f, chunk g): 8*(4*(f>>3) + g) + ((f&7) ^ (g>>1)).  Conflict-free for all three
    // accesses (checked with SQ_LDS_BANK_CONFLICT, tools/microbench/lds_patterns.hip): the MFMA-layout writer
    // (lane = f + 16g; ds_write_b128 serves 8 consecutive lanes per cycle over 32 banks: 8 distinct entries mod 8),
    // the transposed reader (lane = 4f + g; ds_read_b128 serves the 16-lane groups of MI355X_MICROARCH.md over 64
    // banks: frames {0,3,5,6} / {1,2,4,7} of a group x 4 chunks = 16 distinct entries mod 16) and the tile-4 word
    // reads (ds_read_b32, 32 lanes over 32 banks).  The first formula (16*(f>>2) + 4*(f&3) + ((g + (f>>2))&3)) was
    // derived for 64 banks on the write side too and ran every ds_write_b128 two-way conflicted.
    auto entry = [](int f, int gg) { return 8 * (4 * (f >> 3) + gg) + ((f & 7) ^ (gg >> 1)); };
    const int fs = lane >> 2, gs = lane & 3;        // finishing role of this lane
    st.gs = gs;
    st.wr_addr = part_lds + (q * 5 * 64 + entry(nl, g)) * 16;
    st.rd_addr = part_lds + (q * 64 + entry(fs, gs)) * 16;
    st.rc_addr = part_lds + (4 * 64 + entry(fs, q)) * 16 + gs * 4;     // tile 4: wave q takes chunk q, lane its word gs
#pragma unroll
    for (int k = 0; k < 4; ++k) { st.rp[k] = f32x4{0.f, 0.f, 0.f, 0.f}; st.rc[k] = 0.f; }
#pragma unroll
    for (int s = 0; s < 3; ++s) st.L0[s] = u32x4{0u, 0u, 0u, 0u};
    st.L1 = u32x4{0u, 0u, 0u, 0u};
    st.cb[0] = st.cb[1] = st.cb[2] = st.cb[3] = 0u;
    st.tprev = 0.f;

    // LDS init: zero padding entries; the second word of every entry of the upper-k-half lanes (32..63) is the
    // constant (1,1) that meets the conv1 bias
    for (int i = tid; i < 2 * kSImgWords; i += 256) img[i] = ((((i / kS) & 63) >= 32) && ((i % kS) & 1)) ? 0x3F803F80u : 0u;
    __syncthreads();
    const long ngroups = (n + 15) >> 4;
    long grp = blockIdx.x;
    if (grp < ngroups)
        for (int k = 0; k < 4; ++k) sch_stage_write(k, stage_decode<U8>(stage_load<U8>(k, x, n, grp * 16, tid, hop2), tid, scale), n, grp * 16, img, tid);
    __syncthreads();
    const int rng = RANGE ? (int)blockIdx.y : 0;      // position range of this work-group
    const int S = 12 * rng;                           // its first step
    const int wlo = rng ? S + 2 : 0;                  // its first output position

    int buf = 0;
    for (; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        st.im_addr = img_lds + (buf * kSImgWords + lane * kS) * 4;
        unsigned short* fbase = feat + (grp * 16 + fs) * (long)(kW2 * kC2);
        const long gnext = RANGE ? ngroups : grp + gridDim.x;      // (a range work-group has no next group to stage)
        // outputs 0 and 1 never see a fresh MFMA: they start from the bias; acc[2] is step 0's fresh accumulator.
        // The copy is an MFMA (0 x 0 + bias), not `acc = bias`: hipcc turned the assignment into v_accvgpr_mov
        // instructions and sank two of them right in front of the asm MFMA that reads those AGPRs as its C operand
        // -- a VALU-write -> MFMA-read hazard it pads for its own MFMAs but not for asm (wrong output 1, now and then).
        f32x4 acc[3][5];
        {
            const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=&a"(acc[0][b]) : "v"(zero), "a"(st.bias[b]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=&a"(acc[1][b]) : "v"(zero), "a"(st.bias[b]));
            }
        }

        // prologue: entries 0..2 (of the range's first chunk), conv1 of position S packed into Bf[0]
        sch_load_even<0>(st, st.im_addr + (S >> 2) * 16);
        sch_load_odd(st, st.im_addr + (S >> 2) * 16 + 8);
        sch_wait_lds(st);
        sch_conv1_mfma<0, 0, 0>(st); sch_conv1_mfma<0, 0, 1>(st);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(st.X[0]), "+v"(st.X[1]));
        [&]<int... N>(std::integer_sequence<int, N...>) { (sch_packop<0, N>(st), ...); }(std::make_integer_sequence<int, 16>{});
        asm volatile("s_nop 1");

        sch_step<0, true, false, ABL, RANGE>(st, S, q, fbase, acc, wlo);
        // batch form: 10 x 12 steps (v = 1 .. 120), then 121 .. 129.  range form: one pass of the same 12-step body
        // (S+1 .. S+12) and step S+13 for the ranges 0 .. 9, none for the last one (120 was its first step)
        const int iters = RANGE ? (rng < 10 ? 1 : 0) : 10;
        int v = S + 1;
        typename StageRaw<U8>::type sv{};
        for (int it = 0; it < iters; ++it, v += 12) {
            // next group's frames -> the other image buffer, a quarter per iteration; each quarter's global load is
            // issued one iteration (12 steps) before its conversion and LDS writes
            if constexpr (!RANGE) {
                if (it >= 6 && gnext < ngroups) sch_stage_write(it - 6, stage_decode<U8>(sv, tid, scale), n, gnext * 16, img + (buf ^ 1) * kSImgWords, tid);
                if (it >= 5 && it < 9 && gnext < ngroups) sv = stage_load<U8>(it - 5, x, n, gnext * 16, tid, hop2);
            }
            sch_step<1, false, false, ABL, RANGE>(st, v + 0, q, fbase, acc, wlo);
            sch_step<2, false, false, ABL, RANGE>(st, v + 1, q, fbase, acc, wlo);
            sch_step<3, false, false, ABL, RANGE>(st, v + 2, q, fbase, acc, wlo);
            sch_step<4, false, false, ABL, RANGE>(st, v + 3, q, fbase, acc, wlo);
            sch_step<5, false, false, ABL, RANGE>(st, v + 4, q, fbase, acc, wlo);
            sch_step<6, false, false, ABL, RANGE>(st, v + 5, q, fbase, acc, wlo);
            sch_step<7, false, false, ABL, RANGE>(st, v + 6, q, fbase, acc, wlo);
            sch_step<8, false, false, ABL, RANGE>(st, v + 7, q, fbase, acc, wlo);
            sch_step<9, false, false, ABL, RANGE>(st, v + 8, q, fbase, acc, wlo);
            sch_step<10, false, false, ABL, RANGE>(st, v + 9, q, fbase, acc, wlo);
            sch_step<11, false, false, ABL, RANGE>(st, v + 10, q, fbase, acc, wlo);
            sch_step<0, false, false, ABL, RANGE>(st, v + 11, q, fbase, acc, wlo);
        }
        const int vt = RANGE ? v : 121;      // S + 13 for the ranges 0 .. 9
        sch_step<1, false, false, ABL, RANGE>(st, vt, q, fbase, acc, wlo);
        if constexpr (RANGE) {
            if (rng < 10) {      // finish of output S+13; the accumulators of S+14, S+15 are dropped
                FinOut fo;
                sch_wait_lds(st);
                sch_fin_all_clamp<1>(st, fo);      // S + 13 is odd: the pair (S + 12, S + 13)
                sch_store<0>(fo, fbase, vt, q, st.gs);
                sch_store<1>(fo, fbase, vt, q, st.gs);
                asm volatile("s_nop 7\n\ts_nop 7" ::"a"(acc[0][0]), "a"(acc[1][0]), "a"(acc[2][0]));      // in-flight MFMA results land before the registers die
                __syncthreads();
                continue;
            }
        }
        sch_step<2, false, false, ABL, RANGE>(st, 122, q, fbase, acc, wlo);
        sch_step<3, false, false, ABL, RANGE>(st, 123, q, fbase, acc, wlo);
        sch_step<4, false, false, ABL, RANGE>(st, 124, q, fbase, acc, wlo);
        sch_step<5, false, false, ABL, RANGE>(st, 125, q, fbase, acc, wlo);
        sch_step<6, false, false, ABL, RANGE>(st, 126, q, fbase, acc, wlo);
        sch_step<7, false, false, ABL, RANGE>(st, 127, q, fbase, acc, wlo);
        sch_step<8, false, false, ABL, RANGE>(st, 128, q, fbase, acc, wlo);
        sch_step<9, false, true, ABL, RANGE>(st, 129, q, fbase, acc, wlo);
        // tail: finish 129, then outputs 130 and 131 (complete as they are: only zero padding beyond).
        // step 129 (v%3 == 0) left output 130 in acc[1] and output 131 in acc[2].
        auto finish_store = [&](auto odd, int w) {
            FinOut fo;
            sch_wait_lds(st);
            sch_fin_all_clamp<decltype(odd)::value>(st, fo);
            sch_store<0>(fo, fbase, w, q, st.gs);
            if constexpr (decltype(odd)::value) sch_store<1>(fo, fbase, w, q, st.gs);
        };
        using Even = std::integral_constant<int, 0>;
        using Odd = std::integral_constant<int, 1>;
        finish_store(Odd{}, 129);
        asm volatile("s_nop 7\n\ts_nop 7");       // last tap-1/tap-0 MFMAs -> ds_write of their accumulators
        sch_part_write<0, 0>(st, acc[1][0]); sch_part_write<0, 1>(st, acc[1][1]); sch_part_write<0, 2>(st, acc[1][2]);
        sch_part_write<0, 3>(st, acc[1][3]); sch_part_write<0, 4>(st, acc[1][4]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        sch_red_load<0>(st);
        asm volatile("" ::"a"(acc[1][0]), "a"(acc[1][1]), "a"(acc[1][2]), "a"(acc[1][3]), "a"(acc[1][4]));
        finish_store(Even{}, 130);
        sch_part_write<1, 0>(st, acc[2][0]); sch_part_write<1, 1>(st, acc[2][1]); sch_part_write<1, 2>(st, acc[2][2]);
        sch_part_write<1, 3>(st, acc[2][3]); sch_part_write<1, 4>(st, acc[2][4]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        sch_red_load<1>(st);
        asm volatile("" ::"a"(acc[2][0]), "a"(acc[2][1]), "a"(acc[2][2]), "a"(acc[2][3]), "a"(acc[2][4]));
        finish_store(Odd{}, 131);
        __syncthreads();      // next group's image is complete; partial buffers are free again
    }
}

}  // namespace

// Operands of this kernel (d_pack slots 6 and 7).
//   conv2 A fragments [q][((ct*3+j)*2+bb)*5+ot][lane][8]: lane (o' = lane&15, kg = lane>>4), slot jj holds
//     K2[16ot+o'][64q + 32ct + 8*(r>>2) + 4*(kg>>1) + (r&3)][h = kg&1][j],  r = 8bb + jj
//     -- the K order in which conv1's 32x32 result arrives (C/D layout: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
//     column = lane&31 = frame + 16*row h of the I/Q pair)
//   conv1 A operands [q][ct][lane][8]: lane (m = lane&31, khalf = lane>>5) = channel 64q + 32ct + m; slots as in the
//     kernel header: khalf 0 (t0h,t1h | t0h,t1h | t2h,0 | t2h,0), khalf 1 (t0l,t1l | b_hi,b_lo | t2l,0 | 0,0)
int vtcnn2_bf16_pack_sched(mdc_model* m) {
    const float* k1 = m->hk[0].data();   // (256,1,1,3)
    const float* b1 = m->hb[0].data();
    const float* k2 = m->hk[1].data();   // (80,256,2,3)
    int rc;
    std::vector<unsigned short> wq((size_t)4 * kWFrags * 64 * 8);
    for (int q = 0; q < 4; ++q)
        for (int ct = 0; ct < 2; ++ct)
            for (int j = 0; j < 3; ++j)
                for (int bb = 0; bb < 2; ++bb)
                    for (int ot = 0; ot < 5; ++ot)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int o = 16 * ot + (lane & 15), kg = lane >> 4, h = kg & 1, r = 8 * bb + jj;
                                const int ch = 64 * q + 32 * ct + 8 * (r >> 2) + 4 * (kg >> 1) + (r & 3);
                                const size_t idx = ((((size_t)q * kWFrags + ((ct * 3 + j) * 2 + bb) * 5 + ot) * 64) + lane) * 8 + jj;
                                wq[idx] = f2bf(k2[(((size_t)o * kC1 + ch) * 2 + h) * 3 + j]);
                            }
    if ((rc = upload(m, 6, wq.data(), wq.size() * 2))) return rc;
    std::vector<unsigned short> a1((size_t)4 * 2 * 64 * 8, 0);
    for (int q = 0; q < 4; ++q)
        for (int ct = 0; ct < 2; ++ct)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = 64 * q + 32 * ct + (lane & 31), khalf = lane >> 5;
                unsigned short* d = &a1[(((size_t)q * 2 + ct) * 64 + lane) * 8];
                // taps and bias times 2^-kFeatShift (exact: the hi / lo split of a scaled value is the scaled split)
                const float sc = std::ldexp(1.f, -kFeatShift);
                unsigned short th[3], tl[3];
                for (int t = 0; t < 3; ++t) {
                    th[t] = f2bf(k1[ch * 3 + t] * sc);
                    tl[t] = f2bf(k1[ch * 3 + t] * sc - bf2f(th[t]));
                }
                if (khalf == 0) {
                    d[0] = th[0]; d[1] = th[1]; d[2] = th[0]; d[3] = th[1]; d[4] = th[2]; d[6] = th[2];
                } else {
                    const unsigned short bh = f2bf(b1[ch] * sc);
                    d[0] = tl[0]; d[1] = tl[1]; d[2] = bh; d[3] = f2bf(b1[ch] * sc - bf2f(bh)); d[4] = tl[2];
                }
            }
    return upload(m, 7, a1.data(), a1.size() * 2);
}

int vtcnn2_bf16_conv_sched(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s, long hop2, float scale) {
    const long ngroups = (n + 15) / 16;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
#define MDC_LAUNCH_SCHED_V(U, R, GRID) do { \
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_bf16_sched_kernel<0, U, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSchedLds)); \
    hipLaunchKernelGGL((vt_conv_bf16_sched_kernel<0, U, R>), GRID, dim3(256), kSchedLds, s, x, (long)n, \
                       static_cast<const u32x4*>(m->d_pack[6]), static_cast<const u32x4*>(m->d_pack[7]), \
                       static_cast<const float*>(m->d_pack[2]), static_cast<unsigned short*>(feat), hop2 > 0 ? hop2 : 256L, scale); \
    MDC_HIP(hipGetLastError()); return MDC_OK; } while (0)
    if (n <= kConvRangeFrames) {      // small batch: the group's positions over 11 work-groups (results identical)
        if (hop2 > 0) MDC_LAUNCH_SCHED_V(true, true, dim3((unsigned)ngroups, 11));
        else MDC_LAUNCH_SCHED_V(false, true, dim3((unsigned)ngroups, 11));
    }
    if (hop2 > 0) MDC_LAUNCH_SCHED_V(true, false, dim3(grid));      // raw uint8 I/Q straight into the staging
#undef MDC_LAUNCH_SCHED_V
#define MDC_LAUNCH_SCHED(A) do { \
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_bf16_sched_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSchedLds)); \
    hipLaunchKernelGGL(vt_conv_bf16_sched_kernel<A>, dim3(grid), dim3(256), kSchedLds, s, x, (long)n, \
                       static_cast<const u32x4*>(m->d_pack[6]), static_cast<const u32x4*>(m->d_pack[7]), \
                       static_cast<const float*>(m->d_pack[2]), static_cast<unsigned short*>(feat), 256L, 0.f); } while (0)
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
