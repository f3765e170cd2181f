// Canonical VT-CNN2 (T3), fp8 mode (MDC_FP8): conv1+conv2 with conv2 on the block-scaled fp8 MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3, scales 2^0), twice the bf16 MFMA rate; f32 accumulation.
// Round 4: the features leave as E4M3 bytes with one power-of-two scale per tensor (F8OUT, the default) -- 10,560 B per
// frame instead of 21,120 written here and read back by dense1, which converts them to bf16 on the way into its MFMA
// (exact) and keeps bf16 weights; see "FEATURES" below.  With MDC_OPT_FP8_BF16_FEATURES the features leave as bf16
// exactly as in the bf16 mode (rounds 1-3), and dense1 and the head are the bf16 mode's kernels.
//
// Same structure as vtcnn2_bf16_sched.hip (weight-stationary in registers, wave q = input channels [64q, 64q+64), the
// K-quarter partials of one output position exchanged per step, asm-sequenced step, helpers of
// vtcnn2_sched_common.h).  What changes:
//   * one conv2 MFMA covers a whole tap of the wave's K quarter: K = 128 = 2 I/Q rows x 64 channels, so a step has
//     15 conv2 MFMAs (3 taps x 5 output tiles) of 32.6 cycles instead of 60 of 16.3, and the weights take 120
//     registers instead of 240;
//   * its B operand is the two 32x32 conv1 results of the step packed to 32 fp8 per lane.  The operand map of the
//     instruction was probed with exact integer data (tools/microbench/mfma_fp8_probe.hip): row / column on
//     lane&15, and the k index a function of (lane>>4, byte) common to A and B -- so the weight fragment simply
//     stores, at (lane>>4, byte), the weight of the (row h, channel) the activation operand carries there;
//   * ReLU + e4m3 pack of the conv1 results is TWO VALU per pair (round 3; rounds 1-2: two v_med3_f32(x, 0, 448) + one
//     v_cvt_pk_fp8_f32 = three): v_cvt_pk_bf16_f32 with the VOP3 clamp bit (ReLU and clamp to [0, 1] inside the
//     conversion; the conv1 operands carry 2^(sa-9), so the e4m3 range 0 .. 448 is 0 .. 0.875 there) and
//     v_cvt_scalef32_pk_fp8_bf16 with scale 2^-9 (probed: tools/microbench/cvt_fp8_bf16_probe.hip -- the result is
//     e4m3(src / scale), written to the half of the dword op_sel names, and with MODE.FP16_OVFL = 1, which the kernel sets,
//     a value beyond 448 saturates instead of turning into NaN: an input beyond the stated range still clips).  32 pack
//     VALU per step instead of 48, and the finish's three v_pk_max_i16 go the same way (clamp bit): with 17 long MFMAs per
//     step the kernel is VALU-issue-bound, so this is where its time is.  The activations are rounded f32 -> bf16 -> e4m3
//     (the second rounding sees 8 significant bits instead of 24: a value within 2^-9 of an e4m3 tie may land on the
//     other side of it; 2^-9 against e4m3's own 2^-4);
//   * SCALING (host side and the MFMA's own block-scale operand): e4m3 spans 2^-9 .. 448.  conv1's taps and bias carry
//     2^(sa-9) (see above: its e4m3 activations are X * 2^sa), conv2's weights 2^sw (powers of two: exact); the MFMA's
//     E8M0 scale operand of B is 2^-(sa+sw+kFeatShift), so the accumulators -- and the bf16 features -- are the true
//     values times 2^-kFeatShift exactly as in the bf16 mode (vtcnn2_sched_common.h): conv2's bias carries 2^-kFeatShift,
//     dense1's weights 2^+kFeatShift, the finish uses the clamp form.  sa is chosen from the largest |sample| the caller
//     expects (mdc_set_fp8_input_absmax, default 0.02 = the reference's frames); a larger input
//     saturates -- "fp8 (scaled inputs)" in the survey's words.
//   * FEATURES (F8OUT): the finish's last partial-sum add carries the VOP3 clamp bit (the sums are the true values times
//     2^-kFeatShift, far below 1: clamp to [0, 1] IS the ReLU, as in v_cvt_pk_bf16_f32 ... clamp of the bf16 finish), then
//     two v_cvt_scalef32_pk_fp8_f32 (divide by st.fsc = 2^-(kFeatShift + kf), round to e4m3, saturate under
//     MODE.FP16_OVFL; tools/microbench/cvt_bf16_fp8_probe.hip) put a position's four channels into ONE dword: the same
//     VALU count as the bf16 finish.  kf = floor(log2(224 / bound)) with bound the largest conv2 output the weights allow
//     for inputs inside the stated range (sum |w2| x conv1 bound + |b2|): never overflows, and sits 4-6 binades below
//     the best data-fitted scale, where the label agreement is flat (profiles/r03_exp_fp8_features.json: k_best .. -6).
//     Layout (feat8_index): output positions in groups of FOUR, 320 B each -- per position pair 128 B = sixteen 8-byte
//     slots {even position's 4 channels, odd position's 4 channels} (one dwordx2 store per pair and lane, none on even
//     steps), then 64 B = channels 64..79 as (w0, w1, w2, w3) bytes (one dword store per four positions).  The
//     position-range form (small batches) splits a group between two work-groups, so it stores that dword's halves as
//     16-bit stores at every odd position.
// Parity: unpinned like every T3 result (no weights bundled); checked against the f64 oracle at 6e-2 of max|logit|.
#include "vtcnn2_bf16_common.h"
#include "vtcnn2_sched_common.h"

#include <cmath>
#include <cstdlib>

namespace mdc {

namespace {

using u32x8 = __attribute__((ext_vector_type(8))) unsigned;
constexpr int kF8Frags = 15;              // [tap j][output tile ot]
constexpr int kF8NV = 7;                  // fragments kept in VGPRs; the other 8 live in AGPRs

struct Fp8State {
    u32x8 Wv[kF8NV];
    u32x8 Wa[kF8Frags - kF8NV];
    f32x4 bias[5];            // scaled conv2 bias tiles: C operand of the fresh MFMAs (wave 0; zeros on waves 1-3)
    u32x4 A1[2];              // conv1 A operands (scaled), 32 channels x 16 k-slots each
    unsigned Bf[2][8];        // [step parity][dword]: B operand of conv2 (32 fp8), as scalars
    unsigned one;             // E8M0 scales 2^0 for the MFMA's A operand
    unsigned scb;             // E8M0 scales 2^-(sa+sw+kFeatShift) for its B operand (every byte the same)
    float sc9;                // 2^-9: divisor of v_cvt_scalef32_pk_fp8_bf16
    f32x16 X[2];
    unsigned T[16];           // ReLU'd conv1 results as packed bf16 pairs
    f32x4 rp[4];
    float rc[4];
    float tprev;              // fifth-tile value of the last even output position (vtcnn2_sched_common.h, sch_fin_clamp)
    unsigned oe;              // F8OUT: the even position's four e4m3 channels, waiting for the odd position's store
    unsigned tt;              // F8OUT: channel 64 + 4q + gs of the four positions of a group, one byte each
    float fsc;                // F8OUT: 2^-(kFeatShift + kf), divisor of v_cvt_scalef32_pk_fp8_f32
    u32x4 L0[3];
    u32x4 L1;
    unsigned cb[4];
    unsigned wr_addr, rd_addr, rc_addr, im_addr;
    int gs;
};

// conv2 MFMA of tap J, output tile OT (the first one of an output, J == 0, takes the bias as C)
// PAD (range form): two wait states in front of an MFMA that opens a block at a control-flow merge / split, where hipcc
// may put AGPR copies of the accumulator (vtcnn2_bf16_sched.hip, sch_tap)
template <int SP, int J, int OT, bool PAD = false>
__device__ __forceinline__ void f8_tap(Fp8State& st, f32x4 (&acc)[5]) {
    constexpr int IDX = J * 5 + OT;
    const u32x8 b = u32x8{st.Bf[SP][0], st.Bf[SP][1], st.Bf[SP][2], st.Bf[SP][3], st.Bf[SP][4], st.Bf[SP][5], st.Bf[SP][6], st.Bf[SP][7]};
    if constexpr (J == 0) {
        if constexpr (IDX < kF8NV) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b), "a"(st.bias[OT]), "v"(st.one), "v"(st.scb));
        else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&a"(acc[OT]) : "a"(st.Wa[IDX - kF8NV]), "v"(b), "a"(st.bias[OT]), "v"(st.one), "v"(st.scb));
    } else if constexpr (PAD) {
        if constexpr (IDX < kF8NV) asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b), "v"(st.one), "v"(st.scb));
        else asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc[OT]) : "a"(st.Wa[IDX - kF8NV]), "v"(b), "v"(st.one), "v"(st.scb));
    } else {
        if constexpr (IDX < kF8NV) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc[OT]) : "v"(st.Wv[IDX]), "v"(b), "v"(st.one), "v"(st.scb));
        else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc[OT]) : "a"(st.Wa[IDX - kF8NV]), "v"(b), "v"(st.one), "v"(st.scb));
    }
}
// pack of conv1's results into Bf[SP], unit K (0..15) = result registers (r0, r0+1) of channel block ct, destined for
// half K&1 of dword d = K>>1 (byte jb = 16*ct + r of the operand holds result register r of channel block ct):
// PKB<K> = v_cvt_pk_bf16_f32 with the clamp bit (ReLU; values in [0, 0.875] for inputs inside the stated range),
// CV<K> = v_cvt_scalef32_pk_fp8_bf16 (divide by 2^-9, round to e4m3, saturate at 448 under MODE.FP16_OVFL)
template <int K>
__device__ __forceinline__ void f8_pkb(Fp8State& st) {
    constexpr int d = K >> 1, ct = d >> 2, r0 = 4 * (d & 3) + 2 * (K & 1);
    const float lo = st.X[ct][r0], hi = st.X[ct][r0 + 1];
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(st.T[K]) : "v"(lo), "v"(hi));
}
template <int SP, int K>
__device__ __forceinline__ void f8_cvt(Fp8State& st) {
    constexpr int d = K >> 1;
    if constexpr ((K & 1) == 0) asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(st.Bf[SP][d]) : "v"(st.T[K]), "v"(st.sc9));
    else asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(st.Bf[SP][d]) : "v"(st.T[K]), "v"(st.sc9));
}

// F8OUT finish of one output position, one VALU per call (K = 0..17; W4 = position & 3): the sum of the four K-quarter
// partials with the ReLU in the last add's clamp bit, then e4m3 bytes.  An even position leaves its dword in st.oe and its
// fifth-tile value in st.tprev; the odd one packs its own into out.o0 and the pair's fifth-tile bytes into a half of st.tt.
template <int K, int W4, class S>
__device__ __forceinline__ void f8_fin(S& st, FinTmp& f, FinOut& out) {
    constexpr int ODD = W4 & 1;
    if constexpr (K < 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.s[K]) : "v"(st.rp[0][K]), "v"(st.rp[1][K]));
    else if constexpr (K < 8) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.u[K - 4]) : "v"(st.rp[2][K - 4]), "v"(st.rp[3][K - 4]));
    else if constexpr (K < 12) asm volatile("v_add_f32 %0, %0, %1 clamp" : "+v"(f.s[K - 8]) : "v"(f.u[K - 8]));
    else if constexpr (K == 12) {
        // ("=v": the first write of the dword; the instruction keeps the other half, which K == 13 then fills)
        if constexpr (ODD) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(out.o0) : "v"(f.s[0]), "v"(f.s[1]), "v"(st.fsc));
        else asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(st.oe) : "v"(f.s[0]), "v"(f.s[1]), "v"(st.fsc));
    } else if constexpr (K == 13) {
        if constexpr (ODD) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(out.o0) : "v"(f.s[2]), "v"(f.s[3]), "v"(st.fsc));
        else asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(st.oe) : "v"(f.s[2]), "v"(f.s[3]), "v"(st.fsc));
    } else if constexpr (K == 14) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.a) : "v"(st.rc[0]), "v"(st.rc[1]));
    else if constexpr (K == 15) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f.b) : "v"(st.rc[2]), "v"(st.rc[3]));
    else if constexpr (K == 16) {
        if constexpr (ODD) asm volatile("v_add_f32 %0, %1, %2 clamp" : "=v"(f.t) : "v"(f.a), "v"(f.b));
        else asm volatile("v_add_f32 %0, %1, %2 clamp" : "=v"(st.tprev) : "v"(f.a), "v"(f.b));
    } else if constexpr (ODD) {
        if constexpr (W4 == 1) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(st.tt) : "v"(st.tprev), "v"(f.t), "v"(st.fsc));
        else asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(st.tt) : "v"(st.tprev), "v"(f.t), "v"(st.fsc));
    }
}
template <int W4, class S>
__device__ __forceinline__ void f8_fin_all(S& st, FinOut& out) {
    FinTmp f;
    [&]<int... K>(std::integer_sequence<int, K...>) { (f8_fin<K, W4>(st, f, out), ...); }(std::make_integer_sequence<int, 18>{});
}
// F8OUT stores of the finishing lane (frame row frow, bytes; feat8_index layout).  WHICH = 0, after an ODD position w:
// channels [16q+4gs, +4) of positions w-1 and w as one dwordx2.  WHICH = 1: channel 64+4q+gs -- batch form: the group's
// dword after its last position (W4 == 3); range form: the pair's two bytes after every odd position.
template <int WHICH, int W4, bool RANGE, class S>
__device__ __forceinline__ void f8_store(const S& st, const FinOut& fo, unsigned char* frow, int w, int q, int gs) {
    unsigned char* grp = frow + (long)(w >> 2) * (4 * kC2);
    if constexpr (WHICH == 0) {
        if constexpr ((W4 & 1) == 1) *reinterpret_cast<u32x2*>(grp + (W4 >> 1) * 128 + 8 * (4 * q + gs)) = u32x2{st.oe, fo.o0};
    } else if constexpr (RANGE) {
        if constexpr (W4 == 1) *reinterpret_cast<unsigned short*>(grp + 256 + 4 * (4 * q + gs)) = (unsigned short)(st.tt & 0xFFFFu);
        else if constexpr (W4 == 3) *reinterpret_cast<unsigned short*>(grp + 256 + 4 * (4 * q + gs) + 2) = (unsigned short)(st.tt >> 16);
    } else {
        if constexpr (W4 == 3) *reinterpret_cast<unsigned*>(grp + 256 + 4 * (4 * q + gs)) = st.tt;
    }
}

// One position step: 15 conv2 MFMAs + 2 conv1 MFMAs, every gap 32 cycles (2 VALU, or 1 VALU + 1 LDS, ride for free).
// RANGE / wlo: the position-range form for small batches (vtcnn2_bf16_sched.hip): outputs below wlo are not stored
template <int V12, bool FIRST, bool LAST, bool RANGE = false, bool F8OUT = true>
__device__ __forceinline__ void f8_step(Fp8State& st, int v, int q, typename std::conditional<F8OUT, unsigned char, unsigned short>::type* fbase,
                                        f32x4 (&acc)[3][5], int wlo = 0) {
    constexpr int PAR = V12 & 1, PN = 1 - PAR;
    constexpr int R1 = (V12 + 1) & 3, S0 = ((V12 + 1) >> 2) % 3, SN = (S0 + 1) % 3;
    constexpr bool kLoadEven = (V12 & 3) == 1 && !LAST, kLoadOdd = (V12 & 3) == 3 && !LAST;
    constexpr int LSLOT = ((V12 >> 2) + 1) % 3;
    f32x4 (&a2)[5] = acc[V12 % 3];
    f32x4 (&a1)[5] = acc[(V12 + 1) % 3];
    f32x4 (&a0)[5] = acc[(V12 + 2) % 3];
    const unsigned load_addr = st.im_addr + ((v >> 2) * 16 + (kLoadEven ? 16 : 24));
    FinTmp ft;
    FinOut fo;
    constexpr int OP = (V12 + 1) & 1;      // parity of the output position v - 1 this step finishes
    constexpr int W4 = (V12 + 3) & 3;      // ... and its place in its group of four (v = V12 mod 12 in every call)
#define FIN(K) do { if (!FIRST) { if constexpr (F8OUT) f8_fin<K, W4>(st, ft, fo); else sch_fin_clamp<K, OP>(st, ft, fo); } } while (0)
#define PREP(I) do { if (!LAST) sch_prep<R1, S0, SN, I>(st); } while (0)
#define C1M(CT) do { if (!LAST) sch_conv1_mfma<R1, S0, CT>(st); } while (0)
#define ST(W) do { if (!FIRST && (W == 0 || OP == 1)) { if (!RANGE || v - 1 >= wlo) { \
        if constexpr (F8OUT) f8_store<W, W4, RANGE>(st, fo, fbase, v - 1, q, st.gs); else sch_store<W>(fo, fbase, v - 1, q, st.gs); } } } while (0)
#define WR(OT) sch_part_write<PAR, OT>(st, a2[OT])
#define PKB(K) do { if (!LAST) f8_pkb<K>(st); } while (0)
#define CV(K) do { if (!LAST) f8_cvt<PN, K>(st); } while (0)
#define RD(R) sch_red_load1<PAR, R>(st)
#define LD() do { if (kLoadEven) sch_load_even<LSLOT>(st, load_addr); else if (kLoadOdd) sch_load_odd(st, load_addr); } while (0)
#ifdef MDC_F8_PROBE_HALF_BARRIER      // timing-only probe (round 5; results wrong by construction): the four-wave rendezvous on EVEN
    // steps only -- what a two-step rendezvous (four partial buffers, a step pattern of period 24) could buy at most
#define HANDOFF() do { \
        if constexpr ((V12 & 1) == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4])); } while (0)
#else
#define HANDOFF() do { \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4])); } while (0)
#endif
    sch_wait_lds(st);
    // 54 VALU per step (rounds 1-2: 73) spread evenly over the 17 gaps: three per gap, four in three of them (a 32-cycle gap
    // takes two VALU for free, a third costs 4 cycles, a fourth 8 more: tools/microbench/mfma_gap.hip).
    // Round 4, timing probes of this step (profiles/r04_fp8_conv_probes.log; code at commit "fp8 conv timing probes"):
    // without the 12 partial-sum adds 15.06 against 14.99 ms -- the K-quarter chain through the MFMA's C operand would buy
    // nothing; without six of the eight partial reads 14.51; without the ReLU / e4m3 pack 12.01; MFMAs, barrier and stores
    // only 11.27.  The pack's 32 VALU are the step's largest item, and the one form that halves them -- relu(a) = a/2 + |a|/2:
    // ONE v_cvt_scalef32_pk_fp8_f32 |x|, |y| per pair, the linear half as six 16x16x32 bf16 MFMAs per step on conv1's own
    // operand -- was built, is bit-level correct and MORE accurate (logits 4.1e-2 against 4.5e-2, labels 0.9873 against
    // 0.9857 on C = 11 noise frames) and is SLOWER: 16.25 ms with the six MFMAs spread between the fp8 ones, 15.8 clustered,
    // against 15.56 (profiles/r04_fp8_abs_form_*_ab.log): six small MFMAs cost what sixteen conversions cost.  Not kept
    // (the kernel file of that form: tools/experiments/vtcnn2_fp8_conv_absform.hip.txt).
    // ORDER INSIDE A GAP (round 4, all bit-identical; profiles/r04_fp8_conv_gap_order*_ab.log): the memory instructions of a gap
    // (ds_write / ds_read / global_store) FIRST, then its VALU -- 15.26 -> 14.2-14.4 ms from that alone, in three steps --; the
    // e4m3 conversions never as the two halves of one dword back to back and never right behind the conversion that produced
    // their source (hipcc puts an s_nop in front of either); conv1's operand dwords (PREP) prepared in the first gaps, far ahead
    // of the MFMA that reads them, and the finish's dependent tail (a, b -> t -> pack) not back to back (-0.7 %).  Measured and
    // not better: memory and VALU alternating; the hand-off one or two gaps earlier; the partial writes two gaps earlier; the
    // feature stores in memory-free gaps; the tile-4 words of the partial read pairwise (ds_read2st64_b32).
    // ---- T2: tap 2 -> a2 complete; finish of output v-1 (18 VALU, ReLU in the conversions' clamp bit), conv1 operand dwords of v+1
    f8_tap<PAR, 2, 0, RANGE && (V12 == 1 || V12 == 2)>(st, a2); PREP(0); FIN(0); FIN(1);
    f8_tap<PAR, 2, 1>(st, a2); PREP(1); FIN(2); FIN(3);
    f8_tap<PAR, 2, 2>(st, a2); PREP(2); FIN(4); FIN(5);
    f8_tap<PAR, 2, 3>(st, a2); PREP(3); FIN(6); FIN(7); FIN(8);
    f8_tap<PAR, 2, 4>(st, a2); FIN(9); FIN(10); FIN(11);
    C1M(0); WR(0); FIN(14); FIN(15); FIN(12);
    C1M(1); WR(1); FIN(16); FIN(13); FIN(17);
    f8_tap<PAR, 1, 0>(st, a1); ST(0); WR(2); PKB(0); PKB(1); PKB(2);
    f8_tap<PAR, 1, 1>(st, a1); ST(1); WR(3); PKB(3); PKB(4); PKB(5);
    f8_tap<PAR, 1, 2>(st, a1); WR(4); LD(); PKB(6); PKB(7); CV(0);
    f8_tap<PAR, 1, 3>(st, a1); CV(2); CV(4); CV(6);
    f8_tap<PAR, 1, 4>(st, a1); CV(1); PKB(8); PKB(9); HANDOFF();
    f8_tap<PAR, 0, 0>(st, a0); RD(0); RD(1); CV(3); PKB(10); PKB(11);
    f8_tap<PAR, 0, 1>(st, a0); RD(2); RD(3); CV(5); PKB(12); PKB(13);
    f8_tap<PAR, 0, 2>(st, a0); RD(4); RD(5); CV(7); CV(8); PKB(14); PKB(15);
    f8_tap<PAR, 0, 3>(st, a0); RD(6); RD(7); CV(10); CV(12); CV(9); CV(14);
    f8_tap<PAR, 0, 4>(st, a0); CV(11); CV(13); CV(15);
#undef FIN
#undef PREP
#undef C1M
#undef ST
#undef WR
#undef PKB
#undef CV
#undef RD
#undef LD
#undef HANDOFF
}

// RANGE: grid (groups, 11), work-group (g, r) runs steps 12r .. 12r+13 (r = 10: 120 .. 129 and the tail) and stores the
// outputs 12r+2 .. 12r+13 (r = 0: from 0; r = 10: 122 .. 131) -- see vt_conv_bf16_sched_kernel
template <bool U8, bool RANGE = false, bool F8OUT = true>
__global__ __launch_bounds__(256, 1) void vt_conv_fp8_kernel(const float* __restrict__ x, long n,
                                                             const u32x8* __restrict__ wq, const u32x4* __restrict__ a1q,
                                                             const float* __restrict__ b2, void* __restrict__ feat_out,
                                                             long hop2, float scale, unsigned scale_b_e8m0, float feat_divisor) {
    using FeatT = typename std::conditional<F8OUT, unsigned char, unsigned short>::type;      // e4m3 bytes / bf16
    FeatT* __restrict__ feat = static_cast<FeatT*>(feat_out);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* img = reinterpret_cast<unsigned*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, g = lane >> 4;

    Fp8State st;
#pragma unroll
    for (int i = 0; i < kF8Frags; ++i) {
        const u32x8 w = wq[(q * kF8Frags + i) * 64 + lane];
        if (i < kF8NV) { st.Wv[i] = w; asm volatile("" : "+v"(st.Wv[i])); }
        else { st.Wa[i - kF8NV] = w; asm volatile("" : "+a"(st.Wa[i - kF8NV])); }
    }
    st.A1[0] = a1q[(q * 2 + 0) * 64 + lane];
    st.A1[1] = a1q[(q * 2 + 1) * 64 + lane];
    // e4m3 conversions saturate at 448 instead of producing NaN (probed: tools/microbench/cvt_fp8_bf16_probe.hip)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    st.one = 0x7F7F7F7Fu;
    st.scb = scale_b_e8m0 * 0x01010101u;
    st.sc9 = 0.001953125f;
    st.fsc = feat_divisor;
    st.oe = st.tt = 0u;
    asm volatile("" : "+v"(st.one), "+v"(st.scb), "+v"(st.sc9), "+v"(st.fsc), "+v"(st.oe), "+v"(st.tt));
#pragma unroll
    for (int k = 0; k < 16; ++k) st.T[k] = 0u;
#pragma unroll
    for (int ot = 0; ot < 5; ++ot) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + 16 * ot + 4 * g);
        st.bias[ot] = q == 0 ? b : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+a"(st.bias[ot]));
    }
    const unsigned part_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (size_t)2 * kSImgWords * 4);
    const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto entry = [](int f, int gg) { return 8 * (4 * (f >> 3) + gg) + ((f & 7) ^ (gg >> 1)); };      // see vtcnn2_bf16_sched.hip
    const int fs = lane >> 2, gs = lane & 3;
    st.gs = gs;
    st.wr_addr = part_lds + (q * 5 * 64 + entry(nl, g)) * 16;
    st.rd_addr = part_lds + (q * 64 + entry(fs, gs)) * 16;
    st.rc_addr = part_lds + (4 * 64 + entry(fs, q)) * 16 + gs * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { st.rp[k] = f32x4{0.f, 0.f, 0.f, 0.f}; st.rc[k] = 0.f; }
#pragma unroll
    for (int s = 0; s < 3; ++s) st.L0[s] = u32x4{0u, 0u, 0u, 0u};
    st.L1 = u32x4{0u, 0u, 0u, 0u};
    st.cb[0] = st.cb[1] = st.cb[2] = st.cb[3] = 0u;
    st.tprev = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) st.Bf[0][d] = st.Bf[1][d] = 0u;

    for (int i = tid; i < 2 * kSImgWords; i += 256) img[i] = ((((i / kS) & 63) >= 32) && ((i % kS) & 1)) ? 0x3F803F80u : 0u;
    __syncthreads();
    const long ngroups = (n + 15) >> 4;
    long grp = blockIdx.x;
    if (grp < ngroups)
        for (int k = 0; k < 4; ++k) sch_stage_write(k, stage_decode<U8>(stage_load<U8>(k, x, n, grp * 16, tid, hop2), tid, scale), n, grp * 16, img, tid);
    __syncthreads();
    const int rng = RANGE ? (int)blockIdx.y : 0;
    const int S = 12 * rng;
    const int wlo = rng ? S + 2 : 0;

    int buf = 0;
    for (; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        st.im_addr = img_lds + (buf * kSImgWords + lane * kS) * 4;
        FeatT* fbase = feat + (grp * 16 + fs) * (long)(kW2 * kC2);
        const long gnext = RANGE ? ngroups : grp + gridDim.x;
        // accumulators of outputs 0 and 1 start from the bias: an MFMA (0 x 0 + bias), never a compiler AGPR copy
        f32x4 acc[3][5];
        {
            const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=&a"(acc[0][b]) : "v"(zero), "a"(st.bias[b]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=&a"(acc[1][b]) : "v"(zero), "a"(st.bias[b]));
            }
        }
        // prologue: entries 0..2 of the range's first chunk, conv1 of position S packed into Bf[0]
        sch_load_even<0>(st, st.im_addr + (S >> 2) * 16);
        sch_load_odd(st, st.im_addr + (S >> 2) * 16 + 8);
        sch_wait_lds(st);
        sch_conv1_mfma<0, 0, 0>(st); sch_conv1_mfma<0, 0, 1>(st);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(st.X[0]), "+v"(st.X[1]));
        [&]<int... K>(std::integer_sequence<int, K...>) { (f8_pkb<K>(st), ...); }(std::make_integer_sequence<int, 16>{});
        [&]<int... K>(std::integer_sequence<int, K...>) { (f8_cvt<0, K>(st), ...); }(std::make_integer_sequence<int, 16>{});
        asm volatile("s_nop 1");

        f8_step<0, true, false, RANGE, F8OUT>(st, S, q, fbase, acc, wlo);
        // batch form: 10 x 12 steps (v = 1 .. 120), then 121 .. 129; range form: one pass of the 12-step body and step
        // S+13 for the ranges 0 .. 9, none for the last one
        const int iters = RANGE ? (rng < 10 ? 1 : 0) : 10;
        int v = S + 1;
        typename StageRaw<U8>::type sv{};
        for (int it = 0; it < iters; ++it, v += 12) {
            if constexpr (!RANGE) {
                if (it >= 6 && gnext < ngroups) sch_stage_write(it - 6, stage_decode<U8>(sv, tid, scale), n, gnext * 16, img + (buf ^ 1) * kSImgWords, tid);
                if (it >= 5 && it < 9 && gnext < ngroups) sv = stage_load<U8>(it - 5, x, n, gnext * 16, tid, hop2);
            }
            f8_step<1, false, false, RANGE, F8OUT>(st, v + 0, q, fbase, acc, wlo);
            f8_step<2, false, false, RANGE, F8OUT>(st, v + 1, q, fbase, acc, wlo);
            f8_step<3, false, false, RANGE, F8OUT>(st, v + 2, q, fbase, acc, wlo);
            f8_step<4, false, false, RANGE, F8OUT>(st, v + 3, q, fbase, acc, wlo);
            f8_step<5, false, false, RANGE, F8OUT>(st, v + 4, q, fbase, acc, wlo);
            f8_step<6, false, false, RANGE, F8OUT>(st, v + 5, q, fbase, acc, wlo);
            f8_step<7, false, false, RANGE, F8OUT>(st, v + 6, q, fbase, acc, wlo);
            f8_step<8, false, false, RANGE, F8OUT>(st, v + 7, q, fbase, acc, wlo);
            f8_step<9, false, false, RANGE, F8OUT>(st, v + 8, q, fbase, acc, wlo);
            f8_step<10, false, false, RANGE, F8OUT>(st, v + 9, q, fbase, acc, wlo);
            f8_step<11, false, false, RANGE, F8OUT>(st, v + 10, q, fbase, acc, wlo);
            f8_step<0, false, false, RANGE, F8OUT>(st, v + 11, q, fbase, acc, wlo);
        }
        const int vt = RANGE ? v : 121;
        f8_step<1, false, false, RANGE, F8OUT>(st, vt, q, fbase, acc, wlo);
        if constexpr (RANGE) {
            if (rng < 10) {      // finish of output S+13; the accumulators of S+14, S+15 are dropped
                FinOut fo;
                sch_wait_lds(st);
                if constexpr (F8OUT) {              // S + 13 is odd and second in its group of four: the pair (S + 12, S + 13)
                    f8_fin_all<1>(st, fo);
                    f8_store<0, 1, true>(st, fo, fbase, vt, q, st.gs);
                    f8_store<1, 1, true>(st, fo, fbase, vt, q, st.gs);
                } else {
                    sch_fin_all_clamp<1>(st, fo);
                    sch_store<0>(fo, fbase, vt, q, st.gs);
                    sch_store<1>(fo, fbase, vt, q, st.gs);
                }
                asm volatile("s_nop 7\n\ts_nop 7" ::"a"(acc[0][0]), "a"(acc[1][0]), "a"(acc[2][0]));
                __syncthreads();
                continue;
            }
        }
        f8_step<2, false, false, RANGE, F8OUT>(st, 122, q, fbase, acc, wlo);
        f8_step<3, false, false, RANGE, F8OUT>(st, 123, q, fbase, acc, wlo);
        f8_step<4, false, false, RANGE, F8OUT>(st, 124, q, fbase, acc, wlo);
        f8_step<5, false, false, RANGE, F8OUT>(st, 125, q, fbase, acc, wlo);
        f8_step<6, false, false, RANGE, F8OUT>(st, 126, q, fbase, acc, wlo);
        f8_step<7, false, false, RANGE, F8OUT>(st, 127, q, fbase, acc, wlo);
        f8_step<8, false, false, RANGE, F8OUT>(st, 128, q, fbase, acc, wlo);
        f8_step<9, false, true, RANGE, F8OUT>(st, 129, q, fbase, acc, wlo);
        // tail: finish 129, then outputs 130 and 131 (complete as they are: only zero padding beyond).
        // step 129 (v%3 == 0) left output 130 in acc[1] and output 131 in acc[2].
        auto finish_store = [&](auto w4, int w) {      // w4 = w & 3 (129, 130, 131 -> 1, 2, 3)
            constexpr int W4 = decltype(w4)::value;
            FinOut fo;
            sch_wait_lds(st);
            if constexpr (F8OUT) {
                f8_fin_all<W4>(st, fo);
                f8_store<0, W4, RANGE>(st, fo, fbase, w, q, st.gs);
                f8_store<1, W4, RANGE>(st, fo, fbase, w, q, st.gs);
            } else {
                sch_fin_all_clamp<W4 & 1>(st, fo);
                sch_store<0>(fo, fbase, w, q, st.gs);
                if constexpr (W4 & 1) sch_store<1>(fo, fbase, w, q, st.gs);
            }
        };
        using Even = std::integral_constant<int, 2>;
        using Odd = std::integral_constant<int, 1>;
        using Last = std::integral_constant<int, 3>;
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
        finish_store(Last{}, 131);
        __syncthreads();      // next group's image is complete; partial buffers are free again
    }
}

}  // namespace

// e4m3 (OCP "fn": no infinities, max 448, NaN = 0x7F) with round-to-nearest-even; saturates
static unsigned char f2e4m3(float f) {
    if (std::isnan(f)) return 0x7F;
    const unsigned char sign = std::signbit(f) ? 0x80 : 0x00;
    float a = std::fabs(f);
    if (a >= 448.f) return sign | 0x7E;
    if (a < std::ldexp(1.f, -10)) return sign;                       // below half the smallest subnormal (2^-9)
    int e;
    (void)std::frexp(a, &e);                                         // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;                                                   // a = 1.xxx * 2^E
    if (E < -6) E = -6;                                              // subnormal: fixed exponent
    const float q = std::nearbyint(std::ldexp(a, 3 - E));            // mantissa with 3 fractional bits (RNE by default)
    int m = (int)q;                                                  // 8..16 for normals, 0..8 for subnormals
    if (m == 16) { m = 8; ++E; }
    if (E > 8 || (E == 8 && m > 14)) return sign | 0x7E;
    if (m < 8) return sign | (unsigned char)m;                       // subnormal: exponent field 0
    return sign | (unsigned char)(((E + 7) << 3) | (m - 8));
}

// d_pack slots in fp8 mode: 0 conv2 fp8 fragments, 1 conv1 operands (scaled), 2 conv2 bias (scaled),
// 3 dense1 weights (bf16, divided by the feature scale), 4 dense1 bias, 5 head.  Returns the feature scale exponent.
int vtcnn2_fp8_pack(mdc_model* m) {
    const float* k1 = m->hk[0].data();   // (256,1,1,3)
    const float* b1 = m->hb[0].data();
    const float* k2 = m->hk[1].data();   // (80,256,2,3)
    int rc;
    // scales: powers of two
    float w2max = 0.f, c1bound = 0.f;
    for (size_t i = 0; i < m->hk[1].size(); ++i) w2max = std::fmax(w2max, std::fabs(k2[i]));
    for (int ch = 0; ch < kC1; ++ch) {
        const float bound = m->fp8_input_absmax * (std::fabs(k1[ch * 3]) + std::fabs(k1[ch * 3 + 1]) + std::fabs(k1[ch * 3 + 2])) + std::fabs(b1[ch]);
        c1bound = std::fmax(c1bound, bound);
    }
    if (!(w2max > 0.f) || !(c1bound > 0.f)) { set_error("fp8: degenerate weights (all zero)"); return MDC_EINVAL; }
    const int sw = (int)std::floor(std::log2(224.f / w2max));      // a factor 2 of head-room below 448
    const int sa = (int)std::floor(std::log2(224.f / c1bound));
    const float fsw = std::ldexp(1.f, sw), fsa = std::ldexp(1.f, sa - 9);      // conv1 operands: the e4m3 range sits in [0, 0.875] (see the header)
    // the MFMA's E8M0 block scale of B takes the accumulators from 2^(sa+sw) to 2^-kFeatShift times the true values
    const int e8 = 127 - (sa + sw + kFeatShift);
    if (e8 < 1 || e8 > 254) { set_error("fp8: weight / input scales out of the block-scale range (sa %d, sw %d)", sa, sw); return MDC_EINVAL; }
    m->fp8_feat_scale_log2 = e8;      // (kept under its old name: the E8M0 byte the conv kernel is launched with)
    m->feat_scale_log2 = -kFeatShift;

    std::vector<unsigned char> wq((size_t)4 * kF8Frags * 64 * 32);
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 3; ++j)
            for (int ot = 0; ot < 5; ++ot)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jb = 0; jb < 32; ++jb) {
                        const int o = 16 * ot + (lane & 15), kg = lane >> 4, h = kg & 1, ct = jb >> 4, r = jb & 15;
                        const int ch = 64 * q + 32 * ct + 8 * (r >> 2) + 4 * (kg >> 1) + (r & 3);
                        wq[((((size_t)q * kF8Frags + j * 5 + ot) * 64) + lane) * 32 + jb] = f2e4m3(k2[(((size_t)o * kC1 + ch) * 2 + h) * 3 + j] * fsw);
                    }
    if ((rc = upload(m, 0, wq.data(), wq.size()))) return rc;
    std::vector<unsigned short> a1((size_t)4 * 2 * 64 * 8, 0);      // as vtcnn2_bf16_pack_sched, taps and bias x 2^sa
    for (int q = 0; q < 4; ++q)
        for (int ct = 0; ct < 2; ++ct)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = 64 * q + 32 * ct + (lane & 31), khalf = lane >> 5;
                unsigned short* d = &a1[(((size_t)q * 2 + ct) * 64 + lane) * 8];
                unsigned short th[3], tl[3];
                for (int t = 0; t < 3; ++t) {
                    th[t] = f2bf(k1[ch * 3 + t] * fsa);
                    tl[t] = f2bf(k1[ch * 3 + t] * fsa - bf2f(th[t]));
                }
                if (khalf == 0) {
                    d[0] = th[0]; d[1] = th[1]; d[2] = th[0]; d[3] = th[1]; d[4] = th[2]; d[6] = th[2];
                } else {
                    const unsigned short bh = f2bf(b1[ch] * fsa);
                    d[0] = tl[0]; d[1] = tl[1]; d[2] = bh; d[3] = f2bf(b1[ch] * fsa - bf2f(bh)); d[4] = tl[2];
                }
            }
    if ((rc = upload(m, 1, a1.data(), a1.size() * 2))) return rc;
    std::vector<float> b2s(kC2);
    for (int o = 0; o < kC2; ++o) b2s[o] = std::ldexp(m->hb[1][o], -kFeatShift);
    if ((rc = upload(m, 2, b2s.data(), b2s.size() * sizeof(float)))) return rc;
    // dense1.  bf16 features (MDC_OPT_FP8_BF16_FEATURES): exactly the bf16 mode's (its features carry the same 2^-kFeatShift).
    // E4M3 features: true value x 2^kf, K in feat8_index order; the weights carry 2^-kf (a power of two: exact in bf16).
    // kf = floor(log2(224 / bound)), with `bound` the largest conv2 output to be REPRESENTED (anything beyond 2 x bound
    // saturates at 448 under MODE.FP16_OVFL -- the backstop).  Round 5: the bound is no longer the worst case
    // the weights allow (sum |w2| x conv1 bound: on the synthetic weights 46 x the largest feature N(0, 5e-3) frames produce
    // and 29 x what a full-scale sinusoid produces -- 5 binades of E4M3's 17 unused, 4 % of the non-zero features subnormal),
    // but either what the caller measured on a sample (mdc_set_fp8_feature_absmax), or a statistical estimate from the
    // weights: per output channel |b2| + 12 x the rms of the conv2 sum for independent samples of rms absmax / 4 --
    //   sqrt( sum_{c,h,j} w2^2 x (absmax^2/16 x sum_t k1[c][t]^2 + b1[c]^2) )
    // (12 sigma: a full-scale sinusoid reaches 6.9 of these rms units, signal-shaped frames 6.2, Gaussian noise 4.4;
    // tools/measure_bars.py records the occupancy actually reached).
    m->fp8_e4m3_features = (m->topo.reserved[0] & MDC_OPT_FP8_BF16_FEATURES) == 0;
    int kf = 0;
    if (m->fp8_e4m3_features) {
        float bound = 0.f;
        if (m->fp8_feature_absmax > 0.f) {
            bound = m->fp8_feature_absmax;
        } else {
            const double r2 = (double)m->fp8_input_absmax * m->fp8_input_absmax / 16.0;
            for (int o = 0; o < kC2; ++o) {
                double var = 0.0;
                for (int ch = 0; ch < kC1; ++ch) {
                    const double c1ms = r2 * ((double)k1[ch * 3] * k1[ch * 3] + (double)k1[ch * 3 + 1] * k1[ch * 3 + 1] + (double)k1[ch * 3 + 2] * k1[ch * 3 + 2])
                                        + (double)b1[ch] * b1[ch];
                    double w2s = 0.0;
                    for (int t = 0; t < 6; ++t) { const double wv = k2[((size_t)o * kC1 + ch) * 6 + t]; w2s += wv * wv; }
                    var += w2s * c1ms;
                }
                bound = std::fmax(bound, (float)(std::fabs((double)m->hb[1][o]) + 12.0 * std::sqrt(var)));
            }
        }
        if (!(bound > 0.f) || !std::isfinite(bound)) { set_error("fp8: degenerate conv2 output bound"); return MDC_EINVAL; }
        kf = (int)std::floor(std::log2(224.f / bound));
        if (kFeatShift + kf < -120 || kFeatShift + kf > 120) { set_error("fp8: feature scale 2^%d out of range", kf); return MDC_EINVAL; }
        m->feat_scale_log2 = kf;
        m->fp8_feat_divisor = std::ldexp(1.f, -(kFeatShift + kf));      // sums are true x 2^-kFeatShift; bytes are true x 2^kf
    }
    const float* w1 = m->hk[2].data();
    const float inv = std::ldexp(1.f, m->fp8_e4m3_features ? -kf : kFeatShift);
    std::vector<unsigned short> w1t((size_t)kHid * kFeat);
    for (int w = 0; w < kW2; ++w)
        for (int o = 0; o < kC2; ++o) {
            const float* srcw = w1 + (size_t)(o * kW2 + w) * kHid;
            const int kk = m->fp8_e4m3_features ? feat8_index(w, o) : feat16_index(w, o);
            for (int nn = 0; nn < kHid; ++nn) w1t[((size_t)(kk >> 6) * kHid + nn) * 64 + (kk & 63)] = f2bf(srcw[nn] * inv);
        }
    return upload(m, 3, w1t.data(), w1t.size() * 2);
}

int vtcnn2_fp8_conv(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s, long hop2, float scale) {
    const long ngroups = (n + 15) / 16;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
#define MDC_LAUNCH_F8(U, R, F, GRID) do { \
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_fp8_kernel<U, R, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSchedLds)); \
    hipLaunchKernelGGL((vt_conv_fp8_kernel<U, R, F>), GRID, dim3(256), kSchedLds, s, x, (long)n, \
                       static_cast<const u32x8*>(m->d_pack[0]), static_cast<const u32x4*>(m->d_pack[1]), \
                       static_cast<const float*>(m->d_pack[2]), feat, hop2 > 0 ? hop2 : 256L, scale, \
                       (unsigned)m->fp8_feat_scale_log2, m->fp8_feat_divisor); } while (0)
#define MDC_LAUNCH_F8_UR(U, R, GRID) do { if (m->fp8_e4m3_features) MDC_LAUNCH_F8(U, R, true, GRID); else MDC_LAUNCH_F8(U, R, false, GRID); } while (0)
    // hop2 > 0: raw uint8 I/Q straight into the staging.  Small batches: the position-range form (results identical)
    if (n <= kConvRangeFrames) {
        if (hop2 > 0) MDC_LAUNCH_F8_UR(true, true, dim3((unsigned)ngroups, 11));
        else MDC_LAUNCH_F8_UR(false, true, dim3((unsigned)ngroups, 11));
    } else {
        if (hop2 > 0) MDC_LAUNCH_F8_UR(true, false, dim3(grid));
        else MDC_LAUNCH_F8_UR(false, false, dim3(grid));
    }
#undef MDC_LAUNCH_F8_UR
#undef MDC_LAUNCH_F8
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
