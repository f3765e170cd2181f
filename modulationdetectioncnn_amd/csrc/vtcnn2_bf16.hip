// Canonical VT-CNN2 (T3), bf16 MFMA path (f32 accumulation).  See vtcnn2.hip for the math and
// the "lane = frame" mapping.
//
// vt_conv_bf16_kernel -- WEIGHT-STATIONARY IN REGISTERS.  (This is the hipcc-scheduled statement of the algorithm,
// selectable with MDC_CONV_SCHED=0; the production kernel, vtcnn2_bf16_sched.hip, keeps the algorithm but writes the
// instruction order by hand, runs conv1 on the 32x32 MFMA shape and uses its own image and operand order.)  The conv2
// kernel tensor is 80 x 1536 bf16 = 240 KiB: too big for the 160 KiB LDS, but a CU's four SIMDs hold 512 KiB
// of registers.  One workgroup = 4 waves (one per SIMD, 512 VGPR+AGPR each); wave q keeps the
// conv2 weights of input channels [64q, 64q+64) for all 80 outputs, 2 rows and 3 taps:
// 60 A-fragments x 4 VGPRs = 240 registers, loaded once per kernel.  The workgroup walks
// groups of 16 frames; per group every wave sweeps the 130 conv1 positions:
//     8 x v_mfma_f32_16x16x16_bf16   conv1 of its 64 channels x 2 rows  -> X[channel][frame]
//     ReLU + v_cvt_pk_bf16_f32       X is ALREADY the B-operand layout of the next MFMA
//    60 x v_mfma_f32_16x16x32_bf16   conv2: 2 rows x 2 channel pairs x 3 taps x 5 output tiles,
//                                    tap j accumulates into the registers of output w'-j
// so activations never touch LDS and weights never move.  The only exchange is the sum of the
// four waves' K-quarter partials of ONE output position per step (5 KB each) through LDS,
// after which bias + ReLU + bf16 and the store of feat[frame][w][0..80).
// conv1's MFMA has K = 16 slots and needs 3: the spare slots carry the low-order bf16 halves
// of the input samples, of the conv1 taps and of the conv1 bias, so conv1 is computed to
// ~2^-16 relative accuracy although every operand is bf16.
//
// This file: vt_conv_bf16_kernel (hipcc-scheduled statement of the algorithm), the host-side packing of all
// bf16 operands, and the conv launcher.  vtcnn2_bf16_sched.hip holds the production (asm-sequenced) conv
// kernel, vtcnn2_bf16_dense1.hip the dense1 GEMM.
#include "vtcnn2_bf16_common.h"
#include "vtcnn2_sched_common.h"

#include <cmath>

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mdc {

namespace {

#ifdef MDC_ALTERNATES      // the hipcc-scheduled statement of the algorithm: 36 % slower than the asm-sequenced kernel; test build only
__global__ __launch_bounds__(256, 1) void vt_conv_bf16_kernel(const float* __restrict__ x, long n,
                                                              const u32x4* __restrict__ wq,   // [4][60][64]
                                                              const u32x2* __restrict__ a1q,  // [4][4][64]
                                                              const float* __restrict__ b2,   // [80]
                                                              unsigned short* __restrict__ feat) {   // [ceil16(n)][132][80] bf16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* img = reinterpret_cast<unsigned*>(smem);
    float* part = reinterpret_cast<float*>(smem + (size_t)2 * kImgWords * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, g = lane >> 4;

    // ---- stationary operands ----
    bf16x8 W[2][3][2][5];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int ot = 0; ot < 5; ++ot)
                    W[h][j][cp][ot] = __builtin_bit_cast(bf16x8, wq[(q * kWFrags + ((h * 3 + j) * 2 + cp) * 5 + ot) * 64 + lane]);
    s16x4 A1[4];          // taps at k-slots 0..2 (even positions); odd positions shift the B operand instead
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) A1[ct] = __builtin_bit_cast(s16x4, a1q[(q * 4 + ct) * 64 + lane]);
    // conv2 bias: kept in LDS and read where it is used (it would cost 5 persistent registers in a kernel that
    // sits exactly at the register limit); this wave finishes output tile q and component q of tile 4
    float* bias_lds = part + 2 * kPartFloats;
    if (tid < kC2) bias_lds[tid] = b2[tid];
    const float* bq_lds = bias_lds + 16 * q + 4 * g;
    const float* b4_lds = bias_lds + 64 + 4 * g + q;

    // ---- LDS init: zero padding pairs and the constant bias-slot lanes, both buffers ----
    for (int i = tid; i < 2 * kImgWords; i += 256) img[i] = ((i & 63) >= 48) ? 0x3F803F80u : 0u;
    __syncthreads();

    const long ngroups = (n + 15) >> 4;
    long grp = blockIdx.x;
    if (grp < ngroups)
        for (int k = 0; k < 4; ++k) stage_quarter(k, x, n, grp * 16, img, tid);
    __syncthreads();

    int buf = 0;
    for (; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        const unsigned* im = img + buf * kImgWords + lane;
        const long frame0 = grp * 16;
        const long fme = frame0 + nl;
        unsigned short* fbase = feat + fme * (long)(kW2 * kC2) + 4 * g;
        const long gnext = grp + gridDim.x;

        f32x4 acc[3][5];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 Bf[2][2][2];   // [step parity][row][channel pair]: the pack for step v+1 never overwrites operands of step v
        f32x4 X[4][2];        // conv1 tiles [channel tile][row]; row 0 and row 1 are live at different times
        f32x4 rp[4];
        float rc[4];
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;

        // conv1 at output index v (padded position v+2): pair index i = v>>1, taps start at slot v&1
        // The builtin's result always lands in AGPRs in this kernel (512-register budget) and every value
        // then costs a v_accvgpr_read before the VALU can convert it; the asm form names a VGPR destination.
        // hipcc neither counts wait states for asm nor knows this MFMA's latency: x_fence(h) supplies the
        // XDL-write -> VALU-read wait states before the first pack of row h.
        // operand words of row h for conv1 at output index v: pairs i, i+1 (+ i+2 for odd v), i = v>>1.
        // Loaded one phase BEFORE the conv1 that uses them so the LDS latency is never exposed.
        unsigned bw[2][3];
        auto conv1_load = [&](int v, auto par_tag, auto h_tag) {
            constexpr int PAR = decltype(par_tag)::value, h = decltype(h_tag)::value;
            const unsigned* pi = im + (h * kPairs + (v >> 1)) * 64;
            bw[h][0] = pi[0];
            bw[h][1] = pi[64];
            if (PAR == 1) bw[h][2] = pi[128];
        };
        auto conv1 = [&](auto par_tag, auto h_tag) {
            constexpr int PAR = decltype(par_tag)::value, h = decltype(h_tag)::value;
            u32x2 b;
            if (PAR == 0) b = u32x2{bw[h][0], bw[h][1]};                       // samples v, v+1 | v+2, v+3
            else b = u32x2{__builtin_amdgcn_alignbit(bw[h][1], bw[h][0], 16),    // odd v: start one sample later
                           __builtin_amdgcn_alignbit(bw[h][2], bw[h][1], 16)};
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4& xd = X[ct][h];            // (asm operands cannot name captured arrays directly)
                const s16x4& a1 = A1[ct];
                // "=&v": the result must not share registers with an operand.  s_nop 1 before the first one:
                // a VALU (v_alignbit) may have written the B operand in the previous cycle (VALU write -> MFMA
                // read needs 2 wait states and hipcc does not pad asm).
                if (ct == 0) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=&v"(xd) : "v"(a1), "v"(b));
                else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=&v"(xd) : "v"(a1), "v"(b));
            }
        };
        auto x_fence = [&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value;
            f32x4 &x0 = X[0][h], &x1 = X[1][h], &x2 = X[2][h], &x3 = X[3][h];
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        };
        // ReLU + bf16 of the conv1 tile pair (row h, channel pair cp): already a B operand
        auto pack = [&](auto sp_tag, auto h_tag, auto cp_tag) {
            constexpr int sp = decltype(sp_tag)::value, h = decltype(h_tag)::value, cp = decltype(cp_tag)::value;
            const f32x4 t0 = X[2 * cp][h], t1 = X[2 * cp + 1][h];
            const u32x4 pk = u32x4{pack2relu(t0[0], t0[1]), pack2relu(t0[2], t0[3]), pack2relu(t1[0], t1[1]), pack2relu(t1[2], t1[3])};
            Bf[sp][h][cp] = __builtin_bit_cast(bf16x8, pk);
        };
        // conv2, one tap j of one (row, channel pair): 5 MFMAs into the accumulators of output w'-j
        auto tap = [&](auto sp_tag, auto j_tag, auto h_tag, auto cp_tag, f32x4 (&a)[5], bool fresh) {
            constexpr int sp = decltype(sp_tag)::value, j = decltype(j_tag)::value, h = decltype(h_tag)::value, cp = decltype(cp_tag)::value;
#pragma unroll
            for (int ot = 0; ot < 5; ++ot) {
                const f32x4 c = fresh ? f32x4{0.f, 0.f, 0.f, 0.f} : a[ot];
                a[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[h][j][cp][ot], Bf[sp][h][cp], c, 0, 0, 0);
            }
        };
        // ---- exchange of the four waves' K-quarter partials of ONE output position.  Wave q owns
        // output tile q (a float4 per lane) and component q of every float4 of tile 4, so all four
        // waves do identical, branch-free work.
        auto part_write = [&](auto pb_tag, const f32x4 (&a)[5]) {
            constexpr int PB = decltype(pb_tag)::value;
            float* pw = part + PB * kPartFloats;
#pragma unroll
            for (int ot = 0; ot < 5; ++ot) *reinterpret_cast<f32x4*>(pw + ((q * 5 + ot) * 64 + lane) * 4) = a[ot];
        };
        auto red_load = [&](auto pb_tag) {
            constexpr int PB = decltype(pb_tag)::value;
            __syncthreads();     // s_waitcnt lgkmcnt(0) + s_barrier: every wave's partial(w) is in LDS
            const float* pw = part + PB * kPartFloats;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rp[k] = *reinterpret_cast<const f32x4*>(pw + ((k * 5 + q) * 64 + lane) * 4);
                rc[k] = pw[((k * 5 + 4) * 64 + lane) * 4 + q];
            }
        };
        auto red_finish = [&](int w) {
            const f32x4 s = (rp[0] + rp[1]) + (rp[2] + rp[3]);
            u32x2 o;
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bq_lds);
            o[0] = pack2relu(s[0] + bq[0], s[1] + bq[1]);
            o[1] = pack2relu(s[2] + bq[2], s[3] + bq[3]);
            const float t = ((rc[0] + rc[1]) + (rc[2] + rc[3])) + *b4_lds;
            const unsigned short t16 = (unsigned short)pack2relu(t, 0.f);
            unsigned short* frow = fbase - 4 * g;      // (fbase carries the lane's 4-channel offset of the old [w][o] rows)
            *reinterpret_cast<u32x2*>(frow + feat16_index(w, 16 * q + 4 * g)) = o;
            frow[feat16_index(w, 64 + 4 * g + q)] = t16;
        };
#define MDC_SB() __builtin_amdgcn_sched_barrier(0)
        // One position step v (outputs: a0 = v+2 fresh, a1 = v+1, a2 = v completes).  TAP-MAJOR order:
        //   R1  tap 2 (20 MFMAs) -> a2 is complete; finish of output v-1 rides along
        //   R2  a2 -> LDS (its registers are not written again before R4's lgkmcnt(0) of the NEXT step:
        //       an MFMA that overwrites the source of an in-flight ds_write corrupts it, and hipcc does
        //       not model that hazard); conv1 operand reads for v+1
        //   R3  conv1(v+1) (8 MFMAs) + tap 1 (20 MFMAs)
        //   R4  barrier; owner's reads of partial(v)
        //   R5  tap 0 (20 MFMAs) with the ReLU/bf16 pack of conv1(v+1) trailing one group behind
        // sched_barrier pins the phases: at the register limit hipcc otherwise sinks every LDS read next to
        // its use and exposes the LDS latency several times per step.
        auto step = [&](int v, auto par_next, auto first_tag, auto last_tag, f32x4 (&a0)[5], f32x4 (&a1)[5], f32x4 (&a2)[5]) {
            constexpr bool FIRST = decltype(first_tag)::value != 0, LAST = decltype(last_tag)::value != 0;
            using PB = std::integral_constant<int, 1 - decltype(par_next)::value>;      // = v & 1
            // A: tap 2 (20 MFMAs) + pack of row 1 of this step's conv1 (issued at the end of the previous step)
            if (!FIRST) { x_fence(I1{}); pack(PB{}, I1{}, I0{}); pack(PB{}, I1{}, I1{}); }
            if (!LAST) conv1_load(v + 1, par_next, I0{});
            tap(PB{}, I2{}, I0{}, I0{}, a2, false); tap(PB{}, I2{}, I0{}, I1{}, a2, false);
            tap(PB{}, I2{}, I1{}, I0{}, a2, false); tap(PB{}, I2{}, I1{}, I1{}, a2, false);
            // B: the completed output goes to LDS while tap 1 (20 MFMAs, writes a1 only) runs; conv1(v+1) row 0
            //    and its pack (so row 0 of X is live inside this phase only)
            part_write(PB{}, a2);
            if (!LAST) { conv1(par_next, I0{}); conv1_load(v + 1, par_next, I1{}); }
            tap(PB{}, I1{}, I0{}, I0{}, a1, false); tap(PB{}, I1{}, I0{}, I1{}, a1, false);
            tap(PB{}, I1{}, I1{}, I0{}, a1, false);
            if (!LAST) { x_fence(I0{}); pack(par_next, I0{}, I0{}); pack(par_next, I0{}, I1{}); }
            tap(PB{}, I1{}, I1{}, I1{}, a1, false);
            MDC_SB();
            // C: finish of output v-1; barrier + owner's reads of partial(v); tap 0 (20 MFMAs, fresh a0);
            //    conv1(v+1) row 1
            if (!FIRST) red_finish(v - 1);
            red_load(PB{});
            // HARDWARE HAZARD: a ds_write reads its data registers some time AFTER it issues (longer when the four
            // waves' write bursts queue up), and an MFMA that overwrites them meanwhile corrupts the stored partial
            // (VALU writes are interlocked, XDL writes are not; hipcc does not model it).  Keeping a2 alive until
            // here -- behind the barrier's s_waitcnt lgkmcnt(0) -- stops the register allocator from handing
            // a2's registers to any MFMA before the ds_writes have completed.
            asm volatile("" ::"a"(a2[0]), "a"(a2[1]), "a"(a2[2]), "a"(a2[3]), "a"(a2[4]));
            tap(PB{}, I0{}, I0{}, I0{}, a0, true);  tap(PB{}, I0{}, I0{}, I1{}, a0, false);
            tap(PB{}, I0{}, I1{}, I0{}, a0, false);
            if (!LAST) conv1(par_next, I1{});
            tap(PB{}, I0{}, I1{}, I1{}, a0, false);
            MDC_SB();
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;

        conv1_load(0, P0{}, I0{}); conv1_load(0, P0{}, I1{});
        conv1(P0{}, I0{}); conv1(P0{}, I1{});
        x_fence(I0{}); x_fence(I1{});
        pack(P0{}, I0{}, I0{}); pack(P0{}, I0{}, I1{}); pack(P0{}, I1{}, I0{}); pack(P0{}, I1{}, I1{});
        step(0, P1{}, I1{}, I0{}, acc[2], acc[1], acc[0]);
        int v = 1;
        for (int it = 0; it < 21; ++it, v += 6) {     // v = 1 .. 126
            if (it >= 12 && it < 16 && gnext < ngroups)      // next group's frames -> the other image buffer
                stage_quarter(it - 12, x, n, gnext * 16, img + (buf ^ 1) * kImgWords, tid);
            step(v + 0, P0{}, I0{}, I0{}, acc[0], acc[2], acc[1]);
            step(v + 1, P1{}, I0{}, I0{}, acc[1], acc[0], acc[2]);
            step(v + 2, P0{}, I0{}, I0{}, acc[2], acc[1], acc[0]);
            step(v + 3, P1{}, I0{}, I0{}, acc[0], acc[2], acc[1]);
            step(v + 4, P0{}, I0{}, I0{}, acc[1], acc[0], acc[2]);
            step(v + 5, P1{}, I0{}, I0{}, acc[2], acc[1], acc[0]);
        }
        step(127, P0{}, I0{}, I0{}, acc[0], acc[2], acc[1]);
        step(128, P1{}, I0{}, I0{}, acc[1], acc[0], acc[2]);
        step(129, P0{}, I0{}, I1{}, acc[2], acc[1], acc[0]);
        red_finish(129);
        // outputs 130 and 131 see only zero padding beyond position 131: complete as they are
        part_write(I0{}, acc[1]); red_load(I0{});
        asm volatile("" ::"a"(acc[1][0]), "a"(acc[1][1]), "a"(acc[1][2]), "a"(acc[1][3]), "a"(acc[1][4]));
        red_finish(130);
        part_write(I1{}, acc[2]); red_load(I1{});
        asm volatile("" ::"a"(acc[2][0]), "a"(acc[2][1]), "a"(acc[2][2]), "a"(acc[2][3]), "a"(acc[2][4]));
        red_finish(131);
#undef MDC_SB
        __syncthreads();      // next group's image is complete; partial buffers are free again
    }
}


#endif  // MDC_ALTERNATES

}  // namespace

int vtcnn2_bf16_pack(mdc_model* m) {
    const float* k1 = m->hk[0].data();   // (256,1,1,3)
    const float* b1 = m->hb[0].data();
    const float* k2 = m->hk[1].data();   // (80,256,2,3)
    int rc;
#ifdef MDC_ALTERNATES      // operands of the hipcc-scheduled conv kernel (d_pack slots 0 and 1): test build only
    // conv2 A-fragments: [q][((h*3+j)*2+cp)*5+ot][lane][8]; lane (o' = lane&15, g = lane>>4) slot jj holds
    // K2[16ot+o'][64q + 32cp + (jj<4 ? 4g+jj : 16+4g+jj-4)][h][j]   (the channel order X arrives in)
    std::vector<unsigned short> wq((size_t)4 * kWFrags * 64 * 8);
    for (int q = 0; q < 4; ++q)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 3; ++j)
                for (int cp = 0; cp < 2; ++cp)
                    for (int ot = 0; ot < 5; ++ot)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int o = 16 * ot + (lane & 15), g = lane >> 4;
                                const int ch = 64 * q + 32 * cp + (jj < 4 ? 4 * g + jj : 16 + 4 * g + (jj - 4));
                                const size_t idx = ((((size_t)q * kWFrags + ((h * 3 + j) * 2 + cp) * 5 + ot) * 64) + lane) * 8 + jj;
                                wq[idx] = f2bf(k2[(((size_t)o * kC1 + ch) * 2 + h) * 3 + j]);
                            }
    if ((rc = upload(m, 0, wq.data(), wq.size() * 2))) return rc;
    // conv1 A-fragments (K = 16): [q][ct][lane][4]; lane (c = lane&15, kg = lane>>4), taps at slots 0..2:
    //   kg 0: tap hi (x hi)   kg 1: tap hi (x lo)   kg 2: tap lo (x hi)   kg 3: (b1 hi, b1 lo, 0, 0) (x = 1)
    std::vector<unsigned short> a1((size_t)4 * 4 * 64 * 4, 0);
    for (int q = 0; q < 4; ++q)
        for (int ct = 0; ct < 4; ++ct)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = 64 * q + 16 * ct + (lane & 15), kg = lane >> 4;
                unsigned short* d = &a1[(((size_t)q * 4 + ct) * 64 + lane) * 4];
                const float sc = std::ldexp(1.f, -kFeatShift);      // as the production kernel's operands: activations x 2^-kFeatShift
                if (kg == 3) {
                    const unsigned short hi = f2bf(b1[ch] * sc);
                    d[0] = hi;
                    d[1] = f2bf(b1[ch] * sc - bf2f(hi));
                } else {
                    for (int t = 0; t < 3; ++t) {
                        const float kv = k1[ch * 3 + t] * sc;
                        const unsigned short hi = f2bf(kv);
                        d[t] = (kg == 2) ? f2bf(kv - bf2f(hi)) : hi;
                    }
                }
            }
    if ((rc = upload(m, 1, a1.data(), a1.size() * 2))) return rc;
    (void)k1; (void)b1; (void)k2;
#else
    (void)k1; (void)b1; (void)k2;
#endif
    // The bf16 mode's activations and features carry 2^-kFeatShift (vtcnn2_sched_common.h: ReLU in the bf16 conversion's
    // clamp bit): conv2's bias goes with them, dense1's weights take the factor back -- powers of two, exact everywhere
    {
        std::vector<float> b2s(m->hb[1]);
        for (float& v : b2s) v = std::ldexp(v, -kFeatShift);
        if ((rc = upload(m, 2, b2s.data(), b2s.size() * sizeof(float)))) return rc;
    }
    m->feat_scale_log2 = -kFeatShift;
    // dense1: transposed and K-tiled [k'/64][n][k'%64] with k' = w*80 + o  <-  reference row o*132 + w
    const float* w1 = m->hk[2].data();
    std::vector<unsigned short> w1t((size_t)kHid * kFeat);
    for (int w = 0; w < kW2; ++w)
        for (int o = 0; o < kC2; ++o) {
            const float* src = w1 + (size_t)(o * kW2 + w) * kHid;
            // tile-contiguous: [k-tile of 64][hidden unit][64], so a K-tile of the GEMM's B operand is one 32 KiB block
            const int kk = feat16_index(w, o);      // the K order the conv kernels store their features in
            for (int nn = 0; nn < kHid; ++nn) w1t[((size_t)(kk >> 6) * kHid + nn) * 64 + (kk & 63)] = f2bf(std::ldexp(src[nn], kFeatShift));
        }
    if ((rc = upload(m, 3, w1t.data(), w1t.size() * 2))) return rc;
    return vtcnn2_bf16_pack_sched(m);      // operands of the asm-sequenced conv kernel (its own K order)
}

int vtcnn2_bf16_conv(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s, long hop2, float scale) {
    // the asm-sequenced kernel (vtcnn2_bf16_sched.hip).  Alternates build, model created under MDC_CONV_SCHED=0: the
    // hipcc-scheduled one (same results up to summation order), f32 frames only -- raw bytes always take the production kernel
#ifdef MDC_ALTERNATES
    if ((m->alt & kAltConvHipcc) && hop2 <= 0) {
        const long ngroups = (n + 15) / 16;
        const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvBf16Lds));
        hipLaunchKernelGGL(vt_conv_bf16_kernel, dim3(grid), dim3(256), kConvBf16Lds, s, x, (long)n,
                           static_cast<const u32x4*>(m->d_pack[0]), static_cast<const u32x2*>(m->d_pack[1]),
                           static_cast<const float*>(m->d_pack[2]), static_cast<unsigned short*>(feat));
        MDC_HIP(hipGetLastError());
        return MDC_OK;
    }
#endif
    return vtcnn2_bf16_conv_sched(m, x, n, feat, s, hop2, scale);
}

}  // namespace mdc
