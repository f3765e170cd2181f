// Deployed single-conv nets (T1: F=3, T2: F=10), one fused HBM-bound pass.
//
// Math restated from CNN.ipynb cell 6 / the model_config inside the bundled .h5
// (SURVEY.md 8(a) A1):   y[h,w,f] = relu(b[f] + K0[f]*x[h,w-1] + K1[f]*x[h,w]),  w = 0..128,
// x[h,-1] = x[h,128] = 0;  z[c] = relu(bd[c] + sum_{h,w,f} Wd[h*129F + w*F + f][c] * y[h,w,f]);
// p = softmax(z);  label = first argmax (cnn.py:209).
//
// Mapping (gfx950, wave64): one wave owns 64 consecutive frames.  A frame is 1 KiB =
// 64 lanes x float4, so each frame is ONE fully coalesced global_load_dwordx4 per lane;
// lane l holds samples 4l'..4l'+3 of row h = l>>5 (l' = l&31).  The lane computes conv
// positions w = 4l'+1 .. 4l'+4 (x[w-1] is its own sample, x[w] its next sample; the one it
// lacks comes from lane l+1 by a DPP wave_shl:1) plus, in a 5th slot, w = 0 (only lanes
// with l' = 0 carry non-zero dense weights for it).  The dense weights of a lane's own
// positions (5*F*3 floats) live in its VGPRs for the whole kernel; conv taps are scalar.
// Per frame: 3 partial sums per lane -> DPP wave reduction -> deposited in lane f of
// the wave's result registers; after 64 frames each lane finishes ONE frame (bias, ReLU,
// softmax, argmax) and the wave writes 768 B of probabilities + 256 B of labels coalesced.
// HBM traffic = the algorithmic 1,036 B/frame (1,024 in + 12 out) + 4 B label.
#include "mdc_internal.h"

#include <cstdlib>
#include <type_traits>

namespace mdc {

namespace {

constexpr int kC = 3;
constexpr int kSlots = 5;
constexpr int kHeadFloats = 64;   // conv taps + biases, padded

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // v + (v moved by DPP pattern CTRL; lanes with no source or masked rows contribute 0)
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes; the total is valid in lane 63 (returned as a wave-uniform value).
__device__ __forceinline__ float wave_total(float v) {
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror      -> every lane holds its row's sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Simple one-frame-at-a-time variant: used only when a conv/flat tap is requested (slow path).
template <int F, int TAP>   // TAP: 0 none, 1 conv/flat (model4/model3), 2 dense (model2)
__global__ __launch_bounds__(256) void deployed_tap_kernel(const float* __restrict__ x, long n,
                                                           const float* __restrict__ wp,
                                                           float* __restrict__ probs, int* __restrict__ labels,
                                                           float* __restrict__ tap_conv, float* __restrict__ tap_dense) {
    const int lane = threadIdx.x & 63;
    const int lp = lane & 31;
    const int h = lane >> 5;

    float k0[F], k1[F], cb[F], bd[kC];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        k0[f] = wp[3 * f + 0];
        k1[f] = wp[3 * f + 1];
        cb[f] = wp[3 * f + 2];
    }
#pragma unroll
    for (int c = 0; c < kC; ++c) bd[c] = wp[3 * F + c];

    float wd[kSlots][F][kC];
    {
        const float* wl = wp + kHeadFloats + lane;
#pragma unroll
        for (int s = 0; s < kSlots; ++s)
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int c = 0; c < kC; ++c) wd[s][f][c] = wl[((s * F + f) * kC + c) * 64];
    }

    const long nwaves = (long)gridDim.x * 4;
    const long nblk = (n + 63) >> 6;
    for (long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += nwaves) {
        const long base = blk << 6;
        const int cnt = (int)((n - base) < 64 ? (n - base) : 64);
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        const float4* px = reinterpret_cast<const float4*>(x + base * kFrameFloats) + lane;
        float4 cur = px[0];
        for (int fr = 0; fr < cnt; ++fr) {
            float4 nx = cur;
            if (fr + 1 < cnt) nx = px[(long)(fr + 1) * 64];
            // sample x[4l'+4]: first sample of the next lane; right zero-pad at the row end
            float nxt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cur.x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
            nxt = (lp == 31) ? 0.f : nxt;
            const float xs[6] = {cur.x, cur.y, cur.z, cur.w, nxt, 0.f};
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                const float xm = (s < 4) ? xs[s] : 0.f;        // x[w-1]
                const float xc = (s < 4) ? xs[s + 1] : xs[0];  // x[w]   (slot 4: w = 0)
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    float y = fmaf(k1[f], xc, fmaf(k0[f], xm, cb[f]));
                    y = fmaxf(y, 0.f);
                    if (TAP == 1) {
                        const int w = (s < 4) ? (4 * lp + 1 + s) : 0;
                        if (s < 4 || lp == 0)
                            tap_conv[(((base + fr) * 2 + h) * 129 + w) * F + f] = y;
                    }
                    s0 = fmaf(wd[s][f][0], y, s0);
                    s1 = fmaf(wd[s][f][1], y, s1);
                    s2 = fmaf(wd[s][f][2], y, s2);
                }
            }
            const float t0 = wave_total(s0), t1 = wave_total(s1), t2 = wave_total(s2);
            if (lane == fr) { r0 = t0; r1 = t1; r2 = t2; }
            cur = nx;
        }
        if (lane < cnt) {
            const float z0 = fmaxf(r0 + bd[0], 0.f);   // Dense(3, activation='relu')
            const float z1 = fmaxf(r1 + bd[1], 0.f);
            const float z2 = fmaxf(r2 + bd[2], 0.f);
            const float mx = fmaxf(z0, fmaxf(z1, z2));
            const float e0 = expf(z0 - mx), e1 = expf(z1 - mx), e2 = expf(z2 - mx);
            const float inv = 1.0f / (e0 + e1 + e2);
            const long o = base + lane;
            const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
            if (probs) {
                probs[o * 3 + 0] = p0;
                probs[o * 3 + 1] = p1;
                probs[o * 3 + 2] = p2;
            }
            // int(np.argmax(test_Y_hat[i,:])) (cnn.py:209): FIRST maximum of the PROBABILITIES as returned --
            // two slightly different z can round to the same probability, and then the lower index wins
            if (labels) labels[o] = (p0 >= p1 && p0 >= p2) ? 0 : ((p1 >= p2) ? 1 : 2);
            if (TAP == 2) {
                tap_dense[o * 3 + 0] = z0;
                tap_dense[o * 3 + 1] = z1;
                tap_dense[o * 3 + 2] = z2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path.  Same lane mapping, but (a) frames are processed four at a time and their 12 partial
// sums are reduced with a bank-masked DPP reduce-scatter (each 4-lane bank ends up owning one frame:
// 12+12 adds for the two bank levels, 6 inside the bank, 2 permlane swaps per class across the four
// 16-lane rows) instead of a full 6-step wave reduction per value; (b) the 129th conv position
// (w = 0), which only one lane per row owns, is not a fifth slot any more: the position is evaluated once per 64-frame
// block for all frames at once, with scalar weights (x[h][0] travels through a 512-byte LDS table).
// Result lane of frame j = 4G + f of the block:  16*(G>>2) + 4*bank(f) + (G&3),  bank = {0,2,1,3}[f].
// ---------------------------------------------------------------------------------------------
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_add_banks(float v) {
    // v + (v moved by CTRL) in the banks selected by BANK_MASK; other banks unchanged
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, BANK_MASK, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float rows_sum(float m) {       // sum over the four 16-lane rows, result in all rows
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// TAIL = true: the same arithmetic (bit-identical results, so predict() does not depend on how a batch is
// chunked) on ONE ragged block of n < 64 frames with guarded loads and stores.
// U8 = true: x points at raw interleaved unsigned 8-bit (I,Q) samples (256 B per frame, mdc_forward_iq_u8); a lane
// loads the 8 bytes that hold its four samples of BOTH rows (lanes l and l+32 read the same address) and converts
// the row it owns with the arithmetic of iq_u8_kernel (eval_ops.hip), so the results are bit-identical to
// mdc_iq_u8_to_frames followed by mdc_forward -- with 256 instead of 1,024 B of HBM input per frame.  Window f of the
// capture starts at byte f * hop2 (hop2 = 256: disjoint frames; smaller: overlapping windows of a live stream, whose
// bytes are then fetched from HBM once and re-read from cache); only 2-byte alignment of a window is assumed.
// ReLU rides in the clamp bit of the conv's second fma (round 3): the fast kernel's table carries the conv taps and bias
// times 2^-32 and the dense weights times 2^+32 -- exact powers of two: for conv activations in [2^-94, 2^32) every product
// and every f32 sum is the same bits as with fma, fma, v_max_f32 (outside: the scaled activation saturates at 1 = 2^32
// true, loses bits as an f32 denormal below 2^-94 true, and an Inf activation comes out finite; the MDC_TAP_CONV / FLAT
// kernel reads the UNSCALED table with fmaxf, so there tap and probabilities can disagree: include/mdc.h) -- so a conv
// output is below 1 unless the true value exceeds 2^32, and
// __builtin_amdgcn_fmed3f(fma(...), 0, 1) folds into `v_fma_f32 ... clamp` (hipcc folds it into the plain fma only: the
// 10-filter net's packed path keeps its packed first fma and takes two clamped plain fmas per pair of positions).  One VALU
// fewer per conv output: T1 -4.3 % (0.230 -> 0.220 ms per 2^20 frames), T2 the speed of round 2's re-associated "pivot" form
// (fma + v_med3, sign and scale folded into the dense weights; 0.41 ms) in KERAS' operation order -- the pivot form, its
// weight conditions and its option bit are gone (profiles/r03_dep_f32_clamp_ab.log).
// RING > 0 (round 3; f32 frames, full blocks): the frames reach the lanes through a per-wave LDS ring of RING groups
// (4 frames = 4 KiB each) filled by asm-issued LDS-DMA -- one global_load_lds_dwordx4 = one whole frame, 1 KiB contiguous,
// landing lane-linear, read back by the same lane with one ds_read_b128: the registers hold exactly what the direct
// load would have put there, the arithmetic below is untouched (T1 stays Keras' order; results bit-identical to RING = 0).
// The wave's blocks form ONE stream of groups: at group g the wave (a) waits lgkmcnt(0) -- the reads of group g, issued a
// whole group of arithmetic earlier, are out of their slot --, (b) issues the copies of group g + RING into that slot,
// (c) waits vmcnt(4 (RING - 1)): group g + 1 has landed (vmcnt retires in order; stores of a finished block sit in the
// same queue and only make the wait stricter), (d) reads group g + 1 into registers, (e) computes group g.  RING - 1
// groups stay in flight per wave across group AND block boundaries (the direct-load form drains at every block start);
// groups past the wave's last one are clamped copies of it, so the count of outstanding copies stays uniform, and the
// wave drains with vmcnt(0) before it ends (a copy must not land in LDS that already belongs to another work-group).
// Round 2 tried this ring with the builtin and saw no gain: hipcc put a vmcnt(0) in front of every LDS read (HISTORY.md
// 4.1b, "the LDS-DMA rings were not rings").
template <int F, int TAP, int ABL = 0, bool TAIL = false, bool U8 = false, int RING = 0>   // TAP: 0 none, 2 dense (model2); ABL: timing-only ablations
__global__ __launch_bounds__(256) void deployed_fwd_kernel(const float* __restrict__ x, long n,
                                                           const float* __restrict__ wp,
                                                           float* __restrict__ probs, int* __restrict__ labels,
                                                           float* __restrict__ tap_dense, float scale = 0.f, long hop2 = 256) {
    const int lane = threadIdx.x & 63;
    const int lp = lane & 31;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    constexpr bool kPacked = F >= 8 && !(ABL & 8) && !(ABL & 16);

    float k0[F], k1[F], cb[F], bd[kC];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        k0[f] = wp[3 * f + 0];
        k1[f] = wp[3 * f + 1];
        cb[f] = wp[3 * f + 2];
    }
#pragma unroll
    for (int c = 0; c < kC; ++c) bd[c] = wp[3 * F + c];
    float wd[4][F][kC];      // dense weights of this lane's own four positions
    {
        const float* wl = wp + kHeadFloats + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int c = 0; c < kC; ++c) wd[s][f][c] = wl[((s * F + f) * kC + c) * 64];
    }
    // dense weights of position w = 0 of row 0 / row 1 (wave-uniform): slot 4 of lanes 0 and 32
    const float* we0 = wp + kHeadFloats + (4 * F * kC) * 64;
    __shared__ float e_lds[4][128];
    float* e_tab = e_lds[(threadIdx.x >> 6) & 3];
    // lane -> block-relative frame it finishes
    const int myG = 4 * (lane >> 4) + (lane & 3);
    const int myb = (lane >> 2) & 3;
    const int myframe = 4 * myG + ((myb == 1) ? 2 : (myb == 2) ? 1 : myb);

    // FULL 64-frame blocks only (the host launches the one-frame-at-a-time kernel on a ragged tail), so
    // every load is unconditional: guarded loads made hipcc wait for the prefetch right after issuing it.
    const long nwaves = (long)gridDim.x * 4;
    const long nblk = TAIL ? 1 : (n >> 6);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(RING == 0 || (!TAIL && !U8 && RING >= 2), "the ring form serves full blocks of f32 frames");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_mem[];
    unsigned char* const ring = ring_mem + (RING ? wv * (RING * 4096) : 0);
    const long blk0 = (long)blockIdx.x * 4 + wv;
    const long my_groups = (RING && blk0 < nblk) ? ((nblk - blk0 + nwaves - 1) / nwaves) * 16 : 0;      // this wave's stream
    long gi = 0;                                                                                         // its current group
    auto ring_issue = [&](long g) {      // the four frames of group g (clamped to the stream's last group) -> slot g % RING
        const long gc = g < my_groups ? g : my_groups - 1;
        const float* src = x + (((blk0 + (gc >> 4) * nwaves) << 6) + 4 * (gc & 15)) * (long)kFrameFloats + 4 * lane;
        unsigned char* slot = ring + (g % RING) * 4096;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(slot + f * 1024);
            asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(src + f * kFrameFloats), "{m0}"(l) : "memory");
        }
    };
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    auto ring_read = [&](long g, f32x4 (&dst)[4]) {
        const unsigned char* slot = ring + (g % RING) * 4096 + lane * 16;
#pragma unroll
        for (int f = 0; f < 4; ++f) dst[f] = *reinterpret_cast<const f32x4*>(slot + f * 1024);
    };
    f32x4 ring_cur[4], ring_nx[4];
    if constexpr (RING > 0) {
        if (my_groups == 0) return;      // (a wave without a block has issued nothing)
        for (int g = 0; g < RING; ++g) ring_issue(g);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING - 1)) : "memory");
        ring_read(0, ring_cur);
    }
    using Raw = typename std::conditional<U8, uint2, float4>::type;
    Raw cur_raw[4], nx[4];
    bool first_block = true;      // direct-load form: only a wave's first block loads its first group synchronously
    for (long blk = blk0; blk < nblk; blk += nwaves) {
        const long base = blk << 6;
        float r[kC] = {0.f, 0.f, 0.f};
        // x[0][0] / x[1][0] of every frame go through a 512-byte LDS table (written by lanes 0 and 32 as the
        // frames stream by, read once per block by the lane that finishes the frame): no extra HBM/L2 reads
        float eI = 0.f, eQ = 0.f;
        const float4* px = reinterpret_cast<const float4*>(x + base * kFrameFloats) + lane;
        const unsigned char* pb = reinterpret_cast<const unsigned char*>(x) + base * hop2 + lp * 8;
        float4 cur[4];
        auto load = [&](long j) -> Raw {
            if constexpr (U8) {
                // past the end of a ragged block: bytes whose conversion is not used (those frames are never stored)
                if (TAIL && j >= n) return make_uint2(0u, 0u);
                return load8_unaligned(pb + (long)j * hop2);
            } else {
                if (ABL & 32) return (reinterpret_cast<const float4*>(x) + lane)[(long)(j & 15) * 64];      // timing probe: the batch's first 16 frames over and over (cache-resident)
                if (!TAIL) return px[(long)j * 64];
                return (j < n) ? px[(long)j * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        const unsigned row_shift = (lane >> 5) * 8;      // row 0 = I = even bytes, row 1 = Q = odd bytes
        auto decode = [&](const Raw& r) -> float4 {
            if constexpr (U8) {
                const unsigned a = r.x >> row_shift, b = r.y >> row_shift;
                return make_float4(((float)(a & 0xFFu) - 127.5f) * scale, ((float)((a >> 16) & 0xFFu) - 127.5f) * scale,
                                   ((float)(b & 0xFFu) - 127.5f) * scale, ((float)((b >> 16) & 0xFFu) - 127.5f) * scale);
            } else {
                return r;
            }
        };
        if constexpr (RING == 0) {
            // (later blocks: their first group was fetched during the last group of the block before -- round 3; the
            // stream used to drain at every block start, a full memory latency per 64 frames)
            if (first_block) {
#pragma unroll
                for (int f = 0; f < 4; ++f) cur_raw[f] = load(f);
            }
            first_block = false;
        }
        const bool has_next = !TAIL && blk + nwaves < nblk;
        // a ragged block stops after its last group with a real frame: a single window (n = 1, the reference's
        // deployment) then costs one group, not sixteen; a frame's arithmetic does not depend on what follows it
        const int ngrp = TAIL ? (int)((n + 3) >> 2) : 16;
        for (int G = 0; G < ngrp; ++G) {
            // prefetch the next group -- after a block's last group the first group of the wave's NEXT block (frame
            // indices are linear in the block-relative j); the wave's very last group re-reads itself: harmless, stays
            // in bounds.  One group ahead is the measured optimum; two groups ahead (12 KB per wave in flight) was 10 % slower.
            const long Gn4 = (G < 15) ? 4 * (G + 1) : (has_next ? 64 * nwaves : 60);
            if constexpr (RING > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ring_cur[0]), "+v"(ring_cur[1]), "+v"(ring_cur[2]), "+v"(ring_cur[3]) :: "memory");
                ring_issue(gi + RING);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING - 1)) : "memory");
                ring_read(gi + 1, ring_nx);
                __builtin_amdgcn_sched_barrier(0);     // copies, wait and reads stay at the top of the group
#pragma unroll
                for (int f = 0; f < 4; ++f) cur[f] = make_float4(ring_cur[f].x, ring_cur[f].y, ring_cur[f].z, ring_cur[f].w);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) nx[f] = load(Gn4 + f);
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch at the top of the group
#pragma unroll
                for (int f = 0; f < 4; ++f) cur[f] = decode(cur_raw[f]);
            }
            float v[4][kC];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                // sample x[4l'+4]: first sample of the next lane; right zero-pad at the row end
                float nxt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cur[f].x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
                nxt = (lp == 31) ? 0.f : nxt;
                const float xs[5] = {cur[f].x, cur[f].y, cur[f].z, cur[f].w, nxt};
                if (!(ABL & 1) && lp == 0) e_tab[(4 * G + f) * 2 + (lane >> 5)] = cur[f].x;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                if (ABL & 4) { s0 = xs[0] + xs[1]; s1 = xs[2] + xs[3]; s2 = xs[4]; }     // memory-only probe
                else if constexpr (kPacked) {
                    // F = 10 is VALU-bound: two positions per v_pk_fma_f32 (same issue rate as v_fma_f32,
                    // tools/microbench/valu_rate.hip) -- 140 instead of 240 VALU per frame for the conv + dense part
                    const f32x2 xm[2] = {f32x2{xs[0], xs[1]}, f32x2{xs[2], xs[3]}};
                    const f32x2 xc[2] = {f32x2{xs[1], xs[2]}, f32x2{xs[3], xs[4]}};
                    f32x2 a0 = f32x2{0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
                    for (int ff = 0; ff < F; ++ff)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            // one packed fma + two clamped fmas per pair of positions (ReLU in the clamp bit, see above)
                            const f32x2 t = __builtin_elementwise_fma(f32x2{k0[ff], k0[ff]}, xm[pr], f32x2{cb[ff], cb[ff]});
                            const f32x2 y = f32x2{__builtin_amdgcn_fmed3f(fmaf(k1[ff], xc[pr].x, t.x), 0.f, 1.f),
                                                  __builtin_amdgcn_fmed3f(fmaf(k1[ff], xc[pr].y, t.y), 0.f, 1.f)};
                            a0 = __builtin_elementwise_fma(f32x2{wd[2 * pr][ff][0], wd[2 * pr + 1][ff][0]}, y, a0);
                            a1 = __builtin_elementwise_fma(f32x2{wd[2 * pr][ff][1], wd[2 * pr + 1][ff][1]}, y, a1);
                            a2 = __builtin_elementwise_fma(f32x2{wd[2 * pr][ff][2], wd[2 * pr + 1][ff][2]}, y, a2);
                        }
                    s0 = a0.x + a0.y; s1 = a1.x + a1.y; s2 = a2.x + a2.y;
                } else if constexpr ((ABL & 16) != 0) {
                    // every instruction independent of the one before it: all filters' first fma, then all second fmas,
                    // then all ReLUs, then the dense fmas (three accumulator chains in rotation)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        float t[F];
#pragma unroll
                        for (int ff = 0; ff < F; ++ff) t[ff] = fmaf(k0[ff], xs[s], cb[ff]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ff = 0; ff < F; ++ff) t[ff] = fmaf(k1[ff], xs[s + 1], t[ff]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ff = 0; ff < F; ++ff) t[ff] = fmaxf(t[ff], 0.f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ff = 0; ff < F; ++ff) {
                            s0 = fmaf(wd[s][ff][0], t[ff], s0);
                            s1 = fmaf(wd[s][ff][1], t[ff], s1);
                            s2 = fmaf(wd[s][ff][2], t[ff], s2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ff = 0; ff < F; ++ff) {
                        const float y = __builtin_amdgcn_fmed3f(fmaf(k1[ff], xs[s + 1], fmaf(k0[ff], xs[s], cb[ff])), 0.f, 1.f);      // -> v_fma_f32 ... clamp
                        s0 = fmaf(wd[s][ff][0], y, s0);
                        s1 = fmaf(wd[s][ff][1], y, s1);
                        s2 = fmaf(wd[s][ff][2], y, s2);
                    }
                v[f][0] = s0; v[f][1] = s1; v[f][2] = s2;
            }
            // ---- reduce-scatter with fused, bank-masked DPP adds (one instruction each; lanes outside the
            // bank mask keep their value).  Level 1: banks {0,2} take frames 0,1 from lane+4, banks {1,3} take
            // frames 2,3 from lane-4.  Level 2 writes ONE register per class: bank 0 <- frame 0 (+ lane+8),
            // bank 2 <- frame 1 (+ lane-8), bank 1 <- frame 2 (+ lane+8), bank 3 <- frame 3 (+ lane-8).
            // All ops of a level are issued before the next level, so no DPP reads a register written by the
            // instruction just before it (hipcc does not pad asm with the 2 wait states that would need).
            float m[kC];
            if (ABL & 2) {
#pragma unroll
                for (int c = 0; c < kC; ++c) m[c] = (v[0][c] + v[1][c]) + (v[2][c] + v[3][c]);
            } else {
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(v[0][c]));
                    asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(v[1][c]));
                    asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(v[2][c]));
                    asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(v[3][c]));
                }
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    m[c] = 0.f;
                    asm volatile("v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x1" : "+v"(m[c]) : "v"(v[0][c]));
                    asm volatile("v_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0x4" : "+v"(m[c]) : "v"(v[1][c]));
                    asm volatile("v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x2" : "+v"(m[c]) : "v"(v[2][c]));
                    asm volatile("v_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0x8" : "+v"(m[c]) : "v"(v[3][c]));
                }
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    float t = m[c];
                    t = dpp_add<0xB1, 0xf>(t);                           // inside the bank: quad_perm [1,0,3,2]
                    t = dpp_add<0x4E, 0xf>(t);                           //                  quad_perm [2,3,0,1]
                    m[c] = rows_sum(t);
                }
            }
            if (myG == G) { r[0] = m[0]; r[1] = m[1]; r[2] = m[2]; }
            if constexpr (RING > 0) {
#pragma unroll
                for (int f = 0; f < 4; ++f) ring_cur[f] = ring_nx[f];
                ++gi;
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) cur_raw[f] = nx[f];
            }
        }
        // ---- position w = 0 of both rows, all 64 frames at once: y = relu(b + K1*x[h][0]) (x[h][-1] = 0)
        if (!(ABL & 1)) { eI = e_tab[myframe * 2 + 0]; eQ = e_tab[myframe * 2 + 1]; }
#pragma unroll
        for (int ff = 0; ff < F; ++ff) {
            const float yI = __builtin_amdgcn_fmed3f(fmaf(k1[ff], eI, cb[ff]), 0.f, 1.f);      // x[h][-1] = 0
            const float yQ = __builtin_amdgcn_fmed3f(fmaf(k1[ff], eQ, cb[ff]), 0.f, 1.f);
#pragma unroll
            for (int c = 0; c < kC; ++c) {
                r[c] = fmaf(we0[(ff * kC + c) * 64 + 0], yI, r[c]);
                r[c] = fmaf(we0[(ff * kC + c) * 64 + 32], yQ, r[c]);
            }
        }
        const long o = base + myframe;
        if (!TAIL || o < n) {
            const float z0 = fmaxf(r[0] + bd[0], 0.f);   // Dense(3, activation='relu')
            const float z1 = fmaxf(r[1] + bd[1], 0.f);
            const float z2 = fmaxf(r[2] + bd[2], 0.f);
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
    }
    if constexpr (RING > 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the clamped tail copies have landed
}

}  // namespace

static bool f32_mfma_variant(const mdc_model* m) {
#ifdef MDC_ALTERNATES
    return (m->alt & kAltDepF32Mfma) != 0;
#else
    (void)m;
    return false;
#endif
}

// Pack: [F x (k0,k1,b)] [bd x3] pad to 64 floats, then per-lane dense weights
// wl[((slot*F + f)*3 + c)*64 + lane].
int deployed_pack(mdc_model* m) {
    const int F = m->topo.filters;
    std::vector<float> pk(kHeadFloats + (size_t)kSlots * F * kC * 64, 0.f);
    const float* ck = m->hk[0].data();   // HWIO (1,2,1,F): [kw][f]
    for (int f = 0; f < F; ++f) {
        pk[3 * f + 0] = ck[0 * F + f];
        pk[3 * f + 1] = ck[1 * F + f];
        pk[3 * f + 2] = m->hb[0][f];
    }
    for (int c = 0; c < kC; ++c) pk[3 * F + c] = m->hb[1][c];
    const float* dk = m->hk[1].data();   // (258F, 3), rows h*129F + w*F + f
    for (int lane = 0; lane < 64; ++lane) {
        const int lp = lane & 31, h = lane >> 5;
        for (int s = 0; s < kSlots; ++s) {
            int w;
            if (s < 4) w = 4 * lp + 1 + s;
            else if (lp == 0) w = 0;
            else continue;
            for (int f = 0; f < F; ++f)
                for (int c = 0; c < kC; ++c)
                    pk[kHeadFloats + ((size_t)(s * F + f) * kC + c) * 64 + lane] = dk[((size_t)h * 129 * F + (size_t)w * F + f) * kC + c];
        }
    }
    int rc = upload(m, 0, pk.data(), pk.size() * sizeof(float));
    if (rc != MDC_OK) return rc;
    {   // the fast kernel's table (slot 5): taps and conv bias x 2^-32, dense weights x 2^+32 -- exact; the ReLU then rides in
        // the fma's clamp bit (deployed_fwd_kernel).  Slot 0 stays unscaled: the tap kernel, the f16 mode's conversion
        // of the taps and the alternates' f32-MFMA variant read it.
        std::vector<float> ps(pk);
        for (int f = 0; f < 3 * F; ++f) ps[f] = std::ldexp(pk[f], -32);
        for (size_t i = kHeadFloats; i < ps.size(); ++i) ps[i] = std::ldexp(pk[i], 32);
        if ((rc = upload(m, 5, ps.data(), ps.size() * sizeof(float))) != MDC_OK) return rc;
    }
#ifdef MDC_ALTERNATES
    return deployed_f32m_pack(m);      // + the dense layer as f32 MFMA operands (deployed_f32m.hip)
#else
    return MDC_OK;
#endif
}

int deployed_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                     float* tap, int tap_kind, hipStream_t s) {
    if (tap_kind == MDC_TAP_HIDDEN) { set_error("deployed nets have no hidden dense layer to tap"); return MDC_EINVAL; }
    if (m->dtype != MDC_F32) {      // bf16 / f16 / fp8: the dense layer on the matrix cores (deployed_bf16.hip)
        if (tap_kind != MDC_TAP_NONE) { set_error("layer taps of the deployed nets are served by the f32 kernels (finalize with MDC_F32)"); return MDC_ENOTSUP; }
        return deployed_bf16_forward(m, x, n, probs, labels, s);
    }
    const float* wp0 = static_cast<const float*>(m->d_pack[0]);      // unscaled table: the one-frame-at-a-time tap kernel
    const float* wp = static_cast<const float*>(m->d_pack[5]);       // scaled table of the fast kernel (ReLU in the clamp bit)
    const int F = m->topo.filters;
    float* tap_conv = (tap_kind == MDC_TAP_CONV || tap_kind == MDC_TAP_FLAT) ? tap : nullptr;
    float* tap_dense = (tap_kind == MDC_TAP_DENSE) ? tap : nullptr;
    ProfScope ps(m, 0, s);
    // Alternates build only (MDC_DEP_F32_MFMA=1 when the model was created): the variant with Dense(3) on the f32 matrix
    // pipe (deployed_f32m.hip: same results to the last bits of the summation order, measured SLOWER -- v_mfma_f32_4x4x1
    // holds the SIMD's vector issue for its whole 8 cycles, HISTORY.md section 4.1c); the production f32 path is the
    // all-VALU kernel below.  Conv/flat taps always use the simple one-frame-at-a-time kernel.
#ifdef MDC_ALTERNATES
    if (!tap_conv && f32_mfma_variant(m)) return deployed_f32m_forward(m, x, n, probs, labels, tap_dense, s);
#endif
    const long nfull = tap_conv ? 0 : (n / 64) * 64;      // frames handled by the fast kernel
    if (nfull > 0) {
        long grid = (nfull / 64 + 3) / 4;
        if (grid > 2048) grid = 2048;
#ifdef MDC_ABLATIONS
        static const int abl = getenv("MDC_ABLATE_DEP") ? atoi(getenv("MDC_ABLATE_DEP")) : 0;
        if (abl == 1 && F == 3) { hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 1>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense); return MDC_OK; }
        if (abl == 2 && F == 3) { hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 2>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense); return MDC_OK; }
        if (abl == 3 && F == 3) { hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 3>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense); return MDC_OK; }
        if (abl == 7 && F == 3) { hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 7>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense); return MDC_OK; }
#endif
#ifdef MDC_ABLATIONS   // timing probes of the inner loop: 1 plain fmas in independent phases, 2 the same on cache-resident frames, 3 packed on cache-resident frames
        static const int phased = getenv("MDC_DEP_PHASED") ? atoi(getenv("MDC_DEP_PHASED")) : 0;
#define MDC_DEP_PROBE(P, A) if (phased == P && !tap_dense) { \
            if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, A>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense); \
            else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 0, A>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense); \
            MDC_HIP(hipGetLastError()); return MDC_OK; }
        MDC_DEP_PROBE(1, 16) MDC_DEP_PROBE(2, 48) MDC_DEP_PROBE(3, 32)
#undef MDC_DEP_PROBE
#endif
        // Alternates build, model created under MDC_DEP_RING=N (N in 2, 3, 4, 6, 8): T1's frames through the per-wave
        // LDS-DMA ring (see the kernel) instead of direct loads.  Bit-identical and SLOWER at every depth (round 3,
        // profiles/r03_t1_f32_ring_ab.log: 4.05 / 3.9 / 3.9 / 3.35 / 3.45e9 frames/s against 4.2-4.3e9 in the same run):
        // the ring's LDS limits a CU to 16 / 12 / 8 / 4 waves, and this all-VALU kernel needs its 32 to cover the DPP
        // reduce-scatter's latencies.  The grid is exactly the resident set, so every wave walks one long stream of groups.
        int ring = 0;
#ifdef MDC_ALTERNATES
        if (m->alt_ring > 0 && F == 3 && !tap_dense) ring = m->alt_ring;
#endif
        if (ring > 0) {
#ifdef MDC_ALTERNATES
#define MDC_LAUNCH_RING(R) do { \
            constexpr int lds = 4 * R * 4096, per_cu = (160 * 1024) / (lds + 2048); \
            long g = (nfull / 64 + 3) / 4; if (g > 256L * per_cu) g = 256L * per_cu; \
            MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deployed_fwd_kernel<3, 0, 0, false, false, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 0, false, false, R>), dim3((unsigned)g), dim3(256), lds, s, x, nfull, wp, probs, labels, tap_dense); } while (0)
            switch (ring) {
                case 2: MDC_LAUNCH_RING(2); break;
                case 3: MDC_LAUNCH_RING(3); break;
                case 6: MDC_LAUNCH_RING(6); break;
                case 8: MDC_LAUNCH_RING(8); break;
                default: MDC_LAUNCH_RING(4); break;
            }
#undef MDC_LAUNCH_RING
#endif
        } else if (tap_dense) {
            if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 2>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense);
            else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 2>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense);
        } else {
            if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 0>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense);
            else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 0>), dim3(grid), dim3(256), 0, s, x, nfull, wp, probs, labels, tap_dense);
        }
    }
    if (nfull < n) {        // ragged tail (< 64 frames) or a conv tap: the simple kernel
        const long nt = n - nfull;
        const float* xt = x + nfull * kFrameFloats;
        float* pt = probs ? probs + nfull * 3 : nullptr;
        int* lt = labels ? labels + nfull : nullptr;
        float* tc = tap_conv;                               // (tap_conv implies nfull == 0)
        float* td = tap_dense ? tap_dense + nfull * 3 : nullptr;
        if (tc) {       // conv/flat tap: the one-frame-at-a-time kernel over the whole batch
            const long g2 = ((nt + 63) / 64 + 3) / 4 > 2048 ? 2048 : ((nt + 63) / 64 + 3) / 4;
            if (F == 3) hipLaunchKernelGGL((deployed_tap_kernel<3, 1>), dim3((unsigned)g2), dim3(256), 0, s, xt, nt, wp0, pt, lt, tc, td);
            else        hipLaunchKernelGGL((deployed_tap_kernel<10, 1>), dim3((unsigned)g2), dim3(256), 0, s, xt, nt, wp0, pt, lt, tc, td);
        } else if (td) {
            if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 2, 0, true>), dim3(1), dim3(64), 0, s, xt, nt, wp, pt, lt, td);
            else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 2, 0, true>), dim3(1), dim3(64), 0, s, xt, nt, wp, pt, lt, td);
        } else {
            if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 0, true>), dim3(1), dim3(64), 0, s, xt, nt, wp, pt, lt, td);
            else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 0, 0, true>), dim3(1), dim3(64), 0, s, xt, nt, wp, pt, lt, td);
        }
    }
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

// Raw SDR bytes straight into the deployed nets: full 64-frame blocks by the fast kernel,
// a ragged tail by its TAIL form; no frame buffer in between.
int deployed_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, int64_t hop, float scale, float* probs, int32_t* labels, hipStream_t s) {
    const float* wp = static_cast<const float*>(m->d_pack[5]);       // scaled table of the fast kernel
    const int F = m->topo.filters;
    const long nfull = (n / 64) * 64;
    const long hop2 = 2 * (long)hop;
    const float* xb = reinterpret_cast<const float*>(iq);
    ProfScope ps(m, 0, s);
#ifdef MDC_ALTERNATES
    if (f32_mfma_variant(m)) return deployed_f32m_forward_iq_u8(m, iq, n, hop2, scale, probs, labels, s);
#endif
    if (nfull > 0) {
        long grid = (nfull / 64 + 3) / 4;
        if (grid > 2048) grid = 2048;
        if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 0, false, true>), dim3(grid), dim3(256), 0, s, xb, nfull, wp, probs, labels, nullptr, scale, hop2);
        else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 0, 0, false, true>), dim3(grid), dim3(256), 0, s, xb, nfull, wp, probs, labels, nullptr, scale, hop2);
    }
    if (nfull < n) {
        const long nt = n - nfull;
        const float* xt = reinterpret_cast<const float*>(iq + nfull * hop2);
        float* pt = probs ? probs + nfull * 3 : nullptr;
        int* lt = labels ? labels + nfull : nullptr;
        if (F == 3) hipLaunchKernelGGL((deployed_fwd_kernel<3, 0, 0, true, true>), dim3(1), dim3(64), 0, s, xt, nt, wp, pt, lt, nullptr, scale, hop2);
        else        hipLaunchKernelGGL((deployed_fwd_kernel<10, 0, 0, true, true>), dim3(1), dim3(64), 0, s, xt, nt, wp, pt, lt, nullptr, scale, hop2);
    }
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
