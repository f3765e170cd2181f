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

template <int F, int TAP>   // TAP: 0 none, 1 conv/flat (model4/model3), 2 dense (model2)
__global__ __launch_bounds__(256) void deployed_fwd_kernel(const float* __restrict__ x, long n,
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
            if (probs) {
                probs[o * 3 + 0] = e0 * inv;
                probs[o * 3 + 1] = e1 * inv;
                probs[o * 3 + 2] = e2 * inv;
            }
            // np.argmax: first maximum.  exp and the common scale are monotone, so the
            // argmax of the probabilities is the argmax of z (ties included: equal z give
            // bit-equal e).
            if (labels) labels[o] = (z0 >= z1 && z0 >= z2) ? 0 : ((z1 >= z2) ? 1 : 2);
            if (TAP == 2) {
                tap_dense[o * 3 + 0] = z0;
                tap_dense[o * 3 + 1] = z1;
                tap_dense[o * 3 + 2] = z2;
            }
        }
    }
}

}  // namespace

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
    return upload(m, 0, pk.data(), pk.size() * sizeof(float));
}

int deployed_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                     float* tap, int tap_kind, hipStream_t s) {
    if (tap_kind == MDC_TAP_HIDDEN) { set_error("deployed nets have no hidden dense layer to tap"); return MDC_EINVAL; }
    const float* wp = static_cast<const float*>(m->d_pack[0]);
    const long nblk = (n + 63) / 64;
    long grid = (nblk + 3) / 4;
    if (grid > 2048) grid = 2048;
    const int F = m->topo.filters;
    float* tap_conv = (tap_kind == MDC_TAP_CONV || tap_kind == MDC_TAP_FLAT) ? tap : nullptr;
    float* tap_dense = (tap_kind == MDC_TAP_DENSE) ? tap : nullptr;
    ProfScope ps(m, 0, s);
#define MDC_LAUNCH_DEPLOYED(FF, TT) \
    hipLaunchKernelGGL((deployed_fwd_kernel<FF, TT>), dim3(grid), dim3(256), 0, s, x, (long)n, wp, probs, labels, tap_conv, tap_dense)
    const int tt = tap_conv ? 1 : (tap_dense ? 2 : 0);
    if (F == 3) {
        if (tt == 0) MDC_LAUNCH_DEPLOYED(3, 0); else if (tt == 1) MDC_LAUNCH_DEPLOYED(3, 1); else MDC_LAUNCH_DEPLOYED(3, 2);
    } else {
        if (tt == 0) MDC_LAUNCH_DEPLOYED(10, 0); else if (tt == 1) MDC_LAUNCH_DEPLOYED(10, 1); else MDC_LAUNCH_DEPLOYED(10, 2);
    }
#undef MDC_LAUNCH_DEPLOYED
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
