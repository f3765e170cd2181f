// Training step of the two nets the reference trains (/root/reference citations):
//     model.compile(loss='categorical_crossentropy', optimizer='adam')                       cnn.py:113, CNN.ipynb cell 6
//     model.fit(X_train, Y_train, batch_size=1024, epochs=100, validation_data=..., callbacks=[ModelCheckpoint,
//               EarlyStopping])                                                              cnn.py:122-147, CNN.ipynb cell 8
// for MDC_KIND_DEPLOYED (CNN.ipynb cell 6: T1 F=3 / T2 F=10) and MDC_KIND_CNNPY (cnn.py:104-112: T4), in f32.  Neither net
// has a Dropout layer (`dr` is never used in them), so the training forward IS the inference forward -- unless the caller
// asks for one (mdc_trainer_set_dropout; off by default): inverted Dropout(rate) behind the conv activations (and behind
// cnn.py's Dense(D)), where VT-CNN2 has it, its mask a pure function of (seed, Adam's step count, the frame's index in the
// data set, the element) through murmur3's 32-bit finaliser -- no generator state, so the backward pass, a replayed
// hipGraph and the numpy oracle all see the same bits.  The canonical VT-CNN2 (T3) is trained only in the vendored DeepSig
// notebook: out of scope.
//
// One mini-batch = two launches on the caller's stream, no host synchronisation, nothing allocated:
//   mdc_train_grad   one wave per slice of the batch (frames g, g+G, ...): forward, Keras' categorical cross-entropy on
//                    the softmax rows (row / sum, clip to [1e-7, 1-1e-7], -sum y log q; the clip blocks the gradient
//                    outside the interval, the ReLUs where the pre-activation is <= 0), backward with the activations
//                    RECOMPUTED from the frame still in registers -- nothing but the 1 KiB frame is read per sample and
//                    nothing per-sample is written.  Deployed nets: a lane owns a fixed set of weights (4 (+1) conv
//                    positions x F filters x 3 classes of the dense kernel) for the whole slice, so their gradients
//                    accumulate in its registers and only the per-frame reductions (3 class sums) cross lanes (DPP row
//                    butterflies + the four row sums in a fixed order: every lane ends with the same bits).  cnn.py's net: 16-frame tiles on the f32 MFMA
//                    (see train_cnnpy_kernel).  Each wave writes ONE partial gradient vector; the shuffle of fit() is an
//                    index array (`order`), frames are never moved.
//   mdc_train_adam   sums the G partials in a fixed order (the result depends on (count, G) only -- no float atomics, so
//                    a step is reproducible bit for bit; G = count / 2, or / 4 for the 10-filter net, at most 1,024),
//                    scales by 1/count (gradient of the MEAN loss), and applies
//                    TensorFlow 2.4's Adam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g*g-v)(1-b2);
//                    w -= m*alpha/(sqrt(v)+eps).  t lives on the device (the last work-group of this launch to
//                    finish bumps it, after all have read it), so a captured hipGraph of an epoch replays correctly.
// The batch of the reference (1,024 frames x 2,334 parameters) is launch-latency bound on this chip; that is why a step
// is two launches and an epoch needs no host round trip but the final read of the loss.
#include "dense_chain_common.h"      // mdc_internal.h, f32x4, row_allreduce (the 16-lane DPP butterflies of the softmax head)

#include <cmath>
#include <cstddef>
#include <cstring>
#include <new>
#include <vector>

struct mdc_trainer {
    mdc_topology topo{};
    int device = 0;
    int nlayers = 0;
    size_t nk[4]{}, nb[4]{};
    size_t off_k[4]{}, off_b[4]{};      // offsets into the flat parameter vector (Keras layouts, layer by layer: kernel, bias)
    size_t P = 0;
    bool have[4]{};
    float lr = 1e-3f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-7f;      // keras.optimizers.Adam() defaults = the .h5 files' training_config
    float drop_rate = 0.f;       // optional Dropout(rate) behind the conv activations (and cnn.py's Dense(D)): 0 = the reference's nets
    unsigned drop_seed = 0;
    // device state
    float* d_params = nullptr;   // [P] master weights
    float* d_m = nullptr;        // [P] Adam first moment
    float* d_v = nullptr;        // [P] Adam second moment
    float* d_grad = nullptr;     // [P] gradient of the last batch's mean loss
    float* d_partials = nullptr; // [kMaxWaves][Pint]
    size_t Pint = 0;             // floats per partial gradient vector: P for the deployed nets, the MFMA kernel's internal layout for cnn.py's
    int* d_map = nullptr;        // [P][3]: the (up to three) entries of a partial vector that sum to parameter i (-1 = none)
    double* d_loss_partials = nullptr;   // [kMaxWaves]
    void* d_state = nullptr;     // TrainState
};

namespace mdc {

namespace {

constexpr int kMaxWaves = 1024;      // partial gradient vectors per batch
constexpr float kKerasEps = 1e-7f;   // K.epsilon()

struct TrainState {
    long long iterations;      // Adam's `iter` (optimizer_weights/Adam/iter:0 of the .h5): completed updates
    long long train_frames;
    double train_loss;         // sum over frames of the per-sample loss since the last read (fit's running epoch loss)
    long long eval_frames;
    double eval_loss;
    unsigned tickets;          // work-groups of the running reduce + Adam launch that have read `iterations` (the last one bumps it)
    unsigned bad_indices;      // frames skipped because their index (order_dev's, or first + i) lay outside [0, n_frames): mdc_trainer_read reports them
};

// The optional Dropout's counter-based generator (include/mdc.h, mdc_trainer_set_dropout; oracle/oracle_train.py restates it):
// keep an element iff fmix32(k_frame + element * 0xC2B2AE35) >= thr, k_frame = fmix32(k_step ^ (frame * 0x85EBCA6B + site)),
// k_step = fmix32(seed + 0x9E3779B9 * (step + 1)); step = Adam's iteration count, read from the device (a replayed hipGraph
// draws new masks at every step), frame = the frame's index in the data set (a frame's mask does not depend on its batch).
struct DropArgs { unsigned seed, thr; float inv; };
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned drop_step_key(const DropArgs& d, const TrainState* st) {
    return fmix32(d.seed + 0x9E3779B9u * (unsigned)(st->iterations + 1));
}
__device__ __forceinline__ unsigned drop_frame_key(unsigned kstep, long frame, unsigned site) {
    return fmix32(kstep ^ ((unsigned)frame * 0x85EBCA6Bu + site));
}
__device__ __forceinline__ bool drop_keep(unsigned kframe, unsigned element, unsigned thr) {
    return fmix32(kframe + element * 0xC2B2AE35u) >= thr;
}

// The frame a batch position names, or -1 if that is no frame of the caller's buffers (a shuffle index out of range): such a
// position is SKIPPED -- all-zero samples and targets contribute nothing to loss or gradient below -- and counted, once, by the
// lane `count_it` names; mdc_trainer_read turns the count into an error.  An unchecked index would be a GPU fault instead.
__device__ __forceinline__ long frame_index(const int* __restrict__ order, long first, int i, long n_frames, TrainState* st, bool count_it) {
    const long idx = order ? (long)order[first + i] : first + i;
    if ((unsigned long)idx < (unsigned long)n_frames) return idx;
    if (count_it) atomicAdd(&st->bad_indices, 1u);
    return -1;
}

// Sum over the wave, the same bits in every lane: the DPP butterfly over each 16-lane row (no LDS crossbar: __shfl_xor compiles to
// ds_bpermute_b32, six dependent LDS round trips per sum), then the four row sums through scalar registers in a fixed order.
__device__ __forceinline__ float wave_allsum(float v) {
    v = row_allreduce(v, [](float a, float b) { return a + b; });
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// softmax -> Keras categorical_crossentropy on the probabilities -> gradient w.r.t. the softmax INPUT.  y: the target row.
template <int CMAX>
__device__ __forceinline__ float softmax_xent(const float (&d)[CMAX], const float (&y)[CMAX], int C, float (&gd)[CMAX]) {
    float mx = d[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c) if (c < C) mx = fmaxf(mx, d[c]);
    float p[CMAX], se = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { p[c] = c < C ? expf(d[c] - mx) : 0.f; se += p[c]; }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { p[c] = p[c] / se; s += p[c]; }
    float li = 0.f, sg = 0.f, gq[CMAX], q[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        q[c] = p[c] / s;
        const float qc = fminf(fmaxf(q[c], kKerasEps), 1.f - kKerasEps);
        const bool in = c < C && q[c] >= kKerasEps && q[c] <= 1.f - kKerasEps;
        if (c < C && y[c] != 0.f) li -= y[c] * logf(qc);
        gq[c] = in ? -y[c] / qc : 0.f;
        sg += gq[c] * q[c];
    }
    float sp = 0.f, gp[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { gp[c] = (gq[c] - sg) / s; sp += gp[c] * p[c]; }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) gd[c] = p[c] * (gp[c] - sp);
    return li;
}

// ---- deployed net (CNN.ipynb cell 6): lane = (row h = lane>>5, positions w = 4l'..4l'+3, l' = lane&31; lane l' = 31 also w = 128)
// WLDS: the lane's 5 x F x 3 dense weights live in LDS ([entry][lane]: conflict-free) instead of registers -- the gradient
// kernel of the 10-filter net would otherwise hold 150 weights + 150 gradient sums per lane and spill.
template <int F, bool GRAD, bool WLDS, bool DROP = false>
__global__ __launch_bounds__(64) void train_deployed_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           long n_frames, const int* __restrict__ order, long first, int count,
                                                           const float* __restrict__ params, float* __restrict__ partials,
                                                           double* __restrict__ loss_partials, TrainState* __restrict__ st, DropArgs drop) {
    constexpr int C = 3, S = 5;
    constexpr int offCb = 2 * F, offWd = 3 * F, offBd = 3 * F + 258 * F * C, P = offBd + C;
    __shared__ float sW[WLDS ? S * F * C * 64 : 1];
    const int lane = threadIdx.x, h = lane >> 5, lp = lane & 31;
    const int G = gridDim.x, g = blockIdx.x;
    float k0[F], k1[F], cb[F], bd[C];
#pragma unroll
    for (int f = 0; f < F; ++f) { k0[f] = params[f]; k1[f] = params[F + f]; cb[f] = params[offCb + f]; }
#pragma unroll
    for (int c = 0; c < C; ++c) bd[c] = params[offBd + c];
    // dense kernel row of (h, w, f): h*129F + w*F + f   (channels_last Flatten).  A lane's positions w = 4l'..4l'+3 are ONE
    // contiguous run of 4 F C floats (entry (s, f, c) at (s F + f) C + c), lane 31's position 128 the F C floats behind its run:
    // the run is fetched (and its gradient written) as 16-byte pieces -- a quarter of the instructions and line requests of
    // one dword per entry across 64 lanes 4 F C floats apart (4-byte aligned only: row 1 starts 129 F C floats in).
    struct __attribute__((packed, aligned(4))) Run4 { float v[4]; };
    constexpr int FC = F * C, RUN = 4 * FC;
    static_assert(RUN % 4 == 0, "a lane's run is whole 16-byte pieces");
    const long run0 = offWd + (long)(h * 129 + 4 * lp) * FC;
    float Wr[WLDS ? 1 : S][WLDS ? 1 : F][WLDS ? 1 : C], dW[GRAD ? S : 1][GRAD ? F : 1][GRAD ? C : 1];
    auto put = [&](int e, float v) {       // e: compile-time entry index (s F + f) C + c
        if (WLDS) sW[e * 64 + lane] = v;   // (read back by this lane only: no barrier needed)
        else Wr[WLDS ? 0 : e / FC][WLDS ? 0 : (e / C) % F][WLDS ? 0 : e % C] = v;
    };
#pragma unroll
    for (int q = 0; q < RUN / 4; ++q) {
        const Run4 r = *reinterpret_cast<const Run4*>(params + run0 + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) put(4 * q + i, r.v[i]);
    }
#pragma unroll
    for (int e = 0; e < FC; ++e) put(RUN + e, lp == 31 ? params[run0 + RUN + e] : 0.f);
    if (GRAD) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int c = 0; c < C; ++c) dW[s][f][c] = 0.f;
    }
    auto Wv = [&](int s, int f, int c) -> float { return WLDS ? sW[((s * F + f) * C + c) * 64 + lane] : Wr[WLDS ? 0 : s][WLDS ? 0 : f][WLDS ? 0 : c]; };
    float gk0[F], gk1[F], gcb[F], gbd[C];
#pragma unroll
    for (int f = 0; f < F; ++f) gk0[f] = gk1[f] = gcb[f] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) gbd[c] = 0.f;
    double loss = 0.0;
    const unsigned kstep = DROP ? drop_step_key(drop, st) : 0u;

    // the next frame's samples and targets are loaded while this frame is computed (a wave walks its frames one at a time;
    // the sched_barrier keeps hipcc from sinking the loads to their first use)
    long idx_n = 0;
    auto fetch = [&](int i, float4& xv, float (&yv)[C]) {
        const long idx = frame_index(order, first, i, n_frames, st, lane == 0);
        idx_n = idx;
        xv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < C; ++c) yv[c] = 0.f;
        if (idx >= 0) {
            xv = *reinterpret_cast<const float4*>(x + idx * kFrameFloats + h * kSamples + 4 * lp);
#pragma unroll
            for (int c = 0; c < C; ++c) yv[c] = y[idx * C + c];
        }
    };
    float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
    float yn[C] = {0.f, 0.f, 0.f};
    if (g < count) fetch(g, xn, yn);
    for (int i = g; i < count; i += G) {
        if (WLDS) asm volatile("" ::: "memory");      // keep the LDS weight reads inside the loop (hoisted, they are 150 registers again)
        const float4 xv = xn;
        const long idx_cur = idx_n;
        float yv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) yv[c] = yn[c];
        if (i + G < count) fetch(i + G, xn, yn);
        __builtin_amdgcn_sched_barrier(0);
        // Dropout behind the conv activations: one keep bit per (slot, filter) of this lane, element = the Flatten index
        unsigned long long keep = ~0ull;
        if (DROP) {
            const unsigned kf = drop_frame_key(kstep, idx_cur, 0u);
            keep = 0ull;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const unsigned e = (unsigned)((h * 129 + (s < 4 ? 4 * lp + s : 128)) * F + f);
                    if (drop_keep(kf, e, drop.thr)) keep |= 1ull << (s * F + f);
                }
        }
        auto scale_of = [&](int s, int f) -> float { return DROP ? (((keep >> (s * F + f)) & 1ull) ? drop.inv : 0.f) : 1.f; };
        float xprev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xv.w), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
        if (lp == 0) xprev = 0.f;                                       // ZeroPadding2D((0,1)): x[h][-1] = 0
        const float xin[S] = {xprev, xv.x, xv.y, xv.z, xv.w};           // x[h][w-1]
        const float xcu[S] = {xv.x, xv.y, xv.z, xv.w, 0.f};             // x[h][w]   (x[h][128] = 0)
        float z[C] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                float a = fmaxf(fmaf(k1[f], xcu[s], k0[f] * xin[s]) + cb[f], 0.f);
                if (DROP) a *= scale_of(s, f);
#pragma unroll
                for (int c = 0; c < C; ++c) z[c] = fmaf(a, Wv(s, f, c), z[c]);
            }
        float d[C], gz[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { z[c] = wave_allsum(z[c]) + bd[c]; d[c] = fmaxf(z[c], 0.f); }      // Dense(3, activation='relu')
        loss += (double)softmax_xent<C>(d, yv, C, gz);
        if (GRAD) {
            if (WLDS) asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < C; ++c) { gz[c] = z[c] > 0.f ? gz[c] : 0.f; gbd[c] += gz[c]; }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const float pre = fmaf(k1[f], xcu[s], k0[f] * xin[s]) + cb[f];
                    float a = fmaxf(pre, 0.f);
                    if (DROP) a *= scale_of(s, f);
                    float da = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { da = fmaf(Wv(s, f, c), gz[c], da); dW[s][f][c] = fmaf(a, gz[c], dW[s][f][c]); }
                    if (DROP) da *= scale_of(s, f);
                    const float dpre = pre > 0.f ? da : 0.f;            // (W = 0 in the slots a lane does not own: da = 0 there)
                    gk0[f] = fmaf(dpre, xin[s], gk0[f]);
                    gk1[f] = fmaf(dpre, xcu[s], gk1[f]);
                    gcb[f] += dpre;
                }
        }
    }
    if (lane == 0) loss_partials[g] = loss;
    if (!GRAD) return;
    float* out = partials + (size_t)g * P;
    auto dWe = [&](int e) -> float { return dW[GRAD ? e / FC : 0][GRAD ? (e / C) % F : 0][GRAD ? e % C : 0]; };
#pragma unroll
    for (int q = 0; q < RUN / 4; ++q) {
        Run4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.v[i] = dWe(4 * q + i);
        *reinterpret_cast<Run4*>(out + run0 + 4 * q) = r;
    }
    if (lp == 31) {
#pragma unroll
        for (int e = 0; e < FC; ++e) out[run0 + RUN + e] = dWe(RUN + e);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float a = wave_allsum(gk0[f]), b = wave_allsum(gk1[f]), c = wave_allsum(gcb[f]);
        if (lane == 0) { out[f] = a; out[F + f] = b; out[offCb + f] = c; }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) out[offBd + c] = gbd[c];
    }
}

// ---- cnn.py's literal net (cnn.py:104-112 as TensorFlow builds it: H=1, W=2, C=128) on v_mfma_f32_16x16x4_f32, 16 frames at a time.
// pad + Conv2D(F,(1,2)) over (1, 2+2, 128) is a LINEAR map of the frame's 256 floats to 3F values (csrc/cnnpy.hip folds it the
// same way): pre = x . Mc,  Mc[c][w=0] = K1[c], Mc[c][w=1] = K0[c] (I half), Mc[128+c][w=1] = K1[c], Mc[128+c][w=2] = K0[c].
// Everything is a small GEMM with M = the 16 frames of a tile:
//   forward   PRE[16][32] = X[16][256] Mc  (128 MFMAs) -> ReLU -> Z1 = A1 W1 (8) -> ReLU -> LG = H W2 (4) -> softmax / loss
//   backward  dW2 += H^T dLG, dW1 += A1^T dZ1, dMc += X^T dPRE: sums over FRAMES, i.e. K = the tile's 16 frames.  With k-step i
//             carrying frames 4 (lane>>4) + i, the MFMA's A and B operands are exactly registers i of the C/D-layout results
//             (lane = column, rows 4 (lane>>4) + r): no transposition, no LDS.  dH = dLG W2^T and dA1 = dZ1 W1^T need the
//             row-major form of a C/D result: a 2 KiB LDS round trip each, as in the forward chain (dense_chain.hip).
// A wave keeps Mc (128 registers as B operands), the small layers' operands and the gradient accumulators (dMc: 128
// registers) for its whole slice of the batch and writes ONE partial vector in an internal layout ([256][32] for dMc, 16-wide
// small layers); train_adam_kernel folds it into Keras' layout through an index map (each conv weight is the sum of two dMc
// entries, each conv bias of three column sums).  A 1,024-frame batch is 64 waves x one tile.
constexpr int kTF = 10, kTD = 16, kTC = 16;      // bounds of MDC_KIND_CNNPY (mdc_create): filters <= 10, hidden <= 16, classes <= 16
constexpr int kT4Mc = 0, kT4Cb = 256 * 32, kT4W1 = kT4Cb + 32, kT4B1 = kT4W1 + 32 * 16, kT4W2 = kT4B1 + 16, kT4B2 = kT4W2 + 16 * 16,
              kT4Pint = kT4B2 + 16;              // 9,024 floats per partial
constexpr int kT4Xld = kChainXld, kT4Yld = 36;

template <bool GRAD, bool DROP = false>
__global__ __launch_bounds__(64) void train_cnnpy_kernel(const float* __restrict__ x, const float* __restrict__ y, long n_frames,
                                                        const int* __restrict__ order, long first, int count, int F, int D, int C,
                                                        const float* __restrict__ params, float* __restrict__ partials,
                                                        double* __restrict__ loss_partials, TrainState* __restrict__ st, DropArgs drop) {
    const int A = 3 * F;
    const int offCb = 256 * F, offW1 = offCb + F, offB1 = offW1 + A * D, offW2 = offB1 + D, offB2 = offW2 + D * C;
    __shared__ __attribute__((aligned(16))) float xs[16 * kT4Xld];
    __shared__ float ys[16 * kT4Yld];
    __shared__ double sloss[4];
    const int lane = threadIdx.x, fr = lane & 15, g = lane >> 4;
    const int G = gridDim.x;

    // ---- operands (B[k = 4i + g][col = fr] per k-step i), rebuilt from the master weights at every launch
    float Mw[2][64], cbv[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = 16 * jt + fr, w = j / F, f = j - w * F;
        const bool col_ok = j < A;
        cbv[jt] = col_ok ? params[offCb + f] : 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int k = 4 * i + g, c = k & 127;
            // I half (k < 128): w 0 -> tap 1, w 1 -> tap 0; Q half: w 1 -> tap 1, w 2 -> tap 0
            const int kw = k < 128 ? (w == 0 ? 1 : w == 1 ? 0 : -1) : (w == 1 ? 1 : w == 2 ? 0 : -1);
            Mw[jt][i] = (col_ok && kw >= 0) ? params[(kw * 128 + c) * F + f] : 0.f;
        }
    }
    float W1b[8], W2b[4], W2t[4], W1t[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int j = 4 * i + g; W1b[i] = (j < A && fr < D) ? params[offW1 + j * D + fr] : 0.f; }      // Z1 = A1 W1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 4 * i + g;
        W2b[i] = (k < D && fr < C) ? params[offW2 + k * C + fr] : 0.f;            // LG = H W2:      B[k = d][col = c]
        W2t[i] = (k < C && fr < D) ? params[offW2 + fr * C + k] : 0.f;            // dH = dLG W2^T:  B[k = c][col = d]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) { const int j = 16 * jt + fr; W1t[jt][i] = (j < A && k < D) ? params[offW1 + j * D + k] : 0.f; }      // dA1 = dZ1 W1^T: B[k = d][col = j]
    }
    const float b1v = fr < D ? params[offB1 + fr] : 0.f, b2v = fr < C ? params[offB2 + fr] : 0.f;
    const bool cls = fr < C;

    f32x4 dM[GRAD ? 16 : 1][2], dW1a[2], dW2a;
    float gcb[2] = {0.f, 0.f}, gb1 = 0.f, gb2 = 0.f;
    if (GRAD) {
#pragma unroll
        for (int T = 0; T < 16; ++T) { dM[T][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dM[T][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    dW1a[0] = dW1a[1] = dW2a = f32x4{0.f, 0.f, 0.f, 0.f};
    double loss = 0.0;
    const unsigned kstep = DROP ? drop_step_key(drop, st) : 0u;

    const int ntiles = (count + 15) >> 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int f0 = tile << 4;
        // ---- stage the tile's 16 frames (1 KiB each, one float4 per lane), zeros past the end of the batch
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f0 + r < count) {
                const long idx = frame_index(order, first, f0 + r, n_frames, st, lane == 0);
                if (idx >= 0) v = reinterpret_cast<const float4*>(x + idx * kFrameFloats)[lane];
            }
            *reinterpret_cast<float4*>(xs + r * kT4Xld + 4 * lane) = v;
        }
        // targets in the C/D layout: lane (class fr, rows 4g + r); zero rows past the end contribute nothing anywhere below
        float yv[4], m0[2][4], m1[4];      // targets; Dropout scales of the lane's conv activations (site 0) and hidden units (site 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int fi = f0 + 4 * g + r;
            yv[r] = 0.f;
            m0[0][r] = m0[1][r] = m1[r] = 1.f;
            const long idx = fi < count ? frame_index(order, first, fi, n_frames, st, false) : -1;      // (counted where the frame is staged)
            if (idx >= 0) {
                if (cls) yv[r] = y[idx * C + fr];
                if (DROP) {
                    const unsigned k0f = drop_frame_key(kstep, idx, 0u), k1f = drop_frame_key(kstep, idx, 1u);
                    m0[0][r] = drop_keep(k0f, (unsigned)fr, drop.thr) ? drop.inv : 0.f;
                    m0[1][r] = drop_keep(k0f, (unsigned)(16 + fr), drop.thr) ? drop.inv : 0.f;
                    m1[r] = drop_keep(k1f, (unsigned)fr, drop.thr) ? drop.inv : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- forward
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        {   // A operands a pair of k-steps ahead of the MFMAs that use them (pinned: hipcc otherwise re-uses two registers and waits
            // out the LDS latency in front of every four MFMAs -- one wave per SIMD here, nothing else to fill it; dense_chain.hip)
            float a_cur[2] = {xs[fr * kT4Xld + g], xs[fr * kT4Xld + 4 + g]}, a_nxt[2] = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j + 1 < 32) { a_nxt[0] = xs[fr * kT4Xld + 8 * (j + 1) + g]; a_nxt[1] = xs[fr * kT4Xld + 8 * (j + 1) + 4 + g]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u], Mw[0][2 * j + u], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u], Mw[1][2 * j + u], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a_cur[0] = a_nxt[0];
                a_cur[1] = a_nxt[1];
            }
        }
        f32x4 pre[2], a1[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pre[jt][r] = acc[jt][r] + cbv[jt];
                a1[jt][r] = fmaxf(pre[jt][r], 0.f);
                if (DROP) a1[jt][r] *= m0[jt][r];
                ys[(4 * g + r) * kT4Yld + 16 * jt + fr] = a1[jt][r];
            }
        __syncthreads();
        f32x4 z1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kT4Yld + 4 * i + g], W1b[i], z1, 0, 0, 0);
        f32x4 h;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // Dense(D, relu)
            z1[r] += b1v;
            h[r] = fmaxf(z1[r], 0.f);
            if (DROP) h[r] *= m1[r];
            ys[(4 * g + r) * kT4Yld + fr] = h[r];
        }
        __syncthreads();
        f32x4 lg = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) lg = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kT4Yld + 4 * i + g], W2b[i], lg, 0, 0, 0);
        // ---- softmax over the classes (the 16 lanes of a DPP row), Keras' cross-entropy on the probabilities, d loss / d logits
        f32x4 glg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            auto sum16 = [](float v) { return row_allreduce(v, [](float a, float b) { return a + b; }); };
            const float zz = cls ? lg[r] + b2v : -INFINITY;
            const float mx = row_allreduce(zz, [](float a, float b) { return fmaxf(a, b); });
            const float e = cls ? expf(zz - mx) : 0.f;
            const float pv = e / sum16(e);
            const float s = sum16(pv);
            const float q = pv / s;
            const float qc = fminf(fmaxf(q, kKerasEps), 1.f - kKerasEps);
            const bool in = cls && q >= kKerasEps && q <= 1.f - kKerasEps;
            const float li = sum16((cls && yv[r] != 0.f) ? -yv[r] * logf(qc) : 0.f);
            const float gq = in ? -yv[r] / qc : 0.f;
            const float gp = (gq - sum16(gq * q)) / s;
            glg[r] = pv * (gp - sum16(gp * pv));
            if (fr == 0) loss += (double)li;
        }
        if (GRAD) {
            // ---- sums over the tile's frames: C/D-layout registers ARE the operands (k-step i = frames 4 (lane>>4) + i)
#pragma unroll
            for (int i = 0; i < 4; ++i) dW2a = __builtin_amdgcn_mfma_f32_16x16x4f32(h[i], glg[i], dW2a, 0, 0, 0);      // rows d, columns c
            gb2 += (glg[0] + glg[1]) + (glg[2] + glg[3]);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[(4 * g + r) * kT4Yld + fr] = glg[r];
            __syncthreads();
            f32x4 dz1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) dz1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kT4Yld + 4 * i + g], W2t[i], dz1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[r] = z1[r] > 0.f ? (DROP ? dz1[r] * m1[r] : dz1[r]) : 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int i = 0; i < 4; ++i) dW1a[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[jt][i], dz1[i], dW1a[jt], 0, 0, 0);      // rows j, columns d
            gb1 += (dz1[0] + dz1[1]) + (dz1[2] + dz1[3]);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[(4 * g + r) * kT4Yld + fr] = dz1[r];
            __syncthreads();
            f32x4 dpre[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                f32x4 da = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) da = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kT4Yld + 4 * i + g], W1t[jt][i], da, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) dpre[jt][r] = pre[jt][r] > 0.f ? (DROP ? da[r] * m0[jt][r] : da[r]) : 0.f;
                gcb[jt] += (dpre[jt][0] + dpre[jt][1]) + (dpre[jt][2] + dpre[jt][3]);
            }
            {   // A[row = input 16T + fr][k = frame 4g + i]: the next input tile's four values are read before this one's MFMAs issue
                float x_cur[4], x_nxt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) x_cur[i] = xs[(4 * g + i) * kT4Xld + fr];
#pragma unroll
                for (int T = 0; T < 16; ++T) {
                    if (T + 1 < 16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x_nxt[i] = xs[(4 * g + i) * kT4Xld + 16 * (T + 1) + fr];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dM[T][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x_cur[i], dpre[0][i], dM[T][0], 0, 0, 0);
                        dM[T][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x_cur[i], dpre[1][i], dM[T][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) x_cur[i] = x_nxt[i];
                }
            }
        }
        __syncthreads();      // xs / ys are rewritten by the next tile
    }
    // ---- the wave's loss (lanes fr == 0 hold the rows of their g) and its partial gradient vector (internal layout)
    if (fr == 0) sloss[g] = loss;
    __syncthreads();
    if (lane == 0) loss_partials[blockIdx.x] = (sloss[0] + sloss[1]) + (sloss[2] + sloss[3]);
    if (!GRAD) return;
    float* out = partials + (size_t)blockIdx.x * kT4Pint;
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[kT4Mc + (16 * T + 4 * g + r) * 32 + 16 * jt + fr] = dM[GRAD ? T : 0][jt][r];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[kT4W1 + (16 * jt + 4 * g + r) * 16 + fr] = dW1a[jt][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) out[kT4W2 + (4 * g + r) * 16 + fr] = dW2a[r];
    // column sums: over the four lane groups (xor 16, 32: the same bits in every lane)
    auto over_g = [](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
    const float c0 = over_g(gcb[0]), c1 = over_g(gcb[1]), s1 = over_g(gb1), s2 = over_g(gb2);
    if (g == 0) { out[kT4Cb + fr] = c0; out[kT4Cb + 16 + fr] = c1; out[kT4B1 + fr] = s1; out[kT4B2 + fr] = s2; }
}

// ---- fixed-order reduction of the partials + TensorFlow 2.4's Adam.  Block = 64 parameters x 16 slices of the G partials: a
// thread's <= 64 loads are independent (unrolled 16 at a time: the first version's one-load-at-a-time chain of 64 took 30 us,
// twice the gradient kernel), the sums are taken in a fixed order -- slice by slice, then the 16 slices in order.
// mode: 0 = losses only (evaluation), 1 = gradient stored, 2 = gradient stored and applied
constexpr int kRedSlices = 16;

__global__ __launch_bounds__(1024) void train_adam_kernel(const float* __restrict__ partials, const double* __restrict__ loss_partials, int G,
                                                         int P, int Pint, const int* __restrict__ map, int count, int mode, float lr,
                                                         float beta1, float beta2, float eps,
                                                         float* __restrict__ params, float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ grad, TrainState* __restrict__ st) {
    __shared__ float part[kRedSlices][64];
    __shared__ double lpart[1024];
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const long long it0 = st->iterations;      // every thread reads it before its work-group takes a ticket (below)
    if (mode != 0) {
        float acc = 0.f;
        if (i < P) {
            const int per = (G + kRedSlices - 1) / kRedSlices, lo = slice * per, hi = min(G, lo + per);
            // parameter i = the sum of up to three entries of a partial vector (one for the deployed nets; cnn.py's conv weights
            // are two entries of dMc, its conv bias three column sums): entry by entry, partials lo .. hi in order
            for (int e = 0; e < 3; ++e) {
                const int me = map[3 * i + e];
                if (me < 0) break;
                const float* src = partials + (size_t)lo * Pint + me;
                int w = lo;
                for (; w + 16 <= hi; w += 16, src += (size_t)16 * Pint) {
                    float t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t[u] = src[(size_t)u * Pint];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += t[u];
                }
                for (; w < hi; ++w, src += Pint) acc += *src;
            }
        }
        part[slice][col] = acc;
        __syncthreads();
        if (slice == 0 && i < P) {
            float gsum = part[0][col];
#pragma unroll
            for (int sl = 1; sl < kRedSlices; ++sl) gsum += part[sl][col];
            const float gmean = gsum / (float)count;
            grad[i] = gmean;
            if (mode == 2) {
                const float t = (float)(it0 + 1);            // this update's number; `iterations` is bumped by the last work-group to finish
                const float alpha = lr * sqrtf(1.f - powf(beta2, t)) / (1.f - powf(beta1, t));
                const float mi = m[i] + (gmean - m[i]) * (1.f - beta1);
                const float vi = v[i] + (gmean * gmean - v[i]) * (1.f - beta2);
                m[i] = mi;
                v[i] = vi;
                params[i] -= (mi * alpha) / (sqrtf(vi) + eps);
            }
        }
    }
    if (blockIdx.x == 0) {      // the batch's summed loss: G <= 1,024 partials, one per thread, a fixed tree
        lpart[threadIdx.x] = (int)threadIdx.x < G ? loss_partials[threadIdx.x] : 0.0;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) lpart[threadIdx.x] += lpart[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (mode == 0) { st->eval_loss += lpart[0]; st->eval_frames += count; }
            else { st->train_loss += lpart[0]; st->train_frames += count; }
        }
    }
    if (mode == 2) {
        // Adam's step count advances ONCE, after every work-group of this launch has read it: the gradient kernels only ever read
        // it (their Dropout masks are keyed by it), so it is stable for the whole of the next launch
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(&st->tickets, 1u) == gridDim.x - 1) {
                st->tickets = 0u;
                st->iterations = it0 + 1;
            }
        }
    }
}

int trainer_layout(mdc_trainer* t) {
    const mdc_topology& tp = t->topo;
    if (tp.kind == MDC_KIND_DEPLOYED) {
        if (tp.filters != 3 && tp.filters != 10) { set_error("trainer: deployed filters must be 3 or 10 (got %d)", tp.filters); return MDC_ENOTSUP; }
        if (tp.classes != 3) { set_error("trainer: deployed classes must be 3 (got %d)", tp.classes); return MDC_ENOTSUP; }
        t->nlayers = 2;
        t->nk[0] = 2 * (size_t)tp.filters;                  t->nb[0] = tp.filters;
        t->nk[1] = 258 * (size_t)tp.filters * tp.classes;   t->nb[1] = tp.classes;
    } else if (tp.kind == MDC_KIND_CNNPY) {
        if (tp.filters < 1 || tp.filters > kTF || tp.hidden < 1 || tp.hidden > kTD || tp.classes < 2 || tp.classes > kTC) {
            set_error("trainer: cnnpy needs filters 1..%d, hidden 1..%d, classes 2..%d (got %d,%d,%d)", kTF, kTD, kTC, tp.filters, tp.hidden, tp.classes);
            return MDC_ENOTSUP;
        }
        t->nlayers = 3;
        t->nk[0] = 2 * 128 * (size_t)tp.filters;            t->nb[0] = tp.filters;
        t->nk[1] = 3 * (size_t)tp.filters * tp.hidden;      t->nb[1] = tp.hidden;
        t->nk[2] = (size_t)tp.hidden * tp.classes;          t->nb[2] = tp.classes;
    } else {
        set_error("trainer: the reference trains MDC_KIND_DEPLOYED (CNN.ipynb cell 6) and MDC_KIND_CNNPY (cnn.py:104-112) only");
        return MDC_ENOTSUP;
    }
    size_t off = 0;
    for (int l = 0; l < t->nlayers; ++l) {
        t->off_k[l] = off; off += t->nk[l];
        t->off_b[l] = off; off += t->nb[l];
    }
    t->P = off;
    t->Pint = tp.kind == MDC_KIND_CNNPY ? (size_t)kT4Pint : off;
    return MDC_OK;
}

// [P][3] index map from Keras' parameter order into a partial vector (see train_adam_kernel)
std::vector<int> trainer_map(const mdc_trainer* t) {
    std::vector<int> m(t->P * 3, -1);
    if (t->topo.kind != MDC_KIND_CNNPY) {
        for (size_t i = 0; i < t->P; ++i) m[3 * i] = (int)i;
        return m;
    }
    const int F = t->topo.filters, D = t->topo.hidden, C = t->topo.classes;
    auto set = [&](size_t i, int a, int b = -1, int c = -1) { m[3 * i] = a; m[3 * i + 1] = b; m[3 * i + 2] = c; };
    for (int c = 0; c < 128; ++c)
        for (int f = 0; f < F; ++f) {
            set(t->off_k[0] + (size_t)(0 * 128 + c) * F + f, kT4Mc + c * 32 + F + f, kT4Mc + (128 + c) * 32 + 2 * F + f);      // tap 0: (I, w = 1), (Q, w = 2)
            set(t->off_k[0] + (size_t)(1 * 128 + c) * F + f, kT4Mc + c * 32 + f, kT4Mc + (128 + c) * 32 + F + f);              // tap 1: (I, w = 0), (Q, w = 1)
        }
    for (int f = 0; f < F; ++f) set(t->off_b[0] + f, kT4Cb + f, kT4Cb + F + f, kT4Cb + 2 * F + f);
    for (int j = 0; j < 3 * F; ++j)
        for (int d = 0; d < D; ++d) set(t->off_k[1] + (size_t)j * D + d, kT4W1 + j * 16 + d);
    for (int d = 0; d < D; ++d) set(t->off_b[1] + d, kT4B1 + d);
    for (int d = 0; d < D; ++d)
        for (int c = 0; c < C; ++c) set(t->off_k[2] + (size_t)d * C + c, kT4W2 + d * 16 + c);
    for (int c = 0; c < C; ++c) set(t->off_b[2] + c, kT4B2 + c);
    return m;
}

int waves_for(const mdc_trainer* t, int64_t count) {
    // A wave walks its frames one at a time, so the batch's latency is (frames per wave) x (one frame): two frames per wave
    // for the nets whose partial vector is small (T1: 2,334 floats, T4: 2,935 -- 512 waves for the reference's 1,024-frame
    // batch), four for the 10-filter net (7,773 floats per partial: the reduction's traffic would double); never more than
    // kMaxWaves (evaluation of a large set then walks count / 1,024 frames per wave with every CU busy).
    // cnn.py's net: one wave per 16-frame MFMA tile.
    const int per_wave = t->topo.kind == MDC_KIND_CNNPY ? 16 : (t->topo.filters == 10 ? 4 : 2);
    int64_t g = (count + per_wave - 1) / per_wave;
    if (g < 1) g = 1;
    if (g > kMaxWaves) g = kMaxWaves;
    return (int)g;
}

int launch_batch(mdc_trainer* t, const float* x, const float* y, int64_t n_frames, const int32_t* order, int64_t first, int64_t count, int mode, hipStream_t s) {
    const int G = waves_for(t, count);
    auto* st = static_cast<TrainState*>(t->d_state);
    const int cnt = (int)count;
    // Dropout acts in training batches only (mode != 0: Keras' fit reports the loss WITH it, evaluates val_loss without)
    const bool drop_on = mode != 0 && t->drop_rate > 0.f;
    const DropArgs drop{t->drop_seed, (unsigned)std::floor((double)t->drop_rate * 4294967296.0), 1.f / (1.f - t->drop_rate)};
#define MDC_TRAIN_DEP(F, GR, WL, DR) hipLaunchKernelGGL((train_deployed_kernel<F, GR, WL, DR>), dim3(G), dim3(64), 0, s, x, y, (long)n_frames, order, (long)first, cnt, \
                                                         t->d_params, t->d_partials, t->d_loss_partials, st, drop)
#define MDC_TRAIN_T4(GR, DR) hipLaunchKernelGGL((train_cnnpy_kernel<GR, DR>), dim3(G), dim3(64), 0, s, x, y, (long)n_frames, order, (long)first, cnt, t->topo.filters, \
                                                 t->topo.hidden, t->topo.classes, t->d_params, t->d_partials, t->d_loss_partials, st, drop)
    if (t->topo.kind == MDC_KIND_DEPLOYED) {
        if (t->topo.filters == 3) {
            if (!mode) MDC_TRAIN_DEP(3, false, false, false);
            else if (drop_on) MDC_TRAIN_DEP(3, true, false, true);
            else MDC_TRAIN_DEP(3, true, false, false);
        } else {
            if (!mode) MDC_TRAIN_DEP(10, false, false, false);
            else if (drop_on) MDC_TRAIN_DEP(10, true, true, true);
            else MDC_TRAIN_DEP(10, true, true, false);
        }
    } else {
        if (!mode) MDC_TRAIN_T4(false, false);
        else if (drop_on) MDC_TRAIN_T4(true, true);
        else MDC_TRAIN_T4(true, false);
    }
#undef MDC_TRAIN_DEP
#undef MDC_TRAIN_T4
    MDC_HIP(hipGetLastError());
    const int P = (int)t->P;
    const int blocks = mode == 0 ? 1 : (P + 63) / 64;
    hipLaunchKernelGGL(train_adam_kernel, dim3(blocks), dim3(1024), 0, s, t->d_partials, t->d_loss_partials, G, P, (int)t->Pint, t->d_map, cnt, mode, t->lr, t->beta1, t->beta2,
                       t->eps, t->d_params, t->d_m, t->d_v, t->d_grad, st);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

int check_batch_args(const char* what, mdc_trainer* t, const float* x, const float* y, int64_t n_frames, const int32_t* order, int64_t first,
                     int64_t count) {
    if (!t) { set_error("%s: null trainer", what); return MDC_EINVAL; }
    for (int l = 0; l < t->nlayers; ++l)
        if (!t->have[l]) { set_error("%s: layer %d has no weights (mdc_trainer_set_weights)", what, l); return MDC_ESTATE; }
    if (first < 0 || count < 0) { set_error("%s: negative range", what); return MDC_EINVAL; }
    if (count > (int64_t)1 << 30) { set_error("%s: at most 2^30 frames per call", what); return MDC_EINVAL; }
    if (count > 0 && (!x || !y)) { set_error("%s: null frames or targets", what); return MDC_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) { set_error("%s: frames must be 16-byte aligned", what); return MDC_EINVAL; }
    if (n_frames < 0) { set_error("%s: negative n_frames", what); return MDC_EINVAL; }
    // without a shuffle the positions ARE the frames: checked here; with one, its VALUES are checked on the device (frame_index)
    if (!order && first + count > n_frames) {
        set_error("%s: frames [%lld, %lld) outside the %lld of the buffers", what, (long long)first, (long long)(first + count), (long long)n_frames);
        return MDC_EINVAL;
    }
    return MDC_OK;
}

}  // namespace

}  // namespace mdc

using namespace mdc;

extern "C" {

int mdc_trainer_create(const mdc_topology* topo, int device, mdc_trainer** out) {
    return guarded("mdc_trainer_create", [&]() -> int {
        if (!topo || !out) { set_error("mdc_trainer_create: null argument"); return MDC_EINVAL; }
        *out = nullptr;
        for (int i = 0; i < 4; ++i) if (topo->reserved[i] != 0) { set_error("mdc_trainer_create: reserved[] must be 0"); return MDC_EINVAL; }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available"); return MDC_ENODEV; }
        if (device < 0 || device >= ndev) { set_error("device %d out of range (have %d)", device, ndev); return MDC_ENODEV; }
        hipDeviceProp_t prop;
        MDC_HIP(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_error("device %d is %s; libmdc.so carries gfx950 (MI355X) code only", device, prop.gcnArchName);
            return MDC_ENODEV;
        }
        mdc_trainer* t = new (std::nothrow) mdc_trainer();
        if (!t) { set_error("out of host memory"); return MDC_ENOMEM; }
        t->topo = *topo;
        t->device = device;
        int rc = trainer_layout(t);
        if (rc != MDC_OK) { delete t; return rc; }
        DeviceScope dev(device);
        if (!dev.ok) { set_error("mdc_trainer_create: cannot select device %d", device); delete t; return MDC_EIO; }
        const size_t pb = t->P * sizeof(float);
        auto fail = [&](int code) { mdc_trainer_destroy(t); return code; };
        if (hipMalloc(&t->d_params, pb) != hipSuccess || hipMalloc(&t->d_m, pb) != hipSuccess || hipMalloc(&t->d_v, pb) != hipSuccess ||
            hipMalloc(&t->d_grad, pb) != hipSuccess || hipMalloc(&t->d_partials, t->Pint * sizeof(float) * kMaxWaves) != hipSuccess ||
            hipMalloc(&t->d_map, t->P * 3 * sizeof(int)) != hipSuccess ||
            hipMalloc(&t->d_loss_partials, sizeof(double) * kMaxWaves) != hipSuccess || hipMalloc(&t->d_state, sizeof(TrainState)) != hipSuccess) {
            set_error("mdc_trainer_create: out of device memory");
            return fail(MDC_ENOMEM);
        }
        if (hipMemset(t->d_params, 0, pb) != hipSuccess || hipMemset(t->d_m, 0, pb) != hipSuccess || hipMemset(t->d_v, 0, pb) != hipSuccess ||
            hipMemset(t->d_grad, 0, pb) != hipSuccess || hipMemset(t->d_state, 0, sizeof(TrainState)) != hipSuccess) {
            set_error("mdc_trainer_create: hipMemset failed");
            return fail(MDC_EIO);
        }
        const std::vector<int> map = trainer_map(t);
        if (hipMemcpy(t->d_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("mdc_trainer_create: hipMemcpy failed");
            return fail(MDC_EIO);
        }
        *out = t;
        return MDC_OK;
    });
}

int mdc_trainer_num_layers(const mdc_trainer* t) {
    if (!t) { set_error("null trainer"); return MDC_EINVAL; }
    return t->nlayers;
}

int mdc_trainer_layer_sizes(const mdc_trainer* t, int layer, size_t* kernel_elems, size_t* bias_elems) {
    if (!t || layer < 0 || layer >= t->nlayers) { set_error("mdc_trainer_layer_sizes: bad layer %d", layer); return MDC_EINVAL; }
    if (kernel_elems) *kernel_elems = t->nk[layer];
    if (bias_elems) *bias_elems = t->nb[layer];
    return MDC_OK;
}

int mdc_trainer_set_adam(mdc_trainer* t, float lr, float beta1, float beta2, float eps) {
    if (!t) { set_error("null trainer"); return MDC_EINVAL; }
    if (!(lr > 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) {
        set_error("mdc_trainer_set_adam: need lr > 0, 0 <= beta < 1, eps >= 0");
        return MDC_EINVAL;
    }
    t->lr = lr; t->beta1 = beta1; t->beta2 = beta2; t->eps = eps;
    return MDC_OK;
}

int mdc_trainer_set_dropout(mdc_trainer* t, float rate, uint32_t seed) {
    if (!t) { set_error("null trainer"); return MDC_EINVAL; }
    if (!(rate >= 0.f && rate < 1.f)) { set_error("mdc_trainer_set_dropout: need 0 <= rate < 1"); return MDC_EINVAL; }
    t->drop_rate = rate;
    t->drop_seed = seed;
    return MDC_OK;
}

// which: 0 weights, 1 Adam m, 2 Adam v, 3 gradient of the last batch
static float* trainer_vec(mdc_trainer* t, int which) {
    switch (which) {
        case MDC_TRAIN_WEIGHTS: return t->d_params;
        case MDC_TRAIN_ADAM_M:  return t->d_m;
        case MDC_TRAIN_ADAM_V:  return t->d_v;
        case MDC_TRAIN_GRADIENT: return t->d_grad;
        default: return nullptr;
    }
}

int mdc_trainer_set_tensor(mdc_trainer* t, int which, int layer, const float* kernel_host, size_t kernel_elems, const float* bias_host,
                           size_t bias_elems, void* hip_stream) {
    return guarded("mdc_trainer_set_tensor", [&]() -> int {
        if (!t || !kernel_host || !bias_host) { set_error("mdc_trainer_set_tensor: null argument"); return MDC_EINVAL; }
        if (layer < 0 || layer >= t->nlayers) { set_error("mdc_trainer_set_tensor: layer %d out of range 0..%d", layer, t->nlayers - 1); return MDC_EINVAL; }
        float* vec = trainer_vec(t, which);
        if (!vec || which == MDC_TRAIN_GRADIENT) { set_error("mdc_trainer_set_tensor: `which` must be MDC_TRAIN_WEIGHTS, _ADAM_M or _ADAM_V"); return MDC_EINVAL; }
        if (kernel_elems != t->nk[layer] || bias_elems != t->nb[layer]) {
            set_error("mdc_trainer_set_tensor: layer %d expects kernel %zu / bias %zu elements, got %zu / %zu", layer, t->nk[layer], t->nb[layer],
                      kernel_elems, bias_elems);
            return MDC_EINVAL;
        }
        DeviceScope dev(t->device);
        if (!dev.ok) { set_error("mdc_trainer_set_tensor: cannot select device %d", t->device); return MDC_EIO; }
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        MDC_HIP(hipMemcpyAsync(vec + t->off_k[layer], kernel_host, kernel_elems * sizeof(float), hipMemcpyHostToDevice, s));
        MDC_HIP(hipMemcpyAsync(vec + t->off_b[layer], bias_host, bias_elems * sizeof(float), hipMemcpyHostToDevice, s));
        MDC_HIP(hipStreamSynchronize(s));      // the host buffers are the caller's again on return
        if (which == MDC_TRAIN_WEIGHTS) t->have[layer] = true;
        return MDC_OK;
    });
}

int mdc_trainer_get_tensor(mdc_trainer* t, int which, int layer, float* kernel_host, size_t kernel_elems, float* bias_host, size_t bias_elems,
                           void* hip_stream) {
    return guarded("mdc_trainer_get_tensor", [&]() -> int {
        if (!t || !kernel_host || !bias_host) { set_error("mdc_trainer_get_tensor: null argument"); return MDC_EINVAL; }
        if (layer < 0 || layer >= t->nlayers) { set_error("mdc_trainer_get_tensor: layer %d out of range 0..%d", layer, t->nlayers - 1); return MDC_EINVAL; }
        float* vec = trainer_vec(t, which);
        if (!vec) { set_error("mdc_trainer_get_tensor: unknown tensor kind %d", which); return MDC_EINVAL; }
        if (kernel_elems != t->nk[layer] || bias_elems != t->nb[layer]) {
            set_error("mdc_trainer_get_tensor: layer %d holds kernel %zu / bias %zu elements, asked for %zu / %zu", layer, t->nk[layer], t->nb[layer],
                      kernel_elems, bias_elems);
            return MDC_EINVAL;
        }
        DeviceScope dev(t->device);
        if (!dev.ok) { set_error("mdc_trainer_get_tensor: cannot select device %d", t->device); return MDC_EIO; }
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        MDC_HIP(hipMemcpyAsync(kernel_host, vec + t->off_k[layer], kernel_elems * sizeof(float), hipMemcpyDeviceToHost, s));
        MDC_HIP(hipMemcpyAsync(bias_host, vec + t->off_b[layer], bias_elems * sizeof(float), hipMemcpyDeviceToHost, s));
        MDC_HIP(hipStreamSynchronize(s));
        return MDC_OK;
    });
}

int mdc_trainer_set_iterations(mdc_trainer* t, int64_t iterations, void* hip_stream) {
    return guarded("mdc_trainer_set_iterations", [&]() -> int {
        if (!t || iterations < 0) { set_error("mdc_trainer_set_iterations: bad argument"); return MDC_EINVAL; }
        DeviceScope dev(t->device);
        if (!dev.ok) { set_error("mdc_trainer_set_iterations: cannot select device %d", t->device); return MDC_EIO; }
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        long long v = iterations;
        MDC_HIP(hipMemcpyAsync(&static_cast<TrainState*>(t->d_state)->iterations, &v, sizeof(v), hipMemcpyHostToDevice, s));
        MDC_HIP(hipStreamSynchronize(s));
        return MDC_OK;
    });
}

int mdc_train_batch(mdc_trainer* t, const float* x_dev, const float* y_dev, int64_t n_frames, const int32_t* order_dev, int64_t first,
                    int64_t count, int apply, void* hip_stream) {
    return guarded("mdc_train_batch", [&]() -> int {
        int rc = check_batch_args("mdc_train_batch", t, x_dev, y_dev, n_frames, order_dev, first, count);
        if (rc != MDC_OK) return rc;
        if (count == 0) return MDC_OK;
        DeviceScope dev(t->device);
        if (!dev.ok) { set_error("mdc_train_batch: cannot select device %d", t->device); return MDC_EIO; }
        return launch_batch(t, x_dev, y_dev, n_frames, order_dev, first, count, apply ? 2 : 1, static_cast<hipStream_t>(hip_stream));
    });
}

int mdc_trainer_evaluate(mdc_trainer* t, const float* x_dev, const float* y_dev, int64_t n_frames, const int32_t* order_dev, int64_t first,
                         int64_t count, void* hip_stream) {
    return guarded("mdc_trainer_evaluate", [&]() -> int {
        int rc = check_batch_args("mdc_trainer_evaluate", t, x_dev, y_dev, n_frames, order_dev, first, count);
        if (rc != MDC_OK) return rc;
        if (count == 0) return MDC_OK;
        DeviceScope dev(t->device);
        if (!dev.ok) { set_error("mdc_trainer_evaluate: cannot select device %d", t->device); return MDC_EIO; }
        return launch_batch(t, x_dev, y_dev, n_frames, order_dev, first, count, 0, static_cast<hipStream_t>(hip_stream));
    });
}

int mdc_trainer_read(mdc_trainer* t, int reset, double* train_loss_sum, int64_t* train_frames, double* eval_loss_sum, int64_t* eval_frames,
                     int64_t* iterations, void* hip_stream) {
    return guarded("mdc_trainer_read", [&]() -> int {
        if (!t) { set_error("null trainer"); return MDC_EINVAL; }
        DeviceScope dev(t->device);
        if (!dev.ok) { set_error("mdc_trainer_read: cannot select device %d", t->device); return MDC_EIO; }
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        TrainState h{};
        MDC_HIP(hipMemcpyAsync(&h, t->d_state, sizeof(h), hipMemcpyDeviceToHost, s));
        MDC_HIP(hipStreamSynchronize(s));
        if (train_loss_sum) *train_loss_sum = h.train_loss;
        if (train_frames) *train_frames = h.train_frames;
        if (eval_loss_sum) *eval_loss_sum = h.eval_loss;
        if (eval_frames) *eval_frames = h.eval_frames;
        if (iterations) *iterations = h.iterations;
        if (reset) {
            // zero the four accumulators behind `iterations` (which is optimizer state, not a statistic)
            MDC_HIP(hipMemsetAsync(reinterpret_cast<char*>(t->d_state) + offsetof(TrainState, train_frames), 0,
                                   sizeof(TrainState) - offsetof(TrainState, train_frames), s));
            MDC_HIP(hipStreamSynchronize(s));
        }
        if (h.bad_indices) {      // (the statistics above are those of the frames that WERE addressable)
            set_error("mdc_trainer_read: %u batch positions named frames outside [0, n_frames) -- skipped, not trained on; check order_dev", h.bad_indices);
            return MDC_EINVAL;
        }
        return MDC_OK;
    });
}

void mdc_trainer_destroy(mdc_trainer* t) {
    if (!t) return;
    {
        DeviceScope dev(t->device);
        for (void* p : {(void*)t->d_params, (void*)t->d_m, (void*)t->d_v, (void*)t->d_grad, (void*)t->d_partials, (void*)t->d_loss_partials, t->d_state, (void*)t->d_map})
            if (p) (void)hipFree(p);
    }
    delete t;
}

}  // extern "C"
