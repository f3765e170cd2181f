// Pieces of the dense_chain kernel (dense_chain.hip) that the fused epilogue of the bf16 dense1 GEMM
// (vtcnn2_bf16_dense1.hip) runs as well: the same instructions on the same operands, so the VT-CNN2 head gives the
// same bits whether it runs as its own launch (small batches, layer taps, the f32 mode) or inside dense1.
#pragma once
#include "mdc_internal.h"

#include <cmath>

namespace mdc {

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int kChainXld = 260;    // LDS row stride (floats) of a staged 256-float row: A-operand reads (row = lane&15,
                                  // k = 4i + (lane>>4)) fall on 64 different banks

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// butterfly over the 16 lanes of a DPP row: afterwards every lane holds op over the row
template <typename Op>
__device__ __forceinline__ float row_allreduce(float v, Op op) {
    v = op(v, dpp_mov<0xB1>(v));     // quad_perm [1,0,3,2]
    v = op(v, dpp_mov<0x4E>(v));     // quad_perm [2,3,0,1]
    v = op(v, dpp_mov<0x141>(v));    // row_half_mirror
    v = op(v, dpp_mov<0x140>(v));    // row_mirror
    return v;
}

// softmax + first-max argmax over the classes (the 16 lanes of a DPP row) of a 16-row tile: lane (class = fr), rows
// 4g + r hold the pre-softmax values z[r].  int(np.argmax(test_Y_hat[i,:])) (cnn.py:209): the FIRST index attaining the
// maximum of the PROBABILITIES as returned (slightly different logits can round to the same probability).
__device__ __forceinline__ void chain_softmax_store(const f32x4& z, int fr, int g, long row0, long n, int n_out,
                                                    float* __restrict__ probs, int* __restrict__ labels, float* __restrict__ tap_logits) {
    const bool cls = fr < n_out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = row0 + 4 * g + r;
        const float zz = cls ? z[r] : -INFINITY;
        const float mx = row_allreduce(zz, [](float a, float b) { return fmaxf(a, b); });
        const float e = cls ? expf(zz - mx) : 0.f;
        const float sum = row_allreduce(e, [](float a, float b) { return a + b; });
        const float pv = e / sum;
        const float pmx = row_allreduce(cls ? pv : -1.f, [](float a, float b) { return fmaxf(a, b); });
        const float cand = (cls && pv == pmx) ? (float)fr : 1e9f;
        const float arg = row_allreduce(cand, [](float a, float b) { return fminf(a, b); });
        if (row < n) {
            if (cls) {
                if (probs) probs[row * n_out + fr] = pv;
                if (tap_logits) tap_logits[row * n_out + fr] = z[r];
            }
            if (fr == 0 && labels) labels[row] = (int)arg;
        }
    }
}

}  // namespace

}  // namespace mdc
