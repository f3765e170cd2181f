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

// max / min over the 16 lanes of a DPP row with the row operand folded INTO the instruction (v_max_f32_dpp): hipcc fuses the add
// butterfly above but emits v_mov_b32_dpp + v_max_f32 for these -- four instructions more per reduction, and the softmax below is
// three of them per row in tails where every vector instruction competes with the MFMAs for the SIMD's issue.  Same values.
// (s_nop 1: a DPP read of a VGPR needs two wait states behind the vector instruction that wrote it, and hipcc pads that hazard
// for its own instructions only)
#define MDC_DPP_STEP(OP, CTRL) asm("s_nop 1\n\tv_" OP "_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(v)); v = t;
__device__ __forceinline__ float row_allmax(float v) {
    float t;
    MDC_DPP_STEP("max", "quad_perm:[1,0,3,2]") MDC_DPP_STEP("max", "quad_perm:[2,3,0,1]") MDC_DPP_STEP("max", "row_half_mirror") MDC_DPP_STEP("max", "row_mirror")
    return v;
}
__device__ __forceinline__ float row_allmin(float v) {
    float t;
    MDC_DPP_STEP("min", "quad_perm:[1,0,3,2]") MDC_DPP_STEP("min", "quad_perm:[2,3,0,1]") MDC_DPP_STEP("min", "row_half_mirror") MDC_DPP_STEP("min", "row_mirror")
    return v;
}
#undef MDC_DPP_STEP

// One dependent MFMA chain over a staged row of 256 floats: acc += A[row][k] B[k][col], k-steps 0 .. 63 in order (the order is the
// result).  xrow = &image[row * ld + g]; the A operands are requested from LDS a pair of k-steps AHEAD of the MFMAs that use them,
// pinned with sched_barrier: left alone, hipcc reads each pair into the same two registers right after the MFMAs that consumed the
// last one and waits out the whole LDS latency in front of the next two (dense_chain.hip; profiles/r05_dense_chain2_ab.log).
__device__ __forceinline__ f32x4 chain_k256(const float* xrow, const float (&w)[64], f32x4 acc) {
    float a_cur[2] = {xrow[0], xrow[4]}, a_nxt[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j + 1 < 32) { a_nxt[0] = xrow[8 * (j + 1)]; a_nxt[1] = xrow[8 * (j + 1) + 4]; }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0], w[2 * j], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1], w[2 * j + 1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
    }
    return acc;
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
        const float mx = row_allmax(zz);
        const float e = cls ? expf(zz - mx) : 0.f;
        const float sum = row_allreduce(e, [](float a, float b) { return a + b; });
        const float pv = e / sum;
        const float pmx = row_allmax(cls ? pv : -1.f);
        const float cand = (cls && pv == pmx) ? (float)fr : 1e9f;
        const float arg = row_allmin(cand);
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
