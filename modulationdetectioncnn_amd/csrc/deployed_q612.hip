// Q6.12 integer forward of the deployed nets (T1/T2): the arithmetic of the reference's FPGA datapath
// (cnn_test_latest1.sv) on the GPU, so that ROM tables and test vectors written with the `float2fix` exporter can
// be validated against the float path and against each other at batch scale.
//
// Rules followed (citations into /root/reference/cnn_test_latest1.sv):
//   operands: 18-bit two's-complement Q6.12;  quantisation of floats = float2fix (CNN.ipynb cell 23): trunc(v*4096)
//   conv neuron  signed_mult1 (sv:642-658): m = a*b + c*d (36 bit); out = {m[35], m[28:12]}; out+bias wraps to 18 bit;
//                ReLU = 0 when bit 17 is set
//   dense term   signed_mult  (sv:664-675): the same bit selection of I_act*W_i + Q_act*W_q, sign-extended into a
//                32-bit accumulator that starts at the sign-extended bias (sv:293-343); ReLU on bit 31 (sv:171-176)
// Activation/weight ORDER is Keras' (the RTL's clocking and index reversal are not modelled).
//
// Mapping: one wave per frame, lane = conv position (w = lane, lane+64, and 128 on lane 0).  32-bit wrap-around
// addition is associative, so the per-lane partial sums are combined with a butterfly and the result is
// bit-identical to the FPGA's sequential accumulation.  HBM: 1 KiB in, 4*C+4 B out per frame; the weight tables
// (int32, [class][filter][129] for I and Q) stay in L1/L2.
#include "mdc_internal.h"

#include <cmath>

namespace mdc {

namespace {

constexpr int kMaxClasses = 8;

__device__ __forceinline__ int wrap18(int v) { return (v << 14) >> 14; }
__device__ __forceinline__ int select18(long long m) {          // {m[35], m[28:12]} as a signed 18-bit value
    // mult_out is a 36-bit wire (sv:650, 671): the sign is BIT 35 of the sum, not the sign of the unwrapped value --
    // they differ exactly when a*b + c*d = +2^35 (all four operands -2^17), which wraps to -2^35
    return (int)((m >> 12) & 0x1FFFF) - (int)(((m >> 35) & 1) << 17);
}
__device__ __forceinline__ int quant(float v) {                  // float2fix: truncate toward zero, wrap to 18 bits
    return wrap18((int)truncf(v * 4096.f));
}

struct QParams {
    const void* x;          // (n,2,128) f32 (quantised on load) or int32 Q6.12
    int x_is_q;
    long n;
    const int* tab;         // [3F conv: k0[F], k1[F], b[F]] [C dense bias] pad to 64; then Wi[C][F][129], Wq[C][F][129]
    int C;
    int* dense;             // (n,C) or NULL
    int* labels;            // (n) or NULL
};

template <int F>
__global__ __launch_bounds__(256) void deployed_q612_kernel(QParams p) {
    const int lane = threadIdx.x & 63;
    const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= p.n) return;      // whole wave
    const int* k0 = p.tab;
    const int* k1 = p.tab + F;
    const int* cb = p.tab + 2 * F;
    const int* db = p.tab + 3 * F;
    const int* wi = p.tab + 64;
    const int* wq = wi + p.C * F * 129;
    unsigned acc[kMaxClasses];
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) acc[c] = 0u;
    auto sample = [&](int h, int s) -> int {      // x[h][s], zero outside 0..127
        if (s < 0 || s > 127) return 0;
        const long idx = frame * kFrameFloats + h * kSamples + s;
        return p.x_is_q ? wrap18(static_cast<const int*>(p.x)[idx]) : quant(static_cast<const float*>(p.x)[idx]);
    };
    for (int it = 0; it < 3; ++it) {
        const int w = it < 2 ? lane + 64 * it : 128;
        if (it == 2 && lane != 0) break;
        const int xi0 = sample(0, w - 1), xi1 = sample(0, w), xq0 = sample(1, w - 1), xq1 = sample(1, w);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const long long a = k0[f], b = k1[f];
            int ai = wrap18(select18(xi0 * a + xi1 * b) + cb[f]);
            int aq = wrap18(select18(xq0 * a + xq1 * b) + cb[f]);
            ai = ai < 0 ? 0 : ai;
            aq = aq < 0 ? 0 : aq;
            for (int c = 0; c < p.C; ++c) {
                const long long m = (long long)ai * wi[(c * F + f) * 129 + w] + (long long)aq * wq[(c * F + f) * 129 + w];
                acc[c] += (unsigned)select18(m);       // sign-extended 18-bit term into the 32-bit sum (wraps)
            }
        }
    }
    // wave sum (wrap-around adds commute), then bias, ReLU, first-max label
    int best = 0, bestv = 0;
    for (int c = 0; c < p.C; ++c) {
        unsigned v = acc[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += (unsigned)__shfl_xor((int)v, off);
        int s = (int)(v + (unsigned)db[c]);
        s = s < 0 ? 0 : s;
        if (lane == 0 && p.dense) p.dense[frame * p.C + c] = s;
        if (c == 0 || s > bestv) { bestv = s; best = c; }      // strict >: the first maximum wins (cnn.py:209)
    }
    if (lane == 0 && p.labels) p.labels[frame] = best;
}

inline int host_quant(float v) {
    long long q = (long long)std::trunc((double)v * 4096.0);
    q = ((q + (1 << 17)) & ((1 << 18) - 1)) - (1 << 17);
    return (int)q;
}

}  // namespace

// d_pack slot 1 of a deployed model: the integer tables (built at finalize from the float weights, which are exact
// multiples of 2^-12 when they came from a .txt table)
int deployed_q612_pack(mdc_model* m) {
    const int F = m->topo.filters, C = m->topo.classes;
    if (C > kMaxClasses) return MDC_OK;      // no integer path for such a net; mdc_forward_q612 reports it
    std::vector<int> tab(64 + (size_t)2 * C * F * 129, 0);
    const float* ck = m->hk[0].data();   // HWIO (1,2,1,F)
    for (int f = 0; f < F; ++f) {
        tab[f] = host_quant(ck[f]);
        tab[F + f] = host_quant(ck[F + f]);
        tab[2 * F + f] = host_quant(m->hb[0][f]);
    }
    for (int c = 0; c < C; ++c) tab[3 * F + c] = host_quant(m->hb[1][c]);
    const float* dk = m->hk[1].data();   // (258F, C), rows h*129F + w*F + f
    for (int h = 0; h < 2; ++h)
        for (int c = 0; c < C; ++c)
            for (int f = 0; f < F; ++f)
                for (int w = 0; w < 129; ++w)
                    tab[64 + (size_t)h * C * F * 129 + ((size_t)c * F + f) * 129 + w] = host_quant(dk[((size_t)h * 129 * F + (size_t)w * F + f) * C + c]);
    return upload(m, 1, tab.data(), tab.size() * sizeof(int));
}

int deployed_q612_forward(const mdc_model* m, const void* x, int x_is_q, int64_t n, int32_t* dense, int32_t* labels, hipStream_t s) {
    const int F = m->topo.filters, C = m->topo.classes;
    if (C > kMaxClasses || 3 * F + C > 64) { set_error("Q6.12 path supports at most %d classes (got %d)", kMaxClasses, C); return MDC_ENOTSUP; }
    if (!m->d_pack[1]) { set_error("Q6.12 tables missing"); return MDC_ESTATE; }
    if (n == 0) return MDC_OK;
    QParams p{x, x_is_q, (long)n, static_cast<const int*>(m->d_pack[1]), C, dense, labels};
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (F == 3) hipLaunchKernelGGL(deployed_q612_kernel<3>, dim3(grid), dim3(256), 0, s, p);
    else if (F == 10) hipLaunchKernelGGL(deployed_q612_kernel<10>, dim3(grid), dim3(256), 0, s, p);
    else { set_error("Q6.12 path is built for F = 3 and F = 10 (got %d)", F); return MDC_ENOTSUP; }
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
