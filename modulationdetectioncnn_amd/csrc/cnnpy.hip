// The literal cnn.py model (cnn.py:104-115; SURVEY.md 8(a) A0, "T4") on the dense_chain kernel.
//
// With the TensorFlow backend `Reshape([1,2,128])` is (H=1, W=2, C=128): the 128 time samples are
// CHANNELS and the I/Q rows are the width.  ZeroPadding2D((0,1)) -> W=4 = [0, I, Q, 0];
// Conv2D(F,(1,2)) HWIO (1,2,128,F) -> (1,3,F):
//     y[0][f] = b[f] + K[1][:,f].I          y[1][f] = b[f] + K[0][:,f].I + K[1][:,f].Q
//     y[2][f] = b[f] + K[0][:,f].Q
// i.e. a linear map of the frame's 256 floats to 3F values, folded here into a 256 x 3F matrix
// (row = w'*128 + c, column = wo*F + f, entry K[w'+1-wo][c][f] when that tap index is 0 or 1).
// Then ReLU, Flatten (w,f), Dense(D, relu), Dense(C), softmax.
#include "mdc_internal.h"

namespace mdc {

void chain_pack_layer(std::vector<float>& dst, const float* w, int k_in, int n_out, int ksteps, int tiles);
int chain_launch(int t1, int nl, const float* x, long n, const float* wpack, int n_out, int relu1, int relu2,
                 float* probs, int* labels, float* tap_logits, float* tap_h1, float* tap_h2, int n1, int n2, hipStream_t s);

int cnnpy_pack(mdc_model* m) {
    const int F = m->topo.filters, D = m->topo.hidden, C = m->topo.classes;
    if (3 * F > 32 || D > 16 || C > 16) {
        set_error("cnnpy: the HIP path covers 3*filters <= 32, hidden <= 16, classes <= 16 (got %d, %d, %d)", F, D, C);
        return MDC_ENOTSUP;
    }
    const float* ck = m->hk[0].data();       // HWIO (1,2,128,F): [kw][c][f]
    std::vector<float> fold((size_t)256 * 3 * F, 0.f);
    for (int wp = 0; wp < 2; ++wp)
        for (int c = 0; c < 128; ++c)
            for (int wo = 0; wo < 3; ++wo) {
                const int kw = wp + 1 - wo;
                if (kw < 0 || kw > 1) continue;
                for (int f = 0; f < F; ++f) fold[(size_t)(wp * 128 + c) * 3 * F + wo * F + f] = ck[((size_t)kw * 128 + c) * F + f];
            }
    std::vector<float> pk;
    chain_pack_layer(pk, fold.data(), 256, 3 * F, 64, 2);
    chain_pack_layer(pk, m->hk[1].data(), 3 * F, D, 8, 1);
    chain_pack_layer(pk, m->hk[2].data(), D, C, 8, 1);
    std::vector<float> bias(96, 0.f);
    for (int wo = 0; wo < 3; ++wo)
        for (int f = 0; f < F; ++f) bias[wo * F + f] = m->hb[0][f];
    for (int d = 0; d < D; ++d) bias[32 + d] = m->hb[1][d];
    for (int c = 0; c < C; ++c) bias[64 + c] = m->hb[2][c];
    pk.insert(pk.end(), bias.begin(), bias.end());
    return upload(m, 0, pk.data(), pk.size() * sizeof(float));
}

int cnnpy_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                  float* tap, int tap_kind, hipStream_t s) {
    const int F = m->topo.filters, D = m->topo.hidden, C = m->topo.classes;
    ProfScope ps(m, 0, s);
    return chain_launch(2, 3, x, (long)n, static_cast<const float*>(m->d_pack[0]), C, 1, 1, probs, labels,
                        tap_kind == MDC_TAP_DENSE ? tap : nullptr,
                        (tap_kind == MDC_TAP_CONV || tap_kind == MDC_TAP_FLAT) ? tap : nullptr,
                        tap_kind == MDC_TAP_HIDDEN ? tap : nullptr, 3 * F, D, s);
}

}  // namespace mdc
