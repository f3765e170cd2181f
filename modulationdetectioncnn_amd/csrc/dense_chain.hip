// dense_chain: rows of 256 f32 -> [Dense(N1) (ReLU?)] -> [Dense(N2) (ReLU?)] -> [Dense(N3)] -> softmax ->
// first-max argmax, on 16-row tiles with v_mfma_f32_16x16x4_f32 (exact f32 fma chains).
//
// Two users:
//  * the VT-CNN2 head (one layer, 256 -> C): Dense(C) + softmax + np.argmax of
//    RML2016.10a_VTCNN2_example.ipynb:241-243 / cnn.py:209;
//  * the literal cnn.py model (cnn.py:104-115): under TensorFlow's channels_last
//    `Reshape([1,2,128])` is H=1, W=2, C=128, so pad(0,1)+Conv2D(F,(1,2)) is a LINEAR map of the 256
//    input floats to 3F outputs.  It is folded into a dense 256 x 3F matrix at pack time, followed by
//    ReLU, Dense(D, relu), Dense(C), softmax: three layers of this kernel, input = the raw frame.
//
// One wave = one 16-row tile at a time.  The 16 KiB of input rows are loaded with coalesced 16-B
// loads, staged in LDS, and read back as MFMA A operands (A[row][k]); every layer's weights sit in
// registers as B operands (B[k][col]); a layer's 16x16 result (col on the lane) goes through a
// 2 KiB LDS transpose to become the next layer's A operand.  HBM-bound: 1 KiB read per row.
#include "dense_chain_common.h"

namespace mdc {

namespace {

constexpr int kK0 = 256;          // input width (floats per row)
constexpr int kXld = kChainXld;   // LDS row stride of the staged input (16-B aligned rows)
constexpr int kYld = 36;          // LDS row stride of an inter-layer activation tile (<= 32 columns)

struct ChainParams {
    const float* x;        // [n][256]
    long n;
    const float* w;        // packed: layer 1 [T1][64 ksteps][64 lanes]; layer 2 [8][64]; layer 3 [8][64]; biases [3][32]
    int nl;                // number of layers (1 or 3)
    int n_out;             // classes C (columns of the last layer)
    int relu1, relu2;      // ReLU after layer 1 / 2
    float* probs;
    int* labels;
    float* tap_logits;     // last layer pre-softmax (or NULL)
    float* tap_h1;         // layer-1 output after its activation, [n][n1] (or NULL)
    float* tap_h2;         // layer-2 output after its activation, [n][n2] (or NULL)
    int n1, n2;            // real widths of layers 1, 2 (for the taps)
};

// 16 B per lane from global memory straight into LDS (destination = wave-uniform base + 16 lane), issued from asm: hipcc puts
// an s_waitcnt vmcnt(0) in front of every LDS read that follows a BUILTIN LDS-DMA (vtcnn2_bf16_common.h, glds16_async), which
// would drain the prefetch below at once.  The kernel orders both ways itself: vmcnt(0) at the end of a tile before the next one
// reads the buffer, lgkmcnt(0) before a buffer is refilled.
__device__ __forceinline__ void chain_glds16(const void* gsrc, void* lds_wave_base) {
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gsrc), "{m0}"(l) : "memory");
}

// Round 5.  Before: a wave loaded its 16 rows into registers, waited, wrote them to LDS, then ran layer 1 as ONE or two dependent
// MFMA chains (the K order is the result) -- 268 registers, so one wave per SIMD, nothing to fill the chain's latency or the
// load's: cnn.py's model streamed at 0.34 of the HBM peak (93 cycles per 32-cycle MFMA).  Now the rows travel by LDS-DMA (no
// staging registers: 219, TWO waves per SIMD whose chains interleave), and a wave refills its buffer with the NEXT tile as soon as
// layer 1 has read it, under layers 2 - 3, the softmax and the stores.  A second LDS buffer per wave instead (the whole tile ahead)
// costs the second wave per SIMD and bought 12 %; this form 29 %: 3.4e9 frames/s = 0.44 of the HBM peak, same bits
// (profiles/r05_dense_chain_ab.log).  What is left is the chain itself: two waves x two dependent MFMA chains per SIMD.
// Rows past the end of the batch: the last row again (a valid address); they are computed and never stored.
template <int T1, int NL>
__global__ __launch_bounds__(256) void dense_chain_kernel(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    float* xs = smem + wv * (16 * kXld + 16 * kYld);
    float* ys = xs + 16 * kXld;
    auto stage = [&](long tile) {
        const long r0 = tile << 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = r0 + r < p.n ? r0 + r : p.n - 1;
            chain_glds16(p.x + row * kK0 + 4 * lane, xs + r * kXld);
        }
    };

    // ---- weights into registers (B operands: lane (col = fr, k-group g) holds W[4i + g][col]) ----
    float w1[T1][64];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int i = 0; i < 64; ++i) w1[t][i] = p.w[(t * 64 + i) * 64 + lane];
    const float* wp2 = p.w + T1 * 64 * 64;
    float w2[8], w3[8];
    if (NL == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { w2[i] = wp2[i * 64 + lane]; w3[i] = wp2[(8 + i) * 64 + lane]; }
    }
    const float* bias = wp2 + 16 * 64;
    float b1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) b1[t] = bias[t * 16 + fr];
    const float b2 = bias[32 + fr], b3 = bias[64 + fr];

    const long ntiles = (p.n + 15) >> 4;
    const long nwaves = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wv;
    if (tile < ntiles) stage(tile);
    for (; tile < ntiles; tile += nwaves) {
        const long row0 = tile << 4;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile has landed (issued in the prologue / under the previous tile's tail)

        // ---- layer 1: K = 256 ----
        f32x4 acc[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float a = xs[fr * kXld + 4 * i + g];
#pragma unroll
            for (int t = 0; t < T1; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1[t][i], acc[t], 0, 0, 0);
        }
        // every read of xs has fed an MFMA above; say so, then refill the buffer with the wave's next tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tile + nwaves < ntiles) stage(tile + nwaves);
        f32x4 z;      // final-layer pre-softmax values: lane (class = fr), rows 4g + r
        if (NL == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = acc[0][r] + b1[0];
        } else {
            // activation, transpose through LDS: ys[row][col]
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float h = acc[t][r] + b1[t];
                    if (p.relu1) h = fmaxf(h, 0.f);
                    ys[(4 * g + r) * kYld + t * 16 + fr] = h;
                    if (p.tap_h1 && t * 16 + fr < p.n1 && row0 + 4 * g + r < p.n) p.tap_h1[(row0 + 4 * g + r) * p.n1 + t * 16 + fr] = h;
                }
            // ---- layer 2: K <= 32 ----
            f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kYld + 4 * i + g], w2[i], a2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h = a2[r] + b2;
                if (p.relu2) h = fmaxf(h, 0.f);
                a2[r] = h;
                if (p.tap_h2 && fr < p.n2 && row0 + 4 * g + r < p.n) p.tap_h2[(row0 + 4 * g + r) * p.n2 + fr] = h;
            }
            // the layer-3 A operand reads ys after every lane's layer-2 reads are done (same wave, in order)
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[(4 * g + r) * kYld + fr] = a2[r];
            // ---- layer 3: K <= 16 ----
            f32x4 a3 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kYld + 4 * i + g], w3[i], a3, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = a3[r] + b3;
        }
        chain_softmax_store(z, fr, g, row0, p.n, p.n_out, p.probs, p.labels, p.tap_logits);
    }
}

}  // namespace

// Host-side packing of one layer's (in, out) kernel into B-operand order [kstep][lane]:
// lane (col = lane&15, g = lane>>4) of k-step i holds W[4i + g][tile*16 + col] (0 outside the matrix).
void chain_pack_layer(std::vector<float>& dst, const float* w, int k_in, int n_out, int ksteps, int tiles) {
    for (int t = 0; t < tiles; ++t)
        for (int i = 0; i < ksteps; ++i)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = 4 * i + (lane >> 4), c = t * 16 + (lane & 15);
                dst.push_back((k < k_in && c < n_out) ? w[(size_t)k * n_out + c] : 0.f);
            }
}

int chain_launch(int t1, int nl, const float* x, long n, const float* wpack, int n_out, int relu1, int relu2,
                 float* probs, int* labels, float* tap_logits, float* tap_h1, float* tap_h2, int n1, int n2, hipStream_t s) {
    ChainParams p{x, n, wpack, nl, n_out, relu1, relu2, probs, labels, tap_logits, tap_h1, tap_h2, n1, n2};
    const long ntiles = (n + 15) / 16;
    long grid = (ntiles + 3) / 4;
    if (grid > 512) grid = 512;      // persistent: two work-groups per CU (LDS and registers allow exactly that), each wave walking its tiles
    const size_t lds = (size_t)4 * (16 * kXld + 16 * kYld) * sizeof(float);    // 75,776 B
    const void* fn = nullptr;
    if (t1 == 1 && nl == 1) fn = reinterpret_cast<const void*>(dense_chain_kernel<1, 1>);
    else if (t1 == 2 && nl == 3) fn = reinterpret_cast<const void*>(dense_chain_kernel<2, 3>);
    else { set_error("dense_chain: unsupported shape (t1=%d, layers=%d)", t1, nl); return MDC_ENOTSUP; }
    MDC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (t1 == 1) hipLaunchKernelGGL((dense_chain_kernel<1, 1>), dim3((unsigned)grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((dense_chain_kernel<2, 3>), dim3((unsigned)grid), dim3(256), lds, s, p);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
