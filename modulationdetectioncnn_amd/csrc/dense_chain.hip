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
constexpr int kYld = 36;          // LDS row stride of an inter-layer activation tile (<= 32 columns)
// The staged tile, laid out for PROGRESSIVE refill: four k-blocks (64 columns = 16 k-steps each); a block is four 1-KiB chunks,
// one per group of four rows, each written by ONE LDS-DMA instruction whose lane p = 4 i' + r4 carries the 16 bytes
// (row 4 rg + r4, columns 64 b + 4 i' .. + 3) -- exactly the four A-operand values of one k-step of one row.  Chunks are 1,088
// bytes apart: k-step i then reads its 16 pieces from all 8 four-bank slots, two each (the minimum for 64 lanes on 32 banks).
// Element (row fr, column k = 64 b + 4 i' + g) sits at float b kBlk + (fr >> 2) kChunk + (4 i' + (fr & 3)) 4 + g.
constexpr int kChunk = 272, kBlk = 4 * kChunk, kXsFloats = 4 * kBlk;      // 4,352 floats = 17,408 B per wave

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
// The same with the address split as the hardware takes it: a wave-uniform 64-bit base in scalar registers + a 32-bit per-lane
// byte offset -- no vector arithmetic per instruction (the 64-bit per-lane form costs two to eight VALU each, and a tile is sixteen).
__device__ __forceinline__ void chain_glds16_s(const void* sbase, unsigned lane_off, void* lds_wave_base) {
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sbase), "{m0}"(l) : "memory");
}

// Round 5.  Before: a wave loaded its 16 rows into registers, waited, wrote them to LDS, then ran layer 1 as ONE or two dependent
// MFMA chains (the K order is the result) -- 268 registers, so one wave per SIMD, nothing to fill the chain's latency or the
// load's: cnn.py's model streamed at 0.34 of the HBM peak (93 cycles per 32-cycle MFMA).  Now (profiles/r05_dense_chain_ab.log,
// r05_dense_chain2_ab.log; every step the same bits):
//  * the rows travel by LDS-DMA (no staging registers: TWO waves per SIMD whose chains interleave);
//  * the tile is staged as four k-blocks and a block is refilled with the wave's NEXT tile as soon as its last k-step has issued
//    (counted vmcnt per block), so a load runs under the rest of layer 1 as well as under the tail;
//  * layer 1's A operands are requested a pair of k-steps ahead of the MFMAs that use them (pinned: hipcc re-used two registers
//    and exposed the LDS latency 31 times per tile);
//  * the DMA addresses are a scalar base + a constant per-lane offset, and taps / ReLU flags are template parameters: the tile's
//    non-MFMA instructions share the SIMD's issue with the MFMAs (a no-load probe runs at 0.73 of the full kernel's time), so
//    every vector instruction taken out of the tile counts.
// 2.63e9 -> 3.4e9 -> 4.2e9 frames/s on one box (with the fused-DPP softmax reductions of dense_chain_common.h).  What is left:
// ~300 vector instructions of tail per tile beside 140 MFMAs.
// TAPS: the instantiation that also writes the layer taps (CNN.ipynb cell 17's sub-models); the batch path's has none of their
// address arithmetic and branches in its tail.  The three-layer chain is cnn.py's net: ReLU after layers 1 and 2 (cnn.py:108-110).
template <int T1, int NL, bool TAPS>
__global__ __launch_bounds__(256) void dense_chain_kernel(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the tile walk below is wave-uniform)
    const int fr = lane & 15, g = lane >> 4;
    float* xs = smem + wv * (kXsFloats + 16 * kYld);
    float* ys = xs + kXsFloats;
    // one k-block of a tile: four LDS-DMA instructions (rows past the end of the batch: the last row again, computed, never stored)
    const unsigned lane_off = (unsigned)((lane & 3) * kK0 + 4 * (lane >> 2)) * 4u;      // bytes: row r4, columns 4 i' .. + 3
    auto stage_block = [&](long tile, int b) {
        const long r0 = tile << 4;
        const int ip = lane >> 2, r4 = lane & 3;
        if (r0 + 16 <= p.n) {      // a whole tile (all but possibly the last): scalar base + the lane's constant offset
            const float* tb = p.x + r0 * kK0 + 64 * b;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) chain_glds16_s(tb + rg * 4 * kK0, lane_off, xs + b * kBlk + rg * kChunk);
        } else {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const long row = r0 + 4 * rg + r4 < p.n ? r0 + 4 * rg + r4 : p.n - 1;
                chain_glds16(p.x + row * kK0 + 64 * b + 4 * ip, xs + b * kBlk + rg * kChunk);
            }
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
    if (tile < ntiles) {
#pragma unroll
        for (int b = 0; b < 4; ++b) stage_block(tile, b);
    }
    const float* xa = xs + (fr >> 2) * kChunk + (fr & 3) * 4 + g;      // this lane's A-operand column: + b kBlk + 16 i'
    for (; tile < ntiles; tile += nwaves) {
        const long row0 = tile << 4;
        const bool has_next = tile + nwaves < ntiles;

        // ---- layer 1: K = 256, block by block.  A block is read as soon as ITS four DMAs have landed and refilled with the wave's
        // next tile as soon as its last k-step has issued, so a tile's load runs under the rest of this tile's layer 1 as well as
        // under the tail -- a buffer no longer waits out a whole memory latency between being read and being readable again.
        // Counted waits: the DMAs of this tile are the 16 oldest loads; behind them at most the 4 b DMAs already issued for the next
        // tile.  "At most 12 outstanding" therefore means this tile's first 4 (b + 1) have landed, whatever the previous tile's
        // output STORES (which share the counter and may retire in any order against loads) are doing: they can only make the wait
        // longer, never shorter than needed.  The wave's last tile has nothing behind it: 12 - 4 b.
        f32x4 acc[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#define MDC_CHAIN_BLOCK(B)                                                                                                        \
        {                                                                                                                          \
            if (has_next) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                        \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 - 4 * (B)) : "memory");                                               \
            const float* xb = xa + (B) * kBlk;                                                                                     \
            /* the NEXT pair of k-steps is requested before this pair's MFMAs issue (pinned: left alone, hipcc reads each pair  */ \
            /* into the same two registers right after the MFMAs that consumed the last one and waits out the LDS latency)      */ \
            float a_cur[2] = {xb[0], xb[16]}, a_nxt[2] = {0.f, 0.f};                                                              \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                                     \
                if (jj + 1 < 8) { a_nxt[0] = xb[32 * (jj + 1)]; a_nxt[1] = xb[32 * (jj + 1) + 16]; }                              \
                __builtin_amdgcn_sched_barrier(0);                                                                                 \
                _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                      \
                    _Pragma("unroll") for (int t = 0; t < T1; ++t)                                                                 \
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u], w1[t][16 * (B) + 2 * jj + u], acc[t], 0, 0, 0);    \
                __builtin_amdgcn_sched_barrier(0);                                                                                 \
                a_cur[0] = a_nxt[0];                                                                                               \
                a_cur[1] = a_nxt[1];                                                                                               \
            }                                                                                                                      \
            /* every read of this block has fed an MFMA that has issued: the block is free */                                     \
            if (has_next) stage_block(tile + nwaves, (B));                                                                         \
        }
        MDC_CHAIN_BLOCK(0)
        MDC_CHAIN_BLOCK(1)
        MDC_CHAIN_BLOCK(2)
        MDC_CHAIN_BLOCK(3)
#undef MDC_CHAIN_BLOCK
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
                    h = fmaxf(h, 0.f);
                    ys[(4 * g + r) * kYld + t * 16 + fr] = h;
                    if (TAPS && p.tap_h1 && t * 16 + fr < p.n1 && row0 + 4 * g + r < p.n) p.tap_h1[(row0 + 4 * g + r) * p.n1 + t * 16 + fr] = h;
                }
            // ---- layer 2: K <= 32 ----
            f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ys[fr * kYld + 4 * i + g], w2[i], a2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h = a2[r] + b2;
                h = fmaxf(h, 0.f);
                a2[r] = h;
                if (TAPS && p.tap_h2 && fr < p.n2 && row0 + 4 * g + r < p.n) p.tap_h2[(row0 + 4 * g + r) * p.n2 + fr] = h;
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
        chain_softmax_store(z, fr, g, row0, p.n, p.n_out, p.probs, p.labels, TAPS ? p.tap_logits : nullptr);
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
    const size_t lds = (size_t)4 * (kXsFloats + 16 * kYld) * sizeof(float);    // 78,848 B: two work-groups per CU still fit
    if (nl == 3 && !(relu1 && relu2)) { set_error("dense_chain: the three-layer chain is cnn.py's (ReLU after layers 1 and 2)"); return MDC_ENOTSUP; }
    const bool taps = tap_logits || tap_h1 || tap_h2;
    const void* fn = nullptr;
    if (t1 == 1 && nl == 1) fn = taps ? reinterpret_cast<const void*>(dense_chain_kernel<1, 1, true>) : reinterpret_cast<const void*>(dense_chain_kernel<1, 1, false>);
    else if (t1 == 2 && nl == 3) fn = taps ? reinterpret_cast<const void*>(dense_chain_kernel<2, 3, true>) : reinterpret_cast<const void*>(dense_chain_kernel<2, 3, false>);
    else { set_error("dense_chain: unsupported shape (t1=%d, layers=%d)", t1, nl); return MDC_ENOTSUP; }
    MDC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {&p};
    MDC_HIP(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(256), args, lds, s));
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
