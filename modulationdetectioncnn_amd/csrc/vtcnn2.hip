// Canonical VT-CNN2 (T3): host-side packing, f32 kernels, the softmax head and the
// per-call dispatcher.  The bf16 kernels live in vtcnn2_bf16.hip.
//
// Math restated from examples-master/.../RML2016.10a_VTCNN2_example.ipynb:229-243
// (shapes :190-210), SURVEY.md 8(a) A2:
//   xp   = pad2(x)                                  (2,132)
//   y1   = relu(b1[c] + sum_t K1[c,t] xp[h,v+t])    (256,2,130)   v = 0..129
//   y1p  = pad2(y1)                                 (256,2,134)
//   y2   = relu(b2[o] + sum_{c,h,j} K2[o,c,h,j] y1p[c,h,w+j])   (80,132)   w = 0..131
//   hid  = relu(W1^T flat(y2) + c1)                 (256)         flat index o*132 + w
//   p    = softmax(W2^T hid + c2)                   (C)
//
// "Lane = frame" mapping shared by both dtypes: an MFMA's 16-wide N dimension is 16
// FRAMES at one time position.  A conv1 MFMA then yields X[channel][frame] for one padded
// position w'; in the 16x16 C/D layout a lane holds channels 4g..4g+3 of its frame, which is
// exactly a B operand of the next MFMA (k = channel), so conv1's output feeds conv2 from
// registers with no LDS round trip and no im2col buffer.  The three conv2 taps of one X are
// three MFMAs into the accumulators of output positions w', w'-1, w'-2: the tap shift is a
// choice of accumulator register, never a data movement.  Zero padding of y1p is free
// (those positions are simply skipped).
//
// Intermediate layout in HBM (workspace): feat[frame][w][o]  (o fastest, 80 per position).
// dense1's weight rows are permuted to that order at pack time, so the reference's
// channels_first Flatten is never materialised (only the 'flat'/'conv' taps un-permute).
#include "dense_chain_common.h"      // (mdc_internal.h, f32x4, the head's softmax + argmax epilogue)

#include <algorithm>
#include <cmath>

namespace mdc {

namespace {

// async global -> LDS copy of 16 B per lane: LDS destination = wave-uniform base + lane*16.  Issued from asm (round 5): hipcc
// tracks the BUILTIN form as an LDS store it cannot tell apart from the buffer being read, and put an s_waitcnt vmcnt(0) right
// behind the prefetch of the next weight chunk, in front of the first LDS read of the current one (ISA of rounds 1-4: the
// "double buffer" drained at the start of every chunk).  The kernel's own waits order both ways: vmcnt(0) in front of the barrier
// that ends a chunk (the next chunk has landed before anyone reads it), and that same barrier before a buffer is refilled.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(g), "{m0}"(l) : "memory");
}
__device__ __forceinline__ void glds_drain_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ------------------------------------------------------------------------------------
// f32 conv1+conv2: v_mfma_f32_16x16x4_f32 (exact f32 fma chains).
// Workgroup = 4 waves, each wave owns 16 frames (no cross-wave reduction).  Loop nest:
// position block (11 outputs) x 16-channel chunk (conv2 weights of the chunk, 30 KB, staged
// in LDS and shared by the 4 waves) x 13 padded positions x 2 rows x [1 conv1 + 60 conv2 MFMAs].
// ------------------------------------------------------------------------------------
constexpr int kChunk = 16;             // channels per chunk
constexpr int kNChunk = kC1 / kChunk;  // 16
constexpr int kWChunkFloats = 2 * 3 * 5 * 4 * 64;   // [h][j][ot][r][lane] = 7680
constexpr int kXld = 133;              // LDS row stride of a padded input row (132 + 1)
constexpr int kXinFloats = 4 * 2 * 16 * kXld;
constexpr size_t kConvF32Lds = (2 * kWChunkFloats + kXinFloats) * sizeof(float);

// U8 = true (mdc_forward_iq_u8): x points at raw interleaved uint8 (I,Q) pairs, window f at byte f*hop2; a lane loads
// the 8 bytes holding its four samples of both rows and converts its row with iq_u8_kernel's arithmetic (eval_ops.hip).
// KP = output positions per block (register blocking: 5*KP accumulator tiles), YSPLIT = one position block per
// work-group (blockIdx.y) instead of a loop over the 132/KP blocks.  Every form runs the SAME instruction sequence per
// output element (chunk order, tap order, fma chain), so the results are bit-identical; what changes is how many
// work-groups share a frame group.  A single window (the reference classifies one window per start pulse,
// cnn_test_latest1.sv:144-209) with KP = 11 and no split streams all of conv2 through ONE CU (4.8 ms); KP = 1 with
// the split spreads its 132 positions over 132 CUs.
template <bool U8, int KP, bool YSPLIT>
__global__ __launch_bounds__(256, 1) void vt_conv_f32_kernel(const float* __restrict__ x, long n,
                                                             const float* __restrict__ wpack,   // [16 chunks][7680]
                                                             const float* __restrict__ a1pack,  // [16 chunks][64 lanes]
                                                             const float* __restrict__ b2,      // [80]
                                                             float* __restrict__ feat,          // [n][132][80]
                                                             long hop2, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wbuf = smem;
    float* xin = smem + 2 * kWChunkFloats;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const long frame0 = (long)blockIdx.x * 64 + q * 16;

    // ---- stage this wave's 16 frames, zero-padded by 2 on both sides of each row ----
    float* xw = xin + q * (2 * 16 * kXld);
    {
        const int h = lane >> 5, s = (lane & 31) * 4;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const long f = frame0 + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (U8) {
                if (f < n) {
                    const uint2 r = load8_unaligned(reinterpret_cast<const unsigned char*>(x) + f * hop2 + (lane & 31) * 8);
                    const unsigned a = r.x >> (8 * h), b = r.y >> (8 * h);      // row 0 = I = even bytes, row 1 = Q = odd bytes
                    v = make_float4(((float)(a & 0xFFu) - 127.5f) * scale, ((float)((a >> 16) & 0xFFu) - 127.5f) * scale,
                                    ((float)(b & 0xFFu) - 127.5f) * scale, ((float)((b >> 16) & 0xFFu) - 127.5f) * scale);
                }
            } else {
                if (f < n) v = reinterpret_cast<const float4*>(x + f * kFrameFloats)[lane];
            }
            float* d = xw + (h * 16 + i) * kXld + 2 + s;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        for (int r = lane; r < 32 * 5; r += 64) {
            const int row = r / 5, k = r % 5;
            xw[row * kXld + (k < 2 ? k : 128 + k)] = 0.f;     // 0,1,130,131,132
        }
    }
    // ---- chunk 0 of the conv2 weights: LDS-DMA, 30 pieces of 1 KiB, piece p by wave p%4 ----
    for (int p = q; p < kWChunkFloats / 256; p += 4) glds16(wpack + p * 256 + lane * 4, wbuf + p * 256);
    glds_drain_and_barrier();

    const float* xrow0 = xw + (0 * 16 + nl) * kXld + g;    // lane reads xp[n][h][v + g]
    const float* xrow1 = xw + (1 * 16 + nl) * kXld + g;

    int it = 0;
    // conv1's A operand of the NEXT chunk is fetched one chunk ahead as well: a plain load issued behind the LDS-DMA would make
    // hipcc's wait for it (vmcnt(0): loads return in order) a wait for the whole prefetch
    float a1n = a1pack[lane];
    constexpr int kNPB = kW2 / KP;
    static_assert(kNPB * KP == kW2, "KP must divide 132");
    for (int pb = YSPLIT ? (int)blockIdx.y : 0; pb < (YSPLIT ? (int)blockIdx.y + 1 : kNPB); ++pb) {
        const int w0 = pb * KP;
        f32x4 acc[KP][5];
#pragma unroll
        for (int a = 0; a < KP; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int cc = 0; cc < kNChunk; ++cc, ++it) {
            const float* wb = wbuf + (it & 1) * kWChunkFloats + lane;
            // prefetch the next chunk's weights into the other buffer by LDS-DMA (wraps to chunk 0
            // for the next position block); every wave finished reading that buffer before the
            // barrier that ended the previous chunk
            {
                const int nc = (cc + 1) & (kNChunk - 1);
                const float* src = wpack + (size_t)nc * kWChunkFloats;
                float* dst = wbuf + ((it + 1) & 1) * kWChunkFloats;
                for (int p = q; p < kWChunkFloats / 256; p += 4) glds16(src + p * 256 + lane * 4, dst + p * 256);
            }
            const float a1 = a1n;                      // conv1 A operand: K1[c][g] (g<3) | b1[c] (g=3)
            a1n = a1pack[((cc + 1) & (kNChunk - 1)) * 64 + lane];
#pragma unroll
            for (int u = 0; u < KP + 2; ++u) {
                const int wp = w0 + u;                 // padded position w' of y1p
                if (wp < 2 || wp > 131) continue;      // zero padding of y1p: contributes nothing
                const int v = wp - 2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xv = (h == 0 ? xrow0 : xrow1)[v];
                    const float b1f = (g == 3) ? 1.0f : xv;          // k = 3 carries the bias
                    f32x4 X = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1f, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[r] = fmaxf(X[r], 0.f);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int wo = u - j;          // local output position: w = w' - j
                        if (wo < 0 || wo >= KP) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ot = 0; ot < 5; ++ot) {
                                const float a = wb[(((h * 3 + j) * 5 + ot) * 4 + r) * 64];
                                acc[wo][ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, X[r], acc[wo][ot], 0, 0, 0);
                            }
                    }
                }
            }
            // the next chunk has landed (this wave's pieces: vmcnt(0)), then everybody's (barrier)
            glds_drain_and_barrier();
        }
        // ---- epilogue of the block: bias, ReLU, store feat[frame][w][o] ----
        const long f = frame0 + nl;
        if (f < n) {
#pragma unroll
            for (int a = 0; a < KP; ++a)
#pragma unroll
                for (int ot = 0; ot < 5; ++ot) {
                    const int o = ot * 16 + g * 4;
                    const float4 bb = *reinterpret_cast<const float4*>(b2 + o);
                    float4 r;
                    r.x = fmaxf(acc[a][ot][0] + bb.x, 0.f);
                    r.y = fmaxf(acc[a][ot][1] + bb.y, 0.f);
                    r.z = fmaxf(acc[a][ot][2] + bb.z, 0.f);
                    r.w = fmaxf(acc[a][ot][3] + bb.w, 0.f);
                    *reinterpret_cast<float4*>(feat + (f * kW2 + (w0 + a)) * kC2 + o) = r;
                }
        }
    }
}

// ------------------------------------------------------------------------------------
// f32 dense1: hid[f][n] = relu(sum_k feat[f][k] W1p[k][n] + c1[n]),  M = frames, K = 10560, N = 256, on
// v_mfma_f32_16x16x4_f32 (exact f32 fma chains, K ascending: the order IS the result).
//
// Round 5: one work-group holds a 128-frame x 256-unit tile -- ALL hidden units of its frames -- so
//  * every feature row is read from HBM once (the 128 x 128 tiles of rounds 1-4 read it once per column half: 86,978 B per
//    frame through the PMC counters against 42,240 + 1,024 needed);
//  * HEAD = true runs dense2 + softmax + first-max argmax in the epilogue, on the hidden tile in LDS, with the head
//    kernel's own instruction chain (dense_chain_common.h: same operands, same order, same bits as the mdc_vt_head
//    launch): the hidden layer neither goes to HBM nor comes back, and the f32 batch path is two launches, not three.
// 8 waves (2 x 4), each 64 x 64 = 4 x 4 MFMA tiles, BK = 32, two LDS buffers filled through registers one K-tile ahead
// (global loads issued before a tile's MFMAs, LDS writes after them: one barrier per K-tile).  LDS images chosen for the
// fragment reads (ds_read_b32 / ds_read2_b32: two groups of 32 lanes over 32 banks): A as [row][34] -- a lane reads
// A[row = lane&15][k = kk + (lane>>4)], bank 2 row + k: the 32 lanes of a group on 32 banks --, B as [k][272] -- bank
// 16 (lane>>4) + (lane&15): likewise.  A rows are 136 B (8-byte aligned): written with two ds_write_b64, B with ds_write_b128.
// The per-element accumulation order (K-tiles ascending, k-steps of 4 ascending, one MFMA chain per output tile) is that
// of the old kernel and of vt_dense1_f32_small_kernel: bit-identical hidden layer (tests/test_fullsize_gpu.py).
// ------------------------------------------------------------------------------------
// ROWS = 128 (the batch kernel) or 64: the same kernel on half-height tiles for batches that would leave CUs idle at 128 rows
// (n <= 16,384: at most 256 tiles of 64 rows; the 128 x 128 tiles of rounds 1-4 had twice the work-groups per frame, and at
// n = 4,096 the 128-row form alone was slower than they were).  Same per-element order: bit-identical at every size.
constexpr int kDM = 128, kDK = 32;
constexpr int kDAld = 34, kDBld = 272;
constexpr int kDBBuf = kDK * kDBld;                                          // floats per B buffer
template <int ROWS> constexpr size_t dense1_f32_lds() {                      // the epilogue's [ROWS][260] image or the staging buffers, whichever is larger
    const size_t img = (size_t)ROWS * kChainXld * sizeof(float), stg = 2 * ((size_t)ROWS * kDAld + kDBBuf) * sizeof(float);
    return img > stg ? img : stg;
}
constexpr size_t kDense1F32Lds = dense1_f32_lds<128>();                      // 133,120 B

template <bool HEAD, int ROWS>
__global__ __launch_bounds__(512) void vt_dense1_f32_kernel(const float* __restrict__ feat, long n,
                                                            const float* __restrict__ w1p,   // [10560][256]
                                                            const float* __restrict__ c1,    // [256]
                                                            float* __restrict__ hid,         // [n][256] (HEAD = false)
                                                            const float* __restrict__ w2pack, int n_out,
                                                            float* __restrict__ probs, int* __restrict__ labels) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    constexpr int kDABuf = ROWS * kDAld, RI = ROWS / 32;      // floats per A buffer; 16-row fragments per wave (its ROWS/2 rows)
    float* As = dsm;                    // [2][ROWS][34]
    float* Bs = dsm + 2 * kDABuf;       // [2][32][272]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const long row0 = (long)blockIdx.x * ROWS;
    f32x4 acc[RI][4];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: A tile ROWS rows x 32 k = ROWS * 8 float4, one (64 rows) or two (128) per thread (8 consecutive threads = one
    //          row's 128 B); B tile 32 k x 256 cols = 2048 float4, four per thread (one wave = one k row's 1 KiB)
    constexpr bool kTwoA = ROWS == 128;
    const int ar0 = tid >> 3, ak4 = (tid & 7) * 4;           // A rows ar0 and (128-row tiles) ar0 + 64
    const int bk0 = tid >> 6, bc4 = (tid & 63) * 4;           // B rows bk0, +8, +16, +24
    const bool aok0 = row0 + ar0 < n, aok1 = kTwoA && row0 + ar0 + 64 < n;      // rows past the end: zeros, computed, never stored
    const float* ap0 = feat + (aok0 ? row0 + ar0 : 0) * (long)kFeat + ak4;
    const float* ap1 = feat + (aok1 ? row0 + ar0 + 64 : 0) * (long)kFeat + ak4;
    // (named registers, not arrays: hipcc kept float4 arrays captured by these lambdas in scratch)
    float4 ga0, ga1, gb0, gb1, gb2, gb3;
    const float* bp = w1p + (size_t)bk0 * kHid + bc4;
    auto fetch = [&](int k0) {
        // (the address is always valid -- clamped to row 0 --, the VALUE is selected: `ok ? *p : zero` made hipcc select
        // between two addresses and park the zeros in scratch
        // ... and selected in stash(), a K-tile later: selecting here would wait for the load right behind its issue)
        ga0 = *reinterpret_cast<const float4*>(ap0 + k0);
        if (kTwoA) ga1 = *reinterpret_cast<const float4*>(ap1 + k0);
        const float* b = bp + (size_t)k0 * kHid;
        gb0 = *reinterpret_cast<const float4*>(b);
        gb1 = *reinterpret_cast<const float4*>(b + 8 * kHid);
        gb2 = *reinterpret_cast<const float4*>(b + 16 * kHid);
        gb3 = *reinterpret_cast<const float4*>(b + 24 * kHid);
    };
    auto stash = [&](int b) {
        float* a = As + b * kDABuf + ar0 * kDAld + ak4;
        const float4 v0 = aok0 ? ga0 : make_float4(0.f, 0.f, 0.f, 0.f), v1 = aok1 ? ga1 : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float2*>(a) = make_float2(v0.x, v0.y);
        *reinterpret_cast<float2*>(a + 2) = make_float2(v0.z, v0.w);
        if (kTwoA) {
            *reinterpret_cast<float2*>(a + 64 * kDAld) = make_float2(v1.x, v1.y);
            *reinterpret_cast<float2*>(a + 64 * kDAld + 2) = make_float2(v1.z, v1.w);
        }
        float* d = Bs + b * kDBBuf + bk0 * kDBld + bc4;
        *reinterpret_cast<float4*>(d) = gb0;
        *reinterpret_cast<float4*>(d + 8 * kDBld) = gb1;
        *reinterpret_cast<float4*>(d + 16 * kDBld) = gb2;
        *reinterpret_cast<float4*>(d + 24 * kDBld) = gb3;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    constexpr int kTiles = kFeat / kDK;      // 330
    for (int t = 0; t < kTiles; ++t) {
        const int b = t & 1;
        if (t + 1 < kTiles) fetch((t + 1) * kDK);
        // the loads above are ISSUED here, one K-tile ahead of their use: left to itself hipcc sinks them behind the MFMAs and
        // waits for them at once (the whole global-load latency exposed at the end of every K-tile: 3.30 ms per 65,536 frames)
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = As + b * kDABuf + (wr * (ROWS / 2) + fr) * kDAld + fq;
        const float* Bb = Bs + b * kDBBuf + fq * kDBld + wc * 64 + fr;
#pragma unroll
        for (int kk = 0; kk < kDK; kk += 4) {
            float a[RI], bv[4];
#pragma unroll
            for (int i = 0; i < RI; ++i) a[i] = Ab[i * 16 * kDAld + kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = Bb[kk * kDBld + j * 16];
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < kTiles) stash(b ^ 1);      // buffer b^1 was last read in iteration t-1, behind that iteration's barrier
        __syncthreads();
    }
    // C/D layout: col = lane&15, row = 4*(lane>>4) + reg
    if constexpr (!HEAD) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = wc * 64 + j * 16 + fr;
                const float bias = c1[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = row0 + wr * (ROWS / 2) + i * 16 + fq * 4 + r;
                    if (row < n) hid[row * kHid + col] = fmaxf(acc[i][j][r] + bias, 0.f);
                }
            }
    } else {
        // fused head: the staging buffers (every wave is past its last fragment read: the loop's final barrier) become the
        // head kernel's [row][260] image of the 128 x 256 hidden tile; wave wv then runs the head's chain on rows 16 wv ..
        float* xs = dsm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wc * 64 + j * 16 + fr;
            const float bias = c1[col];
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[(wr * (ROWS / 2) + i * 16 + fq * 4 + r) * kChainXld + col] = fmaxf(acc[i][j][r] + bias, 0.f);
        }
        float w2[64];                                  // dense2 as B operands, the head kernel's packing (chain_pack_layer)
#pragma unroll
        for (int i = 0; i < 64; ++i) w2[i] = w2pack[i * 64 + lane];
        const float b2 = w2pack[64 * 64 + 16 * 64 + fr];
        __syncthreads();
        if (wv < ROWS / 16) {      // (64-row tiles: waves 0 .. 3; wave-uniform)
            const f32x4 a2 = chain_k256(xs + (wv * 16 + fr) * kChainXld + fq, w2, f32x4{0.f, 0.f, 0.f, 0.f});
            f32x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = a2[r] + b2;
            chain_softmax_store(z, fr, fq, row0 + wv * 16, n, n_out, probs, labels, nullptr);
        }
    }
}

#ifdef MDC_ALTERNATES
// Rounds 1-4's f32 dense1 (128 x 128 tiles, one LDS buffer, the head as its own launch), kept in the alternates build as the
// bit-identity screen of the kernel above (MDC_DENSE1_PHASED=0 selects it, as it selects the one-barrier bf16 kernel).
constexpr int kDN = 128, kDld = 129;

__global__ __launch_bounds__(256) void vt_dense1_f32_tiles128_kernel(const float* __restrict__ feat, long n,
                                                                     const float* __restrict__ w1p,   // [10560][256]
                                                                     const float* __restrict__ c1,    // [256]
                                                                     float* __restrict__ hid) {       // [n][256]
    __shared__ float As[kDK][kDld];     // As[k][row]
    __shared__ float Bs[kDK][kDld + 3]; // Bs[k][col]  (ld 132: float4 aligned rows)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const long row0 = (long)blockIdx.x * kDM;
    const int col0 = blockIdx.y * kDN;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < kFeat; k0 += kDK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 3, k4 = (idx & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < n) v = *reinterpret_cast<const float4*>(feat + (row0 + r) * kFeat + k0 + k4);
            As[k4 + 0][r] = v.x; As[k4 + 1][r] = v.y; As[k4 + 2][r] = v.z; As[k4 + 3][r] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int k = idx >> 5, c4 = (idx & 31) * 4;
            const float4 v = *reinterpret_cast<const float4*>(w1p + (size_t)(k0 + k) * kHid + col0 + c4);
            *reinterpret_cast<float4*>(&Bs[k][c4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kDK; kk += 4) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk + fq][wr * 64 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk + fq][wc * 64 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col0 + wc * 64 + j * 16 + fr;
            const float bias = c1[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + wr * 64 + i * 16 + fq * 4 + r;
                if (row < n) hid[row * kHid + col] = fmaxf(acc[i][j][r] + bias, 0.f);
            }
        }
}
#endif

// Small batches: one wave = a 16-frame x 16-unit tile over the whole K, both operands straight from global memory in the
// MFMA's lane order (A[row = lane&15][k = kk + (lane>>4)], B[k][col = lane&15], as the tiled kernel reads them from LDS):
// the same instruction and the same K order per output element, so the results are bit-identical, on 16 x ceil(n/16)
// CUs instead of one or two.
__global__ __launch_bounds__(64) void vt_dense1_f32_small_kernel(const float* __restrict__ feat, long n,
                                                                 const float* __restrict__ w1p,   // [10560][256]
                                                                 const float* __restrict__ c1, float* __restrict__ hid) {
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    const long row0 = (long)blockIdx.x * 16;
    const int col0 = blockIdx.y * 16;
    long ar = row0 + fr;
    if (ar >= n) ar = n - 1;                                  // rows past the end are computed, not stored
    const float* ap = feat + ar * (long)kFeat + fq;
    const float* bp = w1p + (size_t)fq * kHid + col0 + fr;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // software pipeline: the operands of the next kU k-steps are in flight while this block's MFMAs (one dependent
    // chain: the K order is the result) run -- loading and multiplying in turns took 114-307 us per launch, K = 10,560
    constexpr int kU = 24;                                    // MFMAs (k-steps of 4) per block: 48 loads in flight
    constexpr int kBlocks = kFeat / (4 * kU);
    static_assert(kBlocks * 4 * kU == kFeat, "block size must divide K / 4");
    float a[2][kU], b[2][kU];
    auto fetch = [&](int blk, float (&aa)[kU], float (&bb)[kU]) {
        const int k0 = blk * 4 * kU;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            aa[u] = ap[k0 + 4 * u];
            bb[u] = bp[(size_t)(k0 + 4 * u) * kHid];
        }
    };
    fetch(0, a[0], b[0]);
    for (int blk = 0; blk < kBlocks; blk += 2) {
        if (blk + 1 < kBlocks) fetch(blk + 1, a[1], b[1]);
#pragma unroll
        for (int u = 0; u < kU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][u], b[0][u], acc, 0, 0, 0);
        if (blk + 1 < kBlocks) {
            if (blk + 2 < kBlocks) fetch(blk + 2, a[0], b[0]);
#pragma unroll
            for (int u = 0; u < kU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][u], b[1][u], acc, 0, 0, 0);
        }
    }
    const int col = col0 + fr;
    const float bias = c1[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = row0 + fq * 4 + r;
        if (row < n) hid[row * kHid + col] = fmaxf(acc[r] + bias, 0.f);
    }
}

// feat[f][w][o] (f32) / feat[f][feat16_index(w, o)] (bf16) -> reference layout (80,132) channels_first, f32   ('conv'/'flat' taps)
template <typename T>
__global__ void vt_unpermute_kernel(const T* __restrict__ feat, long n, float* __restrict__ out, float unscale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kFeat) return;
    const long f = i / kFeat;
    const int r = (int)(i % kFeat);
    const int o = r / kW2, w = r % kW2;
    if constexpr (sizeof(T) == 2) {      // 16-bit modes: positions in pairs (feat16_index), values times a power of two
        const T v = feat[f * kFeat + feat16_index(w, o)];
        out[i] = __uint_as_float(((unsigned)v) << 16) * unscale;      // exact
    } else {
        out[i] = feat[(f * kW2 + w) * kC2 + o];
    }
}

// E4M3 features (fp8 mode): feat[f][feat8_index(w, o)], one byte each, true value x 2^k -> the same reference layout
__global__ void vt_unpermute8_kernel(const unsigned char* __restrict__ feat, long n, float* __restrict__ out, float unscale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kFeat) return;
    const long f = i / kFeat;
    const int r = (int)(i % kFeat);
    const int o = r / kW2, w = r % kW2;
    const unsigned v = feat[f * kFeat + feat8_index(w, o)];
    const int e = (int)((v >> 3) & 15u), mant = (int)(v & 7u);      // OCP E4M3: bias 7, subnormals at e = 0, 0x7F / 0xFF = NaN
    float mag = e == 0 ? ldexpf((float)mant, -9) : ldexpf(1.f + (float)mant * 0.125f, e - 7);
    if ((v & 0x7Fu) == 0x7Fu) mag = __uint_as_float(0x7FC00000u);
    out[i] = ((v & 0x80u) ? -mag : mag) * unscale;                   // exact (a power of two)
}

}  // namespace

// ---- dense_chain.hip: Dense(C) + softmax + first-max argmax of the head -----------------
void chain_pack_layer(std::vector<float>& dst, const float* w, int k_in, int n_out, int ksteps, int tiles);
int chain_launch(int t1, int nl, const float* x, long n, const float* wpack, int n_out, int relu1, int relu2,
                 float* probs, int* labels, float* tap_logits, float* tap_h1, float* tap_h2, int n1, int n2, hipStream_t s);

// ---- bf16 entry points (vtcnn2_bf16.hip) ------------------------------------------------
int vtcnn2_bf16_pack(mdc_model* m);
int vtcnn2_bf16_conv(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s, long hop2 = 0, float scale = 0.f);
int vtcnn2_bf16_dense1(const mdc_model* m, const void* feat, int64_t n, float* hid, hipStream_t s,
                       bool fuse_head, float* probs, int32_t* labels, bool* fused);

// d_pack slots: 0 conv2 weights, 1 conv1 operand, 2 conv2 bias, 3 dense1 weights (permuted),
//               4 dense1 bias, 5 head (dense2) pack for dense_chain
int vtcnn2_pack(mdc_model* m) {
    const float* k1 = m->hk[0].data();   // OIHW (256,1,1,3)
    const float* b1 = m->hb[0].data();
    const float* k2 = m->hk[1].data();   // OIHW (80,256,2,3): ((o*256 + c)*2 + h)*3 + j
    int rc;
    if ((rc = upload(m, 2, m->hb[1].data(), kC2 * sizeof(float)))) return rc;
    if ((rc = upload(m, 4, m->hb[2].data(), kHid * sizeof(float)))) return rc;
    {   // head: dense2 as the single layer of the dense_chain kernel
        const int C = m->topo.classes;
        if (C > 16) { set_error("vtcnn2: the HIP head covers up to 16 classes (got %d)", C); return MDC_ENOTSUP; }
        std::vector<float> pk;
        chain_pack_layer(pk, m->hk[3].data(), kHid, C, 64, 1);
        pk.resize(pk.size() + 16 * 64, 0.f);                 // layers 2/3 unused
        std::vector<float> bias(96, 0.f);
        for (int c = 0; c < C; ++c) bias[c] = m->hb[3][c];
        pk.insert(pk.end(), bias.begin(), bias.end());
        if ((rc = upload(m, 5, pk.data(), pk.size() * sizeof(float)))) return rc;
    }
    if (m->dtype == MDC_BF16) return vtcnn2_bf16_pack(m);
    if (m->dtype == MDC_FP8) return vtcnn2_fp8_pack(m);      // overwrites slot 2 with the scaled conv2 bias

    // conv2 weights: [chunk][h][j][ot][r][lane] = K2[16ot + (lane&15)][16chunk + 4(lane>>4) + r][h][j]
    std::vector<float> wp((size_t)kNChunk * kWChunkFloats);
    for (int cc = 0; cc < kNChunk; ++cc)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 3; ++j)
                for (int ot = 0; ot < 5; ++ot)
                    for (int r = 0; r < 4; ++r)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int o = 16 * ot + (lane & 15), c = 16 * cc + 4 * (lane >> 4) + r;
                            wp[(size_t)cc * kWChunkFloats + ((((h * 3 + j) * 5 + ot) * 4 + r) * 64) + lane] =
                                k2[(((size_t)o * kC1 + c) * 2 + h) * 3 + j];
                        }
    if ((rc = upload(m, 0, wp.data(), wp.size() * sizeof(float)))) return rc;
    // conv1 A operand per chunk: lane (c = lane&15, g = lane>>4): K1[16chunk + c][g] for g<3, b1 for g=3
    std::vector<float> a1((size_t)kNChunk * 64);
    for (int cc = 0; cc < kNChunk; ++cc)
        for (int lane = 0; lane < 64; ++lane) {
            const int c = 16 * cc + (lane & 15), g = lane >> 4;
            a1[cc * 64 + lane] = (g < 3) ? k1[c * 3 + g] : b1[c];
        }
    if ((rc = upload(m, 1, a1.data(), a1.size() * sizeof(float)))) return rc;
    // dense1 rows permuted to the workspace order: row (w*80 + o) <- reference row (o*132 + w)
    const float* w1 = m->hk[2].data();
    std::vector<float> w1p((size_t)kFeat * kHid);
    for (int w = 0; w < kW2; ++w)
        for (int o = 0; o < kC2; ++o)
            std::copy(w1 + (size_t)(o * kW2 + w) * kHid, w1 + (size_t)(o * kW2 + w + 1) * kHid,
                      w1p.begin() + (size_t)(w * kC2 + o) * kHid);
    return upload(m, 3, w1p.data(), w1p.size() * sizeof(float));
}

// bytes per feature: f32; bf16 in the bf16 mode (and in the fp8 mode under MDC_OPT_FP8_BF16_FEATURES); E4M3 in the fp8 mode
static size_t feat_elem(const mdc_model* m) { return m->dtype == MDC_F32 ? 4 : (m->dtype == MDC_FP8 && m->fp8_e4m3_features) ? 1 : 2; }

size_t vtcnn2_workspace_bytes(const mdc_model* m, int64_t n) {
    const size_t np = ((size_t)n + 255) & ~(size_t)255;   // kernels write whole 16-frame groups / 256-row tiles
    return np * kFeat * feat_elem(m) + np * kHid * sizeof(float);
}

// hop2 > 0: x points at raw uint8 I/Q windows hop2 bytes apart (mdc_forward_iq_u8); otherwise f32 frames
static int vtcnn2_run(const mdc_model* m, const float* x, long hop2, float scale, int64_t n, float* probs, int32_t* labels,
                      float* tap, int tap_kind, void* ws, size_t ws_bytes, hipStream_t s) {
    const size_t need = vtcnn2_workspace_bytes(m, n);
    if (!ws || ws_bytes < need) { set_error("vtcnn2 forward of %lld frames needs %zu workspace bytes (got %zu)", (long long)n, need, ws_bytes); return MDC_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) { set_error("workspace must be 256-byte aligned"); return MDC_EINVAL; }
    const size_t fbytes = (((size_t)n + 255) & ~(size_t)255) * kFeat * feat_elem(m);
    void* feat = ws;
    float* hid = reinterpret_cast<float*>(static_cast<char*>(ws) + fbytes);
    const int C = m->topo.classes;
    int rc;
    bool head_done = false;      // dense2 + softmax + argmax already ran in dense1's epilogue (16-bit modes, batch kernels, no dense/hidden tap)
    if (m->dtype == MDC_BF16 || m->dtype == MDC_FP8) {
        { ProfScope ps(m, 0, s); if ((rc = m->dtype == MDC_FP8 ? vtcnn2_fp8_conv(m, x, n, feat, s, hop2, scale) : vtcnn2_bf16_conv(m, x, n, feat, s, hop2, scale))) return rc; }
        const bool fuse = tap_kind != MDC_TAP_DENSE && tap_kind != MDC_TAP_HIDDEN;
        { ProfScope ps(m, 1, s); if ((rc = vtcnn2_bf16_dense1(m, feat, n, hid, s, fuse, probs, labels, &head_done))) return rc; }
    } else {
        {
            ProfScope ps(m, 0, s);
            // per launch, like every other kernel here: the attribute is per DEVICE, and one process may drive several.
            // Form by batch size (results identical): up to 128 frames one position per work-group (132 work-groups per
            // 64-frame group), up to 2,048 three, up to 8,192 the batch blocking with one block per work-group, else the loop.
            const unsigned groups = (unsigned)((n + 63) / 64);
            constexpr long t1 = 128, t3 = 2048, t11 = 8192;      // measured crossovers (conv us at n = 128 / 2,048 / 8,192: 90 / 668 / 2,140; the loop form 3,990 / 4,006 / 4,059)
#define MDC_LAUNCH_CONV_F32(U, KP, YS) do { \
                MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_f32_kernel<U, KP, YS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvF32Lds)); \
                hipLaunchKernelGGL((vt_conv_f32_kernel<U, KP, YS>), dim3(groups, YS ? kW2 / KP : 1), dim3(256), kConvF32Lds, s, x, (long)n, \
                                   static_cast<const float*>(m->d_pack[0]), static_cast<const float*>(m->d_pack[1]), \
                                   static_cast<const float*>(m->d_pack[2]), static_cast<float*>(feat), hop2 > 0 ? hop2 : 256L, scale); } while (0)
            if (hop2 > 0) {
                if (n <= t1) MDC_LAUNCH_CONV_F32(true, 1, true);
                else if (n <= t3) MDC_LAUNCH_CONV_F32(true, 3, true);
                else if (n <= t11) MDC_LAUNCH_CONV_F32(true, 11, true);
                else MDC_LAUNCH_CONV_F32(true, 11, false);
            } else {
                if (n <= t1) MDC_LAUNCH_CONV_F32(false, 1, true);
                else if (n <= t3) MDC_LAUNCH_CONV_F32(false, 3, true);
                else if (n <= t11) MDC_LAUNCH_CONV_F32(false, 11, true);
                else MDC_LAUNCH_CONV_F32(false, 11, false);
            }
#undef MDC_LAUNCH_CONV_F32
            MDC_HIP(hipGetLastError());
        }
        {
            ProfScope ps(m, 1, s);
            const float* featf = static_cast<const float*>(feat);
            const float* w1p = static_cast<const float*>(m->d_pack[3]);
            const float* c1 = static_cast<const float*>(m->d_pack[4]);
            const float* w2pack = static_cast<const float*>(m->d_pack[5]);
            bool done = false;
#ifdef MDC_ALTERNATES
            if (n > 2048 && (m->alt & kAltDense1Simple)) {      // the screen: rounds 1-4's tiles, head as its own launch below
                hipLaunchKernelGGL(vt_dense1_f32_tiles128_kernel, dim3((unsigned)((n + kDM - 1) / kDM), kHid / kDN), dim3(256), 0, s, featf, (long)n, w1p, c1, hid);
                done = true;
            }
#endif
            if (done) {
            } else if (n <= 2048) {      // per-wave 16 x 16 tiles: 108 us at n = 256, 617 at 2,048; the work-group tiles take > 1 ms whatever the batch
                hipLaunchKernelGGL(vt_dense1_f32_small_kernel, dim3((unsigned)((n + 15) / 16), kHid / 16), dim3(64), 0, s, featf, (long)n, w1p, c1, hid);
            } else {
                // the head rides in the epilogue unless a tap wants the hidden layer or the logits in HBM
                bool fuse = tap_kind != MDC_TAP_DENSE && tap_kind != MDC_TAP_HIDDEN;
#ifdef MDC_ALTERNATES
                if (m->alt & kAltSeparateHead) fuse = false;
#endif
#define MDC_LAUNCH_D1F32(H, R) do { \
                    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_f32_kernel<H, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dense1_f32_lds<R>())); \
                    hipLaunchKernelGGL((vt_dense1_f32_kernel<H, R>), dim3((unsigned)((n + R - 1) / R)), dim3(512), dense1_f32_lds<R>(), s, featf, (long)n, w1p, c1, hid, \
                                       w2pack, C, probs, labels); } while (0)
                const bool half = n <= 16384;      // at most 256 tiles of 64 rows: every tile on its own CU
                if (fuse) {
                    if (half) MDC_LAUNCH_D1F32(true, 64); else MDC_LAUNCH_D1F32(true, 128);
                    head_done = true;
                } else {
                    if (half) MDC_LAUNCH_D1F32(false, 64); else MDC_LAUNCH_D1F32(false, 128);
                }
#undef MDC_LAUNCH_D1F32
            }
            MDC_HIP(hipGetLastError());
        }
    }
    if (!head_done) {
        ProfScope ps(m, 2, s);
        if ((rc = chain_launch(1, 1, hid, (long)n, static_cast<const float*>(m->d_pack[5]), C, 0, 0, probs, labels,
                               tap_kind == MDC_TAP_DENSE ? tap : nullptr, nullptr, nullptr, 0, 0, s)))
            return rc;
    }
    if (tap_kind == MDC_TAP_CONV || tap_kind == MDC_TAP_FLAT) {
        const long total = (long)n * kFeat;
        // fp8 mode keeps its features multiplied by a power of two (dense1's weights carry the inverse)
        const float unscale = std::ldexp(1.f, -m->feat_scale_log2);      // the 16-bit modes keep their features times a power of two
        if (feat_elem(m) == 1)
            hipLaunchKernelGGL(vt_unpermute8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned char*>(feat), (long)n, tap, unscale);
        else if (m->dtype != MDC_F32)
            hipLaunchKernelGGL(vt_unpermute_kernel<unsigned short>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned short*>(feat), (long)n, tap, unscale);
        else
            hipLaunchKernelGGL(vt_unpermute_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(feat), (long)n, tap, unscale);
        MDC_HIP(hipGetLastError());
    } else if (tap_kind == MDC_TAP_HIDDEN) {
        MDC_HIP(hipMemcpyAsync(tap, hid, (size_t)n * kHid * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return MDC_OK;
}

int vtcnn2_forward(const mdc_model* m, const float* x, int64_t n, float* probs, int32_t* labels,
                   float* tap, int tap_kind, void* ws, size_t ws_bytes, hipStream_t s) {
    return vtcnn2_run(m, x, 0, 0.f, n, probs, labels, tap, tap_kind, ws, ws_bytes, s);
}

// raw SDR bytes straight into the conv kernels' staging: window i = the 128 (I,Q) pairs from
// pair i*hop of one contiguous capture; no frame buffer in between.  Same workspace as mdc_forward.
int vtcnn2_forward_iq_u8(const mdc_model* m, const uint8_t* iq, int64_t n, int64_t hop, float scale, float* probs, int32_t* labels,
                         void* ws, size_t ws_bytes, hipStream_t s) {
    return vtcnn2_run(m, reinterpret_cast<const float*>(iq), 2 * (long)hop, scale, n, probs, labels, nullptr, MDC_TAP_NONE, ws, ws_bytes, s);
}

}  // namespace mdc
