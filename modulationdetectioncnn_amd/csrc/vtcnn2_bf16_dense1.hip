// Canonical VT-CNN2 (T3), bf16 path: dense1 (10560 -> 256, bias + ReLU) as a tiled MFMA GEMM.
//
// Two kernels, bit-identical results (same tiles, same accumulation order):
//   vt_dense1_bf16_kernel         256x256x64 tiles, LDS-DMA staging with an XOR-swizzled source (so ds_read_b128
//                                 fragments spread over the banks), two LDS buffers, 8 waves (2 x 4), one
//                                 __syncthreads per K-tile.  Simple; LDS-DMA issue + fragment reads and the MFMAs do
//                                 not overlap (tools/ablate_dense1.py).  Kept as the race screen (MDC_DENSE1_PHASED=0).
//   vt_dense1_bf16_phased_kernel  the production kernel; see the comment above it.
//
// Round 4, F8 = true (the fp8 mode's default): the feature matrix holds E4M3 bytes (one power-of-two scale per tensor,
// feat8_index order; vtcnn2_fp8_conv.hip) -- half the bytes of dense1's HBM stream, which is what bounds it.  The MFMA
// stays v_mfma_f32_16x16x32_bf16 with bf16 weights (e4m3 WEIGHTS miss a label floor, profiles/r03_exp_fp8_features.json):
// a feature fragment is 8 bytes per lane (ds_read_b64 / global_load_dwordx2) and becomes the bf16x8 operand through four
// v_cvt_scalef32_pk_bf16_fp8 (x 1.0: exact; tools/microbench/cvt_bf16_fp8_probe.hip) -- 64 VALU per wave and K-tile next
// to 64 MFMAs of 16 cycles.  Every form (batch, per-wave, four-wave ring) converts the same bytes to the same operands
// and keeps the K order, so they stay bit-identical to each other.
#include "vtcnn2_bf16_common.h"
#include "dense_chain_common.h"

#include <cstdlib>

namespace mdc {

namespace {

// ------------------------------------------------------------------------------------
// dense1 bf16 GEMM
// ------------------------------------------------------------------------------------
constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kTileBytes = kBM * kBK * 2;                     // 32 KiB per operand tile
constexpr size_t kDenseBf16Lds = (size_t)4 * kTileBytes;      // A,B x 2 buffers = 128 KiB
constexpr size_t kDenseHeadLds = (size_t)128 * kChainXld * 4;  // fused head: [128 rows][260] f32 image of a wave row's hidden tile (133,120 B)
static_assert(kDenseHeadLds >= kDenseBf16Lds, "the fused-head form allocates the larger of the two");
constexpr int kNT = kFeat / kBK;                              // 165 K-tiles

// 8 E4M3 bytes (k ascending) -> the bf16x8 MFMA operand; x 1.0, exact.  The builtin (not inline asm) so that hipcc's
// scheduler sees four VALU instructions it may place between MFMAs (__builtin_amdgcn_sched_group_barrier below).
__device__ __forceinline__ bf16x8 cvt_e4m3x8(u32x2 r) {
    const bf16x2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], 1.0f, false);
    const bf16x2 p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], 1.0f, true);
    const bf16x2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], 1.0f, false);
    const bf16x2 p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], 1.0f, true);
    return bf16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}


#if defined(MDC_ALTERNATES) || defined(MDC_ABLATIONS)      // the race screen of the phased kernel: test / probe builds only
template <int ABL>   // 0 = product; 1..3 = timing-only probes (tools/ablate_dense1.py, -DMDC_ABLATIONS; results wrong)
__global__ __launch_bounds__(512) void vt_dense1_bf16_kernel(const unsigned short* __restrict__ feat, long n,
                                                             const unsigned short* __restrict__ w1t,   // [165 k-tiles][256][64] bf16
                                                             const float* __restrict__ c1,
                                                             float* __restrict__ hid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const long row0 = (long)blockIdx.x * kBM;

    // staging: each wave moves 4 pieces (8 rows x 128 B) of A and 4 of B per K-tile.  LDS is linear
    // (row*128 + pos*16); the SOURCE chunk is pos ^ (row & 7), and readers apply the same XOR.
    const int srow = lane >> 3, spos = lane & 7;
    const unsigned short* asrc[4];
    const unsigned short* bsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = (wv * 4 + p) * 8 + srow;              // tile row 0..255
        long gr = (ABL == 1 ? 0 : row0) + r;                // probe 1: every work-group streams the same rows (L2 hits)
        if (gr >= n) gr = n - 1;                            // clamp: rows past the end are computed, not stored
        asrc[p] = feat + gr * (long)kFeat + ((spos ^ (r & 7)) * 8);
        bsrc[p] = w1t + r * kBK + ((spos ^ (r & 7)) * 8);
    }
    auto stage = [&](int t, int b) {
        if (ABL == 3 && t > 1) return;                      // probe 3: no staging traffic (compute on stale tiles)
        unsigned char* A = smem + (size_t)b * 2 * kTileBytes;
        unsigned char* B = A + kTileBytes;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            glds16(asrc[p] + t * kBK, A + (wv * 4 + p) * 1024);
            glds16(bsrc[p] + (long)t * (kBN * kBK), B + (wv * 4 + p) * 1024);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int b) {
        const unsigned char* A = smem + (size_t)b * 2 * kTileBytes;
        const unsigned char* B = A + kTileBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = wr * 128 + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(A + r * 128 + (((ks * 4 + fg) ^ (r & 7)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wc * 64 + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(B + r * 128 + (((ks * 4 + fg) ^ (r & 7)) * 16));
            }
            if (ABL == 2) {                                  // probe 2: staging + fragment reads, no MFMA
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(bfr[j]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    stage(0, 0);
    __syncthreads();                 // drains the LDS-DMA (vmcnt(0)) and orders it for every wave
    int cur = 0;
    for (int t = 0; t < kNT - 1; ++t) {
        stage(t + 1, cur ^ 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wc * 64 + j * 16 + fr;
        const float bias = c1[col];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + wr * 128 + i * 16 + fg * 4 + r;
                if (row < n) hid[row * kHid + col] = fmaxf(acc[i][j][r] + bias, 0.f);
            }
    }
}


#endif

// The tile's epilogue, shared by the GEMM kernels below.  HEAD = false: bias + ReLU, hidden layer to HBM.  HEAD = true: dense2 +
// softmax + argmax on the tile (see the comment above vt_dense1_bf16_phased_kernel); the caller guarantees that no LDS-DMA is
// in flight and that every wave is past its last fragment read once it has passed the first barrier in here.
template <bool HEAD>
__device__ __forceinline__ void d1_epilogue(f32x4 (&acc)[8][4], unsigned char* smem, long row0, long n, int lane, int wv,
                                            const float* __restrict__ c1, float* __restrict__ hid, const float* __restrict__ w2pack,
                                            int n_out, float* __restrict__ probs, int* __restrict__ labels) {
    const int wr = wv >> 2, wc = wv & 3, fr = lane & 15, fg = lane >> 4;
    if constexpr (!HEAD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wc * 64 + j * 16 + fr;
            const float bias = c1[col];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = row0 + wr * 128 + i * 16 + fg * 4 + r;
                    if (row < n) hid[row * kHid + col] = fmaxf(acc[i][j][r] + bias, 0.f);
                }
        }
    } else {
        // ---- fused head.  No LDS-DMA is in flight here (the last tile waited vmcnt(0)) and every wave is past its last
        // fragment read once it has passed the barrier below: the staging buffers become the [128][260] f32 image
        float* xs = reinterpret_cast<float*>(smem);
        float w2[64];                                  // dense2 as B operands, the head kernel's packing (chain_pack_layer)
#pragma unroll
        for (int i = 0; i < 64; ++i) w2[i] = w2pack[i * 64 + lane];
        const float b2 = w2pack[64 * 64 + 16 * 64 + fr];
        float bias1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias1[j] = c1[wc * 64 + j * 16 + fr];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if (wr == half) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xs[(i * 16 + fg * 4 + r) * kChainXld + wc * 64 + j * 16 + fr] = fmaxf(acc[i][j][r] + bias1[j], 0.f);
            }
            __syncthreads();
            const f32x4 a2 = chain_k256(xs + (wv * 16 + fr) * kChainXld + fg, w2, f32x4{0.f, 0.f, 0.f, 0.f});
            f32x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = a2[r] + b2;
            chain_softmax_store(z, fr, fg, row0 + half * 128 + wv * 16, n, n_out, probs, labels, nullptr);
        }
    }
}

// ------------------------------------------------------------------------------------
// vt_dense1_bf16_phased_kernel -- the same 256x256x64 tiles and the same accumulation order (bit-identical results),
// restructured after the "8-phase" schedule of cdna_hip_programming.md section 5:
//   * a K-tile is staged as four 16-KiB UNITS, one per phase, in the order the MFMA quadrants need them:
//       U0 = rows 0..63 of each wave row-half (a0), U1 = columns 0..31 of each wave column-quarter (b0),
//       U2 = columns 32..63 (b1), U3 = rows 64..127 (a1);
//     phase p of tile T stages unit p of tile T+1 (other LDS buffer), so three to four units are always in flight
//     and every wait is a counted s_waitcnt vmcnt(4), never 0, behind a raw s_barrier;
//   * phase 0 computes quadrant (a0,b0), phase 1 (a0,b1), phase 2 (a1,b1), phase 3 (a1,b0): 4 or 8 (or 12)
//     ds_read_b128 and 16 MFMAs per phase;
//   * the two wave rows run half a phase apart (wave row 1 passes one extra barrier first; two barriers per phase):
//     while one does its LDS reads / LDS-DMA issue / wait, the other issues MFMAs on the same SIMDs.
// Ordering of LDS-DMA data for a ds_read: the issuing wave's counted vmcnt, then a barrier the reader has passed
// (plus one more barrier for the other wave row) -- reads of a unit sit in the phase AFTER the wait that retires it.
// ------------------------------------------------------------------------------------
constexpr int kUnitBytes = 128 * 128;      // 128 rows x 64 bf16

// Round 3: (1) the feature rows (read exactly once) are fetched non-temporally -- 4.70 -> 4.50 ms per 2^20 frames, the
// weights keep the L2 to themselves; fetching every unit seven phases ahead instead of four (80 KiB in flight per CU,
// vmcnt(10)) did NOT help (4.71-4.84 ms): the stream is not latency-bound (profiles/r03_d1_prefetch_nt_ab.log).
// (2) HEAD = true fuses dense2 + softmax + argmax (the mdc_vt_head launch) into the epilogue: a work-group holds all 256
// hidden units of its 256 frames, so each wave row in turn puts its half of the hidden tile (bias + ReLU applied) into
// LDS in the head kernel's [row][260] image and every wave runs the head's own 64-step v_mfma_f32_16x16x4_f32 chain and
// softmax code (dense_chain_common.h) on one 16-row tile -- same operands, same order, same bits as the separate launch;
// the 1 KiB/frame hidden layer is then neither written to HBM nor read back.
// F8 (round 4): the A units hold E4M3 bytes -- 128 rows x 64 B = 8 KiB, ONE LDS-DMA instruction per wave (16 rows x 64 B:
// lane l -> row l>>2, 16-byte chunk (l&3) ^ ((row>>2)&3) of the source, so that the ds_read_b64 fragment reads of 16 rows
// x 2 halves spread over all 64 banks); the units keep their 16-KiB slots.  With 1 + 2 + 2 + 1 copies per K-tile
// the counted waits become vmcnt(2) / (3) / (4) / (3) (last tile: (1) / (0)): phase p still retires exactly the unit the
// next phase reads.  The raw fragments (ra) are converted once, right after their wait: a1 in phase 2 before its first
// quadrant, the next tile's a0 at the end of phase 3 under that phase's MFMAs.
// ABL 0 = product; timing-only probes (results wrong by construction; tools/ablate_dense1.py): 1 = every work-group streams the same A
// rows (L2 hits); round 5: 2 = the WEIGHT units staged on even K-tiles only (half the weight LDS-DMA: the most a 512-row tile could
// save on that side), 3 = the fragment reads on even K-tiles only (half the LDS read volume: the most 32x32x16 quadrants could save)
template <int ABL, bool HEAD = false, bool F8 = false>
__global__ __launch_bounds__(512) void vt_dense1_bf16_phased_kernel(const unsigned short* __restrict__ feat, long n,
                                                                    const unsigned short* __restrict__ w1t,   // [165][256][64]
                                                                    const float* __restrict__ c1, float* __restrict__ hid,
                                                                    const float* __restrict__ w2pack = nullptr, int n_out = 0,
                                                                    float* __restrict__ probs = nullptr, int* __restrict__ labels = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 buffers][4 units][16 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const long row0 = (long)blockIdx.x * kBM;

    // ---- staging addresses: per phase this wave moves pieces 2wv and 2wv+1 (8 unit rows x 128 B each) of the unit
    const int srow = lane >> 3, spos = lane & 7;
    const unsigned short* src[4][2];      // [unit][piece]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = (wv * 2 + j) * 8 + srow;                 // unit row 0..127
        const int sw = (spos ^ (u & 7)) * 8;                   // swizzled source chunk (LDS destination is linear)
        const int arow_lo = 128 * (u >> 6) + (u & 63);         // tile row of U0; U3 = +64
        const int brow_lo = 64 * (u >> 5) + (u & 31);          // B row (output column) of U1; U2 = +32
        long g0 = (ABL == 1 ? 0 : row0) + arow_lo, g3 = (ABL == 1 ? 0 : row0) + arow_lo + 64;
        if (g0 >= n) g0 = n - 1;                               // rows past the end are computed, not stored
        if (g3 >= n) g3 = n - 1;
        src[0][j] = feat + g0 * (long)kFeat + sw;
        src[3][j] = feat + g3 * (long)kFeat + sw;
        src[1][j] = w1t + brow_lo * kBK + sw;
        src[2][j] = w1t + (brow_lo + 32) * kBK + sw;
    }
    const unsigned char* src8[2] = {nullptr, nullptr};      // F8: [a0 unit, a1 unit], one 16-row piece per wave
    if constexpr (F8) {
        const int u = wv * 16 + (lane >> 2);                   // unit row 0..127
        const int sw = ((lane & 3) ^ ((u >> 2) & 3)) * 16;     // swizzled source chunk (bytes)
        const int arow_lo = 128 * (u >> 6) + (u & 63);
        long g0 = row0 + arow_lo, g3 = row0 + arow_lo + 64;
        if (g0 >= n) g0 = n - 1;
        if (g3 >= n) g3 = n - 1;
        const unsigned char* f8 = reinterpret_cast<const unsigned char*>(feat);
        src8[0] = f8 + g0 * (long)kFeat + sw;
        src8[1] = f8 + g3 * (long)kFeat + sw;
    }
    auto stage_unit = [&](int t, int unit, int b) {
        if (ABL == 2 && (unit == 1 || unit == 2) && (t & 1)) return;
        unsigned char* dst = smem + ((size_t)b * 4 + unit) * kUnitBytes + (wv * 2) * 1024;
        if constexpr (F8) {
            if (unit == 0 || unit == 3) {      // E4M3 feature rows: 64 B per row and K-tile, one piece per wave.  NOT non-temporal:
                // a row's 128-byte line holds TWO K-tiles, and with the streaming hint the second half came from HBM again
                // (PMC: 19.1 KB/frame read against 10.6 algorithmic; 4.23 -> 4.0 ms without the hint, profiles/r04_d1f8_temporal_ab.log)
                glds16(src8[unit == 3] + (long)t * kBK, smem + ((size_t)b * 4 + unit) * kUnitBytes + wv * 1024);
                return;
            }
        }
        if (unit == 0 || unit == 3) {      // feature rows: read once, non-temporal
            glds16_nt(src[unit][0] + (long)t * kBK, dst);
            glds16_nt(src[unit][1] + (long)t * kBK, dst + 1024);
        } else {                           // weights: re-read by every work-group, from L2
            glds16(src[unit][0] + (long)t * (kBN * kBK), dst);
            glds16(src[unit][1] + (long)t * (kBN * kBK), dst + 1024);
        }
    };

    // ---- fragment read offsets inside a unit (bytes): row u, 16-B chunk (ks*4 + fg) ^ (u & 7)
    auto frag = [&](const unsigned char* unit_base, int u, int ks) {
        return *reinterpret_cast<const bf16x8*>(unit_base + u * 128 + (((ks * 4 + fg) ^ (u & 7)) * 16));
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[4][2], a1[4][2], b0[2][2], b1[2][2];
    u32x2 ra[4][2];      // F8: a unit's raw fragments between the ds_read_b64 and the conversion

    // F8 fragment reads are issued as asm ds_read_b64, one per fragment (round 5).  Left to hipcc, pairs of them 1 KiB x k
    // apart were merged into ds_read2st64_b64 (8 of them + 16 ds_read_b64 in round 4's ISA), and a ds_read2_b64 is served
    // in four 16-lane groups over 32 banks -- not the 2 x 32 lanes over 64 banks the source swizzle was derived for:
    // SQ_LDS_BANK_CONFLICT was 0.29 of the LDS-active cycles against 0.015 in bf16 mode (profiles/r04_mfma.json).
    // Address of fragment (i, ks): row u = 64 wr + 16 i + fr (64 B each), chunk ((2 ks + fg/2) ^ ((u>>2)&3)) * 16, half
    // (fg & 1) * 8.  (u>>2)&3 = (fr>>2)&3 whatever i and wr, and 2 ks only flips bit 1 of the chunk: one per-lane
    // address for ks = 0, the same with bit 5 flipped for ks = 1, i * 1024 as the instruction's offset.
    unsigned ra_lane[2] = {0u, 0u};
    if constexpr (F8) {
        ra_lane[0] = (unsigned)((64 * wr + fr) * 64 + ((((fg >> 1) ^ ((fr >> 2) & 3)) * 16) + (fg & 1) * 8));
        ra_lane[1] = ra_lane[0] ^ 32u;
    }
    auto read_a = [&](bf16x8 (&a)[4][2], int unit, int b) {
        if (ABL == 3 && b == 1) return;
        const unsigned char* base = smem + ((size_t)b * 4 + unit) * kUnitBytes;
        if constexpr (F8) {
            const unsigned bl = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)const_cast<unsigned char*>(base);
            const unsigned ad0 = bl + ra_lane[0], ad1 = bl + ra_lane[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ra[i][0]) : "v"(ad0), "i"(i * 1024) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ra[i][1]) : "v"(ad1), "i"(i * 1024) : "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[i][ks] = frag(base, 64 * wr + 16 * i + fr, ks);
        }
    };
    auto cvt_a = [&](bf16x8 (&a)[4][2]) {      // F8 only, after the lgkmcnt wait that covers ra
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[i][ks] = cvt_e4m3x8(ra[i][ks]);
        }
    };
    auto read_b = [&](bf16x8 (&bq)[2][2], int unit, int b) {
        if (ABL == 3 && b == 1) return;
        const unsigned char* base = smem + ((size_t)b * 4 + unit) * kUnitBytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bq[j][ks] = frag(base, 32 * wc + 16 * j + fr, ks);
    };
    // one C quadrant x K = 64: rows i0..i0+3, columns j0..j0+1 of the wave's 8 x 4 fragment grid (ks inner per
    // accumulator keeps the k order of the simple kernel)
    auto quadrant = [&](const bf16x8 (&a)[4][2], const bf16x8 (&bq)[2][2], int i0, int j0) {
#ifndef D1_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bq[j][ks], acc[i0 + i][j0 + j], 0, 0, 0);
#ifndef D1_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // F8: a quadrant whose 16 MFMAs carry the conversion of eight raw fragments (32 VALU) in their issue shadow -- two
    // conversions behind every MFMA, pinned with sched_group_barrier (a block of 32 conversions in front of or behind the
    // MFMAs sits on the wave row's critical path between two barriers: measured, no gain over bf16 features).
    // OWN: ra -> a, the fragments this very quadrant multiplies (a's first fragment converted up front, fragment f + 1
    // under the MFMAs of fragment f); otherwise ra -> dst while a x bq is multiplied (the next tile's a0 in phase 3).
    auto quadrant_cv = [&](bf16x8 (&a)[4][2], const bf16x8 (&bq)[2][2], int i0, int j0, bf16x8 (&dst)[4][2], auto own) {
        constexpr bool OWN = decltype(own)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (OWN) a[0][0] = cvt_e4m3x8(ra[0][0]);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int ks = f >> 2, i = f & 3;
            if constexpr (OWN) { if (f + 1 < 8) a[(f + 1) & 3][(f + 1) >> 2] = cvt_e4m3x8(ra[(f + 1) & 3][(f + 1) >> 2]); }
            else dst[i][ks] = cvt_e4m3x8(ra[i][ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bq[j][ks], acc[i0 + i][j0 + j], 0, 0, 0);
        }
        if constexpr (OWN) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU: the first fragment
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                      // two conversions
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define D1_WAIT_BARRIER(N) do { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define D1_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    // ... and where the asm-issued E4M3 fragments (ra) are consumed: hipcc neither counts those reads nor knows that the wait
    // produces them, so the wait names them as in/out operands -- no conversion can be scheduled in front of it
#define D1_LGKM_RA() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), \
                                  "+v"(ra[2][0]), "+v"(ra[2][1]), "+v"(ra[3][0]), "+v"(ra[3][1]) :: "memory")

    // ---- prologue: tile 0 complete in buffer 0; a0 of tile 0 in registers
#pragma unroll
    for (int u = 0; u < 4; ++u) stage_unit(0, u, 0);
    D1_WAIT_BARRIER(0);
    read_a(a0, 0, 0);
    if constexpr (F8) { D1_LGKM_RA(); cvt_a(a0); }
#ifndef D1_NOSTAGGER
    if (wr == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs half a phase behind
#endif

    // Per phase: issue the LDS-DMA of one unit of tile t+1 FIRST (longest latency), then this phase's 4 or 8 fragment
    // reads, then the counted wait for the unit the NEXT phase reads, barrier, 16 MFMAs, barrier.
    // With U_p(t+1) issued in phase p of tile t, the six LDS-DMA instructions outstanding at each wait are the last
    // three units issued; vmcnt(4) retires the oldest of them:
    //   phase 0 retires U2(t) (b1, read in phase 1)      phase 1 retires U3(t) (a1, read in phase 2)
    //   phase 2 retires U0(t+1) (a0, read in phase 3)    phase 3 retires U1(t+1) (b0, read in phase 0 of t+1)
    for (int t = 0; t < kNT; ++t) {
        const int b = t & 1;
        const bool more = t + 1 < kNT;
        // phase 0: quadrant (a0, b0)
        if (more) stage_unit(t + 1, 0, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        read_b(b0, 1, b);
        if constexpr (F8) { if (more) D1_WAIT_BARRIER(2); else D1_WAIT_BARRIER(1); }
        else { if (more) D1_WAIT_BARRIER(4); else D1_WAIT_BARRIER(2); }
        D1_LGKM();
        quadrant(a0, b0, 0, 0);
        __builtin_amdgcn_s_barrier();
        // phase 1: quadrant (a0, b1)
        if (more) stage_unit(t + 1, 1, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        read_b(b1, 2, b);
        if constexpr (F8) { if (more) D1_WAIT_BARRIER(3); else D1_WAIT_BARRIER(0); }
        else { if (more) D1_WAIT_BARRIER(4); else D1_WAIT_BARRIER(0); }
        D1_LGKM();
        quadrant(a0, b1, 0, 2);
        __builtin_amdgcn_s_barrier();
        // phase 2: quadrant (a1, b1)
        if (more) stage_unit(t + 1, 2, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        read_a(a1, 3, b);
        if (more) D1_WAIT_BARRIER(4); else __builtin_amdgcn_s_barrier();
        if constexpr (F8) { D1_LGKM_RA(); quadrant_cv(a1, b1, 4, 2, a1, std::true_type{}); }
        else { D1_LGKM(); quadrant(a1, b1, 4, 2); }
        __builtin_amdgcn_s_barrier();
        // phase 3: quadrant (a1, b0); a0 of tile t+1
        if (more) stage_unit(t + 1, 3, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_a(a0, 0, b ^ 1);
        if constexpr (F8) { if (more) D1_WAIT_BARRIER(3); else __builtin_amdgcn_s_barrier(); }
        else { if (more) D1_WAIT_BARRIER(4); else __builtin_amdgcn_s_barrier(); }
        if constexpr (F8) {
            // the next tile's a0 bytes (read before the barrier above) are converted under this quadrant's MFMAs
            if (more) { D1_LGKM_RA(); quadrant_cv(a1, b0, 4, 0, a0, std::false_type{}); }
            else { D1_LGKM(); quadrant(a1, b0, 4, 0); }
        } else {
            quadrant(a1, b0, 4, 0);
            D1_LGKM();
        }
        __builtin_amdgcn_s_barrier();
    }
#ifndef D1_NOSTAGGER
    if (wr == 0) __builtin_amdgcn_s_barrier();      // match wave row 1's extra barrier
#endif
#undef D1_WAIT_BARRIER
#undef D1_LGKM
#undef D1_LGKM_RA

    d1_epilogue<HEAD>(acc, smem, row0, n, lane, wv, c1, hid, w2pack, n_out, probs, labels);
}

// (Round 3's variant with the weight fragments taken straight from L2 into registers -- bit-identical, 6.4 against 4.27 ms,
// profiles/r03_d1_wreg_ab.log -- screened nothing the one-barrier kernel does not and left the tree in round 4: HISTORY.md 4.3.)

// ------------------------------------------------------------------------------------
// Small batches (a single window, or a few): the 256-row tile above would run its 165 K-tiles on ONE CU with 1/16 or
// less of its rows in use.  Here one wave owns a 16-frame x 16-unit tile over the whole K and takes both operands
// straight from global memory in MFMA fragment order (16 B per lane per operand per MFMA; the weights stay in L2), so
// 16 x ceil(n/16) waves on as many CUs share the layer.  Same instruction (v_mfma_f32_16x16x32_bf16), same fragment
// contents and the same K order per output element as the tiled kernels: bit-identical results.
// ------------------------------------------------------------------------------------
// F8: the feature fragment is the lane's 8 E4M3 bytes (global_load_dwordx2), converted in front of its MFMA
template <bool F8>
__global__ __launch_bounds__(64) void vt_dense1_bf16_small_kernel(const unsigned short* __restrict__ feat, long n,
                                                                  const unsigned short* __restrict__ w1t,   // [165 k-tiles][256][64] bf16
                                                                  const float* __restrict__ c1, float* __restrict__ hid) {
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    const long row0 = (long)blockIdx.x * 16;
    const int col0 = blockIdx.y * 16;
    long ar = row0 + fr;
    if (ar >= n) ar = n - 1;                                  // rows past the end are computed, not stored
    using AFrag = typename std::conditional<F8, u32x2, bf16x8>::type;
    // element pointer: bf16 elements, or bytes (F8: kFeat bytes per row, 8 per fragment)
    const unsigned char* ap = reinterpret_cast<const unsigned char*>(feat) + (ar * (long)kFeat + fg * 8) * (F8 ? 1 : 2);
    const unsigned short* bp = w1t + (long)(col0 + fr) * kBK + fg * 8;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // software pipeline over blocks of kU K-tiles, two register sets: the 20 fragment loads of block i+1 are in flight
    // while the 10 MFMAs of block i issue (the MFMA chain is the same: K-tiles in ascending order into one accumulator)
    constexpr int kU = 5, kBlocks = kNT / kU;
    static_assert(kNT % kU == 0 && kBlocks % 2 == 1, "blocks: an even count in the loop, the last one drained after it");
    AFrag a0[kU][2], a1[kU][2];
    bf16x8 b0[kU][2], b1[kU][2];
    auto load = [&](AFrag (&af)[kU][2], bf16x8 (&bfr)[kU][2], int blk) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                af[u][ks] = *reinterpret_cast<const AFrag*>(ap + ((blk * kU + u) * kBK + ks * 32) * (F8 ? 1 : 2));
                bfr[u][ks] = *reinterpret_cast<const bf16x8*>(bp + (long)(blk * kU + u) * (kBN * kBK) + ks * 32);
            }
    };
    auto mma = [&](const AFrag (&af)[kU][2], const bf16x8 (&bfr)[kU][2]) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a;
                if constexpr (F8) a = cvt_e4m3x8(af[u][ks]); else a = af[u][ks];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[u][ks], acc, 0, 0, 0);
            }
    };
    load(a0, b0, 0);
    for (int blk = 0; blk < kBlocks - 1; blk += 2) {
        load(a1, b1, blk + 1);
        mma(a0, b0);
        load(a0, b0, blk + 2);
        mma(a1, b1);
    }
    mma(a0, b0);
    const int col = col0 + fr;
    const float bias = c1[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = row0 + fg * 4 + r;
        if (row < n) hid[row * kHid + col] = fmaxf(acc[r] + bias, 0.f);
    }
}

// ------------------------------------------------------------------------------------
// The same tile for the smallest batches (<= 256 frames: at most one tile per CU), fed through LDS: the per-wave kernel
// above is bound by load latency (two register sets of 20 fragment loads = 40 KB in flight per tile: 88 ns per MFMA of
// a 330-MFMA dependent chain, 29 us for one window).  Here FOUR waves serve one tile: wave w moves fragment w of every
// K-tile (A k 0..31, A k 32..63, B k 0..31, B k 32..63; 1 KiB each) by LDS-DMA into a ring of kRing K-tiles, kRing - 1
// tiles ahead -- 4 x 23 KiB in flight, no registers spent on it -- and all four run the identical MFMA chain out of LDS
// (reads of tile t+1 issued before the two MFMAs of tile t).  The fragments land in LDS in lane order and are read
// back in lane order, so they are the per-wave kernel's fragments and the chain is its chain: bit-identical.  Wave 0
// stores.  Ordering: a wave's counted vmcnt retires its own fragment of tile t+1, the barrier then covers the other
// three; a slot is refilled two barriers after its last read.  kGrp K-tiles share one wait + barrier.
// ------------------------------------------------------------------------------------
constexpr int kRing = 24;                  // K-tiles in the ring
constexpr int kGrp = 4;                    // K-tiles per step (one counted wait + one barrier per step)
constexpr size_t kCoopLds = (size_t)kRing * 4096;

// F8: a K-tile's feature fragment pair is ONE KiB (16 rows x 64 E4M3 bytes): wave 0 copies it (lane (fr, c) the 16-byte
// chunk c of row fr), wave 1 copies nothing, and a reading lane takes its 8 bytes -- k = 32 ks + 8 fg .. +7 = chunk
// 2 ks + fg/2, half fg & 1 -- with a ds_read_b64 and converts them in front of the MFMA.
template <bool F8>
__global__ __launch_bounds__(256) void vt_dense1_bf16_coop_kernel(const unsigned short* __restrict__ feat, long n,
                                                                  const unsigned short* __restrict__ w1t,   // [165 k-tiles][256][64] bf16
                                                                  const float* __restrict__ c1, float* __restrict__ hid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * 16;
    const int col0 = blockIdx.y * 16;
    long ar = row0 + fr;
    if (ar >= n) ar = n - 1;                                  // rows past the end are computed, not stored
    // this wave's fragment: A (wv = 0, 1) or B (wv = 2, 3), k half wv & 1; per-lane source as in the per-wave kernel
    // (byte pointers: F8 feature rows are kFeat BYTES long and wave 0 moves both k halves, see above)
    const unsigned char* src;
    long kstep;
    if (wv < 2) {
        if constexpr (F8) { src = reinterpret_cast<const unsigned char*>(feat) + ar * (long)kFeat + fg * 16; kstep = kBK; }
        else { src = reinterpret_cast<const unsigned char*>(feat + ar * (long)kFeat + fg * 8 + (wv & 1) * 32); kstep = 2L * kBK; }
    } else {
        src = reinterpret_cast<const unsigned char*>(w1t + (long)(col0 + fr) * kBK + fg * 8 + (wv & 1) * 32);
        kstep = 2L * kBN * kBK;
    }
    const bool copies = !(F8 && wv == 1);
    static_assert(kRing % kGrp == 0, "groups do not straddle the ring's end");
    constexpr int kAhead = kRing / kGrp - 1;                 // groups in flight
    constexpr int kGroups = kNT / kGrp, kTail = kNT - kGroups * kGrp;      // 41 full groups + 1 tile
    auto issue = [&](int g) {      // group g (tiles clamped: the tail keeps the count of outstanding copies uniform) -> its ring slots
#pragma unroll
        for (int j = 0; j < kGrp; ++j) {
            const int t = g * kGrp + j, tc = t < kNT ? t : kNT - 1;
            if (copies) glds16_async(src + tc * kstep, smem + (size_t)(t % kRing) * 4096 + wv * 1024);
        }
    };
    auto read = [&](bf16x8 (&f)[kGrp][4], int g) {
        const unsigned char* slot = smem + (size_t)((g * kGrp) % kRing) * 4096;
        const unsigned char* base = slot + lane * 16;
#pragma unroll
        for (int j = 0; j < kGrp; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (F8) {
                    if (q < 2) {      // raw bytes in the low half of the register set; converted in mma()
                        const u32x2 r = *reinterpret_cast<const u32x2*>(slot + j * 4096 + (fr + 16 * (2 * q + (fg >> 1))) * 16 + (fg & 1) * 8);
                        f[j][q] = __builtin_bit_cast(bf16x8, u32x4{r[0], r[1], 0u, 0u});
                        continue;
                    }
                }
                f[j][q] = *reinterpret_cast<const bf16x8*>(base + j * 4096 + q * 1024);
            }
    };
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    auto afrag = [&](const bf16x8& raw) {
        if constexpr (F8) { const u32x4 r = __builtin_bit_cast(u32x4, raw); return cvt_e4m3x8(u32x2{r[0], r[1]}); }
        else return raw;
    };
    auto mma = [&](const bf16x8 (&f)[kGrp][4], int tiles) {
#pragma unroll
        for (int j = 0; j < kGrp; ++j)
            if (j < tiles) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag(f[j][0]), f[j][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag(f[j][1]), f[j][3], acc, 0, 0, 0);
            }
    };
    bf16x8 f0[kGrp][4], f1[kGrp][4];
    for (int g = 0; g < kAhead; ++g) issue(g);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((kAhead - 1) * kGrp) : "memory");      // own fragments of group 0, then everybody's
    read(f0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    auto step = [&](bf16x8 (&cur)[kGrp][4], bf16x8 (&nxt)[kGrp][4], int g) {
        issue(g + kAhead);                                                       // outstanding now: groups g+1 .. g+kAhead
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((kAhead - 1) * kGrp) : "memory");   // own fragments of group g+1 landed, then everybody else's
        read(nxt, g + 1);
        mma(cur, kGrp);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    static_assert(kGroups % 2 == 1 && kTail >= 0 && kTail < kGrp, "pairs of steps, a last full group, then the tail tiles");
    for (int g = 0; g < kGroups - 1; g += 2) {
        step(f0, f1, g);
        step(f1, f0, g + 1);
    }
    step(f0, f1, kGroups - 1);         // last full group; f1 <- the tail group (tiles past the end are clamped copies, not used)
    mma(f1, kTail);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the clamped tail copies: nothing in flight at exit
    if (wv != 0) return;
    const int col = col0 + fr;
    const float bias = c1[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = row0 + fg * 4 + r;
        if (row < n) hid[row * kHid + col] = fmaxf(acc[r] + bias, 0.f);
    }
}

constexpr long kCoopBatch = 256;       // frames up to which one tile per CU is the whole layer: the four-wave form

// frames up to which the per-wave kernel is used: 16 x n/16 waves of 16 x 16 outputs beat the 256-row tiles (one
// work-group per 256 frames, 178 us whatever the batch) up to about 2,048 frames (tools/latency.py: 260 vs 296 us there)
constexpr long kSmallBatch = 2048;

}  // namespace

// probs / labels non-null and fuse_head: dense2 + softmax + argmax run in the GEMM's epilogue (*fused = true) and the hidden
// layer is not written; otherwise (small batches, a layer tap downstream, the alternates) hid is, and the caller launches the head.
int vtcnn2_bf16_dense1(const mdc_model* m, const void* feat, int64_t n, float* hid, hipStream_t s,
                       bool fuse_head, float* probs, int32_t* labels, bool* fused) {
    if (fused) *fused = false;
    const bool f8 = m->dtype == MDC_FP8 && m->fp8_e4m3_features;      // the feature matrix holds E4M3 bytes (feat8_index order)
    if (n <= kCoopBatch) {
#define MDC_LAUNCH_COOP(F) do { \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_coop_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLds)); \
        hipLaunchKernelGGL(vt_dense1_bf16_coop_kernel<F>, dim3((unsigned)((n + 15) / 16), kHid / 16), dim3(256), kCoopLds, s, \
                           static_cast<const unsigned short*>(feat), (long)n, static_cast<const unsigned short*>(m->d_pack[3]), \
                           static_cast<const float*>(m->d_pack[4]), hid); } while (0)
        if (f8) MDC_LAUNCH_COOP(true); else MDC_LAUNCH_COOP(false);
#undef MDC_LAUNCH_COOP
        MDC_HIP(hipGetLastError());
        return MDC_OK;
    }
    if (n <= kSmallBatch) {
#define MDC_LAUNCH_SMALL(F) hipLaunchKernelGGL(vt_dense1_bf16_small_kernel<F>, dim3((unsigned)((n + 15) / 16), kHid / 16), dim3(64), 0, s, \
                           static_cast<const unsigned short*>(feat), (long)n, static_cast<const unsigned short*>(m->d_pack[3]), \
                           static_cast<const float*>(m->d_pack[4]), hid)
        if (f8) MDC_LAUNCH_SMALL(true); else MDC_LAUNCH_SMALL(false);
#undef MDC_LAUNCH_SMALL
        MDC_HIP(hipGetLastError());
        return MDC_OK;
    }
    const unsigned short* f = static_cast<const unsigned short*>(feat);
    const unsigned short* w1t = static_cast<const unsigned short*>(m->d_pack[3]);
    const float* c1 = static_cast<const float*>(m->d_pack[4]);
    const dim3 grid((unsigned)((n + kBM - 1) / kBM));
#if defined(MDC_ALTERNATES) || defined(MDC_ABLATIONS)
#define MDC_LAUNCH_D1(A) do { \
    MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseBf16Lds)); \
    hipLaunchKernelGGL(vt_dense1_bf16_kernel<A>, grid, dim3(512), kDenseBf16Lds, s, f, (long)n, w1t, c1, hid); } while (0)
#endif
#ifdef MDC_ABLATIONS
    static const int abl = getenv("MDC_ABLATE_D1") ? atoi(getenv("MDC_ABLATE_D1")) : 0;
#define MDC_LAUNCH_PROBE(A) do { \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_phased_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseBf16Lds)); \
        hipLaunchKernelGGL(vt_dense1_bf16_phased_kernel<A>, grid, dim3(512), kDenseBf16Lds, s, f, (long)n, w1t, c1, hid); \
        MDC_HIP(hipGetLastError()); \
        return MDC_OK; } while (0)
    if (!f8 && abl == 10) MDC_LAUNCH_PROBE(0);      // the phased kernel, unfused (the probes' own baseline)
    if (!f8 && abl == 11) MDC_LAUNCH_PROBE(1);      // ... with its A rows from L2
    if (!f8 && abl == 12) MDC_LAUNCH_PROBE(2);      // ... with the weight units staged on even K-tiles only
    if (!f8 && abl == 13) MDC_LAUNCH_PROBE(3);      // ... with the fragment reads on even K-tiles only
#undef MDC_LAUNCH_PROBE
    if (abl >= 1 && abl <= 3) {
        switch (abl) { case 1: MDC_LAUNCH_D1(1); break; case 2: MDC_LAUNCH_D1(2); break; default: MDC_LAUNCH_D1(3); }
        MDC_HIP(hipGetLastError());
        return MDC_OK;
    }
#endif
#ifdef MDC_ALTERNATES
    if (!f8 && (m->alt & kAltDense1Simple)) {      // one barrier per K-tile (bit-identical results; the phased kernel's race screen)
        MDC_LAUNCH_D1(0);
        MDC_HIP(hipGetLastError());
        return MDC_OK;
    }
    if (m->alt & kAltSeparateHead) fuse_head = false;
#endif
#undef MDC_LAUNCH_D1
#define MDC_LAUNCH_PHASED(F) do { \
    if (fuse_head && (probs || labels)) { \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_phased_kernel<0, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseHeadLds)); \
        hipLaunchKernelGGL((vt_dense1_bf16_phased_kernel<0, true, F>), grid, dim3(512), kDenseHeadLds, s, f, (long)n, w1t, c1, hid, \
                           static_cast<const float*>(m->d_pack[5]), (int)m->topo.classes, probs, labels); \
        if (fused) *fused = true; \
    } else { \
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_phased_kernel<0, false, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseBf16Lds)); \
        hipLaunchKernelGGL((vt_dense1_bf16_phased_kernel<0, false, F>), grid, dim3(512), kDenseBf16Lds, s, f, (long)n, w1t, c1, hid); \
    } } while (0)
    if (f8) MDC_LAUNCH_PHASED(true); else MDC_LAUNCH_PHASED(false);
#undef MDC_LAUNCH_PHASED
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
