// Canonical VT-CNN2 (T3), bf16 MFMA path (f32 accumulation).  See vtcnn2.hip for the math and
// the "lane = frame" mapping.
//
// vt_conv_bf16_kernel -- WEIGHT-STATIONARY IN REGISTERS.  The conv2 kernel tensor is
// 80 x 1536 bf16 = 240 KiB: too big for the 160 KiB LDS, but a CU's four SIMDs hold 512 KiB
// of registers.  One workgroup = 4 waves (one per SIMD, 512 VGPR+AGPR each); wave q keeps the
// conv2 weights of input channels [64q, 64q+64) for all 80 outputs, 2 rows and 3 taps:
// 60 A-fragments x 4 VGPRs = 240 registers, loaded once per kernel.  The workgroup walks
// groups of 16 frames; per group every wave sweeps the 130 conv1 positions:
//     8 x v_mfma_f32_16x16x16_bf16   conv1 of its 64 channels x 2 rows  -> X[channel][frame]
//     ReLU + v_cvt_pk_bf16_f32       X is ALREADY the B-operand layout of the next MFMA
//    60 x v_mfma_f32_16x16x32_bf16   conv2: 2 rows x 2 channel pairs x 3 taps x 5 output tiles,
//                                    tap j accumulates into the registers of output w'-j
// so activations never touch LDS and weights never move.  The only exchange is the sum of the
// four waves' K-quarter partials of ONE output position per step (5 KB each) through LDS,
// after which bias + ReLU + bf16 and the store of feat[frame][w][0..80).
// conv1's MFMA has K = 16 slots and needs 3: the spare slots carry the low-order bf16 halves
// of the input samples, of the conv1 taps and of the conv1 bias, so conv1 is computed to
// ~2^-16 relative accuracy although every operand is bf16.
//
// vt_dense1_bf16_kernel -- 256x256x64-tile bf16 GEMM (M = frames, N = 256 hidden units,
// K = 10560), LDS-DMA staging with an XOR-swizzled source (so ds_read_b128 fragments spread
// over the banks), two LDS buffers, 8 waves (2 x 4), fused bias + ReLU epilogue.
#include "mdc_internal.h"

#include <cstring>
#include <type_traits>

namespace mdc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x2 = __attribute__((ext_vector_type(2))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

namespace {

constexpr int kPairs = 66;                        // 132 padded samples per row, as bf16 pairs
constexpr int kImgWords = 2 * kPairs * 64;        // [row h][pair][lane]  u32
constexpr int kPartFloats = 4 * 5 * 64 * 4;       // [wave][ot][lane][4]  f32
constexpr size_t kConvBf16Lds = (size_t)2 * kImgWords * 4 + (size_t)2 * kPartFloats * 4;   // 108,544 B
constexpr int kWFrags = 2 * 3 * 2 * 5;            // [h][j][cp][ot] = 60 fragments per wave

__device__ __forceinline__ unsigned pack2(float a, float b) {          // two f32 -> packed bf16 (RNE)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ unsigned pack2relu(float a, float b) {      // + ReLU on the packed halves
    s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{a, b}, bf16x2));
    s = __builtin_elementwise_max(s, s16x2{0, 0});                     // negative bf16 <=> negative int16
    return __builtin_bit_cast(unsigned, s);
}
__device__ __forceinline__ float bf16_hi_as_f32(float a) {             // value of bf16(a), as f32
    return __uint_as_float(pack2(a, 0.f) << 16);
}

struct Stage {            // one thread's share of a 16-frame group: 4 float4 loads
    float4 v[4];
};

__device__ __forceinline__ void stage_load(Stage& st, const float* __restrict__ x, long n, long frame0, int tid) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k;
        const long f = frame0 + (idx >> 6);
        st.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) st.v[k] = reinterpret_cast<const float4*>(x + f * kFrameFloats)[idx & 63];
    }
}

// image word for lane (frame i, k-group kg): kg 0 = bf16 hi pair, kg 1 = lo pair (x - hi), kg 2 = hi pair
// again (multiplied by the low halves of the taps), kg 3 = constant (1,1) (bias slots; written once).
__device__ __forceinline__ void stage_write(const Stage& st, unsigned* __restrict__ im, int tid) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k;
        const int i = idx >> 6, l = idx & 63;
        const int h = l >> 5, m = l & 31;
        const float xs[4] = {st.v[k].x, st.v[k].y, st.v[k].z, st.v[k].w};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float a = xs[2 * e], b = xs[2 * e + 1];
            const unsigned hi = pack2(a, b);
            const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xFFFF0000u);
            const unsigned lo = pack2(a - ah, b - bh);
            unsigned* d = im + (h * kPairs + 2 * m + 1 + e) * 64 + i;   // samples 4m+2e, +1 -> padded 4m+2e+2
            d[0] = hi;
            d[16] = lo;
            d[32] = hi;
        }
    }
}

__global__ __launch_bounds__(256, 1) void vt_conv_bf16_kernel(const float* __restrict__ x, long n,
                                                              const u32x4* __restrict__ wq,   // [4][60][64]
                                                              const u32x2* __restrict__ a1q,  // [4][4][2][64]
                                                              const float* __restrict__ b2,   // [80]
                                                              unsigned short* __restrict__ feat) {   // [n][132][80] bf16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* img = reinterpret_cast<unsigned*>(smem);
    float* part = reinterpret_cast<float*>(smem + (size_t)2 * kImgWords * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, g = lane >> 4;

    // ---- stationary operands ----
    bf16x8 W[2][3][2][5];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int ot = 0; ot < 5; ++ot)
                    W[h][j][cp][ot] = __builtin_bit_cast(bf16x8, wq[(q * kWFrags + ((h * 3 + j) * 2 + cp) * 5 + ot) * 64 + lane]);
    s16x4 A1[4][2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int par = 0; par < 2; ++par) A1[ct][par] = __builtin_bit_cast(s16x4, a1q[((q * 4 + ct) * 2 + par) * 64 + lane]);
    const float4 bq = *reinterpret_cast<const float4*>(b2 + 16 * q + 4 * g);     // this wave reduces output tile q ...
    const float4 b4 = *reinterpret_cast<const float4*>(b2 + 64 + 4 * g);         // ... and tile 4 on its turn

    // ---- LDS init: zero padding pairs and the constant bias-slot lanes, both buffers ----
    for (int i = tid; i < 2 * kImgWords; i += 256) img[i] = ((i & 63) >= 48) ? 0x3F803F80u : 0u;
    __syncthreads();

    const long ngroups = (n + 15) >> 4;
    long grp = blockIdx.x;
    Stage st;
    if (grp < ngroups) {
        stage_load(st, x, n, grp * 16, tid);
        stage_write(st, img, tid);
    }
    __syncthreads();

    int buf = 0;
    for (; grp < ngroups; grp += gridDim.x, buf ^= 1) {
        const unsigned* im = img + buf * kImgWords + lane;
        const long frame0 = grp * 16;
        const long fme = frame0 + nl;
        const bool fvalid = fme < n;
        unsigned short* fbase = feat + fme * (long)(kW2 * kC2) + 4 * g;
        const long gnext = grp + gridDim.x;

        f32x4 acc[3][5];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 Bf[2][2];

        // conv1 at output index v (padded position v+2): pair index i = v>>1, taps start at slot v&1
        auto conv1_pack = [&](int v, auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            const int i = v >> 1;
            s16x4 bI = __builtin_bit_cast(s16x4, u32x2{im[(0 * kPairs + i) * 64], im[(0 * kPairs + i + 1) * 64]});
            s16x4 bQ = __builtin_bit_cast(s16x4, u32x2{im[(1 * kPairs + i) * 64], im[(1 * kPairs + i + 1) * 64]});
            f32x4 X[4][2];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                X[ct][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A1[ct][PAR], bI, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                X[ct][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A1[ct][PAR], bQ, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const f32x4 t0 = X[2 * cp][h], t1 = X[2 * cp + 1][h];
                    Bf[h][cp] = __builtin_bit_cast(bf16x8, u32x4{pack2relu(t0[0], t0[1]), pack2relu(t0[2], t0[3]),
                                                                 pack2relu(t1[0], t1[1]), pack2relu(t1[2], t1[3])});
                }
        };
        // conv2 of one position: tap j accumulates into the output at w'-j
        auto conv2 = [&](f32x4 (&a0)[5], f32x4 (&a1)[5], f32x4 (&a2)[5]) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                    for (int ot = 0; ot < 5; ++ot) {
                        const f32x4 c0 = (h == 0 && cp == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : a0[ot];
                        a0[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[h][0][cp][ot], Bf[h][cp], c0, 0, 0, 0);
                        a1[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[h][1][cp][ot], Bf[h][cp], a1[ot], 0, 0, 0);
                        a2[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[h][2][cp][ot], Bf[h][cp], a2[ot], 0, 0, 0);
                    }
        };
        // publish this wave's K-quarter partial of output position w, then (after the barrier)
        // finish the tiles this wave owns: sum of the 4 partials, bias, ReLU, bf16, store
        auto flush = [&](int w, const f32x4 (&a)[5]) {
            float* pw = part + (w & 1) * kPartFloats;
#pragma unroll
            for (int ot = 0; ot < 5; ++ot) *reinterpret_cast<f32x4*>(pw + ((q * 5 + ot) * 64 + lane) * 4) = a[ot];
            __syncthreads();
            auto finish = [&](int ot, const float4& bias) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(pw + ((0 * 5 + ot) * 64 + lane) * 4);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(pw + ((1 * 5 + ot) * 64 + lane) * 4);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(pw + ((2 * 5 + ot) * 64 + lane) * 4);
                const f32x4 p3 = *reinterpret_cast<const f32x4*>(pw + ((3 * 5 + ot) * 64 + lane) * 4);
                const f32x4 s = (p0 + p1) + (p2 + p3);
                u32x2 o;
                o[0] = pack2relu(s[0] + bias.x, s[1] + bias.y);
                o[1] = pack2relu(s[2] + bias.z, s[3] + bias.w);
                if (fvalid) *reinterpret_cast<u32x2*>(fbase + (long)w * kC2 + 16 * ot) = o;
            };
            finish(q, bq);
            if (q == (w & 3)) finish(4, b4);
        };

        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        conv1_pack(0, P0{});
        int v = 0;
        for (int it = 0; it < 21; ++it, v += 6) {
            if (it == 8 && gnext < ngroups) stage_load(st, x, n, gnext * 16, tid);
            if (it == 16 && gnext < ngroups) stage_write(st, img + (buf ^ 1) * kImgWords, tid);
            conv2(acc[2], acc[1], acc[0]); conv1_pack(v + 1, P1{}); flush(v + 0, acc[0]);
            conv2(acc[0], acc[2], acc[1]); conv1_pack(v + 2, P0{}); flush(v + 1, acc[1]);
            conv2(acc[1], acc[0], acc[2]); conv1_pack(v + 3, P1{}); flush(v + 2, acc[2]);
            conv2(acc[2], acc[1], acc[0]); conv1_pack(v + 4, P0{}); flush(v + 3, acc[0]);
            conv2(acc[0], acc[2], acc[1]); conv1_pack(v + 5, P1{}); flush(v + 4, acc[1]);
            conv2(acc[1], acc[0], acc[2]); conv1_pack(v + 6, P0{}); flush(v + 5, acc[2]);
        }
        // v = 126..129 (conv1_pack(126) already done), then the two outputs fed only by padding beyond
        conv2(acc[2], acc[1], acc[0]); conv1_pack(127, P1{}); flush(126, acc[0]);
        conv2(acc[0], acc[2], acc[1]); conv1_pack(128, P0{}); flush(127, acc[1]);
        conv2(acc[1], acc[0], acc[2]); conv1_pack(129, P1{}); flush(128, acc[2]);
        conv2(acc[2], acc[1], acc[0]);                         flush(129, acc[0]);
        flush(130, acc[1]);
        flush(131, acc[2]);
        __syncthreads();      // next group's image is complete; partial buffers are free again
    }
}

// ------------------------------------------------------------------------------------
// dense1 bf16 GEMM
// ------------------------------------------------------------------------------------
constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kTileBytes = kBM * kBK * 2;                     // 32 KiB per operand tile
constexpr size_t kDenseBf16Lds = (size_t)4 * kTileBytes;      // A,B x 2 buffers = 128 KiB
constexpr int kNT = kFeat / kBK;                              // 165 K-tiles

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(512) void vt_dense1_bf16_kernel(const unsigned short* __restrict__ feat, long n,
                                                             const unsigned short* __restrict__ w1t,   // [256][10560] bf16
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
        long gr = row0 + r;
        if (gr >= n) gr = n - 1;                            // clamp: rows past the end are computed, not stored
        asrc[p] = feat + gr * (long)kFeat + ((spos ^ (r & 7)) * 8);
        bsrc[p] = w1t + (long)r * kFeat + ((spos ^ (r & 7)) * 8);
    }
    auto stage = [&](int t, int b) {
        unsigned char* A = smem + (size_t)b * 2 * kTileBytes;
        unsigned char* B = A + kTileBytes;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            glds16(asrc[p] + t * kBK, A + (wv * 4 + p) * 1024);
            glds16(bsrc[p] + t * kBK, B + (wv * 4 + p) * 1024);
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

inline unsigned short f2bf(float f) {          // host RNE f32 -> bf16
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline float bf2f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

}  // namespace

// d_pack slots for bf16: 0 conv2 fragments, 1 conv1 fragments, 3 dense1 weights transposed+permuted
int vtcnn2_bf16_pack(mdc_model* m) {
    const float* k1 = m->hk[0].data();   // (256,1,1,3)
    const float* b1 = m->hb[0].data();
    const float* k2 = m->hk[1].data();   // (80,256,2,3)
    int rc;
    // conv2 A-fragments: [q][((h*3+j)*2+cp)*5+ot][lane][8]; lane (o' = lane&15, g = lane>>4) slot jj holds
    // K2[16ot+o'][64q + 32cp + (jj<4 ? 4g+jj : 16+4g+jj-4)][h][j]   (the channel order X arrives in)
    std::vector<unsigned short> wq((size_t)4 * kWFrags * 64 * 8);
    for (int q = 0; q < 4; ++q)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 3; ++j)
                for (int cp = 0; cp < 2; ++cp)
                    for (int ot = 0; ot < 5; ++ot)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int o = 16 * ot + (lane & 15), g = lane >> 4;
                                const int ch = 64 * q + 32 * cp + (jj < 4 ? 4 * g + jj : 16 + 4 * g + (jj - 4));
                                const size_t idx = ((((size_t)q * kWFrags + ((h * 3 + j) * 2 + cp) * 5 + ot) * 64) + lane) * 8 + jj;
                                wq[idx] = f2bf(k2[(((size_t)o * kC1 + ch) * 2 + h) * 3 + j]);
                            }
    if ((rc = upload(m, 0, wq.data(), wq.size() * 2))) return rc;
    // conv1 A-fragments (K = 16): [q][ct][par][lane][4]; lane (c = lane&15, kg = lane>>4):
    //   kg 0: tap hi at slots par..par+2 (x hi)   kg 1: tap hi (x lo)   kg 2: tap lo (x hi)   kg 3: (b1 hi, b1 lo, 0, 0)
    std::vector<unsigned short> a1((size_t)4 * 4 * 2 * 64 * 4, 0);
    for (int q = 0; q < 4; ++q)
        for (int ct = 0; ct < 4; ++ct)
            for (int par = 0; par < 2; ++par)
                for (int lane = 0; lane < 64; ++lane) {
                    const int ch = 64 * q + 16 * ct + (lane & 15), kg = lane >> 4;
                    unsigned short* d = &a1[((((size_t)q * 4 + ct) * 2 + par) * 64 + lane) * 4];
                    if (kg == 3) {
                        const unsigned short hi = f2bf(b1[ch]);
                        d[0] = hi;
                        d[1] = f2bf(b1[ch] - bf2f(hi));
                    } else {
                        for (int t = 0; t < 3; ++t) {
                            const float kv = k1[ch * 3 + t];
                            const unsigned short hi = f2bf(kv);
                            d[par + t] = (kg == 2) ? f2bf(kv - bf2f(hi)) : hi;
                        }
                    }
                }
    if ((rc = upload(m, 1, a1.data(), a1.size() * 2))) return rc;
    // dense1: transposed [n][k'] with k' = w*80 + o  <-  reference row o*132 + w
    const float* w1 = m->hk[2].data();
    std::vector<unsigned short> w1t((size_t)kHid * kFeat);
    for (int w = 0; w < kW2; ++w)
        for (int o = 0; o < kC2; ++o) {
            const float* src = w1 + (size_t)(o * kW2 + w) * kHid;
            for (int nn = 0; nn < kHid; ++nn) w1t[(size_t)nn * kFeat + (w * kC2 + o)] = f2bf(src[nn]);
        }
    return upload(m, 3, w1t.data(), w1t.size() * 2);
}

int vtcnn2_bf16_conv(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_conv_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvBf16Lds));
        attr_set = true;
    }
    const long ngroups = (n + 15) / 16;
    const unsigned grid = (unsigned)(ngroups < 256 ? ngroups : 256);
    hipLaunchKernelGGL(vt_conv_bf16_kernel, dim3(grid), dim3(256), kConvBf16Lds, s, x, (long)n,
                       static_cast<const u32x4*>(m->d_pack[0]), static_cast<const u32x2*>(m->d_pack[1]),
                       static_cast<const float*>(m->d_pack[2]), static_cast<unsigned short*>(feat));
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

int vtcnn2_bf16_dense1(const mdc_model* m, const void* feat, int64_t n, float* hid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        MDC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vt_dense1_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseBf16Lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(vt_dense1_bf16_kernel, dim3((unsigned)((n + kBM - 1) / kBM)), dim3(512), kDenseBf16Lds, s,
                       static_cast<const unsigned short*>(feat), (long)n, static_cast<const unsigned short*>(m->d_pack[3]),
                       static_cast<const float*>(m->d_pack[4]), hid);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
