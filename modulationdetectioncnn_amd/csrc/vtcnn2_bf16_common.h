// Shared by the bf16 VT-CNN2 kernel files (vtcnn2_bf16.hip, vtcnn2_bf16_sched.hip, vtcnn2_bf16_dense1.hip):
// vector typedefs, the LDS image geometry of the conv kernels, bf16 packing helpers and the frame staging.
#pragma once
#include "mdc_internal.h"

#include <cstring>

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

// 132 padded samples per row = 66 bf16 pairs, plus one more zero pair: conv1 at an odd position v reads pairs
// i, i+1, i+2 (i = v>>1) and v = 129 touches pair 66.  Its sample only meets a zero tap, but 0 x (Inf/NaN bit
// pattern from whatever follows the image in LDS) is NaN, so the pair has to exist and hold a finite value.
constexpr int kPairs = 67;
constexpr int kImgWords = 2 * kPairs * 64;        // [row h][pair][lane]  u32
constexpr int kPartFloats = 4 * 5 * 64 * 4;       // [wave][ot][lane][4]  f32
constexpr size_t kConvBf16Lds = (size_t)2 * kImgWords * 4 + (size_t)2 * kPartFloats * 4 + 512;   // 110,080 B (+ conv2 bias)
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

// Staging of a 16-frame group: 1024 float4 = 4 per thread, done one float4 ("quarter" k) at a time so
// that the few registers it needs are live only briefly (the main loop sits at the register limit).
// Image word for lane (frame i, k-group kg): kg 0 = bf16 hi pair, kg 1 = lo pair (x - hi), kg 2 = hi pair
// again (multiplied by the low halves of the taps), kg 3 = constant (1,1) (bias slots; written once).
// The two halves of it are separate so that the asm-sequenced kernel can issue the global load several steps
// before it converts and stores the values (a load waited for right away sits behind every feature store in flight).
// U8 = true (mdc_forward_iq_u8): x points at raw interleaved unsigned 8-bit (I,Q) pairs of one contiguous capture;
// window f starts hop2 = 2*hop bytes after window f-1 (hop = 128 pairs: disjoint 256-byte frames).  The lane loads the
// 8 bytes that hold its four samples of BOTH rows (lanes l and l+32 read the same address; only 2-byte alignment is
// guaranteed -- gfx950 global loads take unaligned addresses) and stage_decode converts the row it owns with the
// arithmetic of iq_u8_kernel (eval_ops.hip), so everything downstream is bit-identical to convert-then-forward.
template <bool U8> struct StageRaw { using type = float4; };
template <> struct StageRaw<true> { using type = uint2; };
template <bool U8 = false>
__device__ __forceinline__ typename StageRaw<U8>::type stage_load(int k, const float* __restrict__ x, long n, long frame0, int tid, long hop2 = 256) {
    const int idx = tid + 256 * k;
    const long f = frame0 + (idx >> 6);
    // unconditional (clamped) load: a load under an exec mask gets its s_waitcnt vmcnt(0) right at the join
    if constexpr (U8) return load8_unaligned(reinterpret_cast<const unsigned char*>(x) + (f < n ? f : n - 1) * hop2 + (idx & 31) * 8);
    else return reinterpret_cast<const float4*>(x + (f < n ? f : n - 1) * kFrameFloats)[idx & 63];
}
template <bool U8 = false>
__device__ __forceinline__ float4 stage_decode(typename StageRaw<U8>::type r, int tid, float scale) {
    if constexpr (U8) {
        const unsigned sh = ((tid >> 5) & 1) * 8;      // row 0 = I = even bytes, row 1 = Q = odd bytes
        const unsigned a = r.x >> sh, b = r.y >> sh;
        return make_float4(((float)(a & 0xFFu) - 127.5f) * scale, ((float)((a >> 16) & 0xFFu) - 127.5f) * scale,
                           ((float)(b & 0xFFu) - 127.5f) * scale, ((float)((b >> 16) & 0xFFu) - 127.5f) * scale);
    } else {
        return r;
    }
}
__device__ __forceinline__ void stage_write(int k, float4 v, long n, long frame0, unsigned* __restrict__ im, int tid) {
    const int idx = tid + 256 * k;
    const int i = idx >> 6, l = idx & 63;
    if (frame0 + i >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);      // frames past the end of the batch are zeros
    const int h = l >> 5, m = l & 31;
    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float a = xs[2 * e], b = xs[2 * e + 1];
        const unsigned hi = pack2(a, b);
        const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xFFFF0000u);
        const unsigned lo = pack2(a - ah, b - bh);
        unsigned* d = im + (h * kPairs + 2 * m + 1 + e) * 64 + i;   // samples 4m+2e, +1 -> padded 4m+2e+2, +3
        d[0] = hi;
        d[16] = lo;
        d[32] = hi;
    }
}
template <bool U8 = false>
__device__ __forceinline__ void stage_quarter(int k, const float* __restrict__ x, long n, long frame0,
                                              unsigned* __restrict__ im, int tid, long hop2 = 256, float scale = 0.f) {
    stage_write(k, stage_decode<U8>(stage_load<U8>(k, x, n, frame0, tid, hop2), tid, scale), n, frame0, im, tid);
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {      // non-temporal source (read once)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// The same copy issued from inline asm (M0 = the wave's LDS destination through the "{m0}" constraint).  hipcc tracks a
// builtin LDS-DMA as an asynchronous LDS store and, having no alias scopes to tell ring slots apart, puts an
// s_waitcnt vmcnt(0) in front of EVERY later LDS read -- so a ring of several groups "in flight" drained completely at
// each step, counted waits or not (found in round 2 in the ISA of the deployed kernels: `s_waitcnt vmcnt(12)` followed by
// `s_waitcnt vmcnt(0)`).  With the asm form the kernel's own counted waits are the only ones; the kernel is then
// responsible for both orderings: DMA landed before its slot is read (counted vmcnt), slot read out before it is
// refilled (lgkmcnt(0) before the refill is issued).
__device__ __forceinline__ void glds16_async(const void* gsrc, void* lds_wave_base) {
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gsrc), "{m0}"(l) : "memory");
}

// host-side bf16 conversion for the operand packing
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

// launcher of the asm-sequenced conv kernel (vtcnn2_bf16_sched.hip); same arguments as vtcnn2_bf16_conv
// (hop2 > 0: x points at raw uint8 I/Q, windows hop2 bytes apart, samples (byte - 127.5) * scale)
int vtcnn2_bf16_conv_sched(const mdc_model* m, const float* x, int64_t n, void* feat, hipStream_t s, long hop2 = 0, float scale = 0.f);
// its operand packing (d_pack slots 6 and 7); called by vtcnn2_bf16_pack
int vtcnn2_bf16_pack_sched(mdc_model* m);

}  // namespace mdc
