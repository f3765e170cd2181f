// Microbenchmark: what does ONE wave per SIMD pay for a filler instruction placed between back-to-back
// v_mfma_f32_16x16x32_bf16 (accumulators in AGPRs, as in vt_conv_bf16_sched_kernel)?
// Every CU runs a 4-wave workgroup (the chip is loaded as in the product, so the clock is the product's);
// wave 0 of each workgroup times its loop with s_memtime.  Output: cycles per MFMA for each filler.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_gap.hip -o tools/microbench/mfma_gap && ./mfma_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kIters = 2000;
constexpr int kMfmaPerIter = 20;

template <int F, bool A_IN_AGPR, int SMALL = 0>
__global__ __launch_bounds__(256, 1) void gap_kernel(unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
    __syncthreads();
    f32x4 acc[5];
    for (int i = 0; i < 5; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
    u32x4 wa = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b = u32x4{(unsigned)lane, 0u, 0u, 0u};
    asm volatile("" : "+v"(b));
    u32x4 wv = wa;
    asm volatile("" : "+a"(wa));
    asm volatile("" : "+v"(wv));
    using u32x8 = __attribute__((ext_vector_type(8))) unsigned;
    u32x8 w8 = u32x8{0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u}, b8 = w8;
    unsigned one = 0x7F7F7F7Fu;
    asm volatile("" : "+v"(w8), "+v"(b8), "+v"(one));
    u32x8 w8a = w8, b8a = w8;
    u32x4 ba = b;
    asm volatile("" : "+a"(w8a), "+a"(b8a), "+a"(ba));
    f32x4 xs[8];
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    f32x16 xl[2];
    f32x2 a16 = f32x2{1.f, 1.f}, b16 = f32x2{1.f, 1.f};
    asm volatile("" : "+v"(a16), "+v"(b16));
    f32x2 p0 = f32x2{1.f, 2.f}, p1 = f32x2{3.f, 4.f}, p2, p3;
    float s0 = 1.f, s1 = 2.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f;
    unsigned c0 = 0, c1 = 0, u0 = 0, u1 = 0, u2 = 0;
    f32x4 r0 = f32x4{0.f, 0.f, 0.f, 0.f};
    float r1 = 0.f;
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(s0), "+v"(s1));
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + lane * 16;
    float* gp = sink + (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
    const unsigned lds_addr4 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + lane * 4;
    unsigned long long u64a = 0, u64b = threadIdx.x;
    const float* sbase = sink + (size_t)blockIdx.x * 4096;       // uniform -> SGPR pair
    // transposed layout of the product: quad of lanes = 32 (8) contiguous bytes, quads 21 KB apart is not reproducible
    // in this small sink, so quads sit 64 B apart
    const unsigned lds_addr8 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + lane * 8;
    const unsigned voff3 = (lane >> 2) * 64 + (lane & 2) * 2 + (threadIdx.x >> 6) * 1024;      // lane pairs write the same dword
    const unsigned voff4 = lane * 16 + (threadIdx.x >> 6) * 1024;
    const unsigned voff = (lane >> 2) * 64 + (lane & 3) * 8 + (threadIdx.x >> 6) * 1024, voff2 = (lane >> 2) * 64 + (lane & 3) * 2 + (threadIdx.x >> 6) * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int m = 0; m < kMfmaPerIter; ++m) {
            if (SMALL == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 5]) : "a"(wa), "a"(ba));
            else if (SMALL == 5) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m % 5]) : "a"(w8a), "a"(b8a), "v"(one));
            else if (SMALL == 3) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m % 5]) : "v"(w8), "v"(b8), "v"(one));
            else if (SMALL == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(xl[m % 2]) : "v"(wv), "v"(b));
            else if (SMALL) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0" : "=&v"(xs[m % 8]) : "v"(a16), "v"(b16));
            else if (A_IN_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 5]) : "a"(wa), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 5]) : "v"(wv), "v"(b));
            if (F == 1 || F == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2) : "v"(p0), "v"(p1));
            if (F == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p3) : "v"(p0), "v"(p1));
            if (F == 3 || F == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c0) : "v"(s0), "v"(s1));
            if (F == 4) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(c0));
            if (F == 5 || F == 6 || F == 7) asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1));
            if (F == 6 || F == 7) asm volatile("v_add_f32 %0, %1, %2" : "=v"(s3) : "v"(s0), "v"(s1));
            if (F == 7) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s4) : "v"(s0), "v"(s1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s5) : "v"(s0), "v"(s1)); }
            if (F == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory");
            if (F == 9) asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(lds_addr) : "memory");
            if (F == 10 && (m & 3) == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "a"(acc[(m + 2) % 5]) : "memory");
            if (F == 11 && (m & 3) == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(wv) : "memory");
            if (F == 12 && m == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gp), "v"(p0) : "memory");
            if (F == 13 && m == 0) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(gp), "v"(p0) : "memory");
                                     asm volatile("global_store_short %0, %1, off offset:128" ::"v"(gp), "v"(c1) : "memory"); }
            if (F == 14) asm volatile("s_nop 0");
            if (F >= 40 && F < 50) {      // N one-VGPR-source VALU: v_mov (40+N) ; v_med3 with inline constants (45+N)
                constexpr int N = F >= 45 ? F - 45 : F - 40;
                if (F < 45) { if (N > 0) asm volatile("v_mov_b32 %0, %1" : "=v"(c0) : "v"(c1)); if (N > 1) asm volatile("v_mov_b32 %0, %1" : "=v"(s2) : "v"(s0));
                              if (N > 2) asm volatile("v_mov_b32 %0, %1" : "=v"(s3) : "v"(s1)); if (N > 3) asm volatile("v_mov_b32 %0, %1" : "=v"(s4) : "v"(s0)); }
                else { if (N > 0) asm volatile("v_med3_f32 %0, %1, 0, 4.0" : "=v"(s2) : "v"(s0)); if (N > 1) asm volatile("v_med3_f32 %0, %1, 0, 4.0" : "=v"(s3) : "v"(s1));
                       if (N > 2) asm volatile("v_med3_f32 %0, %1, 0, 4.0" : "=v"(s4) : "v"(s0)); if (N > 3) asm volatile("v_med3_f32 %0, %1, 0, 4.0" : "=v"(s5) : "v"(s1)); }
            }
            if (F >= 50 && F < 55) {      // N three-VGPR-source VALU (v_fma_f32)
                constexpr int N = F - 50;
                if (N > 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s2) : "v"(s0), "v"(s1), "v"(r1));
                if (N > 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s3) : "v"(s0), "v"(s1), "v"(r1));
                if (N > 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s4) : "v"(s0), "v"(s1), "v"(r1));
                if (N > 3) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s5) : "v"(s0), "v"(s1), "v"(r1));
            }
            // ---- round 4: what the fp8 conv step's gaps hold.  60+N: N x v_cvt_scalef32_pk_fp8_bf16; 65+N: N x v_cvt_pk_bf16_f32 clamp;
            // 70 / 71: three v_add_f32 then a ds_read_b128 / the read first; 72 / 73: three VALU then 2 reads / the two reads first;
            // 74 / 75: a ds_write_b128(agpr) behind / ahead of three v_add_f32; 76: 2 cvt_bf16 + 1 cvt_fp8 (a pack gap)
            if (F >= 60 && F < 65) {
                constexpr int N = F - 60;
                if (N > 0) asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(c0) : "v"(c1), "v"(s0));
                if (N > 1) asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(u0) : "v"(c1), "v"(s0));
                if (N > 2) asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(u1) : "v"(c1), "v"(s0));
                if (N > 3) asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(u2) : "v"(c1), "v"(s0));
            }
            if (F >= 65 && F < 70) {
                constexpr int N = F - 65;
                if (N > 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(c0) : "v"(s0), "v"(s1));
                if (N > 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(u0) : "v"(s0), "v"(s1));
                if (N > 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(u1) : "v"(s0), "v"(s1));
                if (N > 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(u2) : "v"(s0), "v"(s1));
            }
#define G3ADD() do { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s3) : "v"(s0), "v"(s1)); \
                     asm volatile("v_add_f32 %0, %1, %2" : "=v"(s4) : "v"(s0), "v"(s1)); } while (0)
            if (F == 70) { G3ADD(); asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory"); }
            if (F == 71) { asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory"); G3ADD(); }
            if (F == 72) { G3ADD(); asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory"); asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(lds_addr4) : "memory"); }
            if (F == 73) { asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory"); asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(lds_addr4) : "memory"); G3ADD(); }
            if (F == 74) { G3ADD(); asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "a"(acc[(m + 2) % 5]) : "memory"); }
            if (F == 75) { asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "a"(acc[(m + 2) % 5]) : "memory"); G3ADD(); }
            if (F == 76) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(u0) : "v"(s0), "v"(s1)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(u1) : "v"(s0), "v"(s1));
                           asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(c0) : "v"(c1), "v"(s0)); }
            if (F == 28) { asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory"); asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(lds_addr4) : "memory"); }
            if (F == 29 && m == 0) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(p0), "s"(sbase) : "memory"); }
            if (F == 30) asm volatile("ds_read_b64 %0, %1" : "=v"(p2) : "v"(lds_addr8) : "memory");
            if (F == 31) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(c0) : "v"(c1));
            if (F == 32 && m == 0) { asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(p0), "s"(sbase) : "memory");
                                     asm volatile("global_store_dword %0, %1, %2 offset:128" ::"v"(voff3), "v"(c1), "s"(sbase) : "memory"); }
            if (F == 33) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s3) : "v"(s0), "v"(s1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s4) : "v"(s0), "v"(s1)); }
            if (F == 34 && m == 0) asm volatile("global_store_short %0, %1, %2 offset:128" ::"v"(voff2), "v"(c1), "s"(sbase) : "memory");
            if (F == 35 && m == 0) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff4), "v"(b), "s"(sbase) : "memory");
            if (F == 15 && m == 10) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (F == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(c0) : "v"(c1));
            if (F == 17) asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(lds_addr4) : "memory");
            if (F == 18) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(u64a) : "v"(u64b), "s"(sbase));
            if (F == 19) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(c0) : "v"(c1), "v"(b[0]));
            if (F == 20) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(c0) : "v"(c1));
            if (F == 21) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(lds_addr) : "memory"); }
            if (F == 22) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); if ((m & 3) == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "a"(acc[(m + 2) % 5]) : "memory"); }
            if (F == 23 && m == 0) { asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(p0), "s"(sbase) : "memory");
                                     asm volatile("global_store_short %0, %1, %2 offset:128" ::"v"(voff2), "v"(c1), "s"(sbase) : "memory"); }
            if (F == 24) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(lds_addr4) : "memory"); }
            if (F == 25 && (m & 3) == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(p0) : "memory");
            if (F == 26) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c0) : "v"(s0), "v"(s1)); }
            if (F == 27 && (m & 1) == 0) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s0), "v"(s1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s3) : "v"(s0), "v"(s1)); }
        }
        if (F == 8 || F == 9 || F == 10 || F == 11 || F == 17 || F == 21 || F == 22 || F == 24 || F == 25 || F == 28 || F == 30 || (F >= 70 && F <= 75)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("" ::"v"(p2), "v"(p3), "v"(c0), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(r0), "v"(r1), "v"(u64a), "v"(u0), "v"(u1), "v"(u2));
    float keep = 0.f;
    for (int i = 0; i < 5; ++i) keep += acc[i][0];
    if (SMALL == 1) for (int i = 0; i < 8; ++i) { asm volatile("s_nop 7" : "+v"(xs[i])); keep += xs[i][0]; }
    if (SMALL == 2) for (int i = 0; i < 2; ++i) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(xl[i])); keep += xl[i][0]; }
    if (keep == 123.456f) sink[0] = keep;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int F, bool A, int SMALL = 0>
static void run(const char* name, unsigned long long* d_out, float* d_sink) {
    hipLaunchKernelGGL((gap_kernel<F, A, SMALL>), dim3(256), dim3(256), 0, 0, d_out, d_sink);      // warm
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((gap_kernel<F, A, SMALL>), dim3(256), dim3(256), 0, 0, d_out, d_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d_out, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= 256;
    const double n = (double)kIters * kMfmaPerIter;
    // s_memtime counts at a fixed 100 MHz-class reference on some parts; report both it and the wall time
    printf("%-44s A=%s  memtime/MFMA %7.3f   wall ns/MFMA %7.3f\n", name, A ? "agpr" : "vgpr", c / n, ms * 1e6 / n);
}

int main() {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 256 * 8); hipMalloc(&d_sink, (size_t)256 * 4096 * 4 + 4096);
    run<0, true>("none", d_out, d_sink);
    run<0, false>("none", d_out, d_sink);
    run<14, true>("s_nop 0 per gap", d_out, d_sink);
    run<16, true>("1 v_mov_b32 per gap", d_out, d_sink);
    run<5, true>("1 v_add_f32 per gap", d_out, d_sink);
    run<6, true>("2 v_add_f32 per gap", d_out, d_sink);
    run<7, true>("4 v_add_f32 per gap", d_out, d_sink);
    run<1, true>("1 v_pk_add_f32 per gap", d_out, d_sink);
    run<1, false>("1 v_pk_add_f32 per gap", d_out, d_sink);
    run<2, true>("2 v_pk_add_f32 per gap", d_out, d_sink);
    run<3, true>("1 v_cvt_pk_bf16_f32 per gap", d_out, d_sink);
    run<4, true>("cvt_pk + pk_max per gap", d_out, d_sink);
    run<8, true>("1 ds_read_b128 per gap", d_out, d_sink);
    run<9, true>("1 ds_read_b32 per gap", d_out, d_sink);
    run<10, true>("ds_write_b128(agpr) every 4th gap", d_out, d_sink);
    run<11, true>("ds_write_b128(vgpr) every 4th gap", d_out, d_sink);
    run<12, true>("1 global_store_dwordx2 per 20 MFMAs", d_out, d_sink);
    run<13, true>("store x2 + short per 20 MFMAs", d_out, d_sink);
    run<15, true>("lgkmcnt(0)+s_barrier per 20 MFMAs", d_out, d_sink);
    run<17, true>("1 ds_read_b32 (conflict-free) per gap", d_out, d_sink);
    run<18, true>("1 v_lshl_add_u64 per gap", d_out, d_sink);
    run<19, true>("1 v_alignbit_b32 per gap", d_out, d_sink);
    run<20, true>("1 v_pk_max_i16 per gap", d_out, d_sink);
    run<21, true>("v_add_f32 + ds_read_b128 per gap", d_out, d_sink);
    run<24, true>("v_add_f32 + ds_read_b32 per gap", d_out, d_sink);
    run<22, true>("v_add_f32 per gap + ds_write_b128 every 4th", d_out, d_sink);
    run<25, true>("ds_write_b64 every 4th gap", d_out, d_sink);
    run<23, true>("saddr store x2 + short per 20 MFMAs", d_out, d_sink);
    run<26, true>("v_add_f32 + v_cvt_pk per gap", d_out, d_sink);
    run<27, true>("2 v_add_f32 every 2nd gap", d_out, d_sink);
    run<33, true>("3 v_add_f32 per gap", d_out, d_sink);
    run<28, true>("ds_read_b128 + ds_read_b32 per gap", d_out, d_sink);
    run<30, true>("1 ds_read_b64 per gap", d_out, d_sink);
    run<31, true>("1 v_mov_b32_dpp per gap", d_out, d_sink);
    run<29, true>("v_add + store x2 in one gap per 20 MFMAs", d_out, d_sink);
    run<34, true>("store short alone per 20 MFMAs", d_out, d_sink);
    run<32, true>("store x2 + dword(dup lanes) per 20 MFMAs", d_out, d_sink);
    run<35, true>("store dwordx4 per 20 MFMAs", d_out, d_sink);
    run<0, true, 1>("16x16x16: none", d_out, d_sink);
    run<5, true, 1>("16x16x16: 1 v_add_f32 per gap", d_out, d_sink);
    run<6, true, 1>("16x16x16: 2 v_add_f32 per gap", d_out, d_sink);
    run<20, true, 1>("16x16x16: 1 v_pk_max_i16 per gap", d_out, d_sink);
    run<8, true, 1>("16x16x16: 1 ds_read_b128 per gap", d_out, d_sink);
    run<0, true, 2>("32x32x16 (VGPR dst): none", d_out, d_sink);
    run<5, true, 2>("32x32x16: 1 v_add_f32 per gap", d_out, d_sink);
    run<6, true, 2>("32x32x16: 2 v_add_f32 per gap", d_out, d_sink);
    run<7, true, 2>("32x32x16: 4 v_add_f32 per gap", d_out, d_sink);
    run<8, true, 2>("32x32x16: 1 ds_read_b128 per gap", d_out, d_sink);
    run<21, true, 2>("32x32x16: v_add + ds_read_b128 per gap", d_out, d_sink);
    run<0, true, 3>("f8 16x16x128: none", d_out, d_sink);
    run<5, true, 3>("f8 16x16x128: 1 v_add_f32 per gap", d_out, d_sink);
    run<6, true, 3>("f8 16x16x128: 2 v_add_f32 per gap", d_out, d_sink);
    run<7, true, 3>("f8 16x16x128: 4 v_add_f32 per gap", d_out, d_sink);
    run<8, true, 3>("f8 16x16x128: 1 ds_read_b128 per gap", d_out, d_sink);
    run<21, true, 3>("f8 16x16x128: v_add + ds_read_b128 per gap", d_out, d_sink);
    run<4, true, 3>("f8 16x16x128: cvt_pk + pk_max per gap", d_out, d_sink);
    // does it matter for the free VALU slots which register file the A/B operands come from?
    run<6, false>("2 v_add_f32 per gap", d_out, d_sink);
    run<33, false>("3 v_add_f32 per gap", d_out, d_sink);
    run<0, true, 4>("bf16 A,B,C all AGPR: none", d_out, d_sink);
    run<5, true, 4>("bf16 A,B,C all AGPR: 1 v_add_f32", d_out, d_sink);
    run<6, true, 4>("bf16 A,B,C all AGPR: 2 v_add_f32", d_out, d_sink);
    run<33, true, 4>("bf16 A,B,C all AGPR: 3 v_add_f32", d_out, d_sink);
    run<7, true, 4>("bf16 A,B,C all AGPR: 4 v_add_f32", d_out, d_sink);
    run<0, true, 5>("f8 A,B,C all AGPR: none", d_out, d_sink);
    run<6, true, 5>("f8 A,B,C all AGPR: 2 v_add_f32", d_out, d_sink);
    run<33, true, 5>("f8 A,B,C all AGPR: 3 v_add_f32", d_out, d_sink);
    run<7, true, 5>("f8 A,B,C all AGPR: 4 v_add_f32", d_out, d_sink);
    run<33, true, 3>("f8 16x16x128: 3 v_add_f32 per gap", d_out, d_sink);
    run<42, true>("2 v_mov_b32 per gap", d_out, d_sink);
    run<43, true>("3 v_mov_b32 per gap", d_out, d_sink);
    run<44, true>("4 v_mov_b32 per gap", d_out, d_sink);
    run<47, true>("2 v_med3 (1 vgpr src) per gap", d_out, d_sink);
    run<48, true>("3 v_med3 (1 vgpr src) per gap", d_out, d_sink);
    run<52, true>("2 v_fma_f32 (3 vgpr src) per gap", d_out, d_sink);
    run<53, true>("3 v_fma_f32 (3 vgpr src) per gap", d_out, d_sink);
    run<44, true, 3>("f8: 4 v_mov_b32 per gap", d_out, d_sink);
    run<49, true, 3>("f8: 4 v_med3 (1 vgpr src) per gap", d_out, d_sink);
    run<54, true, 3>("f8: 4 v_fma_f32 per gap", d_out, d_sink);
    // ---- round 4: the fp8 conv step's gap contents (32-cycle fp8 MFMA gaps)
    run<61, true, 3>("f8: 1 v_cvt_scalef32_pk_fp8_bf16 per gap", d_out, d_sink);
    run<62, true, 3>("f8: 2 v_cvt_scalef32_pk_fp8_bf16 per gap", d_out, d_sink);
    run<63, true, 3>("f8: 3 v_cvt_scalef32_pk_fp8_bf16 per gap", d_out, d_sink);
    run<64, true, 3>("f8: 4 v_cvt_scalef32_pk_fp8_bf16 per gap", d_out, d_sink);
    run<66, true, 3>("f8: 1 v_cvt_pk_bf16_f32 clamp per gap", d_out, d_sink);
    run<67, true, 3>("f8: 2 v_cvt_pk_bf16_f32 clamp per gap", d_out, d_sink);
    run<68, true, 3>("f8: 3 v_cvt_pk_bf16_f32 clamp per gap", d_out, d_sink);
    run<69, true, 3>("f8: 4 v_cvt_pk_bf16_f32 clamp per gap", d_out, d_sink);
    run<76, true, 3>("f8: 2 cvt_bf16 + 1 cvt_fp8 per gap", d_out, d_sink);
    run<70, true, 3>("f8: 3 v_add THEN ds_read_b128 per gap", d_out, d_sink);
    run<71, true, 3>("f8: ds_read_b128 THEN 3 v_add per gap", d_out, d_sink);
    run<72, true, 3>("f8: 3 v_add THEN b128 + b32 reads per gap", d_out, d_sink);
    run<73, true, 3>("f8: b128 + b32 reads THEN 3 v_add per gap", d_out, d_sink);
    run<74, true, 3>("f8: 3 v_add THEN ds_write_b128 per gap", d_out, d_sink);
    run<75, true, 3>("f8: ds_write_b128 THEN 3 v_add per gap", d_out, d_sink);
    return 0;
}
