// Two waves on one SIMD: wave A issues back-to-back fp8 MFMAs (16x16x128), wave B only VALU.  How much VALU issue does
// the SIMD have left beside a saturated matrix pipe?  (A single wave gets 2 free VALU per 32-cycle MFMA gap; the question
// is whether a partner wave gets more.)  Waves i and i+4 of a 512-thread workgroup share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x8 = __attribute__((ext_vector_type(8))) unsigned;

template <int MODE>   // 0: A and B together, 1: A alone (B idle), 2: B alone (A idle)
__global__ __launch_bounds__(512, 1) void k(unsigned long long* out, float* sink, int b_iters) {
    const int wv = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    if (wv < 4) {
        if (MODE != 2) {
            u32x8 a = u32x8{0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u}, b = a;
            unsigned one = 0x7F7F7F7Fu;
            asm volatile("" : "+v"(a), "+v"(b), "+v"(one));
            f32x4 acc[5];
            for (int i = 0; i < 5; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < 2000; ++it)
#pragma unroll
                for (int m = 0; m < 20; ++m)
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m % 5]) : "v"(a), "v"(b), "v"(one));
            t1 = __builtin_readcyclecounter();
            float keep = 0.f;
            for (int i = 0; i < 5; ++i) keep += acc[i][0];
            if (keep == 123.456f) sink[0] = keep;
        }
    } else {
        if (MODE != 1) {
            float v[8];
            for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; asm volatile("" : "+v"(v[i])); }
            float s = 1.0001f;
            asm volatile("" : "+v"(s));
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < b_iters; ++it)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(s));
            t1 = __builtin_readcyclecounter();
            float keep = 0.f;
            for (int i = 0; i < 8; ++i) keep += v[i];
            if (keep == 123.456f) sink[1] = keep;
        }
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}

// Two (or one) waves per SIMD, each running the product's mix: one fp8 MFMA followed by NV VALU, back to back.
// Is the VALU cost behind an MFMA a per-wave or a per-SIMD budget?
template <int NV, bool TWO>
__global__ __launch_bounds__(512, 1) void mix(unsigned long long* out, float* sink) {
    const int wv = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    if (TWO || wv < 4) {
        u32x8 a = u32x8{0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u}, b = a;
        unsigned one = 0x7F7F7F7Fu;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(one));
        f32x4 acc[5];
        for (int i = 0; i < 5; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
        float v[8];
        for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; asm volatile("" : "+v"(v[i])); }
        float s = 1.0001f;
        asm volatile("" : "+v"(s));
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 2000; ++it)
#pragma unroll
            for (int m = 0; m < 20; ++m) {
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m % 5]) : "v"(a), "v"(b), "v"(one));
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(s));
            }
        t1 = __builtin_readcyclecounter();
        float keep = 0.f;
        for (int i = 0; i < 5; ++i) keep += acc[i][0];
        for (int i = 0; i < 8; ++i) keep += v[i];
        if (keep == 123.456f) sink[0] = keep;
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}
template <int NV, bool TWO>
static void run_mix(unsigned long long* d, float* sink) {
    hipLaunchKernelGGL((mix<NV, TWO>), dim3(256), dim3(512), 0, 0, d, sink);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<NV, TWO>), dim3(256), dim3(512), 0, 0, d, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0;
    const int nw = TWO ? 8 : 4;
    printf("[wall %.3f ms = %.2f ns per MFMA per SIMD] ", ms, ms * 1e6 / (40000.0 * (TWO ? 2 : 1)));
    for (int i = 0; i < 256; ++i) for (int w = 0; w < nw; ++w) a += (double)h[i * 8 + w];
    a /= 256.0 * nw;
    printf("mix: MFMA + %d VALU, %d wave(s)/SIMD: %7.2f cycles per MFMA per wave = %7.2f per MFMA per SIMD\n", NV, TWO ? 2 : 1, a / 40000.0, a / 40000.0 / (TWO ? 2 : 1));
}

template <int MODE>
static void run(const char* name, unsigned long long* d, float* sink, int b_iters) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, sink, b_iters);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, sink, b_iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) { for (int w = 0; w < 4; ++w) a += (double)h[i * 8 + w]; for (int w = 4; w < 8; ++w) b += (double)h[i * 8 + w]; }
    a /= 1024; b /= 1024;
    printf("%-22s  A: %7.2f cycles per MFMA    B: %6.2f cycles per VALU (%d VALU)\n", name, a / 40000.0, b_iters ? b / (16.0 * b_iters) : 0.0, 16 * b_iters);
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 64);
    run<1>("MFMA wave alone", d, sink, 0);
    run<2>("VALU wave alone", d, sink, 10000);
    for (int bi : {2500, 5000, 10000, 15000, 20000}) run<0>("both on one SIMD", d, sink, bi);
    run_mix<0, false>(d, sink); run_mix<0, true>(d, sink);
    run_mix<2, false>(d, sink); run_mix<2, true>(d, sink);
    run_mix<4, false>(d, sink); run_mix<4, true>(d, sink);
    run_mix<5, false>(d, sink); run_mix<5, true>(d, sink);
    run_mix<6, false>(d, sink); run_mix<6, true>(d, sink);
    return 0;
}
