// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_max_f32) for W waves per SIMD, no MFMA around.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int kIters = 4000;

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink) {
    float a[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{a[i], a[i] + 1.f}; asm volatile("" : "+v"(a[i]), "+v"(p[i])); }
    float s = 1.0001f, t = 0.5f;
    f32x2 s2 = f32x2{1.0001f, 1.0002f}, t2 = f32x2{0.5f, 0.25f};
    asm volatile("" : "+v"(s), "+v"(t), "+v"(s2), "+v"(t2));
    f32x2 ks = f32x2{1.0001f, 0.9999f};
    asm volatile("" : "+s"(ks));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i][0]) : "v"(s), "v"(t)); }
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(s2), "v"(t2));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(s2), "v"(t2));
            if (MODE == 3) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(p[i][0]) : "v"(t)); }
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            // the forms the deployed bf16 kernel uses: tap pair in an SGPR pair, sample broadcast from one half of a VGPR pair
            if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(ks), "v"(t2));
            if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(ks), "v"(t2));
            if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(p[i]) : "s"(ks), "v"(t2), "v"(s2));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i][0] + p[i][1];
    if (acc == 123.456f) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, unsigned long long* d, float* sink) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= 256;
    const int per_iter = (MODE == 0 || MODE == 3) ? 16 : 8;
    const int waves_per_simd = threads / 256;
    printf("%-34s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per SIMD\n", name, waves_per_simd, c / (kIters * per_iter), c / (kIters * per_iter) / waves_per_simd);
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 64);
    for (int threads : {256, 512, 1024}) {
        if (threads == 256) { run<0>("v_fma_f32", 256, d, sink); run<1>("v_pk_fma_f32", 256, d, sink); run<2>("v_pk_fma_f32 (broadcast src)", 256, d, sink); run<3>("v_max_f32", 256, d, sink); run<4>("v_pk_mul_f32", 256, d, sink);
                              run<5>("v_pk_fma_f32 (sgpr pair src0)", 256, d, sink); run<6>("v_pk_fma_f32 (sgpr + broadcast)", 256, d, sink);
                              run<7>("v_pk_fma_f32 (sgpr+bcast, independent)", 256, d, sink); }
        if (threads == 512) { run<0>("v_fma_f32", 512, d, sink); run<1>("v_pk_fma_f32", 512, d, sink); run<6>("v_pk_fma_f32 (sgpr + broadcast)", 512, d, sink); }
        if (threads == 1024) { run<0>("v_fma_f32", 1024, d, sink); run<1>("v_pk_fma_f32", 1024, d, sink); }
    }
    return 0;
}
