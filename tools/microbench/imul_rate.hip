// Microbenchmark: issue rate of the integer multiplies the Q6.12 kernel can choose from (round 3):
// v_mad_i64_i32 (exact 64-bit product-sum, what deployed_q612.hip uses), v_mul_i32_i24 / v_mad_i32_i24 (24-bit operands,
// 32-bit result), v_mul_lo_u32, with v_add_u32 as the plain-VALU yardstick.  W waves per SIMD, independent chains, wall
// time by the cycle counter.      hipcc -O3 --offload-arch=gfx950 tools/microbench/imul_rate.hip -o tools/microbench/imul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 4000;

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int* sink) {
    int a[8];
    long long w[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 37 + i * 1001; w[i] = a[i]; asm volatile("" : "+v"(a[i]), "+v"(w[i])); }
    int s = 77771, t = -12345;
    asm volatile("" : "+v"(s), "+v"(t));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 1) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
            if (MODE == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 4) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(s), "v"(t) : "vcc");
            if (MODE == 5) asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + (int)w[i];
    if (acc == 123456789) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, unsigned long long* d, int* sink) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= 256;
    const int waves_per_simd = threads / 256;
    printf("%-18s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per SIMD\n", name, waves_per_simd, c / (kIters * 8), c / (kIters * 8) / waves_per_simd);
}

int main() {
    unsigned long long* d; int* sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 64);
    for (int threads : {256, 1024}) {
        if (threads == 256) { run<0>("v_add_u32", 256, d, sink); run<1>("v_mul_i32_i24", 256, d, sink); run<2>("v_mad_i32_i24", 256, d, sink); run<3>("v_mul_lo_u32", 256, d, sink); run<4>("v_mad_i64_i32", 256, d, sink); run<5>("v_mul_hi_i32_i24", 256, d, sink); }
        else { run<0>("v_add_u32", 1024, d, sink); run<1>("v_mul_i32_i24", 1024, d, sink); run<2>("v_mad_i32_i24", 1024, d, sink); run<3>("v_mul_lo_u32", 1024, d, sink); run<4>("v_mad_i64_i32", 1024, d, sink); run<5>("v_mul_hi_i32_i24", 1024, d, sink); }
    }
    return 0;
}
