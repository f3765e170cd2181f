// What one hipFuncSetAttribute(MaxDynamicSharedMemorySize) per launch costs on the host (the library sets it per launch:
// the attribute is per device and one process may drive several).  Build: hipcc -O2 --offload-arch=gfx950 attr_cost.hip -o attr_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p) { extern __shared__ int s[]; if (p) p[0] = s[0]; }
int main() {
    const int N = 20000;
    hipStream_t st; hipStreamCreate(&st);
    for (int i = 0; i < 100; ++i) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, st, nullptr); }
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, st, nullptr); if ((i & 255) == 255) hipStreamSynchronize(st); }
    hipStreamSynchronize(st);
    auto t2 = std::chrono::steady_clock::now();
    int dev;
    for (int i = 0; i < N; ++i) hipGetDevice(&dev);
    auto t3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("hipFuncSetAttribute %.2f us/call, launch (async) %.2f us/call, hipGetDevice %.3f us/call\n", us(t0, t1) / N, us(t1, t2) / N, us(t2, t3) / N);
    return 0;
}
