// What read bandwidth does a plain streaming kernel reach on this chip?  (Context for the T1 kernel's HBM fraction:
// MI355X_MICROARCH.md quotes 8 TB/s peak; this measures the practical ceiling of coalesced 16-B/lane reads of a 1 GiB buffer.)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ p, long n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int UNROLL>
static void run(const float4* p, long n4, float* out, int grid) {
    hipLaunchKernelGGL(rd<UNROLL>, dim3(grid), dim3(256), 0, 0, p, n4, out);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(rd<UNROLL>, dim3(grid), dim3(256), 0, 0, p, n4, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("unroll %d grid %5d: %.2f TB/s\n", UNROLL, grid, (double)n4 * 16 * 10 / (ms * 1e-3) / 1e12);
}

int main() {
    const long bytes = 1L << 30, n4 = bytes / 16;
    float4* p; float* out;
    hipMalloc(&p, bytes); hipMalloc(&out, 64);
    hipMemset(p, 0, bytes);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) { run<1>(p, n4, out, grid); run<4>(p, n4, out, grid); run<8>(p, n4, out, grid); }
    return 0;
}
