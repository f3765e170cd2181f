// Which LDS access patterns of the asm-sequenced conv kernel conflict?  One kernel per pattern, run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d out -- tools/microbench/lds_patterns
// and read conflict cycles / active cycles per kernel (tools/microbench/README or DESIGN.md quote the result).
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ int entry_old(int f, int g) { return 16 * (f >> 2) + 4 * (f & 3) + ((g + (f >> 2)) & 3); }      // first attempt: 2-way on writes
__device__ __forceinline__ int entry(int f, int g) { return 8 * (4 * (f >> 3) + g) + ((f & 7) ^ (g >> 1)); }

template <int PAT>
__global__ __launch_bounds__(64) void k(float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nl = lane & 15, g = lane >> 4, fs = lane >> 2, gs = lane & 3;
    unsigned addr = base;
    if (PAT == 0) addr += entry(nl, g) * 16;            // partial write, MFMA layout through entry()
    if (PAT == 1) addr += lane * 16;                    // linear write
    if (PAT == 2) addr += entry(fs, gs) * 16;           // partial read b128, transposed reader
    if (PAT == 3) addr += entry(fs, 1) * 16 + gs * 4;   // tile-4 word read b32
    if (PAT == 4) addr += lane * 560;                   // image b128 read, stride 140 words
    if (PAT == 5) addr += lane * 560 + 8;               // image read2_b64
    if (PAT == 6) addr += entry_old(nl, g) * 16;        // the first entry formula, for comparison
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    u32x4 r = {0, 0, 0, 0};
    float r1 = 0.f;
    asm volatile("" : "+v"(v));
    for (int it = 0; it < 20000; ++it) {
        if (PAT == 0 || PAT == 1 || PAT == 6) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
        if (PAT == 2 || PAT == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
        if (PAT == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(addr) : "memory");
        if (PAT == 5) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(r) : "v"(addr) : "memory");
        if ((it & 15) == 15) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r), "+v"(r1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r), "+v"(r1));
    if (r[0] == 12345u || r1 == 1.5f) sink[0] = 1.f;
}

int main() {
    float* sink; hipMalloc(&sink, 64);
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(64), 0, 0, sink);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, sink);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(64), 0, 0, sink);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(64), 0, 0, sink);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(64), 0, 0, sink);
    hipLaunchKernelGGL(k<5>, dim3(256), dim3(64), 0, 0, sink);
    hipLaunchKernelGGL(k<6>, dim3(256), dim3(64), 0, 0, sink);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
