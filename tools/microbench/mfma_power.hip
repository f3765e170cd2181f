// Microbenchmark: which bf16 MFMA shape holds the higher SUSTAINED rate under the power limit?
// The conv kernel of the headline runs v_mfma_f32_16x16x32_bf16 at 85 % pipe-busy but the chip holds ~1.83 GHz under it
// (tools/clock_probe.sh); v_mfma_f32_32x32x16_bf16 does the same FLOPs per cycle with half the operand-register traffic
// per FLOP.  This loop runs each shape for about a second on every SIMD (1 or 2 waves per SIMD) with random bf16 operands
// rotating through four register sets, and reports TFLOP/s and the clock that rate implies (1024 FLOP/cycle/SIMD dense).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_power.hip -o tools/microbench/mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// two random bf16 in [-2, 2) with random mantissas: sign | exponent 126..127 | 7 random bits
__device__ inline unsigned rnd_bf16x2(unsigned s) {
    const unsigned r = mix(s);
    const unsigned lo = (r & 0x807fu) | (0x3f00u + ((r >> 8) & 0x80u));
    const unsigned hi = ((r >> 16) & 0x807fu) | (0x3f00u + ((r >> 24) & 0x80u));
    return lo | (hi << 16);
}

// SHAPE 0: 16x16x32 (8 accumulators of 4), SHAPE 1: 32x32x16 (4 accumulators of 16), SHAPE 2: 16x16x32 with the A operand in
// AGPRs (as 36 of the conv kernel's 60 weight fragments are), SHAPE 3: the block-scaled e4m3 MFMA 16x16x128 (fp8 mode; 2x the
// FLOPs per instruction, 32 cycles); ZERO: all-zero operands (power floor)
template <int SHAPE, bool ZERO>
__global__ __launch_bounds__(512) void power_kernel(float* sink, int iters) {
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = ZERO ? 0u : rnd_bf16x2(threadIdx.x * 131u + blockIdx.x * 7919u + i * 17u + j);
            b[i][j] = ZERO ? 0u : rnd_bf16x2(threadIdx.x * 257u + blockIdx.x * 104729u + i * 29u + j + 99u);
        }
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    float total = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 8]) : "v"(a[m % 4]), "v"(b[(m / 4) % 4]));
        }
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (SHAPE == 2) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
        for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(a[i]));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 8]) : "a"(a[m % 4]), "v"(b[(m / 4) % 4]));
        }
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (SHAPE == 3) {
        using u32x8 = __attribute__((ext_vector_type(8))) unsigned;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
        u32x8 a8[2], b8[2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 8; ++j) {      // random e4m3 bytes with exponents 5..8 (|x| in [0.25, 4)): no NaN (0x7F / 0xFF) patterns
                const unsigned ra = mix(threadIdx.x * 977u + blockIdx.x * 31u + i * 8u + j), rb = mix(ra + 12345u);
                a8[i][j] = ZERO ? 0u : ((ra & 0x87878787u) | 0x28282828u | ((ra >> 3) & 0x10101010u));
                b8[i][j] = ZERO ? 0u : ((rb & 0x87878787u) | 0x28282828u | ((rb >> 3) & 0x10101010u));
            }
        unsigned one = 0x7F7F7F7Fu;      // E8M0 scale 2^0
        asm volatile("" : "+v"(a8[0]), "+v"(a8[1]), "+v"(b8[0]), "+v"(b8[1]), "+v"(one));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m)      // 16 x (16*16*128*2) = 524,288 x 2 FLOP per iteration: counted below
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m % 8]) : "v"(a8[m % 2]), "v"(b8[(m / 2) % 2]), "v"(one));
        }
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) acc[i][j] = 0.f; asm volatile("" : "+a"(acc[i])); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m % 4]) : "v"(a[m % 4]), "v"(b[(m / 4) % 4]));
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
    }
    if (total == 123.456f) sink[0] = total;      // keep the loop
}

template <int SHAPE, bool ZERO>
static void run(const char* name, int waves_per_simd, float* sink) {
    const int threads = 256 * waves_per_simd;
    const int grid = 256;      // one workgroup per CU
    const int iters = 400000;      // 32 x 16x16x32 (or 16 x 32x32x16) MFMAs per iteration and wave: ~0.1 s per launch
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((power_kernel<SHAPE, ZERO>), dim3(grid), dim3(threads), 0, 0, sink, 2000);      // warm
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((power_kernel<SHAPE, ZERO>), dim3(grid), dim3(threads), 0, 0, sink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // FLOPs per wave per iteration: 32 MFMAs x 16*16*32*2 = 16 MFMAs x 32*32*16*2 = 524,288
        const double flops = 4.0 * iters * 524288.0 * (SHAPE == 3 ? 2.0 : 1.0) * (double)grid * 4.0 * waves_per_simd;
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf("%-34s %d wave/SIMD  %8.1f ms  %7.1f TFLOP/s  implied clock %.3f GHz (of %d FLOP/cycle/SIMD)\n", name, waves_per_simd, ms, tf,
               tf * 1e12 / ((SHAPE == 3 ? 2048.0 : 1024.0) * 1024.0) / 1e9, SHAPE == 3 ? 2048 : 1024);
        fflush(stdout);
    }
}

int main() {
    float* sink;
    CK(hipMalloc(&sink, 4096));
    for (int w = 1; w <= 2; ++w) {
        run<0, true>("16x16x32 bf16, zero operands", w, sink);
        run<1, true>("32x32x16 bf16, zero operands", w, sink);
        run<0, false>("16x16x32 bf16, random", w, sink);
        run<1, false>("32x32x16 bf16, random", w, sink);
        run<2, false>("16x16x32 bf16, A in AGPRs, random", w, sink);
        run<3, true>("16x16x128 e4m3 scaled, zero", w, sink);
        run<3, false>("16x16x128 e4m3 scaled, random", w, sink);
    }
    return 0;
}
