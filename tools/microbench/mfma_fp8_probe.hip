// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, scales 2^0) on gfx950:
//  (1) operand map: is D[row][col] = sum over (kg, j) of A(lane = row + 16 kg, byte j) * B(lane = col + 16 kg, byte j)?
//      (i.e. row/col on lane&15, and the k index a function of (lane>>4, byte) shared by A and B) -- exact integer data;
//  (2) issue interval, one wave per SIMD, AGPR accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void probe(const int* a, const int* b, float* d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[lane * 8 + i]; bv[i] = b[lane * 8 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) d[lane * 4 + r] = c[r];
}

constexpr int kIters = 2000;
__global__ __launch_bounds__(256, 1) void rate(unsigned long long* out, float* sink) {
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = 0x38383838; bv[i] = threadIdx.x; }
    asm volatile("" : "+v"(av), "+v"(bv));
    f32x4 acc[5];
    for (int i = 0; i < 5; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
    int sc = 0x7F7F7F7F;
    asm volatile("" : "+v"(sc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it)
#pragma unroll
        for (int m = 0; m < 20; ++m)
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[m % 5]) : "v"(av), "v"(bv), "v"(sc));
    const unsigned long long t1 = __builtin_readcyclecounter();
    float keep = 0.f;
    for (int i = 0; i < 5; ++i) keep += acc[i][0];
    if (keep == 123.456f) sink[0] = keep;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

static unsigned char fp8_of(int v) {      // exact e4m3 codes of -2..2 and +-0.5
    switch (v) { case 0: return 0x00; case 1: return 0x38; case 2: return 0x40; case -1: return 0xB8; case -2: return 0xC0; case 3: return 0x30; default: return 0xB0; }
}
static float val_of(int v) { return v == 3 ? 0.5f : v == 4 ? -0.5f : (float)v; }

int main() {
    std::vector<int> a(64 * 8), b(64 * 8);
    std::vector<int> ca(64 * 32), cb(64 * 32);
    srand(7);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            int va = rand() % 7 - 2, vb = rand() % 7 - 2;      // -2..4
            ca[l * 32 + j] = va; cb[l * 32 + j] = vb;
            reinterpret_cast<unsigned char*>(a.data())[l * 32 + j] = fp8_of(va);
            reinterpret_cast<unsigned char*>(b.data())[l * 32 + j] = fp8_of(vb);
        }
    int *da, *db; float* dd;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 64 * 16);
    hipMemcpy(da, a.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 64 * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> d(64 * 4);
    hipMemcpy(d.data(), dd, 64 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int col = lane & 15, row = (lane >> 4) * 4 + r;      // C/D layout of the 16x16 shapes
            float want = 0.f;
            for (int kg = 0; kg < 4; ++kg)
                for (int j = 0; j < 32; ++j) want += val_of(ca[(row + 16 * kg) * 32 + j]) * val_of(cb[(col + 16 * kg) * 32 + j]);
            if (want != d[lane * 4 + r]) { if (bad < 5) printf("mismatch lane %d r %d: got %g want %g\n", lane, r, d[lane * 4 + r], want); ++bad; }
        }
    printf("operand-map hypothesis: %s (%d mismatches of 256)\n", bad ? "WRONG" : "holds", bad);
    unsigned long long* dout; float* sink;
    hipMalloc(&dout, 256 * 8); hipMalloc(&sink, 64);
    hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, dout, sink);
    hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, dout, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= 256;
    printf("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8): %.2f cycles per instruction, one wave per SIMD\n", c / (kIters * 20.0));
    return 0;
}
