// Probe (round 3): does gfx950 honour the VOP3 clamp bit on v_cvt_pk_bf16_f32?  If it clamps the converted values to
// [0, 1], a ReLU + bf16 pack of values known to stay below 1 (conv1's outputs under a power-of-two scale folded into the
// weights) is ONE instruction per pair instead of v_cvt_pk_bf16_f32 + v_pk_max_i16.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cvt_bf16_clamp_probe.hip -o tools/microbench/cvt_bf16_clamp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__global__ void k(const float* in, unsigned* out_clamp, unsigned* out_plain, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float a = in[2 * i], b = in[2 * i + 1];
    unsigned c, p;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(c) : "v"(a), "v"(b));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
    asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(p));
    out_clamp[i] = c;
    out_plain[i] = p;
}

static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main() {
    const float vals[] = {0.f, -0.f, 1e-30f, -1e-30f, 0.25f, -0.25f, 0.999f, 0.9999999f, 1.0f, 1.5f, 3e10f, -3e10f, 1e-40f, -1e-40f,
                          INFINITY, -INFINITY, NAN, -NAN, 0.00390625f * 1.0039f, 0.5f + 0.001953125f, 6.1e-13f, -6.1e-13f, 7.7e-5f, 0.99609375f + 0.001953125f};
    const int n = sizeof(vals) / sizeof(float) / 2;
    float* din; unsigned *dc, *dp;
    (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dc, n * 4); (void)hipMalloc(&dp, n * 4);
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dc, dp, n);
    unsigned hc[64], hp[64];
    (void)hipMemcpy(hc, dc, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hp, dp, n * 4, hipMemcpyDeviceToHost);
    int same = 0, total = 0;
    for (int i = 0; i < n; ++i)
        for (int h = 0; h < 2; ++h) {
            const unsigned short c = (unsigned short)(hc[i] >> (16 * h)), p = (unsigned short)(hp[i] >> (16 * h));
            printf("in % .9g : cvt+clamp 0x%04x (% .7g)   cvt + pk_max_i16 0x%04x (% .7g) %s\n", vals[2 * i + h], c, bf(c), p, bf(p), c == p ? "" : "  <-- differ");
            ++total; same += c == p;
        }
    printf("%d of %d identical\n", same, total);
    return 0;
}
