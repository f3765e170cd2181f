// Probe (round 3): v_cvt_scalef32_pk_fp8_bf16 -- packed bf16 pair -> two e4m3 with an f32 scale operand.  Questions:
// does it multiply or divide by the scale, what does it do above 448 (with and without MODE.FP16_OVFL), which half does
// op_sel pick.  Purpose: the fp8 conv kernel's ReLU + pack as v_cvt_pk_bf16_f32 (clamp) + this = 2 VALU per pair
// instead of 2 v_med3_f32 + v_cvt_pk_fp8_f32 = 3.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cvt_fp8_bf16_probe.hip -o tools/microbench/cvt_fp8_bf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

template <bool OVFL>
__global__ void k(const float* in, const float* scales, unsigned* out, int n, int ns) {
    const int i = threadIdx.x;
    if (i >= n) return;
    if (OVFL) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const float a = in[2 * i], b = in[2 * i + 1];
    unsigned pk;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(pk) : "v"(a), "v"(b));
    for (int s = 0; s < ns; ++s) {
        const float sc = scales[s];
        unsigned d = 0xAAAAAAAAu;
        asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(d) : "v"(pk), "v"(sc));
        unsigned e = 0xAAAAAAAAu;
        asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(e) : "v"(pk), "v"(sc));
        out[(s * 64 + i) * 2] = d;
        out[(s * 64 + i) * 2 + 1] = e;
    }
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if ((v & 0x7F) == 0x7F) return NAN;
    const float mag = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -mag : mag;
}

int main() {
    const float vals[] = {0.f, -0.3f, 0.001953125f, 0.0009765625f, 0.25f, 0.3f, 0.4375f, 0.5f, 0.8f, 0.875f, 0.9f, 1.0f, 1.7f, 3e-5f, 0.0146484375f, 0.01513671875f};
    const float scales[] = {1.0f, 0.001953125f /* 2^-9 */, 512.f};
    const int n = sizeof(vals) / sizeof(float) / 2, ns = 3;
    float *din, *dsc; unsigned* dout;
    (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dsc, sizeof(scales)); (void)hipMalloc(&dout, ns * 64 * 2 * 4);
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice); (void)hipMemcpy(dsc, scales, sizeof(scales), hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        if (ov) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, din, dsc, dout, n, ns);
        else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, din, dsc, dout, n, ns);
        unsigned h[3 * 64 * 2];
        (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        printf("---- MODE.FP16_OVFL = %d\n", ov);
        for (int s = 0; s < ns; ++s)
            for (int i = 0; i < n; ++i) {
                const unsigned d = h[(s * 64 + i) * 2], e = h[(s * 64 + i) * 2 + 1];
                printf("scale %-11g in (% .6g, % .6g): plain 0x%08x -> (% .5g, % .5g)   op_sel[2] 0x%08x -> hi half (% .5g, % .5g)\n", scales[s], vals[2 * i], vals[2 * i + 1],
                       d, e4m3(d & 0xFF), e4m3((d >> 8) & 0xFF), e, e4m3((e >> 16) & 0xFF), e4m3((e >> 24) & 0xFF));
            }
    }
    return 0;
}
