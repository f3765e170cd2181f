// Probe (round 4): semantics of the three instructions the e4m3-feature path of the fp8 mode leans on.
//   v_cvt_scalef32_pk_bf16_fp8 d, s, scale [op_sel:[1,0,0]]   two e4m3 bytes of the low (high) half of s -> two bf16 (x scale?)
//   v_cvt_scalef32_pk_fp8_f32  d, a, b, scale [op_sel:[0,0,0,1]]   two f32 -> two e4m3 in the low (high) half of d, other half kept?
//   v_add_f32 d, a, b clamp                                       [0, 1] clamp of an f32 sum (ReLU for values below 1)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cvt_bf16_fp8_probe.hip -o tools/microbench/cvt_bf16_fp8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__global__ void k_dec(unsigned* out, float scale) {      // thread i: bytes (i, 255 - i) in both halves
    const unsigned i = threadIdx.x;
    const unsigned lo = i | ((255u - i) << 8), hi = ((i * 7u + 3u) & 255u) | (((i * 13u + 5u) & 255u) << 8);
    const unsigned s = lo | (hi << 16);
    unsigned a = 0xAAAAAAAAu, b = 0xAAAAAAAAu;
    asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(a) : "v"(s), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(b) : "v"(s), "v"(scale));
    out[2 * i] = a;
    out[2 * i + 1] = b;
}
__global__ void k_enc(const float* in, unsigned* out, float scale) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const unsigned i = threadIdx.x;
    const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    unsigned r = 0xAAAAAAAAu;
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(r) : "v"(a), "v"(b), "v"(scale));
    const unsigned after_lo = r;
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d), "v"(scale));
    float s;
    asm volatile("v_add_f32 %0, %1, %2 clamp" : "=v"(s) : "v"(a), "v"(b));
    out[3 * i] = after_lo;
    out[3 * i + 1] = r;
    out[3 * i + 2] = __float_as_uint(s);
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if ((v & 0x7F) == 0x7F) return NAN;
    const float mag = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -mag : mag;
}
static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 3 * 4);
    for (float scale : {1.0f, 4.0f, 0.25f}) {
        hipLaunchKernelGGL(k_dec, dim3(1), dim3(256), 0, 0, d, scale);
        unsigned h[512];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad_mul = 0, bad_div = 0, nan_ok = 0;
        for (unsigned i = 0; i < 256; ++i) {
            const unsigned char by[4] = {(unsigned char)i, (unsigned char)(255 - i), (unsigned char)((i * 7 + 3) & 255), (unsigned char)((i * 13 + 5) & 255)};
            const float got[4] = {bf(h[2 * i] & 0xFFFF), bf(h[2 * i] >> 16), bf(h[2 * i + 1] & 0xFFFF), bf(h[2 * i + 1] >> 16)};
            for (int j = 0; j < 4; ++j) {
                const float v = e4m3(by[j]);
                if (std::isnan(v)) { nan_ok += std::isnan(got[j]); continue; }
                bad_mul += !(got[j] == v * scale && std::signbit(got[j]) == std::signbit(v));
                bad_div += !(got[j] == v / scale && std::signbit(got[j]) == std::signbit(v));
            }
        }
        printf("fp8 -> bf16, scale %g: mismatches if result = value * scale: %d, if value / scale: %d (of 1016 finite; %d of 8 NaN bytes gave NaN)\n", scale, bad_mul, bad_div, nan_ok);
    }
    const float vals[] = {0.3f, -0.2f, 0.9f, 0.5f,   -0.3f, 0.1f, 3.f, 500.f,   0.f, -0.f, 1e-9f, -1e-9f,   0.4f, 0.7f, -2.f, 1000.f,   0.001953125f, 0.0009765625f, 0.00146484375f, 0.0029296875f};
    const int n = sizeof(vals) / 16;
    float* din; (void)hipMalloc(&din, sizeof(vals));
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (float scale : {1.0f, 0.5f}) {
        hipLaunchKernelGGL(k_enc, dim3(1), dim3(n), 0, 0, din, d, scale);
        unsigned h[64];
        (void)hipMemcpy(h, d, n * 12, hipMemcpyDeviceToHost);
        printf("---- f32 -> fp8, scale %g (MODE.FP16_OVFL = 1)\n", scale);
        for (int i = 0; i < n; ++i) {
            const unsigned lo = h[3 * i], r = h[3 * i + 1];
            float s; std::memcpy(&s, &h[3 * i + 2], 4);
            printf("in (%g, %g | %g, %g): after low cvt 0x%08x, after high cvt 0x%08x -> (%g, %g | %g, %g);  a + b clamp = %g (bits 0x%08x)\n",
                   vals[4 * i], vals[4 * i + 1], vals[4 * i + 2], vals[4 * i + 3], lo, r, e4m3(r & 255), e4m3((r >> 8) & 255), e4m3((r >> 16) & 255), e4m3(r >> 24), s, h[3 * i + 2]);
        }
    }
    return 0;
}
