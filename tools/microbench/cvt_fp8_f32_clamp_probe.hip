// Probe (round 3): does the VOP3 clamp bit do anything on v_cvt_scalef32_pk_fp8_f32 (two f32 -> two e4m3 with a scale)?
// If it clamped to [0, 1] as it does on v_cvt_pk_bf16_f32, the fp8 conv kernel's ReLU + pack would be ONE VALU per pair
// (activations kept in [0, 1]) instead of two (v_cvt_pk_bf16_f32 clamp + v_cvt_scalef32_pk_fp8_bf16).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cvt_fp8_f32_clamp_probe.hip -o tools/microbench/cvt_fp8_f32_clamp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

template <bool OVFL>
__global__ void k(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    if (OVFL) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const float a = in[2 * i], b = in[2 * i + 1], one = 1.0f;
    unsigned plain = 0xAAAAAAAAu, cl = 0xAAAAAAAAu, ng = 0xAAAAAAAAu;
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(plain) : "v"(a), "v"(b), "v"(one));
    // the assembler has no clamp operand for this opcode: the VOP3 word with bit 15 set by hand, on fixed registers
    // (v_cvt_scalef32_pk_fp8_f32 v13, v10, v11, v12 = 0xd235000d 0x0432170a; clamp = 0x00008000 in the first dword)
    asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %0\n\ts_nop 1\n\t.long 0xd235800d, 0x0432170a\n\ts_nop 1\n\tv_mov_b32 %0, v13"
                 : "+v"(cl) : "v"(a), "v"(b), "v"(one) : "v10", "v11", "v12", "v13");
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, |%1|, |%2|, %3" : "+v"(ng) : "v"(a), "v"(b), "v"(one));
    out[i * 3] = plain; out[i * 3 + 1] = cl; out[i * 3 + 2] = ng;
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if ((v & 0x7F) == 0x7F) return NAN;
    const float mag = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -mag : mag;
}

int main() {
    const float vals[] = {0.f, -0.3f, -2.f, -500.f, 0.25f, 0.3f, 0.9f, 1.0f, 1.5f, 3.f, 100.f, 447.f, 460.f, 1000.f, -0.001f, 0.002f};
    const int n = sizeof(vals) / sizeof(float) / 2;
    float* din; unsigned* dout;
    (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dout, 64 * 3 * 4);
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        if (ov) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, din, dout, n);
        else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, din, dout, n);
        unsigned h[64 * 3];
        (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        printf("---- MODE.FP16_OVFL = %d\n", ov);
        for (int i = 0; i < n; ++i) {
            const unsigned p = h[i * 3], c = h[i * 3 + 1], g = h[i * 3 + 2];
            printf("in (% 9.4g, % 9.4g): plain 0x%08x -> (% .5g, % .5g)   clamp 0x%08x -> (% .5g, % .5g)   |x| 0x%08x -> (% .5g, % .5g)\n", vals[2 * i], vals[2 * i + 1],
                   p, e4m3(p & 0xFF), e4m3((p >> 8) & 0xFF), c, e4m3(c & 0xFF), e4m3((c >> 8) & 0xFF), g, e4m3(g & 0xFF), e4m3((g >> 8) & 0xFF));
        }
    }
    return 0;
}
