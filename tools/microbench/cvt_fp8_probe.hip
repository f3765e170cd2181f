#include <hip/hip_runtime.h>
__global__ void k(const float* a, unsigned* out) {
    float x = a[threadIdx.x], y = a[threadIdx.x + 64];
    unsigned d = 0;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    asm volatile("v_cvt_pk_fp8_f32 %0, |%1|, |%2|" : "=v"(d) : "v"(x), "v"(y));
    asm volatile("v_cvt_pk_fp8_f32 %0, |%1|, |%2| op_sel:[0,0,1]" : "+v"(d) : "v"(y), "v"(x));
    out[threadIdx.x] = d;
}
int main() {
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = (i % 2 ? -1.f : 1.f) * (i < 8 ? 1000.f : 0.37f * i);
    float* a; unsigned* o; hipMalloc(&a, 512); hipMalloc(&o, 256); hipMemcpy(a, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o);
    unsigned r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("x=%g y=%g -> %08x\n", h[i], h[i + 64], r[i]);
    return 0;
}
