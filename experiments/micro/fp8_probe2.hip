#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_cvt(const float *in, unsigned *out) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = threadIdx.x;
    unsigned lo = 0;
    float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
    asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(lo) : "v"(c), "v"(d));
    out[i] = lo;
}
int main() {
    float hin[16] = {1.0f, 448.f, 449.f, 500.f, 1000.f, 1e8f, 0.0156f, 3.3f, 1e30f, 0.f, 17.f, 240.f, 465.f, 479.f, 480.f, 0.26f};
    float *din; unsigned *dout; hipMalloc(&din, 64); hipMalloc(&dout, 16);
    hipMemcpy(din, hin, 64, hipMemcpyHostToDevice);
    k_cvt<<<1, 4>>>(din, dout);
    unsigned ho[4]; hipMemcpy(ho, dout, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%g -> fp8 0x%02x\n", hin[i], (ho[i / 4] >> (8 * (i & 3))) & 0xff);
    return 0;
}
