#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
__global__ void k(const float *in, unsigned *out, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
    h2 r;
    r[0] = (_Float16)a; r[1] = (_Float16)b;
    out[threadIdx.x] = __builtin_bit_cast(unsigned, r);
}
__global__ void km(const h8 *a, const h8 *b, float *o) {
    f16v d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], d, 0, 0, 0);
    for (int e = 0; e < 16; ++e) o[threadIdx.x * 16 + e] = d[e];
}
int main() {
    float h[8] = {1.0f, 70000.f, 1e9f, 65504.f, 65520.f, 3.0e-8f, 1.5f, INFINITY};
    float *d; unsigned *o; unsigned ho[4];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, ov);
        hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        printf("ovfl=%d:", ov);
        for (int i = 0; i < 4; ++i) printf(" %04x %04x", ho[i] & 0xffff, ho[i] >> 16);
        printf("\n");
    }
    return 0;
}
