#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) short s2;
__global__ void k(const float *in, unsigned *out, float *back) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    h2 a = {(_Float16)in[2 * threadIdx.x], (_Float16)in[2 * threadIdx.x + 1]};
    s2 old = {0, 0};
    s2 lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, a, 8.0f, false);
    unsigned u = __builtin_bit_cast(unsigned, lo);
    out[threadIdx.x] = u;
    h2 r = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(u, 8.0f, false);
    back[2 * threadIdx.x] = (float)r[0]; back[2 * threadIdx.x + 1] = (float)r[1];
}
int main() {
    float hin[8] = {1.0f, 448.f, 3000.f, 4000.f, 60000.f, 0.13f, 0.05f, 0.01f};
    float *din, *db; unsigned *dout; hipMalloc(&din, 32); hipMalloc(&db, 32); hipMalloc(&dout, 16);
    hipMemcpy(din, hin, 32, hipMemcpyHostToDevice);
    k<<<1, 4>>>(din, dout, db);
    float hb[8]; unsigned ho[4]; hipMemcpy(hb, db, 32, hipMemcpyDeviceToHost); hipMemcpy(ho, dout, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%g -> fp8(scale 8) 0x%02x -> %g\n", hin[i], (ho[i / 2] >> (8 * (i & 1))) & 0xff, hb[i]);
    return 0;
}
