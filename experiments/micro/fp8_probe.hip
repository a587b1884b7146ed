#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
// (1) ds_read_b64_tr_b8: which bytes does lane l of a 16-lane group receive?  LDS image: byte at address a holds (a & 0xff)
//     with 32-byte rows: byte (row, col) = row * 32 + col.  Lane 2q + p of a group supplies the address of row q, cols 8p..8p+7.
__global__ void k_tr(unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned char img[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) img[i] = (unsigned char)(i & 0xff);
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15, q = j >> 1, p = j & 1;
    // group g: rows 8*(g>>1).., physical cols 16*(g&1)..
    const unsigned addr = (unsigned)(uintptr_t)img + (8 * (g >> 1) + q) * 32 + 16 * (g & 1) + 8 * p;
    u2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[2 * lane] = v[0];
    out[2 * lane + 1] = v[1];
}
// (2) fp8 conversions: saturation and the f16 route back
__global__ void k_cvt(const float *in, unsigned *out, float *back) {
    const int i = threadIdx.x;
    int packed = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
    out[i] = (unsigned)packed;
    h2 r = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((unsigned)packed, 1.0f, false);
    back[2 * i] = (float)r[0];
    back[2 * i + 1] = (float)r[1];
}
int main() {
    unsigned *d; hipMalloc(&d, 64 * 8); unsigned h[128];
    k_tr<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        if ((l & 15) < 3 || (l & 15) == 15) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) { unsigned byte = (h[2 * l + (b >> 2)] >> (8 * (b & 3))) & 0xff; printf(" (r%u,c%u)", byte / 32, byte % 32); }
            printf("\n");
        }
    }
    float hin[16] = {1.0f, 1.06f, 0.5f, 448.f, 449.f, 500.f, 1000.f, 1e8f, 0.0156f, 0.001f, 1e-5f, 0.f, 3.3f, 17.f, 240.f, 0.26f};
    float *din, *dback; unsigned *dout; hipMalloc(&din, 64); hipMalloc(&dback, 64); hipMalloc(&dout, 32);
    hipMemcpy(din, hin, 64, hipMemcpyHostToDevice);
    k_cvt<<<1, 8>>>(din, dout, dback);
    float hb[16]; unsigned ho[8]; hipMemcpy(hb, dback, 64, hipMemcpyDeviceToHost); hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%g -> fp8 0x%02x -> %g\n", hin[i], (ho[i / 2] >> (8 * (i & 1))) & 0xff, hb[i]);
    return 0;
}
