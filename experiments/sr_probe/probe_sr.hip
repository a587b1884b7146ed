#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint16_t *xb, const unsigned *seeds, int ns, unsigned char *out) {
    const int i = blockIdx.x;       // which x
    const int j = threadIdx.x;      // which seed
    if (j >= ns) return;
    _Float16 x = __builtin_bit_cast(_Float16, xb[i]);
    int r = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(0, x, seeds[j], 1.0f, 0);
    out[i * ns + j] = (unsigned char)(r & 0xff);
}
__global__ void kopsel(unsigned char *out) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 m = {(_Float16)1.0f, (_Float16)2.0f};
    const unsigned u = __builtin_bit_cast(unsigned, m);
    unsigned w = 0;
    const float scale = 1.0f;
    asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3" : "+v"(w) : "v"(u), "v"(0u), "s"(scale));
    asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3 op_sel:[1,0,1,0]" : "+v"(w) : "v"(u), "v"(0u), "s"(scale));
    asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(w) : "v"(u), "v"(0u), "s"(scale));
    asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3 op_sel:[1,0,1,1]" : "+v"(w) : "v"(u), "v"(0u), "s"(scale));
    if (threadIdx.x == 0) { out[0] = w & 0xff; out[1] = (w >> 8) & 0xff; out[2] = (w >> 16) & 0xff; out[3] = (w >> 24) & 0xff; }
}
__global__ void ksat(const uint16_t *xb, int nx, unsigned char *out) {
    const int i = threadIdx.x;
    if (i >= nx) return;
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    _Float16 x = __builtin_bit_cast(_Float16, xb[i]);
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, f16x2{x, x}, 1.0f, false);
    out[3 * i] = (unsigned char)(w[0] & 0xff);
    out[3 * i + 1] = (unsigned char)(__builtin_amdgcn_cvt_scalef32_sr_fp8_f16(0, x, 0u, 1.0f, 0) & 0xff);
    out[3 * i + 2] = (unsigned char)(__builtin_amdgcn_cvt_scalef32_sr_fp8_f16(0, x, 0xffffffffu, 1.0f, 0) & 0xff);
}
int main() {
    // x = 1 + d/1024 (f16 bits 0x3C00 | d): e4m3 neighbours 1.0 (0x38) and 1.125 (0x39); discarded bits d & 127
    const int ds[] = {0, 1, 2, 32, 64, 96, 127, 128 + 64};
    const int nx = 8;
    uint16_t hx[nx];
    for (int i = 0; i < nx; ++i) hx[i] = 0x3C00 | ds[i];
    const int ns = 32 + 8;
    unsigned hs[ns];
    for (int b = 0; b < 32; ++b) hs[b] = 1u << b;
    hs[32] = 0; hs[33] = 0xffffffffu; hs[34] = 0x7f; hs[35] = 0x3f; hs[36] = 0xfe000000u; hs[37] = 0x01ffffffu; hs[38] = 0x0000ff00u; hs[39] = 0x00ff0000u;
    uint16_t *dx; unsigned *dsd; unsigned char *dout;
    hipMalloc(&dx, sizeof hx); hipMalloc(&dsd, sizeof hs); hipMalloc(&dout, nx * ns);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dsd, hs, sizeof hs, hipMemcpyHostToDevice);
    k<<<nx, 64>>>(dx, dsd, ns, dout);
    unsigned char ho[nx * ns];
    hipMemcpy(ho, dout, nx * ns, hipMemcpyDeviceToHost);
    for (int i = 0; i < nx; ++i) {
        printf("d=%3d:", ds[i]);
        for (int j = 0; j < ns; ++j) printf(" %02x", ho[i * ns + j]);
        printf("\n");
    }
    {
        const float vals[] = {440.f, 448.f, 460.f, 464.f, 480.f, 500.f, 1000.f, 60000.f, 0.001f, 0.0015f, 0.0f, 1e-5f};
        const int nv = 12;
        uint16_t hv[nv];
        for (int i = 0; i < nv; ++i) { _Float16 h = (_Float16)vals[i]; hv[i] = __builtin_bit_cast(uint16_t, h); }
        hv[7] = 0x7bff;   // 65504
        uint16_t *dv; unsigned char *do2; hipMalloc(&dv, sizeof hv); hipMalloc(&do2, 3 * nv);
        hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice);
        ksat<<<1, 64>>>(dv, nv, do2);
        unsigned char h2[3 * nv]; hipMemcpy(h2, do2, 3 * nv, hipMemcpyDeviceToHost);
        for (int i = 0; i < nv; ++i) printf("x=%g: nearest %02x  sr(seed 0) %02x  sr(seed ~0) %02x\n", vals[i], h2[3 * i], h2[3 * i + 1], h2[3 * i + 2]);
    }
    {
        unsigned char *d4; hipMalloc(&d4, 4); kopsel<<<1, 64>>>(d4);
        unsigned char h4[4]; hipMemcpy(h4, d4, 4, hipMemcpyDeviceToHost);
        printf("op_sel probe {lo = 1.0, hi = 2.0}: bytes %02x %02x %02x %02x (expected 38 40 38 40)\n", h4[0], h4[1], h4[2], h4[3]);
    }
    return 0;
}
