// Probe (round 4): does v_cvt_scalef32_pk_fp8_f16 use the whole f32 scale or only its exponent?  Converts 1.0 and 1.5 with the
// scales 8, 5.656854 (8 / sqrt 2) and 6: e4m3 bytes and what they decode to.
// hipcc --offload-arch=gfx950 -O3 -o scale_probe scale_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) short s2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void k(float scale, unsigned *o) {
    s2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, h2{(_Float16)1.0f, (_Float16)1.5f}, scale, false);
    o[0] = __builtin_bit_cast(unsigned, w);
}
static float dec(unsigned b) { int ex = (b >> 3) & 15, man = b & 7; return ex == 0 ? ldexpf((float)man, -9) : ldexpf(1.f + 0.125f * man, ex - 7); }
int main() {
    unsigned *o; hipMalloc(&o, 4);
    for (float s : {8.f, 5.656854f, 6.f, 4.f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, s, o);
        unsigned h; hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost);
        printf("scale %.6f: 1.0 -> byte 0x%02x = %.6f (1/scale = %.6f)   1.5 -> byte 0x%02x = %.6f (1.5/scale = %.6f)\n", s, h & 255, dec(h & 255), 1 / s, (h >> 8) & 255, dec((h >> 8) & 255), 1.5 / s);
    }
    return 0;
}
