// Probe (round 4): which bits of the random operand does v_cvt_scalef32_sr_fp8_f16 use, and is its mean the input?
// For x = 1.03 (ratio / 8 = 0.12875, between the e4m3 codes 0.125 and 0.140625: P(up) should be 0.24) the random operand takes the
// values i << s for i = 0 .. 1023 and shifts s = 0, 4, 8, ... 28; prints the fraction rounded up per shift.
// hipcc --offload-arch=gfx950 -O3 -o sr_probe sr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float xin, int shift, unsigned *up) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, (_Float16)xin, i << shift, 8.f, 0);
    if ((w & 0xffu) != 0x20u) atomicAdd(up, 1u);
    if (i == 0 && shift == 0) printf("x = %f -> byte 0x%02x at rnd 0\n", xin, w & 0xffu);
}
int main() {
    unsigned *up; hipMalloc(&up, 4);
    for (float x : {1.03f, 1.0625f, 1.10f}) {
        for (int s = 0; s <= 28; s += 4) {
            hipMemset(up, 0, 4);
            hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, 0, x, s, up);
            unsigned h; hipMemcpy(&h, up, 4, hipMemcpyDeviceToHost);
            printf("x = %.4f  rnd = i << %2d, i < 1024: rounded up %4u / 1024 = %.3f   (expected %.3f)\n", x, s, h, h / 1024.0, (x / 8 - 0.125) / 0.015625);
        }
    }
    return 0;
}
