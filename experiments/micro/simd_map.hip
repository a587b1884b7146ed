// Which waves of an 8-wave (and 4-wave) workgroup share a SIMD?  Reads HW_REG_HW_ID per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o simd_map simd_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(unsigned *out, int lds_bytes_hint) {
    extern __shared__ unsigned char smem[];
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = hwid;
    if (lds_bytes_hint < 0) smem[threadIdx.x] = 1;
}
static void run(int waves, int lds) {
    const int blocks = 512;
    unsigned *d; hipMalloc(&d, blocks * waves * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), lds, 0, d, 0);
    unsigned h[512 * 8];
    hipMemcpy(h, d, blocks * waves * 4, hipMemcpyDeviceToHost);
    int hist[8][4]; memset(hist, 0, sizeof(hist));
    int same_pair[8][8]; memset(same_pair, 0, sizeof(same_pair));
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < waves; ++w) {
            const unsigned simd = (h[b * waves + w] >> 4) & 3;
            hist[w][simd]++;
            for (int v = 0; v < waves; ++v)
                if (((h[b * waves + v] >> 4) & 3) == simd) same_pair[w][v]++;
        }
    }
    printf("== %d waves per workgroup, %d B LDS: first blocks (simd ids by wave):", waves, lds);
    for (int b = 0; b < 4; ++b) { printf("  ["); for (int w = 0; w < waves; ++w) printf("%u", (h[b * waves + w] >> 4) & 3); printf("]"); }
    printf("\n   how often wave v shares the SIMD of wave 0 (of %d blocks):", blocks);
    for (int v = 0; v < waves; ++v) printf(" w%d:%d", v, same_pair[0][v]);
    printf("\n");
    hipFree(d);
}
int main() { run(8, 60000); run(8, 120000); run(4, 60000); return 0; }
