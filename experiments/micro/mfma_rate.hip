// Microbenchmark: MFMA issue patterns of the KL-NMF tile loops, no memory traffic.
//   pattern 0: 14 independent accumulators, round-robin
//   pattern 1: one 13-deep dependent chain (MFMA1) then 7 accumulators x 2 (MFMA2)  -- the row-pass pattern
//   pattern 2: as 1, plus ~130 VALU/TRANS instructions per tile between the two MFMA groups (epilogue stand-in)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int PATTERN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k(float *out, int iters, float seed) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j); }
    f32x16 acc[7], d;
    for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    for (int e = 0; e < 16; ++e) d[e] = 0.f;
    float s1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {
#pragma unroll
            for (int j = 0; j < 27; ++j) acc[j % 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 13; ++s) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
            bf16x8 q = b;
            if (PATTERN == 2) {
                float x[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float xx = seed + e;
                    const float qq = (xx + 1e-8f) * __builtin_amdgcn_rcpf(d[e] + 1e-8f);
                    s1 = fmaf(xx, __builtin_amdgcn_logf(qq), s1);
                    x[e] = qq;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = (__bf16)(x[j] + x[j + 8]);
            } else {
                asm volatile("" : "+v"(d));
            }
#pragma unroll
            for (int j = 0; j < 14; ++j) acc[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q, acc[j >> 1], 0, 0, 0);
        }
    }
    float r = s1;
    for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) r += acc[m][e];
    for (int e = 0; e < 16; ++e) r += d[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int P, int W>
void run(const char *name) {
    float *out; hipMalloc(&out, 256 * 64 * W * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<P, W><<<256, 64 * W>>>(out, 100, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<P, W><<<256, 64 * W>>>(out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 27 * (W / 4);
    const double tf = 256.0 * W * iters * 27 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-44s waves/SIMD %d: %.3f ms  %.1f ns per MFMA per SIMD  %.0f TFLOP/s\n", name, W / 4, ms,
           ms * 1e6 / mfma_per_simd, tf);
    hipFree(out);
}
int main() {
    run<0, 4>("independent accumulators");
    run<0, 8>("independent accumulators");
    run<1, 4>("13-chain + 7x2");
    run<1, 8>("13-chain + 7x2");
    run<2, 4>("13-chain + epilogue VALU + 7x2");
    run<2, 8>("13-chain + epilogue VALU + 7x2");
    return 0;
}
