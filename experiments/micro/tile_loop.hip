// Microbenchmark of the row-pass tile loop structure, built up feature by feature
// (2 waves per SIMD, 8-wave workgroups, one workgroup per CU):
//   F_VALU  epilogue VALU/TRANS between the MFMA groups
//   F_LDS   operand fragments read from LDS (transposed + row reads, 3-slot ring, distance 2)
//   F_BAR   __syncthreads() every 2 tiles
//   F_DMA   global_load_lds of the next stage (30 KiB + V tiles) every 2 tiles
// build: hipcc --offload-arch=gfx950 -O3 -o tile_loop tile_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))
constexpr int RB = 136, KP = 224, KS = 13, KT = 7, STG = 32768, VAREA = 32768;

__device__ __forceinline__ bf16x8 tr_pair(const LDS unsigned char *p0, const LDS unsigned char *p1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4 *)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4 *)p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 b64_pair(const LDS unsigned char *p0, const LDS unsigned char *p1) {
    s16x4 lo = *(const LDS s16x4 *)p0, hi = *(const LDS s16x4 *)p1;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int F_VALU, int F_LDS, int F_BAR, int F_DMA, int ROWT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k(float *out, const unsigned char *gsrc, int stages, float seed) {
    __shared__ __attribute__((aligned(16))) unsigned char bufA[STG + VAREA];
    __shared__ __attribute__((aligned(16))) unsigned char bufB[STG + VAREA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    const int off_tr = (8 * h + tq) * RB + (16 * half + 4 * tp) * 2, off_row = r * RB + 8 * h;
    for (int i = tid; i < (STG + VAREA) / 4; i += 64 * WAVES) { ((LDS float *)bufA)[i] = 0.001f * (i & 255); ((LDS float *)bufB)[i] = 0.002f * (i & 127); }
    __syncthreads();
    bf16x8 wf[ROWT][KS];
    for (int t = 0; t < ROWT; ++t) for (int s = 0; s < KS; ++s) for (int j = 0; j < 8; ++j) wf[t][s][j] = (__bf16)(seed + 0.01f * (lane + s + j + t));
    f32x16 acc[ROWT][KT];
    for (int t = 0; t < ROWT; ++t) for (int m = 0; m < KT; ++m) for (int e = 0; e < 16; ++e) acc[t][m][e] = 0.f;
    float s1 = 0.f;
    const unsigned char *g = gsrc + (size_t)blockIdx.x * 65536;
    auto dma = [&](LDS unsigned char *buf) {
        if (F_DMA) {
            const int wb = (tid & ~63) * 16;
            for (int rr = 0; rr < 32 / (WAVES * 4 / 4) / 1; ++rr) if (rr * WAVES * 1024 < 32768)
                __builtin_amdgcn_global_load_lds((const GLB void *)(g + rr * WAVES * 1024 + tid * 16), (LDS void *)(buf + rr * WAVES * 1024 + wb), 16, 0, 0);
            for (int t = 0; t < 2 * ROWT; ++t) for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_global_load_lds((const GLB void *)(g + 32768 + (wave * 2 * ROWT + t) * 2048 % 32768 + lane * 32 + 16 * p),
                                                 (LDS void *)(buf + STG + (wave * 2 * ROWT + t) * 2048 % 32768 + 1024 * p), 16, 0, 0);
        }
    };
    auto compute = [&](const LDS unsigned char *img) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const LDS unsigned char *p1 = img + off_tr + 64 * u, *p2 = img + off_row + 64 * u;
            bf16x8 ring[3];
            auto fetch = [&](int idx) {
                if (!F_LDS) { if (idx < 27) ring[idx % 3] = wf[0][idx % KS]; return; }
                if (idx < KS) ring[idx % 3] = tr_pair(p1 + 16 * idx * RB, p1 + (16 * idx + 4) * RB);
                else if (idx < 27) { const int j = idx - KS, m = j >> 1, hh = j & 1; ring[idx % 3] = b64_pair(p2 + 32 * m * RB + 32 * hh, p2 + 32 * m * RB + 32 * hh + 16); }
            };
            fetch(0); fetch(1);
            float x[ROWT][16], q[ROWT][16];
            typedef __attribute__((ext_vector_type(4))) float f4;
#pragma unroll
            for (int t = 0; t < ROWT; ++t) {
                const f4 va = *(const LDS f4 *)(img + STG + ((wave * 2 * ROWT + 2 * t + u) * 2048) % 32768 + lane * 16);
                for (int e = 0; e < 16; ++e) x[t][e] = va.x + e + t;
            }
            f32x16 d[ROWT];
#pragma unroll
            for (int t = 0; t < ROWT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) d[t][e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                fetch(s + 2);
#pragma unroll
                for (int t = 0; t < ROWT; ++t) d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3], wf[t][s], d[t], 0, 0, 0);
            }
            bf16x8 b0[ROWT], b1[ROWT];
#pragma unroll
            for (int t = 0; t < ROWT; ++t) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (F_VALU == 1) { const float qq = (x[t][e] + 1e-8f) * __builtin_amdgcn_rcpf(d[t][e] + 1e-8f); q[t][e] = qq; s1 = fmaf(x[t][e], __builtin_amdgcn_logf(qq), s1); }
                    else if (F_VALU == 2) { const float qq = x[t][e] * __builtin_amdgcn_rcpf(d[t][e]); q[t][e] = qq; s1 = fmaf(x[t][e], __builtin_amdgcn_logf(qq), s1); }
                    else q[t][e] = d[t][e];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { b0[t][j] = (__bf16)q[t][j]; b1[t][j] = (__bf16)q[t][8 + j]; }
            }
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                fetch(KS + j + 2);
#pragma unroll
                for (int t = 0; t < ROWT; ++t) acc[t][j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(KS + j) % 3], (j & 1) ? b1[t] : b0[t], acc[t][j >> 1], 0, 0, 0);
            }
        }
    };
    LDS unsigned char *A = (LDS unsigned char *)bufA, *B = (LDS unsigned char *)bufB;
    for (int st = 0; st < stages; st += 2) {
        dma(B); compute(A); if (F_BAR) __syncthreads();
        dma(A); compute(B); if (F_BAR) __syncthreads();
    }
    float rsum = s1;
    for (int t = 0; t < ROWT; ++t) for (int m = 0; m < KT; ++m) for (int e = 0; e < 16; ++e) rsum += acc[t][m][e];
    out[blockIdx.x * 64 * WAVES + tid] = rsum;
}
template <int V, int L, int Bf, int D, int ROWT, int WAVES>
void run(const char *name, float *out, unsigned char *g) {
    const int stages = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, L, Bf, D, ROWT, WAVES><<<256, 64 * WAVES>>>(out, g, 20, 1.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<V, L, Bf, D, ROWT, WAVES><<<256, 64 * WAVES>>>(out, g, stages, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = stages * 2.0 * 27 * ROWT * (WAVES / 4);
    printf("%-46s rows/wave %d waves %d: %.3f ms  %.1f ns/MFMA  %.0f TFLOP/s\n", name, 32 * ROWT, WAVES, ms,
           ms * 1e6 / mfma_per_simd, 1024.0 * mfma_per_simd * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float *out; unsigned char *g; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&g, 256 * 65536 + 65536); hipMemset(g, 0x3c, 256 * 65536 + 65536);
    run<1, 1, 1, 1, 1, 8>("gen1: VALU+LDS+bar+DMA", out, g);
    run<2, 1, 1, 1, 1, 8>("gen1, lean VALU", out, g);
    run<0, 1, 1, 1, 1, 8>("gen1, no VALU", out, g);
    run<1, 1, 1, 1, 2, 4>("64 rows/wave, 1 wave/SIMD", out, g);
    run<2, 1, 1, 1, 2, 4>("64 rows/wave, 1 wave/SIMD, lean VALU", out, g);
    run<0, 1, 1, 1, 2, 4>("64 rows/wave, 1 wave/SIMD, no VALU", out, g);
    run<2, 0, 1, 0, 2, 4>("64 rows/wave, lean VALU, no LDS/DMA", out, g);
    run<1, 1, 1, 1, 1, 4>("32 rows/wave, 1 wave/SIMD", out, g);
    return 0;
}
