// Microbenchmark: can the VALU work of one wave overlap the MFMA work of the other wave on the
// same SIMD (gfx950)?  8 waves per workgroup = 2 per SIMD; waves 0-3 run an MFMA-only loop,
// waves 4-7 a VALU-only loop of one instruction kind; each role is timed alone and together.
// build: hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip ; run: ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { K_FMA = 0, K_TRANS = 1, K_CVT = 2, K_PKFMA = 3, K_MIX = 4, K_FMAMIX = 5, K_CVTSDWA = 6, K_PKADD = 7, K_ADD = 8, K_LOG = 9, K_RCP = 10, K_CVTF16 = 11, K_MOV = 12 };

template <int KIND>
__device__ __forceinline__ void valu_block(float (&x)[16], float c) {
    // 64 VALU instructions per call, 16 independent chains
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[e]) : "v"(c));
            if (KIND == K_TRANS) {
                if (rep & 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[e]));
                else asm volatile("v_log_f32 %0, %0" : "+v"(x[e]));
            }
            if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[e]) : "v"(c));
            if (KIND == K_PKFMA) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&x[e & ~1]) : "v"(*(double *)&x[e & ~1]));
            if (KIND == K_FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, %0, %0 op_sel_hi:[1,0,0]" : "+v"(x[e]) : "v"(c));
            if (KIND == K_CVTSDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x[e]));
            if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&x[e & ~1]) : "v"(*(double *)&x[e & ~1]));
            if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[e]) : "v"(c));
            if (KIND == K_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(x[e]));
            if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[e]));
            if (KIND == K_CVTF16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[e]));
            if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x[e]) : "v"(c));
            if (KIND == K_MIX) {      // the epilogue's mix: add, rcp, mul, log, fma per element
                if (rep == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[e]) : "v"(c));
                if (rep == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[e]));
                if (rep == 2) asm volatile("v_log_f32 %0, %0" : "+v"(x[e]));
                if (rep == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[e]) : "v"(c));
            }
        }
}

// mode bit0: MFMA waves active, bit1: VALU waves active; SAMEWAVE: every wave does both, interleaved
template <int KIND, int SAMEWAVE, int NACC = 4, int VARYOP = 0>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, int mode, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j); }
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    bf16x8 av[14], bv[14];
    for (int i = 0; i < 14; ++i) for (int j = 0; j < 8; ++j) { av[i][j] = (__bf16)(seed + i + j); bv[i][j] = (__bf16)(seed - i + j); }
    float x[16];
    for (int e = 0; e < 16; ++e) x[e] = seed + e + threadIdx.x;
    float r = 0.f;
    if (SAMEWAVE) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {     // 8 MFMAs (256 cycles of pipe) with 8 VALU each in between
                acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
                if (mode & 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[e + 8 * (j & 1)]) : "v"(seed));
                        if (KIND == K_TRANS) asm volatile("v_log_f32 %0, %0" : "+v"(x[e + 8 * (j & 1)]));
                    }
                }
            }
        }
    } else if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < (VARYOP ? 14 : 8); ++j)
                    acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VARYOP ? av[j] : a, VARYOP ? bv[j] : b, acc[j % NACC], 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) valu_block<KIND>(x, seed);
    }
    for (int m = 0; m < 8; ++m) for (int e = 0; e < 16; ++e) r += acc[m][e];
    for (int e = 0; e < 16; ++e) r += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int SAMEWAVE, int NACC = 4, int VARYOP = 0>
float run(int mode, int iters) {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, SAMEWAVE, NACC, VARYOP><<<256, 512>>>(out, 10, mode, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND, SAMEWAVE, NACC, VARYOP><<<256, 512>>>(out, iters, mode, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
template <int KIND>
void report(const char *name) {
    const int iters = 20000;
    const float m = run<KIND, 0>(1, iters), v = run<KIND, 0>(2, iters), b = run<KIND, 0>(3, iters);
    // per iteration: 8 MFMAs (wave group X) / 64 VALU instructions (wave group Y)
    printf("%-10s MFMA alone %.3f ms (%.1f ns/MFMA)  VALU alone %.3f ms (%.2f ns/instr)  both %.3f ms  -> overlap %.0f%% of the shorter\n",
           name, m, m * 1e6 / (iters * 8.0), v, v * 1e6 / (iters * 64.0), b, 100.0 * (m + v - b) / (m < v ? m : v));
}
int main() {
    report<K_FMA>("v_fma");
    report<K_TRANS>("rcp/log");
    report<K_CVT>("cvt_pk");
    report<K_PKFMA>("pk_mul");
    report<K_MIX>("mix");
    report<K_FMAMIX>("fma_mix");
    report<K_CVTSDWA>("cvt_sdwa");
    report<K_PKADD>("pk_add");
    report<K_ADD>("v_add");
    report<K_LOG>("v_log");
    report<K_RCP>("v_rcp");
    report<K_CVTF16>("cvt_f16");
    report<K_MOV>("v_mov");
    {
        const int it2 = 20000;
        const float m1 = run<K_MIX, 0, 1>(1, it2), v1 = run<K_MIX, 0, 1>(2, it2), b1 = run<K_MIX, 0, 1>(3, it2);
        const float m2 = run<K_MIX, 0, 2>(1, it2), b2 = run<K_MIX, 0, 2>(3, it2);
        printf("dependent MFMA chain (1 accumulator): MFMA alone %.3f  VALU(mix) alone %.3f  both %.3f\n", m1, v1, b1);
        const float m3 = run<K_MIX, 0, 7, 1>(1, it2), v3 = run<K_MIX, 0, 7, 1>(2, it2 * 14 / 8), b3 = run<K_MIX, 0, 7, 1>(3, it2);
        printf("14 distinct A/B operand sets, 7 accumulators (14 MFMAs per iteration): MFMA alone %.3f  both %.3f (VALU alone for the same iterations %.3f)\n", m3, b3, run<K_MIX, 0, 7, 1>(2, it2));
        (void)v3;
        printf("2 accumulators alternating:           MFMA alone %.3f  VALU(mix) alone %.3f  both %.3f\n", m2, v1, b2);
    }
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        const float m = kind ? run<K_TRANS, 1>(1, iters) : run<K_FMA, 1>(1, iters);
        const float b = kind ? run<K_TRANS, 1>(3, iters) : run<K_FMA, 1>(3, iters);
        printf("same wave, 8 %s between MFMAs (2 waves/SIMD): MFMA only %.3f ms, with VALU %.3f ms (64 VALU instr per 8 MFMA per wave)\n",
               kind ? "v_log" : "v_fma", m, b);
    }
    return 0;
}
