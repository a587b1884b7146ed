// Microbenchmark (round 2): issue cost of the fp8 conversion instructions and of their possible replacements, alone and
// beside an MFMA-only partner wave on the same SIMD (same method as overlap.hip: 8 waves per workgroup, waves 0-3 run
// MFMAs, waves 4-7 one instruction kind; each role alone and both together).
// build: hipcc --offload-arch=gfx950 -O3 -o cvt_overlap cvt_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) short s2v;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { K_ENC_SC_F16 = 0, K_ENC_F32, K_ENC_SC_F32, K_DEC_SC_F16, K_DEC_PK_F32, K_DEC_F32, K_PERM, K_LSHR, K_ANDOR, K_PKMULF16, K_CVTPKF16, K_MUL, K_BFI, K_PKADDF16 };

template <int KIND>
__device__ __forceinline__ void valu_block(unsigned (&x)[16], float c) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            unsigned v = x[e];
            if (KIND == K_ENC_SC_F16) v = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(__builtin_bit_cast(s2v, v), __builtin_bit_cast(h2, x[(e + 1) & 15]), 8.f, false));
            if (KIND == K_ENC_F32) v = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_bit_cast(float, x[(e + 1) & 15]), c, (int)v, false);
            if (KIND == K_ENC_SC_F32) v = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(s2v, v), __builtin_bit_cast(float, x[(e + 1) & 15]), c, 8.f, false));
            if (KIND == K_DEC_SC_F16) v = (rep & 1) ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 8.f, true)) : __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 8.f, false));
            if (KIND == K_DEC_PK_F32) { f2 r = (rep & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)v, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)v, false); v = __builtin_bit_cast(unsigned, r[0]); x[(e + 1) & 15] ^= __builtin_bit_cast(unsigned, r[1]) & 1u; }
            if (KIND == K_DEC_F32) v = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_f32_fp8((int)v, 1));
            if (KIND == K_PERM) v = __builtin_amdgcn_perm(v, x[(e + 1) & 15], 0x0c010c00u);
            if (KIND == K_LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v));
            if (KIND == K_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v) : "v"(0x80008000u), "v"(x[(e + 1) & 15]));
            if (KIND == K_PKMULF16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v) : "v"(0x3c003c00u));
            if (KIND == K_PKADDF16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v) : "v"(0x3c003c00u));
            if (KIND == K_CVTPKF16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v) : "v"(c));
            if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(c));
            if (KIND == K_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v) : "v"(0x3f803f80u), "v"(x[(e + 1) & 15]));
            asm volatile("" : "+v"(v));
            x[e] = v;
        }
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, int mode, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed * 0.01f + threadIdx.x * 0.0001f + j * 0.01f); b[j] = (_Float16)(seed * 0.005f + j * 0.01f); }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    unsigned x[16];
    for (int e = 0; e < 16; ++e) x[e] = 0x38003c00u + e * 0x00010001u + threadIdx.x;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) valu_block<KIND>(x, seed);
    }
    for (int m = 0; m < 4; ++m) for (int e = 0; e < 16; ++e) r += acc[m][e];
    for (int e = 0; e < 16; ++e) r += (float)(x[e] & 0xff);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND> float run(int mode, int iters) {
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(out, 10, mode, 1.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND><<<256, 512>>>(out, iters, mode, 1.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}
template <int KIND> void report(const char *name) {
    const int iters = 10000;
    const float m = run<KIND>(1, iters), v = run<KIND>(2, iters), b = run<KIND>(3, iters);
    printf("%-28s MFMA alone %.3f ms (%.1f ns/MFMA)  VALU alone %.3f ms (%.2f ns/instr = %.1f cycles at 1.9 GHz)  both %.3f ms  -> overlap %.0f%% of the shorter\n",
           name, m, m * 1e6 / (iters * 8.0), v, v * 1e6 / (iters * 64.0), v * 1e6 / (iters * 64.0) * 1.9, b, 100.0 * (m + v - b) / (m < v ? m : v));
    fflush(stdout);
}
int main() {
    report<K_MUL>("v_mul_f32");
    report<K_CVTPKF16>("v_cvt_pk_f16_f32");
    report<K_ENC_SC_F16>("v_cvt_scalef32_pk_fp8_f16");
    report<K_ENC_SC_F32>("v_cvt_scalef32_pk_fp8_f32");
    report<K_ENC_F32>("v_cvt_pk_fp8_f32");
    report<K_DEC_SC_F16>("v_cvt_scalef32_pk_f16_fp8");
    report<K_DEC_PK_F32>("v_cvt_pk_f32_fp8");
    report<K_DEC_F32>("v_cvt_f32_fp8");
    report<K_PERM>("v_perm_b32");
    report<K_LSHR>("v_lshrrev_b32");
    report<K_ANDOR>("v_and_or_b32");
    report<K_BFI>("v_bfi_b32");
    report<K_PKMULF16>("v_pk_mul_f16");
    report<K_PKADDF16>("v_pk_add_f16");
    return 0;
}
