// Microbenchmark (round 2): does the epilogue's VALU mix of one wave overlap the LDS-fed MFMA segment of the other wave
// on the same SIMD -- as the ping-pong row pass (mfma4.hip.h) assumes -- and does it depend on where the accumulators
// live (VGPR vs AGPR form of the MFMA) or on where the operands come from (registers vs LDS reads with counted waits)?
//   waves 0-3 ("M"): ITER x [27 MFMA 32x32x16 f16], operands from registers or one 1 KiB LDS fragment per MFMA
//   waves 4-7 ("E"): ITER x [16 x (v_rcp, v_fma_mix, v_log, v_fma_mix) + 8 v_cvt_pk_f16_f32]   (one tile's epilogue)
// each role alone and both together; no barriers.  build: hipcc --offload-arch=gfx950 -O3 -o pingpong_overlap pingpong_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int AG> __device__ __forceinline__ void mfma(f16v &acc, const h8 &a, const h8 &b) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// mode bit0: M waves run, bit1: E waves run.  AG: accumulators in AGPRs.  LDSF: operands through LDS.
template <int AG, int LDSF>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, int mode, float seed) {
    __shared__ __attribute__((aligned(16))) unsigned char img[32768];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += 512) ((float *)img)[i] = 0.001f * (i & 255);
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f16v acc[7];
            for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
            h8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + lane * 0.001f + j); b[j] = (_Float16)(seed * 0.5f + j); }
            const unsigned base = (unsigned)(uintptr_t)img + lane * 16;
            for (int it = 0; it < iters; ++it) {
                if (LDSF) {
                    h8 ring[4];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(ring[0]) : "v"(base));
                    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(ring[1]) : "v"(base));
                    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(ring[2]) : "v"(base));
#pragma unroll
                    for (int p = 0; p < 27; ++p) {
                        if (p + 3 < 27) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(p + 3) & 3]) : "v"(base), "n"(((p + 3) * 1024) & 32767));
                        if (p + 3 < 27) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ring[p & 3]));
                        else if (p + 2 < 27) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ring[p & 3]));
                        else if (p + 1 < 27) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ring[p & 3]));
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ring[p & 3]));
                        mfma<AG>(acc[p % 7], ring[p & 3], b);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 27; ++p) mfma<AG>(acc[p % 7], a, b);
                }
            }
            for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) r += acc[m][e];
        }
    } else {
        if (mode & 2) {
            float d[16], s1 = 0.f;
            h8 xa, xb;
            for (int e = 0; e < 16; ++e) d[e] = seed + 1.f + e + lane;
            for (int j = 0; j < 8; ++j) { xa[j] = (_Float16)(1.f + j); xb[j] = (_Float16)(2.f + j); }
            unsigned pk[8];
            for (int it = 0; it < iters; ++it) {
                float q[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float x = (float)(e < 8 ? xa[e & 7] : xb[e & 7]);
                    float rinv;
                    asm volatile("v_rcp_f32 %0, %1" : "=v"(rinv) : "v"(d[e]));
                    q[e] = __builtin_fmaf(x, rinv, 1e-8f * rinv);
                    float lg;
                    asm volatile("v_log_f32 %0, %1" : "=v"(lg) : "v"(q[e]));
                    s1 = __builtin_fmaf(x, lg, s1);
                    d[e] += 1e-3f * q[e];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(q[2 * j]), "v"(q[2 * j + 1]));
                for (int j = 0; j < 8; ++j) s1 += __builtin_bit_cast(float, pk[j]) * 1e-30f;
            }
            r = s1;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int AG, int LDSF>
float run(int mode, int iters) {
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<AG, LDSF><<<256, 512>>>(out, 10, mode, 1.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<AG, LDSF><<<256, 512>>>(out, iters, mode, 1.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}
template <int AG, int LDSF>
void report(const char *name) {
    const int iters = 4000;
    const float m = run<AG, LDSF>(1, iters), v = run<AG, LDSF>(2, iters), b = run<AG, LDSF>(3, iters);
    printf("%-44s M alone %.3f ms (%.1f ns / 27 MFMA)  E alone %.3f ms (%.1f ns / tile)  both %.3f ms  -> %.0f%% of the shorter hidden\n",
           name, m, m * 1e6 / iters, v, v * 1e6 / iters, b, 100.0 * (m + v - b) / (m < v ? m : v));
}
int main() {
    report<0, 0>("VGPR accumulators, register operands");
    report<1, 0>("AGPR accumulators, register operands");
    report<0, 1>("VGPR accumulators, LDS-fed operands");
    report<1, 1>("AGPR accumulators, LDS-fed operands");
    return 0;
}
