// Issue rate of the fp8 conversions on gfx950: packed round-to-nearest (v_cvt_scalef32_pk_fp8_f16), stochastic single value
// (v_cvt_scalef32_sr_fp8_f16), and the seed arithmetic around it.  One wave per SIMD-slot, N dependent-free conversions per trip.
//   hipcc --offload-arch=gfx950 -O2 -o cvt_rate.bin cvt_rate.hip && ./cvt_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int trips, float scale) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned x[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = 0x3c003a00u + 0x00010001u * (threadIdx.x + i); w[i] = 0; }
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {        // one group = 4 values -> one dword
            const int j = (i + 1) & 7;
            if (MODE == 0) {             // 2 packed conversions
                asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(w[i]) : "v"(x[i]), "s"(scale));
                asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w[i]) : "v"(x[j]), "s"(scale));
            } else if (MODE == 1 || MODE == 2) {
                unsigned r1 = 0x12345678u, r1b = 0x9abcdef0u, r2 = 0x0fedcba9u, r2b = 0x87654321u;
                if (MODE == 2) {         // the product's seed arithmetic: 2 LCG steps, 2 shifts
                    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(s) : "s"(0x6C8E95u), "v"(0x3C6EF35Fu)); r1 = s;
                    asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(r1b) : "v"(r1));
                    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(s) : "s"(0x6C8E95u), "v"(0x3C6EF35Fu)); r2 = s;
                    asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(r2b) : "v"(r2));
                }
                asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3" : "+v"(w[i]) : "v"(x[i]), "v"(r1), "s"(scale));
                asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3 op_sel:[1,0,1,0]" : "+v"(w[i]) : "v"(x[i]), "v"(r1b), "s"(scale));
                asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(w[i]) : "v"(x[j]), "v"(r2), "s"(scale));
                asm volatile("v_cvt_scalef32_sr_fp8_f16 %0, %1, %2, %3 op_sel:[1,0,1,1]" : "+v"(w[i]) : "v"(x[j]), "v"(r2b), "s"(scale));
            } else if (MODE == 4) {      // 4 plain full-rate vector instructions (the yardstick)
                asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(w[i]) : "v"(x[i]));
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s;
}

template <int MODE> float run(unsigned *d, int trips) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<1024, 256>>>(d, 16, 1.0f);
    hipEventRecord(a);
    k<MODE><<<1024, 256>>>(d, trips, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    unsigned *d; hipMalloc(&d, 1024 * 256 * 4);
    const int trips = 4096;
    const float t3 = run<3>(d, trips), t4 = run<4>(d, trips), t0 = run<0>(d, trips), t1 = run<1>(d, trips), t2 = run<2>(d, trips);
    // waves per SIMD: 1024 blocks x 4 waves / (256 CUs x 4 SIMDs) = 4 waves per SIMD in turn; groups of 4 values per wave: trips x 8
    const double groups = (double)trips * 8 * 4;      // per SIMD
    printf("loop alone %.3f ms | 4 x v_xor %.3f | packed nearest (2 instr) %.3f ms | stochastic (4 instr) %.3f ms | stochastic + seeds (8 instr) %.3f ms\n", t3, t4, t0, t1, t2);
    printf("per group of 4 values and wave (ns): packed %.2f  stochastic %.2f  stochastic + seeds %.2f  (loop alone %.2f)\n",
           1e6 * t0 / groups, 1e6 * t1 / groups, 1e6 * t2 / groups, 1e6 * t3 / groups);
    return 0;
}
