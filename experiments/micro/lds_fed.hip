// Microbenchmark: rate of a dependent MFMA chain whose A operand comes from LDS
// (ds_read_b64_tr_b16 pairs or ds_read_b128), as a function of the read-ahead distance D
// and of the waves per SIMD.  No VALU, no barriers, no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS __attribute__((address_space(3)))
constexpr int RB = 144;
__device__ __forceinline__ bf16x8 tr_pair(const LDS unsigned char *p0, const LDS unsigned char *p1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4 *)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4 *)p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// MODE 0: register operands; 1: tr pairs; 2: b128
template <int MODE, int D, int WAVES, int CHAINS>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k(float *out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) unsigned char img[224 * RB];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 224 * RB / 4; i += 64 * WAVES) ((LDS float *)img)[i] = 0.001f * (i & 255);
    __syncthreads();
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    const LDS unsigned char *p1 = (const LDS unsigned char *)img + (8 * h + tq) * RB + (16 * half + 4 * tp) * 2;
    const LDS unsigned char *p2 = (const LDS unsigned char *)img + r * RB + 16 * h;
    bf16x8 wf[13];
    for (int s = 0; s < 13; ++s) for (int j = 0; j < 8; ++j) wf[s][j] = (__bf16)(seed + 0.01f * (lane + s + j));
    f32x16 d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) d[c][e] = 0.f;
    constexpr int R = D + 1;
    for (int it = 0; it < iters; ++it) {
        bf16x8 ring[R];
        auto fetch = [&](int idx) {
            if (idx >= 13) return;
            if (MODE == 0) ring[idx % R] = wf[(idx + 1) % 13];
            else if (MODE == 1) ring[idx % R] = tr_pair(p1 + 16 * idx * RB, p1 + (16 * idx + 4) * RB);
            else ring[idx % R] = *(const LDS bf16x8 *)(p2 + (16 * idx) * RB);
        };
#pragma unroll
        for (int i = 0; i < D; ++i) fetch(i);
#pragma unroll
        for (int s = 0; s < 13; ++s) {
            fetch(s + D);
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % R], wf[s], d[c], 0, 0, 0);
        }
    }
    float rs = 0;
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) rs += d[c][e];
    out[blockIdx.x * 64 * WAVES + tid] = rs;
}
template <int MODE, int D, int WAVES, int CHAINS>
void run(const char *name, float *out) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, D, WAVES, CHAINS><<<256, 64 * WAVES>>>(out, 50, 1.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE, D, WAVES, CHAINS><<<256, 64 * WAVES>>>(out, iters, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 13 * CHAINS * (WAVES / 4);
    printf("%-22s D=%d waves/SIMD=%d MFMAs/fragment=%d : %6.1f ns per MFMA per SIMD\n", name, D, WAVES / 4, CHAINS, ms * 1e6 / mf);
}
int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 2, 4, 1>("registers", out);
    run<1, 1, 4, 1>("tr pairs", out); run<1, 2, 4, 1>("tr pairs", out); run<1, 4, 4, 1>("tr pairs", out); run<1, 8, 4, 1>("tr pairs", out);
    run<2, 2, 4, 1>("b128", out); run<2, 4, 4, 1>("b128", out); run<2, 8, 4, 1>("b128", out);
    run<1, 2, 8, 1>("tr pairs", out); run<1, 4, 8, 1>("tr pairs", out); run<2, 2, 8, 1>("b128", out); run<2, 4, 8, 1>("b128", out);
    run<1, 2, 4, 2>("tr pairs", out); run<2, 2, 4, 2>("b128", out); run<1, 2, 8, 2>("tr pairs", out);
    return 0;
}
