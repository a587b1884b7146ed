// Microbenchmark: the ping-pong row pass's M segment in isolation -- 14 ds_read_b128-fed MFMAs
// (7 accumulators x 2) + 13 ds_read_b64_tr_b16-pair-fed MFMAs (one dependent chain), reads issued by
// inline asm D fragments ahead with counted lgkmcnt waits.  No VALU, no barriers, no global traffic.
// Reports s_memtime cycles per MFMA for 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mseg mseg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS __attribute__((address_space(3)))

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
template <int OFF> __device__ __forceinline__ void rd128(bf16x8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF0, int OFF1> __device__ __forceinline__ void rdtr(bf16x8 &dst, unsigned addr) {
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF1));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    dst = __builtin_bit_cast(bf16x8, v);
}
template <int N> __device__ __forceinline__ void lwait(bf16x8 &v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N)); }

// KIND 0: 14 b128 + 13 tr (the real mix); 1: 27 b128; 2: 27 tr pairs; 3: register operands only
template <int KIND, int D, int RB, int FEAT = 0>
__global__ __launch_bounds__(512, 2) void k(float *out, unsigned long long *cyc, int iters, int nwaves, float seed, const unsigned char *gsrc = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned char dmabuf[FEAT & 2 ? 2 * 40960 : 16];
    __shared__ __attribute__((aligned(16))) unsigned char img[224 * RB];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 224 * RB / 4; i += 512) ((LDS float *)img)[i] = 0.001f * (i & 255);
    __syncthreads();
    if (wave >= nwaves) {
        if (FEAT & 32) {
            float x[16];
            for (int e = 0; e < 16; ++e) x[e] = seed + e + tid;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 7; ++rep)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (rep == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[e]) : "v"(seed));
                        if (rep == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[e]));
                        if (rep == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[e]) : "v"(seed));
                        if (rep == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[e]) : "v"(seed));
                        if (rep == 4) asm volatile("v_log_f32 %0, %0" : "+v"(x[e]));
                        if (rep == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[e]) : "v"(seed));
                        if (rep == 6 && (e & 1)) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[e]) : "v"(seed));
                    }
                if (FEAT & 1) asm volatile("s_barrier" ::: "memory");
            }
            float rr = 0; for (int e = 0; e < 16; ++e) rr += x[e];
            out[blockIdx.x * 512 + tid] = rr;
            return;
        }
        if (FEAT & 1) for (int it = 0; it < iters; ++it) asm volatile("s_barrier" ::: "memory");
        return;
    }
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    const unsigned base = (unsigned)(uintptr_t)(LDS unsigned char *)img;
    const unsigned ta = base + (8 * h + tq) * RB + (16 * half + 4 * tp) * 2;
    const unsigned ra = base + r * RB + 16 * h;
    constexpr int N2 = KIND == 2 ? 0 : (KIND == 1 ? 27 : 14), NF = 27, R = D + 1;
    bf16x8 wf[13], b0, b1, ring[R];
    for (int s = 0; s < 13; ++s) for (int j = 0; j < 8; ++j) wf[s][j] = (__bf16)(seed + 0.01f * (lane + s + j));
    for (int j = 0; j < 8; ++j) { b0[j] = (__bf16)(seed + j); b1[j] = (__bf16)(seed - j); }
    for (int i = 0; i < R; ++i) ring[i] = b0;
    f32x16 acc[7], d;
    for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    for (int e = 0; e < 16; ++e) d[e] = 0.f;
    auto issue = [&](auto P) {
        constexpr int p = decltype(P)::value;
        if constexpr (KIND == 3) return;
        if constexpr (p < N2) rd128<(32 * ((p >> 1) % 7)) * RB + 32 * (p & 1)>(ring[p % R], ra);
        else if constexpr (p < NF) rdtr<(16 * ((p - N2) % 13)) * RB, (16 * ((p - N2) % 13) + 4) * RB>(ring[p % R], ta);
    };
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        const unsigned char *gq[5];
        if (FEAT & 2) {
            const unsigned char *g = gsrc + ((size_t)blockIdx.x * 64 + ((FEAT & 4) ? (it & 1) : (it & 63))) * 40960 + tid * 16;
            const unsigned char *gl2 = gsrc + (size_t)(it & 127) * 24576 + tid * 16;      // image shared by every block (L2)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                gq[q] = ((FEAT & 8) && q < 3) ? gl2 + q * 8192 : g + q * 8192;
                if (!(FEAT & 16))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gq[q],
                                                     (LDS void *)(dmabuf + (it & 1) * 40960 + q * 8192 + (tid & ~63) * 16), 16, 0, 0);
            }
            if (!(FEAT & 16)) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
        static_for<0, D>([&](auto P) { issue(P); });
        static_for<0, NF>([&](auto P) {
            constexpr int p = decltype(P)::value;
            if constexpr ((FEAT & 16) != 0 && p % 5 == 0 && p < 25) {
                constexpr int q = p / 5;
                if (q == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // previous iteration's copies
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gq[q],
                                                 (LDS void *)(dmabuf + (it & 1) * 40960 + q * 8192 + (tid & ~63) * 16), 16, 0, 0);
            }
            issue(std::integral_constant<int, p + D>{});
            constexpr int last = (p + D < NF - 1) ? p + D : NF - 1;
            constexpr int nb = (last < N2 ? last : N2 - 1) - p > 0 ? (last < N2 ? last : N2 - 1) - p : 0;
            constexpr int nt = (last - p) - nb;
            if constexpr (KIND != 3) lwait<(nb + 2 * nt > 15 ? 15 : nb + 2 * nt)>(ring[p % R]);
            if constexpr (p < 14) acc[(p >> 1) % 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[p % R], (p & 1) ? b1 : b0, acc[(p >> 1) % 7], 0, 0, 0);
            else d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[p % R], wf[p - 14], d, 0, 0, 0);
        });
        if (FEAT & 1) asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float rr = 0.f;
    for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) rr += acc[m][e];
    for (int e = 0; e < 16; ++e) rr += d[e];
    out[blockIdx.x * 512 + tid] = rr;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int D, int RB, int FEAT = 0>
void run(const char *name, int nwaves) {
    unsigned char *gsrc = nullptr;
    if (FEAT & 2) { (void)hipMalloc(&gsrc, (size_t)256 * 64 * 40960 + 65536); (void)hipMemset(gsrc, 0, (size_t)256 * 64 * 40960 + 65536); }
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipMemset(cyc, 0, 256 * 8 * 8);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND, D, RB, FEAT><<<256, 512>>>(out, cyc, 10, nwaves, 1.5f, gsrc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND, D, RB, FEAT><<<256, 512>>>(out, cyc, iters, nwaves, 1.5f, gsrc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[256 * 8];
    (void)hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int i = 0; i < 256 * 8; ++i) if (hc[i]) { s += (double)hc[i]; ++n; }
    const double per_simd = (double)iters * 27 * (nwaves > 4 ? 2 : 1);
    printf("%-26s feat=%d D=%d rowB=%3d waves/CU %d: %.3f ms  %.1f ns/MFMA/SIMD  %.1f cycles/MFMA/wave (s_memtime)\n", name, FEAT, D, RB, nwaves, ms,
           ms * 1e6 / per_simd, s / n / (iters * 27.0));
    (void)hipFree(out); (void)hipFree(cyc); if (gsrc) (void)hipFree(gsrc);
}
int main() {
    run<3, 3, 80>("register operands", 4);
    run<3, 3, 80>("register operands", 8);
    run<0, 3, 80>("14 b128 + 13 tr", 4);
    run<0, 5, 80>("14 b128 + 13 tr", 4);
    run<0, 8, 80>("14 b128 + 13 tr", 4);
    run<0, 3, 80>("14 b128 + 13 tr", 8);
    run<0, 5, 80>("14 b128 + 13 tr", 8);
    run<1, 3, 80>("27 b128", 4);
    run<1, 6, 80>("27 b128", 4);
    run<2, 3, 80>("27 tr pairs", 4);
    run<2, 6, 80>("27 tr pairs", 4);
    run<0, 3, 80, 1>("mix + barrier/partner", 4);
    run<0, 3, 80, 2>("mix + DMA 40KB/iter", 4);
    run<0, 3, 80, 3>("mix + barrier + DMA", 4);
    run<3, 3, 80, 2>("regs + DMA 40KB/iter", 4);
    run<3, 3, 80, 1>("regs + barrier/partner", 4);
    run<3, 3, 80, 6>("regs + DMA from L2", 4);
    run<0, 3, 80, 6>("mix + DMA from L2", 4);
    run<3, 3, 80, 10>("regs + DMA 3 L2 + 2 HBM", 4);
    run<0, 3, 80, 10>("mix + DMA 3 L2 + 2 HBM", 4);
    run<0, 3, 80, 10>("mix + DMA 3 L2 + 2 HBM", 8);
    run<0, 3, 80, 11>("mix + DMA 3 L2 + 2 HBM + bar", 8);
    run<0, 3, 80, 26>("mix + DMA 3L2+2HBM spread", 4);
    run<0, 5, 80, 26>("mix + DMA 3L2+2HBM spread", 4);
    run<0, 5, 80, 10>("mix + DMA 3 L2 + 2 HBM", 4);
    run<0, 8, 80, 10>("mix + DMA 3 L2 + 2 HBM", 4);
    run<1, 6, 80, 10>("27 b128 + DMA 3L2+2HBM", 4);
    run<2, 6, 80, 10>("27 tr + DMA 3L2+2HBM", 4);
    run<3, 3, 80, 32>("regs MFMA + partner VALU epilogue", 4);
    run<0, 3, 80, 32>("LDS-fed MFMA + partner VALU epilogue", 4);
    run<0, 3, 80, 33>("LDS-fed MFMA + partner VALU + barrier", 4);
    run<0, 3, 80, 42>("LDS-fed + partner VALU + DMA(3 L2+2 HBM)", 4);
    run<0, 5, 144>("14 b128 + 13 tr", 4);
    run<1, 6, 144>("27 b128", 4);
    run<2, 6, 144>("27 tr pairs", 4);
    return 0;
}
