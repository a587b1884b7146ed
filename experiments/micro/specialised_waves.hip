// Microbenchmark (round 4): the row pass's interval with SPECIALISED waves, next to the product schedule (pingpong_steps.hip,
// whose harness this is).  Product: every wave runs M (27 MFMAs) then E (the epilogue) for its own 32 rows, the two waves of a
// SIMD half a tile apart.  SPEC (feature bit 262144): the two waves of a SIMD share ONE block of 32 rows --
//   E-wave (waves 0-3):  MFMA-1 of tile t (13 MFMAs, W fragments in registers) -> epilogue of tile t -> packed ratios to LDS
//   M-wave (waves 4-7):  packed ratios of tile t-1 from LDS -> MFMA-2 of tile t-1 (14 MFMAs) into the G accumulators
// one s_barrier per tile; a workgroup then covers 128 rows per tile step instead of 256, so the figure to compare is TWO steps
// against one interval of the product schedule (both = 2 tiles of 32 rows per SIMD).  SPEC2 (524288): the E-wave issues MFMA-1 of
// tile t+1 between the quarters of the epilogue of tile t (two W.H buffers) instead of in front of it.
// (original header follows)
// Microbenchmark (round 2, after pingpong_overlap.hip): the row pass's interval rebuilt from nothing, one ingredient at a
// time, to see which of them costs the time the kernel loses against "two MFMA segments per SIMD and interval".
//   workgroup = 8 waves (two per SIMD: X = waves 0-3, Y = waves 4-7), one workgroup per CU, each wave owns 32 rows
//   per tile and wave:  M = 14 MFMA-2 (B = ratio operands) + 13 MFMA-1 (B = W registers), every A fragment 1 KiB from LDS
//                       E = 16 x (rcp, mul, fma_mix, log, fma_mix) + 8 cvt_pk_f16 (+ 8 fp8 conversions with QST)
//   order as in k_rowpass4:  X: M E | barrier      Y: M | barrier | E      (one s_barrier per tile and wave)
// feature bits (template parameter F):
//   1  BAR   the barrier (without it the waves free-run; Y starts with an E to be out of phase)
//   2  DMA   every wave copies its slices of the next dictionary tile image (13 KiB per tile and workgroup, L2-resident
//            source) into a 4-object LDS ring with global_load_lds in its E segment; the M segments read that ring
//   4  VLD   V tiles streamed from HBM (2 x 16 B per lane and tile, issued one E segment ahead, vmcnt(0) at E's start)
//   8  QST   fp8 ratio tile stored (16 B per lane and tile, non-temporal) one tile late, as the kernel does
//   16 DEP   real data flow: E consumes the MFMA-1 accumulator, MFMA-2 consumes E's packed ratios (else constants)
//   32 PRIO  s_setprio 1 for the Y waves
//   64 TR    MFMA-1 fragments read with two ds_read_b64_tr_b16 instead of one ds_read_b128
//   128 PLAIN  ordinary instead of non-temporal ratio stores      256 QSMALL  ratio tiles overwrite 4 slots per wave (cache-resident)
//   1024 NOCVT the stored 16 bytes are the first half of the packed fp16 ratios (no fp8 conversions)    2048 NOSTORE conversions only
//   4096 BATCH2 VMEM batched: V loads and dictionary copies for two tiles issued every second E segment, vmcnt(0) only there
//   8192 VSMALL V tiles re-read from 4 slots per wave (cache hits)
//   16384 SPREAD the E segment's memory instructions issued between quarters of its arithmetic instead of ahead of it
//   131072 BAR2 the barrier only behind every second tile (timing only: the copies' ring would need twice the objects)
//   65536 MCV  only the fp8 conversions of the previous tile's ratios in M (E stores the result)
//   32768 MVM  memory instructions + fp8 conversions in the M segment (between its MFMAs), vmcnt(0) at its start; E arithmetic only
//   512 NOWAIT no s_waitcnt vmcnt(0) at the start of E (only meaningful without VLD / DMA: isolates issue cost from the wait)
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pingpong_steps pingpong_steps.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) short s2v;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
#define LDSP __attribute__((address_space(3)))

constexpr int kImg = 13312;          // 32 columns x 208 components x 2 B
constexpr int kObj = 13312;
constexpr int kTilesL2 = 128;        // dictionary: 128 tile images = 1.7 MB
#ifndef MN2
#define MN2 14
#endif
constexpr int N2 = MN2, N1 = 13, NF = N1 + N2, D = 3;

template <int OFF> __device__ __forceinline__ void rd128(h8 &r, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void rdtr(h8 &r, unsigned a) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    h4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 512));
    r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int N> __device__ __forceinline__ void lwait(h8 &r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N));
}
template <int I, int E, class Fn> __device__ __forceinline__ void sfor(Fn &&f) {
    if constexpr (I < E) { f(std::integral_constant<int, I>{}); sfor<I + 1, E>(f); }
}

template <int F>
__global__ __launch_bounds__(512, 1) void k(float *out, const unsigned char *ht, const unsigned char *vt, unsigned char *qt,
                                            int iters, float seed) {
    constexpr bool BAR = F & 1, DMA = F & 2, VLD = F & 4, QST = F & 8, DEP = F & 16, PRIO = F & 32, TR = F & 64;
    constexpr bool SPEC = (F & 262144) != 0, SPEC2 = (F & 524288) != 0;
    constexpr bool PLAIN = F & 128, QSMALL = F & 256, NOWAIT = F & 512, NOCVT = F & 1024, NOSTORE = F & 2048, BATCH2 = F & 4096, VSMALL = F & 8192, SPREAD = F & 16384, MVM = F & 32768, MCV = F & 65536, BAR2 = F & 131072;
    __shared__ __attribute__((aligned(16))) unsigned char img[4 * kObj + 32768];      // ring of 4 objects (+ pad: one WG per CU)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpY = wave >= 4;
    for (int i = tid; i < (4 * kObj) / 2; i += 512) ((LDSP _Float16 *)img)[i] = (_Float16)(0.0078125f * (1 + ((i * 37) & 31)));
    __syncthreads();
    if (PRIO && grpY) __builtin_amdgcn_s_setprio(1);
    const int64_t wslot = ((int64_t)blockIdx.x * 8 + wave) * iters;
    const unsigned char *vlane = vt + wslot * 2048 + lane * 32;
    unsigned char *qlane = qt + wslot * 1024 + lane * 16;
    f16v acc[7], d;
    h8 wf[N1], ring[4], b0, b1, va[4], vb[4];
    for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    for (int e = 0; e < 16; ++e) d[e] = seed + 4.f + e;
    for (int s = 0; s < N1; ++s) for (int j = 0; j < 8; ++j) wf[s][j] = (_Float16)(0.01f * (1 + ((lane + s + j) & 15)));
    for (int j = 0; j < 8; ++j) { b0[j] = (_Float16)(1.f + 0.01f * j); b1[j] = (_Float16)(0.9f + 0.01f * j); va[0][j] = va[1][j] = va[2][j] = va[3][j] = (_Float16)(1.f + j); vb[0][j] = vb[1][j] = vb[2][j] = vb[3][j] = (_Float16)(2.f + j); }
    float s1 = 0.f;
    const float eps = 1e-8f * seed;
    const unsigned lbase = (unsigned)(uintptr_t)img + lane * 16;
    const unsigned tbase = (unsigned)(uintptr_t)img + lane * 8;       // transposed reads: 8 B per lane, contiguous (no bank conflicts)
    if (VLD) { va[0] = *(const h8 *)vlane; vb[0] = *(const h8 *)(vlane + 16); }
    if (VLD && BATCH2) { va[1] = *(const h8 *)(vlane + 2048); vb[1] = *(const h8 *)(vlane + 2048 + 16); }

    auto dma = [&](int o, int tg) {
        if (!DMA) return;
        const unsigned char *g = ht + (int64_t)(tg & (kTilesL2 - 1)) * kImg;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int off = r * 8192 + tid * 16;
            if (off < kImg)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(g + off),
                                                 (LDSP unsigned *)(img + o * kObj + r * 8192 + wave * 1024), 16, 0, 0);
        }
    };
    u4 qpk = {0u, 0u, 0u, 0u};
    // MVM: the wave's memory instructions and the fp8 conversions of the previous tile's ratios live in the M segment
    // (vmcnt(0) at its start covers what the previous M segment issued, a whole interval ago); E is arithmetic only
    auto m_cvt_store = [&](int it) {
        if (!QST) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const h8 &src = j < 2 ? b0 : b1;
            const int o = 4 * (j & 1);
            s2v w = {0, 0};
            w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, h2{src[o], src[o + 1]}, 8.f, false);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, h2{src[o + 2], src[o + 3]}, 8.f, true);
            qpk[j] = __builtin_bit_cast(unsigned, w);
        }
        if (it > 0) __builtin_nontemporal_store(qpk, (u4 *)(qlane + (int64_t)(it - 1) * 1024));
    };
    auto m_cvt = [&]() {          // MCV: only the conversions move into M (the store of the result stays at the start of E)
        if (!QST) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const h8 &src = j < 2 ? b0 : b1;
            const int o = 4 * (j & 1);
            s2v w = {0, 0};
            w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, h2{src[o], src[o + 1]}, 8.f, false);
            w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, h2{src[o + 2], src[o + 3]}, 8.f, true);
            qpk[j] = __builtin_bit_cast(unsigned, w);
        }
    };
    auto m_vload = [&](int slot, int it) {
        if (!VLD) return;
        const unsigned char *p = vlane + (int64_t)min(it + 1, iters - 1) * 2048;
        va[slot] = *(const h8 *)p;
        vb[slot] = *(const h8 *)(p + 16);
    };
    auto seg_M = [&](auto TS, int it) {
        constexpr int ts = decltype(TS)::value;
        if (MVM) asm volatile("s_waitcnt vmcnt(0)" : "+v"(va[ts & 1]), "+v"(vb[ts & 1])::"memory");
        const unsigned ra = lbase + (DMA ? ((ts + 3) % 4) * kObj : 0);
        const unsigned ta = (TR ? tbase : lbase) + (DMA ? (ts % 4) * kObj : 0);
        auto issue = [&](auto P) {
            constexpr int p = decltype(P)::value;
            if constexpr (p < N2) rd128<(p % 13) * 1024>(ring[p % 4], ra);
            else if constexpr (p < NF) {
                if constexpr (TR) rdtr<((p - N2) % 13) * 1024>(ring[p % 4], ta);
                else rd128<((p - N2) % 13) * 1024>(ring[p % 4], ta);
            }
        };
        sfor<0, D>([&](auto P) { issue(P); });
        sfor<0, NF>([&](auto P) {
            constexpr int p = decltype(P)::value;
            issue(std::integral_constant<int, p + D>{});
            constexpr int last = (p + D < NF - 1) ? p + D : NF - 1;
            constexpr int n_a = (last < N2 ? last : N2 - 1) - p > 0 ? (last < N2 ? last : N2 - 1) - p : 0;
            constexpr int n_b = (last - p) - n_a;
            lwait<n_a + (TR ? 2 : 1) * n_b>(ring[p % 4]);
            if constexpr (p < N2) acc[(p >> 1) % 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[p % 4], (p & 1) ? b1 : b0, acc[(p >> 1) % 7], 0, 0, 0);
            else {
                if constexpr (p == N2 && DEP) for (int e = 0; e < 16; ++e) d[e] = 0.f;
                if constexpr (DEP) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[p % 4], wf[p - N2], d, 0, 0, 0);
                else acc[(p - N2) % 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[p % 4], wf[p - N2], acc[(p - N2) % 7], 0, 0, 0);
            }
            if constexpr (MCV && p == 3) {
                __builtin_amdgcn_sched_barrier(0);
                m_cvt();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MVM && (p == 3 || p == 9 || p == 16)) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (p == 3) m_cvt_store(it);
                if constexpr (p == 9) m_vload((ts + 1) & 1, it);
                if constexpr (p == 16) dma((ts + 2) % 4, it + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if (BAR && grpY && (!BAR2 || (ts & 1))) { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    };
    auto seg_E = [&](auto TS, int it) {
        constexpr int ts = decltype(TS)::value;
        h8 &xa = va[BATCH2 ? ts : (ts & 1)], &xb = vb[BATCH2 ? ts : (ts & 1)];
        if (BATCH2) {
            if ((ts & 1) == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(va[ts]), "+v"(vb[ts]), "+v"(va[ts + 1]), "+v"(vb[ts + 1])::"memory");
        } else if (!NOWAIT && !MVM) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa), "+v"(xb)::"memory");
        auto vm_store = [&]() {
        if (QST && NOSTORE) asm volatile("" ::"v"(qpk));
            else if (QST && it > 0) {
                u4 *qp = (u4 *)(qlane + (int64_t)(QSMALL ? ((it - 1) & 3) : (it - 1)) * 1024);
                if (PLAIN) *qp = qpk;
                else __builtin_nontemporal_store(qpk, qp);
            }
        };
        auto vm_vload = [&]() {
        if (VLD && BATCH2) {
                if ((ts & 1) == 0) {
#pragma unroll
                    for (int u = 2; u < 4; ++u) {
                        const int tn = VSMALL ? ((it + u) & 3) : min(it + u, iters - 1);
                        const unsigned char *p = vlane + (int64_t)tn * 2048;
                        va[(ts + u) & 3] = *(const h8 *)p;
                        vb[(ts + u) & 3] = *(const h8 *)(p + 16);
                    }
                }
            } else if (VLD) {
                const int tn = VSMALL ? ((it + 1) & 3) : min(it + 1, iters - 1);
                const unsigned char *p = vlane + (int64_t)tn * 2048;
                va[(ts + 1) & 1] = *(const h8 *)p;
                vb[(ts + 1) & 1] = *(const h8 *)(p + 16);
            }
        };
        auto vm_dma = [&]() {
        if (BATCH2) {
                if ((ts & 1) == 0) { dma((ts + 2) % 4, it + 2); dma((ts + 3) % 4, it + 3); }
            } else dma(grpY ? (ts + 3) % 4 : (ts + 2) % 4, it + (grpY ? 3 : 2));
        };
        if (!SPREAD && !MVM) { vm_store(); vm_vload(); vm_dma(); }
        float q[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = (float)(e < 8 ? xa[e & 7] : xb[e & 7]);
            const float rinv = __builtin_amdgcn_rcpf(d[e]);
            q[e] = __builtin_fmaf(x, rinv, eps * rinv);
            s1 = __builtin_fmaf(x, __builtin_amdgcn_logf(q[e]), s1);
            if (SPREAD && (e & 3) == 3) {        // the memory instructions spread over the arithmetic instead of ahead of it
                asm volatile("" : "+v"(s1));
                __builtin_amdgcn_sched_barrier(0);
                if (e == 3) vm_store();
                if (e == 7) vm_vload();
                if (e == 11) vm_dma();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // packed as the kernel's compiler output has it: one v_cvt_pk_f16_f32 per pair, the fp8 conversions from those registers
        u4 p0, p1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0[j]) : "v"(q[2 * j]), "v"(q[2 * j + 1]));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1[j]) : "v"(q[8 + 2 * j]), "v"(q[8 + 2 * j + 1]));
        }
        const h8 n0 = __builtin_bit_cast(h8, p0), n1 = __builtin_bit_cast(h8, p1);
        if (DEP) { b0 = n0; b1 = n1; }
        else { asm volatile("" ::"v"(n0), "v"(n1)); }
        if (QST && NOCVT) qpk = p0;
        else if (QST && !MVM && !MCV) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u4 &src = j < 2 ? p0 : p1;
                const int o = 2 * (j & 1);
                s2v w = {0, 0};
                w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_bit_cast(h2, src[o]), 8.f, false);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_bit_cast(h2, src[o + 1]), 8.f, true);
                qpk[j] = __builtin_bit_cast(unsigned, w);
            }
        }
        asm volatile("" : "+v"(s1));
        __builtin_amdgcn_sched_barrier(0);
        if (BAR && !grpY && (!BAR2 || (ts & 1))) { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    };
    if constexpr (SPEC) {
        // ---- specialised waves: E-wave = waves 0-3 (MFMA-1 + epilogue), M-wave = waves 4-7 (MFMA-2), pair (w, w + 4) on one SIMD
        const bool isM = grpY;
        const int pair = wave & 3;
        LDSP unsigned char *qx = (LDSP unsigned char *)img + 4 * kObj;        // [2 slots][4 pairs][2 operands][64 lanes][16 B] = 16 KiB
        const unsigned qaddr = (unsigned)(uintptr_t)qx + pair * 2048 + lane * 16;
        f16v d2;
        for (int e = 0; e < 16; ++e) d2[e] = seed + 5.f + e;
        auto mfma1 = [&](auto TS, f16v &dd) {          // W.H of the tile in ring object ts % 4
            constexpr int ts = decltype(TS)::value;
            const unsigned ta = (TR ? tbase : lbase) + (DMA ? (ts % 4) * kObj : 0);
            auto issue = [&](auto P) {
                constexpr int p = decltype(P)::value;
                if constexpr (p < N1) {
                    if constexpr (TR) rdtr<(p % 13) * 1024>(ring[p % 4], ta);
                    else rd128<(p % 13) * 1024>(ring[p % 4], ta);
                }
            };
            sfor<0, D>([&](auto P) { issue(P); });
            sfor<0, N1>([&](auto P) {
                constexpr int p = decltype(P)::value;
                issue(std::integral_constant<int, p + D>{});
                constexpr int last = (p + D < N1 - 1) ? p + D : N1 - 1;
                lwait<(TR ? 2 : 1) * (last - p)>(ring[p % 4]);
                if constexpr (p == 0) for (int e = 0; e < 16; ++e) dd[e] = 0.f;
                dd = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[p % 4], wf[p], dd, 0, 0, 0);
            });
        };
        auto mfma2 = [&](auto TS) {                    // G += H_tile . Q_tile^T of the tile in ring object (ts + 3) % 4
            constexpr int ts = decltype(TS)::value;
            const unsigned ra = lbase + (DMA ? ((ts + 3) % 4) * kObj : 0);
            auto issue = [&](auto P) {
                constexpr int p = decltype(P)::value;
                if constexpr (p < N2) rd128<(p % 13) * 1024>(ring[p % 4], ra);
            };
            sfor<0, D>([&](auto P) { issue(P); });
            sfor<0, N2>([&](auto P) {
                constexpr int p = decltype(P)::value;
                issue(std::integral_constant<int, p + D>{});
                constexpr int last = (p + D < N2 - 1) ? p + D : N2 - 1;
                lwait<last - p>(ring[p % 4]);
                acc[(p >> 1) % 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[p % 4], (p & 1) ? b1 : b0, acc[(p >> 1) % 7], 0, 0, 0);
            });
        };
        // the epilogue of tile `it` on dd; with AHEAD, the 13 MFMAs of the NEXT tile's W.H (into dn) are issued between its quarters
        auto epilogue = [&](auto TS, auto AHEAD, int it, const f16v &dd, f16v &dn) {
            constexpr int ts = decltype(TS)::value;
            constexpr bool ahead = decltype(AHEAD)::value;
            constexpr int tn = (ts + 1) % 4;
            const unsigned ta = (TR ? tbase : lbase) + (DMA ? tn * kObj : 0);
            auto issue = [&](auto P) {
                constexpr int p = decltype(P)::value;
                if constexpr (p < N1) {
                    if constexpr (TR) rdtr<(p % 13) * 1024>(ring[p % 4], ta);
                    else rd128<(p % 13) * 1024>(ring[p % 4], ta);
                }
            };
            auto step = [&](auto P) {
                constexpr int p = decltype(P)::value;
                if constexpr (ahead && p < N1) {
                    issue(std::integral_constant<int, p + D>{});
                    constexpr int last = (p + D < N1 - 1) ? p + D : N1 - 1;
                    lwait<(TR ? 2 : 1) * (last - p)>(ring[p % 4]);
                    if constexpr (p == 0) for (int e = 0; e < 16; ++e) dn[e] = 0.f;
                    dn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[p % 4], wf[p], dn, 0, 0, 0);
                }
            };
            h8 &xa = va[ts & 1], &xb = vb[ts & 1];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa), "+v"(xb)::"memory");
            if (QST && it > 0) __builtin_nontemporal_store(qpk, (u4 *)(qlane + (int64_t)(it - 1) * 1024));
            if (VLD) {
                const unsigned char *p = vlane + (int64_t)min(it + 1, iters - 1) * 2048;
                va[(ts + 1) & 1] = *(const h8 *)p;
                vb[(ts + 1) & 1] = *(const h8 *)(p + 16);
            }
            dma((ts + 2) % 4, it + 2);
            if constexpr (ahead) sfor<0, D>([&](auto P) { issue(P); });
            float q[16];
            sfor<0, 4>([&](auto Q) {
                constexpr int qi = decltype(Q)::value;
                // MFMAs 0..12 spread 4/3/3/3 over the quarters, each in front of four elements of VALU work
                sfor<0, (qi == 0 ? 4 : 3)>([&](auto J) { step(std::integral_constant<int, (qi == 0 ? 0 : 1 + 3 * qi) + decltype(J)::value>{}); });
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 4 * qi; e < 4 * qi + 4; ++e) {
                    const float x = (float)(e < 8 ? xa[e & 7] : xb[e & 7]);
                    const float rinv = __builtin_amdgcn_rcpf(dd[e]);
                    q[e] = __builtin_fmaf(x, rinv, eps * rinv);
                    s1 = __builtin_fmaf(x, __builtin_amdgcn_logf(q[e]), s1);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            u4 p0, p1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0[j]) : "v"(q[2 * j]), "v"(q[2 * j + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1[j]) : "v"(q[8 + 2 * j]), "v"(q[8 + 2 * j + 1]));
            }
            if (QST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u4 &src = j < 2 ? p0 : p1;
                    const int o = 2 * (j & 1);
                    s2v w = {0, 0};
                    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_bit_cast(h2, src[o]), 8.f, false);
                    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_bit_cast(h2, src[o + 1]), 8.f, true);
                    qpk[j] = __builtin_bit_cast(unsigned, w);
                }
            }
            // the packed ratios to the pair's LDS slot (it & 1): the M-wave multiplies them in the next interval
            const unsigned dst = qaddr + (unsigned)(it & 1) * 8192;
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" ::"v"(dst), "v"(p0), "v"(p1) : "memory");
            asm volatile("" : "+v"(s1));
        };
        auto tile_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // two loops, one per role, so that neither role carries the other's registers (G accumulators / W fragments)
        if (!isM) {
            for (int t4 = 0; t4 < iters; t4 += 4)
                sfor<0, 4>([&](auto I) {
                    constexpr int ts = decltype(I)::value;
                    const int it = t4 + ts;
                    if constexpr (SPEC2) {
                        if (ts & 1) epilogue(I, std::true_type{}, it, d2, d);
                        else epilogue(I, std::true_type{}, it, d, d2);
                    } else {
                        mfma1(I, d);
                        epilogue(I, std::false_type{}, it, d, d2);
                    }
                    tile_barrier();
                });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            out[(int64_t)blockIdx.x * 512 + tid] = s1 + d[3] + d2[5];
        } else {
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
            for (int t4 = 0; t4 < iters; t4 += 4)
                sfor<0, 4>([&](auto I) {
                    constexpr int ts = decltype(I)::value;
                    const int it = t4 + ts;
                    if (it > 0) {
                        const unsigned src = qaddr + (unsigned)((it - 1) & 1) * 8192;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1) : "v"(src) : "memory");
                    }
                    dma((ts + 3) % 4, it + 3);
                    mfma2(I);
                    tile_barrier();
                });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float r = 0.f;
            for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) r += acc[m][e];
            out[(int64_t)blockIdx.x * 512 + tid] = r;
        }
        return;
    }
    if (!BAR && grpY) seg_E(std::integral_constant<int, 3>{}, 0);      // free-running: start the Y waves out of phase
    for (int t4 = 0; t4 < iters; t4 += 4)
        sfor<0, 4>([&](auto I) {
            seg_M(I, t4 + decltype(I)::value);
            seg_E(I, t4 + decltype(I)::value);
        });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = s1 + d[3];
    for (int m = 0; m < 7; ++m) for (int e = 0; e < 16; ++e) r += acc[m][e];
    out[(int64_t)blockIdx.x * 512 + tid] = r;
}

__global__ void fill_h(_Float16 *p, int64_t n, float lo, float step, int mask) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = (_Float16)(lo + step * (float)((i * 29) & mask));
}

static float *g_out; static unsigned char *g_ht, *g_vt, *g_qt;
constexpr int kGrid = 2048, kIters = 128;

template <int F> void report(const char *name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<F><<<256, 512>>>(g_out, g_ht, g_vt, g_qt, 8, 1.5f);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<F><<<kGrid, 512>>>(g_out, g_ht, g_vt, g_qt, kIters, 1.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double us = best * 1e3 / ((kGrid / 256.0) * kIters);
    printf("F=%3d %-58s %7.3f ms  %6.3f us / interval (2 tiles per SIMD; 54 MFMAs = 0.91 us at 1.9 GHz) %s\n", F, name, best, us,
           err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main() {
    const int64_t nv = (int64_t)kGrid * 8 * kIters * 1024, nq = (int64_t)kGrid * 8 * kIters * 1024;   // halves / bytes
    (void)hipMalloc(&g_out, (size_t)kGrid * 512 * 4);
    (void)hipMalloc(&g_ht, (size_t)kTilesL2 * kImg);
    (void)hipMalloc(&g_vt, (size_t)nv * 2);
    (void)hipMalloc(&g_qt, (size_t)nq);
    fill_h<<<1024, 256>>>((_Float16 *)g_ht, (int64_t)kTilesL2 * kImg / 2, 0.0078125f, 0.0078125f, 31);
    fill_h<<<4096, 256>>>((_Float16 *)g_vt, nv, 0.5f, 0.0625f, 63);
    (void)hipDeviceSynchronize();
    report<17 + 2 + 4 + 8 + 64>("product schedule: all + transposed reads (= the kernel's interval)");
    report<17 + 2 + 4 + 8 + 64 + 32>("product schedule + priority for Y");
    report<262144 + 16 + 64>("SPEC: MFMA + epilogue + LDS hand-off only (x 2 for one product interval)");
    report<262144 + 16 + 64 + 2>("SPEC + dictionary copies (x 2)");
    report<262144 + 16 + 64 + 2 + 4 + 8>("SPEC + copies + V from HBM + ratio store (x 2)");
    report<262144 + 16 + 64 + 2 + 4 + 8 + 32>("SPEC all + priority for the M-waves (x 2)");
    report<262144 + 524288 + 16 + 64>("SPEC2 (W.H one tile ahead): MFMA + epilogue + hand-off only (x 2)");
    report<262144 + 524288 + 16 + 64 + 2 + 4 + 8>("SPEC2 all (x 2)");
    report<262144 + 524288 + 16 + 64 + 2 + 4 + 8 + 32>("SPEC2 all + priority for the M-waves (x 2)");
    return 0;
}
