// Third-generation row pass: two waves per SIMD, phase-staggered.
//
// Why (profiles/r01_*): with two waves per SIMD running the same barrier-
// synchronised program (mfma.hip.h) the partners stay in lockstep -- both in
// their MFMA block, then both in their VALU block (ratio / log epilogue): only
// 8 % of cycles had VALU and MFMA co-executing, matrix pipe 38 %.  One wave per
// SIMD (mfma2.hip.h) trades that for a serial issue stream (~240 instructions per
// tile) and measured no better.  Here all 8 waves run the SAME cyclic sequence
//
//   ... M2(b,k-1) | M1(a,k) E(a,k) M2(a,k) M1(b,k) E(b,k) | M2(b,k) ...
//
// but take their single barrier per stage at different points of the cycle:
// waves 0-3 after M2(b,k-1), waves 4-7 before it.  A barrier only counts arrivals,
// so this is legal, and it shifts the two SIMD partners by one MFMA2 block: while
// one issues MFMAs the other is mostly in its epilogue.  M2(b,k-1) then reads the
// PREVIOUS stage's dictionary image after the barrier, hence three image objects.
//
// LDS: 3 dictionary images + 2 V areas, five distinct objects (distinct objects let
// hipcc prove that ds_reads do not alias the global_load_lds in flight); the loop
// body covers 6 stages so every object index is static.  KT <= 7 (160 KiB).
#pragma once
#include "mfma.hip.h"

namespace klnmf {

constexpr int kRow3Cols = 3 * kStageCols;          // feature axis padded to a multiple of 192

template <int KT, int ODD, int MODE>
__global__ __launch_bounds__(kThreads, 2) void k_rowpass3(RowPassArgs a) {
    constexpr int KP = 32 * KT;
    constexpr int KS = 2 * KT - ODD;
    constexpr int STG = h_stage_lds(KP);
    constexpr int ROUNDS = STG / kGldsRound;
    constexpr int WLD = w_ld(KP);
    constexpr int TB = 2048;                               // one fp16 V tile
    constexpr int VAREA = kWavesPerWG * 2 * TB;
    typedef VTraits<_Float16> VTr;
    constexpr int N1 = (MODE == ROW_INIT) ? 0 : KS;        // MFMA1 fragments per tile
    constexpr int N2 = (MODE == ROW_LOSS) ? 0 : 2 * KT;    // MFMA2 fragments per tile
    constexpr int N = N1 + N2;
    __shared__ __attribute__((aligned(16))) unsigned char h0[STG];
    __shared__ __attribute__((aligned(16))) unsigned char h1[STG];
    __shared__ __attribute__((aligned(16))) unsigned char h2[STG];
    __shared__ __attribute__((aligned(16))) unsigned char v0[VAREA];
    __shared__ __attribute__((aligned(16))) unsigned char v1[VAREA];
    if (a.st->stop) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // waves 4-7 (the SIMD partners of 0-3).  Kept in an SGPR and re-laundered through an empty asm
    // at each test so hipcc cannot correlate the two tests of a stage and clone the code per group.
    int grp_s = __builtin_amdgcn_readfirstlane(tid >> 8);   // 0 or 1, wave-uniform
    auto is_y = [&]() -> bool {
        asm volatile("; group test" : "+s"(grp_s));
        return grp_s != 0;
    };
    const int rt_raw = blockIdx.x * kWavesPerWG + wave;
    const bool active = rt_raw < a.nrt;
    const int rt = active ? rt_raw : a.nrt - 1;

    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    const int off_tr = (8 * h + tq) * kHRowB + (16 * half + 4 * tp) * 2;
    const int off_row = r * kHRowB + (4 * h) * 2;
    const int voff = wave * 2 * TB;

    bf16x8 wf[KS];
    if (MODE != ROW_INIT) {
        const __bf16 *wrow = a.Wb_old + (int64_t)(rt * 32 + r) * WLD + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = *(const bf16x8 *)(wrow + 16 * s);
    }
    f32x16 acc[KT];
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    float s1 = 0.f, s2 = 0.f;
    const float eps = a.eps;

    const unsigned char *ht = (const unsigned char *)a.Ht;
    const unsigned char *vt = (const unsigned char *)a.VtA + (int64_t)rt * a.nct * TB;

    auto Hobj = [&](int k) -> KL_LDS unsigned char * {      // k: stage index within the 6-stage body (constant)
        const int o = ((k % 3) + 3) % 3;
        return (KL_LDS unsigned char *)(o == 0 ? h0 : (o == 1 ? h1 : h2));
    };
    auto Vobj = [&](int k) -> KL_LDS unsigned char * {
        return (KL_LDS unsigned char *)((k & 1) ? v1 : v0);
    };
    auto stage_in = [&](int k, int st) {                    // copies of global stage `st` into the objects of body stage k
        st = min(st, a.nst - 1);                            // past the end: re-copy the last stage (never read)
        glds_copy(ht + (int64_t)st * h_stage_bytes(KP), Hobj(k), ROUNDS, tid);
        const unsigned char *vn = vt + (int64_t)(2 * st) * TB;
        stage_v_tile<32>(vn, Vobj(k) + voff, lane);
        stage_v_tile<32>(vn + TB, Vobj(k) + voff + TB, lane);
    };

    // Operand fragments of body stage k in consumption order (position p):
    //   [0,N1) MFMA1 tile a | [N1,N) MFMA2 tile a | [N,N+N1) MFMA1 tile b | [N+N1,2N) MFMA2 tile b
    // read two positions ahead into a 3-slot register ring (slot = p % 3).
    bf16x8 ring[3];
    auto fetch = [&](int k, int p) {
        if (p >= 2 * N) return;
        const int u = p >= N ? 1 : 0;
        const int q = p - u * N;
        const KL_LDS unsigned char *img = Hobj(k);
        if (q < N1) {
            const KL_LDS unsigned char *pp = img + off_tr + (32 * u) * 2 + (16 * q) * kHRowB;
            ring[p % 3] = tr_pair(pp, pp + 4 * kHRowB);
        } else {
            const int j = q - N1, m = j >> 1, hh = j & 1;
            const KL_LDS unsigned char *pp = img + off_row + (32 * u) * 2 + (32 * m) * kHRowB + 32 * hh;
            ring[p % 3] = b64_pair(pp, pp + 16);
        }
    };

    bf16x8 cb0, cb1;                                        // Q operands of tile b, carried to the next stage's M2

    auto mfma2 = [&](int k, int u, const bf16x8 &b0, const bf16x8 &b1) {
        const int base = u * N + N1;
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            fetch(k, base + j + 2);
            acc[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(base + j) % 3], (j & 1) ? b1 : b0,
                                                                  acc[j >> 1], 0, 0, 0);
        }
    };
    // MFMA1 + epilogue of tile u of body stage k -> Q operands (b0, b1)
    auto front = [&](int k, int u, bf16x8 &b0, bf16x8 &b1) {
        float x[16], q[16];
        const typename VTr::Regs vr = VTr::load_lds(Vobj(k) + voff + u * TB, lane);
        VTr::unpack(vr, x);
        if (MODE == ROW_INIT) {
#pragma unroll
            for (int e = 0; e < 16; ++e) q[e] = x[e];
        } else {
            f32x16 d;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                fetch(k, u * N + s + 2);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(u * N + s) % 3], wf[s], d, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float qq = (x[e] + eps) * __builtin_amdgcn_rcpf(d[e] + eps);
                q[e] = qq;
                s1 = fmaf(x[e], __builtin_amdgcn_logf(qq), s1);
            }
            // pin the loss terms to this tile: otherwise hipcc sinks the log/fma of several
            // tiles to the end of the block and keeps all their x/q registers alive (spills)
            asm volatile("" : "+v"(s1));
        }
        b0 = pack8(q);
        b1 = pack8(q + 8);
    };

    // one stage of the cyclic sequence; `carried`: a tile b of the previous stage awaits its M2
    auto stage = [&](int k, int st, bool carried) {
        if (is_y()) {                                       // waves 4-7 synchronise BEFORE the carried MFMA2
            __syncthreads();
            stage_in(k + 1, st + 1);
        }
        if (MODE != ROW_LOSS && carried) mfma2(k - 1, 1, cb0, cb1);
        if (!is_y()) {                                      // waves 0-3 after it
            __syncthreads();
            stage_in(k + 1, st + 1);
        }
        fetch(k, 0);
        fetch(k, 1);
        bf16x8 b0, b1;
        front(k, 0, b0, b1);
        if (MODE != ROW_LOSS) mfma2(k, 0, b0, b1);
        front(k, 1, cb0, cb1);
    };

    stage_in(0, 0);
    for (int st = 0; st < a.nst; st += 6) {                 // nst is a multiple of 3
        stage(0, st, st > 0);
        stage(1, st + 1, true);
        stage(2, st + 2, true);
        if (st + 3 < a.nst) {
            stage(3, st + 3, true);
            stage(4, st + 4, true);
            stage(5, st + 5, true);
        }
    }
    if (MODE != ROW_LOSS) mfma2(2, 1, cb0, cb1);            // the last stage is body stage 2 or 5: image object 2

    if (!active) return;
    if (MODE != ROW_INIT) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) s2 = fmaf((float)wf[s][j], a.hsum[16 * s + 8 * h + j], s2);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) a.loss_part[rt] = make_float2(s1, s2);
    }
    if (MODE != ROW_LOSS) {
        const int64_t row = (int64_t)rt * 32 + r;
#pragma unroll
        for (int m = 0; m < KT; ++m) {
            f32x4 w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int comp = 32 * m + 8 * g + 4 * h;
                if (MODE == ROW_UPDATE) {
                    w[g] = *(const f32x4 *)(a.W32_old + row * KP + comp);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) w[g][t] = 1.f;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int comp = 32 * m + 8 * g + 4 * h;
#pragma unroll
                for (int t = 0; t < 4; ++t) w[g][t] *= acc[m][4 * g + t];
                *(f32x4 *)(a.W32_new + row * KP + comp) = w[g];
                bf16x4 wb;
#pragma unroll
                for (int t = 0; t < 4; ++t) wb[t] = (__bf16)w[g][t];
                *(bf16x4 *)(a.Wb_new + row * WLD + comp) = wb;
            }
        }
    }
}

}  // namespace klnmf
