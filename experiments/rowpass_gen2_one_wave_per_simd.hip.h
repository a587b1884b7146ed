// Second-generation bf16-MFMA passes: one wave per SIMD, software-pipelined.
//
// PMC on the first generation (profiles/r01_*): two waves per SIMD running the
// same barrier-synchronised program stay in lockstep, so their MFMA phases
// collide and then their VALU phases (ratio / log epilogue) collide -- only 8 %
// of cycles had VALU and MFMA co-executing and the matrix pipe sat at 38-47 %.
// Here each SIMD runs ONE wave (4 waves / workgroup, 512-VGPR budget) whose
// instruction stream itself interleaves the phases of neighbouring tiles:
//
//     iteration T:  slots 0..KS-1     MFMA1(T+1)[s]  +  ratio VALU of tile T
//                   slots KS..N-1     MFMA2(T)[j]    +  loss  VALU of tile T
//
// An MFMA occupies the matrix pipe for 32 cycles but vector issue for only 8, so
// the ~5 VALU instructions placed behind it in program order issue in its shadow.
// Every slot also issues the LDS operand read of the MFMA five slots ahead (6-slot
// register ring, static indices: the loop body is 3 stages = 6 tiles, 6N slots).
// `sched_barrier(0)` between slots pins this order; inside a slot hipcc is free.
//
// Memory: three distinct LDS stage objects ([dictionary image | the 4 waves' V
// tiles]); every byte arrives by global_load_lds; distinct objects let hipcc prove
// that the ds_reads of one stage do not alias the DMA into another (no vmcnt drain
// before reads).  One barrier per stage (64 feature columns).
#pragma once
#include "mfma.hip.h"

namespace klnmf {

constexpr int kWaves2 = 4;
constexpr int kThreads2 = 64 * kWaves2;          // 256
constexpr int kGlds2 = kThreads2 * 16;           // 4 KiB per global_load_lds round
constexpr int kRing = 6;                         // operand-fragment ring (read distance 5)
constexpr int kColsPerBody = 3 * kStageCols;     // the row-pass loop body covers 192 columns

__host__ __device__ constexpr int h_img2(int kp) { return round_up(kp * kHRowB, kGlds2); }
__host__ __device__ constexpr int row2_obj(int kp) { return h_img2(kp) + kWaves2 * 2 * 2048; }

__device__ __forceinline__ void glds_copy2(const unsigned char *gsrc, KL_LDS unsigned char *ldst,
                                           int rounds, int tid) {
    const int wave_base = (tid & ~63) * 16;
#pragma unroll
    for (int r = 0; r < rounds; ++r)
        __builtin_amdgcn_global_load_lds((const KL_GLB void *)(gsrc + r * kGlds2 + tid * 16),
                                         (KL_LDS void *)(ldst + r * kGlds2 + wave_base), 16, 0, 0);
}

// ------------------------------------------------------------------ row pass ---
// 4 waves x 32 sample rows.  V storage: fp16 (scaled), staged through LDS.
// Requires nst % 3 == 0 (the host pads the feature axis to a multiple of 192).
template <int KT, int ODD, int MODE>
__global__ __launch_bounds__(kThreads2, 1) void k_rowpass2(RowPassArgs a) {
    constexpr int KP = 32 * KT;
    constexpr int KS = 2 * KT - ODD;
    constexpr int IMG = h_img2(KP);
    constexpr int ROUNDS = IMG / kGlds2;
    constexpr int OBJ = row2_obj(KP);
    constexpr int WLD = w_ld(KP);
    constexpr int TB = 2048;                               // one fp16 V tile
    constexpr int N1 = (MODE == ROW_INIT) ? 0 : KS;        // MFMA1 slots per iteration
    constexpr int N2 = (MODE == ROW_LOSS) ? 0 : 2 * KT;    // MFMA2 slots per iteration
    constexpr int N = N1 + N2;
    constexpr int D = (N < kRing - 1) ? N : kRing - 1;   // never read further ahead than the next tile
    __shared__ __attribute__((aligned(16))) unsigned char obj0[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char obj1[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char obj2[OBJ];
    if (a.st->stop) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rt_raw = blockIdx.x * kWaves2 + wave;
    const bool active = rt_raw < a.nrt;
    const int rt = active ? rt_raw : a.nrt - 1;

    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    const int off_tr = (8 * h + tq) * kHRowB + (16 * half + 4 * tp) * 2;
    const int off_row = r * kHRowB + (4 * h) * 2;
    const int voff = IMG + wave * 2 * TB + lane * 16;

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
    KL_LDS unsigned char *O0 = (KL_LDS unsigned char *)obj0;
    KL_LDS unsigned char *O1 = (KL_LDS unsigned char *)obj1;
    KL_LDS unsigned char *O2 = (KL_LDS unsigned char *)obj2;

    auto stage_in = [&](KL_LDS unsigned char *obj, int st) {
        st = min(st, a.nst - 1);                           // past the end: re-copy the last stage (unused)
        glds_copy2(ht + (int64_t)st * h_stage_bytes(KP), obj, ROUNDS, tid);
        const unsigned char *vn = vt + (int64_t)(2 * st) * TB;
        stage_v_tile<32>(vn, obj + IMG + wave * 2 * TB, lane);
        stage_v_tile<32>(vn + TB, obj + IMG + wave * 2 * TB + TB, lane);
    };
    // object / tile-in-stage of body tile t (t may run past 5: wraps into the next body)
    auto obj_of = [&](int t) -> const KL_LDS unsigned char * {
        const int o = (t % 6) / 2;
        return o == 0 ? O0 : (o == 1 ? O1 : O2);
    };

    // The operand fragment consumed at (body tile t, slot j):
    //   j <  N1 : MFMA1 of tile t+1, k-step j      (transposed read of that tile's image)
    //   j >= N1 : MFMA2 of tile t,  fragment j-N1  (row read)
    bf16x8 ring[kRing];
    auto fetch = [&](int t, int j) {                       // all arguments are constants after unrolling
        t += j / N;
        j = j % N;
        const int g = (t * N + j) % kRing;
        if (j < N1) {
            const KL_LDS unsigned char *p = obj_of(t + 1) + off_tr + (32 * ((t + 1) & 1)) * 2 + (16 * j) * kHRowB;
            ring[g] = tr_pair(p, p + 4 * kHRowB);
        } else {
            const int jj = j - N1, m = jj >> 1, hh = jj & 1;
            const KL_LDS unsigned char *p = obj_of(t) + off_row + (32 * (t & 1)) * 2 + (32 * m) * kHRowB + 32 * hh;
            ring[g] = b64_pair(p, p + 16);
        }
    };

    f32x16 dA, dB;                 // W.H accumulators of the even / odd tile of a stage

    // ---- prologue: three stages in flight; MFMA1 of tile 0 un-pipelined; ring primed
    stage_in(O0, 0);
    stage_in(O1, 1);
    stage_in(O2, 2);
    __syncthreads();
    if (MODE != ROW_INIT) {
        const KL_LDS unsigned char *p1 = O0 + off_tr;
#pragma unroll
        for (int e = 0; e < 16; ++e) dA[e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 f = tr_pair(p1 + (16 * s) * kHRowB, p1 + (16 * s + 4) * kHRowB);
            dA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, wf[s], dA, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) fetch(0, j);

    // ---- one pipelined iteration (body tile t)
    auto iter = [&](int t, f32x16 &d_this, f32x16 &d_next) {
        const KL_LDS unsigned char *cur = obj_of(t);
        const int u = t & 1;
        // this tile's 16 V values: two 16-byte LDS reads, converted element by element below
        const f16x8 va = *(const KL_LDS f16x8 *)(cur + voff + u * TB);
        const f16x8 vb = *(const KL_LDS f16x8 *)(cur + voff + u * TB + 1024);
        float x[16], q[16];
        bf16x8 b0, b1;
        auto ratio = [&](int e) {
            x[e] = (float)(e < 8 ? va[e & 7] : vb[e & 7]);
            if (MODE == ROW_INIT) q[e] = x[e];
            else q[e] = (x[e] + eps) * __builtin_amdgcn_rcpf(d_this[e] + eps);
        };
        auto lossterm = [&](int e) {
            if (MODE != ROW_INIT) s1 = fmaf(x[e], __builtin_amdgcn_logf(q[e]), s1);
        };
        // split of the 16 elements over the slots of each phase
        constexpr int P1 = (N1 > 0) ? N1 : N2;             // slots that carry the ratio work
#pragma unroll
        for (int j = 0; j < N; ++j) {
            fetch(t, j + D);
            if (j < N1) {
                if (j == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) d_next[e] = 0.f;
                }
                d_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(t * N + j) % kRing], wf[j], d_next, 0, 0, 0);
            } else {
                const int jj = j - N1;
                acc[jj >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(t * N + j) % kRing], (jj & 1) ? b1 : b0,
                                                                       acc[jj >> 1], 0, 0, 0);
            }
            // VALU in this MFMA's shadow
            if (N1 > 0) {
                if (j < N1) {
#pragma unroll
                    for (int e = (16 * j) / N1; e < (16 * (j + 1)) / N1; ++e) ratio(e);
                    if ((16 * (j + 1)) / N1 >= 8 && (16 * j) / N1 < 8) b0 = pack8(q);
                    if (j == N1 - 1) b1 = pack8(q + 8);
                } else {
                    const int jj = j - N1;
#pragma unroll
                    for (int e = (16 * jj) / N2; e < (16 * (jj + 1)) / N2; ++e) lossterm(e);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (N1 > 0 && N2 == 0) {                           // ROW_LOSS: no MFMA2 slots to hide the loss terms in
#pragma unroll
            for (int e = 0; e < 16; ++e) lossterm(e);
        }
        (void)P1;
    };
    // ROW_INIT has no MFMA1: its B operands must exist before the first MFMA2 slot
    auto iter_init = [&](int t) {
        const KL_LDS unsigned char *cur = obj_of(t);
        const int u = t & 1;
        const f16x8 va = *(const KL_LDS f16x8 *)(cur + voff + u * TB);
        const f16x8 vb = *(const KL_LDS f16x8 *)(cur + voff + u * TB + 1024);
        bf16x8 b0, b1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { b0[e] = (__bf16)(float)va[e]; b1[e] = (__bf16)(float)vb[e]; }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            fetch(t, j + D);
            acc[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(t * N + j) % kRing], (j & 1) ? b1 : b0,
                                                                  acc[j >> 1], 0, 0, 0);
        }
    };

    // ---- main loop: 3 stages = 6 tiles per body, no branches inside
    for (int st = 0; st < a.nst; st += 3) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (MODE == ROW_INIT) {
                iter_init(2 * o);
                iter_init(2 * o + 1);
            } else {
                iter(2 * o, dA, dB);
                iter(2 * o + 1, dB, dA);
            }
            __syncthreads();       // everyone is done with object o; hipcc drains the DMAs (issued >= 1 stage ago)
            stage_in(o == 0 ? O0 : (o == 1 ? O1 : O2), st + o + 3);
        }
    }

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
