// Row pass, generation 5: the ping-pong kernel of mfma4.hip.h without a barrier in the loop.
//
// Measured on generation 4: one s_barrier per 32-column tile costs 18 % of the
// kernel (3.4 ms with, 2.8 ms without -- invalid results) even with V served from cache and even when
// only the four waves of a 4-wave workgroup meet at it: every tile all waves wait for the slowest one.
// The barrier is only there for the shared dictionary tiles in LDS:
//   (a) a wave may read tile t only after ALL waves' slices of its copy have landed,
//   (b) a wave may overwrite an LDS object only after ALL waves have finished reading its old tile.
// Both are turned into counters in LDS ("ready" per object: +1 per wave when its slices have landed;
// "done" per object: +1 per wave when it has finished the last read of the tile), with 8 rotating
// objects and copies issued three tiles ahead, so that a wave only ever waits when it is more than
// two tiles ahead of, or more than three tiles behind, the slowest / fastest wave -- in steady state
// the polls succeed at once and the waves drift freely (which also makes the matrix segment of one
// wave of a SIMD fall into the VALU segment of its partner without any forced schedule).
//
// Every poll is bounded: after kPollCap unsuccessful reads the wave gives up, marks the launch as failed
// (NaN loss partial -> the host sees a non-finite loss) and continues; the kernel cannot hang.
#pragma once
#include "mfma4.hip.h"

namespace klnmf {

constexpr int kObj5 = 8;                          // rotating dictionary tile objects
constexpr int kDist5 = 3;                         // a tile is copied this many tiles before its first use
constexpr int kPollCap = 1 << 14;

template <int KT, int ODD, int MODE>
__global__ __launch_bounds__(kThreads, 2) void k_rowpass5(RowPass4Args aa) {
    const RowPassArgs &a = aa.base;
    constexpr int KP = 32 * KT;
    constexpr int KS = 2 * KT - ODD;
    constexpr int WLD = w_ld(KP);
    constexpr int TB = 2048;
    constexpr int N1 = (MODE == ROW_INIT) ? 0 : KS;
    constexpr int N2 = (MODE == ROW_LOSS) ? 0 : 2 * KT;
    constexpr int NF = N1 + N2;
    constexpr int D = KL_PF < NF - 1 ? KL_PF : NF - 1;
    constexpr int R = D + 1;
    constexpr int DP = D < N2 ? D : N2;
    constexpr int IMG = KP * kRow4B;
    constexpr int NW = kWavesPerWG;                 // waves that share the objects
    static_assert(kObj5 * IMG + 4096 <= 160 * 1024, "dictionary tile ring exceeds LDS");
    __shared__ __attribute__((aligned(16))) unsigned char hring[kObj5 * IMG];
    __shared__ __attribute__((aligned(16))) float hsum_lds[KP];
    __shared__ unsigned flags[2 * kObj5 + 1];       // ready[0..7], done[8..15], failed
    if (a.st->stop) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bool lateY = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;     // waves 4-7 start half a tile late
    const int rt_raw = blockIdx.x * NW + wave;
    const bool active = rt_raw < a.nrt;
    const int rt = active ? rt_raw : a.nrt - 1;

    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    const unsigned off_tr0 = 2 * h4_elem(8 * h + tq, 16 * half + 4 * tp);
    const unsigned off_tr1 = 2 * h4_elem(8 * h + tq + 4, 16 * half + 4 * tp);
    const unsigned off_row0 = 2 * h4_elem(r, 4 * h);
    const unsigned off_row1 = 2 * h4_elem(r, 16 + 4 * h);

    bf16x8 wf[KS > 0 ? KS : 1];
    if (MODE != ROW_INIT) {
        const __bf16 *wrow = a.Wb_old + (int64_t)(rt * 32 + r) * WLD;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = *(const bf16x8 *)(wrow + wb_col(r, 16 * s + 8 * h));
    }
    f32x16 acc[KT];
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    float s1 = 0.f, s2 = 0.f;
    const float eps = a.eps;
    const unsigned char *ht = (const unsigned char *)aa.Ht4;
    const unsigned char *vt = (const unsigned char *)a.VtA + (int64_t)rt * a.nct * TB;
    const unsigned char *vlane = vt + lane * 32;
    const int nct = a.nct;

    const unsigned hbase = (unsigned)(uintptr_t)(KL_LDS unsigned char *)hring;
    const unsigned fbase = (unsigned)(uintptr_t)(KL_LDS unsigned *)flags;
    auto obj_of = [](int t) -> int { return t & (kObj5 - 1); };       // tile -1 -> object 7
    auto copy_tile = [&](int t) {                                      // this thread's slices of tile t
        glds_copy_exact<IMG>(ht + (int64_t)t * IMG, (KL_LDS unsigned char *)hring + obj_of(t) * IMG, tid);
    };
    // counters: one lane adds, every lane of a polling wave reads the same word (broadcast)
    auto signal = [&](unsigned word) {
#ifdef KL_ABL_NOSIGNAL
        return;
#endif
        if (lane == 0) {
            const unsigned one = 1u;
            asm volatile("ds_add_u32 %0, %1" ::"v"(fbase + 4 * word), "v"(one) : "memory");
        }
    };
    auto flag_read = [&](unsigned &dst, unsigned word) {               // issue only
        asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(fbase + 4 * word) : "memory");
    };
    bool failed = false;
    auto poll = [&](unsigned first, unsigned word, unsigned target) {  // `first`: a value already read
#ifdef KL_ABL_NOPOLL      // ablation build: no waiting on the counters (invalid results, timing only)
        return;
#endif
        unsigned v = __builtin_amdgcn_readfirstlane(first);
        int tries = 0;
        while (v < target) {
            __builtin_amdgcn_s_sleep(2);
            unsigned x;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(fbase + 4 * word) : "memory");
            v = __builtin_amdgcn_readfirstlane(x);
            if (++tries > kPollCap) { failed = true; break; }
        }
    };
    // targets: tile t is the (t/8 + 1)-th occupant of its object; object 7 was first occupied by the zero
    // image of "tile -1", whose readers also count as done
    auto ready_target = [&](int t) -> unsigned { return (unsigned)NW * (unsigned)(t / kObj5 + 1); };
    auto done_target = [&](int t) -> unsigned {     // before tile t may be copied into its object
        return (unsigned)NW * (unsigned)(t / kObj5 + ((N2 > 0 && obj_of(t) == kObj5 - 1) ? 1 : 0));
    };

    bf16x8 ring[R];
    f32x16 d;
    bf16x8 b0, b1;
    f16x8 vreg[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = 0.f;

    auto issue = [&](auto P, unsigned robj, unsigned tobj) {
        constexpr int p = decltype(P)::value;
        if constexpr (p < N2) {
            lds_read_b128<(32 * (p >> 1)) * kRow4B>(ring[p % R], robj + ((p & 1) ? off_row1 : off_row0));
        } else if constexpr (p < NF) {
            constexpr int s = p - N2;
            lds_read_tr_pair<(16 * s) * kRow4B>(ring[p % R], tobj + off_tr0, tobj + off_tr1);
        }
    };
    unsigned ready_pf = 0;          // ready counter of the tile of the next M segment, read ahead in the E segment
    unsigned done_pf = 0;           // done counter of the object the next E segment overwrites, read ahead likewise
    // M segment of tile t: MFMA-2 of tile t-1 (its image), then MFMA-1 of tile t.  LAST: only MFMA-2.
    auto seg_M = [&](int t, auto LAST) {
        constexpr bool last = decltype(LAST)::value;
        const unsigned robj = hbase + obj_of(t - 1) * IMG;
        const unsigned tobj = hbase + obj_of(t) * IMG;
        {
            // the newest tile this segment reads: t (MFMA-1) or, without MFMA-1, t-1.  Its ready counter was read
            // in the E segment before the MFMA-2 lead reads (in order): it has arrived once at most DP LGKM
            // operations are outstanding
            const int need = (N1 > 0 && !last) ? t : t - 1;
            // both counters read ahead in the E segment are older than its DP lead reads (in-order LDS)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ready_pf), "+v"(done_pf) : "n"(DP));
            if (need >= 0 && (N1 == 0 || !last)) poll(ready_pf, obj_of(need), ready_target(need));
        }
        static_for<DP, D>([&](auto P) { if constexpr (!last || decltype(P)::value < N2) issue(P, robj, tobj); });
        static_for<0, NF>([&](auto P) {
            constexpr int p = decltype(P)::value;
            if constexpr (!last || p < N2) {
                if constexpr (!last || p + D < N2) issue(std::integral_constant<int, p + D>{}, robj, tobj);
                constexpr int lim = last ? N2 : NF;
                constexpr int lastf = (p + D < lim - 1) ? p + D : lim - 1;
                constexpr int n_b128 = (lastf < N2 ? lastf : N2 - 1) - p > 0 ? (lastf < N2 ? lastf : N2 - 1) - p : 0;
                constexpr int n_tr = (lastf - p) - n_b128;
                lds_wait<n_b128 + 2 * n_tr>(ring[p % R]);
                if constexpr (p < N2) {
                    acc[p >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[p % R], (p & 1) ? b1 : b0, acc[p >> 1], 0, 0, 0);
                } else {
                    if constexpr (p == N2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) d[e] = 0.f;
                    }
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[p % R], wf[p - N2], d, 0, 0, 0);
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        // every read of the oldest tile this wave still used has been consumed by an MFMA
        if (N2 > 0) signal(kObj5 + obj_of(t - 1));
        else if (!last) signal(kObj5 + obj_of(t));
    };
    // E segment of tile t (TS: parity of t, static for the V register pair)
    auto seg_E = [&](auto TS, int t) {
        constexpr int ts = decltype(TS)::value;
        f16x8 &va = vreg[2 * (ts & 1)], &vb = vreg[2 * (ts & 1) + 1];
        const int tn = t + kDist5;                      // the tile this segment copies
        // everything in flight lands: V(t) and this wave's slices of tile t + kDist5 - 1 (issued one E ago)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(va), "+v"(vb)::"memory");
        if (t + kDist5 - 1 < nct && t >= 1) signal(obj_of(t + kDist5 - 1));
        v_tile_load(vreg[2 * ((ts + 1) & 1)], vreg[2 * ((ts + 1) & 1) + 1], vlane + (int64_t)min(t + 1, nct - 1) * TB);
        if (tn < nct) {
            // done_pf was read at the end of the previous E segment, ahead of a whole M segment of in-order LDS
            // reads that have all been waited for: it has arrived.  (A stale value only makes this conservative.)
            poll(done_pf, kObj5 + obj_of(tn), done_target(tn));
            copy_tile(tn);
        }
        float q[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = (float)(e < 8 ? va[e & 7] : vb[e & 7]);
            if (MODE == ROW_INIT) {
                q[e] = x;
            } else {
                const float rinv = __builtin_amdgcn_rcpf(d[e] + eps);
                q[e] = fmaf(x, rinv, eps * rinv);
                s1 = fmaf(x, __builtin_amdgcn_logf(q[e]), s1);
            }
        }
        b0 = pack8(q);
        b1 = pack8(q + 8);
        asm volatile("" : "+v"(s1));
        __builtin_amdgcn_sched_barrier(0);
        // read ahead for the next M segment: its ready counter first, then the lead of its MFMA-2 reads
        {
            const int need = (N1 > 0) ? t + 1 : t;              // what the next M segment polls (see seg_M)
            if (need < nct) flag_read(ready_pf, obj_of(need));
            if (tn + 1 < nct) flag_read(done_pf, kObj5 + obj_of(tn + 1));
        }
        if constexpr (N2 > 0) {
            const unsigned robj = hbase + obj_of(t) * IMG;
            static_for<0, DP>([&](auto P) { issue(P, robj, 0u); });
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: counters, zero image of "tile -1" (object 7), tiles 0..2 and V tile 0; one barrier
    if (tid < 2 * kObj5 + 1) flags[tid] = (tid < kDist5) ? (unsigned)NW : 0u;     // tiles 0..2 are ready after the barrier
    for (int t = 0; t < kDist5; ++t) copy_tile(min(t, nct - 1));
    v_tile_load(vreg[0], vreg[1], vlane);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        KL_LDS u32x4 *zi = (KL_LDS u32x4 *)(hring + (kObj5 - 1) * IMG);
        for (int e = tid; e < IMG / 16; e += kThreads) zi[e] = z;
        if (MODE != ROW_INIT && tid < KP) hsum_lds[tid] = a.hsum[tid];
#pragma unroll
        for (int j = 0; j < 8; ++j) { b0[j] = (__bf16)0.f; b1[j] = (__bf16)0.f; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(vreg[0]), "+v"(vreg[1])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (N1 > 0) flag_read(ready_pf, 0u);          // (without MFMA-1 the first M segment reads only the zero image)
    if (kDist5 < nct) flag_read(done_pf, kObj5 + obj_of(kDist5));
    if constexpr (N2 > 0) {
        const unsigned robj = hbase + (kObj5 - 1) * IMG;
        static_for<0, DP>([&](auto P) { issue(P, robj, 0u); });
    }
    if (lateY) {                    // start waves 4-7 roughly one M segment late (drift does the rest)
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(2);
    }
    // ---- main loop: two tiles per body (V register pairs alternate); nct is a multiple of 4
    for (int t2 = 0; t2 < nct; t2 += 2) {
        seg_M(t2, std::false_type{});
        seg_E(std::integral_constant<int, 0>{}, t2);
        seg_M(t2 + 1, std::false_type{});
        seg_E(std::integral_constant<int, 1>{}, t2 + 1);
    }
    seg_M(nct, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (!active) return;
    if (MODE != ROW_INIT) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) s2 = fmaf((float)wf[s][j], hsum_lds[16 * s + 8 * h + j], s2);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (failed) s1 = __builtin_nanf("");            // a poll gave up: make the launch visibly invalid
        if (lane == 0) a.loss_part[rt] = make_float2(s1, s2);
    }
    if (MODE != ROW_LOSS) {
        const int64_t row = (int64_t)rt * 32 + r;
        f32x4 wold[KT][4];
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int comp = 32 * m + 8 * g + 4 * h;
                if (MODE == ROW_UPDATE) {
                    wold[m][g] = *(const f32x4 *)(a.W32_old + row * KP + comp);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) wold[m][g][t] = 1.f;
                }
            }
        if (MODE == ROW_UPDATE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int comp = 32 * m + 8 * g + 4 * h;
                f32x4 w = wold[m][g];
#pragma unroll
                for (int t = 0; t < 4; ++t) w[t] *= acc[m][4 * g + t];
                if (failed) w[0] = __builtin_nanf("");
                *(f32x4 *)(a.W32_new + row * KP + comp) = w;
                bf16x4 wb;
#pragma unroll
                for (int t = 0; t < 4; ++t) wb[t] = (__bf16)w[t];
                *(bf16x4 *)(a.Wb_new + row * WLD + wb_col(r, comp)) = wb;
            }
    }
}

}  // namespace klnmf
