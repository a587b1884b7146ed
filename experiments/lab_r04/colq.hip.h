// Column pass on stored ratios, pipelined (the product kernel of the H rule in the bf16 mode).
//
//   numer[a][j] = sum_i W_new[i][a] * Q[i][j]          (nmf.py:347-349, Q = ratio of the OLD W and H)
//
// Q comes from the ping-pong row pass (mfma4.hip.h), which stores its packed bf16 MFMA operands as it applies
// the W rule; see k_colpass_q in mfma.hip.h for the tile layout (Qt) and for how the transposition
// "row-pass lane = one row, several columns" -> "this kernel's lane = one column, several rows" happens on
// the way through LDS (regrouping global_load_lds copy + ds_read_b64_tr_b16).  That first kernel is kept as
// the readable reference of the data path (KLNMF_COLPASS=3); it runs one 64-row stage ahead and drains all
// copies at every stage barrier, so each stage pays the full HBM latency (2.1 us per stage measured,
// against 0.9 us of matrix work).  This one keeps NB-1 stages of 32 rows in flight:
//
//   * NB distinct LDS objects, each [W_new rows of the stage | the 8 waves' ratio tiles]; a workgroup is 8 waves,
//     wave w owns column tile 8*cb + w and all KP components (KT accumulator blocks);
//   * stage s computes on object s % NB while the copies of stages s+1 .. s+NB-1 are in flight; its copy of
//     stage s+NB-1 goes into the object stage s-1 used, which every wave has left (barrier at the end of s-1);
//   * every wave issues the same number of copy instructions per stage (the W_new copy is rounded up to whole
//     8 KiB rounds, reading into the next stage's rows), so "my copies of stage s+1 have landed" is the counted
//     wait vmcnt((NB-2) * OPS); the barrier after it makes that true for all waves.  Only LDS copies are in
//     flight in the loop (no VGPR loads, no stores): counted waits are valid among operations of one kind
//     (DESIGN.md section 8, h3);
//   * operand fragments are read with inline-asm LDS reads and counted lgkmcnt waits (helpers of mfma4.hip.h):
//     the compiler cannot track which object a copy is still in flight into and would drain all of them
//     before the first read.
//   * KSPLIT = 2 (KT > 8, k <= 512): the 8 waves are 4 column tiles x 2 halves of the component range, so that a
//     wave's accumulators (KT/2 blocks) fit 2 waves per SIMD; the two waves of a column tile read the same ratio
//     tile.  The W_new stage is 40 KiB at KP = 512 and only NB = 3 objects fit: two stages in flight.
#pragma once
#include "mfma4.hip.h"

// the ratio tiles are written once by the row pass and read once here; as non-temporal copies (-DKL_Q_NT=1) the fp8 x fp8 column
// pass ran 7 % SLOWER (0.92 vs 0.86 ms; the row pass behind it 2 % faster, the iteration equal): off (profiles/r03_ab_nontemporal.txt)
#ifndef KL_Q_NT
#define KL_Q_NT 0
#endif
#if KL_Q_NT
#define KL_Q_NT_MOD " nt"
#else
#define KL_Q_NT_MOD ""
#endif

namespace klnmf {

__host__ __device__ constexpr int colq_w_area(int kp) { return round_up(32 * w_ld(kp) * 2, kGldsRound); }
__host__ __device__ constexpr int colq_obj_bytes(int kp, int ksplit, int qtile = kQTile) { return colq_w_area(kp) + kWavesPerWG / ksplit * qtile; }
constexpr int kQTile8 = 1024;                    // bytes of one 32 x 32 fp8 ratio tile (mfma4.hip.h, Q8)
#ifndef KL_COLQ_NB
#define KL_COLQ_NB 4
#endif
#ifndef KL_COLQ8_PAIR       // fp8 ratio tiles (24 KiB objects at KT = 7): stages per fence and objects
#define KL_COLQ8_PAIR 1
#endif
#ifndef KL_COLQ8_NB
#define KL_COLQ8_NB (KL_COLQ8_PAIR == 2 ? 6 : 4)
#endif

// Ratio-tile bytes the H numerator cannot take as they are: 0x7E = e4m3 448 = a ratio of 3584 or MORE (the row pass's
// conversion saturates: the excess would be missing), and every byte >= 0x60 = a ratio >= 256 -- a single entry that large
// is not averaged over the rows any more (it can dominate its column's numerator: a spike the model has not fitted), so its
// 3-bit significand (+-6 %) would show.  The column passes only keep a branch-free sticky flag per feature column while they
// stream (a rarely-taken branch with a call inside the stage loop cost the fp8 x fp8 pass 12 %: profiles/r03_ab_colpass_detection.txt)
// and report (row chunk, column tile, physical column) SUSPECTS when they finish; k_q8_fixup (colq8x.hip.h) re-reads the
// suspects' tile bytes, recomputes every large ratio exactly from V and the masters and corrects the summed numerator.
__device__ __forceinline__ unsigned q8_sat_mask(unsigned dw) { return (dw + 0x20202020u) & 0x80808080u; }      // bytes <= 0x7E: no carries
__device__ __forceinline__ void q8_suspect_append(DevState *st, uint2 *list, int chunk, int ct, int pcol) {
    const int at = atomicAdd(&st->q8_list_n, 1);
    if (at < kQ8ListCap) list[at] = make_uint2((unsigned)chunk | ((unsigned)pcol << 16), (unsigned)ct);
}

__device__ __forceinline__ void lds_read_tr(s16x4 &dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}

// Q8: fp8 ratio tiles (1 KiB, row-major with the row pass's column permutation).  The tile is copied linearly (one piece
// per wave and stage instead of two); ds_read_b64_tr_b8 hands lane l the 8 rows 16 s + 8 h .. + 7 of PHYSICAL column l & 31
// (probed: experiments/micro/fp8_probe.hip), four v_cvt_scalef32_pk_f16_fp8 make the B operand of k-step s, the W_new
// fragments are read in the same row order (8 h + t, 8 h + 4 + t), and the accumulator of lane l belongs to the LOGICAL
// column 8 g + 4 h' + t of its physical column 16 h' + 4 g + t.
// PAIR = 2: two 32-row stages per fence (copies of the next PAIR stages issued, PAIR stages computed, then ONE counted wait +
// barrier): half the workgroup barriers per row; NB (even) objects hold NB / 2 such pairs, one pair in flight behind the
// pair being computed.
template <int KT, int NB, int KSPLIT = 1, int Q8 = 0, int PAIR = 1>
__global__ __launch_bounds__(kThreads, 1) void k_colpass_q2(ColPassQArgs a) {
    static_assert(kWavesPerWG == 8 && (KSPLIT == 1 || KSPLIT == 2) && KT % KSPLIT == 0, "wave decomposition");
    static_assert(Q8 == 0 || sizeof(opnd_t) == 2, "fp8 ratio tiles: 16-bit operand builds");
    // (Q8 with KSPLIT = 2 -- k > 256 -- exists only as the fallback of the fp8 x fp8 pass: both waves of a column tile copy
    // the same 1 KiB tile, so that every wave issues the same number of copies and the counted wait stays valid)
    constexpr int QTB = Q8 ? kQTile8 : kQTile;      // bytes of a ratio tile
    constexpr int CTW = kWavesPerWG / KSPLIT;      // column tiles per workgroup
    constexpr int KTW = KT / KSPLIT;               // accumulator blocks per wave
    constexpr int KP = 32 * KT;
    constexpr int WLD = w_ld(KP);
    constexpr int WLDB = WLD * 2;
    constexpr int WST = 32 * WLDB;                 // bytes of W_new per 32-row stage in global memory
    constexpr int WA = colq_w_area(KP);            // copied per stage (whole rounds)
    constexpr int OBJ = colq_obj_bytes(KP, KSPLIT, QTB);
    constexpr int QP = Q8 ? 1 : 2 / KSPLIT;        // 1 KiB pieces of its column tile's ratio tile a wave copies
    constexpr int OPS = WA / kGldsRound + QP;      // copy instructions per wave and stage
    constexpr int N3 = 2 * KTW;
    static_assert(NB >= 3 && NB <= 6 && NB * OBJ <= 160 * 1024, "LDS budget");
    static_assert((NB - 2) * OPS <= 63, "vmcnt range");
    static_assert(PAIR == 1 || (PAIR == 2 && NB % 2 == 0 && NB >= 4), "stage pairs");
    __shared__ __attribute__((aligned(16))) unsigned char o0[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char o1[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char o2[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char o3[NB > 3 ? OBJ : 16];
    __shared__ __attribute__((aligned(16))) unsigned char o4[NB > 4 ? OBJ : 16];
    __shared__ __attribute__((aligned(16))) unsigned char o5[NB > 5 ? OBJ : 16];
    if (a.st->stop) return;
    if (a.guard == 2 && a.st->w8_sat == 0) return;           // fallback of the fp8 x fp8 pass: only when its W image clipped
    if (a.guard == 2 && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&a.st_rw->w8_fallbacks, 1);
        atomicAdd(&a.st_rw->w8_sat_total, a.st->w8_sat);
    }
#ifdef KL_COL_PRIO       // experiment: static priority for the second-dispatched half of the workgroup
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(KL_COL_PRIO);
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: scalar
    const int r = lane & 31, h = lane >> 5;
    const int G = gridDim.x;                       // XCD-aware block -> (row chunk, column block)
    int lin = blockIdx.x;
    if ((G & 7) == 0) lin = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = lin / a.ncb, cb = lin % a.ncb;
    const int ctl = wave % CTW, kh = wave / CTW;   // column tile inside the workgroup, half of the component range
    const int ct_raw = cb * CTW + ctl;
    const bool active = ct_raw < a.nct;            // wave-uniform
    const int ct = active ? ct_raw : a.nct - 1;
    // stages of 32 rows; the chunk decomposition of the host counts 64-row stages
    const int sbeg = 2 * chunk * a.stages_per_chunk;
    const int send = min(a.nrt, sbeg + 2 * a.stages_per_chunk);
    if (sbeg >= send) {                            // empty chunk: its slab of partials must still be defined
        if (active) {
            float *np = a.Npart + (int64_t)chunk * KP * a.f_pad + (int64_t)ct * 32 + r;
            for (int c = 32 * KTW * kh + h; c < 32 * KTW * (kh + 1); c += 2) np[(int64_t)c * a.f_pad] = 0.f;
        }
        return;
    }

    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    //  W_new (A operand): rows = samples 4h+tq (+8 for the second read), cols = components
    const int ra0 = Q8 ? 8 * h + tq : 4 * h + tq, ra1 = Q8 ? 8 * h + tq + 4 : 4 * h + tq + 8;      // rows of the two A reads
    const unsigned off_tr0 = ra0 * WLDB + 2 * wb_col(ra0, 16 * half + 4 * tp) + 64 * KTW * kh;
    const unsigned off_tr1 = ra1 * WLDB + 2 * wb_col(ra1, 16 * half + 4 * tp) + 64 * KTW * kh;
    //  ratios (B operand): row 4h+tq (+8j), columns 16*half + 4*tp.. = slot 2*row + (tp&1), group 2*half + (tp>>1)
    //  Q8: lane 2q + p of a 16-lane group addresses row 8h + q, physical columns 16*half + 8p .. + 7 of the row-major tile
    const unsigned off_q = Q8 ? WA + ctl * QTB + (8 * h + (i16 >> 1)) * 32 + 16 * half + 8 * (i16 & 1)
                              : WA + ctl * QTB + (2 * (4 * h + tq) + (tp & 1)) * 32 + (2 * half + (tp >> 1)) * 8;

    f32x16 acc[KTW];
#pragma unroll
    for (int m = 0; m < KTW; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    const unsigned char *wn = (const unsigned char *)a.Wb_new;
    // copy piece P = 64p + lane of a tile (16 bytes at LDS offset 16P): slot P>>1, operand P&1
    const unsigned char *const qt_s = a.Qt + (int64_t)ct * a.nrt * QTB;       // wave-uniform part
    const unsigned ql32 = Q8 ? (unsigned)lane * 16u : (unsigned)((lane & 1) * 1024 + ((lane >> 2) + 32 * ((lane >> 1) & 1)) * 16);
    const unsigned char *qt = qt_s + ql32;

    auto obj = [&](int o) -> KL_LDS unsigned char * {      // o static after unrolling
        return (KL_LDS unsigned char *)(o == 0 ? o0 : o == 1 ? o1 : o == 2 ? o2 : o == 3 ? o3 : o == 4 ? o4 : o5);
    };
    auto lds_addr = [](KL_LDS unsigned char *p) -> unsigned { return (unsigned)(uintptr_t)p; };
    auto stage_in = [&](int o, int sg) {
        sg = min(sg, send - 1);                    // past the end: re-copy the last stage (uniform instruction count)
#if !(KL_SADDR & 1)
        glds_copy_exact<WA, kWavesPerWG>(wn + (int64_t)sg * WST, obj(o), tid);
        const unsigned char *qs = qt + (int64_t)sg * QTB;
#pragma unroll
        for (int pp = 0; pp < QP; ++pp) {
            const int p = Q8 ? pp : kh * QP + pp;  // KSPLIT = 2: the two waves of a column tile copy one piece each (Q8: the same tile)
            __builtin_amdgcn_global_load_lds((const KL_GLB void *)(qs + p * 256),
                                             (KL_LDS void *)(obj(o) + WA + ctl * QTB + 1024 * p), 16, 0, 0);
        }
#else
        // scalar base + this lane's 32-bit offset, M0 from scalars (mfma4.hip.h `dma`)
        const unsigned char *wbase = wn + (int64_t)sg * WST;
        const unsigned t16 = (unsigned)tid * 16u, m0w = lds_addr(obj(o)) + (unsigned)wave * 1024u;
#pragma unroll
        for (int rr = 0; rr < WA / kGldsRound; ++rr)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0w + rr * kGldsRound), "v"(t16), "s"(wbase + rr * kGldsRound) : "memory");
        const unsigned char *qbase = qt_s + (int64_t)sg * QTB;
#pragma unroll
        for (int pp = 0; pp < QP; ++pp) {
            const int p = Q8 ? pp : kh * QP + pp;  // KSPLIT = 2: the two waves of a column tile copy one piece each (Q8: the same tile)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" KL_Q_NT_MOD ::"s"(lds_addr(obj(o)) + WA + ctl * QTB + 1024 * p), "v"(ql32), "s"(qbase + p * 256) : "memory");
        }
#endif
    };
    // one copy instruction of stage sg into object o (the order stage_in issues them in: W_new rounds, then ratio pieces)
    constexpr int WR = WA / kGldsRound;
    static_assert(WR + QP == OPS, "copy instructions per wave and stage");
    auto piece = [&](int o, int sg, auto I) {
        constexpr int i = decltype(I)::value;
        sg = min(sg, send - 1);
        if constexpr (i < WR) {
            __builtin_amdgcn_global_load_lds((const KL_GLB void *)(wn + (int64_t)sg * WST + i * kGldsRound + tid * 16),
                                             (KL_LDS void *)(obj(o) + i * kGldsRound + (tid & ~63) * 16), 16, 0, 0);
        } else {
            const int p = Q8 ? (i - WR) : kh * QP + (i - WR);
            __builtin_amdgcn_global_load_lds((const KL_GLB void *)(qt + (int64_t)sg * QTB + p * 256),
                                             (KL_LDS void *)(obj(o) + WA + ctl * QTB + 1024 * p), 16, 0, 0);
        }
    };
    unsigned q8_flag = 0u;             // fp8 tiles: sticky "a byte >= 0x60 passed through this lane" (its feature column: physical column r)
    // o_next >= 0: the copies of stage sg_next into object o_next are issued BETWEEN this stage's MFMAs (KL_COLQ_INTERLEAVE)
    auto compute = [&](unsigned base, int stage_row0, int o_next = -1, int sg_next = 0) {
        (void)stage_row0;
        opx8 ring[3];
        s16x4 q0, q1, q2, q3;
        if constexpr (Q8 != 0) {                   // q0 / q2: the 8 fp8 values of k-step 0 / 1 (two dwords each)
            asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(q0) : "v"(base + off_q));
            asm volatile("ds_read_b64_tr_b8 %0, %1 offset:512" : "=v"(q2) : "v"(base + off_q));
        } else {
            lds_read_tr(q0, base + off_q);
            lds_read_tr(q1, base + off_q + 8 * 64);
            lds_read_tr(q2, base + off_q + 16 * 64);
            lds_read_tr(q3, base + off_q + 24 * 64);
        }
        const unsigned t0 = base + off_tr0, t1 = base + off_tr1;
        auto fetch = [&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < N3) lds_read_tr_pair<(16 * (j & 1)) * WLDB + 64 * (j >> 1)>(ring[j % 3], t0, t1);
        };
        fetch(std::integral_constant<int, 0>{});
        fetch(std::integral_constant<int, 1>{});
        // the ratio reads are older than every W_new read: they have landed when fragment 0 has
        if constexpr (Q8 != 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(q0), "+v"(q2), "+v"(ring[0]));
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(ring[0]));
        opx8 b0, b1;
        if constexpr (Q8 != 0) {
#ifndef KL_OPND_BF16
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
            const u32x2 w0 = __builtin_bit_cast(u32x2, q0), w1 = __builtin_bit_cast(u32x2, q2);
            // large / saturated ratio bytes (ratio >= 256 / >= 3584) of this lane's column: sticky, reported at the kernel's end
            q8_flag |= q8_sat_mask(w0[0]) | q8_sat_mask(w0[1]) | q8_sat_mask(w1[0]) | q8_sat_mask(w1[1]);
            // rows 8h + 2j, 8h + 2j + 1 of k-step 0 (b0) and of k-step 1 (b1); the byte-pair selector must be a literal
#define KL_Q8_PAIR(dst, src, j, sel)                                                              \
            { const f16x2 p_ = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(src, kQ8Scale, sel); dst[2 * (j)] = p_[0]; dst[2 * (j) + 1] = p_[1]; }
            KL_Q8_PAIR(b0, w0[0], 0, false) KL_Q8_PAIR(b0, w0[0], 1, true) KL_Q8_PAIR(b0, w0[1], 2, false) KL_Q8_PAIR(b0, w0[1], 3, true)
            KL_Q8_PAIR(b1, w1[0], 0, false) KL_Q8_PAIR(b1, w1[0], 1, true) KL_Q8_PAIR(b1, w1[1], 2, false) KL_Q8_PAIR(b1, w1[1], 3, true)
#undef KL_Q8_PAIR
#endif
        } else {
            // rows {4h+t, 8+4h+t} and {16+4h+t, 24+4h+t} of column r: the contraction order of the W_new reads
            b0 = __builtin_bit_cast(opx8, __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7));
            b1 = __builtin_bit_cast(opx8, __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        static_for<0, N3>([&](auto J) {
            constexpr int j = decltype(J)::value;
            fetch(std::integral_constant<int, j + 2>{});
            constexpr int younger = (j + 2 < N3 ? 2 : N3 - 1 - j);      // fragments issued after fragment j
            lds_wait<2 * younger>(ring[j % 3]);
#ifdef KL_EMU_W8          // precision experiment: the W_new operand rounded to fp8 (e4m3, image / 256, saturating) as an fp8 column pass would see it
            {
                typedef __attribute__((ext_vector_type(2))) short s16x2_;
                typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
                opx8 &fr = ring[j % 3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s16x2_ w8 = {0, 0};
                    w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w8, f16x2_{fr[2 * u], fr[2 * u + 1]}, 256.f, false);
                    const f16x2_ back = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(__builtin_bit_cast(unsigned, w8), 256.f, false);
                    fr[2 * u] = back[0]; fr[2 * u + 1] = back[1];
                }
            }
#endif
            acc[j >> 1] = KL_MFMA_BUILTIN(ring[j % 3], (j & 1) ? b1 : b0, acc[j >> 1], 0, 0, 0);
#ifdef KL_COLQ_INTERLEAVE
            static_for<0, OPS>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int pos = (i * N3) / OPS < N3 - 1 ? (i * N3) / OPS : N3 - 1;
                if constexpr (pos == j) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (o_next >= 0) piece(o_next, sg_next, I);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
#endif
        });
    };
    auto fence = [&]() {          // my copies of the next stage (pair of stages) have landed; then everybody's
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NB - 2 * PAIR) * OPS) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: stages sbeg .. sbeg+NB-PAIR-1 in flight, the first one (pair) awaited
    static_for<0, NB - PAIR>([&](auto I) { stage_in(decltype(I)::value, sbeg + decltype(I)::value); });
    fence();
    for (int s0 = sbeg; s0 < send; s0 += NB) {
        static_for<0, NB / PAIR>([&](auto I) {
            constexpr int i = PAIR * decltype(I)::value;
            if (s0 + i < send) {                                   // uniform
#ifdef KL_COLQ_INTERLEAVE
                if constexpr (PAIR == 1) compute(lds_addr(obj(i)), 32 * (s0 + i), (i + NB - 1) % NB, s0 + i + NB - 1);
                else
#endif
                {
                static_for<0, PAIR>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    stage_in((i + NB - PAIR + u) % NB, s0 + i + NB - PAIR + u);
                });
                compute(lds_addr(obj(i)), 32 * (s0 + i));
                }
                if constexpr (PAIR == 2) {
                    if (s0 + i + 1 < send) compute(lds_addr(obj((i + 1) % NB)), 32 * (s0 + i + 1));
                }
                fence();
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS copy may outlive the workgroup

    if (!active) return;
    if constexpr (Q8 != 0) {
        // suspects: the two lanes of a physical column (h = 0 / 1: different rows) report once; the two waves of a column tile
        // (KSPLIT = 2) saw the same bytes: the first reports
        if (a.q8_list != nullptr && kh == 0) {
            const unsigned both = q8_flag | (unsigned)__shfl_xor((int)q8_flag, 32, 64);
            if (h == 0 && both != 0u) q8_suspect_append(a.st_rw, a.q8_list, chunk, ct, r);
        }
    }
    // acc[m] reg (g,t): component 32m + 8g + 4h + t, feature column ct*32 + r (Q8: the logical column of physical column r)
    const int rcol = Q8 ? 8 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3) : r;
    float *np = a.Npart + (int64_t)chunk * KP * a.f_pad + (int64_t)ct * 32 + rcol;
#pragma unroll
    for (int m = 0; m < KTW; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int comp = 32 * (KTW * kh + m) + 8 * (e >> 2) + 4 * h + (e & 3);
            np[(int64_t)comp * a.f_pad] = acc[m][e];
        }
}

}  // namespace klnmf
