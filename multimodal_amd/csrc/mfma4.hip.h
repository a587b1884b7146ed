// Row pass, ping-pong schedule (generation 4): loss + ratio + W rule in one stream over V.
//
// A wave owns 32 sample rows (its bf16 W fragments stay in registers) and walks the 32-column tiles of the
// dictionary, which stream through 4 rotating LDS objects.  Per tile t its work splits into
//   M(t) = MFMA-2 of tile t-1 (G^T += H_tile . Q_tile^T, 2*KT MFMAs) + MFMA-1 of tile t (W.H, KS MFMAs)
//   E(t) = the VALU epilogue of tile t: ratio, loss terms, bf16 Q operands for the next M segment.
// The two wave groups of the workgroup (waves 0-3 = X, 4-7 = Y; wave w and w+4 share a SIMD,
// experiments/micro/simd_map.hip) run half a tile apart, with ONE s_barrier per tile:
//
//   barrier interval      I_0          I_1          I_2
//   X                 M(0) E(0)  |  M(1) E(1)  |  M(2) E(2)  | ...
//   Y                E(-1) M(0)  |  E(0) M(1)  |  E(1) M(2)  | ...        (Y's E(-1) only issues a copy)
//
// so on every SIMD one wave is in its matrix segment while its partner is in its VALU segment.
//   * dictionary tile t+2 is copied by global_load_lds in exactly its size, each thread its slices, issued
//     from the E segments (X: tile t+2 in E(t); Y: tile t+3 in its E(t), i.e. the same interval);
//   * V tiles go straight to registers, one tile ahead (ordinary loads);
//   * `s_waitcnt vmcnt(0)` at the start of every E segment covers both: the slices issued one E segment ago
//     have landed before the barrier that precedes their first read (two intervals later for X);
//   * all LDS operand reads are inline asm with counted lgkmcnt waits (see lds_read_b128 below);
//   * tile images are unpadded and XOR-swizzled (h4_elem): no bank conflicts for either read pattern.
// What was measured on the way (two barriers per tile, V through LDS, deeper prefetch, 4-wave workgroups,
// LDS semaphores instead of the barrier, copies through registers) is in experiments/README.md.
//
// 256 < k <= 512 (KT = 10..16): the accumulators of 32 rows fill half the register file, so a workgroup is 4 waves,
// one per SIMD, and there is no partner wave.  The update pass then runs the FUSED order (seg_F below): the wave
// issues the epilogue of tile t between the MFMA-2 of tile t-1, computes W.H one tile ahead (inline-asm MFMAs with a
// VGPR accumulator), takes ONE vmcnt(0) + barrier per tile between the two halves and issues all its VMEM work in
// the gaps of the MFMA-1 half.  Rule for the asm MFMAs: every VGPR operand comes from LDS / memory or from VALU code
// that a sched_barrier keeps at least two instructions away (hipcc pads only its own MFMAs).
#pragma once
#include <type_traits>

#include "mfma.hip.h"

namespace klnmf {

// Dictionary tile image [component a][32 columns], 64-byte rows, no padding.  A column c goes to position
// p = h_col_perm(c) (4-column blocks of each 16-column group in the order 0,2,1,3: the MFMA-2 fragment of
// a lane half is then 16 contiguous bytes), and the 16-byte chunk p>>3 of row a is stored at chunk
// (p>>3) ^ ((a>>2)&3).  With that XOR both read patterns are bank-conflict free (64 banks x 4 B):
//  * ds_read_b64_tr_b16 (MFMA-1): a 32-lane group reads 4 consecutive rows x 64 B = all 64 banks once
//    (with padded 80-byte rows the 4th row wrapped onto the 1st: 2-way conflict, PMC: a third of all
//    LDS cycles), the XOR is the same for the 4 rows of a group;
//  * ds_read_b128 (MFMA-2): the 16 rows of a lane group differ in (a&3, (a>>2)&3), i.e. in (bank/16, chunk).
constexpr int kRow4 = 32;                        // image row, elements
constexpr int kRow4B = kRow4 * 2;
__host__ __device__ constexpr int h4_elem(int a, int c) {       // element offset of (component a, column c) in a tile image
    return a * kRow4 + ((((h_col_perm(c) >> 3) ^ ((a >> 2) & 3)) << 3) | (h_col_perm(c) & 7));
}
constexpr int kObj4 = 4 * kGldsRound;            // upper bound of one dictionary tile image (KP <= 512 -> 32768 B)
// Workgroup shape of the ping-pong row pass.  8 waves: the two wave groups share every dictionary copy and one
// barrier.  4 waves (one per SIMD, two workgroups per CU): each workgroup copies its own dictionary tiles (twice
// the L2 -> LDS traffic per CU) but only four waves meet at a barrier and the two workgroups of a CU drift
// freely against each other -- the matrix segment of one overlaps the epilogue of the other statistically.
constexpr int kWaves4 = 8;
constexpr int kThreads4 = 64 * kWaves4;
constexpr int kRound4 = kThreads4 * 16;          // bytes one global_load_lds round of the workgroup moves
__host__ __device__ constexpr int h4_tile_bytes(int kp) { return kp * kRow4B; }

// MFMA-2 fragment p of a tile -> (32-component block, k-step).  Paired order (p>>1, p&1): the two k-steps of a block back to
// back on one accumulator; split order (p % KT, p / KT): all blocks with the first k-step, then all with the second, so
// that consecutive MFMAs never share an accumulator.
__host__ __device__ constexpr int m2_block(int p, int kt, bool split) { return split ? p % kt : p >> 1; }
__host__ __device__ constexpr int m2_kstep(int p, int kt, bool split) { return split ? p / kt : p & 1; }

struct RowPass4Args {
    RowPassArgs base;
    const opnd_t *Ht4;        // [nct][KP][kRow4] per-tile dictionary images
};

// prefetch distance of the operand-fragment stream, in fragments (ring = distance + 1 register sets): the FUSED order's ring is
// fixed by its interval at 3; the 8-wave kernels run 4 ahead (round 3, C4, one box: 2 / 3 / 4 / 5 fragments ahead = 3.66 / 3.69 /
// 3.62 / 3.70 ms, profiles/r03_ab_rowpass_experiments.txt; 4 = 256 registers, no scratch)
constexpr int kPrefetchFused = 3, kPrefetch8 = 4;

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// LDS reads the compiler does not see as memory accesses: its wait-count pass would otherwise
// drain ALL outstanding global_load_lds copies (vmcnt(0)) before any read of an object a copy
// was ever issued into.  Completion is awaited with explicit counted lgkmcnt waits that carry
// the destination as an in/out operand, so no use can be scheduled above its wait.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(opx8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr_pair(opx8 &dst, unsigned addr0, unsigned addr1) {
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr0), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr1), "n"(OFF));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    dst = __builtin_bit_cast(opx8, v);
}
// V tile of this lane straight to registers (2 x 16 B), ordinary loads.  Two measured reasons not to hand-issue
// them: (1) an inline-asm load whose destination overlaps the address pair of the load before it misses the
// wait state the compiler inserts for its own loads (tiles fetched from a stale address); (2) counted vmcnt
// waits are only valid among operations of one kind -- global_load_lds copies (L2 hits) overtake older
// VGPR loads (HBM), so "all but the newest N" does not mean the older V load has landed.  The row pass
// therefore waits vmcnt(0) once per E segment and lets the compiler place the waits of these loads.
// p: wave-uniform (scalar) address of the tile, off: this lane's byte offset -- the form the compiler turns into
// `global_load_dwordx4 v, v_off, s[base]` (no 64-bit VALU address arithmetic in the epilogue segment)
// V is read once per iteration: non-temporal loads (round 3: row pass -0.7 %, iteration -0.5 % in two interleaved A/Bs,
// bit-identical: profiles/r03_ab_nontemporal.txt).  The fp32 master of W -- read once (old) and written once (new) per iteration
// by the row pass's tail -- is non-temporal too (iteration -0.3 %, bit-identical).  Tiles are piece-major (k_tile_V): each
// instruction = 1 KiB of contiguous memory.
__device__ __forceinline__ void v_tile_load(f16x8 &a, f16x8 &b, const unsigned char *p, unsigned off) {
    a = __builtin_nontemporal_load((const f16x8 *)(p + off));
    b = __builtin_nontemporal_load((const f16x8 *)(p + off + 1024));
}
// global -> LDS copy of exactly BYTES (multiple of 16): full 8 KiB rounds of all 512 threads + one partial round
template <int BYTES, int NW = kWaves4>
__device__ __forceinline__ void glds_copy_exact(const unsigned char *gsrc, KL_LDS unsigned char *ldst, int tid) {
    constexpr int kRound4 = NW * 1024;       // bytes one round of the NW-wave workgroup moves
    constexpr int FULL = BYTES / kRound4, REM = BYTES % kRound4;
    const unsigned wave_base = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6) * 1024u;      // scalar: M0 without a VALU detour
    const unsigned t16 = (unsigned)tid * 16u;                                                    // zero-extended lane offset: saddr form
#pragma unroll
    for (int r = 0; r < FULL; ++r)
        __builtin_amdgcn_global_load_lds((const KL_GLB void *)(gsrc + r * kRound4 + t16),
                                         (KL_LDS void *)(ldst + r * kRound4 + wave_base), 16, 0, 0);
    if (REM > 0 && tid * 16 < REM)
        __builtin_amdgcn_global_load_lds((const KL_GLB void *)(gsrc + FULL * kRound4 + t16),
                                         (KL_LDS void *)(ldst + FULL * kRound4 + wave_base), 16, 0, 0);
}

// MFMA-1 of the FUSED order with its accumulator pinned to VGPRs.  A kernel that may use more than 256 registers has
// every compiler-generated MFMA accumulate in AGPRs; at KT = 16 the G accumulators alone are all 256 of them, and
// hipcc then shuttles one block between the two files around every tile (32 moves and a 12-cycle hazard stall).
// The epilogue reads W.H with VALU instructions anyway, so this product lives in VGPRs: written as inline asm (the
// compiler does not know these are matrix instructions -- the wait states between the last one and the first VALU
// read of d are provided by hand, see seg_F).
//
// FUSED order: the counted wait and the matrix instruction as ONE asm statement.  Behind a separate wait statement with
// the fragment as an in/out operand hipcc adds an `s_nop 0` before every MFMA (it must assume a VALU write), and a lone
// wave's issue slots are the scarce resource of that schedule.  G accumulates in AGPRs ("a"), W.H in VGPRs ("v").
template <int N>
__device__ __forceinline__ void mfma2_w(f32x16 &acc, const opx8 &a, const opx8 &b) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n\t" KL_MFMA_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma1_first_w(f32x16 &d, const opx8 &a, const opx8 &b) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n\t" KL_MFMA_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void mfma1_acc_w(f32x16 &d, const opx8 &a, const opx8 &b) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n\t" KL_MFMA_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b), "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(opx8 &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}

// NW = waves per workgroup: 8 (two per SIMD, the X / Y groups of the schedule above; KT <= 7) or 4 (one per SIMD with
// the whole 512-register file -- accumulators in AGPRs -- for 8 <= KT <= 16, i.e. k <= 512: every wave then runs the X
// order alone on its SIMD, matrix and epilogue segments in sequence; with 2*KT + KS >= 48 matrix instructions per tile
// the epilogue is the smaller part).
constexpr int kTailEarly = 3;      // component blocks of the old master requested before the loss sums (k_rowpass4's tail)
// Q8: the ratio tiles left for the column pass are fp8 (e4m3, saturating) instead of the 16-bit MFMA operands: 1 KiB per
// 32 x 32 tile, row-major [row i][16 h' + 4 g + t] = column 8 g + 4 h' + t -- each lane's 16 values are 16 contiguous
// bytes at 16 (2 i + h'), one store per lane and tile.  Only the H numerator (a sum over ALL rows) sees these 4-bit
// significands; the W rule and the loss use the ratio as it stands in the registers (colq.hip.h).
template <int KT, int ODD, int MODE, int EP = 0, int NW = kWaves4, int SPLIT = 0, int Q8 = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void k_rowpass4(RowPass4Args aa) {
    constexpr int kWaves4 = NW, kThreads4 = 64 * NW;
    const RowPassArgs &a = aa.base;
    constexpr int KP = 32 * KT;
    constexpr int KS = 2 * KT - ODD;
    constexpr int WLD = w_ld(KP);
    constexpr int TB = 2048;
    constexpr int N1 = (MODE == ROW_INIT) ? 0 : KS;
    constexpr int N2 = (MODE == ROW_LOSS) ? 0 : 2 * KT;
    constexpr int NF = N1 + N2;              // fragments (= MFMAs) per M segment
    constexpr int PFD = NW == 8 ? kPrefetch8 : kPrefetchFused;
    constexpr int D = PFD < NF - 1 ? PFD : NF - 1;      // reads run D fragments ahead of their MFMA
    constexpr int R = D + 1;
    constexpr int DP = D < N2 ? D : N2;      // fragments of the lead that are MFMA-2 reads (issued one segment early)
    // One wave per SIMD (NW = 4) has no partner whose matrix segment could cover its epilogue: the update pass then runs
    // the FUSED order below, in which the wave itself issues the epilogue of tile t between the MFMA-2 of tile t-1.
    constexpr bool FUSED = NW == 4 && MODE == ROW_UPDATE && N2 >= 16;
    constexpr bool FUSED_ORDER = FUSED;      // MFMA-2 fragments of the FUSED order: split k-steps (m2_block)
    constexpr int IMG = KP * kRow4B;          // bytes of one dictionary tile image
    static_assert(IMG <= kObj4 && IMG % 16 == 0, "dictionary tile image size");
    constexpr int OBJ = IMG;
    // bytes of a tile image that are actually copied per tile: component rows >= 16 KS are zero in every image (k and the
    // eps carrier lie below), so they are zero-filled ONCE in the prologue and never copied -- at k = 200 (KT = 7, KS = 13)
    // 13 instead of 14 copy instructions per tile and workgroup
    constexpr int CPY = KS * 1024 < IMG ? KS * 1024 : IMG;
    // one arena: the four dictionary tile objects of the main loop; after it, the waves' 32 x 32 fp32 transposition
    // buffers of the W rule (kTLD dwords per row: conflict-free 16-byte writes of the accumulator layout)
    constexpr int kTLD = 36;
    constexpr int TBUF = (MODE != ROW_LOSS && SPLIT == 0) ? NW * 32 * kTLD * 4 : 0;
    constexpr int ARENA = 4 * OBJ > TBUF ? 4 * OBJ : TBUF;
    static_assert(ARENA + KP * 16 + 64 <= 160 * 1024, "LDS budget of the row pass");
    __shared__ __attribute__((aligned(16))) unsigned char arena[ARENA];
    KL_LDS unsigned char *const h0 = (KL_LDS unsigned char *)arena, *const h1 = h0 + OBJ, *const h2 = h0 + 2 * OBJ, *const h3 = h0 + 3 * OBJ;
    __shared__ __attribute__((aligned(16))) double hsum_lds[KP];     // row sums of H, for the sum(W.H) term of the loss
    __shared__ __attribute__((aligned(16))) float tc_lds[KP], tn_lds[KP];     // per-component image scales (W rule): current, next
    if (a.st->stop) return;
    KL_FP16_SATURATE();

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: scalar
    const int r = lane & 31, h = lane >> 5;
    const int grpY_s = kWaves4 == 8 ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;       // wave-uniform, kept as a scalar INTEGER:
#define grpY (grpY_s != 0)                                                                  // (as a bool hipcc round-trips it through a VGPR per tile)
    // Static priority for the second-dispatched half of the workgroup (the arbitration loser on every segment otherwise:
    // MI355X_MICROARCH.md, two waves per SIMD, item 4), set once, never flipped: row pass 4.33 -> 4.13 ms at C4, same bits
    // (profiles/r02_ab_static_priority.txt; 1, 2 and 3 measure the same; per-segment flips were no gain in round 1).
    if (grpY) __builtin_amdgcn_s_setprio(1);
    const int rpw = a.rpw > 0 ? a.rpw : kWaves4;                      // row tiles per workgroup (scalar)
    const int rt_raw = ((int)blockIdx.x + a.wg0) * rpw + wave;
    const bool active = wave < rpw && rt_raw < a.nrt;
    const int rt = active ? rt_raw : a.nrt - 1;

    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, half = (lane >> 4) & 1;
    // per-lane byte offsets inside a tile image (see h4_elem): MFMA-1 transposed reads (rows 8h+tq and +4 of a
    // 16-component step), MFMA-2 row reads (row r of a 32-component block, k-step hh = 0 / 1)
    const unsigned off_tr0 = 2 * h4_elem(8 * h + tq, 16 * half + 4 * tp);
    const unsigned off_tr1 = 2 * h4_elem(8 * h + tq + 4, 16 * half + 4 * tp);
    const unsigned off_row0 = 2 * h4_elem(r, 4 * h);             // logical columns 4h.. and 8+4h.. = one permuted chunk
    const unsigned off_row1 = 2 * h4_elem(r, 16 + 4 * h);

    opx8 wf[KS > 0 ? KS : 1];
    if (MODE != ROW_INIT) {
        const opnd_t *wrow = a.Wb_old + (int64_t)(rt * 32 + r) * WLD;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = *(const opx8 *)(wrow + wb_col(r, 16 * s + 8 * h));
    }
    f32x16 acc[KT];
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    float s1 = 0.f;
    double s2 = 0.0;
    const float eps = a.eps;                                            // numerator: (x + eps)
    const float eps_d = a.cq_on ? ldexpf(a.eps, a.st->cq_e) : a.eps;     // denominator of a ratio-scaled image (mfma.hip.h, k_ratio_scale)
    const unsigned char *ht = (const unsigned char *)aa.Ht4;
    const unsigned char *vt = (const unsigned char *)a.VtA + (int64_t)rt * a.nct * TB;
    // column tiles [ct0, ct1) of this workgroup: all of them, or one chunk of the column-split update pass
    // SPLIT is a template parameter, not a runtime flag: the whole-row instantiation must stay the instruction stream it
    // was (a runtime branch cost 2 % at the headline shape and spilled at KT = 16)
    static_assert(SPLIT == 0 || (NW == 8 && MODE == ROW_UPDATE), "column-split pass: 8-wave update kernels only");
    static_assert(Q8 == 0 || (MODE == ROW_UPDATE && sizeof(opnd_t) == 2), "fp8 ratio tiles: update kernels");
    static_assert(Q8 >= 0 && Q8 <= 2, "Q8: 0 = 16-bit ratio tiles, 1 = fp8 tiles, 2 = fp8 tiles and the ratio without the numerator's eps (NE)");
    constexpr bool split = SPLIT != 0;
    const int ct0 = split ? (int)blockIdx.y * a.ct_chunk : 0;
    const int ct1 = split ? min(a.nct, ct0 + a.ct_chunk) : a.nct;

    auto Hobj = [&](int o) -> KL_LDS unsigned char * {      // o in 0..3 (static after unrolling)
        return o == 0 ? h0 : (o == 1 ? h1 : (o == 2 ? h2 : h3));
    };
    auto lds_addr = [](KL_LDS unsigned char *p) -> unsigned { return (unsigned)(uintptr_t)p; };
    // copy of dictionary tile `tg` (global index, clamped) into object o: this thread's slices
    auto dma = [&](int o, int tg) {
        tg = min(tg, a.nct - 1);
        // scalar base + this lane's 32-bit offset, LDS destination (M0) from scalars: no VALU address arithmetic in the
        // epilogue segment (hipcc makes a 64-bit per-lane pointer of the builtin's operand: v_mad_i64_i32 + v_readfirstlane)
        constexpr int kRound4 = NW * 1024, FULL = CPY / kRound4, REM = CPY % kRound4;
        const unsigned char *gbase = ht + (int64_t)tg * IMG;
        const unsigned m0b = lds_addr(Hobj(o)) + (unsigned)wave * 1024u, t16 = (unsigned)tid * 16u;
#pragma unroll
        for (int rr = 0; rr < FULL; ++rr)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0b + rr * kRound4), "v"(t16), "s"(gbase + rr * kRound4) : "memory");
        if (REM > 0 && wave * 1024 < REM)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0b + FULL * kRound4), "v"(t16), "s"(gbase + FULL * kRound4) : "memory");
    };
    // one round (this thread's 16 bytes of every NW KiB) of the copy of dictionary tile tg into object o: the FUSED
    // order issues the rounds one per MFMA instead of back to back (a lone wave's VMEM issue is not covered by a partner)
    constexpr int kRounds = IMG / (NW * 1024);
    unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave index as a scalar
    auto dma_round = [&](int o, int tg, auto Rr) {
        constexpr int rr = decltype(Rr)::value;
        tg = min(tg, a.nct - 1);
        // scalar base + 32-bit lane offset, LDS destination (M0) from scalars: 3 instructions per round instead of the
        // 5 (v_or, v_readfirstlane, s_mov, 64-bit VALU add, load) hipcc produces for the builtin with a per-lane pointer
        const unsigned char *gbase = ht + (int64_t)tg * IMG + rr * (NW * 1024);
        unsigned m0v = lds_addr(Hobj(o)) + rr * (NW * 1024) + wave_u * 1024;
        unsigned t16 = tid * 16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(t16), "s"(gbase) : "memory");
    };
    const unsigned vl32 = (unsigned)lane * 16u;              // this lane's 16 bytes of each of the two pieces of a V tile (vt is wave-uniform)
    // this lane's two 16-byte pieces of a ratio tile (see k_colpass_q); tiles of one column tile are consecutive in rt
    // ratio tiles: wave-uniform base (tiles of one column tile are consecutive in rt) + this lane's 16-byte piece(s)
    const bool qon = a.Qt != nullptr && active;                                      // scalar
    unsigned char *const qbase = a.Qt + (int64_t)rt * (Q8 ? 1024 : 2048);
    const unsigned ql32 = Q8 ? (unsigned)(2 * (lane & 31) + (lane >> 5)) * 16u : (unsigned)lane * 16u;
    const int64_t qstride = (int64_t)a.nrt * (Q8 ? 1024 : 2048);
    f16x8 vreg[4];                                           // V tiles of even / odd column tiles (2 x 16 B each)
    // segment boundary: nothing may be scheduled across it (the MFMAs of an M segment must not sink into the
    // following E segment and vice versa -- that is the whole point of the schedule)
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    opx8 bq[2][2];                // FUSED: Q operands of the even / odd tile slot (one consumed while the other is produced)
    float qv[2] = {0.f, 0.f};       // FUSED: the ratio pair being built
    opx8 ring[R];
    f32x16 d;                       // W.H of the tile between its M and E segments
    opx8 b0, b1;                  // Q operands of the tile between its E segment and the next M segment
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = 0.f;

    // fp8 ratio tiles are rounded STOCHASTICALLY (v_cvt_scalef32_sr_fp8_f16: up when the 7 discarded significand bits + the
    // seed's bits [31:25] reach 128 -- probed, experiments/sr_probe): every stored entry is unbiased, E[e4m3(r)] = r.  Round to
    // nearest leaves a MEAN error in a column's numerator that depends on where the peak of the column's ratios sits in e4m3's
    // 8.8 %-wide cells around 1: it does not fall with the row count, stays from one iteration to the next and is integrated by
    // the slow modes of the update (2e-4 .. 5e-4 of the final KL on few-component / low-rank data after
    // 37 iterations; 20 x less with unbiased entries, whose error is noise of 2^-4 / sqrt(rows), fresh every iteration).
    // Random bits: a 24-bit LCG per lane (v_mad_u32_u24; one step serves two entries: bits [31:25] and [24:18]), started from
    // a hash of (workgroup, lane, the launch's seed).
    // (the e4m3 image of W_new of the fp8 x fp8 column pass is rounded the same way: its entries' round-to-nearest errors are
    // a component-wide factor on data whose coefficients cluster -- 7e-3 on round 4's exactly fitted columns.)
    unsigned sr_state;
    {
        unsigned s = ((unsigned)blockIdx.x * (unsigned)blockDim.x + (unsigned)threadIdx.x + (unsigned)blockIdx.y * 0x632BE5ABu) * 0x9E3779B1u ^ a.sr_seed;
        s = (s ^ (s >> 15)) * 0x85EBCA6Bu;
        sr_state = s ^ (s >> 13);
    }
    auto sr_next = [&]() { sr_state = (sr_state & 0xffffffu) * 0x6C8E95u + 0x3C6EF35Fu; return sr_state; };
    auto sr_pack4 = [&](auto m01, auto m23, auto SCALE) {      // two packed f16 pairs -> e4m3 bytes 0 .. 3 of one dword
        constexpr float scale = (float)decltype(SCALE)::value;
        const unsigned r1 = sr_next(), r2 = sr_next();
        int w;                                // all four bytes are written below: no zero-fill instruction for the tied operand
        asm volatile("" : "=v"(w));
        // (hipcc extracts the odd entries with a shift; the instruction's op_sel[0] would read them in place -- probed in
        // experiments/sr_probe and tried as inline assembly: 8 of 48 vector instructions per tile less, row pass -0.5 .. -0.8 %, but
        // an asm statement gets no wait state for gfx940's dst-sel forwarding hazard and the interleaved two-dword form that
        // provides it by construction spills: not adopted)
        w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, m01[0], r1, scale, 0);
        w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, m01[1], r1 << 7, scale, 1);
        w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, m23[0], r2, scale, 2);
        w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, m23[1], r2 << 7, scale, 3);
        return (unsigned)w;
    };
    auto sr_cvt4 = [&](_Float16 x0, _Float16 x1, _Float16 x2, _Float16 x3) {      // four ratios -> one dword of the tile
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
        const f16x2 m01 = f16x2{x0, x1} * f16x2{(_Float16)kQ8Mid, (_Float16)kQ8Mid};
        const f16x2 m23 = f16x2{x2, x3} * f16x2{(_Float16)kQ8Mid, (_Float16)kQ8Mid};
        return sr_pack4(m01, m23, std::integral_constant<int, (int)kQ8Scale>{});
    };
    auto cvt8_of = [&](const opx8 &b0, const opx8 &b1) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const opx8 &src = j < 2 ? b0 : b1;
            const int o = 4 * (j & 1);
            pk[j] = sr_cvt4(src[o], src[o + 1], src[o + 2], src[o + 3]);
        }
        return pk;
    };
    // Fragment P of the M segment of tile slot TS: [0,N2) MFMA-2 of the previous tile (row reads of ITS image),
    // [N2,NF) MFMA-1 of this tile (transposed reads).  ra / ta: per-lane base addresses in the two objects.
    auto issue = [&](auto P, unsigned robj, unsigned tobj) {     // robj / tobj: LDS addresses of the two tile images
        constexpr int p = decltype(P)::value;
        constexpr int pm = m2_block(p, KT, FUSED_ORDER), ph2 = m2_kstep(p, KT, FUSED_ORDER);
        if constexpr (p < N2) {
            lds_read_b128<(32 * pm) * kRow4B>(ring[p % R], robj + (ph2 ? off_row1 : off_row0));
        } else if constexpr (p < NF) {
            constexpr int s = p - N2;
            lds_read_tr_pair<(16 * s) * kRow4B>(ring[p % R], tobj + off_tr0, tobj + off_tr1);
        }
    };
    // M segment of tile slot TS (global tile tg).  Uniform for every tile: before the first tile the "previous"
    // operands are zeros (b0 = b1 = 0 and a zero-filled image object); after the last one MFMA-1 runs on the
    // clamped copy and is discarded.
    auto seg_M = [&](auto TS, int tg, auto TAIL) {
        constexpr int ts = decltype(TS)::value;
        constexpr bool tail = decltype(TAIL)::value;       // the kernel's last M segment: no copies in flight, no barrier
        const unsigned ra = lds_addr(Hobj((ts + 3) % 4));
        const unsigned ta = lds_addr(Hobj(ts % 4));
        static_for<DP, D>([&](auto P) { issue(P, ra, ta); });      // the part of the lead the E segment could not issue
        static_for<0, NF>([&](auto P) {
            constexpr int p = decltype(P)::value;
            issue(std::integral_constant<int, p + D>{}, ra, ta);
            // LGKM operations younger than fragment p's: fragments p+1 .. min(p+D, NF-1)
            constexpr int last = (p + D < NF - 1) ? p + D : NF - 1;
            constexpr int n_b128 = (last < N2 ? last : N2 - 1) - p > 0 ? (last < N2 ? last : N2 - 1) - p : 0;
            constexpr int n_tr = (last - p) - n_b128;
            lds_wait<n_b128 + 2 * n_tr>(ring[p % R]);
            if constexpr (p < N2) {
                acc[p >> 1] = KL_MFMA_BUILTIN(ring[p % R], (p & 1) ? b1 : b0, acc[p >> 1], 0, 0, 0);
            } else {
                if constexpr (p == N2) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) d[e] = 0.f;
                }
                d = KL_MFMA_BUILTIN(ring[p % R], wf[p - N2], d, 0, 0, 0);
            }
        });
        if (grpY && !tail) barrier();
        else __builtin_amdgcn_sched_barrier(0);
    };
    // the ratios of tile tg as packed in b0 / b1, for the column pass (k_colpass_q2): written once, read once by
    // another kernel -> non-temporal
    auto store_q2 = [&](int tg, const opx8 &b0, const opx8 &b1) {
        if (MODE == ROW_UPDATE && qon) {
            unsigned char *qp = qbase + (int64_t)tg * qstride + ql32;
            unsigned char *const qp_s = qbase + (int64_t)tg * qstride;      // the wave-uniform part (saddr form of the store)
            (void)qp; (void)qp_s;
            if constexpr (Q8 != 0) {
                // from the packed halves (the fp32 ratios are gone by the time the tile leaves): 8 conversions, 2 values each
                const u32x4 pk = cvt8_of(b0, b1);
                // (a VALU write of the data registers of a store wider than 8 bytes needs a wait state after its issue:
                // hipcc pads its own stores, nobody pads an asm statement -- cf. DESIGN.md section 8, h4 and h9 vii)
                asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(ql32), "v"(pk), "s"(qp_s) : "memory");
                return;
            }
            // b0 / b1 are the live MFMA-2 operands: nothing writes them before the next E segment packs the next tile
            asm volatile("global_store_dwordx4 %0, %1, %3 nt\n\tglobal_store_dwordx4 %0, %2, %3 offset:1024 nt\n\ts_nop 1"
                         ::"v"(ql32), "v"(b0), "v"(b1), "s"(qp_s) : "memory");
        }
    };
    auto store_q = [&](int tg) { store_q2(tg, b0, b1); };
    auto store_q_half = [&](int tg, const opx8 &b, int off) {
        if (MODE == ROW_UPDATE && qon) {
            unsigned char *qp = qbase + (int64_t)tg * qstride + ql32;
            if constexpr (Q8 != 0) {       // FUSED order: this half of the lane's 16 bytes of the fp8 tile (off = 0 / 1024 -> + 0 / 8)
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2h;
                u32x2h pk;
#pragma unroll
                for (int j = 0; j < 2; ++j) pk[j] = sr_cvt4(b[4 * j], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3]);
                (void)qp;
                unsigned char *const qp8 = qbase + (int64_t)tg * qstride + (off ? 8 : 0);      // wave-uniform
                asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" ::"v"(ql32), "v"(pk), "s"(qp8) : "memory");
                return;
            }
            (void)qp;
            unsigned char *const qp_s = qbase + (int64_t)tg * qstride + off;      // wave-uniform
            asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(ql32), "v"(b), "s"(qp_s) : "memory");
        }
    };
    // E segment of tile slot TS: ratio + loss terms from d and V, Q operands for the next M segment
    auto seg_E = [&](auto TS, int tg) {
        constexpr int ts = decltype(TS)::value;
        // everything this wave has in flight lands here: V of this tile (issued one E segment ago) and its
        // slices of the dictionary copy issued one E segment ago (first read two or more intervals from now)
        f16x8 &va = vreg[2 * (ts & 1)], &vb = vreg[2 * (ts & 1) + 1];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(va), "+v"(vb)::"memory");
        // the segment's memory instructions: V of the next tile, this wave's slices of the dictionary copy, the previous tile's
        // ratios (still in b0 / b1: a whole tile interval before the next wait)
        auto vmem_block = [&]() {
            // (these stay compiler loads: as asm statements with register outputs -- scalar base + lane offset, one VALU
            // instruction less per tile -- they gave wrong results at 70 000 rows: nothing keeps the compiler from touching an
            // asm output before the data lands; and a pointer passed through an asm statement comes back as a FLAT one)
            v_tile_load(vreg[2 * ((ts + 1) & 1)], vreg[2 * ((ts + 1) & 1) + 1], vt + (int64_t)min(tg + 1, a.nct - 1) * TB, vl32);
            if (grpY) dma((ts + 3) % 4, tg + 3);
            else dma((ts + 2) % 4, tg + 2);
            // ... and only then the previous tile's ratios: their stochastic conversion is 48 vector instructions -- in front of
            // the loads (round 5's first form) it held this segment's memory requests back by some 250 cycles: row pass +4.2 %
            // (profiles/r05_ab_e_segment_order.txt)
            if (tg > ct0) store_q(tg - 1);
        };
        vmem_block();
        float q[16];
        constexpr bool NE = Q8 == 2;       // ratio x / (W.H + eps): the numerator's eps dropped (16 multiplications per tile), see below
        // NE: the addend of the ratio's multiply-add is not 0 but 2^-100 (opaque: the v_fma_mix form stays, x is consumed in its
        // fp16 storage form).  For x > 0 it is below half an ulp of x * r whatever W.H is (x * r >= 2^-24 * 2^-40): the same bits as
        // x * r.  For x = 0 the ratio is 2^-100 instead of 0: its logarithm is finite (-100), the loss term 0 * (-100) = 0 exactly,
        // and it packs to a zero f16 / e4m3 operand -- so V keeps TRUE zeros (round 3 stored them as 2^-24 to keep log2(0) out of
        // the loss), and an all-zero row of V gives an exactly zero row of W as in the reference (nmf.py:156, 342).
        float zero_f = 0x1p-100f;
        if constexpr (NE) asm volatile("" : "+v"(zero_f));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = (float)(e < 8 ? va[e & 7] : vb[e & 7]);
            if (MODE == ROW_INIT) {
                q[e] = x;
            } else {
                // (x + eps) * r as x*r + eps*r: x is consumed in its fp16 storage form by v_fma_mix_f32 (here
                // and in the loss term), so no conversion instruction is needed -- the epilogue's VALU time
                // adds to the matrix time of the SIMD, every instruction counts
                const float rinv = __builtin_amdgcn_rcpf(EP ? d[e] : d[e] + eps_d);      // EP: eps came in through MFMA-1
                // NE (Q8 = 2; api_loop.hip, begin_fp8_loop, chooses it per loop from the data's mean): the ratio as x * r.  Against the reference's
                // (x + eps) * r that is a relative eps / x per element -- chosen only where eps / mean(V) <= 1e-5 (loss record within
                // 1.5e-6, factors within 2.5e-5 of their maxima over 50 iterations: DESIGN.md section 8, h33) --, no logarithm sees
                // a zero ratio (the 2^-100 addend above), and the loss gets the exact constant sum x ln(1 + eps/x) back
                // (DevState.corr_eps).  Row pass -1.8 %, iteration -1.4 % at the headline shape.
                if constexpr (NE) q[e] = fmaf(x, rinv, zero_f);
                else q[e] = fmaf(x, rinv, eps * rinv);
                s1 = fmaf(x, __builtin_amdgcn_logf(q[e]), s1);
            }
        }
        b0 = pack8(q);
        b1 = pack8(q + 8);
        asm volatile("" : "+v"(s1));                        // keep the loss terms inside this segment
        if constexpr (N2 > 0) {                             // prime the next M segment (its MFMA-2 reads THIS tile's image)
            const unsigned ra = lds_addr(Hobj(ts % 4));
            static_for<0, DP>([&](auto P) { issue(P, ra, 0u); });
        }
        if (!grpY) barrier();
        else __builtin_amdgcn_sched_barrier(0);
    };


    // ---- FUSED order (NW = 4, update pass).  Interval t of a wave:
    //        [ MFMA-2 of tile t-1  with the epilogue of tile t issued between its MFMAs ]  [ MFMA-1 of tile t+1 ]
    // MFMA-1 runs one tile ahead, so that the epilogue (needs W.H of tile t, gives the Q operands of tile t) has an
    // independent matrix stream to hide under: an MFMA occupies the pipe for 32 cycles and its issue slot for 4, the
    // 2-3 VALU instructions placed behind it (half an element of the tile: rcp, mul, fma | log, fma, cvt) issue in
    // its shadow.  Source order is pinned per MFMA with sched_barrier(0).  Costs 8 registers (two Q operand sets);
    // W.H of tile t+1 is written only after the last epilogue step of tile t has read W.H of tile t (same registers).
    // Images: interval t reads those of tiles t-1 (rows) and t+1 (transposed); the copy of tile t+2 goes out behind
    // the interval's barrier into the object of tile t-2 and is awaited (vmcnt(0) + barrier) at the next one.
    // Epilogue of one tile in 33 steps, software-pipelined by one step so that no result of a transcendental is used in
    // the step that issues it (the TRANS -> VALU wait state would otherwise cost an s_nop per use):
    //   even step 2e  : loss term and bf16 packing of element e-1 (its log comes from step 2e-1), rcp for element e
    //   odd step 2e+1 : ratio of element e (rcp from step 2e), its log
    float rinv = 0.f, lg = 0.f;
    auto e_step = [&](auto Hh, const f16x8 &va, const f16x8 &vb, opx8 &o0, opx8 &o1) {
        constexpr int hh = decltype(Hh)::value, e = hh >> 1;
        if constexpr ((hh & 1) == 0) {
            if constexpr (e >= 1) {
                constexpr int ep = e - 1;
                const float xp = (float)(ep < 8 ? va[ep & 7] : vb[ep & 7]);
                s1 = fmaf(xp, lg, s1);
                asm volatile("" : "+v"(s1));      // the loss term is computed HERE (hipcc otherwise sinks all 16 past the interval)
                if constexpr ((ep & 1) == 1) {
                    opx8 &o = ep < 8 ? o0 : o1;
                    o[(ep & 7) - 1] = (opnd_t)qv[0];
                    o[ep & 7] = (opnd_t)qv[1];
                }
            }
            if constexpr (e < 16) {
                rinv = __builtin_amdgcn_rcpf(EP ? d[e] : d[e] + eps_d);
                asm volatile("" : "+v"(rinv));
            }
        } else {
            const float x = (float)(e < 8 ? va[e & 7] : vb[e & 7]);
            qv[e & 1] = fmaf(x, rinv, eps * rinv);
            lg = __builtin_amdgcn_logf(qv[e & 1]);
            asm volatile("" : "+v"(lg));
        }
    };
    auto seg_F = [&](auto TS, int tg) {
        constexpr int ts = decltype(TS)::value;
        static_assert(!FUSED || (NF % R == 0 && D <= N2 && D <= N1), "fragment ring of the FUSED order");
        f16x8 &va = vreg[2 * (ts & 1)], &vb = vreg[2 * (ts & 1) + 1];                 // V of tile tg (landed at mid tg-1)
        f16x8 &na = vreg[2 * ((ts + 1) & 1)], &nb = vreg[2 * ((ts + 1) & 1) + 1];     // V of tile tg+1 (lands at mid tg)
        opx8 &p0 = bq[(ts + 1) & 1][0], &p1 = bq[(ts + 1) & 1][1];      // tile tg-1: consumed
        opx8 &c0 = bq[ts & 1][0], &c1 = bq[ts & 1][1];                  // tile tg: produced
        const unsigned ra = lds_addr(Hobj((ts + 3) % 4));                 // image of tile tg-1 (row reads)
        const unsigned ta = lds_addr(Hobj((ts + 1) % 4));                 // image of tile tg+1 (transposed reads)
        const unsigned rn = lds_addr(Hobj(ts % 4));                       // image of tile tg: the next interval's row reads
        // ---- first half: MFMA-2 of tile tg-1, the epilogue of tile tg between its MFMAs.  No VMEM instruction here: with
        // the epilogue steps the gaps of this half are full (an LDS-DMA piece costs 60-180 issue cycles beside them).
        // Fragments 0..D-1 were requested by the previous interval's last gaps.
        static_for<0, N2>([&](auto P) {
            constexpr int p = decltype(P)::value;
            if constexpr (p + D < N2) issue(std::integral_constant<int, p + D>{}, ra, 0u);
            constexpr int young = (p + D < N2 - 1 ? p + D : N2 - 1) - p;               // row reads younger than fragment p's
            mfma2_w<young>(acc[m2_block(p, KT, FUSED_ORDER)], ring[p % R], m2_kstep(p, KT, FUSED_ORDER) ? p1 : p0);
            constexpr int h_lo = (p * 32 + N2 - 1) / N2, h_hi = ((p + 1) * 32 + N2 - 1) / N2;
            static_for<h_lo, h_hi>([&](auto Hh) { e_step(Hh, va, vb, c0, c1); });
            __builtin_amdgcn_sched_barrier(0);
        });
        e_step(std::integral_constant<int, 32>{}, va, vb, c0, c1);
        // ---- middle: the ONE wait and barrier of the interval.  Everything this wave requested in the second half of the
        // previous interval (at least a whole first half ago) has landed: V of tile tg+1, its slices of image tg+1; behind
        // the barrier image tg+1 is complete and every wave is done with image tg-1.
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(na), "+v"(nb)::"memory");
        barrier();
        // ---- second half: MFMA-1 of tile tg+1 (W.H one tile ahead).  Its gaps have issue slots to spare: the interval's
        // VMEM work goes here -- V of tile tg+2 into the registers the epilogue just released, the ratio tile of tg, the
        // rounds of the copy of dictionary tile tg+2 (into the object of tile tg-2) -- all awaited at the next middle.
        static_for<0, D>([&](auto S) { issue(std::integral_constant<int, N2 + decltype(S)::value>{}, 0u, ta); });
        static_for<0, N1>([&](auto S) {
            constexpr int sI = decltype(S)::value;
            if constexpr (sI + D < N1) issue(std::integral_constant<int, N2 + sI + D>{}, 0u, ta);
            else issue(std::integral_constant<int, sI + D - N1>{}, rn, 0u);             // the next interval's first row reads
            constexpr int n_tr = (sI + D < N1 - 1 ? sI + D : N1 - 1) - sI;             // younger transposed fragments (2 reads each)
            constexpr int n_rd = D - n_tr;                                             // younger row reads of the next interval
            if constexpr (sI == 0) mfma1_first_w<2 * n_tr + n_rd>(d, ring[(N2 + sI) % R], wf[0]);
            else mfma1_acc_w<2 * n_tr + n_rd>(d, ring[(N2 + sI) % R], wf[sI]);
            if constexpr (sI == 1) v_tile_load(va, vb, vt + (int64_t)min(tg + 2, a.nct - 1) * TB, vl32);
            if constexpr (sI >= 3 && (sI - 3) % 3 == 0 && (sI - 3) / 3 < kRounds)
                dma_round((ts + 2) % 4, tg + 2, std::integral_constant<int, (sI - 3) / 3>{});
            if constexpr (sI == N1 - 4) store_q_half(tg, c0, 0);
            if constexpr (sI == N1 - 2) store_q_half(tg, c1, 1024);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 15" ::: "memory");       // wait states between the last MFMA-1 and the first VALU read of d
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: dictionary tiles 0 and 1 and V tile 0 in flight; the object of "tile -1" zero-filled
    dma(0, ct0);
    dma(1, ct0 + 1);
    v_tile_load(vreg[0], vreg[1], vt + (int64_t)ct0 * TB, vl32);
    if constexpr (FUSED) v_tile_load(vreg[2], vreg[3], vt + (int64_t)min(1, a.nct - 1) * TB, vl32);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int e = tid; e < IMG / 16; e += kThreads4) ((KL_LDS u32x4 *)h3)[e] = z;
        if constexpr (CPY < IMG) {             // the never-copied rows of the other three objects
            for (int e = tid; e < 3 * ((IMG - CPY) / 16); e += kThreads4) {
                const int o = e / ((IMG - CPY) / 16), w = e % ((IMG - CPY) / 16);
                ((KL_LDS u32x4 *)(h0 + o * OBJ + CPY))[w] = z;
            }
        }
        if (MODE != ROW_INIT)
            for (int e = tid; e < KP; e += kThreads4) hsum_lds[e] = a.hsum[e];
        if (MODE != ROW_LOSS)
            for (int e = tid; e < KP; e += kThreads4) { tc_lds[e] = a.tcur[e]; tn_lds[e] = a.tnext[e]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { b0[j] = (opnd_t)0.f; b1[j] = (opnd_t)0.f; }
    }
    if constexpr (FUSED) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(vreg[0]), "+v"(vreg[1]), "+v"(vreg[2]), "+v"(vreg[3])::"memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(vreg[0]), "+v"(vreg[1])::"memory");
    barrier();
    if constexpr (FUSED) {          // W.H of tile 0 (every later tile's is computed one interval ahead)
#pragma unroll
        for (int j = 0; j < 8; ++j) { bq[1][0][j] = (opnd_t)0.f; bq[1][1][j] = (opnd_t)0.f; }
        const unsigned ta = lds_addr(Hobj(0));
        static_for<0, N1>([&](auto S) {
            constexpr int sI = decltype(S)::value;
            lds_read_tr_pair<(16 * sI) * kRow4B>(ring[0], ta + off_tr0, ta + off_tr1);
            if constexpr (sI == 0) mfma1_first_w<0>(d, ring[0], wf[0]);
            else mfma1_acc_w<0>(d, ring[0], wf[sI]);
        });
        asm volatile("s_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (N2 > 0) {         // what the E segment of "tile -1" would have primed
        const unsigned ra = lds_addr(Hobj(3));
        static_for<0, DP>([&](auto P) { issue(P, ra, 0u); });
    }
    if (grpY) dma(2, ct0 + 2);      // Y's "E(-1)": its slices of tile 2 (X issues its own in E(0), same interval)
    // ---- main loop: 4 tiles per body (nct is a multiple of 4)
    if constexpr (FUSED) {
        for (int t4 = 0; t4 < a.nct; t4 += 4)
            static_for<0, 4>([&](auto I) { seg_F(I, t4 + decltype(I)::value); });
        // tail: MFMA-2 of the last tile (slot 3: its row reads were primed by the last interval)
        const unsigned ra = lds_addr(Hobj(3));
        static_for<0, N2>([&](auto P) {
            constexpr int p = decltype(P)::value;
            if constexpr (p + D < N2) issue(std::integral_constant<int, p + D>{}, ra, 0u);
            constexpr int last = (p + D < N2 - 1) ? p + D : N2 - 1;
            mfma2_w<last - p>(acc[m2_block(p, KT, FUSED_ORDER)], ring[p % R], m2_kstep(p, KT, FUSED_ORDER) ? bq[1][1] : bq[1][0]);
        });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // wait states before the W rule reads G
    } else {
    for (int t4 = ct0; t4 < ct1; t4 += 4) {
        static_for<0, 4>([&](auto I) {
            seg_M(I, t4 + decltype(I)::value, std::false_type{});
            seg_E(I, t4 + decltype(I)::value);
        });
    }
    // ---- tail: MFMA-2 of the last tile (no copies are in flight into anything it reads; no barrier needed)
    seg_M(std::integral_constant<int, 0>{}, ct1, std::true_type{});
    store_q(ct1 - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS copy may outlive the workgroup
    if (TBUF > 0) barrier();      // the W rule's transposition buffers overlay the tile objects the other waves' last M segment reads

    if (!active) return;
    if (split) {
        // column-split pass: this chunk's part of Q.H^T and of the loss; the W rule runs in k_wrule_slabs.
        // sum(W.H) does not depend on the columns: chunk 0 contributes it.
        if (blockIdx.y == 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) s2 = fma((double)(float)wf[s][j], hsum_lds[16 * s + 8 * h + j], s2);
        }
        const double s1w = wave_sum((double)s1);
        s2 = wave_sum(s2);
        const int trt = a.nrt - a.rt0, lrt = rt - a.rt0;      // row tiles of this (possibly partial) launch, this wave's among them
        if (lane == 0) a.loss_part[blockIdx.y == 0 ? (int64_t)rt : (int64_t)a.nrt + (int64_t)(blockIdx.y - 1) * trt + lrt] = make_double2(s1w, s2);
        float *gp = a.gpart + ((int64_t)blockIdx.y * trt * 32 + (int64_t)lrt * 32 + r) * KP;
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[m][4 * g + t];
                *(f32x4 *)(gp + 32 * m + 8 * g + 4 * h) = v;
            }
        return;
    }
    // W rule, first loads: the old master's rows for the first MB1 component blocks are requested BEFORE the loss sums
    // (6 800 cycles of fp64 arithmetic and lane reductions that touch no memory), the rest after them -- the register file
    // holds the accumulators (16 KT), these (16 MB1) and the W fragments the loss sums still read (4 KS).
    constexpr int kEarly = kTailEarly - (Q8 != 0 ? 1 : 0);      // (the fp8 kernels' W rule also carries the e4m3 image's maxima)
    constexpr int MB1 = (MODE == ROW_UPDATE && KT <= 8) ? (KT < kEarly ? KT : kEarly) : 0;
    f32x4 wold_e[MB1 > 0 ? MB1 : 1][4];
    if constexpr (MB1 > 0) {
        const int c4e = (lane & 7) * 4, rje = lane >> 3;
#pragma unroll
        for (int mm = 0; mm < MB1; ++mm)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wold_e[mm][j] = __builtin_nontemporal_load((const f32x4 *)(a.W32_old + ((int64_t)rt * 32 + 8 * j + rje) * KP + 32 * mm + c4e));
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE != ROW_INIT) {
        // sum over this wave's rows of (W.H) = sum_c W[row][c] * hsum[c]; hsum from LDS (staged in the prologue:
        // read from global memory here it was KS dependent round trips on the kernel's tail).  In DOUBLE: the rows
        // of a fitted dictionary sum to 1 up to the rounding of its 16-bit image (hsum = 1 + 1e-5), the W operands
        // have 8-11 significant bits, so in an fp32 chain every product's deviation from W falls below the
        // rounding step of the running sum in the SAME way for every row -- sum_a colsum(W)_a (hsum_a - 1) was
        // lost or kept as a whole, +-2e-7 of sum(V) per evaluation, 1e-4 of the loss at 250000 x 12288, k = 500
        // (scripts/loss_terms_check.py; this, not operand rounding, is what tripped tol = 0 in round 1).
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) s2 = fma((double)(float)wf[s][j], hsum_lds[16 * s + 8 * h + j], s2);
        const double s1w = wave_sum((double)s1);
        s2 = wave_sum(s2);
        if (lane == 0) a.loss_part[rt] = make_double2(s1w, s2);
    }
    if (MODE != ROW_LOSS) {
        // W rule.  The accumulators hold Q.H^T as (row r = lane & 31; components 32 m + 8 g + 4 h + t): taken as they stand,
        // every load / store of the masters would touch 32 rows x 32 bytes (64 scattered 16-byte pieces per instruction:
        // 20 000 + 21 000 cycles per workgroup for the old master's loads and the new one's stores, 9 % of the kernel at C4,
        // in-kernel cycle stamps, round 2).  So each 32 x 32 block goes through the wave's own LDS buffer and comes back row-major --
        // lane l: row 8 j + (l >> 3), components 4 (l & 7) .. + 3 -- and every global instruction moves 8 whole 128-byte
        // lines.  All loads of the old master are issued before the first use (the operand fragments, ring and V
        // registers of the main loop are dead here, so KT*16 registers are free): one memory round trip per wave.
        // (For KT > 8 in blocks of 8 component tiles: the accumulators alone fill half the register file.)
        KL_LDS float *tb = (KL_LDS float *)arena + wave * (32 * kTLD);
        const int c4 = (lane & 7) * 4, rj = lane >> 3;
        const int64_t row0 = (int64_t)rt * 32;
        constexpr int MB = KT <= 8 ? KT : 8;
#pragma unroll
        for (int m0 = 0; m0 < KT; m0 += MB) {
            f32x4 wold[MB][4];
#pragma unroll
            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == ROW_UPDATE && m0 + mm < MB1) {
                        wold[mm][j] = wold_e[mm][j];                 // requested before the loss sums
                    } else if (MODE == ROW_UPDATE && m0 + mm < KT) {
                        wold[mm][j] = __builtin_nontemporal_load((const f32x4 *)(a.W32_old + (row0 + 8 * j + rj) * KP + 32 * (m0 + mm) + c4));
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) wold[mm][j][t] = 1.f;
                    }
                }
            if (MODE == ROW_UPDATE) __builtin_amdgcn_sched_barrier(0);       // keep the loads ahead of the stores
#pragma unroll
            for (int mm = 0; mm < MB; ++mm) {
                if (m0 + mm >= KT) continue;
                const int m = m0 + mm;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = acc[m][4 * g + t];
                    *(KL_LDS f32x4 *)(tb + r * kTLD + 8 * g + 4 * h) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own writes, read back by other lanes
                const f32x4 tc = *(const KL_LDS f32x4 *)(tc_lds + 32 * m + c4), tn = *(const KL_LDS f32x4 *)(tn_lds + 32 * m + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rl = 8 * j + rj, comp = 32 * m + c4;
                    const f32x4 gq = *(const KL_LDS f32x4 *)(tb + rl * kTLD + c4);
                    f32x4 w = wold[mm][j];
                    opx4 wb;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        w[t] *= gq[t] * tc[t];                              // the accumulator saw the dictionary image H / t
                        wb[t] = (EP && comp + t == a.kc) ? (opnd_t)kCarrierW : (opnd_t)(w[t] * tn[t]);      // eps carrier
                    }
                    __builtin_nontemporal_store(w, (f32x4 *)(a.W32_new + (row0 + rl) * KP + comp));
                    *(opx4 *)(a.Wb_new + (row0 + rl) * WLD + wb_col(rl, comp)) = wb;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // block m read before block m + 1 overwrites it
            }
            if (MODE == ROW_UPDATE) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#undef grpY

// W rule of the column-split update pass: G = sum over the chunks' slabs, W_new = W_old * G (fp32 master + swizzled bf16
// image with the eps carrier column, exactly as the tail of k_rowpass4 writes them).  One thread per 4 components.
KL_GLOBAL __launch_bounds__(256) void k_wrule_slabs(const float *gpart, int nchunk, int64_t slab, const float *W32_old,
                                                     float *W32_new, opnd_t *Wb_new, int64_t rows, int kp, int wld, int kc,
                                                     const DevState *st, const float *tcur, const float *tnext) {
    if (st->stop) return;
    KL_FP16_SATURATE();
    const int64_t total = rows * (kp / 4);
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / (kp / 4);
        const int comp = 4 * (int)(e % (kp / 4));
        const int64_t off = row * kp + comp;
        f32x4 g = *(const f32x4 *)(gpart + off);
        for (int z = 1; z < nchunk; ++z) g += *(const f32x4 *)(gpart + z * slab + off);
        f32x4 w = *(const f32x4 *)(W32_old + off);
        const f32x4 tc = *(const f32x4 *)(tcur + comp), tn = *(const f32x4 *)(tnext + comp);
        opx4 wb;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            w[t] *= g[t] * tc[t];
            wb[t] = (comp + t == kc) ? (opnd_t)kCarrierW : (opnd_t)(w[t] * tn[t]);
        }
        *(f32x4 *)(W32_new + off) = w;
        *(opx4 *)(Wb_new + row * wld + wb_col((int)(row & 31), comp)) = wb;
    }
}

}  // namespace klnmf
