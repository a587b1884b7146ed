// The column pass with fp8 on BOTH sides (KLNMF_COL8=0 switches it off).
//
//   numer[a][j] = sum_i W_new[i][a] * Q[i][j]      with W_new as an e4m3 image (f16 image / 256, saturating) and the fp8 ratio
//                                                  tiles of the row pass, on v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales)
//
// Stage = 64 rows.  Object = [W8 rows: 64 x KP bytes, copied linearly | the 8 waves' two ratio tiles].  Per wave and stage:
// 2 + 2 copy pieces, 4 + 4 KT transposed 8-byte LDS reads, KT MFMAs of K = 64 (64 cycles each: half the matrix time of the
// f16 form), no conversions.  ds_read_b64_tr_b8 gives lane l (component or column l & 31 of the block, half h = l >> 5) the
// 8 rows of its byte column; four of them are the 32 k-values 32 h .. 32 h + 31 of the A / B operand -- the hardware pairs
// byte j of lane half h of A with byte j of lane half h of B, so any k order used for both is right.
// The e4m3 image comes from a conversion kernel behind the row pass (k_w8_from_wb) with a measured power of two per component
// (k_w8_scales: the previous iteration's column maxima); writing it from the W rule itself would save its 0.2 ms.
#pragma once
#include "colq.hip.h"

#define KL_COL8_PF 1      // accumulator blocks of W8 fragments requested ahead of their product (round 4, C4, one box: 1 / 2 / 3 blocks
                          // ahead = 0.845 / 0.857 / 0.870 ms -- the pass waits for HBM, not for LDS: profiles/r04_ab_colpass_prefetch.txt)
#define KL_COL8_NB 4      // LDS objects of the fp8 x fp8 column pass (3 and 5 measured: profiles/r02_ab_fp8_fp8_colpass.txt)

namespace klnmf {

// row stride (bytes) of the e4m3 image: KP, padded so that the 8 rows a transposed 8-byte LDS read touches fall into distinct
// groups of 8 banks (stride in dwords = 8 x odd mod 64) -- the rule of w_ld for the f16 image
__host__ __device__ constexpr int w8_ld(int kp) { return kp + ((kp / 32) % 2 == 0 ? 32 : 0); }

// f16 W image (swizzled rows of w_ld(KP) halves) -> e4m3 image [rows][KP bytes] = image / w8s[component] (saturating), and
// the column maxima of the f16 image for the NEXT iteration's scales (W moves slowly from one update to the next; the
// scale leaves one binade of headroom and the conversion saturates).  Block = 8 rows x (KP / 8) threads: a thread keeps its
// 8 components over all its rows.
// sat != nullptr: entries whose scaled value exceeds e4m3's 448 (they are stored as 448) are counted there -- the image's
// scales come from the PREVIOUS iteration's maxima with one binade of headroom, so a column that more than doubles in one
// update clips; the column passes act on the count (k_colpass_q8x returns, the f16-operand pass runs instead).
// tab64: the block maxima are combined by atomicMax into row (blockIdx & 63) of a
// [64][KP] table -- 16 blocks per address, 64 x KP addresses: no hot line -- which ONE block of
// the launch behind the column pass turns into the next iteration's scales (w8s = those scales: computed by k_post from the
// previous conversion's table into the buffer the host swaps in as w8s).
KL_GLOBAL __launch_bounds__(256) void k_w8_from_wb(const opnd_t *Wb, unsigned char *W8, int64_t rows, int kp, int wld,
                                                    const float *w8s, const DevState *st, int *sat, int probe_col,
                                                    unsigned *tab64, unsigned sr_seed) {
    const int ld8 = w8_ld(kp);
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    if (st->stop) return;
    KL_FP16_SATURATE();
    __shared__ float red[8][256 / 8 * 8];
    const int groups = kp / 8;                       // threads per row
    const int c8 = threadIdx.x % groups, rl = threadIdx.x / groups;
    const int rows_per_block = blockDim.x / groups;
    const int comp = 8 * c8;
    f16x2 inv[4];
    float mx[8];
    int nsat = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) inv[u] = f16x2{(_Float16)(1.f / w8s[comp + 2 * u]), (_Float16)(1.f / w8s[comp + 2 * u + 1])};
#pragma unroll
    for (int e = 0; e < 8; ++e) mx[e] = 0.f;
    if (rl < rows_per_block) {
        const int64_t stride = (int64_t)gridDim.x * rows_per_block;
        constexpr int U = 4;                          // rows in flight per thread (one dependent load per trip was latency-bound; 8 measured in round 6: no gain)
        for (int64_t row0 = (int64_t)blockIdx.x * rows_per_block + rl; row0 < rows; row0 += U * stride) {
            opx8 v[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int64_t row = row0 + q * stride;
                if (row < rows) v[q] = *(const opx8 *)(Wb + row * wld + wb_col((int)(row & 31), comp));
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int64_t row = row0 + q * stride;
                if (row >= rows) break;
                unsigned out[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f16x2 p0 = f16x2{v[q][4 * u], v[q][4 * u + 1]} * inv[2 * u], p1 = f16x2{v[q][4 * u + 2], v[q][4 * u + 3]} * inv[2 * u + 1];
                    // stochastically rounded, as the row pass's own image (mfma4.hip.h, sr_pack4): the seed is a hash of the entry's place
                    unsigned sd = ((unsigned)row * 0x9E3779B1u) ^ ((unsigned)(comp + 4 * u) * 0x85EBCA6Bu) ^ sr_seed;
                    sd = (sd ^ (sd >> 15)) * 0x2C1B3C6Du;
                    const unsigned r1 = (sd ^ (sd >> 12)) * 0x297A2D39u, r2 = (r1 & 0xffffffu) * 0x6C8E95u + 0x3C6EF35Fu;
                    int w = 0;
                    w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, p0[0], r1, 1.f, 0);
                    w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, p0[1], r1 << 7, 1.f, 1);
                    w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, p1[0], r2, 1.f, 2);
                    w = __builtin_amdgcn_cvt_scalef32_sr_fp8_f16(w, p1[1], r2 << 7, 1.f, 3);
                    const f16x2 top = __builtin_elementwise_max(p0, p1);                    // (448 is an f16 number)
                    nsat += (top[0] > (_Float16)448.f || top[1] > (_Float16)448.f) ? ((p0[0] > (_Float16)448.f) + (p0[1] > (_Float16)448.f)
                                                                                     + (p1[0] > (_Float16)448.f) + (p1[1] > (_Float16)448.f)) : 0;
                    out[u] = __builtin_bit_cast(unsigned, w);
                }
                // the probe column (an unused pad component; k_colpass_q8x): e4m3 1.0 in every row -- its accumulator is the sum of the
                // stage's ratio bytes per feature column
                if (probe_col >= comp && probe_col < comp + 8) {
                    const int pc = probe_col - comp;
                    out[pc >> 2] = (out[pc >> 2] & ~(0xffu << (8 * (pc & 3)))) | (0x38u << (8 * (pc & 3)));
                }
                *(uint2 *)(W8 + row * ld8 + comp) = make_uint2(out[0], out[1]);
#pragma unroll
                for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)v[q][e]);
            }
        }
    }
    if (sat != nullptr && nsat != 0) atomicAdd(sat, nsat);
    // block maximum per component (positive floats order like their bit patterns)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[e][threadIdx.x] = mx[e];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float m = red[e][c8];
            for (int q = 1; q < rows_per_block; ++q) m = fmaxf(m, red[e][q * groups + c8]);
            atomicMax(tab64 + (int64_t)(blockIdx.x & 63) * kp + comp + e, __float_as_uint(m));
        }
    }
}

typedef __attribute__((ext_vector_type(2))) int i32x2_t;
template <int OFF>
__device__ __forceinline__ void lds_read_tr8(i32x2_t &dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

struct ColPass8Args {
    ColPassQArgs q;
    const unsigned char *W8;      // [rows][KP] e4m3 = f16 image / w8s[component]
    const float *w8s;             // [KP]
    int probe;                    // != 0: column KP - 1 of the image is the probe (1.0 in every row, k_w8_from_wb)
};



// Ratio entries the fp8 tiles cannot hold or hold too coarsely: the column passes report SUSPECTS -- (row chunk, column tile,
// physical column) whose bytes included one >= 0x60 --, and k_post (post.hip.h) re-reads that column's bytes over the chunk's
// rows (the tiles are still in memory) and corrects its numerator row exactly for every byte >= 0x60 at (row i, column j):
//   q = (x + eps) / (sum_a W_old[i][a] H[a][j] + eps)  from the masters,     held = 8 x e4m3(byte)  (what the tile held)
//   numer[a][j] += W_new image[i][a] * q  -  (what the product added: its W operand x held)            for every component.
// Suspects beyond the list's capacity are counted (q8_unfixed): the loop then gives fp8 up.
__device__ __forceinline__ float e4m3_value(unsigned b) {
    const int ex = (int)((b >> 3) & 15u), man = (int)(b & 7u);
    return ex == 0 ? ldexpf((float)man, -9) : ldexpf(1.f + 0.125f * (float)man, ex - 7);
}
// KSPLIT = 2 (KT > 8, k <= 512): the 8 waves are 4 column tiles x 2 halves of the component range (the accumulators of a
// half fit two waves per SIMD); the two waves of a column tile read the same two ratio tiles and copy one each.
// PROBE: the e4m3 W image carries the probe column (aa.probe, compile-time here: the byte test of the other case would
// otherwise sit in every stage, 3 % of the kernel).
template <int KT, int NB, int KSPLIT = 1, int PROBE = 1>
__global__ __launch_bounds__(kThreads, 1) void k_colpass_q8x(ColPass8Args aa) {
    const ColPassQArgs &a = aa.q;
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    typedef i32x2_t i32x2;
    static_assert((KSPLIT == 1 || KSPLIT == 2) && KT % KSPLIT == 0, "wave decomposition");
    constexpr int CTW = kWavesPerWG / KSPLIT;                 // column tiles per workgroup
    constexpr int KTW = KT / KSPLIT;                          // accumulator blocks per wave
    constexpr int KP = 32 * KT;
    constexpr int LD8 = w8_ld(KP);                            // row stride of the e4m3 image
    constexpr int WST = 64 * LD8;                             // bytes of W8 per 64-row stage
    constexpr int WA = round_up(WST, kGldsRound);             // copied per stage (whole rounds)
    constexpr int WR = WA / kGldsRound;
    constexpr int QA = CTW * 2 * kQTile8;                     // the column tiles' two ratio tiles
    constexpr int OBJ = WA + QA;
    constexpr int QPW = 2 / KSPLIT;                           // ratio tile pieces a wave copies
    constexpr int OPS = WR + QPW;
    static_assert(NB >= 3 && NB <= 5 && NB * OBJ <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char o0[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char o1[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char o2[OBJ];
    __shared__ __attribute__((aligned(16))) unsigned char o3[NB > 3 ? OBJ : 16];
    __shared__ __attribute__((aligned(16))) unsigned char o4[NB > 4 ? OBJ : 16];
    if (a.st->stop) return;
    if (a.guard == 1 && a.st->w8_sat != 0) return;           // this iteration's e4m3 W image clipped: the f16-operand pass behind runs

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    int lin = blockIdx.x;
    if ((G & 7) == 0) lin = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int chunk = lin / a.ncb, cb = lin % a.ncb;
    const int ctl = wave % CTW, kh = wave / CTW;              // column tile inside the workgroup, half of the component range
    const int ct_raw = cb * CTW + ctl;
    const bool active = ct_raw < a.nct;
    const int ct = active ? ct_raw : a.nct - 1;
    // stages of 64 rows (the host's chunk decomposition counts 64-row stages)
    const int sbeg = chunk * a.stages_per_chunk;
    const int send = min(a.nrt / 2, sbeg + a.stages_per_chunk);
    if (sbeg >= send) {
        if (active) {
            float *np = a.Npart + (int64_t)chunk * KP * a.f_pad + (int64_t)ct * 32 + r;
            for (int c = 32 * KTW * kh + h; c < 32 * KTW * (kh + 1); c += 2) np[(int64_t)c * a.f_pad] = 0.f;
        }
        return;
    }
    const int i16 = lane & 15, half = (lane >> 4) & 1;
    // A (W8): rows 32 h + 8 u + (i16 >> 1), byte column 32 m + 16 half + 8 (i16 & 1)
    const unsigned off_a = (32 * h + (i16 >> 1)) * LD8 + 16 * half + 8 * (i16 & 1) + 32 * KTW * kh;
    // B (ratio tiles of stage rows 0..31 for h = 0, 32..63 for h = 1): row 8 u + (i16 >> 1) of tile h
    const unsigned off_b = WA + (2 * ctl + h) * kQTile8 + (i16 >> 1) * 32 + 16 * half + 8 * (i16 & 1);

    f32x16 acc[KTW];
#pragma unroll
    for (int m = 0; m < KTW; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    const unsigned char *const qt_s = a.Qt + (int64_t)ct * a.nrt * kQTile8;
    const unsigned t16 = (unsigned)tid * 16u, l16 = (unsigned)lane * 16u;
    auto obj = [&](int o) -> KL_LDS unsigned char * {
        return (KL_LDS unsigned char *)(o == 0 ? o0 : o == 1 ? o1 : o == 2 ? o2 : o == 3 ? o3 : o4);
    };
    auto lds_addr = [](KL_LDS unsigned char *p) -> unsigned { return (unsigned)(uintptr_t)p; };
    auto stage_in = [&](int o, int sg) {
        sg = min(sg, send - 1);
        const unsigned char *wbase = aa.W8 + (int64_t)sg * WST;
        const unsigned m0w = lds_addr(obj(o)) + (unsigned)wave * 1024u;
#pragma unroll
        for (int rr = 0; rr < WR; ++rr)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0w + rr * kGldsRound), "v"(t16), "s"(wbase + rr * kGldsRound) : "memory");
        const unsigned char *qbase = qt_s + (int64_t)(2 * sg) * kQTile8;
#pragma unroll
        for (int pp = 0; pp < QPW; ++pp) {
            const int p = kh * QPW + pp;               // KSPLIT = 2: the two waves of a column tile copy one tile each
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr(obj(o)) + WA + (2 * ctl + p) * kQTile8), "v"(l16), "s"(qbase + p * kQTile8) : "memory");
        }
    };
    const int rcol_of_r = 8 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3);      // logical column of this lane's physical column
    float probe_prev = 0.f, probe_max = 0.f;
    unsigned q8_flag = 0u;

    auto compute = [&](unsigned base, int stage_row0) {
        (void)stage_row0;
        i32x2 bq[4];
        static_for<0, 4>([&](auto U) {
            constexpr int u = decltype(U)::value;
            lds_read_tr8<u * 8 * 32>(bq[u], base + off_b);
        });
        // W8 fragments run PF accumulator blocks ahead of their product (KL_COL8_PF; ring of PF + 1 register sets of 8)
        constexpr int PF = KL_COL8_PF < KTW ? KL_COL8_PF : (KTW > 1 ? KTW - 1 : 1);
        constexpr int RG = PF + 1;
        i32x2 ring[RG][4];
        // accumulator blocks in DESCENDING order: the last block (it holds the probe component KP - 1) is multiplied first, so that
        // its result is there, without a wait, when the stage's other products have been issued
        auto fetch = [&](auto M) {
            constexpr int m = decltype(M)::value;
            if constexpr (m < KTW) {
                static_for<0, 4>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    lds_read_tr8<u * 8 * LD8 + 32 * (KTW - 1 - m)>(ring[m % RG][u], base + off_a);
                });
            }
        };
        static_for<0, PF>([&](auto M) { fetch(M); });
        constexpr int young0 = 4 * (PF < KTW ? PF : KTW);              // reads younger than the ratio operand's
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) : "n"(young0));
        const i32x8 bo = {bq[0][0], bq[0][1], bq[1][0], bq[1][1], bq[2][0], bq[2][1], bq[3][0], bq[3][1]};
        static_for<0, KTW>([&](auto M) {
            constexpr int m = decltype(M)::value;
            fetch(std::integral_constant<int, m + PF>{});
            constexpr int young = 4 * ((m + PF < KTW ? m + PF : KTW - 1) - m);      // reads of the blocks behind block m
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ring[m % RG][0]), "+v"(ring[m % RG][1]), "+v"(ring[m % RG][2]), "+v"(ring[m % RG][3]) : "n"(young));
            const i32x8 ao = {ring[m % RG][0][0], ring[m % RG][0][1], ring[m % RG][1][0], ring[m % RG][1][1],
                              ring[m % RG][2][0], ring[m % RG][2][1], ring[m % RG][3][0], ring[m % RG][3][1]};
            acc[KTW - 1 - m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ao, bo, acc[KTW - 1 - m], 0, 0, 0, 127, 0, 127);
            // the source order IS the schedule (read, counted wait, product, ...): without the fence hipcc moved the probe block's
            // product behind the others as soon as its accumulator was read below, hoisted every read in front of the products
            // (189 registers) and the read of the probe register then waited a whole product out -- 12 % of the kernel
            __builtin_amdgcn_sched_barrier(0);
        });
        {
            // The probe: component KP - 1 of the e4m3 W image is 1.0 in every row, so register 15 of the last accumulator block in
            // the lanes h = 1 (component 32 (KT - 1) + 31, feature column r) grew by the sum of this stage's 64 ratio bytes of that
            // column: a byte >= 32 (ratio >= 256) cannot hide in a growth below 32 (64 ordinary ratios summing to 256 trip it too:
            // harmless, the fix-up looks at the bytes).  Branch-free and sticky: any control flow here splits the stage into
            // blocks, and hipcc then sinks the products out of their place between the reads (12 % of the kernel).
            if constexpr (PROBE != 0) {
                const float now = acc[KTW - 1][15];
                probe_max = fmaxf(probe_max, now - probe_prev);
                probe_prev = now;
            } else {
                // without a spare component for the probe (k a multiple of 32; the image written by the W rule) the bytes themselves
#pragma unroll
                for (int e = 0; e < 8; ++e) q8_flag |= q8_sat_mask((unsigned)bo[e]);
            }
        }
    };
    auto fence = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NB - 2) * OPS) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, NB - 1>([&](auto I) { stage_in(decltype(I)::value, sbeg + decltype(I)::value); });
    fence();
    for (int s0 = sbeg; s0 < send; s0 += NB) {
        static_for<0, NB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (s0 + i < send) {
                stage_in((i + NB - 1) % NB, s0 + i + NB - 1);
                compute(lds_addr(obj(i)), 64 * (s0 + i));
                fence();
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!active) return;
    if (a.q8_list != nullptr && kh == KSPLIT - 1) {           // suspects of this (row chunk, column tile): one report per physical column
        bool mine;
        if (PROBE != 0) mine = h == 1 && probe_max >= 31.5f;
        else mine = h == 0 && (q8_flag | (unsigned)__shfl_xor((int)q8_flag, 32, 64)) != 0u;
        if (mine) q8_suspect_append(a.st_rw, a.q8_list, chunk, ct, r);
    }
    // acc[m] reg (g,t): component 32m + 8g + 4h + t; lane's column: the LOGICAL column of physical column r (colq.hip.h);
    // the operands were W image / w8s[component] and ratio / 8
    const int rcol = 8 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3);
    float *np = a.Npart + (int64_t)chunk * KP * a.f_pad + (int64_t)ct * 32 + rcol;
#pragma unroll
    for (int m = 0; m < KTW; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int comp = 32 * (KTW * kh + m) + 8 * (e >> 2) + 4 * h + (e & 3);
            np[(int64_t)comp * a.f_pad] = acc[m][e] * (kQ8Scale * aa.w8s[comp]);
        }
}

}  // namespace klnmf
