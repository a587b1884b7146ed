// The in-loop monitor of the fp8 ratio tiles: how much noise does their e4m3 rounding put into the H numerator of THIS data?
//
//   H <- H * (W_new^T . Q) / norm        nmf.py:345-351, with Q = ratio of the OLD factors (nmf.py:325-336)
//
// From a loop's third iteration on the row pass may leave Q as e4m3 of ratio x sqrt(2) / 8 (1 byte per element of V instead
// of 2) and the column pass may multiply an e4m3 image of W_new.  Their 3-bit significands enter ONLY the H numerator, a sum
// over all rows.  Both are rounded STOCHASTICALLY since round 5 (mfma4.hip.h, sr_pack4): every stored entry is unbiased, so
// what reaches the numerator is noise of about 0.036 sqrt(2 / rows) per entry, fresh every iteration -- round to nearest left a
// mean error that depended on where a column's ratios (a component's coefficients) sat in e4m3's cells, did not fall with the
// row count, stayed from one iteration to the next and was integrated by the update's slow modes: every class of round 4's fuzz
// failures (few components in a dead zone, exactly fitted columns cycling between two cells, drift beyond 60 iterations) was
// that bias; DESIGN.md section 6 has the before / after table.  What is left to decide per problem is whether the NOISE is small
// enough: it is not when a column has few stored entries (sparse data stored densely: 5 % stored of 70 000 rows give 1.1e-3
// and end 2e-4 off the oracle), and it is measured, not guessed.  On monitored iterations, for ONE column tile (32 columns,
// rotating with the iteration) and a sample of row tiles (up to 256 x 32 rows, strided over the shard, rotating too), this
// kernel recomputes
//
//   N16[a][j] = sum_i Wimg[i][a] * q[i][j]        q = (x + eps) / (W_old . H_old + eps) from the fp32 masters, x sqrt(2)
//   D[a][j]   = sum_i ( Wop[i][a] * held[i][j] - Wimg[i][a] * q[i][j] )
//
// where held = 8 x e4m3(tile byte) is what the fp8 tile holds and Wop what the column pass multiplies (the e4m3 image x its
// scale when the fp8 x fp8 pass ran, else the f16 image Wimg): D is the error the fp8 regime puts into the numerator of
// these entries, N16 what 16-bit tiles would have given.  (Entries >= 256 are left out: the fix-up list corrects those
// exactly.)  The sample is kept as two halves (even / odd sampled row tiles), which k_post (post.hip.h) combines per
// component row a over the tile's valid columns j:
//
//   bias^2   = sum_j D_A D_B / sum_j N_A N_B   (centred on the component's common factor)   the halves' noise is independent: it drops out
//   noise^2  = sum_j (D_A / N_A - D_B / N_B)^2 / cols   = 2 x (relative variance of one half's numerator)
//   stat_a^2 = max(bias^2, 0) + noise^2 / 2 x rows_half / rows_of_the_shard        (noise scaled to the FULL row sum)
//
// i.e. an estimate of the relative error of the full numerator's entries that does not mistake the sample's own noise for
// a defect.  max_a stat_a above kMonThreshold makes the loop give the fp8 regime up (the host polls DevState.mon_trips with
// the saturation counters; on row shards the count rides in the loss exchange, so that every rank decides alike).  The FIRST
// check is a dry run on the loop's first iteration, still on 16-bit tiles: the bytes the row pass would store are formed here
// by the same conversion, and a loop that fails never takes an fp8 tile.
//
// Cost: the kernel reads 256 x 32 rows of W_old (fp32), W_new (f16, e4m3) and one 32-column tile of V, H_old and the ratio
// tiles -- 13 MB at k = 200 -- and writes 128 x [2 halves][2][KP][32] partial sums (15 MB, read back by k_post), on fp8
// iterations 1, 2, 4, 8, 16 and every 32nd after them (api_loop.hip, monitor_due; measured: DESIGN.md section 6).  The row pass -- the headline kernel -- is
// not touched: the 16-bit ratio is recomputed from the masters here instead of being stored by it.
#pragma once
#include "colq8x.hip.h"

namespace klnmf {

constexpr int kW8TabRows = 64;                  // rows of the conversion kernel's maxima table (blockIdx & 63): [kW8TabRows][KP] float bit patterns
constexpr int kMonBlocks = 128;                  // row tile PAIRS sampled (one tile per half-sample); the monitor launch has one block per tile
// The spread std(q) / mean(q) of each monitored column's ratios is measured as well and reported (klnmf_query_f64): round-to-nearest
// tiles needed it above a quarter of an e4m3 cell (exactly fitted columns cycled between two cells); with unbiased entries the
// H rule keeps its feedback inside a cell and the criterion is off (KLNMF_MON_MIN_SPREAD, development, sets one).
constexpr float kMonMinSpread = 0.f;
// A gross-error check on the other operand: the common factor of a component's numerator row (harmless in itself, the row
// normalisation removes it) is 1e-3 .. 2e-3 on every class measured since the e4m3 image of W_new is rounded stochastically
// (round to nearest: 7e-3 .. 1.9e-2 on coefficients gathered inside one e4m3 cell, and 1e-4 of the loss within ten iterations);
// beyond kMonMaxCommon something other than rounding is wrong with the image, and the fp8 regime ends.
constexpr float kMonMaxCommon = 6.0e-3f;
// On stat_a (above).  Calibrated in round 5 (profiles/r05_monitor_calibration.txt): 2e-4 .. 6e-4 on dense data of 40 000 .. 70 000
// rows whatever the components, the columns or the loop's length (final KL within 8e-5 of the oracle's through 150 .. 200
// iterations), 1.1e-3 / 1.5e-3 on 5 % stored entries (2e-4 off).
constexpr float kMonThreshold = 8.0e-4f;

struct MonArgs {
    const DevState *st;
    const unsigned char *Qt;      // fp8 ratio tiles [col tile][row tile][32 rows][32 physical columns]
    const _Float16 *VtA;          // piece-major 32 x 32 tiles (k_tile_V)
    const float *W32_old;         // [n_pad][KP]
    const float *H_old;           // [KP][f_pad]
    const opnd_t *Wb_new;         // [rows][wld] swizzled f16 image of W_new
    const unsigned char *W8;      // e4m3 image of W_new if this iteration's column pass multiplies it, else nullptr
    const float *w8s;             // [KP]
    float *part;                  // [kMonBlocks][2 halves][2 (N16, D)][KP][32]
    float *spread;                // [2 kMonBlocks][3 (count, sum q, sum q^2)][32]: the ratios' spread per column (entries of V > 0)
    int nrt, nct, kp, k, wld, w8ld;   // nrt / nct: row / column tiles of the tiled buffers (layout)
    int nrt_data;                 // row tiles that hold data (the sample's range)
    int64_t f_pad;
    int ct;                       // the monitored column tile
    int ncols;                    // its valid columns (1 .. 32)
    int nsamp;                    // sampled row tiles (<= kMonBlocks * 2, even)
    int rot;                      // rotation of the sample (row tiles)
    const unsigned *w8tab;        // dry run on a problem whose column pass will multiply the e4m3 image of W_new: the maxima table the
                                  // conversion of THIS iteration filled (k_w8_from_wb; k_post derives the next image's scales from it
                                  // later in the iteration) -- the image is formed here with those scales; nullptr: f16 W operand
    int dry;                      // 1: the loop's second iteration, still on 16-bit tiles -- the e4m3 bytes the row pass WOULD store are
                                  // formed here, by the same conversion instruction: the loop is admitted to the fp8 regime by measurement
    float eps;                    // c * 1e-8
};

// row tile of sample s (0 .. nsamp - 1): strided over the shard, rotated
__device__ __forceinline__ int mon_row_tile(const MonArgs &a, int s) {
    return (int)((((int64_t)s * a.nrt_data) / a.nsamp + a.rot) % a.nrt_data);
}

KL_GLOBAL __launch_bounds__(256) void k_q8_monitor(MonArgs a) {
    if (a.st->stop) return;
    KL_FP16_SATURATE();
    constexpr int AC = 128;                                   // components per staged chunk
    __shared__ float Hs[AC][32];                              // H_old[a0 + .][tile columns]
    __shared__ float Ws[32][AC + 4];                          // W_old[row][a0 + .]
    __shared__ __attribute__((aligned(16))) float Qs[32][32]; // q x sqrt(2) of the current row tile (0 where masked)
    __shared__ __attribute__((aligned(16))) float Hd[32][32]; // held
    __shared__ float Qx[32][33];                              // the plain ratio where the entry of V is > 0, else 0
    __shared__ float Sp[3][32];                               // this block's count / sum / sum of squares of the ratios per column
    __shared__ float Wi_s[32][256], Wo_s[32][256];            // phase B: each thread's 32 image entries / operands of the row tile (its own column)
    if (threadIdx.x < 96) Sp[threadIdx.x >> 5][threadIdx.x & 31] = 0.f;
    const int tid = threadIdx.x;
    const int i = tid >> 3, c4 = (tid & 7) * 4;              // phase A: this thread's row and first column of the tile
    const bool used_w8 = a.W8 != nullptr && a.st->w8_sat == 0;      // (a clipped image: the f16-operand pass runs in its place)
    {
        // one sampled row tile per block (round 6; a block used to take both halves of its pair in sequence: 128 blocks of one
        // wave per SIMD, 123 us of vector arithmetic at k = 200 -- the chip has 256 CUs)
        const int s = (int)blockIdx.x;
        float n16[2][32], dd[2][32];                          // phase B accumulators: components tid and tid + 256
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 32; ++j) { n16[g][j] = 0.f; dd[g][j] = 0.f; }
        if (s < a.nsamp) {
            const int rt = mon_row_tile(a, s);
            const int64_t row0 = (int64_t)rt * 32;
            // ---- phase A: the exact ratio of the tile's 32 x 32 entries from the masters
            float d[4] = {0.f, 0.f, 0.f, 0.f};
            for (int a0 = 0; a0 < a.k; a0 += AC) {
                __syncthreads();
                {
                    // the chunk's 2 x 16 loads of this thread are requested together (as rolled loops they were 32 dependent round
                    // trips per chunk: most of what this launch took -- round 6)
                    constexpr int NL = AC * 32 / 256;
                    float hv[NL], wv[NL];
#pragma unroll
                    for (int u = 0; u < NL; ++u) {
                        const int e = tid + 256 * u, aa = e >> 5, jj = e & 31;
                        hv[u] = (a0 + aa < a.k) ? a.H_old[(int64_t)(a0 + aa) * a.f_pad + (int64_t)a.ct * 32 + jj] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < NL; ++u) {
                        const int e = tid + 256 * u, ii = e / AC, aa = e % AC;
                        wv[u] = (a0 + aa < a.k) ? a.W32_old[(row0 + ii) * a.kp + a0 + aa] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < NL; ++u) {
                        const int e = tid + 256 * u;
                        Hs[e >> 5][e & 31] = hv[u];
                        Ws[e / AC][e % AC] = wv[u];
                    }
                }
                __syncthreads();
                const int na = min(AC, a.k - a0);
                for (int aa = 0; aa < na; ++aa) {
                    const float w = Ws[i][aa];
                    const f32x4 h4 = *(const f32x4 *)&Hs[aa][c4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = fmaf(w, h4[t], d[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cc = c4 + t;
                const int laneA = i + 32 * ((cc >> 2) & 1), eA = 4 * (cc >> 3) + (cc & 3);
                const float x = (float)a.VtA[((int64_t)rt * a.nct + a.ct) * 1024 + (eA >> 3) * 512 + laneA * 8 + (eA & 7)];
                const int pcol = 16 * ((cc >> 2) & 1) + 4 * (cc >> 3) + (cc & 3);      // physical column of logical column cc (mfma4.hip.h, Q8)
                const float rinv = 1.f / (d[t] + a.eps);
                const float q = fmaf(x, rinv, a.eps * rinv);
                unsigned byte;
                if (a.dry) {              // what the row pass's epilogue does to its ratio (mfma4.hip.h, sr_cvt4): f16, x sqrt(2), e4m3 of / 8,
                    // stochastically rounded (the seed: a hash of the entry's place and the check's rotation)
                    unsigned sd = ((unsigned)(row0 + i) * 0x9E3779B1u) ^ ((unsigned)(a.ct * 32 + cc) * 0x85EBCA6Bu) ^ ((unsigned)a.rot * 0xC2B2AE35u);
                    sd = (sd ^ (sd >> 15)) * 0x2C1B3C6Du;
                    sd ^= sd >> 12;
                    const _Float16 qm = (_Float16)q * (_Float16)kQ8Mid;
                    byte = (unsigned)__builtin_amdgcn_cvt_scalef32_sr_fp8_f16(0, qm, sd * 0x297A2D39u, kQ8Scale, 0) & 0xffu;
                } else {
                    byte = a.Qt[((int64_t)a.ct * a.nrt + rt) * 1024 + i * 32 + pcol];
                }
                const bool take = cc < a.ncols && byte < 0x60u;
                Qx[i][cc] = (take && x > 0.f) ? q : 0.f;
                Qs[i][cc] = take ? q * kQ8Mid : 0.f;
                Hd[i][cc] = take ? kQ8Scale * e4m3_value(byte) : 0.f;
            }
            __syncthreads();
            if (tid < 32) {               // the ratios of column tid over the tile's rows, fixed order (entries of V > 0 only: a zero entry's
                float cnt = 0.f, s1 = 0.f, s2 = 0.f;      // ratio is 0 in every format)
                for (int ii = 0; ii < 32; ++ii) { const float qq = Qx[ii][tid]; cnt += qq > 0.f ? 1.f : 0.f; s1 += qq; s2 += qq * qq; }
                Sp[0][tid] += cnt; Sp[1][tid] += s1; Sp[2][tid] += s2;
            }
            // ---- phase B: the two numerators of this row tile, thread = component
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int comp = tid + 256 * g;
                if (comp < a.k) {
                    float dry_scale = 0.f;        // dry run: the scale k_post will give this component's e4m3 image (post.hip.h, w8 block)
                    if (a.dry && a.w8tab != nullptr) {
                        unsigned mb = 0u;
#pragma unroll 16
                        for (int rr = 0; rr < kW8TabRows; ++rr) mb = max(mb, a.w8tab[rr * a.kp + comp]);
                        const float mx = __uint_as_float(mb);
                        dry_scale = 1.f;
                        if (mx > 0.f) {
                            int e;
                            (void)frexpf(mx / 224.f, &e);
                            e = e < -14 ? -14 : (e > 15 ? 15 : e);
                            dry_scale = ldexpf(1.f, e);
                        }
                    }
                    // this component's 32 image entries (and e4m3 bytes) of the row tile: all requested before the first use -- one
                    // dependent load per row made this loop 64 round trips per block, most of the launch's 128 us at k = 200 (round 6).
                    // The operands of the rolled loop below wait in this thread's own LDS column (registers indexed by the loop
                    // counter would live in scratch).
                    {
                        opnd_t wimg_r[32];
                        unsigned w8_r[32];
#pragma unroll
                        for (int ii = 0; ii < 32; ++ii) wimg_r[ii] = a.Wb_new[(row0 + ii) * a.wld + wb_col(ii, comp)];
                        if (used_w8) {
#pragma unroll
                            for (int ii = 0; ii < 32; ++ii) w8_r[ii] = a.W8[(row0 + ii) * a.w8ld + comp];
                        }
                        const float w8s_c = used_w8 ? a.w8s[comp] : 0.f;
#pragma unroll
                        for (int ii = 0; ii < 32; ++ii) {
                            const float wimg = (float)wimg_r[ii];
                            Wi_s[ii][tid] = wimg;
                            Wo_s[ii][tid] = used_w8 ? e4m3_value(w8_r[ii]) * w8s_c : wimg;
                        }
                    }
                    for (int ii = 0; ii < 32; ++ii) {
                        const int64_t row = row0 + ii;
                        const float wimg = Wi_s[ii][tid];
                        float wop = Wo_s[ii][tid];
                        if (dry_scale > 0.f) {        // the conversion of k_w8_from_wb on this one value (stochastically rounded)
                            unsigned sd = ((unsigned)row * 0x9E3779B1u) ^ ((unsigned)comp * 0x85EBCA6Bu) ^ ((unsigned)a.rot * 0xC2B2AE35u);
                            sd = (sd ^ (sd >> 15)) * 0x2C1B3C6Du;
                            const _Float16 wm = (_Float16)wimg * (_Float16)(1.f / dry_scale);
                            const unsigned b8 = (unsigned)__builtin_amdgcn_cvt_scalef32_sr_fp8_f16(0, wm, (sd ^ (sd >> 12)) * 0x297A2D39u, 1.f, 0) & 0xffu;
                            wop = e4m3_value(b8) * dry_scale;
                        }
#pragma unroll
                        for (int j4 = 0; j4 < 32; j4 += 4) {
                            const f32x4 q4 = *(const f32x4 *)&Qs[ii][j4], h4 = *(const f32x4 *)&Hd[ii][j4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const float ref = wimg * q4[t];
                                n16[g][j4 + t] += ref;
                                dd[g][j4 + t] += fmaf(wop, h4[t], -ref);
                            }
                        }
                    }
                }
            }
        }
        // this half's partial sums (zeros for a block beyond the sample: k_post sums all blocks)
        float *pb = a.part + (int64_t)blockIdx.x * 2 * (int64_t)a.kp * 32;      // [pair = block >> 1][half = block & 1]
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int comp = tid + 256 * g;
            if (comp < a.kp) {
#pragma unroll
                for (int j4 = 0; j4 < 32; j4 += 4) {
                    *(f32x4 *)(pb + (int64_t)comp * 32 + j4) = f32x4{n16[g][j4], n16[g][j4 + 1], n16[g][j4 + 2], n16[g][j4 + 3]};
                    *(f32x4 *)(pb + ((int64_t)a.kp + comp) * 32 + j4) = f32x4{dd[g][j4], dd[g][j4 + 1], dd[g][j4 + 2], dd[g][j4 + 3]};
                }
            }
        }
        __syncthreads();
    }
    if (tid < 96) a.spread[(int64_t)blockIdx.x * 96 + tid] = Sp[tid >> 5][tid & 31];
}

// What k_post needs to turn the partial sums into the statistic (post.hip.h)
struct MonPost {
    const float *part;            // [kMonBlocks][2 halves][2][KP][32]; nullptr: no check in this launch
    const float *spread;          // [2 kMonBlocks][3][32]
    float min_spread;             // threshold on the smallest relative spread of a column's ratios (kMonMinSpread)
    float max_common;             // threshold on a component row's common factor over the tile (kMonMaxCommon)
    int ncols;                    // valid columns of the monitored tile
    float noise_scale;            // rows of one half of the sample / rows of this shard (<= 0.5)
    float threshold;
};

}  // namespace klnmf
