// Everything of a fit iteration that follows the column pass, in ONE launch (16-bit modes, ping-pong row pass + stored-ratio
// column pass):
//
//   numerator rows = fixed-order sum of the column pass's slabs            (was k_sum_partials_f32)
//   + the exact correction of ratio entries the fp8 tiles hold too coarsely (was k_q8_fixup: its own launch, float atomics)
//   loss = fixed-order fp64 sum of the row pass's partials, stop rule       (was one extra block of the slab-sum launch / k_decide)
//   H <- normalise_rows(H * numerator), fp16 tile images, row sums, scales  (was k_update_pack_H)          nmf.py:214-220, 345-351
//   scales of the NEXT iteration's e4m3 image of W from this one's maxima   (was k_w8_reduce + k_w8_scales, two launches)
//
// Round 4.  At one rank's shard of the headline shape (125 000 x 4096, k = 200) the iteration was nine launches; the seven
// small ones took 67 us of its 641 (scripts/timeline.py).  Merging them as they were bought nothing -- the launch boundaries
// of back-to-back kernels are almost free on this part, and the first version of this kernel (one block per row, four slabs
// in flight per thread) took the sum of its parts, 32 us: what costs is the CHAIN of memory round trips a block walks.  A
// version that spread every row over four blocks and let the last one finalise the row was worse still (74 us): the
// device-scope fences such an election needs write back / invalidate an XCD's L2 on this part.  So: one block per
// component row, no communication between blocks, and every pass issues all its loads before the first use --
//
//   * 1024 threads (rows of 4096 columns and more), one float4 per thread and slab, up to 16 slabs in flight: the 59 MB of
//     slabs are one or two round trips per block instead of sixteen;
//   * sum x old dictionary -> unnormalised new row + fp64 row sum in the same pass; normalisation + fp16 tile image + image
//     row sum in a second pass over the (L2-resident) row: two dependent passes instead of three;
//   * the new dictionary goes to the OTHER master buffer (H_old read, H_new written): the fix-ups can recompute W.H of
//     single entries from ALL rows of the old dictionary while other blocks are already writing their new rows.
//
// Row shards over several GPUs run it twice per iteration around the all-reduce: POST_SUM launches (slabs -> numerator
// [+ fix-ups], one per column part, so that the all-reduce of part p can run while the column pass of part p + 1 computes),
// then one POST_RULE launch (stop rule on the exchanged loss + H rule on the exchanged numerator).
//
// The stop rule inside a multi-block launch: every block needs the decision, only one may record it.  All evaluate
// `prev - err < tol` from the same two numbers -- err (each reduces the same partials in the same order, or reads the
// exchanged value) and prev = DevState.prev2[(it - 1) & 1], which nobody writes in THIS launch (block 0 records err in
// prev2[it & 1]) -- so they agree by construction; `stop` only ever goes 0 -> 1, and a block that reads block 0's 1 early
// returns exactly as it would have decided itself.
#pragma once
#include "monitor.hip.h"

namespace klnmf {

// Round 6: the slab pass left this kernel again.  One block per component row reads its 16 slab rows in one or two round trips,
// but only k blocks (200 on 256 CUs at the headline shape) were pulling the 59 MB: 18 us = 3.3 TB/s at one rank's shard of
// configuration 4, 147 us for the 200 MB of configuration 5's shard -- and in front of it EVERY block reduced all of the row
// pass's loss partials for the stop rule (5 us at the shard, more at 10^6 rows).  k_slab_sum now does both WIDE (256-thread
// blocks over (1024-column group, component row): 800 blocks at the headline shape; up to 16 blocks for the loss partials)
// in a launch of its own in front of k_post -- the boundary between them is cheap, k_slab_sum writes 4 MB where the column
// pass in front of it leaves 59 MB of dirty lines to write back -- and k_post starts from the numerator rows and from the
// loss blocks' <= 16 pairs.  The sums keep their fixed orders (slabs 0, 1, 2, ...; loss slices in order).
constexpr int kPostMaxParts = 4;
constexpr int kPostSlabBatch = 16;              // slabs in flight per thread (one float4 each)
constexpr int kLossRedMax = 16;                 // blocks of k_slab_sum that reduce the row pass's loss partials (one slice each)
constexpr int kLossRedSlice = 2048;             // ... and the fewest partials worth a block of its own

struct PostPart {
    const float *slabs;       // [nslab][KP][ld] of this part (the column pass's Npart for it)
    float *numer;             // [KP][ld]
    int64_t slab_stride;      // KP * ld
    int ld;                   // row stride of slabs / numer (this part's padded columns; a multiple of 128)
    int col0, ncols;          // dictionary columns [col0, col0 + ncols) -- ncols counts VALID columns (< f); col0 a multiple of 128
    int nslab;
    int ct0;                  // first column tile of the part (ratio tiles: Qt + ct0 * nrt * 1024)
};

struct PostArgs {
    PostPart part[kPostMaxParts];
    int nparts;               // parts this launch covers (POST_SUM launches: 1)
    // ---- what the launch does
    int do_sum;               // the launch behind a column pass (and its k_slab_sum): fix-ups on the numerator rows, the monitor's statistic, the counters
    int do_rule;              // H rule
    int do_decide;            // stop rule (every block evaluates, block 0 records)
    int loss_from_parts;      // err from the row pass's partials as k_slab_sum's loss blocks left them in loss_red; else loss_xchg[0] holds it
    int loss_block;           // one extra block: err -> loss_xchg (the value is exchanged next)
    int w8_block;             // one extra block: scales of the next e4m3 W image from the conversion's maxima table
    int last_sum;             // the iteration's last summing launch: its last block also clears w8_sat for the next conversion
                              // (every summing launch's last block empties the suspect list: the next part's column pass refills it)
    int it;                   // iteration index within the loop (parity of DevState.prev2)
    MonPost mon;              // the fp8 monitor's partial sums of this iteration (monitor.hip.h): every component block turns its row
                              // into the statistic; nullptr: no check in this launch
    // ---- loss / stop rule
    const double2 *loss_red; int nloss; double inv_c; double *loss_xchg; int ne;      // loss_red[nloss]: (sum s1, sum s2) per slice of the partials
    int cq_on;                // the partials come from a pass over a ratio-scaled dictionary image (LossArgs.cq_on, mfma.hip.h)
    double tol_abs; double *errors; int64_t cap;
    DevState *st;
    // ---- H rule
    const float *H_old; float *H_new;      // [KP][f_pad] fp32 masters (ping-pong)
    opnd_t *Ht4; double *hsum; float *tcur, *t_hs;
    int64_t f, f_pad; int kp, k, kc; float eps_pad;
    // ---- fix-ups (fp8 ratio tiles; list == nullptr: none)
    const uint2 *list; const unsigned char *Qt; const _Float16 *VtA; const float *W32_old; const opnd_t *Wb_new;
    const unsigned char *W8; const float *w8s; int w8ld, wld, nrt, nct, stages_per_chunk; float eps;
    // ---- e4m3 scales
    unsigned *w8tab; float *w8s_next;
};

// One slice of the row pass's loss partials, fixed order inside the slice (k_slab_sum's loss blocks; 256 threads, eight
// partials in flight per thread)
__device__ __forceinline__ double2 loss_slice_sum(const double2 *part, int64_t e0, int64_t e1, double *red) {
    constexpr int U = 8;
    double au[U], bu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) au[u] = bu[u] = 0.0;
    const int64_t bd = blockDim.x;
    int64_t e = e0 + threadIdx.x;
    for (; e + (U - 1) * bd < e1; e += U * bd) {
        double2 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = part[e + u * bd];
#pragma unroll
        for (int u = 0; u < U; ++u) { au[u] += p[u].x; bu[u] += p[u].y; }
    }
    for (; e < e1; e += bd) {
        const double2 p = part[e];
        au[0] += p.x;
        bu[0] += p.y;
    }
    const double sa = ((au[0] + au[1]) + (au[2] + au[3])) + ((au[4] + au[5]) + (au[6] + au[7]));
    const double sb = ((bu[0] + bu[1]) + (bu[2] + bu[3])) + ((bu[4] + bu[5]) + (bu[6] + bu[7]));
    const double ta = block_sum(sa, red);
    const double tb = block_sum(sb, red);
    return double2{ta, tb};             // (thread 0 holds the sums)
}

// Slabs of the column pass -> numerator rows (fixed order 0, 1, 2, ...), and the row pass's loss partials -> one pair per slice.
// Grid: x = 1024-column groups of the padded dictionary row, y = component rows [0, k) then the loss slices [k, k + nloss).
struct SlabSumArgs {
    PostPart part[kPostMaxParts];
    int nparts;
    int k;
    int64_t f_pad;
    int nloss;
    const double2 *loss_part; int64_t loss_count; double2 *loss_red;
    const DevState *st;
};

KL_GLOBAL __launch_bounds__(256) void k_slab_sum(SlabSumArgs a) {
    if (a.st->stop) return;
    const int tid = threadIdx.x;
    const int comp = blockIdx.y;
    if (comp >= a.k) {
        if (blockIdx.x != 0) return;
        __shared__ double red[16];
        const int l = comp - a.k;
        const int64_t per = (a.loss_count + a.nloss - 1) / a.nloss;
        const int64_t e0 = (int64_t)l * per, e1 = min(a.loss_count, e0 + per);
        const double2 t = loss_slice_sum(a.loss_part, e0, e1, red);
        if (tid == 0) a.loss_red[l] = t;
        return;
    }
    const int64_t j4 = 4 * ((int64_t)blockIdx.x * blockDim.x + tid);      // this thread's four columns
    if (j4 >= a.f_pad) return;
    int p = 0;                                                            // the part that holds them (parts are multiples of 128 columns)
    while (p + 1 < a.nparts && j4 >= a.part[p + 1].col0) ++p;
    const PostPart &pp = a.part[p];
    const int64_t jl = j4 - pp.col0;
    if (jl < 0 || jl >= pp.ld) return;
    const f32x4 *sl = (const f32x4 *)(pp.slabs + (int64_t)comp * pp.ld + jl);
    const int64_t st4 = pp.slab_stride / 4;
    f32x4 nj = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < pp.nslab; z0 += kPostSlabBatch) {               // all loads of a batch are issued before the first addition;
        f32x4 v[kPostSlabBatch];                                          // the additions keep the order 0, 1, 2, ...
#pragma unroll
        for (int u = 0; u < kPostSlabBatch; ++u)
            if (z0 + u < pp.nslab) v[u] = __builtin_nontemporal_load(sl + (z0 + u) * st4);
#pragma unroll
        for (int u = 0; u < kPostSlabBatch; ++u)
            if (z0 + u < pp.nslab) { if (z0 + u == 0) nj = v[u]; else nj += v[u]; }
    }
    *(f32x4 *)(pp.numer + (int64_t)comp * pp.ld + jl) = nj;
}

// err of the current iteration from the loss slices' pairs: <= 16 uniform loads, the same bits in every thread of every block
__device__ __forceinline__ double post_loss(const PostArgs &a) {
    double ta = 0.0, tb = 0.0;
    for (int l = 0; l < a.nloss; ++l) {
        const double2 p = a.loss_red[l];
        ta += p.x;
        tb += p.y;
    }
    const double ta_q = a.cq_on ? ta + (double)a.st->cq_e * a.st->sum_x : ta;
    return (kLn2 * ta_q + (a.ne ? a.st->corr_eps : 0.0) + tb - a.st->sum_x - a.st->corr_c) * a.inv_c;
}

KL_GLOBAL __launch_bounds__(1024) void k_post(PostArgs a) {
    if (a.st->stop) return;
    KL_FP16_SATURATE();
    __shared__ double red[16];
    __shared__ double bc_s;
    const int tid = threadIdx.x;
    const int nb_main = a.k;
    int b = blockIdx.x;
    // ---- extra blocks --------------------------------------------------------------------------------------------------
    if (b >= nb_main) {
        b -= nb_main;
        if (a.w8_block && b == 0) {
            // scales of the e4m3 image from the maxima the conversion of THIS iteration measured (the next conversion uses them:
            // W moves slowly from one update to the next; one binade of headroom, clipped entries are counted): a power of two
            // with image / scale <= 224 (e4m3 reaches 448).  The table is left empty for the next conversion.
            for (int c = tid; c < a.kp; c += blockDim.x) {
                unsigned mb = 0u;
                const unsigned *tab = a.w8tab + c;
#pragma unroll
                for (int r = 0; r < kW8TabRows; ++r) mb = max(mb, tab[r * a.kp]);      // 64 independent loads, then the stores (interleaved,
                for (int r = 0; r < kW8TabRows; ++r) a.w8tab[r * a.kp + c] = 0u;       // they were a chain of 64 round trips: 30 us for this block alone)
                const float m = __uint_as_float(mb);
                float s = 1.f;
                if (m > 0.f) {
                    int e;
                    (void)frexpf(m / 224.f, &e);
                    e = e < -14 ? -14 : (e > 15 ? 15 : e);
                    s = ldexpf(1.f, e);
                }
                a.w8s_next[c] = s;
            }
            return;
        }
        // the loss partials -> loss_xchg (exchanged between the ranks before anybody decides); [1] carries this rank's count of
        // ratio entries beyond the fix-up list, so that the all-reduced sum tells EVERY rank when a loop must give fp8 up
        // ([1] belongs to ONE writer per launch: in the iteration's last summing launch that is the block that finishes last, below --
        // it alone has seen every fix-up and every monitor row)
        const double err = post_loss(a);
        if (tid == 0) {
            a.loss_xchg[0] = err;
            if (!a.last_sum) a.loss_xchg[1] = (double)(a.st->q8_unfixed + a.st->mon_trips);
        }
        return;
    }
    const int comp = b;
    const int n_sus_all = (a.do_sum && a.list != nullptr) ? a.st->q8_list_n : 0;      // uniform over the grid (reset by the LAST block)
    const bool defer_rule = a.do_sum && a.do_rule && n_sus_all != 0;                   // fix-ups first, then the multiplication
    const float *hold = a.H_old + (int64_t)comp * a.f_pad;
    float *hnew = a.do_rule ? a.H_new + (int64_t)comp * a.f_pad : nullptr;

    // ---- stop rule -------------------------------------------------------------------------------------------------------
    if (a.do_decide) {
        const double err = a.loss_from_parts ? post_loss(a) : a.loss_xchg[0];
        const double prev = a.st->prev2[(a.it + 1) & 1];
        const bool stop_now = prev - err < a.tol_abs;
        if (comp == 0 && tid == 0) {
            if (a.loss_from_parts) {
                a.loss_xchg[0] = err;
                if (!a.last_sum) a.loss_xchg[1] = (double)(a.st->q8_unfixed + a.st->mon_trips);      // (else: the last block's, below)
            }
            if (stop_now) {
                a.st->stop = 1;
            } else {
                a.st->prev2[a.it & 1] = err;
                a.st->prev_err = err;
                if (a.st->n_done < a.cap) a.errors[a.st->n_done] = err;
                a.st->n_done += 1;
            }
        }
        if (stop_now) return;                     // (uniform over the grid: every block computed the same two numbers)
    }

    // ---- fp8 monitor (monitor.hip.h): this component row's statistic from the sampled partial sums --------------------------
    if (a.mon.part != nullptr) {
        __shared__ float mred[8][128], msum[128];
        const int q = tid & 127, sub = tid >> 7, nsub = (int)(blockDim.x >> 7);      // q = (half, array, column); blocks dealt over `sub`
        const int half = q >> 6, arr = (q >> 5) & 1, col = q & 31;
        float acc = 0.f;
        // fixed order; 16 loads in flight (one dependent load per trip made this 16 .. 64 round trips: 25 of the 45 us this kernel
        // took on a monitored iteration -- round 6)
        constexpr int MU = 16;
        for (int blk0 = sub; blk0 < kMonBlocks; blk0 += nsub * MU) {
            float v[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int blk = blk0 + u * nsub;
                v[u] = blk < kMonBlocks ? a.mon.part[(((int64_t)blk * 2 + half) * 2 + arr) * (int64_t)a.kp * 32 + (int64_t)comp * 32 + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < MU; ++u) acc += v[u];
        }
        mred[sub][q] = acc;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
            for (int u = 0; u < nsub; ++u) t += mred[u][tid];
            msum[tid] = t;
        }
        __syncthreads();
        if (tid == 0) {
            // msum: [0..31] N16 of half A, [32..63] D of half A, [64..95] N16 of half B, [96..127] D of half B
            float sd = 0.f, sn = 0.f;
            for (int j = 0; j < a.mon.ncols; ++j) { sd += msum[32 + j] + msum[96 + j]; sn += msum[j] + msum[64 + j]; }
            const float mean = sn > 0.f ? sd / sn : 0.f;         // this row's common factor over the tile: the row normalisation removes it
            float cross = 0.f, crossc = 0.f, nn = 0.f, nz = 0.f;
            int cols = 0;
            for (int j = 0; j < a.mon.ncols; ++j) {
                const float na = msum[j], da = msum[32 + j], nb = msum[64 + j], db = msum[96 + j];
                cross += da * db;
                crossc += (da - mean * na) * (db - mean * nb);
                nn += na * nb;
                if (na > 0.f && nb > 0.f) { const float e = da / na - db / nb; nz += e * e; ++cols; }
            }
            float stat = 0.f, raw = 0.f, noise = 0.f;
            if (nn > 0.f && cols > 0) {
                noise = 0.5f * (nz / (float)cols) * a.mon.noise_scale;
                raw = sqrtf(fmaxf(cross / nn, 0.f));
                stat = sqrtf(fmaxf(crossc / nn, 0.f) + noise);
            }
            if (!(stat <= a.mon.threshold) || fabsf(mean) > a.mon.max_common) {      // (a NaN trips too)
                atomicAdd(&a.st->mon_trips, 1);
                __threadfence();                                  // rare: the count must be there when the last block publishes it
            }
            atomicMax(&a.st->mon_stat_bits, __float_as_uint(stat == stat ? stat : 3.0e38f));
            atomicMax(&a.st->mon_dbg[0], __float_as_uint(raw == raw ? raw : 3.0e38f));
            atomicMax(&a.st->mon_dbg[1], __float_as_uint(noise == noise ? sqrtf(noise) : 3.0e38f));
            atomicMax(&a.st->mon_dbg[2], __float_as_uint(fabsf(mean)));
            if (comp == 0) a.st->mon_checks += 1;
        }
        if (comp == 0) {
            // the relative spread of each monitored column's ratios (monitor.hip.h, kMonMinSpread): block 0, thread = column
            // (32 lanes walking the 128 blocks' sums one after the other were 384 dependent loads: 30 of the 45 us this kernel took on
            // a monitored iteration at one rank's shard; every 32-thread group now takes a contiguous run of blocks -- round 6)
            __shared__ float sp_s[32], sp_g[32][3][32];
            const int ngrp = (int)(blockDim.x >> 5), gi = tid >> 5, ci = tid & 31;      // ngrp = 8, 16 or 32
            {
                const int per = 2 * kMonBlocks / ngrp;                                  // (one entry per block of the monitor launch)
                float cnt = 0.f, s1 = 0.f, s2 = 0.f;
                for (int blk = gi * per; blk < (gi + 1) * per; ++blk) {
                    const float *sp = a.mon.spread + (int64_t)blk * 96;
                    cnt += sp[ci]; s1 += sp[32 + ci]; s2 += sp[64 + ci];
                }
                sp_g[gi][0][ci] = cnt; sp_g[gi][1][ci] = s1; sp_g[gi][2][ci] = s2;
            }
            __syncthreads();
            if (tid < 32) {
                float cnt = 0.f, s1 = 0.f, s2 = 0.f;
                for (int g = 0; g < ngrp; ++g) { cnt += sp_g[g][0][tid]; s1 += sp_g[g][1][tid]; s2 += sp_g[g][2][tid]; }
                float rel = 1.f;                                  // (a column without entries > 0 in the sample: nothing to resolve)
                if (tid < a.mon.ncols && cnt >= 64.f && s1 > 0.f) {
                    const float m = s1 / cnt, var = fmaxf(s2 / cnt - m * m, 0.f);
                    rel = sqrtf(var) / m;
                }
                sp_s[tid] = rel;
            }
            __syncthreads();
            if (tid == 0) {
                float mn = 1.f;
                for (int j = 0; j < 32; ++j) mn = fminf(mn, sp_s[j]);
                if (mn < a.mon.min_spread) { atomicAdd(&a.st->mon_trips, 1); __threadfence(); }
                atomicMin(&a.st->mon_spread_bits, __float_as_uint(mn));
            }
        }
    }

    // ---- the row: numerator (k_slab_sum's, or the exchanged one) x old dictionary -> unnormalised new row + row sum --------------
    double s = 0;
    if (a.do_rule && !defer_rule) {
        for (int64_t j4 = 4 * (int64_t)tid; j4 < a.f_pad; j4 += 4 * (int64_t)blockDim.x) {      // this thread's four columns
            int p = 0;                                                        // the part that holds them (parts are multiples of 128 columns)
            while (p + 1 < a.nparts && j4 >= a.part[p + 1].col0) ++p;
            const PostPart &pp = a.part[p];
            const int64_t jl = j4 - pp.col0;
            if (jl < 0 || jl >= pp.ld) continue;
            const f32x4 nj = *(const f32x4 *)(pp.numer + (int64_t)comp * pp.ld + jl);
            const f32x4 ho = *(const f32x4 *)(hold + j4);
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v[t] = (j4 + t < a.f) ? ho[t] * nj[t] : 0.f;
                s += (double)v[t];
            }
            *(f32x4 *)(hnew + j4) = v;
        }
    }
    const bool stop_now = false;
    // ---- exact correction of large / saturated ratio entries of this row (usually none) ---------------------------------
    if (n_sus_all != 0 && !stop_now) {
        __shared__ float fred[16];
        __shared__ int hits[64], nhit;
        const int n = n_sus_all < kQ8ListCap ? n_sus_all : kQ8ListCap;
        const bool used_w8 = a.W8 != nullptr && a.st->w8_sat == 0;       // (a clipped image: the f16-operand pass ran in its place)
        const PostPart &pp = a.part[0];                                   // (fix-ups: single-part launches and whole-matrix ones)
        __syncthreads();                                                  // the row's sums are in memory
        for (int e = 0; e < n; ++e) {
            const int chunk = (int)(a.list[e].x & 0xffffu), pcol = (int)(a.list[e].x >> 16), ctl = (int)a.list[e].y;
            const int ct = pp.ct0 + ctl;
            const int jl = ctl * 32 + (8 * ((pcol >> 2) & 3) + 4 * (pcol >> 4) + (pcol & 3));      // column inside the part
            const int64_t j = (int64_t)pp.col0 + jl;
            const int row_lo = chunk * a.stages_per_chunk * 64, row_hi = min(a.nrt * 32, row_lo + a.stages_per_chunk * 64);
            float fix = 0.f;                                              // thread 0: this row's correction of column j
            for (int base = row_lo; base < row_hi; base += blockDim.x) {
                if (tid == 0) nhit = 0;
                __syncthreads();
                const int row = base + tid;
                if (row < row_hi) {
                    const unsigned byte = a.Qt[((int64_t)ct * a.nrt + (row >> 5)) * 1024 + (row & 31) * 32 + pcol];
                    if (byte >= 0x60u) {
                        if (byte >= 0x7eu && comp == 0) atomicAdd(&a.st->q8_sat_total, 1);
                        const int at = atomicAdd(&nhit, 1);
                        if (at < 64) hits[at] = (row - base) | ((int)byte << 16);
                        else if (comp == 0) atomicAdd(&a.st->q8_unfixed, 1);      // (65 large ratios of one column within one sweep: not a spike)
                    }
                }
                __syncthreads();
                const int nh = min(nhit, 64);
                if (nh > 1 && tid == 0) {                                  // the order the hits were appended in is a race: sort (fixed sums)
                    for (int x = 1; x < nh; ++x) {
                        const int v = hits[x];
                        int y = x - 1;
                        while (y >= 0 && (hits[y] & 0xffff) > (v & 0xffff)) { hits[y + 1] = hits[y]; --y; }
                        hits[y + 1] = v;
                    }
                }
                __syncthreads();
                for (int t = 0; t < nh; ++t) {
                    const int64_t i = base + (hits[t] & 0xffff);
                    const float held = kQ8Scale * e4m3_value((unsigned)hits[t] >> 16);
                    float part = 0.f;
                    for (int c = tid; c < a.k; c += blockDim.x) part += a.W32_old[i * a.kp + c] * a.H_old[(int64_t)c * a.f_pad + j];
                    part = wave_sum(part);
                    if ((tid & 63) == 0) fred[tid >> 6] = part;
                    __syncthreads();
                    if (tid == 0) {
                        float d = 0.f;
                        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) d += fred[w];
                        const int ii = (int)(i & 31), cc = (int)(j & 31);
                        const int laneA = ii + 32 * ((cc >> 2) & 1), eA = 4 * (cc >> 3) + (cc & 3);
                        const float x = (float)a.VtA[((i >> 5) * a.nct + (j >> 5)) * 1024 + (eA >> 3) * 512 + laneA * 8 + (eA & 7)];
                        const float rinv = 1.f / (d + a.eps);
                        const float q_exact = fmaf(x, rinv, a.eps * rinv);
                        const float wimg = (float)a.Wb_new[i * a.wld + wb_col((int)(i & 31), comp)];
                        const float wprod = used_w8 ? e4m3_value(a.W8[i * a.w8ld + comp]) * a.w8s[comp] : wimg;
                        fix += wimg * (q_exact * kQ8Mid) - wprod * held;      // (the tiles' units: mfma.hip.h, kQ8Mid)
                    }
                    __syncthreads();
                }
            }
            if (tid == 0 && fix != 0.f) pp.numer[(int64_t)comp * pp.ld + jl] += fix;      // (suspects of one column: in list order)
        }
        __syncthreads();
    }
    if (a.do_sum) {
        // the block that finishes last empties the suspect list for the next column pass (the next part's, the next iteration's)
        // and -- in the iteration's last summing launch -- clears the count of clipped entries of the e4m3 W image for the next
        // conversion (every block has read both by now)
        if (tid == 0) {
            // (no fence: every block has long consumed what it read of the list and of w8_sat when it arrives here, and the
            // reset is ordered against the next launch by the kernel boundary -- a device-scope fence per block costs an L2
            // write-back on this part)
            if (atomicAdd(&a.st->q8_fix_done, 1) == nb_main - 1) {
                if (n_sus_all > kQ8ListCap) a.st->q8_unfixed += n_sus_all - kQ8ListCap;
                a.st->q8_fix_done = 0;
                a.st->q8_list_n = 0;
                if (a.last_sum) {
                    a.st->w8_sat = 0;
                    // what tells every rank of a sharded loop to give the fp8 regime up travels as the second double of the loss
                    // exchange: published HERE, by the block that finishes last in the iteration's last summing launch -- every
                    // fix-up and every monitor row of the iteration has been counted (round 4 wrote it from the loss block, which
                    // races with them: the all-reduced count could lag an iteration).  The exchange of [1] must therefore be
                    // ordered BEHIND this launch: the native path's grouped all-reduce is (same stream); the torch path exchanges
                    // [0] alone while the column pass computes and [1] behind it, on the iterations that poll (distributed.py)
                    a.loss_xchg[1] = (double)(atomicAdd(&a.st->q8_unfixed, 0) + atomicAdd(&a.st->mon_trips, 0));
                }
            }
        }
    }
    if (!a.do_rule || stop_now) return;
    // ---- H rule (nmf.py:349-350): H * numerator, row-normalised; fp16 tile image, its row sum, the scales ----------------
    if (a.kc >= 0 && comp == 0) {
        const opnd_t ev = (opnd_t)(a.eps_pad / kCarrierW);          // x the carrier column of the W image = eps
        for (int64_t j = tid; j < a.f_pad; j += blockDim.x) a.Ht4[(j / 32) * (int64_t)a.kp * 32 + h4_elem_rt(a.kc, (int)(j % 32))] = ev;
    }
    if (defer_rule) {             // the row had fix-ups pending: multiply now
        const PostPart &pp = a.part[0];
        const float *nr = pp.numer + (int64_t)comp * pp.ld;
        for (int j = tid; j < pp.ncols; j += blockDim.x) {
            const float v = hold[pp.col0 + j] * nr[j];
            hnew[pp.col0 + j] = v;
            s += (double)v;
        }
    }
    const double total = block_sum(s, red);
    if (tid == 0) bc_s = total;
    __syncthreads();
    const float d = (float)(kEpsNorm + bc_s);
    const float t_a = kOpScaleW;               // a row-normalised dictionary row: sum 1, every entry <= 1
    const float sc = 1.f / t_a;
    double hsm = 0;
    for (int64_t j0 = 4 * tid; j0 < a.f; j0 += 4 * blockDim.x) {
        f32x4 hv = *(const f32x4 *)(hnew + j0);                       // (rows are padded to a multiple of 128 columns)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (j0 + t < a.f) {
                hv[t] = hv[t] / d;
                const opnd_t v16 = (opnd_t)(hv[t] * sc);
                a.Ht4[((j0 + t) / 32) * (int64_t)a.kp * 32 + h4_elem_rt(comp, (int)((j0 + t) % 32))] = v16;
                hsm += (double)(float)v16;
            } else {
                hv[t] = 0.f;
            }
        }
        *(f32x4 *)(hnew + j0) = hv;
    }
    const double ths = block_sum(hsm, red);
    if (tid == 0) {
        a.hsum[comp] = ths;
        a.tcur[comp] = t_a;
        if (a.t_hs) a.t_hs[comp] = t_a;
    }
}

}  // namespace klnmf
