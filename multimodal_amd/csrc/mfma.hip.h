// Shared definitions of the 16-bit (fp16-operand MFMA) mode of the KL-NMF update for gfx950 (CDNA4): operand element and image
// layouts, the kernels' argument blocks, and the small kernels around the two passes (dictionary / coefficient packing, the
// upload of V into tiles, the loss reduction with the stop rule).
//
// One fit iteration = a row pass over V (mfma4.hip.h: loss + ratio + W rule, nmf.py:297-343) that leaves the ratio tiles,
// a column pass over them for the H numerator (colq.hip.h / colq8x.hip.h, nmf.py:349) and one launch behind it
// (post.hip.h: slab sum, stop rule, H rule, nmf.py:214-220, 345-351).  Neither W.H nor an fp32 Q goes to HBM.
//
// V storage: fp16 of c*V, c a power of two chosen from max(V) (klnmf_set_v_max) so that c*max(V) is in [2^14, 2^15): same
// bytes as bf16 but 3 more significand bits (the loss is evaluated on V as stored, see k_tile_V).  The whole problem then
// runs in scaled units (W' = cW, eps' = c*eps; H and Q are scale-free) and W / the loss are divided by c on the way out --
// exact, c is a power of two.  V is pre-tiled so that each lane's 16 elements of a 32x32 tile are contiguous in exactly the
// MFMA accumulator order of the row pass (sample on the lane): every V access is a coalesced 16-byte-per-lane stream.
//
// MFMA: v_mfma_f32_32x32x16_f16.  Operand maps (guide section 3):
//   A[row = l&31][k = 8*(l>>5)+j], B[k = 8*(l>>5)+j][col = l&31], j = 0..7
//   D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg = 0..15
#pragma once
#include "common.hip.h"

namespace klnmf {

// MFMA operand element of the 16-bit modes: IEEE half (11 significant bits) with power-of-two scaling of the images
// and saturating conversion (MODE.FP16_OVFL), not bf16 (8 bits): same matrix rate (v_mfma_f32_32x32x16_f16), 8x smaller
// operand rounding.  Scaling (exact, powers of two): per component a a factor t_a;  W image = half(W32 * t_a),
// H image = half(H / t_a)  -- the product of the two scales is 1 for every component, so W.H, the ratio, the loss terms
// and every rule are unscaled; only the two places that turn an accumulator back into a master apply a per-component
// factor (the W rule: G * t_a; the H rule: its factor is constant along a dictionary row and cancels in the row
// normalisation).  Which t_a:
//   * from the second update of a loop on,  t_a = hs_a * 2^-13  with hs_a the power of two >= rowsum(H_a) (1 for a
//     row-normalised dictionary): H image entries <= 2^13, and after any W rule sum_a W_ia rowsum(H_a) = rowsum(V_i)
//     (nmf.py:342 with the ratio of the same W), so W32_ia hs_a <= 2 rowsum(c V_i) <= f 2^16: W image <= 8 f;
//     (V, hence W32, in the storage factor's units: max(c V) in [2^14, 2^15) in BOTH 16-bit modes, whatever V is stored as);
//   * a W that did not come out of a W rule (W0 = V.H0^T of an unnormalised H0, klnmf_set_W) obeys no such bound: its
//     images use MEASURED, balanced scales -- t_a = the power of two nearest sqrt(max_j H_aj / max_i W32_ia), so that both
//     images of component a peak at the same magnitude sqrt(max W max H) -- for the one update they live (k_colmax_W).
typedef _Float16 opnd_t;
#define KL_MFMA_BUILTIN __builtin_amdgcn_mfma_f32_32x32x16_f16
#define KL_MFMA_ASM "v_mfma_f32_32x32x16_f16"
typedef __attribute__((ext_vector_type(8))) opnd_t opx8;
typedef __attribute__((ext_vector_type(4))) opnd_t opx4;
constexpr float kOpScaleH = 8192.f;              // 2^13
constexpr float kOpScaleW = 1.f / 8192.f;
constexpr float kCarrierW = 1.f / 1024.f;
// The fp8 ratio tiles hold ratio x kQ8Mid / 8 with kQ8Mid = sqrt(2): ratio 1 sits in the middle of an e4m3 binade, where the quantiser is
// uniform, instead of on the boundary 2^-3, where it steps by 6 % below and 12 % above and biases every accurately fitted column
// (experiments/fp8_tiles_mid_binade_emulation.py).  The conversion instruction uses only the exponent of its scale operand
// (experiments/micro/scale_probe.hip), hence a packed multiply in front of it.  The H numerator then comes out sqrt(2) larger as a
// whole (the row normalisation removes it); the exact fix-ups work in the same units.
constexpr float kQ8Mid = 1.41421356f;
constexpr float kQ8Scale = 8.f;                 // fp8 ratio tiles hold ratio x kQ8Mid / 8: e4m3 then covers ratios 2^-6.5 .. 3584 / sqrt(2) (saturating)
// f32 -> f16 conversions that overflow give the largest finite half instead of infinity (MODE bit 23, FP16_OVFL; true
// infinities stay): a ratio beyond 65504 (x > 0 where W.H ~ 0) or an operand beyond the image range then perturbs one
// update instead of poisoning the factors with inf - inf.  Set once per kernel (the mode is per wave).
#define KL_FP16_SATURATE() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1")
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define KL_LDS __attribute__((address_space(3)))
#define KL_GLB __attribute__((address_space(1)))

constexpr int kWavesPerWG = 8;
constexpr int kThreads = 64 * kWavesPerWG;       // 512
// Dictionary tile images [component][32 columns]: inside every group of 16 columns the
// four 4-column blocks are stored in the order 0,2,1,3, so that the MFMA-2 A fragment of
// lane half h (logical columns 4h..4h+3 and 8+4h..8+4h+3 of the group) is 16 contiguous
// bytes -> one ds_read_b128.  (Two 8-byte reads get fused into ds_read2_b64, which costs
// 2-4x the LDS cycles per byte and saturated the LDS array: profiles/r01_*.)  The MFMA-1
// transposed reads address 4-column blocks individually, so they just follow the permutation.
__host__ __device__ constexpr int h_col_perm(int c) {           // logical column (0..63) -> physical
    return (c & ~15) | ((((c >> 2) & 1) << 1 | ((c >> 3) & 1)) << 2) | (c & 3);
}
// element offset of (component a, column c) inside a 32-column tile image of the ping-pong row pass (mfma4.hip.h)
__host__ __device__ constexpr int h4_elem_rt(int a, int c) {
    return a * 32 + ((((h_col_perm(c) >> 3) ^ ((a >> 2) & 3)) << 3) | (h_col_perm(c) & 7));
}
constexpr int kStageRowTiles = 2;                // 32-row tiles per LDS stage (column pass)
constexpr int kGldsRound = kThreads * 16;        // bytes one global_load_lds round moves (8 KiB)

__host__ __device__ constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }
// bf16 W image [sample row][component]: row stride KP elements, +32 when KP/32 is even, so that the stride
// is 16 or 48 dwords (mod 64 banks): the 4 consecutive rows x 64 B of a transposed read (column pass,
// MFMA-3) then cover all 64 banks once.  The 16 rows of a ds_read_b128 lane group (MFMA-1') would collide
// 4-way on such a stride, so the 16-byte chunk c of row i is stored at chunk c ^ ((i>>2)&3) -- the same
// swizzle as the dictionary tile images of mfma4.hip.h.  (Padded 464-byte rows were conflict-free for the
// row reads only: PMC showed a third of the column pass's LDS cycles as bank conflicts.)
__host__ __device__ constexpr int w_ld(int kp) { return kp + ((kp / 32) % 2 == 0 ? 32 : 0); }
// element offset of component `comp` inside row `row` (row stride not included)
__host__ __device__ constexpr int wb_col(int row, int comp) {
    return ((((comp >> 3) ^ ((row >> 2) & 3)) << 3) | (comp & 7));
}

enum RowMode { ROW_UPDATE = 0, ROW_INIT = 1, ROW_LOSS = 2 };

// ---- small device helpers ---------------------------------------------------
__device__ __forceinline__ opx8 pack8(const float *q) {
    opx8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (opnd_t)q[j];
    return r;
}

// 16 V elements of one lane of a 32 x 32 tile as stored (two 16-byte pieces)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct RowPassArgs {
    const void *VtA;          // [nrt][nct] tiles of 16 values per lane (layout A), stored piece-major: [16-byte piece][64 lanes][16 B]
    const opnd_t *Wb_old;     // [n_pad(+pad)][w_ld(KP)]
    const float *W32_old;     // [n_pad][KP]
    opnd_t *Wb_new;
    float *W32_new;
    double2 *loss_part;       // [nrt] (sum x*log2 q, sum y)
    const double *hsum;       // [KP] row sums of the 16-bit dictionary image (for sum(W.H)), see row_sum_wh
    unsigned char *Qt;            // ratio tiles for k_colpass_q ([nct][nrt][2 KiB], see there), or null (ping-pong pass only)
    const float *tcur;            // [KP] per-component scale t_a of the CURRENT images (W image = W32 * t, H image = H / t; see opnd_t)
    const float *tnext;           // [KP] scale the W rule packs the NEW W image with (the next dictionary image's)
    int kc;                       // eps-carrying pad component (see k_update_pack_H), -1 if none
    const DevState *st;
    int nrt, nct;             // row tiles, column tiles (a multiple of 4)
    float eps;                // c * 1e-8 (scaled units)
    int cq_on;                // the dictionary image carries the ratio scale 2^st->cq_e (k_ratio_scale): the denominator's eps is scaled with it
    // Column-split update pass of the ping-pong kernel (few rows: one workgroup per 256 rows would leave the chip idle):
    // blockIdx.y = column chunk of ct_chunk tiles; the workgroup leaves its part of Q.H^T in gpart[chunk][row][KP] and
    // its loss terms in loss_part[chunk * nrt + rt]; k_wrule_slabs sums the chunks and applies the W rule.  null: whole rows.
    // A launch may cover only the workgroups from wg0 on (hybrid update pass: the full rounds of workgroups run whole
    // rows, the last partial round runs column-split so that it fills the chip: api_loop.hip, fast_rowpass); the split
    // launch then addresses gpart and the chunks' extra loss parts relative to its first row tile rt0 = 8 * wg0:
    // gpart[chunk][rt - rt0 ...], loss_part[nrt + (chunk - 1) * (nrt - rt0) + rt - rt0] for chunk >= 1.
    float *gpart;
    int ct_chunk;
    int wg0, rt0;
    // row tiles per workgroup of the ping-pong pass (0 = its wave count).  One round of workgroups that leaves CUs idle
    // (50 000 rows = 196 workgroups of 8 row tiles on 256 CUs) is spread over more of them with 7, 6, ... row tiles per
    // workgroup, the workgroup's last waves idling: such problems are HBM-bound per CU (api_loop.hip, fast_rowpass)
    int rpw;
    // (The e4m3 image of W_new for the fp8 x fp8 column pass is written by the conversion kernel k_w8_from_wb, colq8x.hip.h.  Until
    // round 6 the W rule could write it itself -- KLNMF_COL8=2, measured three times as no gain: what the conversion launch costs
    // came back as un-overlapped tail of the row pass -- and its registers made every KT = 7 update kernel spill 8 of them: a
    // kernel with scratch pays 4 us more per launch on this part: profiles/r06_boundary_probe.txt, r06_ab_w8_from_w_rule.txt.)
    unsigned sr_seed;         // fp8 ratio tiles: this launch's seed of the stochastic rounding (mfma4.hip.h, sr_cvt4)
};

// ---- column pass on stored ratios ----------------------------------------------------------------
// The H rule needs W_new^T . Q with the ratio Q of the OLD W and H (nmf.py:347-349) -- exactly what the row
// pass has in registers, as packed bf16 MFMA operands, when it applies the W rule.  The ping-pong row pass
// (mfma4.hip.h) stores them (Qt, 2 B per element of V) and this kernel is then a plain streaming product:
// no second W.H, no division, no V -- half the matrix work of k_colpass and none of its VALU work, for the
// same bytes read (Q instead of the second copy of V, which is no longer kept) plus the row pass's writes.
//
// Qt layout: [column tile][row tile][2 KiB]; a tile is the row pass's two packed operands as its lanes hold
// them: bytes [16*lane, +16) = b0 and [1024 + 16*lane, +16) = b1 of lane (row i = lane & 31, h' = lane >> 5),
// element 4g + t of the 16 = column 8g + 4h' + t -- two fully contiguous 1 KiB stores per wave and tile.
// The row pass needs "one row, several columns" per lane (its contraction runs over columns), this kernel
// "one column, several rows" (contraction over rows): the transposition happens here, on the way through
// LDS.  The global -> LDS copy (global_load_lds takes a per-lane source address) regroups the 16-byte pieces
// so that the 32 bytes of slot 2i + h' are adjacent, and ds_read_b64_tr_b16 -- whose unit is the 8-byte
// group of 4 consecutive columns of one row, which the layout keeps together -- delivers the transposed
// fragments; the 32 lanes of a read's first pass touch 256 consecutive bytes (conflict-free).
struct ColPassQArgs {
    const unsigned char *Qt;
    const opnd_t *Wb_new;     // [n_pad(+pad)][w_ld(KP)]
    float *Npart;             // [nchunks][KP][f_pad]
    const DevState *st;
    int nrt, nct, ncb, nchunks, stages_per_chunk;
    int64_t f_pad;
    // fp8 iterations: guard = 1: the fp8 x fp8 pass, returns at once when this iteration's e4m3 W image saturated (st->w8_sat);
    // guard = 2: the f16-operand pass launched behind it, runs ONLY then.  0: no guard.
    int guard;
    DevState *st_rw;          // saturation counters and the fix-up list's fill (fp8 ratio tiles only)
    uint2 *q8_list;           // [kQ8ListCap] (row, feature column) of saturated ratio entries
};
constexpr int kQTile = 2048;                                         // bytes of one 32x32 bf16 ratio tile
// ---- dictionary / coefficient packing ----------------------------------------
// One block per component row.  do_update: H <- H*num then row-normalise
// (nmf.py:349-350); always (re)writes the fp16 tile images Ht4 from the fp32 master.
//
// Pad-component eps (kc >= 0): the ratio needs 1/(W.H + eps) for every element.  Instead of one VALU add
// per element in both passes, the otherwise unused component kc (k <= kc < 16*KS) carries it through the
// matrix product: row kc of the dictionary images holds eps in every column, column kc of the bf16 W
// images holds 1 (k_pack_W, row-pass epilogue), so MFMA-1 delivers W.H + eps.  hsum[kc] stays 0 (the
// loss term sum(W.H) must not contain it); the accumulators of component kc are never read.
KL_GLOBAL __launch_bounds__(1024) void k_update_pack_H(float *H32, const float *num,
                                                       opnd_t *Ht4, double *hsum, float *tcur, float *t_hs,
                                                       const unsigned *wmax, int *op_range, int64_t f,
                                                       int64_t f_pad, int kp, int do_update,
                                                       const DevState *st, int kc, float eps_pad,
                                                       int nslab = 0, int64_t slab = 0, const DevState *stq = nullptr) {
    if (st && st->stop) return;
    KL_FP16_SATURATE();
    // stq != nullptr: the image carries the ratio scale 2^cq_e (k_ratio_scale below) -- entries, eps row and the balance of
    // the measured scales; hsum stays the row sum of the UNSCALED image (the W rule and the loss's sum(W.H) use it)
    const int qe = stq ? stq->cq_e : 0;
    if (kc >= 0 && blockIdx.x == 0) {
        const opnd_t ev = (opnd_t)ldexpf(eps_pad / kCarrierW, qe);          // x the carrier column of the W image = eps (x 2^cq_e)
        for (int64_t j = threadIdx.x; j < f_pad; j += blockDim.x) {
            Ht4[(j / 32) * (int64_t)kp * 32 + h4_elem_rt(kc, (int)(j % 32))] = ev;
        }
    }
    __shared__ double red[16];
    __shared__ double total;
    __shared__ float rmax_s;
    const int a = blockIdx.x;
    float *row = H32 + (int64_t)a * f_pad;
    float t_a = kOpScaleW;             // a row-normalised dictionary row: sum 1, every entry <= 1 (hs = 1)
    if (do_update) {
        const float *nrow = num + (int64_t)a * f_pad;
        double s = 0;
        for (int64_t j = threadIdx.x; j < f; j += blockDim.x) {
            float nj = nrow[j];
            for (int z = 1; z < nslab; ++z) nj += nrow[z * slab + j];      // nslab > 0: num = the column pass's slabs,
            const float v = row[j] * nj;                                    // summed here in k_sum_partials_f32's order
            row[j] = v;                // (num carries the W image's per-component scale: constant along the row, it
            s += (double)v;            //  cancels in the normalisation below)
        }
        const double t = block_sum(s, red);
        if (threadIdx.x == 0) total = t;
        __syncthreads();
        const float d = (float)(kEpsNorm + total);
        for (int64_t j = threadIdx.x; j < f; j += blockDim.x) row[j] = row[j] / d;
        __syncthreads();
        if (t_hs && threadIdx.x == 0) t_hs[a] = t_a;
    } else {
        // a dictionary as given (klnmf_set_H: transform on a column slice, an unnormalised initial dictionary): hs = the
        // power of two at or above the row sum, so that the image's entries use the half range as a normalised row's;
        // wmax: column maxima of the W that goes with it (bit patterns) -> the measured, balanced scale (see opnd_t)
        double s = 0;
        float mx = 0.f;
        for (int64_t j = threadIdx.x; j < f; j += blockDim.x) { s += (double)row[j]; mx = fmaxf(mx, row[j]); }
        const double t = block_sum(s, red);
        __shared__ float mred[16];
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
        if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m2 = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m2 = fmaxf(m2, mred[w]);
            rmax_s = m2;
            total = t;
        }
        __syncthreads();
        int e = 0;
        if (total > 0 && total < 1e300) { (void)frexp(total, &e); t_a = ldexpf(kOpScaleW, e); }
        if (t_hs && threadIdx.x == 0) t_hs[a] = t_a;
        if (wmax) {
            const float wm = __uint_as_float(wmax[a]), rmax = rmax_s;
            if (wm > 0.f && wm < 3e38f && rmax > 0.f) {
                int ew = 0, eh = 0;
                (void)frexpf(wm, &ew);
                (void)frexpf(rmax, &eh);
                eh += qe;                                                // the image holds H x 2^cq_e
                int et = (eh - ew) / 2;                                  // t ~ sqrt(max H / max W)
                if (eh - et > 15) et = eh - 15;                          // H image peak <= 2^15 first (W saturates, if anything)
                if (ew + et > 16 && threadIdx.x == 0) atomicAdd(op_range, 1);   // ... and it does: reported by the loop entry points
                t_a = ldexpf(1.f, et);
            }
        }
        __syncthreads();
    }
    const float sc = ldexpf(1.f / t_a, qe);        // exact: t_a is a power of two
    double hsm = 0;
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) {
        const opnd_t v = (opnd_t)(row[j] * sc);
        Ht4[(j / 32) * (int64_t)kp * 32 + h4_elem_rt(a, (int)(j % 32))] = v;   // mfma4.hip.h tile images (swizzled)
        hsm += (double)(float)v;
    }
    const double ths = block_sum(hsm, red);
    if (threadIdx.x == 0) {
        hsum[a] = ldexp(ths, -qe);     // row sum of the IMAGE (scaled by 1 / t_a only): x the W image's scale it is sum_j (W.H)_ij exactly
        tcur[a] = t_a;
    }
}

// Column maxima of a W master (all entries >= 0: the bit pattern of a non-negative float orders like the integer), for the
// measured image scales k_update_pack_H derives from them.
KL_GLOBAL void k_colmax_W(const float *W32, int64_t n, int kp, unsigned *wmax) {
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= kp) return;
    float m = 0.f;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) m = fmaxf(m, W32[i * kp + c]);
    atomicMax(wmax + c, __float_as_uint(m));
}
// The ratio scale of the first update after klnmf_init_W.  W0 = V.H0^T (nmf.py:156) is not the result of a W rule: the model
// W0.H0 is too small by about f / sum_a rowsum(H0_a) (= f / k for a row-normalised dictionary: every entry of W0 is a weighted
// MEAN of its row of V, and k such means replace a SUM over f columns), so the first ratios are that much larger than 1 --
// beyond 65504, the largest f16, from f / k ~ 1000 on with heavy-tailed data (round 4's shape fuzz: k = 1, f = 2755; the
// saturated operands clip the first H numerator and errors[1] is off by a factor 2, the run recovers two iterations later).
// One block: cq_e = floor(log2(f / sum of the dictionary's entries)), 0 below 2^7 (ratios up to 500 x their mean still fit), or
// what brings the worst case f^2 / sum down to 2^15 if that is more; at most 12.  Derived from the dictionary alone, so that every rank of a row-sharded loop takes the same value.  The image
// packed next carries 2^cq_e (k_update_pack_H); the update pass then sees W.H x 2^cq_e and a ratio / 2^cq_e: its second
// product Q.H^T multiplies the two and is unchanged, the H numerator is scaled as a whole and the row normalisation removes
// it, the loss adds cq_e x sum(x) to its sum of x log2(ratio) (loss_from_parts_block).  enable = 0: writes 0.
KL_GLOBAL __launch_bounds__(256) void k_ratio_scale(const double *hsum, const float *tcur, int k, int64_t f, DevState *st, int enable, int e_cap) {
    __shared__ double red[16];
    // sum of the dictionary's entries from the pack that precedes every klnmf_init_W (klnmf_set_H): hsum holds the row sums of the
    // image H / t (k_update_pack_H), tcur its t -- k values instead of k x f (a one-block walk over a 500 x 12 288 dictionary
    // took 3.5 ms)
    double s = 0;
    if (enable)
        for (int a = threadIdx.x; a < k; a += blockDim.x) s += hsum[a] * (double)tcur[a];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) {
        int e = 0;
        if (enable && t > 0 && t < 1e300) {
            int ex = 0;
            (void)frexp((double)f / t, &ex);          // f / t = m 2^ex, m in [0.5, 1)  ->  floor(log2) = ex - 1
            e = ex - 1;
            if (e < 7) e = 0;                  // (a mean ratio below 128: ratios 500 x their mean still fit)
            // ... and the WORST first ratio: a row of V that is one entry x at column j has W0_a = x H0_aj and a ratio of
            // 1 / sum_a H0_aj^2 there -- f^2 / k for a flat dictionary, 3.4e5 at f = 4096, k = 50 (data fuzz, round 4: one entry
            // 1e4 x the rest made errors[1] 38 % wrong; log-normal data with sigma 2 the same in small).  The scale that puts
            // f^2 / sum(H0) at 2^15 leaves the ordinary ratios (f / k) far inside the half range: f16 spans 30 binades.
            int ex2 = 0;
            (void)frexp((double)f / t * (double)f, &ex2);      // = m 2^ex2, m in [0.5, 1): ceil(log2) <= ex2
            if (ex2 - 15 > e) e = ex2 - 15;
            if (e > e_cap) e = e_cap;          // (12, or what the image's eps row can take)
        }
        st->cq_e = e;
    }
}
KL_GLOBAL void k_pack_W(const float *W32, opnd_t *Wb, int64_t n, int kp, int wld, int kc, const float *tcur) {
    KL_FP16_SATURATE();
    const int64_t total = n * kp;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / kp, c = e % kp;
        Wb[i * wld + wb_col((int)(i & 31), (int)c)] = (c == kc) ? (opnd_t)kCarrierW : (opnd_t)(W32[e] * tcur[c]);
    }
}

// Scatter a host-layout fp32/fp64 [n,k] (or [k,f]) array into a padded fp32 master.
template <typename S>
__global__ void k_place_padded(float *dst, int64_t dld, const S *src, int64_t rows, int64_t cols,
                               double mul) {
    const int64_t total = rows * cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / cols, c = e % cols;
        dst[i * dld + c] = (float)(mul * (double)src[e]);
    }
}
// [rows, cols] block between two device matrices of any row strides and element types (device-resident operands:
// klnmf_set_H_device, klnmf_get_W_device)
template <typename D, typename S>
__global__ void k_copy_2d(D *dst, int64_t dld, const S *src, int64_t sld, int64_t rows, int64_t cols, double mul) {
    const int64_t total = rows * cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / cols, c = e % cols;
        dst[i * dld + c] = (D)(mul * (double)src[i * sld + c]);
    }
}
template <typename D>
__global__ void k_gather_padded(D *dst, const float *src, int64_t sld, int64_t rows, int64_t cols,
                                double mul) {
    const int64_t total = rows * cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / cols, c = e % cols;
        dst[e] = (D)(mul * (double)src[i * sld + c]);
    }
}

// ---- V upload: scale, cast, tile (K6) ------------------------------------------
// Tile element map (32x32 tile, i = row in tile, c = col in tile):
//   lane = i + 32*((c>>2)&1), e = 4*(c>>3) + (c&3)      (sample on lane: the row pass's accumulator order)
// Also accumulates sum(x~) and the storage-rounding correction
//   C = KL(x~ || x) = sum( x~ ln(x~/x) - x~ + x )        (0 ln 0 = 0)
// With x~ the value as stored:  KL(x||y) = KL(x~||y) - C + sum (x~-x) ln(y/x)
// exactly; the kernels evaluate KL(x~||y), the host-visible loss is
// KL(x~||y) - C, and the dropped last term is zero-mean, second order in the
// rounding error and scale-free (DESIGN.md "loss with rounded V").
template <typename S>
__global__ __launch_bounds__(256) void k_tile_V(_Float16 *VtA, int nrt, int nct, const S *src,
                                                int64_t rows, int64_t cols, int64_t ld, int64_t row0,
                                                int64_t col0, double scale, DevState *st,
                                                const int64_t *row_idx = nullptr, double eps_s = 0.0) {
    __shared__ double red[16];
    const int64_t total = rows * cols;
    double sx = 0, cc = 0, ce = 0, nz = 0;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ii = e / cols, jj = e % cols;
        const double xv = scale * (double)src[(row_idx ? row_idx[ii] : ii) * ld + jj];   // scale includes the storage factor c
        _Float16 xs = (_Float16)xv;
        if (!(xv <= 65504.0)) {                                // beyond the maximum given to klnmf_set_v_max: saturate
            xs = (_Float16)65504.f;                            // (never inf in the matrix) and report at the next loop
            atomicAdd(&st->v_overflow, 1);
        }
        const double xt = (double)xs;
        const int64_t row = row0 + ii, col = col0 + jj;
        const int64_t rt = row >> 5, ctile = col >> 5;
        const int i = row & 31, c = col & 31;
        const int laneA = i + 32 * ((c >> 2) & 1), eA = 4 * (c >> 3) + (c & 3);
        // a lane's 16 values are stored as 16-byte pieces, piece-major: [tile][piece][lane][8 values] (v_tile_load, mfma4.hip.h)
        VtA[(rt * nct + ctile) * 1024 + (eA / 8) * 512 + laneA * 8 + (eA % 8)] = xs;
        sx += xt;
        nz += xt > 0 ? 1.0 : 0.0;
        cc += (xt > 0 && xv > 0) ? xt * log(xt / xv) - (xt - xv) : (xv - xt);
        if (xt > 0) ce += xt * log1p(eps_s / xt);
    }
    const double tsx = block_sum(sx, red);
    const double tcc = block_sum(cc, red);
    const double tce = block_sum(ce, red);
    const double tnz = block_sum(nz, red);
    if (threadIdx.x == 0) {
        atomicAdd(&st->sum_x, tsx);
        atomicAdd(&st->nnz_x, tnz);
        atomicAdd(&st->corr_c, tcc);
        atomicAdd(&st->corr_eps, tce);
    }
}

// loss_local = (ln2 * sum(s1) + sum(s2) - sum_x - C) / c   (fixed summation order)
// decide != 0 (single-context loop, klnmf_run): the stop rule of nmf.py:214-220 in the same launch (k_decide's body;
// one kernel latency less per iteration, which is what a small problem's iteration consists of).
struct LossArgs {
    const double2 *part;      // nullptr: no loss work in this launch
    int64_t count;
    double inv_c;
    double *out;
    int decide;
    DevState *st_rw;
    double tol_abs;
    double *errors;
    int64_t cap;
    int ne;                   // the partials come from an update pass without the numerator's eps: add DevState.corr_eps
    int cq_on;                // ... from a pass over a ratio-scaled dictionary image: its ratios are 2^cq_e too small (k_ratio_scale)
};
// one block: fixed-order fp64 reduction of the row pass's loss partials, then (decide) the stop rule of nmf.py:214-220
__device__ __forceinline__ void loss_from_parts_block(const LossArgs &la, const DevState *st, double *red) {
    // eight partials in flight per thread: ONE block walks all of them (31 250 at n = 10^6), and with one dependent load per
    // trip that was 64 us -- the whole duration of the slab-sum launch it rides in (the slab sum itself: 15 us)
    constexpr int U = 8;
    double au[U], bu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) au[u] = bu[u] = 0.0;
    const int64_t bd = blockDim.x;
    int64_t e = threadIdx.x;
    for (; e + (U - 1) * bd < la.count; e += U * bd) {
        double2 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = la.part[e + u * bd];
#pragma unroll
        for (int u = 0; u < U; ++u) { au[u] += p[u].x; bu[u] += p[u].y; }
    }
    for (; e < la.count; e += bd) {                    // fewer than U left for this thread
        const double2 p = la.part[e];
        au[0] += p.x;
        bu[0] += p.y;
    }
    const double a = ((au[0] + au[1]) + (au[2] + au[3])) + ((au[4] + au[5]) + (au[6] + au[7]));
    const double b = ((bu[0] + bu[1]) + (bu[2] + bu[3])) + ((bu[4] + bu[5]) + (bu[6] + bu[7]));
    const double ta = block_sum(a, red);
    const double tb = block_sum(b, red);
    if (threadIdx.x == 0) {
        const double ta_q = la.cq_on ? ta + (double)st->cq_e * st->sum_x : ta;      // sum x log2(ratio) of the unscaled ratio
        const double err = (kLn2 * ta_q + (la.ne ? st->corr_eps : 0.0) + tb - st->sum_x - st->corr_c) * la.inv_c;
        la.out[0] = err;
        la.out[1] = (double)(st->q8_unfixed + st->mon_trips);      // (row shards: summed by the loss exchange, so that every rank sees when fp8 tiles must be given up)
        if (la.decide) {
            if (la.st_rw->prev_err - err < la.tol_abs) {
                la.st_rw->stop = 1;
            } else {
                la.st_rw->prev_err = err;
                la.st_rw->prev2[0] = err; la.st_rw->prev2[1] = err;
                if (la.st_rw->n_done < la.cap) la.errors[la.st_rw->n_done] = err;
                la.st_rw->n_done += 1;
            }
        }
    }
}
KL_GLOBAL __launch_bounds__(1024) void k_loss_from_parts(const double2 *part, int64_t count,
                                                          const DevState *st, double inv_c,
                                                          double *out, int decide = 0, DevState *st_rw = nullptr,
                                                          double tol_abs = 0.0, double *errors = nullptr,
                                                          int64_t cap = 0, int ne = 0, int cq_on = 0) {
    if (st->stop) return;
    __shared__ double red[16];
    const LossArgs la{part, count, inv_c, out, decide, st_rw, tol_abs, errors, cap, ne, cq_on};
    loss_from_parts_block(la, st, red);
}

}  // namespace klnmf
