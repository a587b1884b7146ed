// CSR input, blocked for the L2 (round 5): the reference's sparse branch (nmf.py:52-70, 301-308, 331-334, 342, 349) as
//   ONE pass over the stored entries for ratio + loss + W rule (SDDMM and SpMM fused), column-blocked, and
//   one pass over the same entries in CSC order for the H numerator, row-blocked.
//
// Why blocks.  Per stored entry the passes gather a k-vector of H^T (ratio, W rule) or of W (H numerator): 3 x nnz x k x es
// bytes per iteration against nnz x (index + value) compulsory ones -- 13.2 GB against 1.04 GB at 20 000 x 110 000, 0.5 %
// stored, k = 50, fp64.  Round 4 served them from the Infinity Cache (H^T is 44 MB): 3.7 .. 6.5 TB/s, 2.75 ms per iteration,
// 0.047 of the HBM roof on compulsory bytes.  An XCD's 4 MiB L2 delivers gathered rows at twice the Infinity Cache's rate
// (MI355X_MICROARCH.md, "Indexed rows"), so the entries are walked block by block:
//   * column blocks of `cb_cols` columns with cb_cols x k x es <= 3 MiB -- the block's rows of H^T stay in every XCD's L2
//     while ALL workgroups work on that block (blockIdx = block x rows: dispatch order keeps the chip on one block);
//   * the ratio needs the whole dot product W_i . H_:,j, which a column block has -- one entry, one column; the W rule
//     sums over ALL entries of a row: per (row, block) partial sums G[b][i][:] (fp64 / fp32 slabs, summed in block order by
//     k_spb_wrule: fixed order, no atomics), and the second gather of an entry's row of H^T -- for the W rule, right behind
//     the dot product -- hits the L1 it was just loaded into: the two products cost ONE pass of gathers;
//   * row blocks of `rb_rows` rows (rb_rows x k x es <= 3 MiB of W) for the H numerator over the CSC order, partial sums
//     per row block in TRANSPOSED layout [rb][column][k] (one coalesced k-vector per wave), summed and transposed by
//     k_spb_numer.
// Block pointers are built on the device at upload from the sorted CSR / CSC arrays (binary search per row and block);
// indices are kept as int32 (half the index bytes).  Results equal the unblocked kernels' to summation order (G6 / G9 at 1e-9).
#pragma once
#include "sparse.hip.h"

namespace klnmf {

constexpr int64_t kSpBlockBytes = (int64_t)3 << 20;       // of H^T (column block) / of W (row block): fits a 4 MiB L2 beside the streams

// blkptr[r][b] = first position in [ptr[r], ptr[r+1]) whose index is >= b * width (b = 0 .. nb; blkptr[r][nb] = ptr[r+1]);
// idx sorted ascending inside every row.  bad: set when a row is not sorted.
KL_GLOBAL void k_spb_blkptr(const int64_t *ptr, const int64_t *idx, int64_t rows, int nb, int64_t width, int64_t *blkptr, int *bad) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= rows * (nb + 1)) return;
    const int64_t r = e / (nb + 1);
    const int b = (int)(e % (nb + 1));
    const int64_t p0 = ptr[r], p1 = ptr[r + 1];
    const int64_t key = (int64_t)b * width;
    int64_t lo = p0, hi = p1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (idx[mid] < key) lo = mid + 1; else hi = mid;
    }
    blkptr[e] = b == nb ? p1 : lo;
    if (b == 0)
        for (int64_t p = p0 + 1; p < p1; ++p)
            if (idx[p] < idx[p - 1]) { *bad = 1; break; }
}
KL_GLOBAL void k_spb_narrow(const int64_t *src, int *dst, int64_t count) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) dst[e] = (int)src[e];
}

enum SpbMode { SPB_LOSS = 0, SPB_UPDATE = 1, SPB_INIT = 2 };

// One wave per (row i, column block b); blockIdx.x = b * n + i.  Lane l holds components l, l + 64, ... of W_i.
//   SPB_UPDATE  q_p = (x_p + eps) / (W_i . H_:,j_p + eps) -> q (CSR order), loss partial, G[b][i][a] = sum_p q_p H[a][j_p]
//   SPB_LOSS    the loss partial only (klnmf_error)
//   SPB_INIT    G[b][i][a] = sum_p x_p H[a][j_p]                       (W0 = X . H0^T, nmf.py:156)
// 32 entries per trip: 32 gathers of a k-vector at wave-uniform addresses in flight, LaneTransposeSum turns the 32 vectors of
// per-lane partial products into the 32 dot products (both lane halves; the lower one takes division and logarithm), then
// the same 32 rows of H^T -- now in L1 -- are multiplied by their ratios into the W rule's partial sums.
template <typename T, int KC, int MODE>
__global__ __launch_bounds__(64) void k_spb_qw(const int64_t *blkptr, const int *indices, const T *data, const T *W, const T *HT,
                                               T *q, double *loss_part, T *G, int64_t n, int64_t k, int nb, T eps, const DevState *st) {
    // KEEP (k <= 128): the gathered rows stay in registers from the dot products to the W rule -- ONE pass of gathers.  (Reloading
    // them "from L1" was an illusion: 16 waves x 32 rows x 400 B per CU are 200 KB against 32 KB of L1, the second pass missed it
    // and the kernel sat on the gather roof -- 8.5 TB/s -- with twice the bytes: 1.37 ms instead of 0.8 at the bench's shape.)
    constexpr bool KEEP = KC <= 2;
    constexpr int NB = KC == 1 ? 32 : (KC == 2 ? 16 : 32);            // entries per trip (NB x KC rows of registers in the KEEP form)
    if (st && st->stop) return;
    const int64_t i = blockIdx.x % n;
    const int b = (int)(blockIdx.x / n);
    const int lane = threadIdx.x, el = lane & (NB - 1);
    const int64_t p0 = blkptr[i * (nb + 1) + b], p1 = blkptr[i * (nb + 1) + b + 1];
    T w[KC], acc[KC];
    bool live[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        live[c] = 64 * c + lane < k;
        w[c] = (MODE != SPB_INIT && live[c]) ? W[i * k + 64 * c + lane] : T(0);
        acc[c] = T(0);
    }
    double local = 0;
    for (int64_t p = p0; p < p1; p += NB) {
        const bool mine = lane < NB && p + lane < p1;
        const bool have = p + el < p1;
        const int my_j = have ? __builtin_nontemporal_load(indices + p + el) : __builtin_nontemporal_load(indices + p0);      // (past the end: a row of this block, unused)
        const T x = mine ? __builtin_nontemporal_load(data + p + lane) : T(0);
        const int cnt = (int)min((int64_t)NB, p1 - p);
        T qq = x;                                              // SPB_INIT: the W rule multiplies the entry itself
        T hv[KEEP ? NB : 1][KEEP ? KC : 1];
        if (KEEP) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const T *h = HT + (int64_t)__builtin_amdgcn_readlane(my_j, u) * k;
#pragma unroll
                for (int c = 0; c < KC; ++c) hv[KEEP ? u : 0][KEEP ? c : 0] = live[c] ? h[64 * c + lane] : T(0);
            }
        }
        if (MODE != SPB_INIT) {
            T part[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                part[u] = T(0);
                if (KEEP) {
#pragma unroll
                    for (int c = 0; c < KC; ++c) part[u] += w[c] * hv[KEEP ? u : 0][KEEP ? c : 0];
                } else {
                    const T *h = HT + (int64_t)__builtin_amdgcn_readlane(my_j, u) * k;
#pragma unroll
                    for (int c = 0; c < KC; ++c)
                        if (live[c]) part[u] += w[c] * h[64 * c + lane];
                }
            }
            T wh = LaneTransposeSum<T, NB>::run(part, lane);
#pragma unroll
            for (int o = NB; o < 64; o <<= 1) wh += __shfl_xor(wh, o);      // every group of NB lanes summed its own lanes
            qq = T(0);
            if (mine) {
                qq = (x + eps) / (wh + eps);
                if (MODE == SPB_UPDATE) __builtin_nontemporal_store(qq, q + p + lane);
                local += (double)(x * log(qq)) - (double)x;
            }
        }
        if (MODE != SPB_LOSS) {
            if (KEEP) {                    // (entries past the end carry qq = 0)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const T qu = lane_value(qq, u);
#pragma unroll
                    for (int c = 0; c < KC; ++c) acc[c] += qu * hv[KEEP ? u : 0][KEEP ? c : 0];
                }
            } else if (cnt == NB) {
                // k > 128: the rows are gathered a second time, all 32 loads of a component block issued before the first multiply-add
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    if (live[c]) {
                        T hr[NB];
#pragma unroll
                        for (int u = 0; u < NB; ++u) hr[u] = HT[(int64_t)__builtin_amdgcn_readlane(my_j, u) * k + 64 * c + lane];
#pragma unroll
                        for (int u = 0; u < NB; ++u) acc[c] += lane_value(qq, u) * hr[u];
                    }
                }
            } else {
                for (int u = 0; u < cnt; ++u) {                // (wave-uniform trip count)
                    const T *h = HT + (int64_t)__shfl(my_j, u) * k;
                    const T qu = __shfl(qq, u);
#pragma unroll
                    for (int c = 0; c < KC; ++c)
                        if (live[c]) acc[c] += qu * h[64 * c + lane];
                }
            }
        }
    }
    if (MODE != SPB_INIT) {
        local = wave_sum(local);
        if (lane == 0) loss_part[(int64_t)b * n + i] = local;
    }
    if (MODE != SPB_LOSS) {
        T *g = G + ((int64_t)b * n + i) * k;
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (live[c]) g[64 * c + lane] = acc[c];
    }
}

// W_new[i][a] = (multiply ? W[i][a] : 1) * sum_b G[b][i][a]        (fixed block order)
template <typename T>
__global__ __launch_bounds__(256) void k_spb_wrule(const T *G, int nb, const T *Wold, T *Wnew, int64_t n, int64_t k, int multiply,
                                                   const DevState *st) {
    if (st && st->stop) return;
    const int64_t total = n * k;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        T s = G[e];
        for (int b = 1; b < nb; ++b) s += G[(int64_t)b * total + e];
        Wnew[e] = multiply ? Wold[e] * s : s;
    }
}

// One wave per (column j, row block rb); blockIdx.x = rb * f + j.  NT[rb][j][a] = sum_{p in column j, row block rb} W[row_p][a] q[perm_p]
template <typename T, int KC>
__global__ __launch_bounds__(64) void k_spb_n(const int64_t *cblkptr, const int *csc_rows, const int *csc_perm, const T *q, const T *W,
                                              T *NT, int64_t k, int64_t f, int nrb, const DevState *st) {
    if (st && st->stop) return;
    const int64_t j = blockIdx.x % f;
    const int rb = (int)(blockIdx.x / f);
    const int lane = threadIdx.x;
    const int64_t p0 = cblkptr[j * (nrb + 1) + rb], p1 = cblkptr[j * (nrb + 1) + rb + 1];
    T acc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] = T(0);
    for (int64_t p = p0; p < p1; p += 64) {
        const bool mine = p + lane < p1;
        const int my_i = mine ? __builtin_nontemporal_load(csc_rows + p + lane) : 0;
        const T my_q = mine ? q[__builtin_nontemporal_load(csc_perm + p + lane)] : T(0);
        const int cnt = (int)min((int64_t)64, p1 - p);
        // groups of 16 entries, all 16 gathers of a component block issued before the first multiply-add (entries past the end
        // carry q = 0 and row 0: no divergence; one dependent load per entry was a chain of L2 latencies)
#pragma unroll
        for (int u0 = 0; u0 < 64; u0 += 16) {
            if (u0 < cnt) {
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    if (64 * c + lane < k) {
                        T wv[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) wv[u] = W[(int64_t)__builtin_amdgcn_readlane(my_i, u0 + u) * k + 64 * c + lane];
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc[c] += wv[u] * lane_value(my_q, u0 + u);
                    }
                }
            }
        }
    }
    T *o = NT + ((int64_t)rb * f + j) * k;
#pragma unroll
    for (int c = 0; c < KC; ++c)
        if (64 * c + lane < k) o[64 * c + lane] = acc[c];
}

// numer[a][j] = sum_rb NT[rb][j][a]: summed in row-block order and transposed through LDS (32 columns x 32 components per step:
// coalesced on both sides)
template <typename T>
__global__ __launch_bounds__(256) void k_spb_numer(const T *NT, int nrb, T *numer, int64_t k, int64_t f, const DevState *st) {
    if (st && st->stop) return;
    __shared__ T tile[32][33];
    const int64_t j0 = (int64_t)blockIdx.x * 32, a0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8 threads
    for (int r = ty; r < 32; r += 8) {
        const int64_t j = j0 + r, a = a0 + tx;
        T s = T(0);
        if (j < f && a < k)
            for (int rb = 0; rb < nrb; ++rb) s += NT[((int64_t)rb * f + j) * k + a];
        tile[r][tx] = s;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t a = a0 + r, j = j0 + tx;
        if (a < k && j < f) numer[a * f + j] = tile[tx][r];
    }
}

}  // namespace klnmf
