// CSR input (next-row N3): the reference's sparse branch, exact arithmetic (T = double / float).
//
// With scipy-sparse X the reference evaluates W.H and the ratio only on the stored entries of X
// (nmf.py:52-70 `_special_sparse_dot`, 331-334 `_Q`): Q is sparse with X's structure, the W rule is
// Q.H^T (CSR x dense, nmf.py:342), the H rule W^T.Q (nmf.py:349) and the loss
//   sum_nnz x*log((x+eps)/(wh+eps)) - sum x + sum_a colsum(W)_a * rowsum(H)_a      (nmf.py:301-308).
// This is SDDMM + two SpMMs per iteration, HBM/cache bound; no MFMA.  All sums run in a fixed order
// (no atomics): results are reproducible bit for bit.
//
// Layouts: CSR of X (indptr[n+1], indices[nnz], data[nnz], all device arrays), the same entries in
// CSC order (csc_indptr[f+1], csc_rows[nnz], csc_perm[nnz] = position in the CSR arrays), q[nnz] in
// CSR order, HT[f][k] = H transposed (rebuilt once per iteration: the dot products of the SDDMM then
// read two contiguous k-vectors).
#pragma once
#include "common.hip.h"

namespace klnmf {

template <typename T>
__global__ void k_sp_transpose_H(const T *H, T *HT, int64_t k, int64_t f, const DevState *st) {
    if (st && st->stop) return;
    const int64_t total = k * f;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e / k, a = e % k;
        HT[e] = H[a * f + j];
    }
}

// One wave per row: q_p = (x_p + eps) / (W_i . H_:,j_p + eps), loss partial of the row (fp64).
template <typename T>
__global__ __launch_bounds__(64) void k_sp_q(const int64_t *indptr, const int64_t *indices, const T *data, const T *W,
                                              const T *HT, T *q, double *row_loss, int64_t k, T eps, int write_q,
                                              const DevState *st) {
    if (st && st->stop) return;
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const T *w = W + i * k;
    double local = 0;
    for (int64_t p = indptr[i] + lane; p < indptr[i + 1]; p += 64) {
        const T *h = HT + indices[p] * k;
        T wh = T(0);
        for (int64_t a = 0; a < k; ++a) wh += w[a] * h[a];
        const T x = data[p];
        const T qq = (x + eps) / (wh + eps);
        if (write_q) q[p] = qq;
        local += (double)(x * log(qq)) - (double)x;
    }
    local = wave_sum(local);
    if (lane == 0) row_loss[i] = local;
}

// W_new[i][a] = (multiply ? W[i][a] : 1) * sum_{p in row i} v_p * H[a][j_p]      (v = q, or x for W0 = X.H0^T)
template <typename T>
__global__ __launch_bounds__(256) void k_sp_w(const int64_t *indptr, const int64_t *indices, const T *v, const T *Wold,
                                               const T *HT, T *Wnew, int64_t k, int multiply, const DevState *st) {
    if (st && st->stop) return;
    const int64_t i = blockIdx.x;
    for (int64_t a = threadIdx.x; a < k; a += blockDim.x) {
        T s = T(0);
        for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) s += v[p] * HT[indices[p] * k + a];
        Wnew[i * k + a] = multiply ? Wold[i * k + a] * s : s;
    }
}

// numer[a][j] = sum_{p in column j} W[row_p][a] * q[perm_p]
template <typename T>
__global__ __launch_bounds__(256) void k_sp_n(const int64_t *csc_indptr, const int64_t *csc_rows, const int64_t *csc_perm,
                                               const T *q, const T *W, T *numer, int64_t k, int64_t f, const DevState *st) {
    if (st && st->stop) return;
    const int64_t j = blockIdx.x;
    for (int64_t a = threadIdx.x; a < k; a += blockDim.x) {
        T s = T(0);
        for (int64_t p = csc_indptr[j]; p < csc_indptr[j + 1]; ++p) s += W[csc_rows[p] * k + a] * q[csc_perm[p]];
        numer[a * f + j] = s;
    }
}

// Column sums of W in two fixed-order stages: part[b][a] over row blocks of 256, then over b.
template <typename T>
__global__ __launch_bounds__(256) void k_sp_colsum_part(const T *W, double *part, int64_t n, int64_t k, const DevState *st) {
    if (st && st->stop) return;
    const int64_t r0 = blockIdx.x * 256, r1 = min(n, r0 + 256);
    for (int64_t a = threadIdx.x; a < k; a += blockDim.x) {
        double s = 0;
        for (int64_t i = r0; i < r1; ++i) s += (double)W[i * k + a];
        part[blockIdx.x * k + a] = s;
    }
}

// prod[a] = colsum(W)_a * rowsum(H)_a ; one block per component, fixed order.
template <typename T>
__global__ __launch_bounds__(256) void k_sp_dots(const double *wpart, int64_t nblk, const T *H, int64_t k, int64_t f,
                                                  double *prod, const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    const int64_t a = blockIdx.x;
    double hs = 0, ws = 0;
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) hs += (double)H[a * f + j];
    for (int64_t b = threadIdx.x; b < nblk; b += blockDim.x) ws += wpart[b * k + a];
    const double hsum = block_sum(hs, red);
    const double wsum = block_sum(ws, red);
    if (threadIdx.x == 0) prod[a] = hsum * wsum;
}

// loss = sum_i row_loss[i] + sum_a prod[a] ; one block, fixed order.
__global__ __launch_bounds__(1024) void k_sp_loss(const double *row_loss, int64_t n, const double *prod, int64_t k,
                                                   double *out, const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += row_loss[i];
    for (int64_t a = threadIdx.x; a < k; a += blockDim.x) s += prod[a];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) { out[0] = t; out[1] = 0; }
}

}  // namespace klnmf
