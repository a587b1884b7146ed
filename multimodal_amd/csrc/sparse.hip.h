// CSR input (next-row N3): the reference's sparse branch, exact arithmetic (T = double / float) -- the unblocked kernels (k > 512 and
// degenerate shapes), the sums around the passes and the helpers; the product kernels for k <= 512 are in sparseb.hip.h.
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
#include "exact.hip.h"

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

// Lane u of the wave receives the value lane u holds in `v`, as a wave-uniform (scalar) number.
__device__ __forceinline__ int64_t lane_value(int64_t v, int u) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, u);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), u);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float lane_value(float v, int u) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), u));
}
__device__ __forceinline__ double lane_value(double v, int u) {
    return __longlong_as_double(lane_value((int64_t)__double_as_longlong(v), u));
}

// N vectors of 64 lanes -> their N sums, sum u ending in the lanes whose low bits equal u: at each level the
// vectors are paired (u, u + N/2), a lane keeps the half its bit selects and receives the other lane's
// contribution to it, so the number of live vectors halves per exchange: N - 1 exchanges for N sums
// instead of N * log2(64), and a fixed summation order.
template <typename T, int N>
struct LaneTransposeSum {
    static __device__ __forceinline__ T run(const T (&v)[N], int lane) {
        constexpr int H = N / 2;
        T r[H];
        const bool hi = (lane & H) != 0;
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const T keep = hi ? v[u + H] : v[u];
            const T send = hi ? v[u] : v[u + H];
            r[u] = keep + __shfl_xor(send, H);
        }
        return LaneTransposeSum<T, H>::run(r, lane);
    }
};
template <typename T>
struct LaneTransposeSum<T, 1> {
    static __device__ __forceinline__ T run(const T (&v)[1], int) { return v[0]; }
};

// General k (> 512): every lane takes its own entries and loops over the components.
template <typename T>
__global__ __launch_bounds__(64) void k_sp_q_anyk(const int64_t *indptr, const int64_t *indices, const T *data, const T *W,
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
// Thread a of the block owns component a.  Each wave fetches the row's (j_p, v_p) 64 at a time with one
// coalesced load and hands them round as wave-uniform values, so the gathers of H^T's rows are independent
// loads at scalar addresses that the hardware can keep in flight together.
template <typename T>
__global__ __launch_bounds__(256) void k_sp_w(const int64_t *indptr, const int64_t *indices, const T *v, const T *Wold,
                                               const T *HT, T *Wnew, int64_t k, int multiply, const DevState *st) {
    if (st && st->stop) return;
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = indptr[i], p1 = indptr[i + 1];
    for (int64_t a0 = 0; a0 < k; a0 += blockDim.x) {
        const int64_t a = a0 + threadIdx.x;
        const bool live = a < k;
        T s = T(0);
        for (int64_t p = p0; p < p1; p += 64) {
            const bool mine = p + lane < p1;
            const int64_t my_j = mine ? indices[p + lane] : 0;
            const T my_v = mine ? v[p + lane] : T(0);               // past the end: adds 0 * H^T[0][a]
            const int cnt = (int)min((int64_t)64, p1 - p);
            if (cnt == 64) {
#pragma unroll
                for (int u = 0; u < 64; ++u) {
                    const int64_t ju = lane_value(my_j, u);
                    const T vu = lane_value(my_v, u);
                    const T hv = live ? HT[ju * k + a] : T(0);
                    s += vu * hv;
                }
            } else {
                for (int u = 0; u < cnt; ++u) {                     // (exchange outside the `live` select: a
                    const int64_t ju = __shfl(my_j, u);             //  lane-exchange only sees active lanes)
                    const T vu = __shfl(my_v, u);
                    const T hv = live ? HT[ju * k + a] : T(0);
                    s += vu * hv;
                }
            }
        }
        if (live) Wnew[i * k + a] = multiply ? Wold[i * k + a] * s : s;
    }
}

// numer[a][j] = sum_{p in column j} W[row_p][a] * q[perm_p]          (same scheme over the CSC arrays)
template <typename T>
__global__ __launch_bounds__(256) void k_sp_n(const int64_t *csc_indptr, const int64_t *csc_rows, const int64_t *csc_perm,
                                               const T *q, const T *W, T *numer, int64_t k, int64_t f, const DevState *st) {
    if (st && st->stop) return;
    const int64_t j = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = csc_indptr[j], p1 = csc_indptr[j + 1];
    for (int64_t a0 = 0; a0 < k; a0 += blockDim.x) {
        const int64_t a = a0 + threadIdx.x;
        const bool live = a < k;
        T s = T(0);
        for (int64_t p = p0; p < p1; p += 64) {
            const bool mine = p + lane < p1;
            const int64_t my_i = mine ? csc_rows[p + lane] : 0;
            const T my_q = mine ? q[csc_perm[p + lane]] : T(0);
            const int cnt = (int)min((int64_t)64, p1 - p);
            if (cnt == 64) {
#pragma unroll
                for (int u = 0; u < 64; ++u) {
                    const int64_t iu = lane_value(my_i, u);
                    const T qu = lane_value(my_q, u);
                    const T wv = live ? W[iu * k + a] : T(0);
                    s += wv * qu;
                }
            } else {
                for (int u = 0; u < cnt; ++u) {
                    const int64_t iu = __shfl(my_i, u);
                    const T qu = __shfl(my_q, u);
                    const T wv = live ? W[iu * k + a] : T(0);
                    s += wv * qu;
                }
            }
        }
        if (live) numer[a * f + j] = s;
    }
}

// Column sums of W in two fixed-order stages: part[b][a] over row blocks of kSpColsumRows, then over b.  (32 rows per block
// since round 4: with k = 50 only 50 threads of a block work, and 79 blocks of 256 rows left them a serial walk of 100 us.)
constexpr int kSpColsumRows = 32;
template <typename T>
__global__ __launch_bounds__(256) void k_sp_colsum_part(const T *W, double *part, int64_t n, int64_t k, const DevState *st) {
    if (st && st->stop) return;
    const int64_t r0 = blockIdx.x * (int64_t)kSpColsumRows, r1 = min(n, r0 + kSpColsumRows);
    for (int64_t a = threadIdx.x; a < k; a += blockDim.x) {
        double s = 0;
        for (int64_t i = r0; i < r1; ++i) s += (double)W[i * k + a];
        part[blockIdx.x * k + a] = s;
    }
}

// Row sums of H in S segments (long rows: one block per component walked 880 KB at f = 110 000): hpart[a][s].
template <typename T>
__global__ __launch_bounds__(256) void k_sp_hsum_part(const T *H, int64_t f, int64_t seg, double *hpart, const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    const int64_t a = blockIdx.y, j0 = blockIdx.x * seg, j1 = min(f, j0 + seg);
    double hs = 0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) hs += (double)H[a * f + j];
    const double t = block_sum(hs, red);
    if (threadIdx.x == 0) hpart[a * gridDim.x + blockIdx.x] = t;
}

// prod[a] = colsum(W)_a * rowsum(H)_a ; one block per component, fixed order.  hpart != nullptr: the row sum from its S
// segment sums (k_sp_hsum_part) instead of a walk over the row.
template <typename T>
__global__ __launch_bounds__(256) void k_sp_dots(const double *wpart, int64_t nblk, const T *H, int64_t k, int64_t f,
                                                  double *prod, const DevState *st, const double *hpart = nullptr, int S = 0) {
    if (st && st->stop) return;
    __shared__ double red[16];
    const int64_t a = blockIdx.x;
    double hs = 0, ws = 0;
    if (hpart != nullptr) {
        if (threadIdx.x == 0) for (int z = 0; z < S; ++z) hs += hpart[a * S + z];
    } else
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) hs += (double)H[a * f + j];
    for (int64_t b = threadIdx.x; b < nblk; b += blockDim.x) ws += wpart[b * k + a];
    const double hsum = block_sum(hs, red);
    const double wsum = block_sum(ws, red);
    if (threadIdx.x == 0) prod[a] = hsum * wsum;
}

// loss = sum_i row_loss[i] + sum_a prod[a] ; one block, fixed order.
KL_GLOBAL __launch_bounds__(1024) void k_sp_loss(const double *row_loss, int64_t n, const double *prod, int64_t k,
                                                   double *out, const DevState *st,
                                                   DecideArgs dec = DecideArgs{0, nullptr, 0.0, nullptr, 0}) {
    if (st && st->stop) return;
    __shared__ double red[16];
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += row_loss[i];
    for (int64_t a = threadIdx.x; a < k; a += blockDim.x) s += prod[a];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) { out[0] = t; out[1] = 0; decide_here(dec, t); }
}

}  // namespace klnmf
