// Exact-arithmetic kernels (fp64 = the reference's arithmetic, or fp32): the
// four dense contractions of one multiplicative update as LDS-tiled VALU GEMMs
// with the element-wise work fused into their epilogues.  This is the mode the
// tight parity tests run; the throughput path is mfma.hip.h.
//
//   k_gemm<.., EpiQ>  W.H -> Q=(V+eps)/(WH+eps), loss partials   nmf.py:325-336, 297-310
//   k_gemm<.., EpiW>  Q.H^T -> W*(.)                              nmf.py:338-343 (and :156 for W0)
//   k_gemm<.., EpiN>  W^T.Q split over row chunks -> partials     nmf.py:349
//   k_sum_partials / k_update_H                                   nmf.py:349-350, array_utils.py:19-22
#pragma once
#include "common.hip.h"

namespace klnmf {

constexpr int GT = 64;   // output tile edge
constexpr int GK = 16;   // contraction step

// C[M,N] = A[M,K] . B[K,N]; element (r,c) of A is A[r*ars + c*acs] (so a
// transposed operand is a stride swap).  256 threads, 4x4 outputs each.
// blockIdx.z selects a contraction chunk [z*kchunk, (z+1)*kchunk).
template <typename T, typename Epi>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, const T *A, int64_t ars,
                                              int64_t acs, const T *B, int64_t brs, int64_t bcs,
                                              int kchunk, const DevState *st, Epi epi) {
    if (st && st->stop) return;
    __shared__ T As[GK][GT + 4];
    __shared__ T Bs[GK][GT + 4];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T(0);

    const bool a_k_contig = (acs == 1);   // consecutive threads walk the contiguous axis
    const bool b_n_contig = (bcs == 1);
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
        for (int e = tid; e < GT * GK; e += 256) {
            int m, kk;
            if (a_k_contig) { kk = e % GK; m = e / GK; } else { m = e % GT; kk = e / GT; }
            const int gm = m0 + m, gk = k0 + kk;
            As[kk][m] = (gm < M && gk < kend) ? A[gm * ars + gk * acs] : T(0);
        }
#pragma unroll
        for (int e = tid; e < GT * GK; e += 256) {
            int n, kk;
            if (b_n_contig) { n = e % GT; kk = e / GT; } else { kk = e % GK; n = e / GK; }
            const int gn = n0 + n, gk = k0 + kk;
            Bs[kk][n] = (gn < N && gk < kend) ? B[gk * brs + gn * bcs] : T(0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = m0 + ty * 4 + i, c = n0 + tx * 4 + j;
            if (r < M && c < N) epi.apply(r, c, acc[i][j]);
        }
    epi.finish(red);
}

// Q = (V+eps)/(WH+eps) and the loss terms x*log(q) - x + y (metrics.py:18-20).
template <typename T>
struct EpiQ {
    const T *V; T *Q; int64_t f; double *loss_part; int write_q; double local; T eps;
    __device__ void apply(int r, int c, T y) {
        const T x = V[(int64_t)r * f + c];
        const T q = (x + eps) / (y + eps);
        if (write_q) Q[(int64_t)r * f + c] = q;
        local += (double)(x * log(q) - x + y);
    }
    __device__ void finish(double *red) {
        const double t = block_sum(local, red);
        if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
};

// W_new = W_old * acc (update) or acc (W0 = V.H^T).
template <typename T>
struct EpiW {
    const T *Wold; T *Wnew; int64_t k; int multiply;
    __device__ void apply(int r, int c, T g) {
        const int64_t o = (int64_t)r * k + c;
        Wnew[o] = multiply ? Wold[o] * g : g;
    }
    __device__ void finish(double *) {}
};

// Split contraction of the W rule (few rows: n*k/4096 output tiles would leave the chip idle while each walks all of
// f): blockIdx.z = chunk of the feature axis, partial Q.H^T into slab z; k_wrule_exact sums the slabs in a fixed order.
template <typename T>
struct EpiWpart {
    T *P; int64_t k; int64_t slab;      // slab = n*k
    __device__ void apply(int r, int c, T g) { P[blockIdx.z * slab + (int64_t)r * k + c] = g; }
    __device__ void finish(double *) {}
};
template <typename T>
__global__ void k_wrule_exact(const T *part, int nslab, int64_t count, const T *Wold, T *Wnew, int multiply,
                              const DevState *st) {
    if (st && st->stop) return;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        T g = T(0);
        for (int z = 0; z < nslab; ++z) g += part[z * count + e];
        Wnew[e] = multiply ? Wold[e] * g : g;
    }
}

// Plain store C[r][c] = acc (reconstruction GEMM, learner.py:80-84).
template <typename T>
struct EpiStore {
    T *C; int64_t ldc;
    __device__ void apply(int r, int c, T v) { C[(int64_t)r * ldc + c] = v; }
    __device__ void finish(double *) {}
};

// Partial numerator of the H rule for one row chunk.
template <typename T>
struct EpiN {
    T *Npart; int64_t f; int64_t slab;   // slab = k*f
    __device__ void apply(int r, int c, T v) { Npart[blockIdx.z * slab + (int64_t)r * f + c] = v; }
    __device__ void finish(double *) {}
};

template <typename T>
__global__ void k_sum_partials(const T *part, T *out, int64_t count, int nslab,
                               const DevState *st) {
    if (st && st->stop) return;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < count;
         e += (int64_t)gridDim.x * blockDim.x) {
        T s = T(0);
        for (int z = 0; z < nslab; ++z) s += part[z * count + e];
        out[e] = s;
    }
}

// H <- H*num, rows divided by (1e-16 + row sum).  One block per component row.
template <typename T>
__global__ __launch_bounds__(256) void k_update_H(T *H, const T *num, int64_t f,
                                                  const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    __shared__ double total;
    T *row = H + blockIdx.x * f;
    const T *nrow = num + blockIdx.x * f;
    double s = 0;
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) {
        const T v = row[j] * nrow[j];
        row[j] = v;
        s += (double)v;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) total = t;
    __syncthreads();
    const T d = (T)(kEpsNorm + total);
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) row[j] = row[j] / d;
}

// Sum of `count` doubles in a fixed order (deterministic), one block.
__global__ __launch_bounds__(1024) void k_sum_doubles(const double *part, int64_t count,
                                                      double *out, const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    double s = 0;
    for (int64_t e = threadIdx.x; e < count; e += blockDim.x) s += part[e];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) { out[0] = t; out[1] = 0; }
}

// V[row0+i, col0+j] = scale * src[i, j]  (learner.py:53-56 fused into the upload).
template <typename T, typename S>
__global__ void k_place_V(T *V, int64_t f, const S *src, int64_t rows, int64_t cols, int64_t ld,
                          int64_t row0, int64_t col0, double scale, const int64_t *row_idx = nullptr) {
    const int64_t total = rows * cols;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / cols, j = e % cols;
        V[(row0 + i) * f + col0 + j] = (T)(scale * (double)src[(row_idx ? row_idx[i] : i) * ld + j]);
    }
}

template <typename D, typename S>
__global__ void k_convert(D *dst, const S *src, int64_t count) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < count;
         e += (int64_t)gridDim.x * blockDim.x)
        dst[e] = (D)src[e];
}

// generalized_KL of two flat arrays (metrics.py:18-20), partial per block.
template <typename T>
__global__ __launch_bounds__(256) void k_gkl(const T *x, const T *y, int64_t count, double eps,
                                             double *part) {
    __shared__ double red[16];
    double s = 0;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < count;
         e += (int64_t)gridDim.x * blockDim.x) {
        const double xv = (double)x[e], yv = (double)y[e];
        s += xv * log((xv + eps) / (yv + eps)) - xv + yv;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---- pairwise distances for the nearest-neighbour evaluation (next-row N4) --------------------------
// out[i][j] = measure(A[i, :], B[j, :]) for the measures of metrics.py:58-86 (as used by
// evaluation.py:103-116 `all_distances`, which broadcasts [n_a,1,d] x [1,n_b,d]).  One wave per pair.
enum DistMetric { DIST_KL = 0, DIST_REV_KL = 1, DIST_SYM_KL = 2, DIST_FROBENIUS = 3, DIST_COSINE_DIFF = 4 };

template <typename T>
__global__ __launch_bounds__(256) void k_all_distances(const T *A, const T *B, T *out, int64_t na, int64_t nb,
                                                        int64_t d, int metric, double eps, int64_t lda = -1, int64_t ldb = -1) {
    if (lda < 0) lda = d;
    if (ldb < 0) ldb = d;
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= na * nb) return;
    const int64_t i = pair / nb, j = pair % nb;
    const T *a = A + i * lda, *b = B + j * ldb;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int64_t e = lane; e < d; e += 64) {
        const double x = (double)a[e], y = (double)b[e];
        if (metric <= DIST_SYM_KL) {
            const double l = log((x + eps) / (y + eps));
            s0 += x * l - x + y;                    // generalized_KL(a, b)
            s1 += -y * l - y + x;                   // generalized_KL(b, a): log((y+eps)/(x+eps)) = -l
        } else if (metric == DIST_FROBENIUS) {
            s0 += (x - y) * (x - y);
        } else {
            s0 += x * y; s1 += x * x; s2 += y * y;
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane != 0) return;
    double r;
    switch (metric) {
        case DIST_KL: r = s0; break;
        case DIST_REV_KL: r = s1; break;
        case DIST_SYM_KL: r = 0.5 * (s0 + s1); break;
        case DIST_FROBENIUS: r = sqrt(s0); break;
        default: r = -(s0 / (sqrt(s1 * s2) + (s0 == 0.0 ? 1.0 : 0.0))); break;     // metrics.py:71-77
    }
    out[pair] = (T)r;
}

}  // namespace klnmf
