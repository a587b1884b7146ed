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
// transposed operand is a stride swap).  256 threads, TT x TT outputs each: tiles of 64 x 64 (TT = 4) or 128 x 128 (TT = 8).
// blockIdx.z selects a contraction chunk [z*kchunk, (z+1)*kchunk).
// Round 4: with 4 x 4 outputs per thread every contraction step reads 8 operands from LDS for 16 multiply-adds -- in fp64
// that is 128 LDS cycles for 64 VALU cycles per step and workgroup: LDS-bound at a quarter of the fp64 peak (18 TFLOP/s
// measured), with element-wise bounds-checked staging on top.  8 x 8 outputs per thread read 16 operands for 64
// multiply-adds (LDS and VALU time balanced) and stage a quarter of the elements per flop; tiles that do not touch a matrix
// edge load without bounds checks; the next step's operands are prefetched into registers under the current step's
// arithmetic.  The summation order of one output element is unchanged (k ascending within its chunk): for the same chunking
// the results are bit-identical to the 64 x 64 tiles.  128 x 128 tiles are used where their grid still fills the chip
// (klnmf_set_problem); measured (scripts/small_problem_timing.py): see DESIGN.md.
// MF (64 x 64 tiles): the inner product on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 -- wave w owns rows 16 w .. 16 w + 15 of the tile and
// its four 16-column blocks; per 4 contraction steps one double of A and four of B per lane from the SAME LDS images (A[i][k]
// in lane i + 16 k, B[k][j] in lane j + 16 k; result register r of lane l = D[(l >> 4) + 4 r][l & 15]: probed,
// experiments/micro/mfma_f64_probe.hip).  Same fp64 peak as the vector pipe on this part, a sixth of the LDS reads and a
// sixteenth of the instructions: the VALU form is LDS-bound at a quarter of that peak.  Round 1 had tried it and found no
// gain (526 vs 540 us at 2000 x 4096, k = 200) because the bounds-checked synchronous staging bound the kernel then.
template <typename T, typename Epi, int TT = 4, bool MF = false>
__global__ __launch_bounds__(256, (TT == 4 ? 3 : 2)) void k_gemm(int M, int N, int K, const T *A, int64_t ars,
                                              int64_t acs, const T *B, int64_t brs, int64_t bcs,
                                              int kchunk, const DevState *st, Epi epi) {
    if (st && st->stop) return;
    constexpr int TL = 16 * TT;             // tile edge
    __shared__ T As[GK][TL + 4];
    __shared__ T Bs[GK][TL + 4];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * TL, n0 = blockIdx.x * TL;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    T acc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[i][j] = T(0);
    typedef __attribute__((ext_vector_type(4))) double d4_t;
    typedef __attribute__((ext_vector_type(4))) float f4_t;
    d4_t accm[4];
    f4_t accf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { accm[t] = d4_t{0.0, 0.0, 0.0, 0.0}; accf[t] = f4_t{0.f, 0.f, 0.f, 0.f}; }

    const bool a_k_contig = (acs == 1);   // consecutive threads walk the contiguous axis
    const bool b_n_contig = (bcs == 1);
    const bool m_inside = m0 + TL <= M, n_inside = n0 + TL <= N;      // (uniform) this tile's rows of A / columns of B all exist
    // The next contraction step's operands are requested into registers BEFORE this step's arithmetic and written to LDS
    // behind it: the global latency runs under 16 x TT x TT multiply-adds per thread instead of in front of them (rounds
    // 1-3: load -> LDS -> barrier -> compute -> barrier, the latency exposed at every step).
    constexpr int PER = TL * GK / 256;      // elements of A (and of B) per thread and step
    T ra[PER], rb[PER];
    auto fetch = [&](int k0) {
        const bool k_inside = k0 + GK <= kend;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            int m, kk;
            if (a_k_contig) { kk = e % GK; m = e / GK; } else { m = e % TL; kk = e / TL; }
            const int gm = m0 + m, gk = k0 + kk;
            if (m_inside && k_inside) ra[u] = A[(int64_t)gm * ars + (int64_t)gk * acs];
            else ra[u] = (gm < M && gk < kend) ? A[(int64_t)gm * ars + (int64_t)gk * acs] : T(0);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            int n, kk;
            if (b_n_contig) { n = e % TL; kk = e / TL; } else { kk = e % GK; n = e / GK; }
            const int gn = n0 + n, gk = k0 + kk;
            if (n_inside && k_inside) rb[u] = B[(int64_t)gk * brs + (int64_t)gn * bcs];
            else rb[u] = (gn < N && gk < kend) ? B[(int64_t)gk * brs + (int64_t)gn * bcs] : T(0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            int m, kk;
            if (a_k_contig) { kk = e % GK; m = e / GK; } else { m = e % TL; kk = e / TL; }
            As[kk][m] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            int n, kk;
            if (b_n_contig) { n = e % TL; kk = e / TL; } else { kk = e % GK; n = e / GK; }
            Bs[kk][n] = rb[u];
        }
    };
    if (kbeg < kend) { fetch(kbeg); commit(); }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        const bool more = k0 + GK < kend;
        constexpr bool PREF = !(sizeof(T) == 8 && TT == 8);      // (fp64 with 8 x 8 outputs: 128 accumulator registers leave no room for the prefetch)
        if (PREF && more) fetch(k0 + GK);
        if constexpr (MF) {
            static_assert(!MF || TT == 4, "MFMA inner product: 64 x 64 tiles");
            const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
            for (int k4 = 0; k4 < GK / 4; ++k4) {
                const T av = As[4 * k4 + (lane >> 4)][16 * wv + (lane & 15)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const T bv = Bs[4 * k4 + (lane >> 4)][16 * t + (lane & 15)];
                    if constexpr (sizeof(T) == 8) accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, accm[t], 0, 0, 0);
                    else accf[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accf[t], 0, 0, 0);      // (same operand / result layout)
                }
            }
        } else {
        constexpr int UNR = TT == 8 ? 2 : 4;
#pragma unroll UNR
        for (int kk = 0; kk < GK; ++kk) {
            T a[TT], b[TT];
#pragma unroll
            for (int i = 0; i < TT; ++i) a[i] = As[kk][ty * TT + i];
#pragma unroll
            for (int j = 0; j < TT; ++j) b[j] = Bs[kk][tx * TT + j];
#pragma unroll
            for (int i = 0; i < TT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j) acc[i][j] += a[i] * b[j];
        }
        }
        __syncthreads();
        if (more) {
            if (!PREF) fetch(k0 + GK);
            commit();
            __syncthreads();
        }
    }
    if constexpr (MF) {
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                // result register rr of lane l: fp64 D[(l >> 4) + 4 rr][l & 15] (probed); fp32 D[4 (l >> 4) + rr][l & 15]
                const int r = m0 + 16 * wv + (sizeof(T) == 8 ? (lane >> 4) + 4 * rr : 4 * (lane >> 4) + rr), c = n0 + 16 * t + (lane & 15);
                if (r < M && c < N) epi.apply(r, c, sizeof(T) == 8 ? (T)accm[t][rr] : (T)accf[t][rr]);
            }
    } else {
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int r = m0 + ty * TT + i, c = n0 + tx * TT + j;
            if (r < M && c < N) epi.apply(r, c, acc[i][j]);
        }
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

// The same rule for long rows (round 4: the CSR problems have f = 110 000 columns and k = 50 components -- 50 blocks walked
// 880 KB each, three dependent passes: 0.33 ms of a 3.3 ms iteration): the row in S segments, two launches, no communication
// inside a launch.  k_update_H_part: H * num written back + the segment's fp64 partial sum; k_update_H_norm: every block adds
// the row's S partial sums in the same fixed order and divides its segment.
template <typename T>
__global__ __launch_bounds__(256) void k_update_H_part(T *H, const T *num, int64_t f, int64_t seg, double *part, const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    const int64_t a = blockIdx.y, j0 = blockIdx.x * seg, j1 = min(f, j0 + seg);
    T *row = H + a * f;
    const T *nrow = num + a * f;
    double s = 0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const T v = row[j] * nrow[j];
        row[j] = v;
        s += (double)v;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) part[a * gridDim.x + blockIdx.x] = t;
}
template <typename T>
__global__ __launch_bounds__(256) void k_update_H_norm(T *H, int64_t f, int64_t seg, const double *part, const DevState *st) {
    if (st && st->stop) return;
    const int64_t a = blockIdx.y, j0 = blockIdx.x * seg, j1 = min(f, j0 + seg);
    double total = 0;
    for (unsigned z = 0; z < gridDim.x; ++z) total += part[a * gridDim.x + z];      // (every thread: the same order, the same bits)
    const T d = (T)(kEpsNorm + total);
    T *row = H + a * f;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) row[j] = row[j] / d;
}

// The stop rule of nmf.py:214-220 behind a one-block loss reduction (single-context loops of the exact modes: k_decide as a
// launch of its own is a seventh of a small problem's iteration).
struct DecideArgs {
    int on;                   // 0: the loss only (it is exchanged or read by the caller; the rule follows elsewhere)
    DevState *st_rw;
    double tol_abs;
    double *errors;
    int64_t cap;
};
__device__ __forceinline__ void decide_here(const DecideArgs &d, double err) {
    if (!d.on) return;
    if (d.st_rw->prev_err - err < d.tol_abs) {
        d.st_rw->stop = 1;
        return;
    }
    d.st_rw->prev_err = err;
    d.st_rw->prev2[0] = err; d.st_rw->prev2[1] = err;
    if (d.st_rw->n_done < d.cap) d.errors[d.st_rw->n_done] = err;
    d.st_rw->n_done += 1;
}

// Sum of `count` doubles in a fixed order (deterministic), one block.
KL_GLOBAL __launch_bounds__(1024) void k_sum_doubles(const double *part, int64_t count,
                                                      double *out, const DevState *st,
                                                      DecideArgs dec = DecideArgs{0, nullptr, 0.0, nullptr, 0}) {
    if (st && st->stop) return;
    __shared__ double red[16];
    double s = 0;
    for (int64_t e = threadIdx.x; e < count; e += blockDim.x) s += part[e];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) { out[0] = t; out[1] = 0; decide_here(dec, t); }
}

// The H rule straight from the row chunks' slabs (single-context loops): num[j] = sum_z part[z][a][j] in k_sum_partials'
// order, then exactly k_update_H -- one launch instead of two, the same bits.
template <typename T>
__global__ __launch_bounds__(256) void k_update_H_slabs(T *H, const T *part, int nslab, int64_t slab, int64_t f, const DevState *st) {
    if (st && st->stop) return;
    __shared__ double red[16];
    __shared__ double total;
    T *row = H + blockIdx.x * f;
    const T *prow = part + blockIdx.x * f;
    double s = 0;
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) {
        T nj = T(0);
        for (int z = 0; z < nslab; ++z) nj += prow[z * slab + j];
        const T v = row[j] * nj;
        row[j] = v;
        s += (double)v;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) total = t;
    __syncthreads();
    const T d = (T)(kEpsNorm + total);
    for (int64_t j = threadIdx.x; j < f; j += blockDim.x) row[j] = row[j] / d;
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
