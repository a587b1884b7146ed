// Shared device-side definitions for the KL-NMF path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace klnmf {

// eps of the ratio / loss: the reference hard-codes 1e-8 in _Q / error /
// generalized_KL (nmf.py:232,297,325; metrics.py:15), independent of self.eps.
constexpr double kEpsRatio = 1.0e-8;
// eps of normalize_sum (array_utils.py:19).
constexpr double kEpsNorm = 1.0e-16;
constexpr float kLn2f = 0.6931471805599453f;
constexpr double kLn2 = 0.6931471805599453094;

// Loop state kept on the device so the whole loop of nmf.py:212-222 can be
// enqueued without host synchronisation.
struct DevState {
    double prev_err;   // prev_error of nmf.py:206 (starts at +inf)
    double sum_x;      // sum of V as stored (bf16 modes: the loss is assembled from partial sums)
    double corr_c;     // storage-rounding correction of the loss (0 when V is stored exactly)
    int stop;          // stop rule fired (the `break` of nmf.py:216)
    int n_done;        // updates executed == len(errors)
    int v_overflow;    // uploaded values that exceeded the fp16 range announced with klnmf_set_v_max (saturated)
    int op_range;      // components whose measured fp16 operand images could not hold both factors (max W x max H > 2^30)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum of a double; result valid in thread 0.  `red` holds >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// The stop rule of nmf.py:214-220, one thread.
__global__ void k_decide(DevState *st, const double *loss_xchg, double tol_abs,
                         double *errors, int64_t cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->stop) return;
    const double err = loss_xchg[0];
    if (st->prev_err - err < tol_abs) {
        st->stop = 1;
        return;
    }
    st->prev_err = err;
    if (st->n_done < cap) errors[st->n_done] = err;
    st->n_done += 1;
}

__global__ void k_reset_state(DevState *st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->prev_err = __longlong_as_double(0x7ff0000000000000LL);
        st->stop = 0;
        st->n_done = 0;
    }
}

}  // namespace klnmf
