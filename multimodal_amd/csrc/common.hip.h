// Shared device-side definitions for the KL-NMF path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Non-template kernels defined in these headers have internal linkage: the library is several translation units (ctx.hip.h),
// each of which compiles the kernels it launches.
#define KL_GLOBAL static __global__

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
    double corr_eps;   // sum over the stored V of x ln(1 + eps/x): what the loss of an update pass WITHOUT the numerator's eps
                       // (ratio x / (W.H + eps): mfma4.hip.h, NE) lacks against the reference's x ln((x + eps) / (W.H + eps))
    double nnz_x;      // entries of the stored V that are > 0 (counted at upload): the fp8 decision of a loop needs enough of them per
                       // column, not enough ROWS (api_loop.hip, begin_fp8_loop)
    int stop;          // stop rule fired (the `break` of nmf.py:216)
    int n_done;        // updates executed == len(errors)
    int v_overflow;    // uploaded values that exceeded the fp16 range announced with klnmf_set_v_max (saturated)
    int op_range;      // components whose measured fp16 operand images could not hold both factors (max W x max H > 2^30)
    // ---- e4m3 saturation (16-bit modes, fp8 iterations; reset at every loop's entry).  Nothing that saturates reaches an H
    // numerator uncorrected without being counted here (colq8x.hip.h, klnmf_query):
    int w8_sat;        // entries of THIS iteration's e4m3 W image beyond 448 x its scale (reset before each conversion): non-zero ->
                       // the fp8 x fp8 column pass of the iteration returns at once and the f16-operand one runs in its place
    int w8_sat_total;  // ... summed over the loop
    int w8_fallbacks;  // iterations whose column pass ran on the f16 W image for that reason
    int q8_sat_total;  // ratio-tile entries found saturated (ratio > 3584) by the column passes, over the loop
    int q8_list_n;     // entries appended to the fix-up list in this iteration (k_q8_fixup recomputes them exactly and resets it)
    int q8_unfixed;    // saturated ratio entries beyond the list's capacity: their excess over 3584 is missing from an H numerator
    int q8_fix_done;   // blocks of the running fix-up launch that have finished (the last one resets the list)
    int cq_e;          // ratio scale of MEASURED images that follow klnmf_init_W (mfma.hip.h, k_ratio_scale): the dictionary image is
                       // H x 2^cq_e / t, so that W.H comes out 2^cq_e times larger and the ratio 2^cq_e times smaller; 0 otherwise
    // ---- the fp8 monitor (monitor.hip.h; reset at every loop's entry)
    int mon_trips;     // component rows whose statistic exceeded the threshold (any > 0: the loop gives the fp8 regime up)
    int mon_checks;    // monitored iterations so far
    unsigned mon_stat_bits;   // the largest statistic of the loop (bit pattern of a non-negative float)
    unsigned mon_spread_bits; // the smallest relative spread of a monitored column's ratios seen in the loop (bit pattern of a float >= 0)
    unsigned mon_dbg[3];      // its ingredients, largest of the loop each: uncentred bias, noise term, |common factor - 1| of a row
    // prev_error as a two-entry ring for stop rules evaluated inside a multi-block launch (post.hip.h): iteration `it` reads
    // prev2[(it - 1) & 1] -- which no block of its launch writes -- and records its loss in prev2[it & 1]
    double prev2[2];
};
constexpr int kQ8ListCap = 8192;      // (row, column) pairs of saturated ratio entries per iteration that are corrected exactly

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
KL_GLOBAL void k_decide(DevState *st, const double *loss_xchg, double tol_abs,
                         double *errors, int64_t cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->stop) return;
    const double err = loss_xchg[0];
    if (st->prev_err - err < tol_abs) {
        st->stop = 1;
        return;
    }
    st->prev_err = err;
    st->prev2[0] = err; st->prev2[1] = err;
    if (st->n_done < cap) errors[st->n_done] = err;
    st->n_done += 1;
}

KL_GLOBAL void k_reset_state(DevState *st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->prev_err = __longlong_as_double(0x7ff0000000000000LL);
        st->prev2[0] = st->prev_err; st->prev2[1] = st->prev_err;
        st->stop = 0;
        st->n_done = 0;
        st->w8_sat = 0; st->w8_sat_total = 0; st->w8_fallbacks = 0;
        st->q8_sat_total = 0; st->q8_list_n = 0; st->q8_unfixed = 0; st->q8_fix_done = 0;
        st->mon_trips = 0; st->mon_checks = 0; st->mon_stat_bits = 0u; st->mon_dbg[0] = st->mon_dbg[1] = st->mon_dbg[2] = 0u; st->mon_spread_bits = 0x3f800000u;
    }
}

}  // namespace klnmf
