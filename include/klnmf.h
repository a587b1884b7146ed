/* klnmf.h -- C-ABI of the MI355X-native KL-divergence NMF hot path.
 *
 * This is the drop-in boundary for the multiplicative-update loop of
 * omangin/multimodal (`multimodal/lib/nmf.py`).  Every entry point below names
 * the reference interface it replaces (file:line, relative to the reference
 * root).  Plain pointers and sizes only; no exceptions cross the boundary:
 * every function returns an int status (0 = ok, <0 = error class) and the
 * message for the calling thread is available from klnmf_last_error().
 *
 * One context = one GPU (one process per GPU in the multi-GPU case).  The
 * collective between row shards is issued INSIDE the library on the native path
 * (klnmf_comm_init + klnmf_run_sharded / klnmf_run_more: RCCL over xGMI), or by
 * the caller between the klnmf_iter_* pieces (e.g. torch.distributed); see
 * INTEGRATION.md.  A context is used from one thread at a time; different
 * contexts may be used from different threads.  All calls are asynchronous on
 * the context's stream unless they return data to the host.
 *
 * Matrices named as in the reference: V/X [n,f] data (rows = samples),
 * W [n,k] coefficients, H [k,f] dictionary (`components_`), Q [n,f] ratio.
 */
#ifndef KLNMF_H
#define KLNMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLNMF_VERSION 100            /* 0.1.0 */

/* status codes */
#define KLNMF_OK            0
#define KLNMF_ERR_ARG      -1        /* bad argument / wrong state        */
#define KLNMF_ERR_ALLOC    -2        /* device or host allocation failed  */
#define KLNMF_ERR_HIP      -3        /* HIP runtime error                 */
#define KLNMF_ERR_UNSUPP   -4        /* shape not supported by this mode  */
#define KLNMF_ERR_RCCL     -5        /* an RCCL call failed, or librccl could not be opened */

/* arithmetic modes */
#define KLNMF_PREC_F64      0        /* fp64 everywhere (reference arithmetic)      */
#define KLNMF_PREC_F32      1        /* fp32 everywhere                              */
#define KLNMF_PREC_F16      2        /* fp16 MFMA operands (power-of-two-scaled images, saturating conversion), fp32
                                      * accumulate and masters, V stored 16-bit (scaled fp16): the throughput mode */
#define KLNMF_PREC_BF16     KLNMF_PREC_F16      /* round-1 name of the mode (its operands were bf16 then) */
/* (3 was KLNMF_PREC_F16_V32 -- the same kernels on fp32-stored V -- until round 5: retired, it missed the 1e-4 bar on a
 *  BASELINE shape and kept the generation-1 kernels in the library for no measured benefit) */

/* host element types for uploads / downloads */
#define KLNMF_DT_F32        0
#define KLNMF_DT_F64        1

typedef struct klnmf_ctx klnmf_ctx;

/* ---- library / device -------------------------------------------------- */
int         klnmf_version(void);
const char *klnmf_last_error(void);
/* arch string (e.g. "gfx950"), CU count, HBM bytes: roofline constants for bench.py */
int         klnmf_device_info(int device, char *arch, int arch_len,
                              int *cu_count, uint64_t *hbm_bytes);

/* ---- context ----------------------------------------------------------- */
/* stream: a hipStream_t to run on (e.g. torch's current stream), NULL for an own (non-blocking)
 * stream, or KLNMF_STREAM_DEFAULT for the device's default (null) stream -- whose handle is 0 and
 * could not be told from "no stream" otherwise; it is what torch reports as its current stream unless
 * the caller switched streams, and collectives issued through torch are ordered against it.  Replaces constructing KLdivNMF (nmf.py:136-145) + the implicit
 * "everything lives in host numpy arrays" of the reference. */
#define KLNMF_STREAM_DEFAULT ((void *)(intptr_t)-1)
int klnmf_create(klnmf_ctx **out, int device, int precision, void *stream);
int klnmf_destroy(klnmf_ctx *ctx);

/* Declare the local problem: n rows held by this context, f features, k
 * components, and the capacity of the per-iteration loss record.
 * (nmf.py:196-199: shapes taken from X and n_components.) */
int klnmf_set_problem(klnmf_ctx *ctx, int64_t n, int64_t f, int64_t k,
                      int64_t max_iter_capacity);

/* ---- data in ------------------------------------------------------------ */
/* KLNMF_PREC_BF16 stores V as fp16 of c*V with c the power of two that puts
 * c*vmax in [2^14, 2^15); call this with vmax >= max(scale * src) over ALL
 * blocks (and all ranks: the factor must be common) before the first upload.
 * Without it c = 1 (fine for data in fp16's range).  No-op in the other modes. */
int klnmf_set_v_max(klnmf_ctx *ctx, double vmax);
/* Give the problem's device memory back (to the library's block cache) and keep the context -- its stream, its
 * communicator -- for a later klnmf_set_problem.  Hosts that run many short fits in sequence (experiment.py:158-180)
 * keep a few contexts alive this way instead of creating one per fit. */
int klnmf_release_problem(klnmf_ctx *ctx);
/* Forget the uploaded matrix: zero V and the sums the upload kernels accumulate (sum of V as stored, the
 * storage-rounding correction of the loss, the count of values beyond the announced maximum), so that the same
 * context can take another matrix of the same shape; klnmf_set_v_max may be called again afterwards.  (Uploading
 * a block twice without this counts it twice in the reported loss.) */
int klnmf_reset_V(klnmf_ctx *ctx);
/* Upload the sub-block V[row0:row0+rows, col0:col0+cols] = scale * src, src a
 * host array with leading dimension ld (elements).  One call per modality
 * block replaces learner.py:53-56 (`safe_hstack([c * m ...])`): the scale,
 * cast and column placement are fused into the upload (K6). */
int klnmf_upload_V(klnmf_ctx *ctx, const void *src, int dtype,
                   int64_t rows, int64_t cols, int64_t ld,
                   int64_t row0, int64_t col0, double scale);
/* Same from a device-resident fp32 block (bench / device-side pipelines). */
int klnmf_upload_V_device(klnmf_ctx *ctx, const float *dsrc,
                          int64_t rows, int64_t cols, int64_t ld,
                          int64_t row0, int64_t col0, double scale);
/* Same with a row gather: row i of the block is row drow_idx[i] (device array) of the device-resident
 * matrix dsrc.  Replaces the per-run `x[train, :]` / `x[test, :]` copies of experiment.py:163-164 and the
 * re-upload that follows them (next-row N2: the stacked data stays on the device across runs). */
int klnmf_upload_V_device_rows(klnmf_ctx *ctx, const float *dsrc, const int64_t *drow_idx,
                               int64_t rows, int64_t cols, int64_t ld,
                               int64_t row0, int64_t col0, double scale);
/* Dictionary [k,f], C order (nmf.py:149-155 `_init_dictionary`, learner.py:13
 * `components_ = dictionary`). */
int klnmf_set_H(klnmf_ctx *ctx, const void *src, int dtype);
/* Coefficients [n,k], C order (tests call `_update(X, W)` with their own W). */
int klnmf_set_W(klnmf_ctx *ctx, const void *src, int dtype);
/* Ratio [n,f] given by the caller (nmf.py:338-351: the `Q=` argument).
 * F64/F32 modes only. */
int klnmf_set_Q(klnmf_ctx *ctx, const void *src, int dtype);

/* ---- the path ----------------------------------------------------------- */
/* W0 = V . H^T  (nmf.py:156). */
int klnmf_init_W(klnmf_ctx *ctx);

/* The whole loop of nmf.py:212-222 on the device, single GPU:
 *   for it in 1..max_iter: err = error(); if prev - err < tol_abs: break;
 *                          errors.append(err); W (and H if fit) = update
 * errors_out receives n_done losses (each recorded before its update);
 * *stopped = 1 when the stop rule fired (the break), 0 when the limit ended
 * the loop.  Synchronous. */
int klnmf_run(klnmf_ctx *ctx, int64_t max_iter, int fit, double tol_abs,
              double *errors_out, int64_t *n_done, int *stopped);

/* The same loop in pieces, for the row-sharded multi-GPU case: the host issues
 * the collective between the pieces (sum of the exchange buffers over ranks).
 *   klnmf_loop_begin                          reset prev_error / counters
 *   per iteration:
 *     klnmf_iter_rowpass   loss partial + W update        -> loss exchange [0]
 *     [all-reduce loss[0]: may run while the column pass computes]
 *     klnmf_iter_colpass   H numerator W_new^T . Q_old    -> numerator exchange; its last launch writes loss[1], the count of
 *                          fp8-monitor trips and unfixable ratio entries of this iteration
 *     [all-reduce numerator exchange]
 *     [all-reduce loss[1] -- BEHIND klnmf_iter_colpass, never together with loss[0] in front of it -- on the iterations for
 *      which klnmf_query(KLNMF_Q_FP8_POLL_DUE) answers 1: the klnmf_iter_advance below then polls the sum, and every rank
 *      leaves the fp8 regime in the same iteration]
 *     klnmf_iter_decide    stop rule (nmf.py:214-220) on loss[0]
 *     klnmf_iter_update_H  H * num, row-normalise (nmf.py:349-350)
 *     klnmf_iter_advance   swap the W_old / W_new buffers
 *   klnmf_loop_end         sync, fetch errors / counters
 * After the stop rule fires every later piece is a no-op on the device, so the
 * host may enqueue all max_iter iterations without synchronising. */
int klnmf_loop_begin(klnmf_ctx *ctx);
/* The same for one rank of a row-sharded problem: the fp8 decision of the loop (16-bit modes) is taken from the sums over
 * ALL shards -- sum of V (klnmf_query_f64 KLNMF_QF_SUM_V, all-reduced) and its element count -- so that every rank runs
 * the same kernels and the result does not depend on the partition beyond summation order. */
int klnmf_loop_begin_sharded(klnmf_ctx *ctx, double sum_v_all, double cells_all);
/* ... and with the number of entries of V that are > 0 over all shards (klnmf_query_f64 KLNMF_QF_NNZ_V, all-reduced): fp8 ratio
 * tiles need enough of them per column, not enough rows (sparse data stored densely).  nnz_all < 0: as the call above (dense). */
int klnmf_loop_begin_sharded_nnz(klnmf_ctx *ctx, double sum_v_all, double cells_all, double nnz_all);
/* ... and with the conjunction over all ranks of "this shard's SHAPE allows fp8 ratio tiles" (klnmf_query
 * KLNMF_Q_RATIO_TILE_BYTES == 1, all-reduced as a minimum): shards differ by a row tile and the last one takes the remainder,
 * so they can straddle the row threshold -- ranks must not mix tile formats (the numerators of the two differ by sqrt(2)).
 * fp8_shape_all < 0: this rank's own shape decides (the calls above).  klnmf_loop_begin on a communicator agrees it itself. */
int klnmf_loop_begin_agreed(klnmf_ctx *ctx, double sum_v_all, double cells_all, double nnz_all, int fp8_shape_all);
/* `iters` whole iterations of the open loop at once, enqueued exactly as klnmf_run enqueues them (callers that fence between
 * two parts of one loop: warm-up | timed iterations of bench.py).
 * On a context that holds an RCCL communicator of more than one rank (klnmf_comm_init below) klnmf_loop_begin and
 * klnmf_run_more are klnmf_run_sharded in parts: the entry agrees the refusals and the fp8 decision over the communicator,
 * every iteration carries the ONE grouped all-reduce (numerator + loss) between column pass and stop rule; tol_abs is then
 * tol x n_total x f of the GLOBAL shape (nmf.py:207). */
int klnmf_run_more(klnmf_ctx *ctx, int64_t iters, int fit, double tol_abs);
int klnmf_iter_rowpass(klnmf_ctx *ctx, int fit);
int klnmf_iter_decide(klnmf_ctx *ctx, double tol_abs);
int klnmf_iter_colpass(klnmf_ctx *ctx);
int klnmf_iter_update_H(klnmf_ctx *ctx);
/* advance the W ping-pong after one iteration's pieces have been enqueued */
int klnmf_iter_advance(klnmf_ctx *ctx);
int klnmf_loop_end(klnmf_ctx *ctx, double *errors_out, int64_t *n_done,
                   int *stopped);
/* ---- Row shards over the GPUs of one node: the native collective path -----------------------------
 * One process (or thread) per GPU, each with its own context holding n_local rows of V and W; the dictionary is
 * replicated.  The reference has no counterpart (it is a single process); what is exchanged per fit iteration is
 * exactly what the algebra needs: the k x f numerator W_new^T.Q of the H rule (nmf.py:349) and the scalar loss
 * (nmf.py:214), summed over the ranks by ONE grouped RCCL all-reduce on the context's stream (xGMI inside a node).
 * The W rule and a transform exchange only the loss.  librccl is opened at run time on the first klnmf_comm_* call.
 *
 *   rank 0:  klnmf_comm_unique_id(id)  -> send the 128 bytes to every rank (any channel: MPI, a file, torch.distributed)
 *   each:    klnmf_comm_init(ctx, id, rank, nranks)
 *            klnmf_comm_max(ctx, &vmax) -> klnmf_set_v_max(ctx, vmax)      (the storage factor must be common)
 *            uploads, klnmf_set_H (same H on every rank), klnmf_init_W
 *            klnmf_run_sharded(ctx, n_total, ...)                           (errors / n_done / stopped identical on all ranks)
 */
#define KLNMF_COMM_ID_BYTES 128
int klnmf_comm_unique_id(void *id_out /* KLNMF_COMM_ID_BYTES */);
int klnmf_comm_init(klnmf_ctx *ctx, const void *id, int rank, int nranks);
int klnmf_comm_destroy(klnmf_ctx *ctx);
/* all-reduce(max) of one host double over the communicator (no-op without one) */
int klnmf_comm_max(klnmf_ctx *ctx, double *value);
/* klnmf_run over row shards: tol is the RELATIVE tolerance of nmf.py:207 (x n_total x f inside, the global shape).
 * Without a communicator (or with one of size 1) it is klnmf_run on the local rows. */
int klnmf_run_sharded(klnmf_ctx *ctx, int64_t n_total, int64_t max_iter, int fit, double tol,
                      double *errors_out, int64_t *n_done, int *stopped);

/* Device pointers of the two exchange buffers (what the collective sums):
 * loss: 2 doubles ([0] the loss partial, [1] this rank's count of fp8 ratio entries beyond the exact fix-up's list: the
 * summed value tells every rank when a loop gives fp8 tiles up); numerator: *numer_count elements of fp32 (BF16 modes,
 * F32) or fp64 (F64).  The buffers are owned by the context. */
int klnmf_exchange_buffers(klnmf_ctx *ctx, void **loss_ptr, void **numer_ptr,
                           int64_t *numer_count, int *numer_is_f64);
/* Layout of the numerator buffer: component rows of *row_stride elements, of which the first k rows
 * (*valid_count elements from the start) carry data -- what a collective has to move; the rest is padding. */
int klnmf_exchange_layout(klnmf_ctx *ctx, int64_t *row_stride, int64_t *valid_count);
/* Column-range form of the numerator exchange (round 4): with KLNMF_COMM_PARTS = P > 1 (read by klnmf_set_problem; 16-bit
 * modes, fused tail) the H numerator of nmf.py:349 can be produced and exchanged in P column parts, so that the all-reduce of
 * part p runs while the column pass of part p + 1 computes.  Part p = columns [col0[p], col0[p] + ncols[p]) of all k
 * components, stored as ONE contiguous block of counts[p] = k x (padded part width) elements at element offset offsets[p]
 * of the numerator buffer (klnmf_exchange_buffers: *numer_count covers this layout too).  Arrays of KLNMF_MAX_PARTS entries;
 * *nparts = 1 when the problem is not split: the one part is the whole-matrix layout of klnmf_exchange_layout.
 *   per iteration:  klnmf_iter_rowpass | for p in 0 .. P-1: klnmf_iter_colpass_part(p) -> [all-reduce part p] |
 *                   klnmf_iter_decide | klnmf_iter_update_H (applies the parts) | klnmf_iter_advance
 * klnmf_run_sharded / klnmf_run_more on a communicator do the same natively (part all-reduces on a second stream). */
#define KLNMF_MAX_PARTS 4
int klnmf_exchange_parts(klnmf_ctx *ctx, int *nparts, int64_t *offsets, int64_t *counts, int64_t *col0, int64_t *ncols);
int klnmf_iter_colpass_part(klnmf_ctx *ctx, int part);
/* (Round 4: the collective of the native path -- klnmf_run_sharded -- is issued INSIDE the library; these entry points serve
 * callers that sequence the loop's pieces around their own collective, e.g. torch.distributed.)
 * Use caller-owned device buffers as the exchange buffers instead (e.g. torch
 * tensors handed to torch.distributed.all_reduce): loss_ptr >= 2 doubles,
 * numer_ptr >= numer_count elements.  NULL keeps the current buffer. */
int klnmf_bind_exchange(klnmf_ctx *ctx, void *loss_ptr, void *numer_ptr);

/* Step-granular entry points (unit parity with the reference's private
 * methods).  Synchronous. */
/* error(X, W, H)                      nmf.py:297-310 + metrics.py:18-20 */
int klnmf_error(klnmf_ctx *ctx, double *loss);
/* Diagnostic: the terms klnmf_error assembles in the 16-bit modes, in the caller's units:
 * terms[0] = sum x~ ln((x~+eps)/(W.H+eps)), [1] = sum W.H, [2] = sum x~ (V as stored),
 * [3] = the storage-rounding correction C = KL(x~ || x); loss = [0] + [1] - [2] - [3]
 * (metrics.py:18-20 split into its three sums).  KLNMF_ERR_UNSUPP in the exact modes. */
int klnmf_loss_terms(klnmf_ctx *ctx, double *terms);
/* one _update(X, W, _fit) without stop rule: nmf.py:232-257 */
int klnmf_update(klnmf_ctx *ctx, int fit);
/* eps of klnmf_step_Q (the `eps=` argument of _Q, nmf.py:325; default 1e-8).
 * The loop itself always uses 1e-8, as the reference does (nmf.py:232,251). */
int klnmf_set_ratio_eps(klnmf_ctx *ctx, double eps);
/* Q = (V+eps)/(W.H+eps) into the context's Q buffer (F64/F32 modes): nmf.py:325-336 */
int klnmf_step_Q(klnmf_ctx *ctx);
/* W <- W * (Q . H^T) with the Q buffer: nmf.py:338-343 */
int klnmf_step_W(klnmf_ctx *ctx);
/* H <- normalize_rows(H * (W^T . Q)) with the Q buffer: nmf.py:345-351 */
int klnmf_step_H(klnmf_ctx *ctx);
/* generalized_KL(x, y) of two host arrays of `count` elements: metrics.py:18-20 */
int klnmf_generalized_kl(klnmf_ctx *ctx, const void *x, const void *y, int dtype,
                         int64_t count, double eps, double *out);

/* ---- data out ----------------------------------------------------------- */
int klnmf_get_W(klnmf_ctx *ctx, void *dst, int dtype);   /* [n,k] */
int klnmf_get_H(klnmf_ctx *ctx, void *dst, int dtype);   /* [k,f] */
int klnmf_get_Q(klnmf_ctx *ctx, void *dst, int dtype);   /* [n,f], F64/F32 modes */

/* ---- device-resident operands (next-row N1) -------------------------------
 * The evaluation of an experiment (experiment.py:233-277, 332-371) runs 2 M .. 12 transforms per run on a fixed dictionary,
 * reconstructs modalities from the coefficients (learner.py:80-84) and compares in every space (evaluation.py:103-116).
 * With these the dictionary, the coefficients and the reconstructions never leave the GPU: pointers are DEVICE memory on the
 * context's device (e.g. torch tensors), row strides in elements.
 *   klnmf_set_H_device           a column block of `components_` (nmf.py:283-284; get_dico / get_stacked_dicos,
 *                                learner.py:43-51) from a device-resident dictionary: H[:, col0 : col0 + ncols] <- src
 *                                ([k, ncols], rows `ld` apart); `last` != 0 on the block that completes the dictionary
 *   klnmf_get_W_device           the coefficients of the last loop (what transform returns, nmf.py:291) -> [n, k], rows `ld` apart
 *   klnmf_upload_V_device_rows_dt klnmf_upload_V_device_rows for float32 or float64 device-resident modalities
 *   klnmf_matmul_device          reconstruct_modalit{y,ies} (learner.py:80-84): C[m,n] = A[m,kk] . B[kk,n]
 *   klnmf_all_distances_device   all_distances (metrics.py:80-86) of two device matrices -> [na, nb] on the device */
int klnmf_set_H_device(klnmf_ctx *ctx, const void *dsrc, int dtype, int64_t ld, int64_t col0, int64_t ncols, int last);
int klnmf_get_W_device(klnmf_ctx *ctx, void *ddst, int dtype, int64_t ld);
int klnmf_upload_V_device_rows_dt(klnmf_ctx *ctx, const void *dsrc, int dtype, const int64_t *drow_idx, int64_t rows,
                                  int64_t cols, int64_t ld, int64_t row0, int64_t col0, double scale);
int klnmf_matmul_device(int device, int dtype, int64_t m, int64_t n, int64_t kk, const void *dA, int64_t lda,
                        const void *dB, int64_t ldb, void *dC, int64_t ldc);
int klnmf_all_distances_device(int device, int dtype, int metric, int64_t na, int64_t nb, int64_t d, const void *dA,
                               int64_t lda, const void *dB, int64_t ldb, void *dout);

/* ---- introspection ------------------------------------------------------ */
/* What a context decided and what its last loop actually ran -- no reference counterpart (the reference has one
 * arithmetic); bench.py and the parity tests read it instead of mirroring the library's rules on the host.
 *   KLNMF_Q_FP8_LOOP          1 if the current / last loop was allowed fp8 ratio tiles (shape and range rules at the loop's entry;
 *                             what the tiles' rounding noise does to this data is measured while the loop runs: the monitor below)
 *   KLNMF_Q_FP8_TILE_ITERS    iterations of that loop whose ratio tiles were fp8 (e4m3 of ratio x sqrt(2) / 8, stochastically rounded)
 *   KLNMF_Q_FP8_COL_ITERS     iterations whose H-numerator product ran on e4m3 operands on both sides
 *   KLNMF_Q_RATIO_TILE_BYTES  bytes per element of V the stored ratio tiles take in an fp8 iteration (1), else 2; 0: none stored
 *   KLNMF_Q_COMM_RANKS        ranks RCCL reports for the context's communicator (ncclCommCount); 1 without one */
#define KLNMF_Q_FP8_LOOP          0
#define KLNMF_Q_FP8_TILE_ITERS    1
#define KLNMF_Q_FP8_COL_ITERS     2
#define KLNMF_Q_RATIO_TILE_BYTES  3
#define KLNMF_Q_COMM_RANKS        4
/* e4m3 saturation in the last loop -- counted, and kept out of the result:
 *   KLNMF_Q_W8_SATURATED     entries of the e4m3 W image beyond 448 x their component's scale (a column that more than doubled
 *                            in one update), KLNMF_Q_W8_FALLBACKS the iterations whose H-numerator product therefore ran on the
 *                            f16 W image instead;
 *   KLNMF_Q_RATIO_SATURATED  ratio-tile entries at the tiles' maximum (ratio >= 3584 / sqrt(2)): each is recomputed exactly and its excess
 *                            added to the H numerator; KLNMF_Q_RATIO_UNFIXED those beyond the correction list's capacity (8192 per
 *                            iteration) -- non-zero means H numerators of this loop were clipped; the loop then drops fp8 tiles */
#define KLNMF_Q_W8_SATURATED      5
#define KLNMF_Q_W8_FALLBACKS      6
#define KLNMF_Q_RATIO_SATURATED   7
#define KLNMF_Q_RATIO_UNFIXED     8
/*   KLNMF_Q_NO_NUM_EPS       1 if that loop's update passes on fp8 tiles formed the ratio as x / (W.H + eps) instead of the
 *                            reference's (x + eps) / (W.H + eps) (nmf.py:332-336): taken at the loop's entry where eps / mean(V)
 *                            <= 1e-5 (k <= 224), V keeps true zeros (2^-100 addend in the ratio), the loss corrected exactly; KLNMF_NE=0 turns it off */
#define KLNMF_Q_NO_NUM_EPS        9
/* The fp8 monitor of the last loop (csrc/monitor.hip.h): on the loop's first iteration (a dry run, before any fp8 tile is
 * taken), on fp8 iterations 1, 2, 4, 8, 16 and every 32nd after them the library recomputes, for one column tile and a sample
 * of rows, the H numerator (nmf.py:349) the 16-bit ratio tiles would have given and compares it with what the fp8 regime
 * produced -- a measured bound on the noise the (unbiased) e4m3 rounding puts into this data's numerator.
 *   KLNMF_Q_MON_CHECKS       monitored iterations;  KLNMF_Q_MON_TRIPS  component rows whose statistic exceeded the threshold;
 *   KLNMF_Q_MON_GAVE_UP      1 if the loop therefore (or after bulk saturation) continued on 16-bit tiles;
 *   klnmf_query_f64: KLNMF_QF_MON_STAT the largest statistic of the loop (estimated relative error of a numerator entry),
 *                    KLNMF_QF_MON_THRESHOLD the threshold it is compared with */
#define KLNMF_Q_MON_CHECKS        10
#define KLNMF_Q_MON_TRIPS         11
#define KLNMF_Q_MON_GAVE_UP       12
/*   KLNMF_Q_FP8_POLL_DUE     loops sequenced in pieces on row shards (klnmf_iter_*): 1 if the klnmf_iter_advance that follows the
 *                            column pass just enqueued reads the all-reduced count in loss[1] (monitor trips + unfixable ratio entries,
 *                            written by that column pass's last launch).  The caller then all-reduces loss[1] BEHIND
 *                            klnmf_iter_colpass and before klnmf_iter_advance; loss[0] alone may be exchanged earlier, while the
 *                            column pass computes.  The loop state behind the answer is the same on every rank. */
#define KLNMF_Q_FP8_POLL_DUE      13
int klnmf_query(klnmf_ctx *ctx, int what, int64_t *value);
/*   KLNMF_QF_SUM_V  the sum of the uploaded V as stored (16-bit modes; 0 in the exact modes), in the data's own units */
#define KLNMF_QF_SUM_V            0
/*   KLNMF_QF_NNZ_V  how many entries of the uploaded V are > 0 as stored (16-bit modes) */
#define KLNMF_QF_NNZ_V            1
#define KLNMF_QF_MON_STAT         2
#define KLNMF_QF_MON_THRESHOLD    3
/*   KLNMF_QF_MON_PART0 + 0 / 1 / 2: the statistic's ingredients, largest of the loop each (diagnostics: uncentred bias, noise
 *   term scaled to the full row count, |common factor of a component row over the monitored tile|) */
#define KLNMF_QF_MON_PART0        4
/*   KLNMF_QF_MON_SPREAD the smallest relative spread std(q) / mean(q) of a monitored column's ratios in the loop (the e4m3
 *   rounding only averages out over the rows if the ratios are spread over its cells), KLNMF_QF_MON_MIN_SPREAD its threshold */
#define KLNMF_QF_MON_SPREAD       7
#define KLNMF_QF_MON_MIN_SPREAD   8
/*   KLNMF_QF_KL_OVER_SUM_V  the last loop's final recorded loss over the sum of V (all shards), 16-bit modes; -1: none.  Below
 *   about 2e-3 the f16 operands' own rounding noise can exceed 1e-4 of the loss (nmf.py:214 has no
 *   counterpart: the reference has one arithmetic) -- the host layer says so once on stderr */
#define KLNMF_QF_KL_OVER_SUM_V    9
int klnmf_query_f64(klnmf_ctx *ctx, int what, double *value);

/* ---- measurement -------------------------------------------------------- */
/* When enabled, row-pass / column-pass launches are bracketed by HIP events on the context's stream: on = 1 every
 * iteration's, on = N > 1 those of every N-th iteration of a loop (an event record is a stream packet of its own, about
 * 5 us of dispatch gap: four per fit iteration are 3.5 % of a 0.65 ms iteration, so a measurement that must not
 * disturb what it times samples).  klnmf_profile_read returns the bracketed launches' count and summed milliseconds
 * since the last reset (synchronises). */
int klnmf_profile_enable(klnmf_ctx *ctx, int on);
int klnmf_profile_read(klnmf_ctx *ctx, int64_t *rowpass_launches,
                       double *rowpass_ms, int64_t *colpass_launches,
                       double *colpass_ms, int reset);
/* The part of the row-pass time klnmf_profile_read reports that the column-split last partial round of workgroups and its
 * slab W rule took (hybrid update pass: DESIGN.md section 4.1), the launches it was measured over and the (padded) rows
 * that part covers; 0 launches when the context runs whole rows everywhere. */
int klnmf_profile_read_tail(klnmf_ctx *ctx, int64_t *tail_n, double *tail_ms, int64_t *tail_rows, int reset);
int klnmf_synchronize(klnmf_ctx *ctx);

/* ---- CSR input (next-row N3): the reference's sparse branch ----------------------------------- */
/* With scipy-sparse X the reference evaluates W.H and the ratio only on the stored entries of X
 * (nmf.py:52-70, 301-308, 331-334).  Exact modes only.  klnmf_set_problem_sparse replaces
 * klnmf_set_problem; klnmf_upload_csr takes X in CSR (indptr[n+1], indices[nnz], data[nnz]; no explicit
 * zeros, `eliminate_zeros` of nmf.py:66) and the same entries in CSC order (csc_indptr[f+1], csc_rows[nnz],
 * csc_perm[nnz] = position of the entry in the CSR arrays).  Every loop / step / error entry point then
 * works on the stored entries; klnmf_get_Q_values returns the ratio in CSR order (the data of the sparse
 * matrix `_Q` returns, nmf.py:333-334). */
int klnmf_set_problem_sparse(klnmf_ctx *ctx, int64_t n, int64_t f, int64_t k, int64_t max_iter_capacity, int64_t nnz);
int klnmf_upload_csr(klnmf_ctx *ctx, int dtype, const int64_t *indptr, const int64_t *indices, const void *data,
                     const int64_t *csc_indptr, const int64_t *csc_rows, const int64_t *csc_perm);
int klnmf_get_Q_values(klnmf_ctx *ctx, void *dst, int dtype);

/* ---- reconstruction (next-row K7) ---------------------------------------- */
/* C[m x n] = A[m x kk] . B[kk x n], row-major host arrays of `dtype`, computed on `device` in that
 * arithmetic.  Replaces `internal.dot(self.get_dico(dest_mod))` / `internal.dot(self.get_stacked_dicos(..))`
 * of MultimodalLearner.reconstruct_modality / reconstruct_modalities (learner.py:80-84). */
int klnmf_matmul(int device, int dtype, int64_t m, int64_t n, int64_t kk,
                 const void *A, const void *B, void *C);

/* ---- nearest-neighbour evaluation (next-row N4) ------------------------------------------------- */
/* out[na x nb] (row-major) = measure(A[i, :], B[j, :]) for all pairs; A [na x d], B [nb x d] row-major host
 * arrays of `dtype`.  Replaces `all_distances(reco_data, ex_data, measure)` (evaluation.py:103-106: a
 * [na,1,d] x [1,nb,d] broadcast of one of the measures of metrics.py:58-86).  metric: 0 kl_div(a, b),
 * 1 rev_kl_div, 2 sym_kl_div, 3 frobenius, 4 cosine_diff; eps = 1e-8 (metrics.py:15). */
#define KLNMF_DIST_KL 0
#define KLNMF_DIST_REV_KL 1
#define KLNMF_DIST_SYM_KL 2
#define KLNMF_DIST_FROBENIUS 3
#define KLNMF_DIST_COSINE_DIFF 4
int klnmf_all_distances(int device, int dtype, int metric, int64_t na, int64_t nb, int64_t d,
                        const void *A, const void *B, void *out);

/* ---- hardware probes (tests) -------------------------------------------- */
/* Runs the MFMA / LDS-transpose layout self-checks the fused kernels rely on;
 * *failed = bitmask of failing probes (0 = all good). */
int klnmf_selftest(int device, int *failed);

#ifdef __cplusplus
}
#endif
#endif /* KLNMF_H */
