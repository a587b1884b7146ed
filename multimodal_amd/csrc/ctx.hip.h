// Host side of libklnmf.so, shared by its translation units (include/klnmf.h has the contract and the reference interfaces each
// entry point replaces):
//   api_context.hip  contexts, problems, uploads and downloads (K6: learner.py:53-56 stack_data; nmf.py:147-157 _init)
//   api_loop.hip     the loop of nmf.py:212-222 and its pieces: launch sequencing, stop rule, fp8 regime and its monitor
//   api_comm.hip     row shards over the GPUs of a node: the RCCL communicator, the agreed loop entry, the exchange (nmf.py:349)
//   api_eval.hip     evaluation and introspection: reconstruction products (learner.py:80-84), distances, queries, profiling
// This header: error handling, the device block cache, the development switches and the context itself.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>       // types and enums only: the library itself is opened at run time (no link-time dependency)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/klnmf.h"
#include "common.hip.h"
#include "exact.hip.h"
#include "sparseb.hip.h"
#include "mfma.hip.h"
#include "mfma4.hip.h"
// the k_rowpass4 instantiations live in rowpass4_inst_{1,2,3}.hip (built in parallel); here they are only declared
#include "rowpass4_list.hip.h"
namespace klnmf {
KL_RP4_LIST_1(KL_RP4_DECLARE) KL_RP4_LIST_2(KL_RP4_DECLARE) KL_RP4_LIST_3(KL_RP4_DECLARE)
}  // namespace klnmf
#include "colq.hip.h"
#include "colq8x.hip.h"
#include "post.hip.h"

using namespace klnmf;

namespace klnmf_host {

extern thread_local std::string g_err;      // (api_eval.hip: klnmf_last_error)

struct ApiError {
    int code;
    std::string msg;
};

[[noreturn]] inline void fail(int code, const std::string &m) { throw ApiError{code, m}; }

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            fail(e_ == hipErrorOutOfMemory ? KLNMF_ERR_ALLOC : KLNMF_ERR_HIP,              \
                 std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return KLNMF_OK;
    } catch (const ApiError &e) {
        g_err = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        g_err = e.what();
        return KLNMF_ERR_HIP;
    } catch (...) {
        g_err = "unknown error";
        return KLNMF_ERR_HIP;
    }
}

inline int grid_for(int64_t count, int block = 256, int cap = 4096) {
    int64_t g = (count + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

constexpr int kW8Blocks = 1024;             // conversion kernel's grid: per-block column maxima [kW8Blocks][KP]

struct EventPair {
    hipEvent_t a, b;
};

// RCCL entry points, resolved on first use: a process that never shards needs no librccl.
struct RcclApi {
    void *lib = nullptr;
    std::string err;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
};
RcclApi &rccl();              // (api_comm.hip: resolved on first use)
#define RCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) fail(KLNMF_ERR_RCCL, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace klnmf_host
using namespace klnmf_host;

// Device blocks of destroyed / re-shaped contexts are kept for the next one (per process, per device, by size class).
// At the reference's data sizes a fit is a few milliseconds of kernels, and ~25 hipMalloc + hipFree per context cost as
// much again (experiment.py and samples/launcher.py run many small fits and transforms in sequence).  Blocks are handed
// back only after the owning stream has been synchronised (klnmf_destroy, klnmf_set_problem), and every block is
// zero-filled on hand-out as a fresh one is.  KLNMF_ALLOC_CACHE_MB (default 1024; 0 = off) bounds what is kept;
// blocks above 64 MiB are never kept.
struct DevBlockCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_blocks;
    size_t held = 0;
    static size_t limit() {
        static const size_t v = [] {
            const char *e = std::getenv("KLNMF_ALLOC_CACHE_MB");
            return (size_t)(e ? std::max(0, std::atoi(e)) : 1024) << 20;
        }();
        return v;
    }
    static size_t size_class(size_t bytes) {          // next power of two up to 1 MiB, then multiples of 1 MiB
        if (bytes <= 256) return 256;
        if (bytes <= ((size_t)1 << 20)) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
        return (bytes + ((size_t)1 << 20) - 1) >> 20 << 20;
    }
    void *take(int device, size_t cls) {
        std::lock_guard<std::mutex> g(mu);
        auto it = free_blocks.find({device, cls});
        if (it == free_blocks.end() || it->second.empty()) return nullptr;
        void *p = it->second.back();
        it->second.pop_back();
        held -= cls;
        return p;
    }
    void flush(int device) {
        std::lock_guard<std::mutex> g(mu);
        for (auto &kv : free_blocks) {
            if (kv.first.first != device) continue;
            for (void *p : kv.second) { (void)hipFree(p); held -= kv.first.second; }
            kv.second.clear();
        }
    }
    bool give(int device, size_t cls, void *p) {
        if (cls > ((size_t)64 << 20)) return false;
        std::lock_guard<std::mutex> g(mu);
        if (held + cls > limit()) return false;
        free_blocks[{device, cls}].push_back(p);
        held += cls;
        return true;
    }
};
extern DevBlockCache g_block_cache;      // (api_context.hip)

// Development switches: what only measurements and tests need.  Read in ONE place (here), afresh at every klnmf_set_problem and
// loop entry, and honoured only under KLNMF_DEV=1 -- a production process cannot change the library's arithmetic by accident.
// (User-facing environment: KLNMF_QTILE=16 -- never fp8 ratio tiles -- and KLNMF_ALLOC_CACHE_MB, plus the host layer's
// KLNMF_PRECISION / KLNMF_DEVICE / KLNMF_LIB / KLNMF_NO_POOL: INTEGRATION.md section 1.)
struct DevSwitches {
    int qtile = 0;              // KLNMF_QTILE = 8 / 16: fp8 ratio tiles forced on (from a loop's third iteration) / off      [16: also without KLNMF_DEV]
    int col8 = -1;              // KLNMF_COL8 = 0: no fp8 x fp8 column pass; 1: at any size
    int ne = -1;                // KLNMF_NE = 0 / 1: the update pass without the numerator's eps never / in every fp8 loop
    bool q8_fixup = true;       // KLNMF_Q8_FIXUP=0: no exact correction of large ratio entries (the tests' control run)
    bool q8_monitor = true;     // KLNMF_Q8_MONITOR=0: no monitor
    float mon_threshold = 0.f, mon_min_spread = -1.f;      // KLNMF_MON_THRESHOLD / KLNMF_MON_MIN_SPREAD: the monitor's two thresholds (calibration runs)
    bool ratio_scale = true;    // KLNMF_RATIO_SCALE=0: no ratio scale of the first update
    bool eps_pad = true;        // KLNMF_NO_EPS_PAD=1: eps added in the epilogue instead of riding through MFMA-1
    int row_split = -1;         // KLNMF_ROW_SPLIT = 0 / N: column-split update pass off / N chunks
    int row_tail = -1;          // KLNMF_ROW_TAIL = 0: no column-split last partial round
    int comm_parts = 1;         // KLNMF_COMM_PARTS = P: the numerator in P column parts on a communicator (experimental: one-rank runs only)
    bool comm_overlap = true;   // KLNMF_COMM_OVERLAP=0: the parts' all-reduces on the loop's own stream
    bool comm_single = false;   // KLNMF_COMM_SINGLE=1: a one-rank communicator takes the collective path (tests)
    int sp_cb = 0, sp_rb = 0;   // KLNMF_SP_CB / KLNMF_SP_RB: column / row blocks of the CSR kernels (sparseb.hip.h; 0: by the L2's size)
    int graph = 0;              // KLNMF_GRAPH=1: two iterations captured into a hipGraph and replayed (measured: no gain)
    static DevSwitches read() {
        DevSwitches d;
        auto num = [](const char *name, int dflt) { const char *e = std::getenv(name); return e ? std::atoi(e) : dflt; };
        if (num("KLNMF_QTILE", 0) == 16) d.qtile = 16;
        if (num("KLNMF_DEV", 0) == 0) return d;
        d.qtile = num("KLNMF_QTILE", 0);
        d.col8 = num("KLNMF_COL8", -1);
        d.ne = num("KLNMF_NE", -1);
        d.q8_fixup = num("KLNMF_Q8_FIXUP", 1) != 0;
        d.q8_monitor = num("KLNMF_Q8_MONITOR", 1) != 0;
        if (const char *e = std::getenv("KLNMF_MON_THRESHOLD")) d.mon_threshold = (float)std::atof(e);
        if (const char *e = std::getenv("KLNMF_MON_MIN_SPREAD")) d.mon_min_spread = (float)std::atof(e);
        d.ratio_scale = num("KLNMF_RATIO_SCALE", 1) != 0;
        d.eps_pad = num("KLNMF_NO_EPS_PAD", 0) == 0;
        d.row_split = num("KLNMF_ROW_SPLIT", -1);
        d.row_tail = num("KLNMF_ROW_TAIL", -1);
        d.comm_parts = num("KLNMF_COMM_PARTS", 1);
        d.comm_overlap = num("KLNMF_COMM_OVERLAP", 1) != 0;
        d.comm_single = num("KLNMF_COMM_SINGLE", 0) != 0;
        d.graph = num("KLNMF_GRAPH", 0);
        d.sp_cb = num("KLNMF_SP_CB", 0);
        d.sp_rb = num("KLNMF_SP_RB", 0);
        return d;
    }
};

struct klnmf_ctx {
    DevSwitches sw;
    int device = 0;
    int prec = KLNMF_PREC_F64;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 256;

    int64_t n = 0, f = 0, k = 0, cap = 0;
    bool have_problem = false;
    int cur = 0;          // index of the current W buffer
    int loop_start_cur = 0;
    int64_t loop_iters = 0;

    // common device state
    DevState *st = nullptr;
    double *errors = nullptr;
    double *loss_xchg = nullptr;
    double2 *loss_red = nullptr;              // [kLossRedMax] pairs of the loss slices (k_slab_sum -> k_post; 16-bit modes)
    std::vector<std::pair<void *, size_t>> allocs;      // (block, size class)

    // exact modes (T = double or float)
    void *V = nullptr, *W[2] = {nullptr, nullptr}, *H = nullptr, *Q = nullptr;
    void *Npart = nullptr, *numer = nullptr;
    double *loss_part = nullptr;
    int64_t loss_part_count = 0;
    int nsplit = 1, kchunk = 0;
    int wsplit = 1, wchunk = 0;       // exact modes: feature chunks of the W rule's contraction (few rows), slabs in Wpart
    int q_tt = 4, w_tt = 4, n_tt = 4; // exact modes: outputs per thread and axis of the three GEMMs (k_gemm: 4 = 64 x 64 tiles, 8 = 128 x 128)
    void *Wpart = nullptr;

    // CSR input in the exact modes (sparse.hip.h): structure of X in CSR and CSC order, ratio values, H^T
    bool sparse = false;
    int64_t nnz = 0;
    int64_t *sp_indptr = nullptr, *sp_indices = nullptr, *csc_indptr = nullptr, *csc_rows = nullptr, *csc_perm = nullptr;
    void *sp_data = nullptr, *sp_q = nullptr, *HT = nullptr;
    double *sp_row_loss = nullptr, *sp_wpart = nullptr, *sp_prod = nullptr;
    int64_t sp_nblk = 0;
    // ... blocked for the L2 (sparseb.hip.h; k <= 512): int32 copies of the indices, column blocks of the CSR order and row blocks
    // of the CSC order with their pointers, the slabs of partial sums
    bool sp_blocked = false;
    int sp_cb = 1, sp_rb = 1;                 // column blocks / row blocks
    int64_t sp_cb_cols = 0, sp_rb_rows = 0;
    int *sp_idx32 = nullptr, *csc_rows32 = nullptr, *csc_perm32 = nullptr;
    int64_t *sp_blkptr = nullptr, *csc_blkptr = nullptr;      // [n][cb + 1], [f][rb + 1]
    double *sp_loss_part = nullptr;           // [cb][n]
    void *sp_G = nullptr, *sp_NT = nullptr;   // [cb][n][k], [rb][f][k]
    int *sp_bad = nullptr;
    double *hpart = nullptr;          // exact modes, long rows: [k][hseg_n] partial row sums of the H rule / of the CSR loss term
    int hseg_n = 1; int64_t hseg = 0; // segments per dictionary row and their length (1: the one-block-per-row kernels)

    // bf16 modes
    int KT = 0, KP = 0, ks = 0;
    int64_t n_pad = 0, f_pad = 0, w_rows = 0;
    int nrt = 0, nct = 0, nct_used = 0, nst = 0, ncb = 0, nchunks = 0, stages_per_chunk = 0;
    int row_chunks = 1, row_ct_chunk = 0;     // column-split update pass (few rows): chunks, column tiles per chunk
    float *Gpart = nullptr;                   // [row_chunks][nrt * 32][KP] partial Q.H^T
    // hybrid update pass (many rows): the workgroups of the last partial round run column-split (tail_chunks chunks of
    // tail_ct_chunk column tiles each) so that they fill the chip; tail_wg = 0: none.  Gpart then holds the tail's slabs.
    int tail_wg = 0, tail_chunks = 1, tail_ct_chunk = 0;
    unsigned char *W8 = nullptr;              // e4m3 image of W_new for the fp8 x fp8 column pass (colq8x.hip.h; KLNMF_COL8=0: off)
    float *w8s = nullptr;                     // [KP] power-of-two scales of the e4m3 image
    bool w8_meas = false;                     // the maxima table holds a measurement of this loop
    int64_t loss_parts() const {               // entries of loss_part2 an update pass writes
        if (tail_wg > 0) return (int64_t)nrt + (int64_t)(tail_chunks - 1) * (nrt - tail_rt0());
        return (int64_t)nrt * row_chunks;
    }
    int tail_rt0() const { return (((nrt + 7) / 8) - tail_wg) * 8; }
    void *VtA = nullptr;          // V as 32 x 32 fp16 tiles in the row pass's accumulator order (k_tile_V)
    unsigned char *Qt = nullptr;  // ratio tiles the row pass leaves for the column pass
    // fp8 ratio tiles (1 B per element of V instead of the 16-bit operands) for the H rule.  q8_ok: the problem's shape
    // allows them (klnmf_set_problem); q8_loop: this loop's data do (decided at the loop's entry); they are used from the
    // loop's third iteration on (the first updates from W0 = V.H0^T can carry ratios far beyond fp8's range).
    bool q8_ok = false, q8_loop = false;
    int64_t iter_in_loop = 0;
    // what the last loop actually ran (klnmf_query): iterations whose ratio tiles were fp8, whose column pass was fp8 x fp8
    int64_t stat_q8_tiles = 0, stat_col8 = 0;
    uint2 *q8_list = nullptr;                 // [kQ8ListCap] saturated ratio entries of the current iteration (colq.hip.h)
    bool ne_ok = false;                       // the problem's shape has NE kernels (fp16 V, k <= 224, enough rows for fp8 ratio tiles)
    bool ne_loop = false;                     // this loop's fp8-tile update passes drop the numerator's eps (NE kernels; begin_fp8_loop)
    bool last_row_ne = false;                 // ... and the update pass just launched was one of them (its loss needs DevState.corr_eps)
    bool in_capture = false;                  // a hipGraph capture is recording this context's launches (no synchronising polls)
    // the saturation counters of the last loop as its end found them (DevState is reset by the next entry point)
    int64_t stat_w8_sat = 0, stat_w8_fallbacks = 0, stat_q8_sat = 0, stat_q8_unfixed = 0;
    // ---- the fp8 monitor (monitor.hip.h): partial sums of the monitored iteration; what k_post is to do with them; the last
    // loop's record (klnmf_query / klnmf_query_f64)
    float *mon_part = nullptr, *mon_spread = nullptr;
    bool mon_pending = false;                 // this iteration's first summing launch turns the partial sums into the statistic
    bool mon_dry_pending = false;             // ... and it was the dry run of the loop's second iteration: poll before the third
    int mon_ncols = 0; float mon_noise_scale = 0.f;
    int64_t mon_checks = 0, stat_mon_checks = 0, stat_mon_trips = 0;
    double stat_mon_max = 0.0, stat_mon_dbg[3] = {0, 0, 0}, stat_mon_spread = 1.0;
    bool stat_mon_gave_up = false;
    // polls of the monitor's verdict (poll_fp8_overflow).  The dry run's poll synchronises (its answer decides the NEXT iteration's
    // kernels); every later one is deferred: the counts are copied to pinned host memory behind the monitored iteration, an event
    // marks the copy, and the answer is read one iteration later, when the event has long passed -- the stream never drains
    // (a synchronising poll cost a pipeline bubble plus a pageable read-back per check: 5 checks in bench.py's 40 timed iterations)
    void *poll_host = nullptr;                // pinned: a DevState or the two doubles of the loss exchange
    hipEvent_t poll_ev = nullptr;
    bool poll_inflight = false, poll_agreed = false;
    // the 16-bit mode's data condition (below KL / sum(V) of about 2e-3 the f16 operands' own noise can pass 1e-4
    // of the loss): sum of V over ALL shards as the loop's entry was given it (stored units; < 0: this context's own), and the
    // last loop's final KL / sum(V) (klnmf_query_f64 KLNMF_QF_KL_OVER_SUM_V; < 0: no loop yet / exact mode)
    double loop_sum_x_all = -1.0, stat_kl_over_sumv = -1.0;
    // the refusal counters of DevState (v_overflow, op_range) change only on uploads and image measurements: they are read
    // back (one copy + synchronisation) only when one of those happened since the last check
    bool refusals_dirty = true;
    // single-context fit loops: the loss reduction + stop decision of an iteration ride in the slab-sum launch behind the
    // column pass (k_sum_partials_f32) instead of a launch of their own behind the row pass; set by piece_rowpass,
    // consumed by the next fast_colpass.  KLNMF_LOSS_DEFER=0: off.
    LossArgs pending_loss{nullptr, 0, 0.0, nullptr, 0, nullptr, 0.0, nullptr, 0};
    double v_max = 0.0;          // the maximum announced with klnmf_set_v_max (0: none)
    // fp8 ratio tiles in this iteration?  k > 256 (FUSED row pass, KSPLIT = 2 column pass) has only the fp8 x fp8 column pass
    // for them: there the W image's scales must have been measured (the loop's second iteration does that)
    bool q8() const { return q8_loop && iter_in_loop >= 2 && (!big || (W8 != nullptr && w8_meas)); }
    float *W32[2] = {nullptr, nullptr};
    opnd_t *Wb[2] = {nullptr, nullptr};
    float *H32 = nullptr;
    // ---- one launch behind the column pass (post.hip.h) ----
    float *H32alt = nullptr;                  // the dictionary master is ping-pong there: k_post reads H32, writes H32alt, then they swap
    int64_t loop_hswaps = 0;                  // H rules enqueued since the loop's entry (how many the device executed: n_done -- fetch_results)
    float *loop_h0 = nullptr, *loop_h1 = nullptr;      // H32 / H32alt as the loop found them
    unsigned *w8tab = nullptr;                // [kW8TabRows][KP] maxima of the conversion kernel (k_post: -> w8s_next, emptied)
    float *w8s_next = nullptr;                // [KP] scales of the NEXT image: k_post writes them, then w8s / w8s_next swap
    bool conv_ran = false;                    // this iteration's conversion ran: k_post derives the next scales

    // Column parts of the H numerator.  `whole`: all columns as one part (layout [KP][f_pad], what every single-context loop
    // and the exchange API use).  `parts[0 .. nparts_cfg)`: the split layout of loops on a communicator -- part p = a range of
    // column blocks with its own slabs [nchunks][KP][ld] and numerator [KP][ld] (contiguous: one ncclAllReduce each), so that
    // the all-reduce of part p overlaps the column pass of part p + 1 (KLNMF_COMM_PARTS, default 1 = no split)
    struct PartCfg { int cb0, ncb, ct0, nct, col0, ncols, ld, nchunks, spc; int64_t numer_off, slab_off; };
    PartCfg whole{}, parts[kPostMaxParts]{};
    int nparts_cfg = 1;
    bool piece_split = false, piece_use8 = false;      // loop in pieces: the numerator was produced in parts (klnmf_iter_colpass_part)
    hipStream_t comm_stream = nullptr;        // all-reduces of the parts before the last one (overlap)
    hipEvent_t ev_part[kPostMaxParts] = {}, ev_ar[kPostMaxParts] = {};
    opnd_t *Ht4 = nullptr;
    int kc = -1;                 // eps-carrying pad component of the ping-pong path (k_update_pack_H), -1: none
    int kc_shape = -1;           // ... as the shape allows it; kc = kc_shape only while the carrier pair fits fp16 (choose_eps_carrier)
    double *hsum = nullptr;
    // per-component power-of-two scales of the fp16 operand images (mfma.hip.h, opnd_t), [KP] each: of the current images;
    // hs-based (from the dictionary's row sums); the constant 2^-13 of a row-normalised dictionary; measured from a W
    float *tcur = nullptr, *t_hs = nullptr, *t_unit = nullptr;
    unsigned *wmax = nullptr;
    bool images_measured = false;    // the current images carry measured scales: valid for one update (see opnd_t)
    unsigned sr_launches = 0;        // row pass launches of the current loop (the seeds of the tiles' stochastic rounding)
    bool w_is_init = false;          // the current W is W0 = V.H0^T of klnmf_init_W, untouched since: a dictionary set NOW still meets
                                     // ratios of about f / k on its first update (the ratio scale of k_ratio_scale must stay on)
    float *NpartF = nullptr, *numerF = nullptr;
    double2 *loss_part2 = nullptr;

    // profiling
    double ratio_eps = kEpsRatio;   // only the step API honours a non-default value
    double v_scale = 1.0;           // storage factor c of the 16-bit V (power of two)
    bool v_uploaded = false;

    bool profiling = false;
    // every `profile_every`-th iteration of a loop has its row-pass / column-pass launches bracketed by HIP events (1: every one).
    // An event record is a packet of its own in the stream: about 5-6 us of dispatch gap each -- four of them per fit iteration
    // were 3.5 % of one rank's 0.65 ms shard iteration (profiles/r06_timelines_shard.txt), so bench.py samples every 4th
    int profile_every = 1;
    int64_t profile_seq = 0;
    bool prof_now = false;           // this iteration's launches are bracketed (set by piece_rowpass)
    std::vector<EventPair> ev_row, ev_col, ev_tail;      // ev_tail: the column-split tail + slabs part of a hybrid row pass

    // row shards over the GPUs of a node (klnmf_comm_*, klnmf_run_sharded): this rank's RCCL communicator
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    double *comm_scratch = nullptr;       // 8 doubles on the device, owned by the communicator (not by a problem)

    bool is_exact() const { return prec == KLNMF_PREC_F64 || prec == KLNMF_PREC_F32; }
    // fp8 ratio tiles from how many rows per context?  Their e4m3 rounding only enters the H numerator, a sum over all rows
    // (relative error ~ 0.036 sqrt(2 / n)); measured against the fp64 oracle (scripts/fp8_rows_survey.py,
    // profiles/r03_fp8_rows_survey.txt): final-KL deviation 1.7e-5 .. 3.5e-5 from 4096 to 50 000 rows at k = 50, 6.7e-6 .. 1.5e-5
    // at k = 200 -- a floor that does not depend on n, a fifth of the 1e-4 budget.  k <= 224: from 32 769 rows, where the
    // column-split update pass of small problems no longer runs (round 2: 65 536; C2 = 50 000 rows now qualifies).
    // 256 < k <= 512: 65 536, the size fixture G14 pins.
    static bool row_chunks_possible_q8(int64_t n, bool big_k) { return big_k ? n >= 65536 : n > 32768; }
    // ping-pong row pass (mfma4.hip.h): fp16-stored V; 8-wave workgroups for KT <= 7, 4-wave ones for 10 <= KT <= 16 (even)
    // big: 224 < k <= 512 (KT = 8 .. 16, even): 4-wave workgroups, FUSED order, component-split column passes
    bool big = false;
    size_t esize() const { return prec == KLNMF_PREC_F64 ? 8 : 4; }

    void *dalloc(size_t bytes, bool zero = true) {
        if (bytes == 0) bytes = 16;
        const size_t cls = DevBlockCache::size_class(bytes);
        void *p = g_block_cache.take(device, cls);
        if (!p) {
            hipError_t e = hipMalloc(&p, cls);
            if (e == hipErrorOutOfMemory) {        // the cache may be what is in the way: give its blocks back and retry once
                (void)hipGetLastError();
                g_block_cache.flush(device);
                e = hipMalloc(&p, cls);
            }
            HIPCHK(e);
        }
        allocs.push_back({p, cls});
        if (zero) HIPCHK(hipMemsetAsync(p, 0, bytes, stream));
        return p;
    }
    void free_all() {      // callers have synchronised the stream: no kernel of this context still touches the blocks
        for (auto &b : allocs)
            if (!g_block_cache.give(device, b.second, b.first)) (void)hipFree(b.first);
        allocs.clear();
        for (auto &e : ev_row) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (auto &e : ev_col) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (auto &e : ev_tail) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        ev_tail.clear();
        ev_row.clear();
        ev_col.clear();
        have_problem = false;
    }
};

namespace klnmf_host {

inline void use(klnmf_ctx *c) {
    if (!c) fail(KLNMF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
}
inline void comm_release(klnmf_ctx *c) {
    if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    for (int p = 0; p < kPostMaxParts; ++p) {
        if (c->ev_part[p]) { (void)hipEventDestroy(c->ev_part[p]); c->ev_part[p] = nullptr; }
        if (c->ev_ar[p]) { (void)hipEventDestroy(c->ev_ar[p]); c->ev_ar[p] = nullptr; }
    }
    if (c->comm_scratch) { (void)hipFree(c->comm_scratch); c->comm_scratch = nullptr; }
    c->comm_rank = 0; c->comm_size = 1;
}
inline void need_problem(klnmf_ctx *c) {
    use(c);
    if (!c->have_problem) fail(KLNMF_ERR_ARG, "klnmf_set_problem has not been called");
}

inline EventPair begin_event(klnmf_ctx *c, std::vector<EventPair> &v) {
    EventPair e{};
    HIPCHK(hipEventCreate(&e.a));
    HIPCHK(hipEventCreate(&e.b));
    HIPCHK(hipEventRecord(e.a, c->stream));
    v.push_back(e);
    return e;
}

// ---- what the units call across each other (default arguments live here) ---------------------------------------------------------
enum PostMode { POST_FULL = 0, POST_SUM = 1, POST_RULE = 2 };
static const LossArgs kNoLoss{nullptr, 0, 0.0, nullptr, 0, nullptr, 0.0, nullptr, 0, 0};
struct Refusals { int v_overflow = 0, op_range = 0; };
// api_context.hip
void reset_state(klnmf_ctx *c);
void fast_pack_H(klnmf_ctx *c, const unsigned *wmax = nullptr);
void measure_and_pack(klnmf_ctx *c, bool from_init = false);
size_t dt_size(int dtype);
void *stage_to_device(klnmf_ctx *c, const void *src, int dtype, int64_t count);
// api_loop.hip
void piece_rowpass(klnmf_ctx *c, int fit, const double *fused_tol = nullptr, bool defer_to_post = false);
void piece_decide(klnmf_ctx *c, double tol_abs);
void piece_colpass(klnmf_ctx *c);
void piece_update_H(klnmf_ctx *c);
void piece_fit_tail(klnmf_ctx *c);
void fetch_results(klnmf_ctx *c, double *errors_out, int64_t *n_done, int *stopped);
void poll_fp8_overflow(klnmf_ctx *c, bool agreed = false);
bool fp8_poll_due(const klnmf_ctx *c);
bool fused_w8_stage(klnmf_ctx *c);
void launch_monitor(klnmf_ctx *c, bool use8);
void fused_colpass_part(klnmf_ctx *c, const klnmf_ctx::PartCfg &p, bool use8);
void launch_post(klnmf_ctx *c, PostMode mode, const klnmf_ctx::PartCfg *parts, int nparts, const LossArgs &la, bool decide, bool use8,
                 bool last_sum);
Refusals read_refusals(klnmf_ctx *c);
void raise_refusals(klnmf_ctx *c, const Refusals &r);
void check_v_overflow(klnmf_ctx *c);
void begin_fp8_loop(klnmf_ctx *c, double sum_x_global = -1.0, double cells_global = -1.0, double nnz_global = -1.0, int ok_all = -1);
// api_comm.hip
bool comm_multi(const klnmf_ctx *c);
void comm_loop_entry(klnmf_ctx *c);
void comm_iteration(klnmf_ctx *c, int fit, double tol_abs);

}  // namespace klnmf_host
