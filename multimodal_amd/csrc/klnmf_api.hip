// C-ABI of the MI355X-native KL-NMF path (see include/klnmf.h for the contract
// and the reference interfaces each entry point replaces).
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
#include "probe.hip.h"

using namespace klnmf;

namespace {

thread_local std::string g_err;

struct ApiError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const std::string &m) { throw ApiError{code, m}; }

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

int grid_for(int64_t count, int block = 256, int cap = 4096) {
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
RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) { a.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return a; }
#define KL_RCCL_SYM(field, sym)                                                        \
        a.field = (decltype(a.field))dlsym(a.lib, sym);                                \
        if (!a.field && a.err.empty()) a.err = std::string("librccl lacks ") + sym;
        KL_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") KL_RCCL_SYM(CommInitRank, "ncclCommInitRank")
        KL_RCCL_SYM(CommDestroy, "ncclCommDestroy") KL_RCCL_SYM(AllReduce, "ncclAllReduce")
        KL_RCCL_SYM(GroupStart, "ncclGroupStart") KL_RCCL_SYM(GroupEnd, "ncclGroupEnd")
        KL_RCCL_SYM(GetErrorString, "ncclGetErrorString") KL_RCCL_SYM(CommCount, "ncclCommCount")
#undef KL_RCCL_SYM
        return a;
    }();
    if (!api.err.empty()) fail(KLNMF_ERR_RCCL, api.err);
    return api;
}
#define RCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) fail(KLNMF_ERR_RCCL, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace

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
static DevBlockCache g_block_cache;

// Development switches: what only measurements and tests need.  Read in ONE place (here), afresh at every klnmf_set_problem and
// loop entry, and honoured only under KLNMF_DEV=1 -- a production process cannot change the library's arithmetic by accident.
// (User-facing environment: KLNMF_QTILE=16 -- never fp8 ratio tiles -- and KLNMF_ALLOC_CACHE_MB, plus the host layer's
// KLNMF_PRECISION / KLNMF_DEVICE / KLNMF_LIB / KLNMF_NO_POOL: INTEGRATION.md section 1.)
struct DevSwitches {
    int qtile = 0;              // KLNMF_QTILE = 8 / 16: fp8 ratio tiles forced on (from a loop's third iteration) / off      [16: also without KLNMF_DEV]
    int col8 = -1;              // KLNMF_COL8 = 0: no fp8 x fp8 column pass; 1: at any size; 2: the W rule writes the e4m3 image itself
    int ne = -1;                // KLNMF_NE = 0 / 1: the update pass without the numerator's eps never / in every fp8 loop
    bool q8_fixup = true;       // KLNMF_Q8_FIXUP=0: no exact correction of large ratio entries (the tests' control run)
    bool q8_rules_r4 = false;   // KLNMF_Q8_RULES=1: round 4's data rules at the loop's entry as well as the in-loop monitor (A/B runs)
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
        d.q8_rules_r4 = num("KLNMF_Q8_RULES", 0) != 0;
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
    bool w8_tail = false;                     // KLNMF_COL8=2: the W rule writes the e4m3 image itself (whole-row launch); the conversion
                                              // kernel then only covers the rows of the column-split last partial round
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
    int64_t loop_planned = 0;                 // iterations this loop may run (klnmf_run: max_iter; loops in pieces: the capacity of
                                              // klnmf_set_problem): the monitor's threshold depends on it (monitor.hip.h)
    double stat_mon_max = 0.0, stat_mon_dbg[3] = {0, 0, 0}, stat_mon_spread = 1.0;
    bool stat_mon_gave_up = false;
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
    bool tail_use8 = false;                   // KLNMF_COL8=2 on the fused tail: the image the W rule just wrote carries measured scales
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
    float *NpartF = nullptr, *numerF = nullptr;
    double2 *loss_part2 = nullptr;

    // profiling
    double ratio_eps = kEpsRatio;   // only the step API honours a non-default value
    double v_scale = 1.0;           // storage factor c of the 16-bit V (power of two)
    bool v_uploaded = false;

    bool profiling = false;
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

namespace {

void use(klnmf_ctx *c) {
    if (!c) fail(KLNMF_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
}
void comm_release(klnmf_ctx *c) {
    if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    for (int p = 0; p < kPostMaxParts; ++p) {
        if (c->ev_part[p]) { (void)hipEventDestroy(c->ev_part[p]); c->ev_part[p] = nullptr; }
        if (c->ev_ar[p]) { (void)hipEventDestroy(c->ev_ar[p]); c->ev_ar[p] = nullptr; }
    }
    if (c->comm_scratch) { (void)hipFree(c->comm_scratch); c->comm_scratch = nullptr; }
    c->comm_rank = 0; c->comm_size = 1;
}
void need_problem(klnmf_ctx *c) {
    use(c);
    if (!c->have_problem) fail(KLNMF_ERR_ARG, "klnmf_set_problem has not been called");
}

EventPair begin_event(klnmf_ctx *c, std::vector<EventPair> &v) {
    EventPair e{};
    HIPCHK(hipEventCreate(&e.a));
    HIPCHK(hipEventCreate(&e.b));
    HIPCHK(hipEventRecord(e.a, c->stream));
    v.push_back(e);
    return e;
}

// ---------------------------------------------------------------- dispatch ---
template <int MODE>
void launch_rowpass4_kt(klnmf_ctx *c, const RowPass4Args &a, int grid_x, int grid_y = 1) {
    const dim3 grid(grid_x, grid_y);
    const int odd = 2 * c->KT - c->ks;
    const bool ep = c->kc >= 0;
    const bool ne = MODE == ROW_UPDATE && c->ne_loop && !c->big && c->q8() && a.base.Qt != nullptr;      // (NE kernels: Q8 = 2)
    if (MODE == ROW_UPDATE) c->last_row_ne = ne;
#define KL_ROW4_CASE(KTV)                                                                                       \
    case KTV:                                                                                                   \
        if constexpr (MODE == ROW_UPDATE) {                                                                     \
            if (grid_y > 1 && c->q8() && a.base.Qt) {      /* column-split pass leaving fp8 ratio tiles */      \
                if (ne) {      /* ... and the ratio without the numerator's eps */                              \
                    if (ep) {                                                                                   \
                        if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 1, 8, 1, 2>), grid, dim3(kThreads4), 0, c->stream, a);  \
                        else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 8, 1, 2>), grid, dim3(kThreads4), 0, c->stream, a);      \
                    } else {                                                                                    \
                        if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 0, 8, 1, 2>), grid, dim3(kThreads4), 0, c->stream, a);  \
                        else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 8, 1, 2>), grid, dim3(kThreads4), 0, c->stream, a);      \
                    }                                                                                           \
                    break;                                                                                      \
                }                                                                                               \
                if (ep) {                                                                                       \
                    if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 1, 8, 1, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
                    else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 8, 1, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
                } else {                                                                                        \
                    if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 0, 8, 1, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
                    else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 8, 1, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
                }                                                                                               \
                break;                                                                                          \
            }                                                                                                   \
            if (grid_y > 1) {         /* column-split pass: its own instantiations (SPLIT = 1) */               \
                if (ep) {                                                                                       \
                    if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 1, 8, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
                    else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 8, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
                } else {                                                                                        \
                    if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 0, 8, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
                    else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 8, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
                }                                                                                               \
                break;                                                                                          \
            }                                                                                                   \
        }                                                                                                       \
        if constexpr (MODE == ROW_UPDATE) {                                                                     \
            if (c->q8() && a.base.Qt) {     /* fp8 ratio tiles for the column pass */                             \
                if (ne) {                                                                                       \
                    if (ep) {                                                                                   \
                        if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 1, 8, 0, 2>), grid, dim3(kThreads4), 0, c->stream, a);  \
                        else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 8, 0, 2>), grid, dim3(kThreads4), 0, c->stream, a);      \
                    } else {                                                                                    \
                        if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 0, 8, 0, 2>), grid, dim3(kThreads4), 0, c->stream, a);  \
                        else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 8, 0, 2>), grid, dim3(kThreads4), 0, c->stream, a);      \
                    }                                                                                           \
                    break;                                                                                      \
                }                                                                                               \
                if (ep) {                                                                                       \
                    if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 1, 8, 0, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
                    else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 8, 0, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
                } else {                                                                                        \
                    if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 0, 8, 0, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
                    else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 8, 0, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
                }                                                                                               \
                break;                                                                                          \
            }                                                                                                   \
        }                                                                                                       \
        if (ep) {                                                                                               \
            if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 1>), grid, dim3(kThreads4), 0, c->stream, a);  \
            else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1>), grid, dim3(kThreads4), 0, c->stream, a);      \
        } else {                                                                                                \
            if (odd) hipLaunchKernelGGL((k_rowpass4<KTV, 1, MODE, 0>), grid, dim3(kThreads4), 0, c->stream, a);  \
            else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0>), grid, dim3(kThreads4), 0, c->stream, a);      \
        }                                                                                                       \
        break;
#define KL_ROW4_BIG(KTV)                                                                                        \
    case KTV:                                                                                                   \
        if constexpr (MODE == ROW_UPDATE) {                                                                     \
            if (c->q8() && a.base.Qt) {     /* FUSED order leaving fp8 ratio tiles */                           \
                if (ep) hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 4, 0, 1>), grid, dim3(256), 0, c->stream, a);   \
                else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 4, 0, 1>), grid, dim3(256), 0, c->stream, a);      \
                break;                                                                                          \
            }                                                                                                   \
        }                                                                                                       \
        if (ep) hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 1, 4>), grid, dim3(256), 0, c->stream, a);   \
        else hipLaunchKernelGGL((k_rowpass4<KTV, 0, MODE, 0, 4>), grid, dim3(256), 0, c->stream, a);      \
        break;
    switch (c->KT) {
        KL_ROW4_CASE(1) KL_ROW4_CASE(2) KL_ROW4_CASE(3) KL_ROW4_CASE(4)
        KL_ROW4_CASE(5) KL_ROW4_CASE(6) KL_ROW4_CASE(7)
        KL_ROW4_BIG(8) KL_ROW4_BIG(10) KL_ROW4_BIG(12) KL_ROW4_BIG(14) KL_ROW4_BIG(16)
        default: fail(KLNMF_ERR_UNSUPP, "ping-pong row pass: 224 < k <= 256 runs on the generation-1 kernel");
    }
#undef KL_ROW4_BIG
#undef KL_ROW4_CASE
    HIPCHK(hipGetLastError());
}

// the probe column of the e4m3 W image (colq8x.hip.h): the last padded component, if neither a real component nor the eps
// carrier lives there
int w8_probe_col(const klnmf_ctx *c) { return (c->KP - 1 >= c->k && c->KP - 1 != c->kc) ? c->KP - 1 : -1; }

void fast_rowpass(klnmf_ctx *c, int mode, int store_q = 0) {
    RowPassArgs a{};
    a.VtA = c->VtA;
    a.Qt = (store_q && mode == ROW_UPDATE) ? c->Qt : nullptr;
    a.Wb_old = c->Wb[c->cur];
    a.W32_old = c->W32[c->cur];
    a.Wb_new = c->Wb[c->cur ^ 1];
    a.W32_new = c->W32[c->cur ^ 1];
    a.loss_part = c->loss_part2;
    a.hsum = c->hsum;
    a.tcur = c->tcur;
    // the new W image goes with the NEXT dictionary image: row-normalised after an H rule (fit), else the hs-based one
    a.tnext = mode == ROW_UPDATE ? (store_q ? c->t_unit : c->t_hs) : c->tcur;
    a.kc = c->kc;
    a.st = c->st;
    a.nrt = c->nrt;
    a.nct = c->nct;
    a.eps = (float)(kEpsRatio * c->v_scale);
    a.cq_on = c->images_measured ? 1 : 0;
    // e4m3 image of W_new written by the W rule itself (KLNMF_COL8=2, whole-row launches): the maxima go to the 64-row table
    // k_post turns into the next scales, and the image is written from the loop's second iteration on, so that the third can
    // already multiply it
    const bool w8_here = c->w8_tail && c->W8 && store_q && mode == ROW_UPDATE && c->row_chunks == 1 && c->q8_loop && c->iter_in_loop >= 1;
    if (c->w8_tail && mode == ROW_UPDATE) { c->conv_ran = false; c->tail_use8 = false; }
    if (w8_here) {
        c->tail_use8 = c->w8_meas && c->q8();
        a.w8tab = c->w8tab;
        a.W8 = c->W8;
        a.w8s = c->w8s;
        a.w8_sat = &c->st->w8_sat;
        a.w8_probe = w8_probe_col(c);
    }
    if (mode == ROW_UPDATE && a.Qt && c->q8()) c->stat_q8_tiles += 1;      // this update leaves fp8 ratio tiles
    EventPair ev{};
    if (c->profiling) ev = begin_event(c, c->ev_row);
    RowPass4Args a4{a, c->Ht4};
    const int nw = c->big ? 4 : kWaves4;
    const int grid4 = (c->nrt + nw - 1) / nw;
    if (mode == ROW_UPDATE && c->row_chunks > 1) {     // few rows: column chunks in blockIdx.y, W rule from the slabs
        a4.base.gpart = c->Gpart;
        a4.base.ct_chunk = c->row_ct_chunk;
        launch_rowpass4_kt<ROW_UPDATE>(c, a4, grid4, c->row_chunks);
        const int64_t rows = (int64_t)c->nrt * 32;
        hipLaunchKernelGGL(k_wrule_slabs, dim3(grid_for(rows * (c->KP / 4))), dim3(256), 0, c->stream,
                           (const float *)c->Gpart, c->row_chunks, rows * c->KP, (const float *)c->W32[c->cur],
                           c->W32[c->cur ^ 1], c->Wb[c->cur ^ 1], rows, c->KP, (int)w_ld(c->KP), c->kc,
                           (const DevState *)c->st, a.tcur, a.tnext);
        HIPCHK(hipGetLastError());
        if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
        return;
    }
    if (mode == ROW_UPDATE && c->tail_wg > 0) {
        // hybrid: the full rounds of workgroups take whole rows; the last partial round (tail_wg < CUs workgroups
        // that would each run a whole row block's length on an otherwise idle chip) is split into column chunks
        // and its W rule applied from the slabs
        launch_rowpass4_kt<ROW_UPDATE>(c, a4, grid4 - c->tail_wg);
        EventPair evt{};
        if (c->profiling) evt = begin_event(c, c->ev_tail);
        RowPass4Args t4 = a4;
        t4.base.wg0 = grid4 - c->tail_wg;
        t4.base.rt0 = c->tail_rt0();
        t4.base.gpart = c->Gpart;
        t4.base.ct_chunk = c->tail_ct_chunk;
        launch_rowpass4_kt<ROW_UPDATE>(c, t4, c->tail_wg, c->tail_chunks);
        const int64_t row0 = (int64_t)t4.base.rt0 * 32, rows = (int64_t)(c->nrt - t4.base.rt0) * 32;
        hipLaunchKernelGGL(k_wrule_slabs, dim3(grid_for(rows * (c->KP / 4))), dim3(256), 0, c->stream,
                           (const float *)c->Gpart, c->tail_chunks, rows * c->KP,
                           (const float *)c->W32[c->cur] + row0 * c->KP, c->W32[c->cur ^ 1] + row0 * c->KP,
                           c->Wb[c->cur ^ 1] + row0 * w_ld(c->KP), rows, c->KP, (int)w_ld(c->KP), c->kc,
                           (const DevState *)c->st, a.tcur, a.tnext);
        HIPCHK(hipGetLastError());
        if (w8_here) {         // the rows of the split tail: image and maxima by the conversion kernel (a few thousand rows)
            const int groups = c->KP / 8, rpb = 256 / groups;
            const int blocks = (int)std::min<int64_t>((rows + rpb - 1) / rpb, kW8Blocks);
            hipLaunchKernelGGL(k_w8_from_wb, dim3(blocks), dim3(256), 0, c->stream, (const opnd_t *)c->Wb[c->cur ^ 1] + row0 * w_ld(c->KP),
                               c->W8 + row0 * w8_ld(c->KP), rows, c->KP, (int)w_ld(c->KP), (const float *)c->w8s,
                               (const DevState *)c->st, &c->st->w8_sat, w8_probe_col(c), c->w8tab);
            HIPCHK(hipGetLastError());
            c->w8_meas = true;
            c->conv_ran = true;
        }
        if (c->profiling) { HIPCHK(hipEventRecord(evt.b, c->stream)); HIPCHK(hipEventRecord(ev.b, c->stream)); }
        return;
    }
    if (w8_here) { c->w8_meas = true; c->conv_ran = true; }
    switch (mode) {
        case ROW_UPDATE: launch_rowpass4_kt<ROW_UPDATE>(c, a4, grid4); break;
        case ROW_INIT: launch_rowpass4_kt<ROW_INIT>(c, a4, grid4); break;
        default: launch_rowpass4_kt<ROW_LOSS>(c, a4, grid4); break;
    }
    if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
}

// ---- the fused iteration tail (post.hip.h): column pass of one column part, then k_post -----------------------------------
ColPassQArgs colq_part_args(klnmf_ctx *c, const klnmf_ctx::PartCfg &p) {
    ColPassQArgs a{};
    a.Qt = c->Qt + (int64_t)p.ct0 * c->nrt * (c->q8() ? kQTile8 : kQTile);
    a.Wb_new = c->Wb[c->cur ^ 1];
    a.Npart = c->NpartF + p.slab_off;
    a.st = c->st;
    a.nrt = c->nrt;
    a.nct = p.nct;
    a.ncb = p.ncb;
    a.nchunks = p.nchunks;
    a.stages_per_chunk = p.spc;
    a.f_pad = p.ld;
    a.guard = 0;
    a.st_rw = c->st;
    a.q8_list = c->q8() ? c->q8_list : nullptr;
    return a;
}

// ---- the fp8 monitor (monitor.hip.h) -------------------------------------------------------------------------------------------
void monitor_setup(klnmf_ctx *c) {
    c->mon_part = nullptr;
    c->mon_pending = false;
    if (!c->q8_ok) return;
    c->mon_part = (float *)c->dalloc((size_t)kMonBlocks * 2 * 2 * c->KP * 32 * 4);
    c->mon_spread = (float *)c->dalloc((size_t)kMonBlocks * 96 * 4);
}
// When: DRY on a loop's second iteration (16-bit tiles still: the e4m3 bytes are formed by the monitor itself -- the loop enters
// the fp8 regime only if that measurement passes), then on fp8 iterations 1 (the first with the real tiles and the e4m3 W
// image), 2, 3, 4, 6, 8, 12, 16 and every eighth after that: dead zones open as the fit converges (ratios gather inside one e4m3
// step of 1), within a few iterations on small problems, and move slowly afterwards; the poll that acts on the result keeps the
// same cadence (poll_fp8_overflow)
bool monitor_due(int64_t n8) { return (n8 >= 1 && n8 <= 4) || n8 == 6 || n8 == 12 || (n8 >= 8 && (n8 & 7) == 0); }
// behind the row pass (and the conversion of the e4m3 W image) of a fit iteration on fp8 tiles, before its column pass
void launch_monitor(klnmf_ctx *c, bool use8) {
    c->mon_pending = false;
    c->mon_dry_pending = false;
    if (!c->mon_part || !c->sw.q8_monitor || !c->q8_loop || c->in_capture) return;
    const bool dry = c->iter_in_loop == 1 && !c->q8();
    if (!dry && !(c->q8() && monitor_due(c->stat_q8_tiles))) return;
    MonArgs a{};
    a.st = c->st; a.Qt = c->Qt; a.VtA = (const _Float16 *)c->VtA; a.W32_old = c->W32[c->cur]; a.H_old = c->H32;
    a.Wb_new = c->Wb[c->cur ^ 1]; a.W8 = (use8 && !dry) ? c->W8 : nullptr; a.w8s = c->w8s; a.part = c->mon_part;
    a.spread = c->mon_spread;
    a.dry = dry ? 1 : 0;
    a.w8tab = (dry && c->W8 != nullptr && c->conv_ran) ? c->w8tab : nullptr;
    a.nrt = c->nrt; a.nct = c->nct; a.kp = c->KP; a.k = (int)c->k; a.wld = (int)w_ld(c->KP); a.w8ld = (int)w8_ld(c->KP);
    a.f_pad = c->f_pad;
    const int tiles = (int)((c->f + 31) / 32);                       // column tiles that hold data
    a.ct = (int)((c->mon_checks * 5 + 1) % tiles);                   // rotates with the checks (5: coprime to the usual tile counts)
    a.ncols = (int)std::min<int64_t>(32, c->f - (int64_t)a.ct * 32);
    a.nrt_data = (int)((c->n + 31) / 32);                            // row tiles that hold data: the sample walks (and wraps inside) them
    a.nsamp = std::min(2 * kMonBlocks, a.nrt_data) & ~1;
    if (a.nsamp < 2) return;
    a.rot = (int)((c->mon_checks * 7) % a.nrt_data);
    a.eps = (float)(kEpsRatio * c->v_scale);
    hipLaunchKernelGGL(k_q8_monitor, dim3(kMonBlocks), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    c->mon_pending = true;
    c->mon_dry_pending = dry;
    c->mon_ncols = a.ncols;
    c->mon_noise_scale = (float)std::min(0.5, (double)(a.nsamp / 2) * 32.0 / (double)c->n);
    c->mon_checks += 1;
}

// the e4m3 image of W_new for this iteration's fp8 x fp8 column pass (once per iteration, before the first part's pass):
// converted with the scales k_post derived from the PREVIOUS conversion's maxima; returns whether the fp8 x fp8 pass may run
bool fused_w8_stage(klnmf_ctx *c) {
    if (c->w8_tail) return c->tail_use8 && c->conv_ran;      // the row pass's W rule wrote image and maxima itself (fast_rowpass)
    c->conv_ran = false;
    if (!c->W8 || !c->q8_loop) return false;
    const bool measure_only = c->iter_in_loop == 1 && !c->w8_meas;      // the loop's second iteration (16-bit tiles still)
    if (!measure_only && !c->q8()) return false;
    const bool use8 = !measure_only && c->w8_meas;
    const int groups = c->KP / 8, rpb = std::max(1, 256 / groups);
    const int64_t rows = c->n_pad;
    const int blocks = (int)std::min<int64_t>((rows + rpb - 1) / rpb, kW8Blocks);
    hipLaunchKernelGGL(k_w8_from_wb, dim3(blocks), dim3(256), 0, c->stream, (const opnd_t *)c->Wb[c->cur ^ 1], c->W8, rows,
                       c->KP, (int)w_ld(c->KP), (const float *)c->w8s, (const DevState *)c->st,
                       &c->st->w8_sat, w8_probe_col(c), c->w8tab);
    HIPCHK(hipGetLastError());
    c->w8_meas = true;
    c->conv_ran = true;
    return use8;
}

void fused_colpass_part(klnmf_ctx *c, const klnmf_ctx::PartCfg &p, bool use8) {
    ColPassQArgs a = colq_part_args(c, p);
    const int grid = p.ncb * p.nchunks;
    EventPair ev{};
    if (c->profiling) ev = begin_event(c, c->ev_col);
    auto launch_q2 = [&](const ColPassQArgs &g, bool fp8_tiles) {
        switch (c->KT) {
#define KL_PQ2(KTV) case KTV:                                                                                                    \
            if (fp8_tiles) hipLaunchKernelGGL((k_colpass_q2<KTV, KL_COLQ8_NB, 1, 1, KL_COLQ8_PAIR>), dim3(grid), dim3(kThreads), 0, c->stream, g); \
            else hipLaunchKernelGGL((k_colpass_q2<KTV, KL_COLQ_NB>), dim3(grid), dim3(kThreads), 0, c->stream, g);                \
            break;
#define KL_PQ2_BIG(KTV) case KTV:                                                                                                \
            if (fp8_tiles) hipLaunchKernelGGL((k_colpass_q2<KTV, 3, 2, 1, 1>), dim3(grid), dim3(kThreads), 0, c->stream, g);      \
            else hipLaunchKernelGGL((k_colpass_q2<KTV, 3, 2>), dim3(grid), dim3(kThreads), 0, c->stream, g);                      \
            break;
            KL_PQ2(1) KL_PQ2(2) KL_PQ2(3) KL_PQ2(4) KL_PQ2(5) KL_PQ2(6) KL_PQ2(7)
            KL_PQ2_BIG(8) KL_PQ2_BIG(10) KL_PQ2_BIG(12) KL_PQ2_BIG(14) KL_PQ2_BIG(16)
#undef KL_PQ2
#undef KL_PQ2_BIG
            default: fail(KLNMF_ERR_UNSUPP, "stored-ratio column pass: k <= 224 or 256 < k <= 512");
        }
        HIPCHK(hipGetLastError());
    };
    if (use8) {
        a.guard = 1;                       // returns at once if this iteration's e4m3 image clipped; the f16-operand pass behind runs then
        ColPass8Args a8{a, c->W8, c->w8s, w8_probe_col(c) >= 0 ? 1 : 0};
        if (c->KT == 8 && !c->big) fail(KLNMF_ERR_UNSUPP, "fp8 x fp8 column pass: KT = 8 only on the component-split path");
        switch (c->KT) {
#define KL_PQ8X(KTV, NBV, KSV) case KTV:                                                                                          \
            if (a8.probe) hipLaunchKernelGGL((k_colpass_q8x<KTV, NBV, KSV, 1>), dim3(grid), dim3(kThreads), 0, c->stream, a8);     \
            else hipLaunchKernelGGL((k_colpass_q8x<KTV, NBV, KSV, 0>), dim3(grid), dim3(kThreads), 0, c->stream, a8);              \
            break;
            KL_PQ8X(1, KL_COL8_NB, 1) KL_PQ8X(2, KL_COL8_NB, 1) KL_PQ8X(3, KL_COL8_NB, 1) KL_PQ8X(4, KL_COL8_NB, 1)
            KL_PQ8X(5, KL_COL8_NB, 1) KL_PQ8X(6, KL_COL8_NB, 1) KL_PQ8X(7, KL_COL8_NB, 1)
            KL_PQ8X(8, 3, 2) KL_PQ8X(10, 3, 2) KL_PQ8X(12, 3, 2) KL_PQ8X(14, 3, 2) KL_PQ8X(16, 3, 2)
#undef KL_PQ8X
            default: fail(KLNMF_ERR_UNSUPP, "fp8 x fp8 column pass: k <= 224 or 256 < k <= 512");
        }
        HIPCHK(hipGetLastError());
        ColPassQArgs g = a;
        g.guard = 2;
        launch_q2(g, true);
    } else {
        launch_q2(a, c->q8());
    }
    if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
}

enum PostMode { POST_FULL = 0, POST_SUM = 1, POST_RULE = 2 };
// POST_FULL: everything behind the column pass of a single-context fit iteration (`la`: the row pass's loss partials and the
// stop rule's tolerance); POST_SUM: slabs -> numerator of ONE part (+ fix-ups; `la.part` set: one extra block leaves the loss
// in loss_xchg); POST_RULE: the H rule on the numerator as it stands (decide: with the stop rule on loss_xchg[0])
void launch_post(klnmf_ctx *c, PostMode mode, const klnmf_ctx::PartCfg *parts, int nparts, const LossArgs &la, bool decide,
                 bool use8, bool last_sum) {
    PostArgs a{};
    a.nparts = nparts;
    for (int p = 0; p < nparts; ++p) {
        const klnmf_ctx::PartCfg &q = parts[p];
        a.part[p] = PostPart{c->NpartF + q.slab_off, c->numerF + q.numer_off, (int64_t)c->KP * q.ld, q.ld, q.col0, q.ncols,
                             q.nchunks, q.ct0};
    }
    a.do_sum = mode != POST_RULE;
    a.do_rule = mode != POST_SUM;
    a.do_decide = decide ? 1 : 0;
    a.loss_from_parts = (mode == POST_FULL && la.part != nullptr) ? 1 : 0;
    a.loss_block = (mode == POST_SUM && la.part != nullptr) ? 1 : 0;
    a.w8_block = (a.do_sum && last_sum && c->conv_ran && c->w8tab != nullptr) ? 1 : 0;
    a.last_sum = (a.do_sum && last_sum) ? 1 : 0;
    a.it = (int)(c->iter_in_loop & 1);
    a.mon = MonPost{nullptr, nullptr, kMonMinSpread, kMonMaxCommon, 0, 0.f, kMonThreshold};
    if (a.do_sum && c->mon_pending) {
        a.mon = MonPost{c->mon_part, c->mon_spread, c->sw.mon_min_spread >= 0.f ? c->sw.mon_min_spread : kMonMinSpread,
                        c->sw.mon_threshold > 0.f ? 1.0f : kMonMaxCommon, c->mon_ncols,
                        c->mon_noise_scale, c->sw.mon_threshold > 0.f ? c->sw.mon_threshold : mon_threshold_for((float)c->loop_planned)};
        c->mon_pending = false;
    }
    a.loss_part = la.part; a.loss_count = la.count; a.inv_c = la.inv_c; a.loss_xchg = c->loss_xchg; a.ne = la.ne; a.cq_on = la.cq_on;
    a.tol_abs = la.tol_abs; a.errors = c->errors; a.cap = c->cap;
    a.st = c->st;
    a.H_old = c->H32; a.H_new = c->H32alt;
    a.Ht4 = c->Ht4; a.hsum = c->hsum; a.tcur = c->tcur; a.t_hs = c->t_hs;
    a.f = c->f; a.f_pad = c->f_pad; a.kp = c->KP; a.k = (int)c->k; a.kc = c->kc;
    a.eps_pad = (float)(kEpsRatio * c->v_scale);
    const bool fix = a.do_sum && c->q8() && c->q8_list != nullptr && c->sw.q8_fixup;      // (KLNMF_Q8_FIXUP=0: the tests' control run)
    a.list = fix ? c->q8_list : nullptr;
    a.Qt = c->Qt; a.VtA = (const _Float16 *)c->VtA; a.W32_old = c->W32[c->cur]; a.Wb_new = c->Wb[c->cur ^ 1];
    a.W8 = use8 ? c->W8 : nullptr; a.w8s = c->w8s; a.w8ld = (int)w8_ld(c->KP); a.wld = (int)w_ld(c->KP);
    a.nrt = c->nrt; a.nct = c->nct; a.stages_per_chunk = nparts == 1 ? parts[0].spc : 0; a.eps = (float)(kEpsRatio * c->v_scale);
    a.w8tab = c->w8tab; a.w8s_next = c->w8s_next;
    if (a.do_sum && !fix && c->q8() && c->q8_list != nullptr)      // fix-ups switched off: the list must still be emptied
        HIPCHK(hipMemsetAsync(&c->st->q8_list_n, 0, sizeof(int), c->stream));
    // one block per component row, one float4 per thread and trip: 1024 threads for rows of 4096 columns and more
    const int hthreads = c->f_pad >= 4096 ? 1024 : (c->f_pad >= 2048 ? 512 : 256);
    const int blocks = (int)c->k + a.w8_block + a.loss_block;
    hipLaunchKernelGGL(k_post, dim3((unsigned)blocks), dim3(hthreads), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    if (a.do_rule) {
        std::swap(c->H32, c->H32alt);
        c->loop_hswaps += 1;
        c->images_measured = false;
    }
    if (a.w8_block) {
        std::swap(c->w8s, c->w8s_next);      // what this launch derived is what the next image is written with
        c->conv_ran = false;
    }
}

void fast_pack_H(klnmf_ctx *c, const unsigned *wmax = nullptr) {
    // the dictionary's fp16 tile images, row sums and image scales from its fp32 master (no update: the H rule of a loop runs in
    // k_post).  One block per component row; its passes over the row are a chain of memory round trips, so a long row gets more
    // threads (fewer elements per thread and pass)
    const int hthreads = c->f_pad >= 4096 ? 1024 : (c->f_pad >= 2048 ? 512 : 256);
    hipLaunchKernelGGL(k_update_pack_H, dim3((unsigned)c->k), dim3(hthreads), 0, c->stream, c->H32, (const float *)c->numerF,
                       c->Ht4, c->hsum, c->tcur, c->t_hs, wmax, &c->st->op_range, c->f, c->f_pad, c->KP, 0,
                       (const DevState *)nullptr, c->kc, (float)(kEpsRatio * c->v_scale), 0, (int64_t)c->KP * c->f_pad,
                       wmax ? (const DevState *)c->st : (const DevState *)nullptr);      // measured images carry DevState.cq_e
    HIPCHK(hipGetLastError());
    c->images_measured = wmax != nullptr;
}

void fast_pack_W(klnmf_ctx *c) {
    hipLaunchKernelGGL(k_pack_W, dim3(grid_for(c->n_pad * c->KP, 256, 8192)), dim3(256), 0,
                       c->stream, (const float *)c->W32[c->cur], c->Wb[c->cur], c->n_pad, c->KP,
                       w_ld(c->KP), c->kc, (const float *)c->tcur);
    HIPCHK(hipGetLastError());
}

// Both images of the current (W, H) with scales MEASURED from W's column maxima (a W that no W rule produced: W0 = V.H0^T,
// klnmf_set_W -- see opnd_t in mfma.hip.h).  They are valid for one update; the update's W rule packs the next W image
// with the hs-based / row-normalised scale again.
// eps through the matrix product (kc >= 0, k_update_pack_H) is the pair "W image column kc = 2^-10, dictionary image row kc =
// eps x c x 2^10" (c = the storage factor of V, a power of two fixed by klnmf_set_v_max).  Both must be fp16 numbers: with
// max(V) below 5e-6 the row value passes 65504 (round 4's data fuzz: V x 1e-6, k = 40 -- eps came out 25 % small, the losses
// 4 % off), with max(V) above ~1e7 it underflows to 0 and the padded rows of the last row tile divide 0 by 0 (V x 1e6,
// 70 000 rows: NaN).  Outside [2^-20, 2^15] -- max(V) outside about [1e-5, 3e5] -- the carrier is dropped and the kernels add
// eps in their fp32 epilogue (the EP = 0 instantiations every shape has; one more VALU instruction per element).  Below
// 2^-14 the row value is a subnormal half (at 2^-20: 16 steps, eps good to 3 %): V is then 1e11 times eps and more, and all
// that is asked of eps is to keep 0 / 0 out of the empty rows.
static void choose_eps_carrier(klnmf_ctx *c) {
    const double ev = kEpsRatio * c->v_scale / (double)kCarrierW;
    const bool fits = ev >= 9.5367431640625e-07 && ev <= 32768.0;
    c->kc = (c->kc_shape >= 0 && fits) ? c->kc_shape : -1;
}

// from_init: W is W0 = V.H0^T of klnmf_init_W -- the first update's ratios are about f / k times 1, and the dictionary image
// then carries the ratio scale k_ratio_scale derives (mfma.hip.h; KLNMF_RATIO_SCALE=0: never); any other W: scale 1.
void measure_and_pack(klnmf_ctx *c, bool from_init = false) {
    c->refusals_dirty = true;
    const bool cq_ok = c->sw.ratio_scale;
    // (the eps row of the image is scaled too: it must stay an fp16 number)
    int e_cap = 12;
    if (c->kc >= 0) {
        const double ev = kEpsRatio * c->v_scale / (double)kCarrierW;
        int ex = 0;
        (void)std::frexp(32768.0 / ev, &ex);
        e_cap = std::min(12, std::max(0, ex - 1));
    }
    hipLaunchKernelGGL(k_ratio_scale, dim3(1), dim3(256), 0, c->stream, (const double *)c->hsum, (const float *)c->tcur, (int)c->k, c->f,
                       c->st, from_init && cq_ok ? 1 : 0, e_cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->wmax, 0, (size_t)c->KP * 4, c->stream));
    HIPCHK(hipMemsetAsync(&c->st->op_range, 0, sizeof(int), c->stream));
    const int rows_grid = (int)std::min<int64_t>(c->n_pad, 1024);
    hipLaunchKernelGGL(k_colmax_W, dim3(rows_grid, (c->KP + 255) / 256), dim3(256), 0, c->stream,
                       (const float *)c->W32[c->cur], c->n_pad, c->KP, c->wmax);
    HIPCHK(hipGetLastError());
    fast_pack_H(c, c->wmax);
    fast_pack_W(c);
}

// When a loop gives the fp8 regime up for its remaining iterations: bulk saturation of the ratio tiles (more entries per
// iteration than the fix-up list holds: the range rule at the loop's entry normally excludes such matrices), or the monitor's
// statistic above its threshold (monitor.hip.h).  Polled on the monitor's cadence -- fp8 iterations 1 .. 4 and every eighth
// after them (one DevState read-back and stream synchronisation each).
// Row shards: every rank must drop the tiles in the SAME iteration (they would run different kernels otherwise, and the
// replicas of H would drift apart): the count travels as the second double of the loss exchange -- every loss launch leaves
// this rank's q8_unfixed in loss_xchg[1], the all-reduce (native or torch) sums it -- and `agreed` polls read that sum.
void poll_fp8_overflow(klnmf_ctx *c, bool agreed = false) {
    if (!c->q8_loop || c->in_capture) return;
    const bool dry = c->mon_dry_pending;          // the dry run of the iteration just enqueued decides whether the next one takes fp8 tiles
    c->mon_dry_pending = false;
    if (!dry && !(c->q8() && monitor_due(c->stat_q8_tiles))) return;
    if (agreed) {
        double h[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h, c->loss_xchg, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (h[1] > 0) { c->q8_loop = false; c->stat_mon_gave_up = true; }
        return;
    }
    DevState hs{};
    HIPCHK(hipMemcpyAsync(&hs, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // 16-bit tiles from the next iteration on (klnmf_query reports the counts and the statistic)
    if (hs.q8_unfixed > 0 || hs.mon_trips > 0) { c->q8_loop = false; c->stat_mon_gave_up = true; }
}

// ------------------------------------------------------------ exact pieces ---
// CSR input: ratio on the stored entries + loss (nmf.py:301-308, 331-334)
template <typename T>
void sparse_Q(klnmf_ctx *c, int write_q, double eps, const DecideArgs &dec) {
    const int64_t hk = c->k * c->f;
    EventPair ev{};
    if (c->profiling) ev = begin_event(c, c->ev_row);
    hipLaunchKernelGGL((k_sp_transpose_H<T>), dim3(grid_for(hk)), dim3(256), 0, c->stream, (const T *)c->H,
                       (T *)c->HT, c->k, c->f, (const DevState *)c->st);
#define KL_SPQ_ARGS (const int64_t *)c->sp_indptr, (const int64_t *)c->sp_indices, (const T *)c->sp_data, \
        (const T *)c->W[c->cur], (const T *)c->HT, (T *)c->sp_q, c->sp_row_loss, c->k, (T)eps, write_q, (const DevState *)c->st
    if (c->sp_blocked) {
        const int kcb = (int)((c->k + 63) / 64);
        const unsigned gridb = (unsigned)((int64_t)c->sp_cb * c->n);
#define KL_SPB_QW(KCV, MODEV) hipLaunchKernelGGL((k_spb_qw<T, KCV, MODEV>), dim3(gridb), dim3(64), 0, c->stream, (const int64_t *)c->sp_blkptr, \
        (const int *)c->sp_idx32, (const T *)c->sp_data, (const T *)c->W[c->cur], (const T *)c->HT, (T *)c->sp_q, c->sp_loss_part, (T *)c->sp_G,  \
        c->n, c->k, c->sp_cb, (T)eps, (const DevState *)c->st)
#define KL_SPB_QW_MODE(KCV) do { if (write_q) KL_SPB_QW(KCV, SPB_UPDATE); else KL_SPB_QW(KCV, SPB_LOSS); } while (0)
        if (kcb <= 1) KL_SPB_QW_MODE(1); else if (kcb == 2) KL_SPB_QW_MODE(2); else if (kcb <= 4) KL_SPB_QW_MODE(4); else KL_SPB_QW_MODE(8);
#undef KL_SPB_QW_MODE
        HIPCHK(hipGetLastError());
        if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
    } else {
    // (k > 512, or no stored entries: every lane takes its own entries and loops over the components)
    hipLaunchKernelGGL((k_sp_q_anyk<T>), dim3((unsigned)c->n), dim3(64), 0, c->stream, KL_SPQ_ARGS);
#undef KL_SPQ_ARGS
    HIPCHK(hipGetLastError());
    if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
    }
    hipLaunchKernelGGL((k_sp_colsum_part<T>), dim3((unsigned)c->sp_nblk), dim3(256), 0, c->stream,
                       (const T *)c->W[c->cur], c->sp_wpart, c->n, c->k, (const DevState *)c->st);
    if (c->hseg_n > 1) {
        hipLaunchKernelGGL((k_sp_hsum_part<T>), dim3((unsigned)c->hseg_n, (unsigned)c->k), dim3(256), 0, c->stream, (const T *)c->H,
                           c->f, c->hseg, c->hpart, (const DevState *)c->st);
        hipLaunchKernelGGL((k_sp_dots<T>), dim3((unsigned)c->k), dim3(256), 0, c->stream, (const double *)c->sp_wpart,
                           c->sp_nblk, (const T *)c->H, c->k, c->f, c->sp_prod, (const DevState *)c->st, (const double *)c->hpart,
                           c->hseg_n);
    } else {
        hipLaunchKernelGGL((k_sp_dots<T>), dim3((unsigned)c->k), dim3(256), 0, c->stream, (const double *)c->sp_wpart,
                           c->sp_nblk, (const T *)c->H, c->k, c->f, c->sp_prod, (const DevState *)c->st);
    }
    hipLaunchKernelGGL(k_sp_loss, dim3(1), dim3(1024), 0, c->stream, (const double *)(c->sp_blocked ? c->sp_loss_part : c->sp_row_loss),
                       c->sp_blocked ? (int64_t)c->sp_cb * c->n : c->n, (const double *)c->sp_prod, c->k, c->loss_xchg,
                       (const DevState *)c->st, dec);
    HIPCHK(hipGetLastError());
}

// k_gemm: 64 x 64 tiles, the inner product on the fp64 / fp32 MFMA (exact.hip.h)
#define KL_GEMM_TT(tt, T, EPI, grid, stream, ...) hipLaunchKernelGGL((k_gemm<T, EPI, 4, true>), grid, dim3(256), 0, stream, __VA_ARGS__)

// dec.on: the stop rule rides in the one-block loss reduction (single-context loops: no k_decide launch)
template <typename T>
void exact_Q(klnmf_ctx *c, int write_q, double eps = kEpsRatio, DecideArgs dec = DecideArgs{0, nullptr, 0.0, nullptr, 0}) {
    if (c->sparse) { sparse_Q<T>(c, write_q, eps, dec); return; }
    EpiQ<T> epi{(const T *)c->V, (T *)c->Q, c->f, c->loss_part, write_q, 0.0, (T)eps};
    const int TL = 16 * c->q_tt;
    dim3 grid((unsigned)((c->f + TL - 1) / TL), (unsigned)((c->n + TL - 1) / TL), 1);
    EventPair ev{};
    if (c->profiling) ev = begin_event(c, c->ev_row);
    KL_GEMM_TT(c->q_tt, T, EpiQ<T>, grid, c->stream, (int)c->n, (int)c->f,
               (int)c->k, (const T *)c->W[c->cur], (int64_t)c->k, (int64_t)1,
               (const T *)c->H, (int64_t)c->f, (int64_t)1, (int)c->k + GK,
               (const DevState *)c->st, epi);
    HIPCHK(hipGetLastError());
    if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
    hipLaunchKernelGGL(k_sum_doubles, dim3(1), dim3(1024), 0, c->stream,
                       (const double *)c->loss_part, (int64_t)grid.x * grid.y, c->loss_xchg,
                       (const DevState *)c->st, dec);
    HIPCHK(hipGetLastError());
}

// W_new = W * (Qsrc . H^T)   (multiply=0: W_new = Qsrc . H^T, i.e. W0 with Qsrc = V)
template <typename T>
void exact_W(klnmf_ctx *c, const void *qsrc, int multiply) {
    if (c->sparse && c->sp_blocked) {
        // the partial sums of Q . H^T per column block are there (sparse_Q, fused with the ratio); W0 = X . H0^T: the same pass
        // over the entries' values
        if (!multiply) {
            const int kcb = (int)((c->k + 63) / 64);
            const unsigned gridb = (unsigned)((int64_t)c->sp_cb * c->n);
#define KL_SPB_INIT(KCV) hipLaunchKernelGGL((k_spb_qw<T, KCV, SPB_INIT>), dim3(gridb), dim3(64), 0, c->stream, (const int64_t *)c->sp_blkptr,       \
            (const int *)c->sp_idx32, (const T *)qsrc, (const T *)c->W[c->cur], (const T *)c->HT, (T *)nullptr, (double *)nullptr, (T *)c->sp_G,     \
            c->n, c->k, c->sp_cb, (T)0, (const DevState *)c->st)
            if (kcb <= 1) KL_SPB_INIT(1); else if (kcb == 2) KL_SPB_INIT(2); else if (kcb <= 4) KL_SPB_INIT(4); else KL_SPB_INIT(8);
#undef KL_SPB_INIT
        }
        hipLaunchKernelGGL((k_spb_wrule<T>), dim3(grid_for(c->n * c->k, 256, 8192)), dim3(256), 0, c->stream, (const T *)c->sp_G, c->sp_cb,
                           (const T *)c->W[c->cur], (T *)c->W[c->cur ^ 1], c->n, c->k, multiply, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        return;
    }
    if (c->sparse) {        // Q . H^T over the stored entries (qsrc: the ratio values, or X's values for W0)
        const int spw_threads = (int)std::min<int64_t>(256, (c->k + 63) / 64 * 64);
        hipLaunchKernelGGL((k_sp_w<T>), dim3((unsigned)c->n), dim3(spw_threads), 0, c->stream, (const int64_t *)c->sp_indptr,
                           (const int64_t *)c->sp_indices, (const T *)qsrc, (const T *)c->W[c->cur], (const T *)c->HT,
                           (T *)c->W[c->cur ^ 1], c->k, multiply, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        return;
    }
    if (c->wsplit > 1) {     // few rows: contraction over f split into chunks (blockIdx.z), W rule from the slabs
        EpiWpart<T> epip{(T *)c->Wpart, c->k, c->n * c->k};
        const int TLw = 16 * c->w_tt;
        dim3 gridp((unsigned)((c->k + TLw - 1) / TLw), (unsigned)((c->n + TLw - 1) / TLw), (unsigned)c->wsplit);
        KL_GEMM_TT(c->w_tt, T, EpiWpart<T>, gridp, c->stream, (int)c->n, (int)c->k,
                   (int)c->f, (const T *)qsrc, (int64_t)c->f, (int64_t)1, (const T *)c->H,
                   (int64_t)1, (int64_t)c->f, c->wchunk, (const DevState *)c->st, epip);
        HIPCHK(hipGetLastError());
        const int64_t count = c->n * c->k;
        hipLaunchKernelGGL((k_wrule_exact<T>), dim3(grid_for(count)), dim3(256), 0, c->stream, (const T *)c->Wpart,
                           c->wsplit, count, (const T *)c->W[c->cur], (T *)c->W[c->cur ^ 1], multiply,
                           (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        return;
    }
    EpiW<T> epi{(const T *)c->W[c->cur], (T *)c->W[c->cur ^ 1], c->k, multiply};
    const int TLw = 16 * c->w_tt;
    dim3 grid((unsigned)((c->k + TLw - 1) / TLw), (unsigned)((c->n + TLw - 1) / TLw), 1);
    KL_GEMM_TT(c->w_tt, T, EpiW<T>, grid, c->stream, (int)c->n, (int)c->k,
               (int)c->f, (const T *)qsrc, (int64_t)c->f, (int64_t)1, (const T *)c->H,
               (int64_t)1, (int64_t)c->f, (int)c->f + GK, (const DevState *)c->st, epi);
    HIPCHK(hipGetLastError());
}

// numer = W[widx]^T . Q   (sum_slabs = false: the dense row chunks' slabs are left for exact_H to sum: single-context loops)
template <typename T>
void exact_N(klnmf_ctx *c, int widx, bool sum_slabs = true) {
    if (c->sparse && c->sp_blocked) {        // W^T . Q over the CSC order, row block by row block (sparseb.hip.h)
        EventPair evs{};
        if (c->profiling) evs = begin_event(c, c->ev_col);
        const int kcb = (int)((c->k + 63) / 64);
        const unsigned gridb = (unsigned)((int64_t)c->sp_rb * c->f);
#define KL_SPB_N(KCV) hipLaunchKernelGGL((k_spb_n<T, KCV>), dim3(gridb), dim3(64), 0, c->stream, (const int64_t *)c->csc_blkptr, (const int *)c->csc_rows32, \
        (const int *)c->csc_perm32, (const T *)c->sp_q, (const T *)c->W[widx], (T *)c->sp_NT, c->k, c->f, c->sp_rb, (const DevState *)c->st)
        if (kcb <= 1) KL_SPB_N(1); else if (kcb == 2) KL_SPB_N(2); else if (kcb <= 4) KL_SPB_N(4); else KL_SPB_N(8);
#undef KL_SPB_N
        hipLaunchKernelGGL((k_spb_numer<T>), dim3((unsigned)((c->f + 31) / 32), (unsigned)((c->k + 31) / 32)), dim3(256), 0, c->stream,
                           (const T *)c->sp_NT, c->sp_rb, (T *)c->numer, c->k, c->f, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        if (c->profiling) HIPCHK(hipEventRecord(evs.b, c->stream));
        return;
    }
    if (c->sparse) {        // W^T . Q, one block per feature column (CSC order)
        EventPair evs{};
        if (c->profiling) evs = begin_event(c, c->ev_col);
        const int spn_threads = (int)std::min<int64_t>(256, (c->k + 63) / 64 * 64);
        hipLaunchKernelGGL((k_sp_n<T>), dim3((unsigned)c->f), dim3(spn_threads), 0, c->stream, (const int64_t *)c->csc_indptr,
                           (const int64_t *)c->csc_rows, (const int64_t *)c->csc_perm, (const T *)c->sp_q,
                           (const T *)c->W[widx], (T *)c->numer, c->k, c->f, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        if (c->profiling) HIPCHK(hipEventRecord(evs.b, c->stream));
        return;
    }
    EpiN<T> epi{(T *)c->Npart, c->f, c->k * c->f};
    const int TLn = 16 * c->n_tt;
    dim3 grid((unsigned)((c->f + TLn - 1) / TLn), (unsigned)((c->k + TLn - 1) / TLn), (unsigned)c->nsplit);
    EventPair ev{};
    if (c->profiling) ev = begin_event(c, c->ev_col);
    KL_GEMM_TT(c->n_tt, T, EpiN<T>, grid, c->stream, (int)c->k, (int)c->f,
               (int)c->n, (const T *)c->W[widx], (int64_t)1, (int64_t)c->k,
               (const T *)c->Q, (int64_t)c->f, (int64_t)1, c->kchunk,
               (const DevState *)c->st, epi);
    HIPCHK(hipGetLastError());
    if (c->profiling) HIPCHK(hipEventRecord(ev.b, c->stream));
    if (!sum_slabs) return;
    const int64_t count = c->k * c->f;
    hipLaunchKernelGGL((k_sum_partials<T>), dim3(grid_for(count)), dim3(256), 0, c->stream,
                       (const T *)c->Npart, (T *)c->numer, count, c->nsplit,
                       (const DevState *)c->st);
    HIPCHK(hipGetLastError());
}

template <typename T>
void exact_H(klnmf_ctx *c, bool from_slabs = false) {
    if (from_slabs) {             // (dense, short rows: the rule sums the row chunks' slabs itself -- the same bits, one launch less)
        hipLaunchKernelGGL((k_update_H_slabs<T>), dim3((unsigned)c->k), dim3(256), 0, c->stream, (T *)c->H, (const T *)c->Npart,
                           c->nsplit, c->k * c->f, c->f, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        return;
    }
    if (c->hseg_n > 1) {          // long rows: S segments per row, two launches (exact.hip.h)
        hipLaunchKernelGGL((k_update_H_part<T>), dim3((unsigned)c->hseg_n, (unsigned)c->k), dim3(256), 0, c->stream, (T *)c->H,
                           (const T *)c->numer, c->f, c->hseg, c->hpart, (const DevState *)c->st);
        hipLaunchKernelGGL((k_update_H_norm<T>), dim3((unsigned)c->hseg_n, (unsigned)c->k), dim3(256), 0, c->stream, (T *)c->H,
                           c->f, c->hseg, (const double *)c->hpart, (const DevState *)c->st);
    } else {
        hipLaunchKernelGGL((k_update_H<T>), dim3((unsigned)c->k), dim3(256), 0, c->stream, (T *)c->H,
                           (const T *)c->numer, c->f, (const DevState *)c->st);
    }
    HIPCHK(hipGetLastError());
}

#define EXACT_CALL(c, fn, ...)                                         \
    do {                                                               \
        if ((c)->prec == KLNMF_PREC_F64) fn<double>(c, ##__VA_ARGS__); \
        else fn<float>(c, ##__VA_ARGS__);                              \
    } while (0)

// ------------------------------------------------------------- loop pieces ---
// Empty V tile buffers: true zeros (padding rows and columns are inert; an all-zero row of V gives an exactly zero row of W,
// as in the reference -- also under the update pass without the numerator's eps, whose ratio carries a 2^-100 addend instead
// of relying on a stored "zero class": mfma4.hip.h, NE).
void fill_v_tiles(klnmf_ctx *c, void *tiles, size_t bytes) {
    HIPCHK(hipMemsetAsync(tiles, 0, bytes, c->stream));
}

void reset_state(klnmf_ctx *c) {
    hipLaunchKernelGGL(k_reset_state, dim3(1), dim3(1), 0, c->stream, c->st);
    HIPCHK(hipGetLastError());
}

// fused_tol != nullptr (klnmf_run): the stop rule rides in the launch that reduces the loss (no k_decide launch)
void piece_rowpass(klnmf_ctx *c, int fit, const double *fused_tol = nullptr, bool defer_to_post = false) {
    // the W rule is the same for fit and transform (nmf.py:251-253); a fit also keeps the ratios for the H rule
    if (c->is_exact()) {
        // fused_tol (single-context loops): the stop rule in the loss reduction's launch, no k_decide
        DecideArgs dec{0, nullptr, 0.0, nullptr, 0};
        if (fused_tol) dec = DecideArgs{1, c->st, *fused_tol, c->errors, c->cap};
        EXACT_CALL(c, exact_Q, 1, kEpsRatio, dec);
        EXACT_CALL(c, exact_W, c->sparse ? c->sp_q : c->Q, 1);
        return;
    }
    const bool measured = c->images_measured;
    fast_rowpass(c, ROW_UPDATE, fit);
    // measured image scales live for one update: the new W image already carries the hs-based scale (tnext); in a fit
    // the H rule re-packs the dictionary image anyway, in a transform the unchanged dictionary is re-packed here
    if (measured && !fit) fast_pack_H(c);
    if (fit && defer_to_post) {
        // a fit on a communicator: one extra block of the first part's summing launch (k_post) reduces the partials into
        // loss_xchg, which is exchanged with the numerator; the stop rule rides in the launch behind the all-reduce
        c->pending_loss = LossArgs{(const double2 *)c->loss_part2, c->loss_parts(), 1.0 / c->v_scale, c->loss_xchg, 0, c->st,
                                   0.0, c->errors, c->cap, c->last_row_ne ? 1 : 0, measured ? 1 : 0};
        return;
    }
    if (fit && fused_tol) {
        // a fit in one context: nothing needs the loss before the H rule -- it is reduced (and the stop rule applied) by
        // k_post behind the column pass (piece_fit_tail)
        c->pending_loss = LossArgs{(const double2 *)c->loss_part2, c->loss_parts(), 1.0 / c->v_scale, c->loss_xchg, 1, c->st,
                                   *fused_tol, c->errors, c->cap, c->last_row_ne ? 1 : 0, measured ? 1 : 0};
        return;
    }
    hipLaunchKernelGGL(k_loss_from_parts, dim3(1), dim3(1024), 0, c->stream,
                       (const double2 *)c->loss_part2, c->loss_parts(),
                       (const DevState *)c->st, 1.0 / c->v_scale, c->loss_xchg, fused_tol ? 1 : 0, c->st,
                       fused_tol ? *fused_tol : 0.0, c->errors, c->cap, c->last_row_ne ? 1 : 0, measured ? 1 : 0);
    HIPCHK(hipGetLastError());
}

void piece_decide(klnmf_ctx *c, double tol_abs) {
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(1), 0, c->stream, c->st,
                       (const double *)c->loss_xchg, tol_abs, c->errors, c->cap);
    HIPCHK(hipGetLastError());
}

const LossArgs kNoLoss{nullptr, 0, 0.0, nullptr, 0, nullptr, 0.0, nullptr, 0, 0};

void piece_colpass(klnmf_ctx *c) {
    if (c->is_exact()) { EXACT_CALL(c, exact_N, c->cur ^ 1); return; }
    // (the pieces of a loop sequenced by the caller: the numerator is summed here, exchanged by the caller, applied by
    // piece_update_H; the loss was left in loss_xchg by piece_rowpass)
    const bool use8 = fused_w8_stage(c);
    if (use8) c->stat_col8 += 1;
    launch_monitor(c, use8);
    fused_colpass_part(c, c->whole, use8);
    launch_post(c, POST_SUM, &c->whole, 1, kNoLoss, false, use8, true);
}

// one column part of the split layout (the caller exchanges it while the next part computes)
void piece_colpass_part(klnmf_ctx *c, int p) {
    if (c->is_exact() || c->nparts_cfg <= 1) {
        if (p != 0) fail(KLNMF_ERR_ARG, "klnmf_iter_colpass_part: this problem has one part");
        piece_colpass(c);
        return;
    }
    if (p < 0 || p >= c->nparts_cfg) fail(KLNMF_ERR_ARG, "klnmf_iter_colpass_part: no such part");
    if (p == 0) {
        c->piece_use8 = fused_w8_stage(c);
        if (c->piece_use8) c->stat_col8 += 1;
        launch_monitor(c, c->piece_use8);
    }
    fused_colpass_part(c, c->parts[p], c->piece_use8);
    launch_post(c, POST_SUM, &c->parts[p], 1, kNoLoss, false, c->piece_use8, p == c->nparts_cfg - 1);
    c->piece_split = true;
}

void piece_update_H(klnmf_ctx *c) {
    if (c->is_exact()) EXACT_CALL(c, exact_H);
    else if (c->piece_split) launch_post(c, POST_RULE, c->parts, c->nparts_cfg, kNoLoss, false, false, false);
    else launch_post(c, POST_RULE, &c->whole, 1, kNoLoss, false, false, false);
    c->piece_split = false;
}

// column pass + everything behind it of a single-context fit iteration (the row pass has run; its loss partials and the
// tolerance ride in c->pending_loss when the stop rule is deferred to here)
void piece_fit_tail(klnmf_ctx *c) {
    if (c->is_exact()) {
        // the H rule sums the row chunks' slabs itself where that is a few thousand loads per row (the reference's own data
        // scale: one launch less of 7); beyond, one block per row walking the slabs is slower than the wide sum kernel
        // (1000 x 2000, k = 50, 16 slabs: 97 vs 73 us per iteration)
        const bool slabs = !c->sparse && c->hseg_n == 1 && (int64_t)c->nsplit * c->f <= 8192;
        EXACT_CALL(c, exact_N, c->cur ^ 1, !slabs);
        EXACT_CALL(c, exact_H, slabs);
        return;
    }
    const LossArgs la = c->pending_loss;
    c->pending_loss.part = nullptr;
    const bool use8 = fused_w8_stage(c);
    if (use8) c->stat_col8 += 1;
    launch_monitor(c, use8);
    fused_colpass_part(c, c->whole, use8);
    launch_post(c, POST_FULL, &c->whole, 1, la, la.part != nullptr, use8, true);
}

struct HostState {
    DevState st;
};

void fetch_results(klnmf_ctx *c, double *errors_out, int64_t *n_done, int *stopped) {
    DevState hs{};
    HIPCHK(hipMemcpyAsync(&hs, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t nd = hs.n_done;
    if (nd > c->cap) nd = c->cap;
    if (errors_out && nd > 0) {
        HIPCHK(hipMemcpy(errors_out, c->errors, sizeof(double) * nd, hipMemcpyDeviceToHost));
    }
    if (n_done) *n_done = hs.n_done;
    if (stopped) *stopped = hs.stop;
    c->stat_w8_sat = hs.w8_sat_total; c->stat_w8_fallbacks = hs.w8_fallbacks;
    c->stat_q8_sat = hs.q8_sat_total; c->stat_q8_unfixed = hs.q8_unfixed;
    c->stat_mon_checks = hs.mon_checks; c->stat_mon_trips = hs.mon_trips;
    { float m; std::memcpy(&m, &hs.mon_stat_bits, 4); c->stat_mon_max = (double)m; }
    for (int i = 0; i < 3; ++i) { float m; std::memcpy(&m, &hs.mon_dbg[i], 4); c->stat_mon_dbg[i] = (double)m; }
    { float m; std::memcpy(&m, &hs.mon_spread_bits, 4); c->stat_mon_spread = (double)m; }
    // the current W is the one the last *executed* update wrote
    c->cur = (c->loop_start_cur + (int)(hs.n_done & 1)) & 1;
    // ... and so is the current dictionary master: k_post swaps H32 / H32alt per ENQUEUED H rule, the device performed
    // n_done of them (after the stop rule fired the launches return at their first instruction)
    if (c->loop_hswaps > 0 && c->loop_h0 != nullptr) {
        const bool odd = (hs.n_done & 1) != 0;
        c->H32 = odd ? c->loop_h1 : c->loop_h0;
        c->H32alt = odd ? c->loop_h0 : c->loop_h1;
        c->loop_hswaps = 0;
    }
}

// ---------------------------------------------------------------- uploads ---
template <typename S>
void place_block(klnmf_ctx *c, const S *dsrc, int64_t rows, int64_t cols, int64_t ld, int64_t row0,
                 int64_t col0, double scale, const int64_t *row_idx = nullptr) {
    const int64_t total = rows * cols;
    const int grid = grid_for(total, 256, 8192);
    switch (c->prec) {
        case KLNMF_PREC_F64:
            hipLaunchKernelGGL((k_place_V<double, S>), dim3(grid), dim3(256), 0, c->stream,
                               (double *)c->V, c->f, dsrc, rows, cols, ld, row0, col0, scale, row_idx);
            break;
        case KLNMF_PREC_F32:
            hipLaunchKernelGGL((k_place_V<float, S>), dim3(grid), dim3(256), 0, c->stream,
                               (float *)c->V, c->f, dsrc, rows, cols, ld, row0, col0, scale, row_idx);
            break;
        default:
            hipLaunchKernelGGL((k_tile_V<S>), dim3(grid), dim3(256), 0, c->stream,
                               (_Float16 *)c->VtA, c->nrt, c->nct, dsrc, rows, cols,
                               ld, row0, col0, scale * c->v_scale, c->st, row_idx, kEpsRatio * c->v_scale);
            break;
    }
    HIPCHK(hipGetLastError());
    c->v_uploaded = true;
    c->refusals_dirty = true;
}

void check_block(klnmf_ctx *c, int64_t rows, int64_t cols, int64_t ld, int64_t row0, int64_t col0) {
    if (rows < 0 || cols < 0 || row0 < 0 || col0 < 0 || row0 + rows > c->n || col0 + cols > c->f ||
        ld < cols)
        fail(KLNMF_ERR_ARG, "V block out of range");
}

size_t dt_size(int dtype) {
    if (dtype == KLNMF_DT_F32) return 4;
    if (dtype == KLNMF_DT_F64) return 8;
    fail(KLNMF_ERR_ARG, "unknown dtype");
}

// host [rows,cols] (dtype) -> device staging buffer; returns device pointer (freed by caller)
void *stage_to_device(klnmf_ctx *c, const void *src, int dtype, int64_t count) {
    void *d = nullptr;
    const size_t bytes = (size_t)count * dt_size(dtype);
    HIPCHK(hipMalloc(&d, bytes ? bytes : 16));
    hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(d);
        fail(KLNMF_ERR_HIP, std::string("hipMemcpyAsync H2D: ") + hipGetErrorString(e));
    }
    return d;
}

// dense [rows,cols] host array -> device array of the context's element type / padded fp32
void set_matrix(klnmf_ctx *c, const void *src, int dtype, int64_t rows, int64_t cols, void *exact_dst,
                float *fast_dst, int64_t fast_ld, double mul = 1.0) {
    const int64_t count = rows * cols;
    void *d = stage_to_device(c, src, dtype, count);
    const int grid = grid_for(count, 256, 8192);
    if (c->is_exact()) {
        if (c->prec == KLNMF_PREC_F64) {
            if (dtype == KLNMF_DT_F64)
                hipLaunchKernelGGL((k_convert<double, double>), dim3(grid), dim3(256), 0, c->stream, (double *)exact_dst, (const double *)d, count);
            else
                hipLaunchKernelGGL((k_convert<double, float>), dim3(grid), dim3(256), 0, c->stream, (double *)exact_dst, (const float *)d, count);
        } else {
            if (dtype == KLNMF_DT_F64)
                hipLaunchKernelGGL((k_convert<float, double>), dim3(grid), dim3(256), 0, c->stream, (float *)exact_dst, (const double *)d, count);
            else
                hipLaunchKernelGGL((k_convert<float, float>), dim3(grid), dim3(256), 0, c->stream, (float *)exact_dst, (const float *)d, count);
        }
    } else {
        if (dtype == KLNMF_DT_F64)
            hipLaunchKernelGGL((k_place_padded<double>), dim3(grid), dim3(256), 0, c->stream, fast_dst, fast_ld, (const double *)d, rows, cols, mul);
        else
            hipLaunchKernelGGL((k_place_padded<float>), dim3(grid), dim3(256), 0, c->stream, fast_dst, fast_ld, (const float *)d, rows, cols, mul);
    }
    hipError_t e = hipGetLastError();
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    HIPCHK(e);
}

void get_matrix(klnmf_ctx *c, void *dst, int dtype, int64_t rows, int64_t cols, const void *exact_src,
                const float *fast_src, int64_t fast_ld, double mul = 1.0) {
    const int64_t count = rows * cols;
    void *d = nullptr;
    const size_t bytes = (size_t)count * dt_size(dtype);
    HIPCHK(hipMalloc(&d, bytes ? bytes : 16));
    const int grid = grid_for(count, 256, 8192);
    if (c->is_exact()) {
        if (c->prec == KLNMF_PREC_F64) {
            if (dtype == KLNMF_DT_F64)
                hipLaunchKernelGGL((k_convert<double, double>), dim3(grid), dim3(256), 0, c->stream, (double *)d, (const double *)exact_src, count);
            else
                hipLaunchKernelGGL((k_convert<float, double>), dim3(grid), dim3(256), 0, c->stream, (float *)d, (const double *)exact_src, count);
        } else {
            if (dtype == KLNMF_DT_F64)
                hipLaunchKernelGGL((k_convert<double, float>), dim3(grid), dim3(256), 0, c->stream, (double *)d, (const float *)exact_src, count);
            else
                hipLaunchKernelGGL((k_convert<float, float>), dim3(grid), dim3(256), 0, c->stream, (float *)d, (const float *)exact_src, count);
        }
    } else {
        if (dtype == KLNMF_DT_F64)
            hipLaunchKernelGGL((k_gather_padded<double>), dim3(grid), dim3(256), 0, c->stream, (double *)d, fast_src, fast_ld, rows, cols, mul);
        else
            hipLaunchKernelGGL((k_gather_padded<float>), dim3(grid), dim3(256), 0, c->stream, (float *)d, fast_src, fast_ld, rows, cols, mul);
    }
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(e);
    HIPCHK(e2);
    HIPCHK(e3);
}

}  // namespace

// =============================================================== exports ===
template <typename D, typename S>
static void copy_2d(klnmf_ctx *c, D *dst, int64_t dld, const S *src, int64_t sld, int64_t rows, int64_t cols, double mul = 1.0) {
    if (rows * cols == 0) return;
    hipLaunchKernelGGL((k_copy_2d<D, S>), dim3(grid_for(rows * cols, 256, 8192)), dim3(256), 0, c->stream, dst, dld, src, sld,
                       rows, cols, mul);
    HIPCHK(hipGetLastError());
}


template <typename T>
static void distances_on_device(int metric, int64_t na, int64_t nb, int64_t d, const void *A, const void *B, void *out) {
    T *dA = nullptr, *dB = nullptr, *dO = nullptr;
    auto release = [&] { (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dO); };
    try {
        HIPCHK(hipMalloc((void **)&dA, sizeof(T) * (size_t)std::max<int64_t>(1, na * d)));
        HIPCHK(hipMalloc((void **)&dB, sizeof(T) * (size_t)std::max<int64_t>(1, nb * d)));
        HIPCHK(hipMalloc((void **)&dO, sizeof(T) * (size_t)(na * nb)));
        if (d > 0) {
            HIPCHK(hipMemcpy(dA, A, sizeof(T) * (size_t)(na * d), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(dB, B, sizeof(T) * (size_t)(nb * d), hipMemcpyHostToDevice));
        }
        const int64_t pairs = na * nb;
        hipLaunchKernelGGL((k_all_distances<T>), dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, 0, (const T *)dA,
                           (const T *)dB, dO, na, nb, d, metric, kEpsRatio);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(out, dO, sizeof(T) * (size_t)pairs, hipMemcpyDeviceToHost));
    } catch (...) {
        release();
        throw;
    }
    release();
}

template <typename T>
static void matmul_on_device(int64_t m, int64_t n, int64_t kk, const void *A, const void *B, void *C) {
    T *dA = nullptr, *dB = nullptr, *dC = nullptr;
    auto release = [&] { (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); };
    try {
        HIPCHK(hipMalloc((void **)&dA, sizeof(T) * (size_t)(m * kk)));
        HIPCHK(hipMalloc((void **)&dB, sizeof(T) * (size_t)(kk * n)));
        HIPCHK(hipMalloc((void **)&dC, sizeof(T) * (size_t)(m * n)));
        HIPCHK(hipMemcpy(dA, A, sizeof(T) * (size_t)(m * kk), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dB, B, sizeof(T) * (size_t)(kk * n), hipMemcpyHostToDevice));
        EpiStore<T> epi{dC, n};
        dim3 grid((unsigned)((n + GT - 1) / GT), (unsigned)((m + GT - 1) / GT), 1);
        hipLaunchKernelGGL((k_gemm<T, EpiStore<T>>), grid, dim3(256), 0, 0, (int)m, (int)n, (int)kk,
                           (const T *)dA, (int64_t)kk, (int64_t)1, (const T *)dB, (int64_t)n, (int64_t)1,
                           (int)kk + GK, (const DevState *)nullptr, epi);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(C, dC, sizeof(T) * (size_t)(m * n), hipMemcpyDeviceToHost));
    } catch (...) {
        release();
        throw;
    }
    release();
}


extern "C" {

int klnmf_version(void) { return KLNMF_VERSION; }

const char *klnmf_last_error(void) { return g_err.c_str(); }

int klnmf_device_info(int device, char *arch, int arch_len, int *cu_count, uint64_t *hbm_bytes) {
    return guarded([&] {
        hipDeviceProp_t p;
        HIPCHK(hipGetDeviceProperties(&p, device));
        if (arch && arch_len > 0) {
            std::strncpy(arch, p.gcnArchName, arch_len - 1);
            arch[arch_len - 1] = 0;
        }
        if (cu_count) *cu_count = p.multiProcessorCount;
        if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    });
}

int klnmf_create(klnmf_ctx **out, int device, int precision, void *stream) {
    return guarded([&] {
        if (!out) fail(KLNMF_ERR_ARG, "null out pointer");
        if (precision < KLNMF_PREC_F64 || precision > KLNMF_PREC_F16)
            fail(KLNMF_ERR_ARG, "unknown precision mode");
        int ndev = 0;
        HIPCHK(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) fail(KLNMF_ERR_ARG, "no such device");
        HIPCHK(hipSetDevice(device));
        hipDeviceProp_t p;
        HIPCHK(hipGetDeviceProperties(&p, device));
        if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
            fail(KLNMF_ERR_UNSUPP, std::string("this library is built for gfx950 only, device is ") + p.gcnArchName);
        klnmf_ctx *c = new klnmf_ctx();
        c->device = device;
        c->prec = precision;
        c->cu_count = p.multiProcessorCount;
        if (stream == KLNMF_STREAM_DEFAULT) {
            c->stream = nullptr;                    // the default (null) stream
        } else if (stream) {
            c->stream = (hipStream_t)stream;
        } else {
            hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
            if (e != hipSuccess) {
                delete c;
                HIPCHK(e);
            }
            c->own_stream = true;
        }
        *out = c;
    });
}

int klnmf_destroy(klnmf_ctx *c) {
    return guarded([&] {
        if (!c) return;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        c->free_all();
        if (c->comm || c->comm_scratch) {
            try { comm_release(c); } catch (...) {}
        }
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
    });
}

int klnmf_set_problem(klnmf_ctx *c, int64_t n, int64_t f, int64_t k, int64_t cap) {
    return guarded([&] {
        use(c);
        if (n <= 0 || f <= 0 || k <= 0 || cap < 0) fail(KLNMF_ERR_ARG, "n, f, k must be positive");
        if (n > (1LL << 30) || f > (1LL << 30) || k > (1LL << 20))
            fail(KLNMF_ERR_UNSUPP, "dimension too large");
        if (c->is_exact() && n > (int64_t)65535 * GT)
            fail(KLNMF_ERR_UNSUPP, "KLNMF_PREC_F64 / F32: more than 65535 x 64 rows per context (row tiles ride on gridDim.y); "
                                   "shard the rows or use the 16-bit mode");
        HIPCHK(hipStreamSynchronize(c->stream));
        c->free_all();
        c->sw = DevSwitches::read();
        c->Gpart = nullptr; c->row_chunks = 1; c->tail_wg = 0; c->tail_chunks = 1;
        c->Wpart = nullptr; c->wsplit = 1;
        c->hseg_n = 1; c->hpart = nullptr;
        c->n = n; c->f = f; c->k = k; c->cap = cap;
        c->cur = 0;
        c->sparse = false;
        c->v_uploaded = false;
        c->refusals_dirty = true;
        c->v_scale = 1.0;
        c->nnz = 0;
        c->st = (DevState *)c->dalloc(sizeof(DevState));
        c->errors = (double *)c->dalloc(sizeof(double) * (cap > 0 ? cap : 1));
        c->loss_xchg = (double *)c->dalloc(sizeof(double) * 2);
        if (c->is_exact()) {
            const size_t es = c->esize();
            c->V = c->dalloc((size_t)n * f * es);
            c->Q = c->dalloc((size_t)n * f * es);
            c->W[0] = c->dalloc((size_t)n * k * es);
            c->W[1] = c->dalloc((size_t)n * k * es);
            c->H = c->dalloc((size_t)k * f * es);
            // 64 x 64 output tiles (k_gemm).  Measured (profiles/r04_exact_modes.txt): with the register prefetch they win over
            // 128 x 128 tiles at every shape tried (2000 x 4096, k = 200, fp64: 440 us per iteration against 455, 537 before):
            // four waves per SIMD hide more than the halved LDS traffic gains.
            auto tiles_of = [](int64_t M, int64_t N, int64_t TL) { return ((M + TL - 1) / TL) * ((N + TL - 1) / TL); };
            c->q_tt = 4;
            const int64_t smax = (n + 63) / 64;          // at least four contraction steps per chunk
            auto n_split = [&](int tt) {
                const int64_t tiles = tiles_of(k, f, 16 * tt);
                int64_t s = (4 * (int64_t)c->cu_count + tiles - 1) / tiles;
                if (s > smax) s = smax;
                if (s < 1) s = 1;
                return s;
            };
            c->n_tt = 4;
            int64_t s = n_split(c->n_tt);
            int64_t chunk = (n + s - 1) / s;
            chunk = (chunk + GK - 1) / GK * GK;
            s = (n + chunk - 1) / chunk;
            c->nsplit = (int)s;
            c->kchunk = (int)chunk;
            c->Npart = c->dalloc((size_t)s * k * f * es);
            c->hseg = 4096;          // dictionary rows of 16 384 columns and more: the H rule in segments (exact_H)
            c->hseg_n = f >= 16384 ? (int)((f + c->hseg - 1) / c->hseg) : 1;
            c->hpart = c->hseg_n > 1 ? (double *)c->dalloc(sizeof(double) * (size_t)k * c->hseg_n) : nullptr;
            c->numer = c->dalloc((size_t)k * f * es);
            // W rule: n*k/4096 output tiles, each contracting over all of f.  With fewer tiles than CUs split f so that
            // the grid covers the chip about twice.
            {
                auto w_split = [&](int tt) {
                    const int64_t wt = tiles_of(k, n, 16 * tt);
                    int64_t w = wt < c->cu_count ? (2 * (int64_t)c->cu_count + wt - 1) / wt : 1;
                    return std::min<int64_t>(w, (f + 4 * GK - 1) / (4 * GK));
                };
                c->w_tt = 4;
                int64_t ws = w_split(c->w_tt);
                ws = std::min<int64_t>(ws, (f + 4 * GK - 1) / (4 * GK));
                while (ws > 1 && ws * n * k * (int64_t)es > ((int64_t)256 << 20)) --ws;
                int64_t wch = (f + ws - 1) / ws;
                wch = (wch + GK - 1) / GK * GK;
                ws = (f + wch - 1) / wch;
                c->wsplit = (int)ws;
                c->wchunk = (int)wch;
                if (ws > 1) c->Wpart = c->dalloc((size_t)ws * n * k * es);
            }
            c->loss_part_count = ((f + GT - 1) / GT) * ((n + GT - 1) / GT);
            c->loss_part = (double *)c->dalloc(sizeof(double) * c->loss_part_count);
        } else {
            c->KT = (int)((k + 31) / 32);
            c->ks = (int)((k + 15) / 16);
            c->big = false;
            if (k > 512) fail(KLNMF_ERR_UNSUPP, "k > 512 runs in KLNMF_PREC_F32 / F64 (the 16-bit MFMA kernels cover k <= 512)");
            if (c->KT >= 8) {
                // 224 < k <= 512: 4-wave workgroups of the row pass (whole register file per wave, FUSED order) and the
                // component-split column passes; component tiles in pairs, the W.H contraction over all of them
                c->big = true;
                c->KT = 2 * (int)((k + 63) / 64);
                c->ks = 2 * c->KT;
            }
            c->KP = 32 * c->KT;
            // both passes work on 64-row / 64-column stages: pad to 64 (zero padding is inert)
            c->n_pad = (n + 63) / 64 * 64;
            c->f_pad = (f + 127) / 128 * 128;            // the row pass walks 4 column tiles per loop body
            c->nrt = (int)(c->n_pad / 32);
            c->nct = (int)(c->f_pad / 32);
            c->nct_used = (int)((f + 63) / 64 * 2);      // column tiles that hold data (column pass)
            c->v_scale = 1.0;
            c->v_uploaded = false;
            c->refusals_dirty = true;
            const int total_stages = c->nrt / kStageRowTiles;
            // the fp16 W images are streamed by global_load_lds in whole 8 KiB rounds, i.e. a stage's copy reads on into the rows
            // behind it: pad the tail by what ONE copy covers.  (64 rows until round 4: at KP = 32 a row is 64 bytes and a copy 128
            // rows -- the last stage read 2 KiB past the image; found by scripts/shape_fuzz.py as a memory access fault at
            // 16 305 x 28, k = 8, where the image is exactly 1 MiB and ends on a mapping boundary.)
            const int64_t copy_rows = (colq_w_area(c->KP) + (int64_t)w_ld(c->KP) * 2 - 1) / ((int64_t)w_ld(c->KP) * 2);
            c->w_rows = (int64_t)total_stages * 32 * kStageRowTiles + std::max<int64_t>(64, copy_rows);
            const size_t vbytes = (size_t)c->nrt * c->nct * 1024 * 2;
            // fp8 ratio tiles: only the H numerator -- a sum over all rows -- sees their 3-bit significands; its relative
            // error falls like 0.036 sqrt(2 / n), so they are used from 32 769 / 65 536 rows per context on (row_chunks_possible_q8;
            // KLNMF_QTILE = 8 / 16 forces either), where the bytes matter
            const bool col8_off = c->sw.col8 == 0;
            const bool q8_kt = !c->big || !col8_off;      // (k > 224: fp8 tiles only with the fp8 x fp8 column pass)
            // ... and from one column tile of data on: below that the tiles are mostly padding (nothing to gain), and a handful of
            // columns is fitted so exactly that the loss itself goes to 0 (the 16-bit mode's own operand rounding then shows)
            c->q8_ok = q8_kt && f >= 32 && c->row_chunks_possible_q8(n, c->big);
            if (c->sw.qtile != 0) c->q8_ok = q8_kt && c->sw.qtile == 8;
            c->q8_loop = false;
            c->iter_in_loop = 0;
            c->v_max = 0.0;
            c->ne_ok = c->q8_ok && !c->big;      // (q8_ok: enough rows for fp8 ratio tiles -- where the NE kernels exist)
            c->VtA = c->dalloc(vbytes, false);
            fill_v_tiles(c, c->VtA, vbytes);
            c->Qt = (unsigned char *)c->dalloc((size_t)c->nrt * c->nct * kQTile);      // (fp8 tiles use the first half of the buffer)
            c->W8 = nullptr; c->w8s = nullptr; c->w8_meas = false;
            c->q8_list = c->q8_ok ? (uint2 *)c->dalloc(sizeof(uint2) * kQ8ListCap) : nullptr;
            // fp8 x fp8 column pass (e4m3 image of W_new): where the H-numerator product is worth the conversion launch --
            // k > 96 and 65 536 rows or more; below that the f16-operand column pass reads the fp8 tiles (C2, k = 50: 0.053 ms
            // against 0.050 + 0.03 ms of conversions; profiles/r03_c2_schedules.txt)
            const bool col8_size = c->big || (c->KT >= 4 && n >= 65536) || c->sw.col8 >= 1;
            if (c->q8_ok && !col8_off && col8_size) {
                c->W8 = (unsigned char *)c->dalloc((size_t)(c->n_pad + 64) * w8_ld(c->KP) + 65536);
                // who writes the e4m3 image: the conversion kernel behind the row pass (default) or the row pass's W rule itself
                // (KLNMF_COL8=2, k <= 224: one pass over W less and no conversion launch -- +0.6 % at n = 10^6, -0.8 % on a
                // 125 000-row shard where the tail weighs more: profiles/r03_ab_w8_from_w_rule.txt; not the default)
                c->w8_tail = !c->big && c->sw.col8 == 2;
                c->w8s = (float *)c->dalloc((size_t)c->KP * 4);
                const std::vector<float> unit8((size_t)c->KP, 256.f);
                HIPCHK(hipMemcpyAsync(c->w8s, unit8.data(), unit8.size() * 4, hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
            }
            for (int i = 0; i < 2; ++i) {
                c->W32[i] = (float *)c->dalloc((size_t)c->n_pad * c->KP * 4);
                c->Wb[i] = (opnd_t *)c->dalloc((size_t)c->w_rows * w_ld(c->KP) * 2);
            }
            c->H32 = (float *)c->dalloc((size_t)c->KP * c->f_pad * 4);
            c->Ht4 = (opnd_t *)c->dalloc((size_t)c->nct * h4_tile_bytes(c->KP) + kObj4);
            // eps through a pad component (k_update_pack_H): the row pass's W epilogue keeps the carrier column at 2^-10; needs
            // a spare component inside the MFMA-1 contraction range
            c->kc_shape = (k < 16 * c->ks && c->sw.eps_pad) ? (int)k : -1;
            choose_eps_carrier(c);
            c->hsum = (double *)c->dalloc((size_t)c->KP * 8);
            c->tcur = (float *)c->dalloc((size_t)c->KP * 4);
            c->t_hs = (float *)c->dalloc((size_t)c->KP * 4);
            c->t_unit = (float *)c->dalloc((size_t)c->KP * 4);
            c->wmax = (unsigned *)c->dalloc((size_t)c->KP * 4);
            c->images_measured = false;
            {
                const std::vector<float> unit((size_t)c->KP, kOpScaleW);      // until a dictionary is packed (k_update_pack_H)
                for (float *t : {c->tcur, c->t_hs, c->t_unit})
                    HIPCHK(hipMemcpyAsync(t, unit.data(), unit.size() * 4, hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
            }
            // column pass decomposition: column blocks of 8 tiles x row chunks; keep the grid a
            // multiple of 8 (XCD remap) and close to a multiple of the CU count
            const int ctw = c->big ? kWavesPerWG / 2 : kWavesPerWG;      // column tiles per workgroup (colq.hip.h, KSPLIT)
            c->ncb = (c->nct_used + ctw - 1) / ctw;
            auto chunks_for = [&](int ncb) {      // row chunks of a column pass over `ncb` column blocks: the grid fills the chip once
                int nch = 8;                      // (one workgroup is resident per CU; two per CU measured 1-3 % slower)
                while ((int64_t)nch * ncb < c->cu_count && nch * 2 <= total_stages) nch += 8;
                while (nch > 8 && ((int64_t)nch * ncb) % c->cu_count != 0 &&
                       (int64_t)(nch - 8) * ncb >= c->cu_count) nch -= 8;
                if (nch > total_stages) nch = total_stages > 0 ? ((total_stages + 7) / 8) * 8 : 8;
                return nch;
            };
            const int nch = chunks_for(c->ncb);
            c->nchunks = nch;
            c->stages_per_chunk = (total_stages + nch - 1) / nch;
            c->whole = klnmf_ctx::PartCfg{0, c->ncb, 0, c->nct_used, 0, (int)f, (int)c->f_pad, nch, c->stages_per_chunk, 0, 0};
            // column parts for loops on a communicator (overlap of the numerator's all-reduce with the column pass)
            c->nparts_cfg = std::min(std::min(kPostMaxParts, std::max(1, c->sw.comm_parts)), c->ncb);
            int64_t split_numer = 0, split_slabs = 0;
            if (c->nparts_cfg > 1) {
                for (int p = 0; p < c->nparts_cfg; ++p) {
                    klnmf_ctx::PartCfg &q = c->parts[p];
                    q.cb0 = (int)((int64_t)c->ncb * p / c->nparts_cfg);
                    q.ncb = (int)((int64_t)c->ncb * (p + 1) / c->nparts_cfg) - q.cb0;
                    q.ct0 = q.cb0 * ctw;
                    q.nct = std::min(c->nct_used - q.ct0, q.ncb * ctw);
                    q.col0 = q.ct0 * 32;
                    q.ld = q.ncb * ctw * 32;
                    q.ncols = (int)std::min<int64_t>(f - q.col0, q.ld);
                    q.nchunks = chunks_for(q.ncb);
                    q.spc = (total_stages + q.nchunks - 1) / q.nchunks;
                    q.numer_off = split_numer;
                    q.slab_off = split_slabs;
                    split_numer += (int64_t)c->KP * q.ld;
                    split_slabs += (int64_t)q.nchunks * c->KP * q.ld;
                }
            }
            c->NpartF = (float *)c->dalloc((size_t)std::max<int64_t>((int64_t)nch * c->KP * c->f_pad, split_slabs) * 4);
            c->numerF = (float *)c->dalloc((size_t)std::max<int64_t>((int64_t)c->KP * c->f_pad, split_numer) * 4);
            c->H32alt = (float *)c->dalloc((size_t)c->KP * c->f_pad * 4);
            c->loop_hswaps = 0;
            c->w8tab = nullptr; c->w8s_next = nullptr; c->conv_ran = false;
            if (c->W8) {
                c->w8tab = (unsigned *)c->dalloc((size_t)kW8TabRows * c->KP * 4);      // (zero-filled)
                c->w8s_next = (float *)c->dalloc((size_t)c->KP * 4);
                HIPCHK(hipMemcpyAsync(c->w8s_next, c->w8s, (size_t)c->KP * 4, hipMemcpyDeviceToDevice, c->stream));
            }
            monitor_setup(c);
            // Column-split update pass: with fewer than half as many 8-wave workgroups as CUs (n < ~32 000 rows; the
            // reference's own data sets have 10^2..10^3) split every row block's columns over blockIdx.y so that the grid
            // fills the chip once.  KLNMF_ROW_SPLIT = 0 / N (development switch) forces it off / to N chunks.
            c->row_chunks = 1;
            c->row_ct_chunk = c->nct;
            if (!c->big && !c->q8_ok) {
                const int nwg = (c->nrt + kWaves4 - 1) / kWaves4;
                int want = (2 * nwg <= c->cu_count) ? c->cu_count / nwg : 1;
                if (c->sw.row_split >= 0) want = std::max(1, c->sw.row_split);
                want = std::min(want, c->nct / 4);
                const int64_t slab_bytes = (int64_t)c->nrt * 32 * c->KP * 4;
                while (want > 1 && want * slab_bytes > (int64_t)256 << 20) --want;
                if (want > 1) {
                    c->row_ct_chunk = 4 * ((c->nct / 4 + want - 1) / want);
                    c->row_chunks = (c->nct + c->row_ct_chunk - 1) / c->row_ct_chunk;
                }
                if (c->row_chunks > 1) c->Gpart = (float *)c->dalloc((size_t)c->row_chunks * slab_bytes);
            }
            // Hybrid update pass: more workgroups than CUs, and a last partial round of at most half the CUs (one
            // workgroup per CU: 254 registers).  Its workgroups are split into as many column chunks as fill the chip
            // once (n = 10^6: 67 workgroups x 3 chunks; 90 000 rows: 96 x 2).  KLNMF_ROW_TAIL = 0 (development switch): off.
            c->tail_wg = 0; c->tail_chunks = 1; c->tail_ct_chunk = c->nct;
            if (!c->big && c->row_chunks == 1) {
                const int nwg = (c->nrt + kWaves4 - 1) / kWaves4;
                const int rem = nwg % c->cu_count;
                int want = (nwg > c->cu_count && rem > 0) ? c->cu_count / rem : 1;
                if (c->sw.row_tail >= 0) want = std::min(want, std::max(1, c->sw.row_tail));
                want = std::min(std::min(want, 4), c->nct / 4);
                if (want > 1) {
                    c->tail_ct_chunk = 4 * ((c->nct / 4 + want - 1) / want);
                    c->tail_chunks = (c->nct + c->tail_ct_chunk - 1) / c->tail_ct_chunk;
                    if (c->tail_chunks > 1) {
                        c->tail_wg = rem;
                        c->Gpart = (float *)c->dalloc((size_t)c->tail_chunks * (c->nrt - c->tail_rt0()) * 32 * c->KP * 4);
                    }
                }
            }
            c->loss_part2 = (double2 *)c->dalloc(sizeof(double2) * std::max<int64_t>(c->loss_parts(), (int64_t)c->nrt * c->row_chunks));
        }
        reset_state(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        c->have_problem = true;
    });
}

int klnmf_release_problem(klnmf_ctx *c) {
    return guarded([&] {
        use(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        c->free_all();              // device blocks back to the per-process cache (large ones to the driver)
        c->profiling = false;
        c->images_measured = false;
        c->ratio_eps = kEpsRatio;
    });
}

int klnmf_set_problem_sparse(klnmf_ctx *c, int64_t n, int64_t f, int64_t k, int64_t cap, int64_t nnz) {
    return guarded([&] {
        use(c);
        if (!c->is_exact()) fail(KLNMF_ERR_UNSUPP, "CSR input runs in the exact modes (KLNMF_PREC_F64 / F32); densify for the bf16 kernels");
        if (n <= 0 || f <= 0 || k <= 0 || cap < 0 || nnz < 0) fail(KLNMF_ERR_ARG, "n, f, k must be positive, nnz >= 0");
        if (n > (1LL << 30) || f > (1LL << 30) || k > (1LL << 20)) fail(KLNMF_ERR_UNSUPP, "dimension too large");
        HIPCHK(hipStreamSynchronize(c->stream));
        c->free_all();
        c->Gpart = nullptr; c->row_chunks = 1; c->tail_wg = 0; c->tail_chunks = 1;
        c->Wpart = nullptr; c->wsplit = 1;
        c->n = n; c->f = f; c->k = k; c->cap = cap;
        c->cur = 0;
        c->sparse = true;
        c->v_uploaded = false;
        c->refusals_dirty = true;
        c->v_scale = 1.0;
        c->nnz = nnz;
        const size_t es = c->esize();
        c->st = (DevState *)c->dalloc(sizeof(DevState));
        c->errors = (double *)c->dalloc(sizeof(double) * (cap > 0 ? cap : 1));
        c->loss_xchg = (double *)c->dalloc(sizeof(double) * 2);
        c->V = nullptr; c->Q = nullptr; c->Npart = nullptr; c->loss_part = nullptr;
        c->W[0] = c->dalloc((size_t)n * k * es);
        c->W[1] = c->dalloc((size_t)n * k * es);
        c->H = c->dalloc((size_t)k * f * es);
        c->HT = c->dalloc((size_t)k * f * es);
        c->numer = c->dalloc((size_t)k * f * es);
        c->sp_indptr = (int64_t *)c->dalloc(sizeof(int64_t) * (n + 1));
        c->sp_indices = (int64_t *)c->dalloc(sizeof(int64_t) * (nnz > 0 ? nnz : 1));
        c->csc_indptr = (int64_t *)c->dalloc(sizeof(int64_t) * (f + 1));
        c->csc_rows = (int64_t *)c->dalloc(sizeof(int64_t) * (nnz > 0 ? nnz : 1));
        c->csc_perm = (int64_t *)c->dalloc(sizeof(int64_t) * (nnz > 0 ? nnz : 1));
        c->sp_data = c->dalloc((size_t)(nnz > 0 ? nnz : 1) * es);
        c->sp_q = c->dalloc((size_t)(nnz > 0 ? nnz : 1) * es);
        c->sp_row_loss = (double *)c->dalloc(sizeof(double) * n);
        c->sp_nblk = (n + kSpColsumRows - 1) / kSpColsumRows;
        // dictionary rows of 16 384 columns and more: the H rule and the loss term's row sums in segments of 4096
        c->hseg = 4096;
        c->hseg_n = f >= 16384 ? (int)((f + c->hseg - 1) / c->hseg) : 1;
        c->hpart = c->hseg_n > 1 ? (double *)c->dalloc(sizeof(double) * (size_t)k * c->hseg_n) : nullptr;
        c->sp_wpart = (double *)c->dalloc(sizeof(double) * c->sp_nblk * k);
        c->sp_prod = (double *)c->dalloc(sizeof(double) * k);
        // blocks for the L2 (sparseb.hip.h): kSpBlockBytes of H^T per column block / of W per row block, as many blocks as the
        // slabs of partial sums allow (1 GiB each)
        c->sp_blocked = k <= 512 && nnz > 0 && n < ((int64_t)1 << 31) && f < ((int64_t)1 << 31) && nnz < ((int64_t)1 << 31);
        if (c->sp_blocked) {
            const int64_t per = std::max<int64_t>(64, (kSpBlockBytes / (int64_t)(k * es)) / 64 * 64);
            const int64_t slab_cap = (int64_t)1 << 30;
            // Measured (round 5, 20 000 x 110 000, 0.5 %, k = 50, fp64; profiles/r05_sparse_blocks.txt): 1, 2, 4, 8 or 15 column blocks and
            // 1 or 3 row blocks give the same iteration within 5 % -- the passes are bound by the latency chain of a wave's trips
            // (index load -> gathers -> reduction), not by where the gathered rows come from.  One block each unless asked
            // (KLNMF_SP_CB / KLNMF_SP_RB, development switches; `per` = rows of a block that fit the L2).
            int64_t cb = 1, rb = 1;
            (void)per;
            const DevSwitches sw = DevSwitches::read();
            if (sw.sp_cb > 0) cb = std::min<int64_t>(sw.sp_cb, (f + 63) / 64);
            if (sw.sp_rb > 0) rb = std::min<int64_t>(sw.sp_rb, (n + 63) / 64);
            cb = std::max<int64_t>(1, std::min(cb, slab_cap / std::max<int64_t>(1, n * k * (int64_t)es)));
            rb = std::max<int64_t>(1, std::min(rb, slab_cap / std::max<int64_t>(1, f * k * (int64_t)es)));
            cb = std::min<int64_t>(cb, ((int64_t)1 << 31) / std::max<int64_t>(1, n) - 1);      // (blocks x rows ride on gridDim.x)
            rb = std::min<int64_t>(rb, ((int64_t)1 << 31) / std::max<int64_t>(1, f) - 1);
            if (cb < 1 || rb < 1) c->sp_blocked = false;
            c->sp_cb = (int)cb; c->sp_rb = (int)rb;
            c->sp_cb_cols = (f + cb - 1) / cb; c->sp_rb_rows = (n + rb - 1) / rb;
        }
        if (c->sp_blocked) {
            c->sp_idx32 = (int *)c->dalloc(sizeof(int) * nnz);
            c->csc_rows32 = (int *)c->dalloc(sizeof(int) * nnz);
            c->csc_perm32 = (int *)c->dalloc(sizeof(int) * nnz);
            c->sp_blkptr = (int64_t *)c->dalloc(sizeof(int64_t) * n * (c->sp_cb + 1));
            c->csc_blkptr = (int64_t *)c->dalloc(sizeof(int64_t) * f * (c->sp_rb + 1));
            c->sp_loss_part = (double *)c->dalloc(sizeof(double) * (size_t)c->sp_cb * n);
            c->sp_G = c->dalloc((size_t)c->sp_cb * n * k * es);
            c->sp_NT = c->dalloc((size_t)c->sp_rb * f * k * es);
            c->sp_bad = (int *)c->dalloc(sizeof(int));
        }
        reset_state(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        c->have_problem = true;
    });
}

int klnmf_upload_csr(klnmf_ctx *c, int dtype, const int64_t *indptr, const int64_t *indices, const void *data,
                     const int64_t *csc_indptr, const int64_t *csc_rows, const int64_t *csc_perm) {
    return guarded([&] {
        need_problem(c);
        if (!c->sparse) fail(KLNMF_ERR_ARG, "klnmf_upload_csr needs klnmf_set_problem_sparse");
        if (!indptr || !csc_indptr || (c->nnz > 0 && (!indices || !data || !csc_rows || !csc_perm)))
            fail(KLNMF_ERR_ARG, "null pointer");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (indptr[0] != 0 || indptr[c->n] != c->nnz || csc_indptr[0] != 0 || csc_indptr[c->f] != c->nnz)
            fail(KLNMF_ERR_ARG, "index pointers do not match n, f, nnz");
        HIPCHK(hipMemcpyAsync(c->sp_indptr, indptr, sizeof(int64_t) * (c->n + 1), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->csc_indptr, csc_indptr, sizeof(int64_t) * (c->f + 1), hipMemcpyHostToDevice, c->stream));
        if (c->nnz > 0) {
            HIPCHK(hipMemcpyAsync(c->sp_indices, indices, sizeof(int64_t) * c->nnz, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->csc_rows, csc_rows, sizeof(int64_t) * c->nnz, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->csc_perm, csc_perm, sizeof(int64_t) * c->nnz, hipMemcpyHostToDevice, c->stream));
            // values: through the dense setter (dtype conversion) as a 1 x nnz matrix
            set_matrix(c, data, dtype, 1, c->nnz, c->sp_data, nullptr, 0);
        }
        if (c->sp_blocked) {
            // block pointers by binary search in the sorted rows / columns, int32 copies of the indices (sparseb.hip.h)
            HIPCHK(hipMemsetAsync(c->sp_bad, 0, sizeof(int), c->stream));
            hipLaunchKernelGGL(k_spb_blkptr, dim3(grid_for(c->n * (c->sp_cb + 1), 256, 1 << 20)), dim3(256), 0, c->stream,
                               (const int64_t *)c->sp_indptr, (const int64_t *)c->sp_indices, c->n, c->sp_cb, c->sp_cb_cols, c->sp_blkptr, c->sp_bad);
            hipLaunchKernelGGL(k_spb_blkptr, dim3(grid_for(c->f * (c->sp_rb + 1), 256, 1 << 20)), dim3(256), 0, c->stream,
                               (const int64_t *)c->csc_indptr, (const int64_t *)c->csc_rows, c->f, c->sp_rb, c->sp_rb_rows, c->csc_blkptr, c->sp_bad);
            hipLaunchKernelGGL(k_spb_narrow, dim3(grid_for(c->nnz, 256, 8192)), dim3(256), 0, c->stream, (const int64_t *)c->sp_indices, c->sp_idx32, c->nnz);
            hipLaunchKernelGGL(k_spb_narrow, dim3(grid_for(c->nnz, 256, 8192)), dim3(256), 0, c->stream, (const int64_t *)c->csc_rows, c->csc_rows32, c->nnz);
            hipLaunchKernelGGL(k_spb_narrow, dim3(grid_for(c->nnz, 256, 8192)), dim3(256), 0, c->stream, (const int64_t *)c->csc_perm, c->csc_perm32, c->nnz);
            HIPCHK(hipGetLastError());
            int bad = 0;
            HIPCHK(hipMemcpyAsync(&bad, c->sp_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (bad) fail(KLNMF_ERR_ARG, "klnmf_upload_csr: the column indices of every row (and the rows of every column in the CSC arrays) must be sorted");
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        c->v_uploaded = true;
        c->refusals_dirty = true;
    });
}

int klnmf_get_Q_values(klnmf_ctx *c, void *dst, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!c->sparse) fail(KLNMF_ERR_ARG, "klnmf_get_Q_values needs a CSR problem");
        if (!dst && c->nnz > 0) fail(KLNMF_ERR_ARG, "null destination");
        if (c->nnz > 0) get_matrix(c, dst, dtype, 1, c->nnz, c->sp_q, nullptr, 0);
    });
}

int klnmf_set_v_max(klnmf_ctx *c, double vmax) {
    return guarded([&] {
        need_problem(c);
        if (!(vmax >= 0) || !std::isfinite(vmax)) fail(KLNMF_ERR_ARG, "vmax must be finite and >= 0");
        if (c->v_uploaded) fail(KLNMF_ERR_ARG, "klnmf_set_v_max must precede the first upload");
        if (c->is_exact() || vmax == 0) {
            c->v_scale = 1.0;
            if (!c->is_exact()) choose_eps_carrier(c);
            return;
        }
        int e = 0;
        (void)std::frexp(vmax, &e);             // vmax = m * 2^e, m in [0.5, 1)
        c->v_scale = std::ldexp(1.0, 15 - e);   // c * vmax in [2^14, 2^15)
        c->v_max = vmax;
        choose_eps_carrier(c);
        if (c->kc >= 0) fast_pack_H(c);      // the eps row of the dictionary images is in scaled units
    });
}

int klnmf_reset_V(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        if (c->sparse) fail(KLNMF_ERR_ARG, "klnmf_reset_V: CSR problems are re-uploaded whole (klnmf_upload_csr)");
        // the upload kernels ACCUMULATE sum(V as stored), the storage-rounding correction and the overflow count: a second
        // upload into a live context would count a block twice.  Clear the matrix and the three counters.
        if (c->is_exact()) {
            HIPCHK(hipMemsetAsync(c->V, 0, (size_t)c->n * c->f * c->esize(), c->stream));
        } else {
            fill_v_tiles(c, c->VtA, (size_t)c->nrt * c->nct * 1024 * 2);
        }
        HIPCHK(hipMemsetAsync(&c->st->sum_x, 0, sizeof(double) * 4, c->stream));         // sum_x, corr_c, corr_eps, nnz_x
        HIPCHK(hipMemsetAsync(&c->st->v_overflow, 0, sizeof(int), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->v_uploaded = false;
        c->refusals_dirty = true;
    });
}

int klnmf_upload_V(klnmf_ctx *c, const void *src, int dtype, int64_t rows, int64_t cols, int64_t ld,
                   int64_t row0, int64_t col0, double scale) {
    return guarded([&] {
        need_problem(c);
        if (!src) fail(KLNMF_ERR_ARG, "null source");
        check_block(c, rows, cols, ld, row0, col0);
        const size_t es = dt_size(dtype);
        // stream the block through a bounded device staging buffer
        int64_t rows_per = (int64_t)((256ull << 20) / (es * (size_t)ld));
        if (rows_per < 1) rows_per = 1;
        if (rows_per > rows) rows_per = rows;
        void *d = nullptr;
        HIPCHK(hipMalloc(&d, (size_t)rows_per * ld * es + 16));
        try {
            for (int64_t r0 = 0; r0 < rows; r0 += rows_per) {
                const int64_t rr = std::min(rows_per, rows - r0);
                const size_t bytes = ((size_t)(rr - 1) * ld + cols) * es;
                HIPCHK(hipMemcpyAsync(d, (const char *)src + (size_t)r0 * ld * es, bytes,
                                      hipMemcpyHostToDevice, c->stream));
                if (dtype == KLNMF_DT_F64)
                    place_block<double>(c, (const double *)d, rr, cols, ld, row0 + r0, col0, scale);
                else
                    place_block<float>(c, (const float *)d, rr, cols, ld, row0 + r0, col0, scale);
                HIPCHK(hipStreamSynchronize(c->stream));
            }
        } catch (...) {
            (void)hipFree(d);
            throw;
        }
        (void)hipFree(d);
    });
}

int klnmf_upload_V_device(klnmf_ctx *c, const float *dsrc, int64_t rows, int64_t cols, int64_t ld,
                          int64_t row0, int64_t col0, double scale) {
    return guarded([&] {
        need_problem(c);
        if (!dsrc) fail(KLNMF_ERR_ARG, "null source");
        check_block(c, rows, cols, ld, row0, col0);
        place_block<float>(c, dsrc, rows, cols, ld, row0, col0, scale);
    });
}

int klnmf_upload_V_device_rows(klnmf_ctx *c, const float *dsrc, const int64_t *drow_idx, int64_t rows,
                               int64_t cols, int64_t ld, int64_t row0, int64_t col0, double scale) {
    return guarded([&] {
        need_problem(c);
        if (!dsrc || !drow_idx) fail(KLNMF_ERR_ARG, "null source");
        check_block(c, rows, cols, ld, row0, col0);
        place_block<float>(c, dsrc, rows, cols, ld, row0, col0, scale, drow_idx);
    });
}

int klnmf_set_H(klnmf_ctx *c, const void *src, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!src) fail(KLNMF_ERR_ARG, "null source");
        if (!c->is_exact()) HIPCHK(hipMemsetAsync(c->H32, 0, (size_t)c->KP * c->f_pad * 4, c->stream));
        set_matrix(c, src, dtype, c->k, c->f, c->H, c->H32, c->f_pad);
        if (!c->is_exact()) {
            fast_pack_H(c);          // hs-based scales (also leaves them in t_hs)
            measure_and_pack(c);        // the W that is there (zeros, W0 of another dictionary, a klnmf_set_W) goes with it
        }
    });
}

// ---- device-resident operands (next-row N1: the transforms of an evaluation keep dictionary, coefficients and
// reconstructions on the GPU).  Pointers are DEVICE memory of the context's device; row strides in elements.
int klnmf_set_H_device(klnmf_ctx *c, const void *dsrc, int dtype, int64_t ld, int64_t col0, int64_t ncols, int last) {
    return guarded([&] {
        need_problem(c);
        if (!dsrc) fail(KLNMF_ERR_ARG, "null source");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (col0 < 0 || ncols < 0 || col0 + ncols > c->f || ld < ncols) fail(KLNMF_ERR_ARG, "klnmf_set_H_device: column block out of range");
        if (c->sparse) fail(KLNMF_ERR_UNSUPP, "klnmf_set_H_device: dense problems");
        const bool f64 = dtype == KLNMF_DT_F64;
        // the first block of a dictionary (col0 = 0) starts from zeros, as klnmf_set_H does: a pooled or re-used context
        // must not keep padding rows / columns of the previous dictionary in its images
        if (col0 == 0 && !c->is_exact()) HIPCHK(hipMemsetAsync(c->H32, 0, (size_t)c->KP * c->f_pad * 4, c->stream));
        if (c->prec == KLNMF_PREC_F64) {
            if (f64) copy_2d(c, (double *)c->H + col0, c->f, (const double *)dsrc, ld, c->k, ncols);
            else copy_2d(c, (double *)c->H + col0, c->f, (const float *)dsrc, ld, c->k, ncols);
        } else if (c->prec == KLNMF_PREC_F32) {
            if (f64) copy_2d(c, (float *)c->H + col0, c->f, (const double *)dsrc, ld, c->k, ncols);
            else copy_2d(c, (float *)c->H + col0, c->f, (const float *)dsrc, ld, c->k, ncols);
        } else {
            if (f64) copy_2d(c, c->H32 + col0, c->f_pad, (const double *)dsrc, ld, c->k, ncols);
            else copy_2d(c, c->H32 + col0, c->f_pad, (const float *)dsrc, ld, c->k, ncols);
        }
        if (last) {
            if (!c->is_exact()) {
                fast_pack_H(c);
                measure_and_pack(c);
            }
            HIPCHK(hipStreamSynchronize(c->stream));      // the caller's buffer may go away
        }
    });
}

int klnmf_get_W_device(klnmf_ctx *c, void *ddst, int dtype, int64_t ld) {
    return guarded([&] {
        need_problem(c);
        if (!ddst) fail(KLNMF_ERR_ARG, "null destination");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (ld < c->k) fail(KLNMF_ERR_ARG, "klnmf_get_W_device: row stride shorter than k");
        const bool f64 = dtype == KLNMF_DT_F64;
        if (c->prec == KLNMF_PREC_F64) {
            if (f64) copy_2d(c, (double *)ddst, ld, (const double *)c->W[c->cur], c->k, c->n, c->k);
            else copy_2d(c, (float *)ddst, ld, (const double *)c->W[c->cur], c->k, c->n, c->k);
        } else if (c->prec == KLNMF_PREC_F32) {
            if (f64) copy_2d(c, (double *)ddst, ld, (const float *)c->W[c->cur], c->k, c->n, c->k);
            else copy_2d(c, (float *)ddst, ld, (const float *)c->W[c->cur], c->k, c->n, c->k);
        } else {
            if (f64) copy_2d(c, (double *)ddst, ld, (const float *)c->W32[c->cur], (int64_t)c->KP, c->n, c->k, 1.0 / c->v_scale);
            else copy_2d(c, (float *)ddst, ld, (const float *)c->W32[c->cur], (int64_t)c->KP, c->n, c->k, 1.0 / c->v_scale);
        }
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_upload_V_device_rows_dt(klnmf_ctx *c, const void *dsrc, int dtype, const int64_t *drow_idx, int64_t rows,
                                  int64_t cols, int64_t ld, int64_t row0, int64_t col0, double scale) {
    return guarded([&] {
        need_problem(c);
        if (!dsrc) fail(KLNMF_ERR_ARG, "null source");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        check_block(c, rows, cols, ld, row0, col0);
        if (dtype == KLNMF_DT_F64) place_block<double>(c, (const double *)dsrc, rows, cols, ld, row0, col0, scale, drow_idx);
        else place_block<float>(c, (const float *)dsrc, rows, cols, ld, row0, col0, scale, drow_idx);
    });
}

int klnmf_set_W(klnmf_ctx *c, const void *src, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!src) fail(KLNMF_ERR_ARG, "null source");
        set_matrix(c, src, dtype, c->n, c->k, c->W[c->cur], c->W32[c->cur], c->KP, c->v_scale);
        if (!c->is_exact()) measure_and_pack(c);     // (all padded rows too: the eps carrier column in every row a tile can contain)
    });
}

int klnmf_set_Q(klnmf_ctx *c, const void *src, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!c->is_exact()) fail(KLNMF_ERR_UNSUPP, "the ratio Q is never materialised in the bf16 modes");
        if (!src) fail(KLNMF_ERR_ARG, "null source");
        if (c->sparse) fail(KLNMF_ERR_UNSUPP, "klnmf_set_Q: the ratio of a CSR problem lives on X's structure");
        set_matrix(c, src, dtype, c->n, c->f, c->Q, nullptr, 0);
    });
}

int klnmf_init_W(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        reset_state(c);
        if (c->sparse) {     // W0 = X . H0^T over the stored entries (nmf.py:156 with CSR X); needs H^T first
            if (c->prec == KLNMF_PREC_F64)
                hipLaunchKernelGGL((k_sp_transpose_H<double>), dim3(grid_for(c->k * c->f)), dim3(256), 0, c->stream,
                                   (const double *)c->H, (double *)c->HT, c->k, c->f, (const DevState *)nullptr);
            else
                hipLaunchKernelGGL((k_sp_transpose_H<float>), dim3(grid_for(c->k * c->f)), dim3(256), 0, c->stream,
                                   (const float *)c->H, (float *)c->HT, c->k, c->f, (const DevState *)nullptr);
            EXACT_CALL(c, exact_W, c->sp_data, 0);
        } else if (c->is_exact()) EXACT_CALL(c, exact_W, c->V, 0);
        else fast_rowpass(c, ROW_INIT);
        c->cur ^= 1;
        if (!c->is_exact()) measure_and_pack(c, true);     // W0 = V.H0^T scales with H0: images with measured scales for the first update
    });
}

// fp16 storage: values above the maximum announced with klnmf_set_v_max were saturated on upload; a fit on such
// a matrix is not the fit of the caller's data, so every entry point that computes refuses it -- and a pair of factors the
// fp16 operand images cannot hold (op_range).  The counters change only on uploads and image measurements: they are read
// back (one copy + one synchronisation) only when one of those happened since the last check.
struct Refusals { int v_overflow = 0, op_range = 0; };
static Refusals read_refusals(klnmf_ctx *c) {
    if (c->is_exact()) return Refusals{};
    if (!c->refusals_dirty) return Refusals{};          // the last check of this state passed (a failing one stays dirty)
    DevState ds{};
    HIPCHK(hipMemcpyAsync(&ds, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    Refusals r;
    r.op_range = ds.op_range;
    r.v_overflow = c->v_uploaded ? ds.v_overflow : 0;
    return r;
}
static void raise_refusals(klnmf_ctx *c, const Refusals &r) {
    if (r.op_range != 0)
        fail(KLNMF_ERR_UNSUPP, "the factors exceed the fp16 operand range (max W x max H of " + std::to_string(r.op_range) +
                                   " component(s) is more than 2^15 times the largest entry of V: an initial dictionary whose rows "
                                   "sum to far more than 1?); run KLNMF_PREC_F32 / F64");
    if (r.v_overflow != 0)
        fail(KLNMF_ERR_ARG, "uploaded V exceeds the maximum given to klnmf_set_v_max (" + std::to_string(r.v_overflow) +
                                " values out of the fp16 storage range)");
    c->refusals_dirty = false;
}
static void check_v_overflow(klnmf_ctx *c) { raise_refusals(c, read_refusals(c)); }

// Loop entry points only (klnmf_run, klnmf_run_sharded, klnmf_loop_begin): may THIS loop use fp8 ratio tiles (e4m3 of
// ratio x sqrt(2) / 8, from its third iteration on)?  Four things decide, in this order:
//   shape   q8_ok of klnmf_set_problem: enough rows per context that the tiles' bytes matter (32 769 / 65 536);
//   length  at most kQ8MaxLoop (50) planned iterations -- klnmf_run's max_iter, the capacity of klnmf_set_problem for loops in
//           pieces: what the e4m3 rounding does to the loss grows with the square of the iteration count (monitor.hip.h);
//   range   the tiles end at 3584 / sqrt(2) (saturating): data whose largest entry is more than 256 times the mean entry can hold
//           ratios beyond that for many iterations (a spike the model has not fitted yet) -- those keep the 16-bit tiles.  What
//           still saturates in a loop that passed is corrected exactly (fix-up list) or, in bulk, ends the fp8 regime;
//   what the e4m3 rounding does to THIS data's H numerator is not guessed here but MEASURED while the loop runs: the monitor
//           (monitor.hip.h, launch_monitor) -- a loop that fails it continues on 16-bit tiles.  Round 4 held five more data
//           rules at this place (components, columns, stored entries per column, ...), each added after a fuzz case had
//           ended 2e-4 .. 1.2e-3 off the oracle; KLNMF_Q8_RULES=1 (development switch) re-applies the three that round 5's
//           monitor replaced, for A/B runs.
// KLNMF_QTILE = 8 (development) forces the tiles on, = 16 off.  `sum_x_global` / `cells_global` / `nnz_global`: the sums over ALL
// ranks' shards (the sharded loop passes the all-reduced values, so that every rank takes the same path); negative: this
// context's own.  `ok_all`: the conjunction of every rank's q8_ok (shards that straddle the row threshold must not mix tile
// formats: since round 4 fp8-tile numerators are sqrt(2) larger than 16-bit-tile ones); negative: this context's own.
static void begin_fp8_loop(klnmf_ctx *c, double sum_x_global = -1.0, double cells_global = -1.0, double nnz_global = -1.0,
                           int ok_all = -1, int64_t planned = -1) {
    c->sw = DevSwitches::read();
    c->loop_planned = planned > 0 ? planned : std::max<int64_t>(1, c->cap);
    c->q8_loop = false;
    c->iter_in_loop = 0;
    c->w8_meas = false;
    c->stat_q8_tiles = 0;
    c->stat_col8 = 0;
    c->ne_loop = false;
    c->last_row_ne = false;
    c->mon_checks = 0;
    c->mon_pending = false;
    c->mon_dry_pending = false;
    c->stat_mon_gave_up = false;
    if (c->is_exact() || !c->q8_ok || ok_all == 0) return;
    if (c->W8 != nullptr && c->w8tab != nullptr) {
        // a loop that stopped inside k_post can leave the conversion's maxima table filled and the scale buffers swapped an odd
        // number of times: every loop starts from an empty table and unit scales (its second iteration measures)
        const std::vector<float> unit8((size_t)c->KP, 256.f);
        HIPCHK(hipMemsetAsync(c->w8tab, 0, (size_t)kW8TabRows * c->KP * 4, c->stream));
        HIPCHK(hipMemcpyAsync(c->w8s, unit8.data(), unit8.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->w8s_next, unit8.data(), unit8.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->conv_ran = false;
    }
    if (c->sw.qtile != 0) {
        c->q8_loop = c->sw.qtile == 8;
        c->ne_loop = c->q8_loop && c->ne_ok && c->sw.ne == 1;
        return;
    }
    double sum_x = sum_x_global, cells = cells_global, nnz = nnz_global;
    if (sum_x < 0) {
        DevState ds{};
        HIPCHK(hipMemcpyAsync(&ds, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        sum_x = ds.sum_x;
        cells = (double)c->n * (double)c->f;
        nnz = ds.nnz_x;
    }
    if (nnz < 0) nnz = cells;                 // (a caller that all-reduced only the two sums: dense data assumed)
    const double mean = sum_x / c->v_scale / cells;
    c->q8_loop = c->v_max > 0 && mean > 0 && c->v_max <= 256.0 * mean;
    if (c->loop_planned > kQ8MaxLoop) c->q8_loop = false;      // length: see monitor.hip.h
    if (c->sw.q8_rules_r4) {
        // round 4's data rules (kept for A/B runs against the monitor): fewer than four components or less than one column tile
        // of data (dead zones of the e4m3 step around ratio 1), fewer stored entries per column than half the row threshold
        if (c->k < 4) c->q8_loop = false;
        if (nnz / (double)c->f < 0.5 * (c->big ? 65536.0 : 32768.0)) c->q8_loop = false;
    }
    // The ratio without the numerator's eps (NE kernels, k <= 224): x / (W.H + eps) differs from the reference's
    // (x + eps) / (W.H + eps) by a relative eps / x per element.  Simulated in fp64 over 50 iterations (DESIGN_APPENDIX.md, h33)
    // the loss record moves by 0.06 .. 0.15 x eps / mean(V) and the factors by 0.5 .. 2.3 x eps / mean(V) of their maxima:
    // taken where eps / mean(V) <= 1e-5, i.e. 1.5e-6 and 2.5e-5 -- below the fp8 tiles' own floor (h29).
    c->ne_loop = c->q8_loop && c->ne_ok && mean >= 1.0e5 * kEpsRatio;
    if (c->sw.ne >= 0) c->ne_loop = c->q8_loop && c->ne_ok && c->sw.ne == 1;
}

// ---- a loop on this context's RCCL communicator (klnmf_comm_init): entry and iteration, shared by klnmf_run_sharded (the
// whole loop in one call) and by klnmf_loop_begin / klnmf_run_more (the same loop in parts) ---------------------------------
// KLNMF_COMM_SINGLE=1 (tests): a ONE-rank communicator takes the collective path too -- the same agreement block, grouped
// all-reduces (in place, on the loop's own buffers, counts and types) and decision kernel that N ranks execute; RCCL refuses
// two ranks on one device, so this is the only way a one-GPU box ever runs these lines.
static bool comm_multi(const klnmf_ctx *c) { return c->comm != nullptr && (c->comm_size > 1 || DevSwitches::read().comm_single); }

// Loop entry.  Every rank must take the same decisions, or the others block in a collective for ever: the refusal counters
// (a rank-local overflow, a rank-local operand range) are all-reduced (max) and every rank fails TOGETHER; the fp8 decision
// is taken from the all-reduced sums, so that all ranks run the same kernels and N = 1 / N = 8 differ by summation order only.
static void comm_loop_entry(klnmf_ctx *c, int64_t planned = -1) {
    if (c->sparse) fail(KLNMF_ERR_UNSUPP, "loops on a communicator: dense problems only");
    c->refusals_dirty = true;
    const Refusals mine = read_refusals(c);
    DevState ds{};
    HIPCHK(hipMemcpyAsync(&ds, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // (h[2]: a rank whose shard is too short for fp8 ratio tiles -- shards differ by a row tile and the last takes the remainder, so
    // they can straddle the row threshold -- keeps EVERY rank on 16-bit tiles: the numerators of the two formats differ by sqrt(2))
    double h[6] = {(double)(mine.v_overflow != 0), (double)(mine.op_range != 0), c->q8_ok ? 0.0 : 1.0,
                   ds.sum_x, (double)c->n * (double)c->f, ds.nnz_x};
    HIPCHK(hipMemcpyAsync(c->comm_scratch, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rccl().GroupStart());
    ncclResult_t r1 = rccl().AllReduce(c->comm_scratch, c->comm_scratch, 3, ncclDouble, ncclMax, c->comm, c->stream);
    ncclResult_t r2 = rccl().AllReduce(c->comm_scratch + 3, c->comm_scratch + 3, 3, ncclDouble, ncclSum, c->comm, c->stream);
    ncclResult_t r3 = rccl().GroupEnd();            // always closed, whatever the calls inside returned
    RCCLCHK(r1); RCCLCHK(r2); RCCLCHK(r3);
    HIPCHK(hipMemcpyAsync(h, c->comm_scratch, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h[0] != 0 || h[1] != 0) {
        if (mine.v_overflow || mine.op_range) raise_refusals(c, mine);       // this rank's own message
        fail(h[1] != 0 ? KLNMF_ERR_UNSUPP : KLNMF_ERR_ARG,
             h[1] != 0 ? "another rank's factors exceed the fp16 operand range: the sharded loop is refused on every rank"
                       : "another rank's shard of V exceeds the maximum given to klnmf_set_v_max: the sharded loop is refused on every rank");
    }
    c->refusals_dirty = false;
    begin_fp8_loop(c, h[3], h[4], h[5], h[2] == 0.0 ? 1 : 0, planned);
}

// One iteration: row pass -> column pass (it does not depend on the stop decision) -> ONE grouped RCCL launch on the
// context's stream (the k real rows of the numerator -- the 16-bit modes lay it out [KP][f_pad], rows beyond k are padding --
// and the two doubles of the loss) -> stop rule on identical inputs -> H rule.
static void comm_iteration(klnmf_ctx *c, int fit, double tol_abs) {
    if (fit && !c->is_exact()) {
        // Fused tail with column parts (post.hip.h).  Per part: column pass -> k_post(SUM): slabs -> this part's numerator
        // [KP][ld] (contiguous: one ncclAllReduce), fix-ups; the first part's launch also leaves the loss in loss_xchg.  The
        // all-reduce of every part but the last goes to the communicator's own stream behind an event and runs while the
        // next part's column pass computes (KLNMF_COMM_OVERLAP=0: all of them on the context's stream, in sequence -- the
        // same arithmetic, bit for bit).  The last part's all-reduce and the loss travel as ONE grouped RCCL launch on the
        // context's stream, AFTER the earlier all-reduces have completed (no two collectives of one communicator ever run
        // concurrently); k_post(RULE) then takes the stop decision from the exchanged loss and applies the H rule.
        const int P = c->nparts_cfg > 1 ? c->nparts_cfg : 1;
        const klnmf_ctx::PartCfg *parts = P > 1 ? c->parts : &c->whole;
        const bool overlap = P > 1 && c->sw.comm_overlap && c->comm_stream != nullptr;
        piece_rowpass(c, fit, nullptr, true);
        const LossArgs la = c->pending_loss;
        c->pending_loss.part = nullptr;
        const bool use8 = fused_w8_stage(c);
        if (use8) c->stat_col8 += 1;
        launch_monitor(c, use8);
        for (int p = 0; p < P; ++p) {
            fused_colpass_part(c, parts[p], use8);
            launch_post(c, POST_SUM, &parts[p], 1, p == 0 ? la : kNoLoss, false, use8, p == P - 1);
            if (p < P - 1) {
                float *nb = c->numerF + parts[p].numer_off;
                const size_t cnt = (size_t)c->k * (size_t)parts[p].ld;
                if (overlap) {
                    HIPCHK(hipEventRecord(c->ev_part[p], c->stream));
                    HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_part[p], 0));
                    RCCLCHK(rccl().AllReduce(nb, nb, cnt, ncclFloat, ncclSum, c->comm, c->comm_stream));
                    HIPCHK(hipEventRecord(c->ev_ar[p], c->comm_stream));
                } else {
                    RCCLCHK(rccl().AllReduce(nb, nb, cnt, ncclFloat, ncclSum, c->comm, c->stream));
                }
            }
        }
        if (overlap)
            for (int p = 0; p < P - 1; ++p) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_ar[p], 0));
        float *nb = c->numerF + parts[P - 1].numer_off;
        const size_t cnt = (size_t)c->k * (size_t)parts[P - 1].ld;
        RCCLCHK(rccl().GroupStart());
        ncclResult_t ra = rccl().AllReduce(nb, nb, cnt, ncclFloat, ncclSum, c->comm, c->stream);
        ncclResult_t rb = rccl().AllReduce(c->loss_xchg, c->loss_xchg, 2, ncclDouble, ncclSum, c->comm, c->stream);
        ncclResult_t rc = rccl().GroupEnd();       // closed on the error path too
        RCCLCHK(ra); RCCLCHK(rb); RCCLCHK(rc);
        LossArgs lt = kNoLoss;
        lt.tol_abs = tol_abs;
        launch_post(c, POST_RULE, parts, P, lt, true, false, false);
        c->cur ^= 1;
        c->iter_in_loop += 1;
        poll_fp8_overflow(c, true);
        return;
    }
    const size_t ncount = c->is_exact() ? (size_t)(c->k * c->f) : (size_t)c->k * (size_t)c->f_pad;
    void *nbuf = c->is_exact() ? c->numer : (void *)c->numerF;
    const ncclDataType_t ntype = c->prec == KLNMF_PREC_F64 ? ncclDouble : ncclFloat;
    piece_rowpass(c, fit);                     // leaves this rank's part of the loss in loss_xchg
    if (fit) piece_colpass(c);                 // ... and of the numerator
    RCCLCHK(rccl().GroupStart());
    ncclResult_t ra = fit ? rccl().AllReduce(nbuf, nbuf, ncount, ntype, ncclSum, c->comm, c->stream) : ncclSuccess;
    ncclResult_t rb = rccl().AllReduce(c->loss_xchg, c->loss_xchg, 2, ncclDouble, ncclSum, c->comm, c->stream);
    ncclResult_t rc = rccl().GroupEnd();       // closed on the error path too
    RCCLCHK(ra); RCCLCHK(rb); RCCLCHK(rc);
    piece_decide(c, tol_abs);                  // identical inputs on every rank -> identical decisions
    if (fit) piece_update_H(c);
    c->cur ^= 1;
    c->iter_in_loop += 1;
    if (fit) poll_fp8_overflow(c, !c->is_exact());
}

int klnmf_loop_begin(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        if (comm_multi(c)) {
            comm_loop_entry(c);                // the entry of klnmf_run_sharded: agreed refusals, agreed fp8 decision
        } else {
            check_v_overflow(c);
            begin_fp8_loop(c);
        }
        reset_state(c);
        c->loop_start_cur = c->cur;
        c->loop_hswaps = 0; c->loop_h0 = c->H32; c->loop_h1 = c->H32alt;
        c->loop_iters = 0;
    });
}

int klnmf_loop_begin_sharded(klnmf_ctx *c, double sum_x_all, double cells_all) {
    return klnmf_loop_begin_sharded_nnz(c, sum_x_all, cells_all, -1.0);
}

int klnmf_loop_begin_sharded_nnz(klnmf_ctx *c, double sum_x_all, double cells_all, double nnz_all) {
    return klnmf_loop_begin_agreed(c, sum_x_all, cells_all, nnz_all, -1);
}

int klnmf_loop_begin_agreed(klnmf_ctx *c, double sum_x_all, double cells_all, double nnz_all, int fp8_shape_all) {
    return guarded([&] {
        need_problem(c);
        if (!(sum_x_all >= 0) || !(cells_all > 0)) fail(KLNMF_ERR_ARG, "klnmf_loop_begin_sharded: the all-reduced sums must be given");
        check_v_overflow(c);
        begin_fp8_loop(c, sum_x_all * c->v_scale, cells_all, nnz_all, fp8_shape_all);      // (the caller's sums are in the data's own units)
        reset_state(c);
        c->loop_start_cur = c->cur;
        c->loop_hswaps = 0; c->loop_h0 = c->H32; c->loop_h1 = c->H32alt;
        c->loop_iters = 0;
    });
}

/* `iters` more iterations of the loop klnmf_loop_begin opened, enqueued as klnmf_run enqueues them (the loss reduction and
 * the stop rule riding in the slab-sum launch of a fit): the loop in bulk, for callers that want to fence between two parts
 * of ONE loop (bench.py: warm-up iterations | timed iterations).  Results by klnmf_loop_end. */
int klnmf_run_more(klnmf_ctx *c, int64_t iters, int fit, double tol_abs) {
    return guarded([&] {
        need_problem(c);
        if (iters < 0) fail(KLNMF_ERR_ARG, "iters < 0");
        if (comm_multi(c)) {                   // the loop of klnmf_run_sharded, continued
            for (int64_t it = 0; it < iters; ++it) {
                comm_iteration(c, fit, tol_abs);
                c->loop_iters += 1;
            }
            return;
        }
        for (int64_t it = 0; it < iters; ++it) {
            piece_rowpass(c, fit, &tol_abs);      // (every mode: the stop rule rides in the loss reduction's launch)
            if (fit) piece_fit_tail(c);
            c->cur ^= 1;
            c->loop_iters += 1;
            c->iter_in_loop += 1;
            if (fit) poll_fp8_overflow(c);
        }
    });
}

int klnmf_iter_rowpass(klnmf_ctx *c, int fit) {
    return guarded([&] {
        need_problem(c);
        piece_rowpass(c, fit);
    });
}

int klnmf_iter_decide(klnmf_ctx *c, double tol_abs) {
    return guarded([&] {
        need_problem(c);
        piece_decide(c, tol_abs);
    });
}

int klnmf_iter_colpass(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        piece_colpass(c);
    });
}

int klnmf_iter_colpass_part(klnmf_ctx *c, int part) {
    return guarded([&] {
        need_problem(c);
        piece_colpass_part(c, part);
    });
}

int klnmf_exchange_parts(klnmf_ctx *c, int *nparts, int64_t *offsets, int64_t *counts, int64_t *col0, int64_t *ncols) {
    return guarded([&] {
        need_problem(c);
        if (!nparts || !offsets || !counts) fail(KLNMF_ERR_ARG, "klnmf_exchange_parts: null pointer");
        const bool split = !c->is_exact() && c->nparts_cfg > 1;
        *nparts = split ? c->nparts_cfg : 1;
        for (int p = 0; p < *nparts; ++p) {
            if (split) {
                offsets[p] = c->parts[p].numer_off;
                counts[p] = c->k * (int64_t)c->parts[p].ld;
                if (col0) col0[p] = c->parts[p].col0;
                if (ncols) ncols[p] = c->parts[p].ncols;
            } else {
                offsets[p] = 0;
                counts[p] = c->is_exact() ? c->k * c->f : c->k * c->f_pad;
                if (col0) col0[p] = 0;
                if (ncols) ncols[p] = c->f;
            }
        }
    });
}

int klnmf_iter_update_H(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        piece_update_H(c);
    });
}

/* The host calls this after enqueuing each iteration's pieces so the W
 * ping-pong advances; kept separate from the pieces so transform (no H rule)
 * and fit share them. */
int klnmf_iter_advance(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        c->cur ^= 1;
        c->loop_iters += 1;
        c->iter_in_loop += 1;
        // the loop in pieces gives fp8 tiles up after bulk saturation like the loops in one call: the count is read where the
        // caller's exchange left it (loss_xchg[1]: summed over the ranks by the loss all-reduce, this context's own without one)
        if (!c->is_exact()) poll_fp8_overflow(c, true);
    });
}

int klnmf_loop_end(klnmf_ctx *c, double *errors_out, int64_t *n_done, int *stopped) {
    return guarded([&] {
        need_problem(c);
        fetch_results(c, errors_out, n_done, stopped);
    });
}

int klnmf_run(klnmf_ctx *c, int64_t max_iter, int fit, double tol_abs, double *errors_out,
              int64_t *n_done, int *stopped) {
    return guarded([&] {
        need_problem(c);
        if (max_iter < 0) fail(KLNMF_ERR_ARG, "max_iter < 0");
        if (max_iter > c->cap) fail(KLNMF_ERR_ARG, "max_iter exceeds the capacity given to klnmf_set_problem");
        check_v_overflow(c);
        begin_fp8_loop(c, -1.0, -1.0, -1.0, -1, max_iter);
        reset_state(c);
        c->loop_start_cur = c->cur;
        c->loop_hswaps = 0; c->loop_h0 = c->H32; c->loop_h1 = c->H32alt;
        // bf16 modes: the stop rule inside the loss kernel -- one launch fewer per iteration (a small problem's
        // iteration IS its kernel latencies: 7 launches of 4-10 us each).  Summing the column pass's slabs inside the
        // H rule as well (from_slabs) was measured and is NOT used: its k blocks walk the slabs serially, 47 -> 68 us.
        auto one_iteration = [&] {
            piece_rowpass(c, fit, &tol_abs);      // (every mode: the stop rule rides in the loss reduction's launch)
            if (fit) piece_fit_tail(c);
            c->cur ^= 1;
            c->iter_in_loop += 1;
            if (fit) poll_fp8_overflow(c);
        };
        auto stopped_already = [&]() -> bool {        // the stop rule may have fired: the remaining (no-op) iterations need not be enqueued
            DevState hs{};
            HIPCHK(hipMemcpyAsync(&hs, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            return hs.stop != 0;
        };
        // Launch-bound problems (the reference's own data: 10^2..10^4 rows, 4-7 kernels of a few microseconds per
        // iteration): two consecutive iterations -- both positions of the W ping-pong -- captured once into a hipGraph
        // and replayed.  The first two iterations run eagerly (they may carry the measured image scales of W0 and the
        // re-pack that follows them).  KLNMF_GRAPH=0 turns it off, =1 forces it for any size.
        int64_t it = 0;
        // Measured (scripts/small_problem_timing.py, 200 x 450 .. 10 000 x 4096): 29.5 us per iteration replayed against
        // 28.3 eager -- the iteration is the kernels' own few microseconds and their dependent boundaries, which a graph
        // keeps (MI355X_MICROARCH.md: "dependent kernel boundary ... eager = hipGraph"), not host launch cost.  Off unless asked for.
        const bool want_graph = c->stream != nullptr && !c->profiling && max_iter >= 12 && c->sw.graph != 0;
        if (want_graph) {
            for (; it < 2; ++it) one_iteration();
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            const int cur_before = c->cur;
            hipError_t ge = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
            if (ge == hipSuccess) {
                c->in_capture = true;
                try {
                    one_iteration();
                    one_iteration();
                } catch (...) {
                    c->in_capture = false;
                    (void)hipStreamEndCapture(c->stream, &graph);
                    if (graph) (void)hipGraphDestroy(graph);
                    throw;
                }
                c->in_capture = false;
                ge = hipStreamEndCapture(c->stream, &graph);
                if (ge == hipSuccess) ge = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            }
            c->cur = cur_before;                       // nothing has run yet: the capture only recorded the launches
            if (ge == hipSuccess && exec) {
                int64_t replays = 0;
                for (; it + 2 <= max_iter; it += 2) {
                    HIPCHK(hipGraphLaunch(exec, c->stream));
                    if (tol_abs > 0 && (++replays & 7) == 0 && stopped_already()) { it = max_iter; break; }
                }
            } else {
                (void)hipGetLastError();               // capture not available here: the eager loop below does the work
            }
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
        }
        for (; it < max_iter; ++it) {
            one_iteration();
            if (tol_abs > 0 && (it & 15) == 15 && stopped_already()) break;
        }
        fetch_results(c, errors_out, n_done, stopped);
    });
}

// ---- row shards: the native collective path (RCCL over xGMI) ------------------------------------------------------
int klnmf_comm_unique_id(void *id) {
    return guarded([&] {
        if (!id) fail(KLNMF_ERR_ARG, "null id buffer");
        static_assert(sizeof(ncclUniqueId) == KLNMF_COMM_ID_BYTES, "ncclUniqueId size");
        RCCLCHK(rccl().GetUniqueId((ncclUniqueId *)id));
    });
}

int klnmf_comm_init(klnmf_ctx *c, const void *id, int rank, int nranks) {
    return guarded([&] {
        use(c);
        if (!id || nranks < 1 || rank < 0 || rank >= nranks) fail(KLNMF_ERR_ARG, "klnmf_comm_init: bad rank / size / id");
        HIPCHK(hipStreamSynchronize(c->stream));
        comm_release(c);
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof(uid));
        RCCLCHK(rccl().CommInitRank(&c->comm, nranks, uid, rank));
        c->comm_rank = rank;
        c->comm_size = nranks;
        HIPCHK(hipMalloc((void **)&c->comm_scratch, 8 * sizeof(double)));
        // the parts' all-reduces that overlap the column pass (comm_iteration) run on a stream of their own
        HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        for (int p = 0; p < kPostMaxParts; ++p) {
            HIPCHK(hipEventCreateWithFlags(&c->ev_part[p], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_ar[p], hipEventDisableTiming));
        }
    });
}

int klnmf_comm_destroy(klnmf_ctx *c) {
    return guarded([&] {
        use(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        comm_release(c);
    });
}

int klnmf_comm_max(klnmf_ctx *c, double *value) {
    return guarded([&] {
        use(c);
        if (!value) fail(KLNMF_ERR_ARG, "null value");
        if (!comm_multi(c)) return;
        HIPCHK(hipMemcpyAsync(c->comm_scratch, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
        RCCLCHK(rccl().AllReduce(c->comm_scratch, c->comm_scratch, 1, ncclDouble, ncclMax, c->comm, c->stream));
        HIPCHK(hipMemcpyAsync(value, c->comm_scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_run_sharded(klnmf_ctx *c, int64_t n_total, int64_t max_iter, int fit, double tol, double *errors_out,
                      int64_t *n_done, int *stopped) {
    return guarded([&] {
        need_problem(c);
        if (max_iter < 0 || max_iter > c->cap) fail(KLNMF_ERR_ARG, "max_iter out of range");
        if (n_total < c->n) fail(KLNMF_ERR_ARG, "n_total smaller than this rank's rows");
        if (c->sparse) fail(KLNMF_ERR_UNSUPP, "klnmf_run_sharded: dense problems only");
        const bool multi = comm_multi(c);
        if (multi) {
            comm_loop_entry(c, max_iter);
        } else {
            check_v_overflow(c);
            begin_fp8_loop(c, -1.0, -1.0, -1.0, -1, max_iter);
        }
        reset_state(c);
        c->loop_start_cur = c->cur;
        c->loop_hswaps = 0; c->loop_h0 = c->H32; c->loop_h1 = c->H32alt;
        const double tol_abs = tol * (double)n_total * (double)c->f;          // nmf.py:207 on the GLOBAL shape
        for (int64_t it = 0; it < max_iter; ++it) {
            if (multi) {
                comm_iteration(c, fit, tol_abs);
            } else {
                // one rank: the stop decision rides in the loss kernel, as in klnmf_run (one launch less per iteration)
                piece_rowpass(c, fit, &tol_abs);
                if (fit) piece_fit_tail(c);
                c->cur ^= 1;
                c->iter_in_loop += 1;
                if (fit) poll_fp8_overflow(c);
            }
            if (tol_abs > 0 && (it & 15) == 15) {
                DevState hs{};
                HIPCHK(hipMemcpyAsync(&hs, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (hs.stop) break;
            }
        }
        fetch_results(c, errors_out, n_done, stopped);
    });
}

int klnmf_exchange_buffers(klnmf_ctx *c, void **loss_ptr, void **numer_ptr, int64_t *numer_count,
                           int *numer_is_f64) {
    return guarded([&] {
        need_problem(c);
        if (loss_ptr) *loss_ptr = c->loss_xchg;
        if (c->is_exact()) {
            if (numer_ptr) *numer_ptr = c->numer;
            if (numer_count) *numer_count = c->k * c->f;
            if (numer_is_f64) *numer_is_f64 = c->prec == KLNMF_PREC_F64;
        } else {
            if (numer_ptr) *numer_ptr = c->numerF;
            int64_t cnt = (int64_t)c->KP * c->f_pad;        // (the split layout of klnmf_exchange_parts may be longer: whole column blocks)
            if (c->nparts_cfg > 1)
                cnt = std::max(cnt, c->parts[c->nparts_cfg - 1].numer_off + (int64_t)c->KP * c->parts[c->nparts_cfg - 1].ld);
            if (numer_count) *numer_count = cnt;
            if (numer_is_f64) *numer_is_f64 = 0;
        }
    });
}

int klnmf_exchange_layout(klnmf_ctx *c, int64_t *row_stride, int64_t *valid_count) {
    return guarded([&] {
        need_problem(c);
        // numerator buffer: component rows of row_stride elements; only the first k rows (valid_count elements) carry data
        if (row_stride) *row_stride = c->is_exact() ? c->f : c->f_pad;
        if (valid_count) *valid_count = c->is_exact() ? c->k * c->f : c->k * c->f_pad;
    });
}

int klnmf_bind_exchange(klnmf_ctx *c, void *loss_ptr, void *numer_ptr) {
    return guarded([&] {
        need_problem(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        if (loss_ptr) c->loss_xchg = (double *)loss_ptr;
        if (numer_ptr) {
            if (c->is_exact()) c->numer = numer_ptr;
            else c->numerF = (float *)numer_ptr;
        }
    });
}

int klnmf_error(klnmf_ctx *c, double *loss) {
    return guarded([&] {
        need_problem(c);
        check_v_overflow(c);
        reset_state(c);
        if (c->is_exact()) {
            // the reference's CSR branch uses the caller's eps (nmf.py:301-308); its dense branch ignores it (nmf.py:309-310)
            EXACT_CALL(c, exact_Q, 0, c->sparse ? c->ratio_eps : kEpsRatio);
        } else {
            fast_rowpass(c, ROW_LOSS);
            hipLaunchKernelGGL(k_loss_from_parts, dim3(1), dim3(1024), 0, c->stream,
                               (const double2 *)c->loss_part2, (int64_t)c->nrt,
                               (const DevState *)c->st, 1.0 / c->v_scale, c->loss_xchg, 0, (DevState *)nullptr, 0.0,
                               (double *)nullptr, (int64_t)0, 0, c->images_measured ? 1 : 0);
            HIPCHK(hipGetLastError());
        }
        double h[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h, c->loss_xchg, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (loss) *loss = h[0];
    });
}

int klnmf_loss_terms(klnmf_ctx *c, double *terms) {
    return guarded([&] {
        need_problem(c);
        if (c->is_exact()) fail(KLNMF_ERR_UNSUPP, "klnmf_loss_terms: the exact modes evaluate the loss per element");
        if (!terms) fail(KLNMF_ERR_ARG, "null destination");
        reset_state(c);
        fast_rowpass(c, ROW_LOSS);
        std::vector<double2> parts((size_t)c->nrt);
        DevState hs{};
        HIPCHK(hipMemcpyAsync(parts.data(), c->loss_part2, sizeof(double2) * parts.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&hs, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        double a = 0, b = 0;
        for (const double2 &p : parts) { a += p.x; b += p.y; }
        if (c->images_measured) a += (double)hs.cq_e * hs.sum_x;      // a ratio-scaled dictionary image (k_ratio_scale): the unscaled ratio's logarithms
        terms[0] = kLn2 * a / c->v_scale;
        terms[1] = b / c->v_scale;
        terms[2] = hs.sum_x / c->v_scale;
        terms[3] = hs.corr_c / c->v_scale;
    });
}

int klnmf_update(klnmf_ctx *c, int fit) {
    return guarded([&] {
        need_problem(c);
        check_v_overflow(c);
        reset_state(c);
        // a single step always runs on 16-bit ratio tiles; the fp8 state of a loop around it is left as it was
        struct Keep { klnmf_ctx *c; bool q8; ~Keep() { c->q8_loop = q8; } } keep{c, c->q8_loop};
        c->q8_loop = false;
        piece_rowpass(c, fit);
        if (fit) {
            piece_colpass(c);
            piece_update_H(c);
        }
        c->cur ^= 1;
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_step_Q(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        if (!c->is_exact()) fail(KLNMF_ERR_UNSUPP, "the ratio Q is never materialised in the bf16 modes");
        reset_state(c);
        EXACT_CALL(c, exact_Q, 1, c->ratio_eps);
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_step_W(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        if (!c->is_exact()) fail(KLNMF_ERR_UNSUPP, "step API needs KLNMF_PREC_F64/F32");
        reset_state(c);
        EXACT_CALL(c, exact_W, c->sparse ? c->sp_q : c->Q, 1);
        c->cur ^= 1;
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_step_H(klnmf_ctx *c) {
    return guarded([&] {
        need_problem(c);
        if (!c->is_exact()) fail(KLNMF_ERR_UNSUPP, "step API needs KLNMF_PREC_F64/F32");
        reset_state(c);
        EXACT_CALL(c, exact_N, c->cur);
        EXACT_CALL(c, exact_H);
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_set_ratio_eps(klnmf_ctx *c, double eps) {
    return guarded([&] {
        use(c);
        if (!(eps >= 0)) fail(KLNMF_ERR_ARG, "eps must be >= 0");
        if (!c->is_exact() && eps != kEpsRatio)
            fail(KLNMF_ERR_UNSUPP, "the bf16 kernels use the reference's fixed eps = 1e-8");
        c->ratio_eps = eps;
    });
}

int klnmf_generalized_kl(klnmf_ctx *c, const void *x, const void *y, int dtype, int64_t count,
                         double eps, double *out) {
    return guarded([&] {
        use(c);
        if (!x || !y || count < 0) fail(KLNMF_ERR_ARG, "bad arguments");
        void *dx = stage_to_device(c, x, dtype, count);
        void *dy = nullptr;
        double *part = nullptr;
        const int grid = grid_for(count, 256, 1024);
        hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess;
        double host_part[1024];
        try {
            dy = stage_to_device(c, y, dtype, count);
            HIPCHK(hipMalloc((void **)&part, sizeof(double) * grid));
            if (dtype == KLNMF_DT_F64)
                hipLaunchKernelGGL((k_gkl<double>), dim3(grid), dim3(256), 0, c->stream, (const double *)dx, (const double *)dy, count, eps, part);
            else
                hipLaunchKernelGGL((k_gkl<float>), dim3(grid), dim3(256), 0, c->stream, (const float *)dx, (const float *)dy, count, eps, part);
            e1 = hipGetLastError();
            e2 = hipMemcpyAsync(host_part, part, sizeof(double) * grid, hipMemcpyDeviceToHost, c->stream);
            e3 = hipStreamSynchronize(c->stream);
        } catch (...) {
            (void)hipFree(dx);
            if (dy) (void)hipFree(dy);
            if (part) (void)hipFree(part);
            throw;
        }
        (void)hipFree(dx);
        (void)hipFree(dy);
        (void)hipFree(part);
        HIPCHK(e1);
        HIPCHK(e2);
        HIPCHK(e3);
        double s = 0;
        for (int i = 0; i < grid; ++i) s += host_part[i];
        if (out) *out = s;
    });
}

int klnmf_get_W(klnmf_ctx *c, void *dst, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!dst) fail(KLNMF_ERR_ARG, "null destination");
        get_matrix(c, dst, dtype, c->n, c->k, c->W[c->cur], c->W32[c->cur], c->KP, 1.0 / c->v_scale);
    });
}

int klnmf_get_H(klnmf_ctx *c, void *dst, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!dst) fail(KLNMF_ERR_ARG, "null destination");
        get_matrix(c, dst, dtype, c->k, c->f, c->H, c->H32, c->f_pad);
    });
}

int klnmf_get_Q(klnmf_ctx *c, void *dst, int dtype) {
    return guarded([&] {
        need_problem(c);
        if (!c->is_exact()) fail(KLNMF_ERR_UNSUPP, "the ratio Q is never materialised in the bf16 modes");
        if (!dst) fail(KLNMF_ERR_ARG, "null destination");
        if (c->sparse) fail(KLNMF_ERR_UNSUPP, "klnmf_get_Q: use klnmf_get_Q_values for a CSR problem");
        get_matrix(c, dst, dtype, c->n, c->f, c->Q, nullptr, 0);
    });
}

int klnmf_profile_enable(klnmf_ctx *c, int on) {
    return guarded([&] {
        use(c);
        c->profiling = on != 0;
    });
}

int klnmf_profile_read(klnmf_ctx *c, int64_t *row_n, double *row_ms, int64_t *col_n, double *col_ms,
                       int reset) {
    return guarded([&] {
        use(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        auto total = [&](std::vector<EventPair> &v, int64_t *n, double *ms) {
            double s = 0;
            for (auto &e : v) {
                float t = 0;
                HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
                s += t;
            }
            if (n) *n = (int64_t)v.size();
            if (ms) *ms = s;
            if (reset) {
                for (auto &e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
                v.clear();
            }
        };
        total(c->ev_row, row_n, row_ms);
        total(c->ev_col, col_n, col_ms);
    });
}

int klnmf_profile_read_tail(klnmf_ctx *c, int64_t *tail_n, double *tail_ms, int64_t *tail_rows, int reset) {
    return guarded([&] {
        use(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        double s = 0;
        for (auto &e : c->ev_tail) {
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
            s += t;
        }
        if (tail_n) *tail_n = (int64_t)c->ev_tail.size();
        if (tail_ms) *tail_ms = s;
        if (tail_rows) *tail_rows = c->tail_wg > 0 ? (int64_t)(c->nrt - c->tail_rt0()) * 32 : 0;
        if (reset) {
            for (auto &e : c->ev_tail) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
            c->ev_tail.clear();
        }
    });
}

int klnmf_query(klnmf_ctx *c, int what, int64_t *value) {
    return guarded([&] {
        use(c);
        if (!value) fail(KLNMF_ERR_ARG, "null value");
        switch (what) {
            case KLNMF_Q_FP8_LOOP: *value = c->q8_loop ? 1 : 0; break;
            case KLNMF_Q_FP8_TILE_ITERS: *value = c->stat_q8_tiles; break;
            case KLNMF_Q_FP8_COL_ITERS: *value = c->stat_col8; break;
            case KLNMF_Q_RATIO_TILE_BYTES:          // per element of V: 0 = no stored ratio tiles, 2 = 16-bit, 1 = fp8 once a loop allows them
                *value = (c->have_problem && c->Qt) ? (c->q8_ok ? 1 : 2) : 0;
                break;
            case KLNMF_Q_W8_SATURATED: *value = c->stat_w8_sat; break;
            case KLNMF_Q_W8_FALLBACKS: *value = c->stat_w8_fallbacks; break;
            case KLNMF_Q_RATIO_SATURATED: *value = c->stat_q8_sat; break;
            case KLNMF_Q_RATIO_UNFIXED: *value = c->stat_q8_unfixed; break;
            case KLNMF_Q_NO_NUM_EPS: *value = c->ne_loop ? 1 : 0; break;
            case KLNMF_Q_MON_CHECKS: *value = c->stat_mon_checks; break;
            case KLNMF_Q_MON_TRIPS: *value = c->stat_mon_trips; break;
            case KLNMF_Q_MON_GAVE_UP: *value = c->stat_mon_gave_up ? 1 : 0; break;
            case KLNMF_Q_COMM_RANKS: {
                int cnt = 1;
                if (c->comm) RCCLCHK(rccl().CommCount(c->comm, &cnt));
                *value = cnt;
                break;
            }
            default: fail(KLNMF_ERR_ARG, "klnmf_query: unknown item");
        }
    });
}

int klnmf_query_f64(klnmf_ctx *c, int what, double *value) {
    return guarded([&] {
        need_problem(c);
        if (!value) fail(KLNMF_ERR_ARG, "null value");
        if (what == KLNMF_QF_MON_STAT) { *value = c->stat_mon_max; return; }
        if (what == KLNMF_QF_MON_THRESHOLD) { *value = (double)mon_threshold_for((float)c->loop_planned); return; }
        if (what == KLNMF_QF_MON_SPREAD) { *value = c->stat_mon_spread; return; }
        if (what == KLNMF_QF_MON_MIN_SPREAD) { *value = (double)kMonMinSpread; return; }
        if (what >= KLNMF_QF_MON_PART0 && what < KLNMF_QF_MON_PART0 + 3) { *value = c->stat_mon_dbg[what - KLNMF_QF_MON_PART0]; return; }
        if (what != KLNMF_QF_SUM_V && what != KLNMF_QF_NNZ_V) fail(KLNMF_ERR_ARG, "klnmf_query_f64: unknown item");
        if (c->is_exact()) { *value = 0.0; return; }
        DevState ds{};
        HIPCHK(hipMemcpyAsync(&ds, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *value = what == KLNMF_QF_NNZ_V ? ds.nnz_x : ds.sum_x / c->v_scale;
    });
}

int klnmf_synchronize(klnmf_ctx *c) {
    return guarded([&] {
        use(c);
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

int klnmf_matmul(int device, int dtype, int64_t m, int64_t n, int64_t kk, const void *A, const void *B, void *C) {
    return guarded([&] {
        if (m < 0 || n < 0 || kk < 0 || m > (1LL << 30) || n > (1LL << 30) || kk > (1LL << 30))
            fail(KLNMF_ERR_ARG, "klnmf_matmul: bad shape");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "klnmf_matmul: dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (m == 0 || n == 0) return;
        if (m > (int64_t)65535 * GT) fail(KLNMF_ERR_UNSUPP, "klnmf_matmul: more than 65535 x 64 rows (row tiles ride on gridDim.y)");
        if (!A || !B || !C) fail(KLNMF_ERR_ARG, "klnmf_matmul: null pointer");
        HIPCHK(hipSetDevice(device));
        if (kk == 0) { std::memset(C, 0, (size_t)(m * n) * (dtype == KLNMF_DT_F64 ? 8 : 4)); return; }
        if (dtype == KLNMF_DT_F64) matmul_on_device<double>(m, n, kk, A, B, C);
        else matmul_on_device<float>(m, n, kk, A, B, C);
    });
}

int klnmf_matmul_device(int device, int dtype, int64_t m, int64_t n, int64_t kk, const void *dA, int64_t lda, const void *dB,
                        int64_t ldb, void *dC, int64_t ldc) {
    return guarded([&] {
        if (m < 0 || n < 0 || kk < 0 || m > (1LL << 30) || n > (1LL << 30) || kk > (1LL << 30))
            fail(KLNMF_ERR_ARG, "klnmf_matmul_device: bad shape");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "klnmf_matmul_device: dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (m == 0 || n == 0) return;
        if (m > (int64_t)65535 * GT) fail(KLNMF_ERR_UNSUPP, "klnmf_matmul_device: more than 65535 x 64 rows");
        if (!dA || !dB || !dC || lda < kk || ldb < n || ldc < n) fail(KLNMF_ERR_ARG, "klnmf_matmul_device: null pointer or short stride");
        HIPCHK(hipSetDevice(device));
        dim3 grid((unsigned)((n + GT - 1) / GT), (unsigned)((m + GT - 1) / GT), 1);
        if (dtype == KLNMF_DT_F64) {
            EpiStore<double> epi{(double *)dC, ldc};
            hipLaunchKernelGGL((k_gemm<double, EpiStore<double>>), grid, dim3(256), 0, 0, (int)m, (int)n, (int)kk, (const double *)dA,
                               lda, (int64_t)1, (const double *)dB, ldb, (int64_t)1, (int)kk + GK, (const DevState *)nullptr, epi);
        } else {
            EpiStore<float> epi{(float *)dC, ldc};
            hipLaunchKernelGGL((k_gemm<float, EpiStore<float>>), grid, dim3(256), 0, 0, (int)m, (int)n, (int)kk, (const float *)dA,
                               lda, (int64_t)1, (const float *)dB, ldb, (int64_t)1, (int)kk + GK, (const DevState *)nullptr, epi);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(0));
    });
}

int klnmf_all_distances_device(int device, int dtype, int metric, int64_t na, int64_t nb, int64_t d, const void *dA, int64_t lda,
                               const void *dB, int64_t ldb, void *dout) {
    return guarded([&] {
        if (na < 0 || nb < 0 || d < 0 || na > (1LL << 24) || nb > (1LL << 24) || d > (1LL << 30))
            fail(KLNMF_ERR_ARG, "klnmf_all_distances_device: bad shape");
        if (metric < DIST_KL || metric > DIST_COSINE_DIFF) fail(KLNMF_ERR_ARG, "klnmf_all_distances_device: unknown metric");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "klnmf_all_distances_device: dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (na == 0 || nb == 0) return;
        if (!dout || (d > 0 && (!dA || !dB)) || lda < d || ldb < d) fail(KLNMF_ERR_ARG, "klnmf_all_distances_device: null pointer or short stride");
        if (na * nb > ((int64_t)1 << 32)) fail(KLNMF_ERR_UNSUPP, "klnmf_all_distances_device: more than 2^32 pairs per call (four pairs per block on gridDim.x): split the rows");
        HIPCHK(hipSetDevice(device));
        const int64_t pairs = na * nb;
        if (dtype == KLNMF_DT_F64)
            hipLaunchKernelGGL((k_all_distances<double>), dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, 0, (const double *)dA,
                               (const double *)dB, (double *)dout, na, nb, d, metric, kEpsRatio, lda, ldb);
        else
            hipLaunchKernelGGL((k_all_distances<float>), dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, 0, (const float *)dA,
                               (const float *)dB, (float *)dout, na, nb, d, metric, kEpsRatio, lda, ldb);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(0));
    });
}

int klnmf_all_distances(int device, int dtype, int metric, int64_t na, int64_t nb, int64_t d, const void *A,
                        const void *B, void *out) {
    return guarded([&] {
        if (na < 0 || nb < 0 || d < 0 || na > (1LL << 24) || nb > (1LL << 24) || d > (1LL << 30))
            fail(KLNMF_ERR_ARG, "klnmf_all_distances: bad shape");
        if (metric < DIST_KL || metric > DIST_COSINE_DIFF) fail(KLNMF_ERR_ARG, "klnmf_all_distances: unknown metric");
        if (dtype != KLNMF_DT_F32 && dtype != KLNMF_DT_F64) fail(KLNMF_ERR_ARG, "klnmf_all_distances: dtype must be KLNMF_DT_F32 or KLNMF_DT_F64");
        if (na == 0 || nb == 0) return;
        if (!out || (d > 0 && (!A || !B))) fail(KLNMF_ERR_ARG, "klnmf_all_distances: null pointer");
        if (na * nb > ((int64_t)1 << 32)) fail(KLNMF_ERR_UNSUPP, "klnmf_all_distances: more than 2^32 pairs per call: split the rows");
        HIPCHK(hipSetDevice(device));
        if (dtype == KLNMF_DT_F64) distances_on_device<double>(metric, na, nb, d, A, B, out);
        else distances_on_device<float>(metric, na, nb, d, A, B, out);
    });
}

int klnmf_selftest(int device, int *failed) {
    return guarded([&] {
        HIPCHK(hipSetDevice(device));
        const int bits = run_probes();
        if (failed) *failed = bits;
    });
}

}  // extern "C"
