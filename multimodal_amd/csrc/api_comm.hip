// libklnmf.so, unit 3 of 4: row shards over the GPUs of a node -- the RCCL entry points (opened at run time), the loop entry every
// rank agrees on, the iteration with its ONE grouped all-reduce, and the exchange buffers of the torch path (ctx.hip.h lists the units).
#include "ctx.hip.h"

namespace klnmf_host {

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

// ---- a loop on this context's RCCL communicator (klnmf_comm_init): entry and iteration, shared by klnmf_run_sharded (the
// whole loop in one call) and by klnmf_loop_begin / klnmf_run_more (the same loop in parts) ---------------------------------
// KLNMF_COMM_SINGLE=1 (tests): a ONE-rank communicator takes the collective path too -- the same agreement block, grouped
// all-reduces (in place, on the loop's own buffers, counts and types) and decision kernel that N ranks execute; RCCL refuses
// two ranks on one device, so this is the only way a one-GPU box ever runs these lines.
bool comm_multi(const klnmf_ctx *c) { return c->comm != nullptr && (c->comm_size > 1 || DevSwitches::read().comm_single); }

// Loop entry.  Every rank must take the same decisions, or the others block in a collective for ever: the refusal counters
// (a rank-local overflow, a rank-local operand range) are all-reduced (max) and every rank fails TOGETHER; the fp8 decision
// is taken from the all-reduced sums, so that all ranks run the same kernels and N = 1 / N = 8 differ by summation order only.
void comm_loop_entry(klnmf_ctx *c) {
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
    begin_fp8_loop(c, h[3], h[4], h[5], h[2] == 0.0 ? 1 : 0);
}

// One iteration: row pass -> column pass (it does not depend on the stop decision) -> ONE grouped RCCL launch on the
// context's stream (the k real rows of the numerator -- the 16-bit modes lay it out [KP][f_pad], rows beyond k are padding --
// and the two doubles of the loss) -> stop rule on identical inputs -> H rule.
void comm_iteration(klnmf_ctx *c, int fit, double tol_abs) {
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

}  // namespace klnmf_host

extern "C" {

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
            comm_loop_entry(c);
        } else {
            check_v_overflow(c);
            begin_fp8_loop(c);
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

}  // extern "C"
