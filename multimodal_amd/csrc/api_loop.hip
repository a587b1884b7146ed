// libklnmf.so, unit 2 of 4: the loop of nmf.py:212-222 -- kernel dispatch, the pieces of an iteration, the stop rule's bookkeeping,
// the fp8 regime with its monitor, and the entry points that run loops and single steps (ctx.hip.h lists the units).
#include "ctx.hip.h"

namespace klnmf_host {
namespace {

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
    // stochastic rounding of fp8 ratio tiles: another stream per launch and per rank, the same streams for the same fit
    a.sr_seed = ((unsigned)c->sr_launches++ * 0x9E3779B9u + 0x7F4A7C15u) ^ ((unsigned)c->comm_rank * 0xC2B2AE35u);
    if (mode == ROW_UPDATE && a.Qt && c->q8()) c->stat_q8_tiles += 1;      // this update leaves fp8 ratio tiles
    EventPair ev{};
    if (c->prof_now) ev = begin_event(c, c->ev_row);
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
        if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
        return;
    }
    if (mode == ROW_UPDATE && c->tail_wg > 0) {
        // hybrid: the full rounds of workgroups take whole rows; the last partial round (tail_wg < CUs workgroups
        // that would each run a whole row block's length on an otherwise idle chip) is split into column chunks
        // and its W rule applied from the slabs
        launch_rowpass4_kt<ROW_UPDATE>(c, a4, grid4 - c->tail_wg);
        EventPair evt{};
        if (c->prof_now) evt = begin_event(c, c->ev_tail);
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
        if (c->prof_now) { HIPCHK(hipEventRecord(evt.b, c->stream)); HIPCHK(hipEventRecord(ev.b, c->stream)); }
        return;
    }
    switch (mode) {
        case ROW_UPDATE: launch_rowpass4_kt<ROW_UPDATE>(c, a4, grid4); break;
        case ROW_INIT: launch_rowpass4_kt<ROW_INIT>(c, a4, grid4); break;
        default: launch_rowpass4_kt<ROW_LOSS>(c, a4, grid4); break;
    }
    if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
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

// When: DRY on a loop's first iteration (16-bit tiles still: the e4m3 bytes are formed by the monitor itself -- the loop enters
// the fp8 regime only if that measurement passes), then on fp8 iterations 1 (the first with the real tiles and the e4m3 W
// image), 2, 4, 8, 16 and every 32nd after that.  The noise of stochastically rounded tiles is a property of the data (stored
// entries per column, rows) that barely moves with the iterations: the early checks decide, the later ones guard against gross
// errors.  Every check is followed by a poll (poll_fp8_overflow) -- a host synchronisation, i.e. a pipeline bubble of some
// 50 us: at the cadence of round 5's first monitor (every 8th) that was 3 % of configuration 2's 0.2 ms iterations.
bool monitor_due(int64_t n8) { return n8 == 1 || n8 == 2 || n8 == 4 || n8 == 8 || n8 == 16 || (n8 >= 32 && (n8 & 31) == 0); }






// ------------------------------------------------------------ exact pieces ---
// CSR input: ratio on the stored entries + loss (nmf.py:301-308, 331-334)
template <typename T>
void sparse_Q(klnmf_ctx *c, int write_q, double eps, const DecideArgs &dec) {
    const int64_t hk = c->k * c->f;
    EventPair ev{};
    if (c->prof_now) ev = begin_event(c, c->ev_row);
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
        if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
    } else {
    // (k > 512, or no stored entries: every lane takes its own entries and loops over the components)
    hipLaunchKernelGGL((k_sp_q_anyk<T>), dim3((unsigned)c->n), dim3(64), 0, c->stream, KL_SPQ_ARGS);
#undef KL_SPQ_ARGS
    HIPCHK(hipGetLastError());
    if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
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
    if (c->prof_now) ev = begin_event(c, c->ev_row);
    KL_GEMM_TT(c->q_tt, T, EpiQ<T>, grid, c->stream, (int)c->n, (int)c->f,
               (int)c->k, (const T *)c->W[c->cur], (int64_t)c->k, (int64_t)1,
               (const T *)c->H, (int64_t)c->f, (int64_t)1, (int)c->k + GK,
               (const DevState *)c->st, epi);
    HIPCHK(hipGetLastError());
    if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
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
        if (c->prof_now) evs = begin_event(c, c->ev_col);
        const int kcb = (int)((c->k + 63) / 64);
        const unsigned gridb = (unsigned)((int64_t)c->sp_rb * c->f);
#define KL_SPB_N(KCV) hipLaunchKernelGGL((k_spb_n<T, KCV>), dim3(gridb), dim3(64), 0, c->stream, (const int64_t *)c->csc_blkptr, (const int *)c->csc_rows32, \
        (const int *)c->csc_perm32, (const T *)c->sp_q, (const T *)c->W[widx], (T *)c->sp_NT, c->k, c->f, c->sp_rb, (const DevState *)c->st)
        if (kcb <= 1) KL_SPB_N(1); else if (kcb == 2) KL_SPB_N(2); else if (kcb <= 4) KL_SPB_N(4); else KL_SPB_N(8);
#undef KL_SPB_N
        hipLaunchKernelGGL((k_spb_numer<T>), dim3((unsigned)((c->f + 31) / 32), (unsigned)((c->k + 31) / 32)), dim3(256), 0, c->stream,
                           (const T *)c->sp_NT, c->sp_rb, (T *)c->numer, c->k, c->f, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        if (c->prof_now) HIPCHK(hipEventRecord(evs.b, c->stream));
        return;
    }
    if (c->sparse) {        // W^T . Q, one block per feature column (CSC order)
        EventPair evs{};
        if (c->prof_now) evs = begin_event(c, c->ev_col);
        const int spn_threads = (int)std::min<int64_t>(256, (c->k + 63) / 64 * 64);
        hipLaunchKernelGGL((k_sp_n<T>), dim3((unsigned)c->f), dim3(spn_threads), 0, c->stream, (const int64_t *)c->csc_indptr,
                           (const int64_t *)c->csc_rows, (const int64_t *)c->csc_perm, (const T *)c->sp_q,
                           (const T *)c->W[widx], (T *)c->numer, c->k, c->f, (const DevState *)c->st);
        HIPCHK(hipGetLastError());
        if (c->prof_now) HIPCHK(hipEventRecord(evs.b, c->stream));
        return;
    }
    EpiN<T> epi{(T *)c->Npart, c->f, c->k * c->f};
    const int TLn = 16 * c->n_tt;
    dim3 grid((unsigned)((c->f + TLn - 1) / TLn), (unsigned)((c->k + TLn - 1) / TLn), (unsigned)c->nsplit);
    EventPair ev{};
    if (c->prof_now) ev = begin_event(c, c->ev_col);
    KL_GEMM_TT(c->n_tt, T, EpiN<T>, grid, c->stream, (int)c->k, (int)c->f,
               (int)c->n, (const T *)c->W[widx], (int64_t)1, (int64_t)c->k,
               (const T *)c->Q, (int64_t)c->f, (int64_t)1, c->kchunk,
               (const DevState *)c->st, epi);
    HIPCHK(hipGetLastError());
    if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
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

}  // namespace

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
    hipLaunchKernelGGL(k_q8_monitor, dim3(2 * kMonBlocks), dim3(256), 0, c->stream, a);
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
                       &c->st->w8_sat, w8_probe_col(c), c->w8tab,
                       ((unsigned)c->sr_launches++ * 0x9E3779B9u + 0x7F4A7C15u) ^ ((unsigned)c->comm_rank * 0xC2B2AE35u));
    HIPCHK(hipGetLastError());
    c->w8_meas = true;
    c->conv_ran = true;
    return use8;
}

void fused_colpass_part(klnmf_ctx *c, const klnmf_ctx::PartCfg &p, bool use8) {
    ColPassQArgs a = colq_part_args(c, p);
    const int grid = p.ncb * p.nchunks;
    EventPair ev{};
    if (c->prof_now) ev = begin_event(c, c->ev_col);
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
    if (c->prof_now) HIPCHK(hipEventRecord(ev.b, c->stream));
}

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
                        c->mon_noise_scale, c->sw.mon_threshold > 0.f ? c->sw.mon_threshold : kMonThreshold};
        c->mon_pending = false;
    }
    // the row pass's loss partials are reduced by up to 16 blocks of k_slab_sum (one slice each); k_post sums their pairs
    const int nloss = (a.do_sum && la.part != nullptr)
                          ? (int)std::min<int64_t>(kLossRedMax, std::max<int64_t>(1, (la.count + kLossRedSlice - 1) / kLossRedSlice)) : 0;
    a.loss_red = c->loss_red; a.nloss = nloss; a.inv_c = la.inv_c; a.loss_xchg = c->loss_xchg; a.ne = la.ne; a.cq_on = la.cq_on;
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
    if (a.do_sum) {
        // slabs -> numerator rows of these parts, loss partials -> loss_red: the wide launch in front of k_post (post.hip.h)
        SlabSumArgs sa{};
        for (int p = 0; p < nparts; ++p) sa.part[p] = a.part[p];
        sa.nparts = nparts; sa.k = (int)c->k; sa.f_pad = c->f_pad; sa.nloss = nloss;
        sa.loss_part = la.part; sa.loss_count = la.count; sa.loss_red = c->loss_red; sa.st = c->st;
        hipLaunchKernelGGL(k_slab_sum, dim3((unsigned)((c->f_pad + 1023) / 1024), (unsigned)(c->k + nloss)), dim3(256), 0, c->stream, sa);
        HIPCHK(hipGetLastError());
    }
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

// When a loop gives the fp8 regime up for its remaining iterations: bulk saturation of the ratio tiles (more entries per
// iteration than the fix-up list holds: the range rule at the loop's entry normally excludes such matrices), or the monitor's
// statistic above its threshold (monitor.hip.h).  Polled on the monitor's cadence -- fp8 iterations 1 .. 4 and every eighth
// after them (one DevState read-back and stream synchronisation each).
// Row shards: every rank must drop the tiles in the SAME iteration (they would run different kernels otherwise, and the
// replicas of H would drift apart): the count travels as the second double of the loss exchange -- every loss launch leaves
// this rank's q8_unfixed in loss_xchg[1], the all-reduce (native or torch) sums it -- and `agreed` polls read that sum.
// Will the poll behind the iteration whose column pass has just been enqueued (klnmf_iter_advance) read the agreed count?  The
// torch-sequenced loop exchanges loss_xchg[1] only then, BEHIND the column pass's k_post (distributed.py) -- every rank holds
// the same loop state, so every rank answers alike.
bool fp8_poll_due(const klnmf_ctx *c) {
    if (c->is_exact() || !c->q8_loop || c->in_capture) return false;
    if (c->mon_dry_pending) return true;
    const bool q8_then = c->iter_in_loop + 1 >= 2 && (!c->big || (c->W8 != nullptr && c->w8_meas));      // q8() after the advance
    return q8_then && monitor_due(c->stat_q8_tiles);
}

// what a poll's read-back says: 16-bit tiles from the next iteration on (klnmf_query reports the counts and the statistic)
static void poll_verdict(klnmf_ctx *c, bool agreed) {
    bool give_up;
    if (agreed) {
        const double *h = (const double *)c->poll_host;
        give_up = h[1] > 0;
    } else {
        const DevState *hs = (const DevState *)c->poll_host;
        give_up = hs->q8_unfixed > 0 || hs->mon_trips > 0;
    }
    if (give_up) { c->q8_loop = false; c->stat_mon_gave_up = true; }
}

// the answer of a deferred poll (enqueued behind an EARLIER iteration: the event has passed or is about to)
void poll_resolve(klnmf_ctx *c) {
    if (!c->poll_inflight) return;
    c->poll_inflight = false;
    HIPCHK(hipEventSynchronize(c->poll_ev));
    if (c->q8_loop) poll_verdict(c, c->poll_agreed);
}

void poll_fp8_overflow(klnmf_ctx *c, bool agreed) {
    poll_resolve(c);
    if (!c->q8_loop || c->in_capture) return;
    const bool dry = c->mon_dry_pending;          // the dry run of the iteration just enqueued decides whether the next one takes fp8 tiles
    c->mon_dry_pending = false;
    if (!dry && !(c->q8() && monitor_due(c->stat_q8_tiles))) return;
    if (!c->poll_host) {
        HIPCHK(hipHostMalloc(&c->poll_host, sizeof(DevState) > 16 ? sizeof(DevState) : 16, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&c->poll_ev, hipEventDisableTiming));
    }
    if (agreed) HIPCHK(hipMemcpyAsync(c->poll_host, c->loss_xchg, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    else HIPCHK(hipMemcpyAsync(c->poll_host, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    if (dry) {                                    // the next iteration's kernels depend on the answer: wait for it
        HIPCHK(hipStreamSynchronize(c->stream));
        poll_verdict(c, agreed);
        return;
    }
    // a later check guards a regime the loop is already in: its answer may come one iteration late, and the stream keeps running
    HIPCHK(hipEventRecord(c->poll_ev, c->stream));
    c->poll_inflight = true;
    c->poll_agreed = agreed;
}


// fused_tol != nullptr (klnmf_run): the stop rule rides in the launch that reduces the loss (no k_decide launch)
void piece_rowpass(klnmf_ctx *c, int fit, const double *fused_tol, bool defer_to_post) {
    c->prof_now = c->profiling && (c->profile_seq++ % c->profile_every) == 0;      // (sampled event bracketing: ctx.hip.h)
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
    c->w_is_init = false;
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

void fetch_results(klnmf_ctx *c, double *errors_out, int64_t *n_done, int *stopped) {
    poll_resolve(c);
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
    // the last recorded loss over the sum of V (the loss is in the data's units, sum_x as stored: x v_scale)
    c->stat_kl_over_sumv = -1.0;
    if (!c->is_exact() && hs.n_done > 0) {
        const double sx = c->loop_sum_x_all >= 0 ? c->loop_sum_x_all : hs.sum_x;
        if (sx > 0) c->stat_kl_over_sumv = hs.prev_err * c->v_scale / sx;
    }
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

Refusals read_refusals(klnmf_ctx *c) {
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

void raise_refusals(klnmf_ctx *c, const Refusals &r) {
    if (r.op_range != 0)
        fail(KLNMF_ERR_UNSUPP, "the factors exceed the fp16 operand range (max W x max H of " + std::to_string(r.op_range) +
                                   " component(s) is more than 2^15 times the largest entry of V: an initial dictionary whose rows "
                                   "sum to far more than 1?); run KLNMF_PREC_F32 / F64");
    if (r.v_overflow != 0)
        fail(KLNMF_ERR_ARG, "uploaded V exceeds the maximum given to klnmf_set_v_max (" + std::to_string(r.v_overflow) +
                                " values out of the fp16 storage range)");
    c->refusals_dirty = false;
}

void check_v_overflow(klnmf_ctx *c) { raise_refusals(c, read_refusals(c)); }

// Loop entry points only (klnmf_run, klnmf_run_sharded, klnmf_loop_begin): may THIS loop use fp8 ratio tiles (e4m3 of
// ratio x sqrt(2) / 8, stochastically rounded, from its third iteration on)?  Three things decide, in this order:
//   shape   q8_ok of klnmf_set_problem: enough rows per context that the tiles' bytes matter (32 769 / 65 536), one column
//           tile of data;
//   range   the tiles end at 3584 / sqrt(2) (saturating): data whose largest entry is more than 256 times the mean entry can hold
//           ratios beyond that for many iterations (a spike the model has not fitted yet) -- those keep the 16-bit tiles.  What
//           still saturates in a loop that passed is corrected exactly (fix-up list) or, in bulk, ends the fp8 regime;
//   noise   what the e4m3 rounding does to THIS data's H numerator is not guessed here but MEASURED while the loop runs: the
//           monitor (monitor.hip.h, launch_monitor), first as a dry run on the loop's first iteration -- a loop that fails it
//           continues on 16-bit tiles.  (Round 4 held five data rules at this place, each added after a fuzz case had ended
//           2e-4 .. 1.2e-3 off the oracle; all of them were symptoms of round-to-nearest tiles' bias: DESIGN.md section 6.)
// KLNMF_QTILE = 8 (development) forces the tiles on, = 16 off.  `sum_x_global` / `cells_global` / `nnz_global`: the sums over ALL
// ranks' shards (the sharded loop passes the all-reduced values, so that every rank takes the same path); negative: this
// context's own.  `ok_all`: the conjunction of every rank's q8_ok (shards that straddle the row threshold must not mix tile
// formats: since round 4 fp8-tile numerators are sqrt(2) larger than 16-bit-tile ones); negative: this context's own.
void begin_fp8_loop(klnmf_ctx *c, double sum_x_global, double cells_global, double nnz_global, int ok_all) {
    c->sw = DevSwitches::read();
    c->q8_loop = false;
    c->iter_in_loop = 0;
    c->sr_launches = 0;
    c->w8_meas = false;
    c->stat_q8_tiles = 0;
    c->stat_col8 = 0;
    c->ne_loop = false;
    c->last_row_ne = false;
    c->mon_checks = 0;
    c->mon_pending = false;
    c->mon_dry_pending = false;
    c->stat_mon_gave_up = false;
    if (c->poll_inflight) { (void)hipEventSynchronize(c->poll_ev); c->poll_inflight = false; }      // (a loop abandoned without klnmf_loop_end)
    c->loop_sum_x_all = sum_x_global;              // (stored units; < 0: fetch_results takes this context's own sum)
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
    (void)nnz;                                // (the stored-entry count stays in the ABI: no rule reads it since the monitor measures)
    const double mean = sum_x / c->v_scale / cells;
    c->q8_loop = c->v_max > 0 && mean > 0 && c->v_max <= 256.0 * mean;
    // The ratio without the numerator's eps (NE kernels, k <= 224): x / (W.H + eps) differs from the reference's
    // (x + eps) / (W.H + eps) by a relative eps / x per element.  Simulated in fp64 over 50 iterations (DESIGN_APPENDIX.md, h33)
    // the loss record moves by 0.06 .. 0.15 x eps / mean(V) and the factors by 0.5 .. 2.3 x eps / mean(V) of their maxima:
    // taken where eps / mean(V) <= 1e-5, i.e. 1.5e-6 and 2.5e-5 -- below the fp8 tiles' own floor (h29).
    c->ne_loop = c->q8_loop && c->ne_ok && mean >= 1.0e5 * kEpsRatio;
    if (c->sw.ne >= 0) c->ne_loop = c->q8_loop && c->ne_ok && c->sw.ne == 1;
}

}  // namespace klnmf_host

extern "C" {

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
        c->w_is_init = true;
        if (!c->is_exact()) measure_and_pack(c, true);     // W0 = V.H0^T scales with H0: images with measured scales for the first update
    });
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
        begin_fp8_loop(c);
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
        // (never on a loop that may take fp8 tiles: a replay would reuse the captured iterations' stochastic-rounding seeds and skip the
        // monitor's checks and polls -- the premises of the fp8 regime)
        const bool want_graph = c->stream != nullptr && !c->profiling && max_iter >= 12 && c->sw.graph != 0 && !c->q8_loop;
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

}  // extern "C"
