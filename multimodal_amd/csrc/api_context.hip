// libklnmf.so, unit 1 of 4: contexts, problems, uploads and downloads (ctx.hip.h lists the units).
#include "ctx.hip.h"

DevBlockCache g_block_cache;

namespace {

// ---- the fp8 monitor (monitor.hip.h) -------------------------------------------------------------------------------------------
void monitor_setup(klnmf_ctx *c) {
    c->mon_part = nullptr;
    c->mon_pending = false;
    if (!c->q8_ok) return;
    c->mon_part = (float *)c->dalloc((size_t)kMonBlocks * 2 * 2 * c->KP * 32 * 4);
    c->mon_spread = (float *)c->dalloc((size_t)2 * kMonBlocks * 96 * 4);
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
void choose_eps_carrier(klnmf_ctx *c) {
    const double ev = kEpsRatio * c->v_scale / (double)kCarrierW;
    const bool fits = ev >= 9.5367431640625e-07 && ev <= 32768.0;
    c->kc = (c->kc_shape >= 0 && fits) ? c->kc_shape : -1;
}


// ------------------------------------------------------------- loop pieces ---
// Empty V tile buffers: true zeros (padding rows and columns are inert; an all-zero row of V gives an exactly zero row of W,
// as in the reference -- also under the update pass without the numerator's eps, whose ratio carries a 2^-100 addend instead
// of relying on a stored "zero class": mfma4.hip.h, NE).
void fill_v_tiles(klnmf_ctx *c, void *tiles, size_t bytes) {
    HIPCHK(hipMemsetAsync(tiles, 0, bytes, c->stream));
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

// =============================================================== exports ===
template <typename D, typename S>
void copy_2d(klnmf_ctx *c, D *dst, int64_t dld, const S *src, int64_t sld, int64_t rows, int64_t cols, double mul = 1.0) {
    if (rows * cols == 0) return;
    hipLaunchKernelGGL((k_copy_2d<D, S>), dim3(grid_for(rows * cols, 256, 8192)), dim3(256), 0, c->stream, dst, dld, src, sld,
                       rows, cols, mul);
    HIPCHK(hipGetLastError());
}

}  // namespace

namespace klnmf_host {

void reset_state(klnmf_ctx *c) {
    hipLaunchKernelGGL(k_reset_state, dim3(1), dim3(1), 0, c->stream, c->st);
    HIPCHK(hipGetLastError());
}

void fast_pack_H(klnmf_ctx *c, const unsigned *wmax) {
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

// from_init: W is W0 = V.H0^T of klnmf_init_W -- the first update's ratios are about f / k times 1, and the dictionary image
// then carries the ratio scale k_ratio_scale derives (mfma.hip.h; KLNMF_RATIO_SCALE=0: never); any other W: scale 1.
void measure_and_pack(klnmf_ctx *c, bool from_init) {
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

}  // namespace klnmf_host

extern "C" {

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
        if (c->poll_ev) (void)hipEventDestroy(c->poll_ev);
        if (c->poll_host) (void)hipHostFree(c->poll_host);
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
        c->w_is_init = false;
        c->sparse = false;
        c->v_uploaded = false;
        c->refusals_dirty = true;
        c->v_scale = 1.0;
        c->nnz = 0;
        c->st = (DevState *)c->dalloc(sizeof(DevState));
        c->errors = (double *)c->dalloc(sizeof(double) * (cap > 0 ? cap : 1));
        c->loss_xchg = (double *)c->dalloc(sizeof(double) * 2);
        c->loss_red = (double2 *)c->dalloc(sizeof(double2) * kLossRedMax);
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
            // How many blocks: as many as make a block's gathered rows fit the L2 (`per` rows of k x es bytes) -- but a (row, block)
            // cell must still fill the kernels' trips, or the gather slots of its last trip run empty.  Measured (round 5, 20 000 x
            // 110 000, 0.5 %, k = 50, fp64; profiles/r05_sparse_blocks.txt): 15 column blocks (37 entries per cell) cut the fused
            // pass's fabric traffic from 4.5 to 1.1 GB and cost it 0.87 instead of 0.66 ms; 4 blocks (137 per cell): 0.62 ms; 3 row
            // blocks (33 per cell, groups of 16) take the H-side pass from 0.61 to 0.51 ms, 6 (17 per cell) back to 0.63.  So: at
            // least 128 entries per cell of the CSR order, 32 of the CSC order.  KLNMF_SP_CB / KLNMF_SP_RB (development) override.
            int64_t cb = std::min<int64_t>((f + per - 1) / per, std::max<int64_t>(1, nnz / std::max<int64_t>(1, n) / 128));
            int64_t rb = std::min<int64_t>((n + per - 1) / per, std::max<int64_t>(1, nnz / std::max<int64_t>(1, f) / 32));
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
            // the W that is there (zeros, a klnmf_set_W, W0 = V.H_init^T of klnmf_init_W) goes with it; behind klnmf_init_W the
            // first update still starts from W0 -- about f / k too small whatever dictionary is set now (transform with
            // components_ != _init_dictionary: nmf.py:159-230) --, so the first update's ratio scale stays on
            measure_and_pack(c, c->w_is_init);
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
                measure_and_pack(c, c->w_is_init);
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
        c->w_is_init = false;
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

int klnmf_set_ratio_eps(klnmf_ctx *c, double eps) {
    return guarded([&] {
        use(c);
        if (!(eps >= 0)) fail(KLNMF_ERR_ARG, "eps must be >= 0");
        if (!c->is_exact() && eps != kEpsRatio)
            fail(KLNMF_ERR_UNSUPP, "the bf16 kernels use the reference's fixed eps = 1e-8");
        c->ratio_eps = eps;
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

}  // extern "C"
