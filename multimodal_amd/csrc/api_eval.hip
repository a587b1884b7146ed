// libklnmf.so, unit 4 of 4: evaluation and introspection -- reconstruction products, nearest-neighbour distances, the generalized KL of two
// arrays, the hardware probes, queries and profiling (ctx.hip.h lists the units).
#include "ctx.hip.h"
#include "probe.hip.h"

thread_local std::string klnmf_host::g_err;

namespace {

template <typename T>
void distances_on_device(int metric, int64_t na, int64_t nb, int64_t d, const void *A, const void *B, void *out) {
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
void matmul_on_device(int64_t m, int64_t n, int64_t kk, const void *A, const void *B, void *C) {
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

}  // namespace

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

int klnmf_profile_enable(klnmf_ctx *c, int on) {
    return guarded([&] {
        use(c);
        c->profiling = on != 0;
        c->profile_every = on > 1 ? on : 1;
        c->profile_seq = 0;
        c->prof_now = c->profiling;          // (launches outside a loop's iterations -- klnmf_error, klnmf_init_W -- follow the last setting)
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
            case KLNMF_Q_FP8_POLL_DUE: *value = (c->have_problem && fp8_poll_due(c)) ? 1 : 0; break;
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
        if (what == KLNMF_QF_MON_THRESHOLD) { *value = (double)(c->sw.mon_threshold > 0.f ? c->sw.mon_threshold : kMonThreshold); return; }
        if (what == KLNMF_QF_MON_SPREAD) { *value = c->stat_mon_spread; return; }
        if (what == KLNMF_QF_MON_MIN_SPREAD) { *value = (double)kMonMinSpread; return; }
        if (what == KLNMF_QF_KL_OVER_SUM_V) { *value = c->stat_kl_over_sumv; return; }
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
