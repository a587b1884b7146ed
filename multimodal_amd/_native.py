"""ctypes binding of libklnmf.so (the C-ABI declared in include/klnmf.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device
is usable, everything that computes raises.  The binding mirrors the header one
to one; `Context` is a thin RAII wrapper used by `lib/nmf.py`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KLNMF_LIB') or os.path.join(_HERE, 'csrc', 'libklnmf.so')   # KLNMF_LIB: A/B builds

PREC_F64, PREC_F32, PREC_BF16 = 0, 1, 2
PRECISIONS = {'f64': PREC_F64, 'fp64': PREC_F64, 'float64': PREC_F64,
              'f32': PREC_F32, 'fp32': PREC_F32, 'float32': PREC_F32,
              'f16': PREC_BF16, 'fp16': PREC_BF16, 'float16': PREC_BF16,
              # the 16-bit mode's historical name (its MFMA operands were bf16 in round 1; fp16 with power-of-two
              # scaling since: same matrix rate, 8x smaller operand rounding -- csrc/mfma.hip.h)
              'bf16': PREC_BF16}
DT_F32, DT_F64 = 0, 1
STREAM_DEFAULT = (1 << 64) - 1        # KLNMF_STREAM_DEFAULT: (void *)(intptr_t)-1

ERR_ARG, ERR_ALLOC, ERR_HIP, ERR_UNSUPP, ERR_RCCL = -1, -2, -3, -4, -5
COMM_ID_BYTES = 128

_c = ctypes
_ctx_p = _c.c_void_p
_i64 = _c.c_int64

# name -> (restype, argtypes); must list every symbol of include/klnmf.h
SIGNATURES = {
    'klnmf_version': (_c.c_int, []),
    'klnmf_last_error': (_c.c_char_p, []),
    'klnmf_device_info': (_c.c_int, [_c.c_int, _c.c_char_p, _c.c_int,
                                     _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64)]),
    'klnmf_create': (_c.c_int, [_c.POINTER(_ctx_p), _c.c_int, _c.c_int, _c.c_void_p]),
    'klnmf_destroy': (_c.c_int, [_ctx_p]),
    'klnmf_set_problem': (_c.c_int, [_ctx_p, _i64, _i64, _i64, _i64]),
    'klnmf_release_problem': (_c.c_int, [_ctx_p]),
    'klnmf_set_v_max': (_c.c_int, [_ctx_p, _c.c_double]),
    'klnmf_reset_V': (_c.c_int, [_ctx_p]),
    'klnmf_upload_V': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int, _i64, _i64, _i64,
                                  _i64, _i64, _c.c_double]),
    'klnmf_upload_V_device_rows': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_void_p, _i64, _i64, _i64, _i64, _i64,
                                              _c.c_double]),
    'klnmf_upload_V_device': (_c.c_int, [_ctx_p, _c.c_void_p, _i64, _i64, _i64, _i64,
                                         _i64, _c.c_double]),
    'klnmf_set_H': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_set_W': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_set_Q': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_init_W': (_c.c_int, [_ctx_p]),
    'klnmf_run': (_c.c_int, [_ctx_p, _i64, _c.c_int, _c.c_double,
                             _c.POINTER(_c.c_double), _c.POINTER(_i64),
                             _c.POINTER(_c.c_int)]),
    'klnmf_loop_begin': (_c.c_int, [_ctx_p]),
    'klnmf_loop_begin_sharded': (_c.c_int, [_ctx_p, _c.c_double, _c.c_double]),
    'klnmf_loop_begin_sharded_nnz': (_c.c_int, [_ctx_p, _c.c_double, _c.c_double, _c.c_double]),
    'klnmf_loop_begin_agreed': (_c.c_int, [_ctx_p, _c.c_double, _c.c_double, _c.c_double, _c.c_int]),
    'klnmf_run_more': (_c.c_int, [_ctx_p, _i64, _c.c_int, _c.c_double]),
    'klnmf_iter_rowpass': (_c.c_int, [_ctx_p, _c.c_int]),
    'klnmf_iter_decide': (_c.c_int, [_ctx_p, _c.c_double]),
    'klnmf_iter_colpass': (_c.c_int, [_ctx_p]),
    'klnmf_iter_colpass_part': (_c.c_int, [_ctx_p, _c.c_int]),
    'klnmf_exchange_parts': (_c.c_int, [_ctx_p, _c.POINTER(_c.c_int), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64),
                                        _c.POINTER(_i64)]),
    'klnmf_iter_update_H': (_c.c_int, [_ctx_p]),
    'klnmf_iter_advance': (_c.c_int, [_ctx_p]),
    'klnmf_loop_end': (_c.c_int, [_ctx_p, _c.POINTER(_c.c_double), _c.POINTER(_i64),
                                  _c.POINTER(_c.c_int)]),
    'klnmf_comm_unique_id': (_c.c_int, [_c.c_void_p]),
    'klnmf_comm_init': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int, _c.c_int]),
    'klnmf_comm_destroy': (_c.c_int, [_ctx_p]),
    'klnmf_comm_max': (_c.c_int, [_ctx_p, _c.POINTER(_c.c_double)]),
    'klnmf_run_sharded': (_c.c_int, [_ctx_p, _i64, _i64, _c.c_int, _c.c_double,
                                     _c.POINTER(_c.c_double), _c.POINTER(_i64), _c.POINTER(_c.c_int)]),
    'klnmf_exchange_buffers': (_c.c_int, [_ctx_p, _c.POINTER(_c.c_void_p),
                                          _c.POINTER(_c.c_void_p), _c.POINTER(_i64),
                                          _c.POINTER(_c.c_int)]),
    'klnmf_exchange_layout': (_c.c_int, [_ctx_p, _c.POINTER(_i64), _c.POINTER(_i64)]),
    'klnmf_bind_exchange': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_void_p]),
    'klnmf_error': (_c.c_int, [_ctx_p, _c.POINTER(_c.c_double)]),
    'klnmf_loss_terms': (_c.c_int, [_ctx_p, _c.POINTER(_c.c_double)]),
    'klnmf_update': (_c.c_int, [_ctx_p, _c.c_int]),
    'klnmf_set_ratio_eps': (_c.c_int, [_ctx_p, _c.c_double]),
    'klnmf_step_Q': (_c.c_int, [_ctx_p]),
    'klnmf_step_W': (_c.c_int, [_ctx_p]),
    'klnmf_step_H': (_c.c_int, [_ctx_p]),
    'klnmf_generalized_kl': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_void_p, _c.c_int, _i64,
                                        _c.c_double, _c.POINTER(_c.c_double)]),
    'klnmf_get_W': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_get_H': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_get_Q': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_profile_enable': (_c.c_int, [_ctx_p, _c.c_int]),
    'klnmf_profile_read': (_c.c_int, [_ctx_p, _c.POINTER(_i64), _c.POINTER(_c.c_double),
                                      _c.POINTER(_i64), _c.POINTER(_c.c_double), _c.c_int]),
    'klnmf_profile_read_tail': (_c.c_int, [_ctx_p, _c.POINTER(_i64), _c.POINTER(_c.c_double), _c.POINTER(_i64), _c.c_int]),
    'klnmf_synchronize': (_c.c_int, [_ctx_p]),
    'klnmf_query': (_c.c_int, [_ctx_p, _c.c_int, _c.POINTER(_i64)]),
    'klnmf_set_H_device': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int, _i64, _i64, _i64, _c.c_int]),
    'klnmf_get_W_device': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int, _i64]),
    'klnmf_upload_V_device_rows_dt': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int, _c.c_void_p, _i64, _i64, _i64, _i64, _i64,
                                                 _c.c_double]),
    'klnmf_matmul_device': (_c.c_int, [_c.c_int, _c.c_int, _i64, _i64, _i64, _c.c_void_p, _i64, _c.c_void_p, _i64,
                                       _c.c_void_p, _i64]),
    'klnmf_all_distances_device': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _i64, _i64, _i64, _c.c_void_p, _i64,
                                              _c.c_void_p, _i64, _c.c_void_p]),
    'klnmf_query_f64': (_c.c_int, [_ctx_p, _c.c_int, _c.POINTER(_c.c_double)]),
    'klnmf_set_problem_sparse': (_c.c_int, [_ctx_p, _i64, _i64, _i64, _i64, _i64]),
    'klnmf_upload_csr': (_c.c_int, [_ctx_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                    _c.c_void_p, _c.c_void_p]),
    'klnmf_get_Q_values': (_c.c_int, [_ctx_p, _c.c_void_p, _c.c_int]),
    'klnmf_matmul': (_c.c_int, [_c.c_int, _c.c_int, _i64, _i64, _i64, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'klnmf_all_distances': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _i64, _i64, _i64, _c.c_void_p, _c.c_void_p,
                                       _c.c_void_p]),
    'klnmf_selftest': (_c.c_int, [_c.c_int, _c.POINTER(_c.c_int)]),
}

_lib = None


class NativeError(RuntimeError):
    """A non-zero status from the C-ABI."""

    def __init__(self, code, msg):
        RuntimeError.__init__(self, "klnmf error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libklnmf.so and declare every entry point.  Raises if it is absent:
    the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP extension %s is missing; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if a symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(status):
    if status != 0:
        msg = load().klnmf_last_error()
        msg = msg.decode('utf-8', 'replace') if msg else ''
        if status == ERR_ALLOC:
            raise MemoryError("klnmf: " + msg)
        raise NativeError(status, msg)


def _np_dtype_code(a):
    if a.dtype == np.float64:
        return DT_F64
    if a.dtype == np.float32:
        return DT_F32
    raise TypeError("expected float32/float64, got %s" % a.dtype)


def _as_float_array(a):
    """C-contiguous float32/float64 view or copy of a (ints -> float64)."""
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return np.ascontiguousarray(a)


def device_info(device=0):
    lib = load()
    arch = ctypes.create_string_buffer(64)
    cu = _c.c_int(0)
    mem = _c.c_uint64(0)
    _check(lib.klnmf_device_info(device, arch, 64, ctypes.byref(cu), ctypes.byref(mem)))
    return {'arch': arch.value.decode(), 'cu_count': cu.value, 'hbm_bytes': mem.value}


def matmul(A, B, device=0):
    """A.dot(B) on the GPU (klnmf_matmul): float32 if both operands are float32,
    float64 otherwise -- the reconstruction product of learner.py:80-84."""
    A = np.asarray(A)
    B = np.asarray(B)
    if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[0]:
        raise ValueError('shapes %s and %s not aligned' % (A.shape, B.shape))
    dt = np.float32 if (A.dtype == np.float32 and B.dtype == np.float32) else np.float64
    A = np.ascontiguousarray(A, dtype=dt)
    B = np.ascontiguousarray(B, dtype=dt)
    C = np.empty((A.shape[0], B.shape[1]), dtype=dt)
    lib = load()
    _check(lib.klnmf_matmul(device, DT_F32 if dt == np.float32 else DT_F64, A.shape[0], B.shape[1], A.shape[1],
                            A.ctypes.data_as(_c.c_void_p), B.ctypes.data_as(_c.c_void_p),
                            C.ctypes.data_as(_c.c_void_p)))
    return C


DIST_KL, DIST_REV_KL, DIST_SYM_KL, DIST_FROBENIUS, DIST_COSINE_DIFF = 0, 1, 2, 3, 4


def all_distances(A, B, metric, device=0):
    """[len(A), len(B)] matrix of metric(A[i], B[j]) on the GPU (klnmf_all_distances)."""
    A = np.asarray(A)
    B = np.asarray(B)
    if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[1]:
        raise ValueError('shapes %s and %s do not hold vectors of one length' % (A.shape, B.shape))
    dt = np.float32 if (A.dtype == np.float32 and B.dtype == np.float32) else np.float64
    A = np.ascontiguousarray(A, dtype=dt)
    B = np.ascontiguousarray(B, dtype=dt)
    out = np.empty((A.shape[0], B.shape[0]), dtype=dt)
    lib = load()
    _check(lib.klnmf_all_distances(device, DT_F32 if dt == np.float32 else DT_F64, int(metric), A.shape[0],
                                   B.shape[0], A.shape[1], A.ctypes.data_as(_c.c_void_p),
                                   B.ctypes.data_as(_c.c_void_p), out.ctypes.data_as(_c.c_void_p)))
    return out


def matmul_device(dA, lda, dB, ldb, dC, ldc, m, n, kk, f64=True, device=0):
    """C[m, n] = A[m, kk] . B[kk, n] between DEVICE matrices (pointers, row strides in elements): klnmf_matmul_device."""
    _check(load().klnmf_matmul_device(device, DT_F64 if f64 else DT_F32, int(m), int(n), int(kk), _c.c_void_p(dA), int(lda),
                                      _c.c_void_p(dB), int(ldb), _c.c_void_p(dC), int(ldc)))


def all_distances_device(dA, lda, dB, ldb, dout, na, nb, d, metric, f64=True, device=0):
    """out[na, nb] = metric(A[i], B[j]) between DEVICE matrices: klnmf_all_distances_device."""
    _check(load().klnmf_all_distances_device(device, DT_F64 if f64 else DT_F32, int(metric), int(na), int(nb), int(d),
                                             _c.c_void_p(dA), int(lda), _c.c_void_p(dB), int(ldb), _c.c_void_p(dout)))


Q_FP8_LOOP, Q_FP8_TILE_ITERS, Q_FP8_COL_ITERS, Q_RATIO_TILE_BYTES, Q_COMM_RANKS = 0, 1, 2, 3, 4
Q_W8_SATURATED, Q_W8_FALLBACKS, Q_RATIO_SATURATED, Q_RATIO_UNFIXED = 5, 6, 7, 8
Q_NO_NUM_EPS = 9
Q_MON_CHECKS, Q_MON_TRIPS, Q_MON_GAVE_UP = 10, 11, 12
Q_FP8_POLL_DUE = 13
QF_SUM_V, QF_NNZ_V, QF_MON_STAT, QF_MON_THRESHOLD = 0, 1, 2, 3
QF_KL_OVER_SUM_V = 9


def selftest(device=0):
    lib = load()
    failed = _c.c_int(-1)
    _check(lib.klnmf_selftest(device, ctypes.byref(failed)))
    return failed.value


class Context(object):
    """One GPU-resident KL-NMF problem (V, W, H on the device)."""

    # Handles of closed `pooled` contexts, per (precision, device): experiment.py runs hundreds of short fits and
    # transforms in sequence (experiment.py:158-180, 235-238), each through its own KLdivNMF object; creating and
    # destroying a native context (a HIP stream, ~25 device blocks) per call cost as much as a small transform itself.
    # A pooled context is handed back with its problem released (klnmf_release_problem: device blocks to the block
    # cache); the stream and the handle live on.
    _pool = {}
    _POOL_MAX = 4

    def __init__(self, precision='f64', device=0, stream=None, pooled=False):
        self._lib = load()
        self._h = _ctx_p()
        if isinstance(precision, str):
            precision = PRECISIONS[precision]
        self.precision = precision
        # the mode's canonical name (what device_data picks the fp32 / fp64 source copies by)
        self.precision_name = {PREC_F64: 'f64', PREC_F32: 'f32', PREC_BF16: 'f16'}[precision]
        self.n = self.f = self.k = 0
        self.cap = 0
        self._pool_key = (precision, int(device)) if (pooled and stream is None and os.environ.get('KLNMF_NO_POOL') != '1') else None
        if self._pool_key is not None:
            free = Context._pool.get(self._pool_key)
            if free:
                self._h = free.pop()
                return
        # stream: None -> the context creates its own stream; an integer hipStream_t handle otherwise, where
        # 0 is the device's default (null) stream -- torch's current stream unless the caller switched --
        # and is passed as KLNMF_STREAM_DEFAULT, because the C-ABI reads NULL as "no stream given".
        if stream is None:
            handle = None
        elif int(stream) == 0:
            handle = _c.c_void_p(STREAM_DEFAULT)
        else:
            handle = _c.c_void_p(int(stream))
        _check(self._lib.klnmf_create(ctypes.byref(self._h), device, precision, handle))

    # -- lifetime --
    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            key = getattr(self, '_pool_key', None)
            free = Context._pool.setdefault(key, []) if key is not None else None
            if free is not None and len(free) < Context._POOL_MAX and self._lib.klnmf_release_problem(self._h) == 0:
                free.append(self._h)
            else:
                self._lib.klnmf_destroy(self._h)
            self._h = _ctx_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def exact(self):
        return self.precision in (PREC_F64, PREC_F32)

    # -- data in --
    def set_problem(self, n, f, k, max_iter_capacity):
        _check(self._lib.klnmf_set_problem(self._h, n, f, k, max(1, max_iter_capacity)))
        self.n, self.f, self.k, self.cap = n, f, k, max(1, max_iter_capacity)

    def set_v_max(self, vmax):
        _check(self._lib.klnmf_set_v_max(self._h, float(vmax)))

    def reset_V(self):
        _check(self._lib.klnmf_reset_V(self._h))

    def upload_V(self, block, row0=0, col0=0, scale=1.0):
        """V[row0:, col0:] block = scale * block (any strides; row-major view
        is uploaded without a host copy when rows are contiguous)."""
        a = np.asarray(block)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        if a.ndim != 2:
            raise ValueError("2-D block expected")
        if a.shape[0] == 0 or a.shape[1] == 0:
            return
        if a.strides[1] != a.itemsize or a.strides[0] % a.itemsize or \
                a.strides[0] < a.shape[1] * a.itemsize:
            a = np.ascontiguousarray(a)
        ld = a.strides[0] // a.itemsize
        _check(self._lib.klnmf_upload_V(self._h, a.ctypes.data, _np_dtype_code(a),
                                        a.shape[0], a.shape[1], ld, row0, col0,
                                        float(scale)))

    def upload_blocks(self, blocks, scales=None):
        """Upload hstack([s * b ...]) block by block (one fused scale/cast/place
        kernel per block) after fixing the 16-bit storage factor from the
        global maximum."""
        scales = [1.0] * len(blocks) if scales is None else scales
        vmax = 0.0
        for b, s in zip(blocks, scales):
            if b.size:
                vmax = max(vmax, float(s) * float(np.max(b)))
        self.set_v_max(vmax)
        col = 0
        for b, s in zip(blocks, scales):
            self.upload_V(b, row0=0, col0=col, scale=s)
            col += b.shape[1]

    def upload_V_device(self, dev_ptr, rows, cols, ld, row0=0, col0=0, scale=1.0):
        _check(self._lib.klnmf_upload_V_device(self._h, _c.c_void_p(dev_ptr), rows, cols, ld,
                                               row0, col0, float(scale)))

    # ---- CSR input (exact modes) ----
    def set_problem_sparse(self, X, k, max_iter_capacity):
        """X: scipy CSR (explicit zeros are dropped, indices sorted -- nmf.py:66 does the same).  Uploads the
        structure in CSR and CSC order and the values."""
        import scipy.sparse as sp
        X = sp.csr_matrix(X, copy=True)
        X.eliminate_zeros()
        X.sort_indices()
        n, f = X.shape
        nnz = int(X.nnz)
        dt = np.float32 if X.dtype == np.float32 else np.float64
        _check(self._lib.klnmf_set_problem_sparse(self._h, n, f, k, int(max_iter_capacity), nnz))
        self.n, self.f, self.k, self.cap = n, f, k, int(max_iter_capacity)
        self.nnz = nnz
        indptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(X.indices, dtype=np.int64)
        data = np.ascontiguousarray(X.data, dtype=dt)
        # CSC order of the same entries: a stable sort of the CSR entries by column
        perm = np.argsort(indices, kind='stable').astype(np.int64)
        rows_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
        csc_rows = np.ascontiguousarray(rows_of[perm])
        csc_indptr = np.zeros(f + 1, dtype=np.int64)
        np.cumsum(np.bincount(indices, minlength=f), out=csc_indptr[1:])
        p = lambda a: a.ctypes.data_as(_c.c_void_p)
        _check(self._lib.klnmf_upload_csr(self._h, DT_F32 if dt == np.float32 else DT_F64, p(indptr), p(indices),
                                          p(data), p(csc_indptr), p(csc_rows), p(perm)))
        self._csr = (indptr, indices)
        return X

    def get_Q_values(self, dtype=np.float64):
        out = np.empty(self.nnz, dtype=dtype)
        _check(self._lib.klnmf_get_Q_values(self._h, out.ctypes.data, _np_dtype_code(out)))
        return out

    def upload_V_device_rows(self, dev_ptr, row_idx_ptr, rows, cols, ld, row0=0, col0=0, scale=1.0):
        """Rows row_idx[0..rows) (int64 device array) of a device-resident fp32 matrix."""
        _check(self._lib.klnmf_upload_V_device_rows(self._h, _c.c_void_p(dev_ptr), _c.c_void_p(row_idx_ptr),
                                                    rows, cols, ld, row0, col0, float(scale)))

    def upload_V_device_rows_dt(self, dev_ptr, f64, row_idx_ptr, rows, cols, ld, row0=0, col0=0, scale=1.0):
        """The same for a float32 or float64 device-resident matrix; row_idx_ptr 0: rows 0 .. rows - 1."""
        _check(self._lib.klnmf_upload_V_device_rows_dt(self._h, _c.c_void_p(dev_ptr), DT_F64 if f64 else DT_F32,
                                                       _c.c_void_p(row_idx_ptr) if row_idx_ptr else None, rows, cols, ld,
                                                       row0, col0, float(scale)))

    def set_H_device(self, dev_ptr, f64, ld, col0, ncols, last=True):
        """H[:, col0 : col0 + ncols] from a device-resident dictionary (klnmf_set_H_device)."""
        _check(self._lib.klnmf_set_H_device(self._h, _c.c_void_p(dev_ptr), DT_F64 if f64 else DT_F32, int(ld), int(col0),
                                            int(ncols), 1 if last else 0))

    def get_W_device(self, dev_ptr, f64, ld):
        """The coefficients into device memory ([n, k], rows ld apart): klnmf_get_W_device."""
        _check(self._lib.klnmf_get_W_device(self._h, _c.c_void_p(dev_ptr), DT_F64 if f64 else DT_F32, int(ld)))

    def set_H(self, H):
        H = _as_float_array(H)
        assert H.shape == (self.k, self.f)
        _check(self._lib.klnmf_set_H(self._h, H.ctypes.data, _np_dtype_code(H)))

    def set_W(self, W):
        W = _as_float_array(W)
        assert W.shape == (self.n, self.k)
        _check(self._lib.klnmf_set_W(self._h, W.ctypes.data, _np_dtype_code(W)))

    def set_Q(self, Q):
        Q = _as_float_array(Q)
        assert Q.shape == (self.n, self.f)
        _check(self._lib.klnmf_set_Q(self._h, Q.ctypes.data, _np_dtype_code(Q)))

    # -- the path --
    def init_W(self):
        _check(self._lib.klnmf_init_W(self._h))

    def run(self, max_iter, fit, tol_abs):
        """Returns (errors list, n_done, stopped)."""
        max_iter = int(max_iter)
        if max_iter > self.cap:
            raise ValueError("max_iter exceeds capacity")
        errs = np.zeros(max(1, max_iter), dtype=np.float64)
        nd = _i64(0)
        stopped = _c.c_int(0)
        _check(self._lib.klnmf_run(self._h, max_iter, 1 if fit else 0, float(tol_abs),
                                   errs.ctypes.data_as(_c.POINTER(_c.c_double)),
                                   ctypes.byref(nd), ctypes.byref(stopped)))
        return [float(e) for e in errs[:nd.value]], nd.value, bool(stopped.value)

    # ---- native collective path (RCCL inside the C-ABI) ----
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        _check(load().klnmf_comm_unique_id(buf))
        return bytes(buf.raw)

    def comm_init(self, uid, rank, nranks):
        assert len(uid) == COMM_ID_BYTES
        buf = ctypes.create_string_buffer(bytes(uid), COMM_ID_BYTES)
        _check(self._lib.klnmf_comm_init(self._h, buf, int(rank), int(nranks)))

    def comm_destroy(self):
        _check(self._lib.klnmf_comm_destroy(self._h))

    def comm_max(self, value):
        v = ctypes.c_double(float(value))
        _check(self._lib.klnmf_comm_max(self._h, ctypes.byref(v)))
        return float(v.value)

    def run_sharded(self, n_total, max_iter, fit, tol):
        errs = (ctypes.c_double * max(1, int(max_iter)))()
        n_done = _i64(0)
        stopped = ctypes.c_int(0)
        _check(self._lib.klnmf_run_sharded(self._h, int(n_total), int(max_iter), 1 if fit else 0, float(tol),
                                           errs, ctypes.byref(n_done), ctypes.byref(stopped)))
        nd = int(n_done.value)
        return [float(errs[i]) for i in range(min(nd, int(max_iter)))], nd, bool(stopped.value)

    def loop_begin(self, sum_v_all=None, cells_all=None, nnz_all=None, fp8_shape_all=None):
        """klnmf_loop_begin; with the all-reduced sum of V, element count (and count of entries > 0, and the conjunction of the
        ranks' fp8_shape_ok()): klnmf_loop_begin_sharded(_nnz) / klnmf_loop_begin_agreed (one rank of a row-sharded problem --
        every rank then takes the same fp8 decision)."""
        if sum_v_all is None:
            _check(self._lib.klnmf_loop_begin(self._h))
        elif fp8_shape_all is not None:
            _check(self._lib.klnmf_loop_begin_agreed(self._h, float(sum_v_all), float(cells_all),
                                                     float(-1.0 if nnz_all is None else nnz_all), 1 if fp8_shape_all else 0))
        elif nnz_all is None:
            _check(self._lib.klnmf_loop_begin_sharded(self._h, float(sum_v_all), float(cells_all)))
        else:
            _check(self._lib.klnmf_loop_begin_sharded_nnz(self._h, float(sum_v_all), float(cells_all), float(nnz_all)))

    def fp8_shape_ok(self):
        """This problem's shape allows fp8 ratio tiles (klnmf_query KLNMF_Q_RATIO_TILE_BYTES == 1)."""
        return self.query(Q_RATIO_TILE_BYTES) == 1

    def sum_V(self):
        """Sum of the uploaded V as stored (klnmf_query_f64 KLNMF_QF_SUM_V)."""
        v = _c.c_double(0.0)
        _check(self._lib.klnmf_query_f64(self._h, 0, ctypes.byref(v)))
        return float(v.value)

    def nnz_V(self):
        """How many entries of the uploaded V are > 0 as stored (klnmf_query_f64 KLNMF_QF_NNZ_V)."""
        v = _c.c_double(0.0)
        _check(self._lib.klnmf_query_f64(self._h, 1, ctypes.byref(v)))
        return float(v.value)

    def run_more(self, iters, fit=True, tol_abs=0.0):
        """`iters` whole iterations of the open loop, enqueued as klnmf_run does (klnmf_run_more)."""
        _check(self._lib.klnmf_run_more(self._h, int(iters), 1 if fit else 0, float(tol_abs)))

    def iter_rowpass(self, fit=True):
        _check(self._lib.klnmf_iter_rowpass(self._h, 1 if fit else 0))

    def iter_decide(self, tol_abs):
        _check(self._lib.klnmf_iter_decide(self._h, float(tol_abs)))

    def iter_colpass(self):
        _check(self._lib.klnmf_iter_colpass(self._h))

    def iter_colpass_part(self, part):
        """Column pass + numerator of ONE column part of the split layout (`exchange_parts`): the caller exchanges the part
        while the next one computes."""
        _check(self._lib.klnmf_iter_colpass_part(self._h, int(part)))

    def exchange_parts(self):
        """[(element offset, element count, first column, columns)] of the numerator's column parts (one whole-matrix part
        unless KLNMF_COMM_PARTS split the problem)."""
        n = _c.c_int(0)
        arr = lambda: (_i64 * 4)()
        off, cnt, c0, nc = arr(), arr(), arr(), arr()
        _check(self._lib.klnmf_exchange_parts(self._h, ctypes.byref(n), off, cnt, c0, nc))
        return [(int(off[p]), int(cnt[p]), int(c0[p]), int(nc[p])) for p in range(n.value)]

    def iter_update_H(self):
        _check(self._lib.klnmf_iter_update_H(self._h))

    def iter_advance(self):
        _check(self._lib.klnmf_iter_advance(self._h))

    def loop_end(self, max_iter):
        errs = np.zeros(max(1, int(max_iter)), dtype=np.float64)
        nd = _i64(0)
        stopped = _c.c_int(0)
        _check(self._lib.klnmf_loop_end(self._h, errs.ctypes.data_as(_c.POINTER(_c.c_double)),
                                        ctypes.byref(nd), ctypes.byref(stopped)))
        return [float(e) for e in errs[:nd.value]], nd.value, bool(stopped.value)

    def exchange_buffers(self):
        """(loss_ptr, numer_ptr, numer_count, numer_is_f64) device pointers."""
        lp, npt = _c.c_void_p(), _c.c_void_p()
        cnt = _i64(0)
        is64 = _c.c_int(0)
        _check(self._lib.klnmf_exchange_buffers(self._h, ctypes.byref(lp), ctypes.byref(npt),
                                                ctypes.byref(cnt), ctypes.byref(is64)))
        return lp.value, npt.value, cnt.value, bool(is64.value)

    def exchange_layout(self):
        stride, valid = _i64(0), _i64(0)
        _check(self._lib.klnmf_exchange_layout(self._h, ctypes.byref(stride), ctypes.byref(valid)))
        return int(stride.value), int(valid.value)

    def bind_exchange(self, loss_ptr, numer_ptr):
        _check(self._lib.klnmf_bind_exchange(self._h, _c.c_void_p(loss_ptr),
                                             _c.c_void_p(numer_ptr)))

    def error(self):
        out = _c.c_double(0)
        _check(self._lib.klnmf_error(self._h, ctypes.byref(out)))
        return out.value

    def loss_terms(self):
        out = (ctypes.c_double * 4)()
        _check(self._lib.klnmf_loss_terms(self._h, out))
        return [float(v) for v in out]

    def update(self, fit=True):
        _check(self._lib.klnmf_update(self._h, 1 if fit else 0))

    def set_ratio_eps(self, eps):
        _check(self._lib.klnmf_set_ratio_eps(self._h, float(eps)))

    def step_Q(self):
        _check(self._lib.klnmf_step_Q(self._h))

    def step_W(self):
        _check(self._lib.klnmf_step_W(self._h))

    def step_H(self):
        _check(self._lib.klnmf_step_H(self._h))

    def generalized_kl(self, x, y, eps):
        x = _as_float_array(x).ravel()
        y = _as_float_array(y).ravel()
        if x.dtype != y.dtype:
            x = x.astype(np.float64)
            y = y.astype(np.float64)
        out = _c.c_double(0)
        _check(self._lib.klnmf_generalized_kl(self._h, x.ctypes.data, y.ctypes.data,
                                              _np_dtype_code(x), x.size, float(eps),
                                              ctypes.byref(out)))
        return out.value

    # -- data out --
    def _get(self, fn, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        _check(fn(self._h, out.ctypes.data, _np_dtype_code(out)))
        return out

    def get_W(self, dtype=np.float64):
        return self._get(self._lib.klnmf_get_W, (self.n, self.k), dtype)

    def get_H(self, dtype=np.float64):
        return self._get(self._lib.klnmf_get_H, (self.k, self.f), dtype)

    def get_Q(self, dtype=np.float64):
        return self._get(self._lib.klnmf_get_Q, (self.n, self.f), dtype)

    # -- measurement --
    def profile_enable(self, on=True):
        """True / 1: every iteration's row- and column-pass launches bracketed by HIP events; N > 1: every N-th iteration's."""
        _check(self._lib.klnmf_profile_enable(self._h, int(on) if on else 0))

    def profile_read(self, reset=True):
        rn, cn = _i64(0), _i64(0)
        rms, cms = _c.c_double(0), _c.c_double(0)
        _check(self._lib.klnmf_profile_read(self._h, ctypes.byref(rn), ctypes.byref(rms),
                                            ctypes.byref(cn), ctypes.byref(cms),
                                            1 if reset else 0))
        tn, tr, tms = _i64(0), _i64(0), _c.c_double(0)
        _check(self._lib.klnmf_profile_read_tail(self._h, ctypes.byref(tn), ctypes.byref(tms), ctypes.byref(tr),
                                                 1 if reset else 0))
        return {'rowpass_launches': rn.value, 'rowpass_ms': rms.value,
                'colpass_launches': cn.value, 'colpass_ms': cms.value,
                'tail_launches': tn.value, 'tail_ms': tms.value, 'tail_rows': tr.value}

    def synchronize(self):
        _check(self._lib.klnmf_synchronize(self._h))

    def query(self, what):
        """klnmf_query: what the context decided / what its last loop ran (Q_* items above)."""
        v = _i64(0)
        _check(self._lib.klnmf_query(self._h, int(what), ctypes.byref(v)))
        return int(v.value)

    def fp8_report(self):
        """{'allowed', 'tile_iterations', 'column_pass_iterations'} of the last loop -- read from the library, not re-derived."""
        return {'allowed': bool(self.query(Q_FP8_LOOP)), 'tile_iterations': self.query(Q_FP8_TILE_ITERS),
                'column_pass_iterations': self.query(Q_FP8_COL_ITERS),
                # e4m3 saturation: counted and kept out of the result (see include/klnmf.h)
                'w_image_saturated': self.query(Q_W8_SATURATED), 'w_image_fallback_iterations': self.query(Q_W8_FALLBACKS),
                'ratio_saturated': self.query(Q_RATIO_SATURATED), 'ratio_unfixed': self.query(Q_RATIO_UNFIXED),
                # the loop's update passes formed the ratio without the numerator's eps (large-mean data; loss corrected exactly)
                'no_numerator_eps': bool(self.query(Q_NO_NUM_EPS)),
                # the in-loop monitor (csrc/monitor.hip.h): measured relative error of the sampled H-numerator entries
                'monitor_checks': self.query(Q_MON_CHECKS), 'monitor_trips': self.query(Q_MON_TRIPS),
                'gave_up': bool(self.query(Q_MON_GAVE_UP)), 'monitor_statistic': self.query_f64(QF_MON_STAT),
                'monitor_threshold': self.query_f64(QF_MON_THRESHOLD),
                'monitor_parts': [self.query_f64(4 + i) for i in range(3)],
                'monitor_min_spread': self.query_f64(7), 'monitor_spread_threshold': self.query_f64(8),
                # the loop's final KL / sum(V) (16-bit modes; -1: none): the data condition of the f16 operands' accuracy envelope
                'kl_over_sum_v': self.query_f64(QF_KL_OVER_SUM_V)}

    def fp8_poll_due(self):
        """Loops in pieces on row shards: the next `iter_advance` reads the all-reduced count in loss[1] (KLNMF_Q_FP8_POLL_DUE)."""
        return bool(self.query(Q_FP8_POLL_DUE))

    def query_f64(self, what):
        v = _c.c_double(0.0)
        _check(self._lib.klnmf_query_f64(self._h, int(what), ctypes.byref(v)))
        return float(v.value)
