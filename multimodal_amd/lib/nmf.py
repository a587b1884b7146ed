"""KL-divergence (I-divergence) NMF on MI355X -- host side.

Same public surface as reference multimodal/lib/nmf.py (`KLdivNMF`, `_scale`,
`_special_sparse_dot`, `check_non_negative`), so `multimodal/learner.py`,
`experiment.py` and the sample scripts can use it unchanged.  All arithmetic of
the path -- W0 = X.H0^T, the loss, the ratio Q, the W and H rules, the row
normalisation and the stop rule of the loop -- runs in the HIP kernels of
`csrc/` through the C-ABI of `include/klnmf.h`.  This module only validates
input (the reference's ValueErrors are raised before anything is uploaded),
moves arrays, and keeps the reference's object protocol (`components_`,
`_init_dictionary`, `return_errors`, the stderr warning).

There is no CPU implementation behind these calls: without the HIP extension or
a gfx950 device they raise.

Behaviour kept on purpose: `scale_W` passed to
fit/fit_transform/transform is accepted and ignored (q1, nmf.py:222); the H
rule pairs the ratio of the OLD W with the NEW W (q2, nmf.py:251-256, done that
way inside the kernels); the random initial dictionary comes from the global
`np.random` stream (q3, nmf.py:150); the loop's eps is the hard-coded 1e-8 (q4);
the loss is recorded before each update and the breaking iteration's loss is not
appended (q5, nmf.py:214-220).

Extra (non-reference) constructor arguments: `precision` ('f64' default =
the reference's float64 arithmetic; 'f32'; 'f16' (= 'bf16') = MFMA fast path with V
stored as power-of-two-scaled fp16; 'auto'), `device`.  Environment: KLNMF_PRECISION, KLNMF_DEVICE.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

from .array_utils import normalize_sum
from .sklearn_utils import atleast2d_or_csr
from .. import _native


def _default_precision():
    return os.environ.get('KLNMF_PRECISION', 'f64')


# precision='auto' (KLNMF_PRECISION=auto): the reference's own arithmetic (f64: results equal to the reference's) where a
# fit is cheap anyway, the fp16-operand MFMA path (final KL within 1e-4) from this many multiply-adds per W.H on -- 2e9 is
# 10 000 x 4096 at k = 50: below it an f64 iteration is well under a millisecond.  The default stays 'f64' (SURVEY section 5:
# defaults must reproduce the reference's results).
AUTO_F16_WORK = 2e9
# ... and only on shapes inside the 16-bit mode's accuracy envelope: its rounding noise averages out over the columns (W rule), the
# rows (H rule) and the components (W.H), and with few of either the averaging is weak -- measured against the oracle (round 5,
# profiles/r05_monitor_calibration.txt): f = 8 ends 2.4e-4 off, f = 64 is up to 4.5e-4 off in mid-descent, k <= 8 on low-noise data
# up to 8e-4 after 150 iterations; from f = 256 and k = 16 on every class stays within 1e-4.  Smaller shapes run in fp32 (1e-6).
AUTO_F16_MIN_F = 256
AUTO_F16_MIN_K = 16


# ... and on the data: a fit whose final KL / sum(V) falls below this has so little residual left that the f16 operands' own
# rounding noise (about 3e-4 / sqrt(terms) per product) can pass 1e-4 of the loss (measured: low-noise rank-8 data, 160 000 rows,
# 150 iterations: 8e-4, at KL / sum(V) = 4e-4; the reference's fixtures G18 / G19 sit at 5.5e-4 and 2.9e-3; the BASELINE
# configurations' data at 4.8e-3 .. 9.5e-3).  The library reports the ratio of every loop (klnmf_query_f64
# KLNMF_QF_KL_OVER_SUM_V); `KLdivNMF.last_fp8_report['outside_f16_envelope']` and one stderr line say when a fit ended below it.
F16_MIN_KL_OVER_SUM_V = 3e-3

MAX_K_MFMA = 512          # the 16-bit MFMA kernels hold a wave's accumulators of all components in registers: k <= 512
MAX_ROWS_EXACT = 65535 * 64   # the exact modes' row tiles ride on gridDim.y (csrc/api_context.hip: KLNMF_ERR_UNSUPP beyond)

_NOTED = set()


def _note_once(key, text):
    """One stderr line per process and cause when a problem runs in another arithmetic than the one asked for (never silently)."""
    if key not in _NOTED:
        _NOTED.add(key)
        sys.stderr.write(text)


def resolve_precision(precision, n, f, k):
    """The arithmetic one problem runs in.  'auto' by size (above); the 16-bit modes hand k > 512 to the fp32 kernels of the
    same library (LDS-tiled VALU GEMMs: any k, at least the 16-bit mode's accuracy) instead of refusing the problem -- and
    say so once on stderr (the fp32 kernels are an order of magnitude slower than the MFMA path)."""
    if precision == 'auto':
        if float(n) * float(f) * float(k) < AUTO_F16_WORK:
            precision = 'f64'
        elif f >= AUTO_F16_MIN_F and k >= AUTO_F16_MIN_K:
            precision = 'f16'
        elif n > MAX_ROWS_EXACT:
            # outside the 16-bit mode's envelope, but beyond what one context of the exact modes holds: the 16-bit path runs
            # it (as 'auto' did before the envelope rule existed) and says what that means
            _note_once(('auto-rows', f < AUTO_F16_MIN_F, k < AUTO_F16_MIN_K),
                       "KLdivNMF: precision='auto': %d rows exceed one fp32 context (%d); the problem runs with precision='f16' although "
                       "f = %d, k = %d lie outside its accuracy envelope (f >= %d, k >= %d: final KL within 1e-4 of the reference's; "
                       "here up to 5e-4) -- shard the rows for fp32\n"
                       % (n, MAX_ROWS_EXACT, f, k, AUTO_F16_MIN_F, AUTO_F16_MIN_K))
            precision = 'f16'
        else:
            precision = 'f32'
    elif (_native.PRECISIONS[precision] == _native.PREC_BF16 and k <= MAX_K_MFMA
          and (f < AUTO_F16_MIN_F or k < AUTO_F16_MIN_K)):
        # an explicit 16-bit mode on a shape outside its envelope is honoured -- and never silent about it
        _note_once(('envelope', precision, f < AUTO_F16_MIN_F, k < AUTO_F16_MIN_K),
                   "KLdivNMF: precision=%r on f = %d, k = %d: outside the 16-bit mode's accuracy envelope (f >= %d and k >= %d keep "
                   "the final KL within 1e-4 of the reference's; measured up to 5e-4 below) -- precision='auto' or 'f32' keeps 1e-6\n"
                   % (precision, f, k, AUTO_F16_MIN_F, AUTO_F16_MIN_K))
    code = _native.PRECISIONS[precision]
    if code == _native.PREC_BF16 and k > MAX_K_MFMA:
        _note_once(('k', precision), "KLdivNMF: precision=%r holds k <= %d; k = %d runs on the fp32 kernels (precision='f32')\n"
                   % (precision, MAX_K_MFMA, k))
        return 'f32'
    return precision


def sparse_precision(precision):
    """The arithmetic CSR input runs in: the reference's sparse branch (ratio on the stored entries only, nmf.py:52-70,
    301-308, 331-334) exists as SDDMM + SpMM kernels in the exact modes.  'f64' / 'auto' -> f64, 'f32' -> f32; the 16-bit
    modes run it in fp32 (never the dense rule on a densified matrix, which is a different algorithm: off X's structure the
    dense ratio is eps / (W.H + eps), not 0) and say so once on stderr."""
    if precision in ('auto', 'f64'):
        return 'f64'
    if _native.PRECISIONS[precision] == _native.PREC_F32:
        return 'f32'
    _note_once(('csr', precision), "KLdivNMF: CSR input with precision=%r runs the reference's sparse branch on the fp32 "
               "sparse kernels (precision='f32'); pass a dense array for the 16-bit MFMA path\n" % (precision,))
    return 'f32'


def _default_device():
    return int(os.environ.get('KLNMF_DEVICE', '0'))


def check_non_negative(X, whom):
    """ValueError on negative entries (reference nmf.py:23-26)."""
    X = X.data if sp.issparse(X) else X
    if (X < 0).any():
        raise ValueError("Negative values in data passed to %s" % whom)


def _scale(matrix, factors, axis=0):
    """Scale the columns (axis=0) or the lines (axis=1) of a 2-D array by a
    vector (reference nmf.py:29-49; host helper, not on the GPU loop)."""
    if not (len(matrix.shape) == 2):
        raise ValueError("Wrong array shape: %s, should have only 2 dimensions."
                         % str(matrix.shape))
    if axis not in (0, 1):
        raise ValueError('Wrong axis, should be 0 (scaling lines) '
                         'or 1 (scaling columns).')
    factors = np.squeeze(np.asarray(factors))
    if axis == 1:
        factors = factors[:, np.newaxis]
    return np.multiply(matrix, factors)


def _special_sparse_dot(a, b, refmat):
    """a.b sampled on the non-zeros of refmat, returned as CSR with refmat's
    structure (reference nmf.py:52-70, including its `eliminate_zeros` side
    effect on refmat).  Host scipy helper kept for callers of the module function; the
    fit / transform / error of CSR input run the same SDDMM on the device (csrc/sparse.hip.h)."""
    refmat.eliminate_zeros()
    ii, jj = refmat.nonzero()
    vals = np.einsum('ij,ij->i', a[ii, :], b.T[jj, :])
    return sp.coo_matrix((vals, (ii, jj)), shape=refmat.shape).tocsr()


def _dense(X):
    """Dense ndarray of validated input (only reached with CSR input by the single-step helpers that are handed an
    explicit dense Q; fit / transform / error / _update of CSR input take the sparse kernels: `_sparse_route`)."""
    if sp.issparse(X):
        return np.asarray(X.toarray())
    return np.asarray(X)


def _csr_of(blocks, coefs):
    """CSR of hstack([c * b ...]) (safe_hstack keeps the stack sparse if any block is, array_utils.py:5-9)."""
    mats = [sp.csr_matrix(b) * float(c) if sp.issparse(b) else sp.csr_matrix(np.asarray(b) * float(c))
            for b, c in zip(blocks, coefs)]
    X = mats[0] if len(mats) == 1 else sp.hstack(mats, format='csr')
    return sp.csr_matrix(X)


def _out_dtype(*arrays):
    """The reference computes in float32 only if every operand is float32."""
    if all(np.asarray(a).dtype == np.float32 for a in arrays):
        return np.float32
    return np.float64


class KLdivNMF(object):
    """Non negative factorization with Kullback Leibler divergence cost
    (Lee & Seung multiplicative updates), GPU implementation of reference
    nmf.py:73-351.

    Parameters as in the reference: n_components, tol (1e-6), max_iter (200),
    eps (1e-8; only used by `scale`), subit (unused), random_state (stored,
    unused -- the reference never reads it either).
    """

    def __init__(self, n_components=None, tol=1e-6, max_iter=200, eps=1.e-8,
                 subit=10, random_state=None, precision=None, device=None):
        self.n_components = n_components
        self._init_dictionary = None
        self.random_state = random_state
        self.tol = tol
        self.max_iter = max_iter
        self.eps = eps
        self.subit = subit
        self.precision = precision if precision is not None else _default_precision()
        self.device = device if device is not None else _default_device()
        self.last_fp8_report = None         # set by every loop: what it ran on e4m3 operands (klnmf_query)

    # ------------------------------------------------------------ helpers ---
    def _sparse_route(self, *blocks):
        """CSR input ALWAYS runs the reference's sparse branch (ratio on the stored entries only, nmf.py:52-70,
        301-308, 331-334): SDDMM + SpMM kernels of the exact modes, in the arithmetic `sparse_precision` names."""
        return any(sp.issparse(b) for b in blocks)

    def _context(self, exact=False, shape=None, sparse=False):
        prec = self.precision
        if sparse:                                # CSR input: f64 ('auto', 'f64') or f32 (everything else), never densified
            prec = sparse_precision(prec)
        elif prec == 'auto' and shape is None:    # single steps and loss evaluations: exact
            prec = 'f64'
        elif shape is not None:                   # decided per problem (shape = (n, f, k)): 'auto' by size, k > 512 -> fp32 kernels
            prec = resolve_precision(prec, *shape)
        if exact and _native.PRECISIONS[prec] not in (_native.PREC_F64, _native.PREC_F32):
            prec = 'f64'
        return _native.Context(precision=prec, device=self.device, pooled=True)

    @classmethod
    def _exact_context(cls):
        return _native.Context(precision=os.environ.get('KLNMF_STEP_PRECISION', 'f64'),
                               device=_default_device(), pooled=True)

    def _init_H(self, n_features):
        """Initial dictionary (reference nmf.py:149-155)."""
        if self._init_dictionary is None:
            return normalize_sum(np.abs(np.random.random(
                (self.n_components, n_features))) + .01, axis=1)
        assert(self._init_dictionary.shape ==
               (self.n_components, n_features))
        return self._init_dictionary

    # --------------------------------------------------------------- loop ---
    def fit_transform(self, X, y=None, weights=1., _fit=True,
                      return_errors=False, scale_W=False):
        """Learn a NMF model for X and return the transformed data
        (reference nmf.py:159-230).  `y`, `weights`, `scale_W` are accepted and
        ignored exactly as in the reference."""
        X = atleast2d_or_csr(X)
        check_non_negative(X, "NMF.fit")
        return self._fit_blocks([X], [1.], _fit=_fit, return_errors=return_errors)

    def _fit_blocks(self, blocks, coefs, _fit=True, return_errors=False):
        """fit_transform of hstack([c * b for b, c in zip(blocks, coefs)])
        without building the stacked matrix on the host: each modality block is
        scaled, cast and placed by the upload kernel (learner.py:53-56 fused)."""
        if self._sparse_route(*blocks):
            X = _csr_of(blocks, coefs)
            return self._fit_uploaded(X.shape[0], X.shape[1], None,
                                      lambda H_init: np.float32 if (X.dtype == np.float32 and H_init.dtype == np.float32)
                                      else np.float64, _fit=_fit, return_errors=return_errors, sparse_X=X)
        blocks = [_dense(b) for b in blocks]
        n_samples = blocks[0].shape[0]
        n_features = sum(b.shape[1] for b in blocks)
        return self._fit_uploaded(n_samples, n_features, lambda ctx: ctx.upload_blocks(blocks, coefs),
                                  lambda H_init: _out_dtype(H_init, *blocks), _fit=_fit,
                                  return_errors=return_errors)

    def _fit_uploaded(self, n_samples, n_features, upload, out_dtype_of, _fit=True, return_errors=False,
                      sparse_X=None):
        """The loop of nmf.py:159-230 on a matrix that `upload(ctx)` places in the context: host blocks
        (`_fit_blocks`) or rows gathered from device-resident data (`device_data.DeviceDataset`)."""
        if not self.n_components:
            self.n_components = n_features
        H_init = self._init_H(n_features)
        k = self.n_components
        max_iter = int(self.max_iter)
        out_dtype = out_dtype_of(H_init)

        with self._context(shape=None if sparse_X is not None else (n_samples, n_features, k),
                           sparse=sparse_X is not None) as ctx:
            if sparse_X is not None:
                ctx.set_problem_sparse(sparse_X, k, max_iter)
            else:
                ctx.set_problem(n_samples, n_features, k, max_iter)
                upload(ctx)
            ctx.set_H(H_init)
            ctx.init_W()                       # W0 = X . H_init^T (nmf.py:156)
            if _fit:
                self.components_ = H_init      # nmf.py:203-204
            elif self.components_ is not H_init:
                ctx.set_H(self.components_)    # loop runs on components_ (nmf.py:214)
            tol_abs = self.tol * n_samples * n_features      # nmf.py:207
            errors, n_done, stopped = ctx.run(max_iter, _fit, tol_abs)
            # what the loop ran on e4m3 operands (16-bit modes, large problems; all zero otherwise) -- as the library
            # reports it (klnmf_query); no reference counterpart
            self.last_fp8_report = ctx.fp8_report()
            self._check_f16_envelope(ctx, n_samples, n_features, k)
            W = ctx.get_W(dtype=out_dtype)
            if _fit and n_done > 0:
                self.components_ = ctx.get_H(dtype=out_dtype)

        n_iter = n_done + 1 if stopped else max_iter
        if max_iter > 0 and n_iter == max_iter and tol_abs > 0:   # nmf.py:224-225
            sys.stderr.write("Warning: Iteration limit reached during fit\n")
        if return_errors:
            return W, errors
        return W

    def _check_f16_envelope(self, ctx, n, f, k):
        """The run-time half of the 16-bit mode's envelope: the library holds loss and sum(V) on the device; a loop that ended
        with KL / sum(V) below `F16_MIN_KL_OVER_SUM_V` is reported in `last_fp8_report` and said once on stderr (the shape
        half is `resolve_precision`'s).  No reference counterpart (nmf.py has one arithmetic)."""
        rep = self.last_fp8_report
        if rep is None:
            return
        if ctx.precision != _native.PREC_BF16:
            rep['outside_f16_envelope'] = None      # (not a 16-bit loop: the key is there, the question does not arise)
            return
        r = rep.get('kl_over_sum_v', -1.0)
        reasons = []
        if f < AUTO_F16_MIN_F:
            reasons.append('f < %d' % AUTO_F16_MIN_F)
        if k < AUTO_F16_MIN_K:
            reasons.append('k < %d' % AUTO_F16_MIN_K)
        if 0.0 <= r < F16_MIN_KL_OVER_SUM_V:
            reasons.append('KL / sum(V) = %.2e < %.0e' % (r, F16_MIN_KL_OVER_SUM_V))
            _note_once(('residual',),
                       "KLdivNMF: this 16-bit fit ended with KL / sum(V) = %.2e (< %.0e): so little residual that the f16 operands' "
                       "rounding noise can exceed 1e-4 of the final KL (measured up to 8e-4) -- precision='f32' keeps 1e-6\n"
                       % (r, F16_MIN_KL_OVER_SUM_V))
        rep['outside_f16_envelope'] = reasons

    def fit(self, X, y=None, **params):
        """Learn a NMF model for X; returns self (reference nmf.py:259-273)."""
        self.fit_transform(X, **params)
        return self

    def transform(self, X, **params):
        """Coefficients of X for the fitted dictionary (reference
        nmf.py:275-291; leaves `_init_dictionary` set, like the reference)."""
        self._init_dictionary = self.components_
        params['_fit'] = False
        return self.fit_transform(X, **params)

    def _transform_blocks(self, blocks, coefs, return_errors=False):
        self._init_dictionary = self.components_
        return self._fit_blocks(blocks, coefs, _fit=False, return_errors=return_errors)

    # -------------------------------------------------------- single steps ---
    def _update(self, X, W, _fit=True, scale_W=False, eps=1.e-8):
        """One update iteration (reference nmf.py:232-257)."""
        sparse = self._sparse_route(X)
        Xd = None if sparse else _dense(X)
        if scale_W:
            # dead from every caller in the reference (nmf.py:246-250), kept
            W = _scale(normalize_sum(W, axis=1), np.asarray(X.sum(axis=1)).ravel(), axis=1)
        if (eps != 1.e-8 and not sparse and self.precision != 'auto'
                and _native.PRECISIONS[self.precision] >= _native.PREC_BF16):
            raise ValueError("the bf16 kernels use the reference's fixed eps = 1e-8")
        H = self.components_
        with self._context(sparse=sparse) as ctx:
            if sparse:
                Xc = ctx.set_problem_sparse(X, H.shape[0], 1)
                dt = _out_dtype(Xc.data, W, H)
            else:
                ctx.set_problem(Xd.shape[0], Xd.shape[1], H.shape[0], 1)
                ctx.upload_blocks([Xd])
                dt = _out_dtype(Xd, W, H)
            ctx.set_H(H)
            ctx.set_W(W)
            if eps != 1.e-8:
                ctx.set_ratio_eps(eps)
                ctx.step_Q()
                ctx.step_W()
                if _fit:
                    ctx.step_H()
            else:
                ctx.update(_fit)
            Wn = ctx.get_W(dtype=dt)
            if _fit:
                self.components_ = ctx.get_H(dtype=dt)
        return Wn

    def error(self, X, W, H=None, weights=1., eps=1.e-8):
        """generalized_KL(X, W.H) (reference nmf.py:297-310; `weights` and `eps`
        are ignored by the reference's dense branch too)."""
        X = atleast2d_or_csr(X)
        if H is None:
            H = self.components_
        with self._context(sparse=self._sparse_route(X)) as ctx:
            if self._sparse_route(X):
                ctx.set_problem_sparse(X, np.shape(H)[0], 1)      # nmf.py:301-308
            else:
                Xd = _dense(X)
                ctx.set_problem(Xd.shape[0], Xd.shape[1], np.shape(H)[0], 1)
                ctx.upload_blocks([Xd])
            ctx.set_H(H)
            ctx.set_W(W)
            return ctx.error()

    def scale(self, W, H, factors):
        """Scale W columns and H rows inversely (reference nmf.py:314-321)."""
        safe_factors = factors + self.eps
        s_W = _scale(W, safe_factors, axis=0)
        s_H = _scale(H, 1. / safe_factors, axis=1)
        return s_W, s_H

    @classmethod
    def _Q(cls, X, W, H, eps=1.e-8):
        """(X + eps) / (W.H + eps), element-wise (reference nmf.py:325-336).
        CSR input gives a CSR result on X's structure, as in the reference."""
        if sp.issparse(X):
            with cls._exact_context() as ctx:
                Xc = ctx.set_problem_sparse(X, np.shape(H)[0], 1)
                ctx.set_H(H)
                ctx.set_W(W)
                ctx.set_ratio_eps(eps)
                ctx.step_Q()
                q = ctx.get_Q_values(dtype=_out_dtype(Xc.data, W, H))
            return sp.csr_matrix((q, Xc.indices, Xc.indptr), shape=Xc.shape)
        Xd = _dense(X)
        with cls._exact_context() as ctx:
            ctx.set_problem(Xd.shape[0], Xd.shape[1], np.shape(H)[0], 1)
            ctx.upload_V(Xd)
            ctx.set_H(H)
            ctx.set_W(W)
            ctx.set_ratio_eps(eps)
            ctx.step_Q()
            return ctx.get_Q(dtype=_out_dtype(Xd, W, H))

    @classmethod
    def _step(cls, X, W, H, Q, eps, which):
        if sp.issparse(X) and Q is None:          # the sparse branch end to end (nmf.py:331-351)
            with cls._exact_context() as ctx:
                Xc = ctx.set_problem_sparse(X, np.shape(H)[0], 1)
                ctx.set_H(H)
                ctx.set_W(W)
                ctx.set_ratio_eps(eps)
                ctx.step_Q()
                dt = _out_dtype(Xc.data, W, H)
                if which == 'W':
                    ctx.step_W()
                    return ctx.get_W(dtype=dt)
                ctx.step_H()
                return ctx.get_H(dtype=dt)
        Xd = _dense(X)
        with cls._exact_context() as ctx:
            ctx.set_problem(Xd.shape[0], Xd.shape[1], np.shape(H)[0], 1)
            ctx.upload_V(Xd)
            ctx.set_H(H)
            ctx.set_W(W)
            if Q is None:
                ctx.set_ratio_eps(eps)
                ctx.step_Q()
            else:
                ctx.set_Q(_dense(Q))
            dt = _out_dtype(Xd, W, H)
            if which == 'W':
                ctx.step_W()
                return ctx.get_W(dtype=dt)
            ctx.step_H()
            return ctx.get_H(dtype=dt)

    @classmethod
    def _updated_W(cls, X, W, H, weights=1., Q=None, eps=1.e-8):
        """W * (Q.H^T) (reference nmf.py:338-343)."""
        return cls._step(X, W, H, Q, eps, 'W')

    @classmethod
    def _updated_H(cls, X, W, H, weights=1., Q=None, eps=1.e-8):
        """normalize_rows(H * (W^T.Q)) (reference nmf.py:345-351)."""
        return cls._step(X, W, H, Q, eps, 'H')
