"""CPU oracle for the KL-divergence NMF multiplicative-update path.

TEST INFRASTRUCTURE ONLY.  This module is a float64 numpy restatement of the
algorithm in the reference `multimodal/lib/nmf.py` (+ the two helpers it pulls
from `lib/metrics.py` and `lib/array_utils.py`, and the stacking / slicing
logic of `multimodal/learner.py`).  It exists so that `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` have
something to check the HIP path against.  Nothing under `multimodal_amd/` may
import it: the product path is HIP-only and fails loudly without its extension.

Parity pin: every function here is checked against golden vectors produced by
importing the reference itself in the build container
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`,
`tests/test_oracle_golden.py`), and against the three known-answer tests the
reference holds (`tests/test_metrics.py:48-54`, `tests/test_array_utils.py:31-41`,
`tests/test_nmf_kl.py:56-68`).  The dense arithmetic underneath the reference is
numpy/OpenBLAS (third party, unpinned: reference `setup.py:34`), so agreement is
to 1e-10 relative, not bitwise (dgemm summation order is implementation-defined).

The restatement keeps the reference's behaviour *including* its quirks
 and its redundant work (a separate W.H product for
the loss and for the ratio), because it is also the timed CPU baseline.
"""

import sys

import numpy as np

EPS_RATIO = 1.e-8      # hard-coded default of _Q / error / generalized_KL
                       # (reference nmf.py:232,297,325 and metrics.py:15)
EPS_NORMALIZE = 1.e-16  # reference array_utils.py:19


# ---------------------------------------------------------------- helpers ---

def normalize_sum(a, axis=0, eps=EPS_NORMALIZE):
    """a / (eps + sum(a, axis)) with the summed axis kept for broadcasting.

    Follows reference multimodal/lib/array_utils.py:19-22 (ValueError when the
    axis does not exist).
    """
    a = np.asarray(a)
    if axis >= a.ndim:
        raise ValueError
    return a / (eps + a.sum(axis=axis, keepdims=True))


def generalized_kl(x, y, eps=EPS_RATIO, axis=None):
    """I-divergence sum(x*log((x+eps)/(y+eps)) - x + y).

    Follows reference multimodal/lib/metrics.py:18-20.
    """
    x = np.asarray(x)
    y = np.asarray(y)
    ratio = (x + eps) / (y + eps)
    return (x * np.log(ratio) - x + y).sum(axis=axis)


def scale_matrix(matrix, factors, axis=0):
    """Scale columns (axis=0) or lines (axis=1) of a 2-D array by a vector.

    Follows reference multimodal/lib/nmf.py:29-49.
    """
    matrix = np.asarray(matrix)
    if matrix.ndim != 2:
        raise ValueError("expected a 2-D array, got shape %r" % (matrix.shape,))
    if axis not in (0, 1):
        raise ValueError("axis must be 0 or 1")
    factors = np.squeeze(np.asarray(factors))
    if axis == 1:
        factors = factors[:, None]
    return matrix * factors


def check_input(X):
    """Input contract of fit_transform: >=2-D, finite, non-negative.

    Follows reference sklearn_utils.py:59-97 and nmf.py:23-26,193-194.
    """
    X = np.atleast_2d(np.asarray(X))
    if X.dtype.kind == 'f' and not np.isfinite(X.sum()) \
            and not np.isfinite(X).all():
        raise ValueError("array contains NaN or infinity")
    if (X < 0).any():
        raise ValueError("Negative values in data passed to NMF.fit")
    return X


# ------------------------------------------------------- single-step rules ---

def ratio_q(X, W, H, eps=EPS_RATIO):
    """Q = (X + eps) / (W.H + eps), dense.  Reference nmf.py:325-336."""
    return (X + eps) / (W.dot(H) + eps)


def kl_error(X, W, H):
    """Loss as the reference evaluates it (its own W.H product, eps=1e-8).

    Reference nmf.py:297-310 -> metrics.py:18-20.
    """
    return generalized_kl(X, W.dot(H))


def updated_w(X, W, H, Q=None):
    """W * (Q.H^T) -- no denominator.  Reference nmf.py:338-343."""
    if Q is None:
        Q = ratio_q(X, W, H)
    return W * Q.dot(H.T)


def updated_h(X, W, H, Q=None):
    """normalize_rows(H * (W^T.Q)).  Reference nmf.py:345-351."""
    if Q is None:
        Q = ratio_q(X, W, H)
    return normalize_sum(H * W.T.dot(Q), axis=1)


def update_step(X, W, H, fit=True, scale_W=False):
    """One multiplicative update; returns (W_new, H_new).

    Reference nmf.py:232-257.  Q is computed once from the *old* W; the H rule
    then pairs that old Q with the *new* W (quirk q2).
    """
    if scale_W:  # reference nmf.py:246-250 (dead from every caller, kept)
        W = scale_matrix(normalize_sum(W, axis=1), X.sum(axis=1), axis=1)
    Q = ratio_q(X, W, H)
    W_new = updated_w(X, W, H, Q=Q)
    H_new = updated_h(X, W_new, H, Q=Q) if fit else H
    return W_new, H_new


def fit_iteration_lean32(X32, W, H, eps=EPS_RATIO):
    """The "optimised CPU" baseline of BASELINE.md section 3: the same iteration (loss before the update, old
    ratio with the new W in the H rule, row normalisation: nmf.py:212-222, 232-257) without the reference's
    avoidable work -- ONE W.H product shared by the loss and the ratio, float32 arrays, temporaries reused.
    Returns (loss, W_new, H_new).  bench.py times it next to the faithful restatement so that the GPU figure is
    not only compared with the reference's waste; tests check it against `update_step` to float32 accuracy."""
    WH = W.dot(H)                                  # float32 [n, f]
    sum_wh = WH.sum(dtype=np.float64)
    WH += np.float32(eps)
    Q = X32 + np.float32(eps)
    Q /= WH                                        # the ratio; WH is scratch from here on
    np.log(Q, out=WH)
    WH *= X32
    loss = WH.sum(dtype=np.float64) - X32.sum(dtype=np.float64) + sum_wh
    W_new = W * Q.dot(H.T)
    H_new = H * W_new.T.dot(Q)
    H_new /= (np.float32(EPS_NORMALIZE) + H_new.sum(axis=1, keepdims=True))
    return loss, W_new, H_new


# ------------------------------------------------------------ the hot loop ---

def init_factors(X, k, H0=None, rng=None):
    """(W0, H0): H0 given or drawn, W0 = X.H0^T.  Reference nmf.py:147-157."""
    n, f = X.shape
    if H0 is None:
        draw = np.random.random if rng is None else rng.random_sample
        H0 = normalize_sum(np.abs(draw((k, f))) + .01, axis=1)
    else:
        assert H0.shape == (k, f)
    return X.dot(H0.T), H0


def fit_transform(X, k=None, H0=None, max_iter=200, tol=1e-6, fit=True,
                  components=None, rng=None, warn=True):
    """The loop of reference nmf.py:159-230 as a pure function.

    Returns (W, H, errors).  `errors` has one entry per *executed* update, each
    recorded before that update (q5).  With fit=False the dictionary
    `components` is held fixed (this is what `transform` does after setting
    `_init_dictionary = components_`, reference nmf.py:275-291).
    """
    X = check_input(X)
    n, f = X.shape
    if not k:
        k = f
    if not fit:
        H0 = components if H0 is None else H0
    W, H = init_factors(X, k, H0=H0, rng=rng)
    if not fit:
        H = components
    prev = np.inf
    tol_abs = tol * n * f
    errors = []
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        err = kl_error(X, W, H)
        if prev - err < tol_abs:
            break
        prev = err
        errors.append(err)
        W, H = update_step(X, W, H, fit=fit)
    if warn and n_iter == max_iter and tol_abs > 0:
        sys.stderr.write("Warning: Iteration limit reached during fit\n")
    return W, H, errors


def transform(X, components, max_iter=200, tol=1e-6):
    """Coefficients for a fixed dictionary.  Reference nmf.py:275-291."""
    W, _, errors = fit_transform(X, k=components.shape[0], max_iter=max_iter,
                                 tol=tol, fit=False, components=components)
    return W, errors


# ------------------------------------------------------------ learner side ---

def stack_modalities(blocks, coefs):
    """hstack of c*m.  Reference learner.py:53-56, array_utils.py:5-9."""
    return np.hstack([c * np.asarray(m) for m, c in zip(blocks, coefs)])


def axis_range(dims, idx):
    """Column range of modality idx.  Reference learner.py:58-62."""
    start = sum(dims[:idx])
    return start, start + dims[idx]


def learner_train(blocks, coefs, k, iterations, H0):
    """MultimodalLearner.train: tol=0 fit of the stacked matrix.

    Reference learner.py:31-41.  Returns (dico, W).
    """
    V = stack_modalities(blocks, coefs)
    W, H, _ = fit_transform(V, k=k, H0=H0, max_iter=iterations, tol=0)
    return H, W


def learner_internal(blocks, coefs, dico_slices, iterations):
    """reconstruct_internal_multi: transform against column-sliced dictionary.

    Reference learner.py:67-78 + fit_coefficients learner.py:11-15.
    """
    V = stack_modalities(blocks, coefs)
    D = np.hstack(dico_slices)
    W, _ = transform(V, D, max_iter=iterations, tol=0)
    return W


# ---------------------------------------------------- synthetic workload ---

def synthetic_block(base_seed, b, rows, f, k_true, Ht=None):
    """Row block b of the seeded factorisable+noise V of SURVEY.md section 8d."""
    if Ht is None:
        Ht = synthetic_Ht(base_seed, f, k_true)
    rs = np.random.RandomState(base_seed + 1 + b)
    Wt = rs.gamma(1.0, 1.0, (rows, k_true))
    return Wt.dot(Ht) / k_true + 0.05 * rs.random_sample((rows, f))


def synthetic_Ht(base_seed, f, k_true):
    return np.random.RandomState(base_seed).gamma(0.5, 1.0, (k_true, f))


def synthetic_V(base_seed, n, f, k_true, block=8192):
    Ht = synthetic_Ht(base_seed, f, k_true)
    out = np.empty((n, f))
    for b, r0 in enumerate(range(0, n, block)):
        r1 = min(n, r0 + block)
        out[r0:r1] = synthetic_block(base_seed, b, r1 - r0, f, k_true, Ht)
    return out


def synthetic_H0(base_seed, f, k):
    return normalize_sum(
        np.random.RandomState(base_seed - 1).random_sample((k, f)) + .01,
        axis=1)


# ---- the reference's CSR branch (nmf.py:52-70, 301-308, 331-334) ----------------------------------
# With sparse X the reference evaluates W.H and the ratio ONLY on the stored entries of X: Q is a sparse
# matrix with X's structure (structural zeros of X stay zero in Q, unlike the dense branch where they
# become eps/(WH+eps)), and the loss uses sum(W.H) = sum_a colsum(W)_a * rowsum(H)_a.

def sparse_structure(X):
    """(ii, jj, data) of the stored non-zeros in CSR order, after eliminate_zeros (nmf.py:66-67)."""
    import scipy.sparse as sp
    X = sp.csr_matrix(X, copy=True)
    X.eliminate_zeros()
    X.sort_indices()
    ii, jj = X.nonzero()
    return X, ii, jj


def sparse_wh(X, W, H):
    """W.H on the stored entries of X.  Reference nmf.py:52-70 (_special_sparse_dot)."""
    X, ii, jj = sparse_structure(X)
    return X, ii, jj, np.multiply(W[ii, :], H.T[jj, :]).sum(axis=1)


def sparse_ratio_q(X, W, H, eps=EPS_RATIO):
    """Q values (CSR order of X) = (x + eps) / (wh + eps).  Reference nmf.py:331-334."""
    Xs, ii, jj, wh = sparse_wh(X, W, H)
    return Xs, ii, jj, (Xs.data + eps) / (wh + eps)


def sparse_kl_error(X, W, H, eps=EPS_RATIO):
    """Reference nmf.py:301-308."""
    Xs, ii, jj, wh = sparse_wh(X, W, H)
    wh_sum = np.sum(np.multiply(np.sum(W, axis=0), np.sum(H, axis=1)))
    return (np.multiply(Xs.data, np.log(np.divide(Xs.data + eps, wh + eps)))).sum() - Xs.data.sum() + wh_sum


def sparse_update_step(X, W, H, fit=True):
    """One _update with sparse X.  Reference nmf.py:232-257 with the CSR `_Q` (old ratio for both rules)."""
    import scipy.sparse as sp
    Xs, ii, jj, q = sparse_ratio_q(X, W, H)
    Q = sp.csr_matrix((q, (ii, jj)), shape=Xs.shape)
    W_new = np.multiply(W, np.asarray(Q.dot(H.T)))
    H_new = H
    if fit:
        H_new = normalize_sum(np.multiply(H, np.asarray(Q.T.dot(W_new)).T), axis=1)
    return W_new, H_new


def sparse_fit_transform(X, k, H0, max_iter=200, tol=1e-6, fit=True, components=None):
    """The loop of nmf.py:159-230 for CSR input.  Returns (W, H, errors)."""
    import scipy.sparse as sp
    Xs = sp.csr_matrix(X)
    n, f = Xs.shape
    H = np.array(H0, dtype=np.float64)
    W = np.asarray(Xs.dot(H.T))                               # nmf.py:156
    if not fit:
        H = np.array(components, dtype=np.float64)
    prev = np.inf
    tol_abs = tol * n * f
    errors = []
    for _ in range(max_iter):
        err = sparse_kl_error(Xs, W, H)
        if prev - err < tol_abs:
            break
        prev = err
        errors.append(err)
        W, H = sparse_update_step(Xs, W, H, fit=fit)
    return W, H, errors


# ---- measures of the nearest-neighbour evaluation (metrics.py:58-86, evaluation.py:103-106) ---------
def pairwise_distances(A, B, name, eps=EPS_RATIO):
    """[len(A), len(B)] matrix of the reference's measure `name` through its own broadcast pattern."""
    a = np.asarray(A, dtype=np.float64)[:, np.newaxis, :]
    b = np.asarray(B, dtype=np.float64)[np.newaxis, :, :]
    gkl = lambda x, y: (np.multiply(x, np.log(np.divide(x + eps, y + eps))) - x + y).sum(axis=-1)
    if name == 'kl_div':
        return gkl(a, b)
    if name == 'rev_kl_div':
        return gkl(b, a)
    if name == 'sym_kl_div':
        return .5 * (gkl(a, b) + gkl(b, a))
    if name == 'frobenius':
        return np.sqrt(np.square(a - b).sum(axis=-1))
    if name == 'cosine_diff':
        ab = np.multiply(a, b).sum(axis=-1)
        return -(ab / (np.sqrt(np.square(a).sum(axis=-1) * np.square(b).sum(axis=-1)) + (ab == 0)))
    raise ValueError(name)
