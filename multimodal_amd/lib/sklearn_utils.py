"""Input contract of the NMF entry points (host-side validation).

Mirrors the behaviour of reference multimodal/lib/sklearn_utils.py:59-110: the
ValueErrors are raised in Python before anything is uploaded to the GPU.
"""
import numpy as np
from scipy import sparse


def assert_all_finite(X):
    """ValueError("array contains NaN or infinity") for non-finite float data
    (reference sklearn_utils.py:59-69: cheap sum test first, full test only if
    the sum is not finite)."""
    data = X.data if sparse.issparse(X) else X
    if data.dtype.kind == 'f' and not np.isfinite(data.sum()) \
            and not np.isfinite(data).all():
        raise ValueError("array contains NaN or infinity")


def array2d(X, dtype=None, order=None, copy=False):
    """At-least-2-D ndarray view of dense input (sklearn_utils.py:72-80)."""
    if sparse.issparse(X):
        raise TypeError('A sparse matrix was passed, but dense data '
                        'is required. Use X.todense() to convert to dense.')
    X_2d = np.asarray(np.atleast_2d(X), dtype=dtype, order=order)
    if X is X_2d and copy:
        X_2d = np.copy(X_2d, order='K')
    return X_2d


def atleast2d_or_csr(X, dtype=None, order=None, copy=False):
    """>=2-D ndarray (np.matrix -> ndarray) or CSR; finite-checked
    (sklearn_utils.py:83-97)."""
    if sparse.issparse(X):
        if dtype is None or X.dtype == dtype:
            X = X.tocsr()
        else:
            X = sparse.csr_matrix(X, dtype=dtype)
    else:
        X = array2d(X, dtype=dtype, order=order, copy=copy)
    assert_all_finite(X)
    return X


def safe_sparse_dot(a, b, dense_output=False):
    """Dot product that also accepts scipy sparse operands
    (sklearn_utils.py:102-110).  Host helper, not on the GPU path."""
    if sparse.issparse(a) or sparse.issparse(b):
        ret = a * b
        if dense_output and hasattr(ret, "toarray"):
            ret = ret.toarray()
        return ret
    return np.dot(a, b)
