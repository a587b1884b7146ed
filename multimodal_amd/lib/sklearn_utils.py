"""Input contract of the NMF entry points: what the host accepts before anything is uploaded.

Behaviour (not code) of reference multimodal/lib/sklearn_utils.py:59-110, which the NMF front end relies on:
dense input becomes an ndarray of at least two dimensions (np.matrix and nested lists included), sparse input
becomes CSR, non-finite floats raise ValueError("array contains NaN or infinity") -- the message fixture G8 pins.
"""
import numpy as np
from scipy import sparse

_NONFINITE = "array contains NaN or infinity"


def assert_all_finite(X):
    """Raise ValueError for NaN / inf in float data (sklearn_utils.py:59-69).  One reduction decides the common
    case: a finite sum proves every term finite (inf - inf would be NaN, overflow inf), so the element-wise test
    only runs when the sum itself is not finite."""
    values = X.data if sparse.issparse(X) else np.asarray(X)
    if values.dtype.kind != 'f':
        return
    if np.isfinite(values.sum()):
        return
    if not np.all(np.isfinite(values)):
        raise ValueError(_NONFINITE)


def array2d(X, dtype=None, order=None, copy=False):
    """Dense input as an ndarray with ndim >= 2 (sklearn_utils.py:72-80); refuses scipy sparse matrices."""
    if sparse.issparse(X):
        raise TypeError('A sparse matrix was passed, but dense data is required. Use X.todense() to convert to dense.')
    out = np.asarray(np.atleast_2d(X), dtype=dtype, order=order)
    return out.copy(order='K') if (copy and out is X) else out


def atleast2d_or_csr(X, dtype=None, order=None, copy=False):
    """The validated form of a data matrix (sklearn_utils.py:83-97): CSR for sparse input (converted to `dtype` when
    one is asked for and differs), `array2d` otherwise; finite-checked either way."""
    if not sparse.issparse(X):
        checked = array2d(X, dtype=dtype, order=order, copy=copy)
    elif dtype is not None and X.dtype != dtype:
        checked = sparse.csr_matrix(X, dtype=dtype)
    else:
        checked = X.tocsr()
    assert_all_finite(checked)
    return checked


def safe_sparse_dot(a, b, dense_output=False):
    """a . b for any mix of ndarray and scipy sparse operands (sklearn_utils.py:102-110); host helper of the CPU-side
    API mirror, never on the GPU path."""
    if not (sparse.issparse(a) or sparse.issparse(b)):
        return np.dot(a, b)
    product = a * b
    if dense_output and sparse.issparse(product):
        product = product.toarray()
    return product
