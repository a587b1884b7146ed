"""Array helpers of the path (reference multimodal/lib/array_utils.py:5-22).

`normalize_sum` here is the host helper used to build the *initial*
dictionary (nmf.py:149-151) and by the dead `scale_W` branch; the per-iteration
row normalisation of H (nmf.py:350) runs on the GPU inside the H-update kernel
(csrc: k_update_H / k_update_pack_H).
"""
import numpy as np
import scipy.sparse as sp


def safe_hstack(blocks):
    """hstack that keeps sparsity if any block is sparse (array_utils.py:5-9)."""
    if any(sp.issparse(b) for b in blocks):
        return sp.hstack(blocks)
    return np.hstack(blocks)


def safe_vstack(Xs):
    if any(sp.issparse(X) for X in Xs):
        return sp.vstack(Xs)
    return np.vstack(Xs)


def normalize_sum(a, axis=0, eps=1.e-16):
    """a / (eps + sum over axis), ValueError if the axis does not exist
    (array_utils.py:19-22)."""
    if axis >= len(a.shape):
        raise ValueError
    return a / (eps + np.expand_dims(np.sum(a, axis=axis), axis))
