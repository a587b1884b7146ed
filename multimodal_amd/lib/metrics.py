"""generalized_KL -- the loss of the path (reference multimodal/lib/metrics.py:18-20).

Inside the NMF loop the loss is fused into the W.H kernels; this standalone
entry point (used by tests and by evaluation code) runs the element-wise
reduction kernel `k_gkl` on the GPU.  `axis` reductions are done slice by slice
with the same kernel.
"""
import os

import numpy as np

from .. import _native

EPSILON = 1.e-8


def _ctx():
    return _native.Context(precision='f64', device=int(os.environ.get('KLNMF_DEVICE', '0')))


def generalized_KL(x, y, eps=EPSILON, axis=None):
    """sum(x * log((x + eps) / (y + eps)) - x + y) over `axis` (None = all)."""
    x = np.asarray(x)
    y = np.asarray(y)
    x, y = np.broadcast_arrays(x, y)
    if x.dtype not in (np.float32, np.float64) or x.dtype != y.dtype:
        x = x.astype(np.float64)
        y = y.astype(np.float64)
    with _ctx() as ctx:
        if axis is None:
            return ctx.generalized_kl(x, y, eps)
        xs = np.moveaxis(x, axis, -1)
        ys = np.moveaxis(y, axis, -1)
        out_shape = xs.shape[:-1]
        xs = np.ascontiguousarray(xs).reshape(-1, xs.shape[-1])
        ys = np.ascontiguousarray(ys).reshape(-1, ys.shape[-1])
        out = np.empty(xs.shape[0], dtype=np.float64)
        for i in range(xs.shape[0]):
            out[i] = ctx.generalized_kl(xs[i], ys[i], eps)
        return out.reshape(out_shape)


# ---- measures of the nearest-neighbour evaluation (reference metrics.py:58-86) ------------------------
# The reference calls them on broadcast operands [n_a, 1, d] x [1, n_b, d] with axis=-1
# (evaluation.py:103-106); that pairwise pattern -- and plain [n, d] x [n, d] rows -- is what runs on the
# GPU (`klnmf_all_distances`).  Other broadcast patterns raise: there is no CPU path behind this module.

def _pairwise(a, b, axis, metric):
    a = np.asarray(a)
    b = np.asarray(b)
    if axis not in (-1, a.ndim - 1) or a.ndim != b.ndim:
        raise ValueError('the GPU measures reduce over the last axis of operands of equal rank')
    dev = int(os.environ.get('KLNMF_DEVICE', '0'))
    if a.ndim == 3 and a.shape[1] == 1 and b.shape[0] == 1 and a.shape[2] == b.shape[2]:
        return _native.all_distances(a[:, 0, :], b[0, :, :], metric, dev)
    if a.ndim == 2 and a.shape == b.shape:           # row i against row i
        out = np.empty(a.shape[0], dtype=np.float32 if (a.dtype == np.float32 and b.dtype == np.float32) else np.float64)
        for r0 in range(0, a.shape[0], 1024):        # diagonal of bounded square blocks
            blk = _native.all_distances(a[r0:r0 + 1024], b[r0:r0 + 1024], metric, dev)
            out[r0:r0 + 1024] = np.diagonal(blk)
        return out
    if a.ndim == 1 and a.shape == b.shape:
        return _native.all_distances(a[None, :], b[None, :], metric, dev)[0, 0]
    raise ValueError('unsupported operand shapes %s, %s for a GPU measure' % (a.shape, b.shape))


def kl_div(a, b, axis=-1, eps=EPSILON, normalize=False):
    """generalized_KL(a, b) along the last axis (reference metrics.py:58-62; as there, `eps` is ignored)."""
    if normalize:                                    # in place, as the reference does
        a /= np.expand_dims(a.sum(axis=axis), axis)
        b /= np.expand_dims(b.sum(axis=axis), axis)
    return _pairwise(a, b, axis, _native.DIST_KL)


def rev_kl_div(a, b, **kwargs):
    """kl_div(b, a) (reference metrics.py:65-66)."""
    return kl_div(b, a, **kwargs) if kwargs.get('normalize') else _pairwise(a, b, kwargs.get('axis', -1), _native.DIST_REV_KL)


def sym_kl_div(a, b, **kwargs):
    """0.5 * (kl_div + rev_kl_div) (reference metrics.py:69-70)."""
    if kwargs.get('normalize'):
        return .5 * (kl_div(a, b, **kwargs) + rev_kl_div(a, b, **kwargs))
    return _pairwise(a, b, kwargs.get('axis', -1), _native.DIST_SYM_KL)


def frobenius(a, b, axis=-1):
    """sqrt(sum((a - b)^2)) (reference metrics.py:73-74)."""
    return _pairwise(a, b, axis, _native.DIST_FROBENIUS)


def cosine_diff(a, b, axis=-1):
    """-cosine_similarity, 0 when either vector is 0 (reference metrics.py:77-86)."""
    return _pairwise(a, b, axis, _native.DIST_COSINE_DIFF)


def cosine_similarity(a, b, axis=-1):
    return -cosine_diff(a, b, axis=axis)
