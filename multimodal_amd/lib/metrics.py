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
