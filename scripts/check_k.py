#!/usr/bin/env python3
"""Step-by-step parity of the bf16 mode against the f64 mode at a given small shape (experiment builds: KLNMF_LIB):
W0 = V.H0^T (row pass, INIT mode), the loss (LOSS mode), and fits of 1..iters iterations (UPDATE mode + column pass).
    python scripts/check_k.py N F K [ITERS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from multimodal_amd import _native

n, f, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rs = np.random.RandomState(5)
Ht = rs.gamma(0.5, 1.0, (k, f)); Wt = rs.gamma(1.0, 1.0, (n, k))
X = Wt @ Ht / k + 0.05 * rs.random_sample((n, f))
H0 = rs.random_sample((k, f)) + .01; H0 /= H0.sum(axis=1, keepdims=True)
out = {}
for mode in ('f64', 'bf16'):
    r = {}
    with _native.Context(mode, device=0) as ctx:
        ctx.set_problem(n, f, k, iters)
        ctx.upload_blocks([X]); ctx.set_H(H0); ctx.init_W()
        r['W0'] = ctx.get_W()
        r['err0'] = ctx.error()
        errs, n_done, stopped = ctx.run(iters, True, -1e300)
        r['errs'] = np.array(errs); r['W'] = ctx.get_W(); r['H'] = ctx.get_H()
    out[mode] = r
a, b = out['f64'], out['bf16']
rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
print('W0 rel %.2e | err0 rel %.2e | errs rel %s | W rel %.2e | H rel %.2e | H min %.2e' % (
    rel(b['W0'], a['W0']), abs(b['err0'] - a['err0']) / abs(a['err0']),
    np.array2string(np.abs(b['errs'] - a['errs']) / np.abs(a['errs']), precision=1), rel(b['W'], a['W']), rel(b['H'], a['H']), b['H'].min()))
