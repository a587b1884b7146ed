#!/usr/bin/env python3
"""Quick parity check of the k = 500 bf16 path against the f32 exact mode on the same data (experiment builds:
set KLNMF_LIB).  Prints the per-iteration relative loss difference and the relative difference of W and H."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from multimodal_amd import _native

n, f, k, iters = 640, 1024, 500, 6
rs = np.random.RandomState(5)
Ht = rs.gamma(0.5, 1.0, (k, f)); Wt = rs.gamma(1.0, 1.0, (n, k))
X = Wt @ Ht / k + 0.05 * rs.random_sample((n, f))
H0 = rs.random_sample((k, f)) + .01; H0 /= H0.sum(axis=1, keepdims=True)
res = {}
for mode in ('f32', 'bf16'):
    with _native.Context(mode, device=0) as ctx:
        ctx.set_problem(n, f, k, iters)
        ctx.upload_blocks([X]); ctx.set_H(H0); ctx.init_W()
        errs, n_done, stopped = ctx.run(iters, True, -1e300)
        res[mode] = (np.array(errs), ctx.get_W(), ctx.get_H())
e0, W0, Hh0 = res['f32']; e1, W1, Hh1 = res['bf16']
print('loss rel diff per iter', np.abs(e1 - e0) / np.abs(e0))
print('W rel', np.linalg.norm(W1 - W0) / np.linalg.norm(W0), 'H rel', np.linalg.norm(Hh1 - Hh0) / np.linalg.norm(Hh0))
ok = np.all(np.abs(e1 - e0) / np.abs(e0) < 2e-3) and np.linalg.norm(W1 - W0) / np.linalg.norm(W0) < 2e-2
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
