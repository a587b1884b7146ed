#!/usr/bin/env python3
"""Per-iteration time of klnmf_run at the reference's real-data scale (10^2..10^3 rows), where the
loop is bound by kernel launches rather than by the kernels:  python scripts/small_problem_timing.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from multimodal_amd import _native

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for (n, f, k) in [(200, 450, 10), (1000, 2000, 50), (2000, 4096, 200), (10000, 4096, 50)]:
    rs = np.random.RandomState(3)
    X = rs.random_sample((n, f)) + 0.01
    H0 = rs.random_sample((k, f)) + .01; H0 /= H0.sum(axis=1, keepdims=True)
    for mode in ('bf16', 'f32', 'f64'):
        with _native.Context(mode, device=0) as ctx:
            ctx.set_problem(n, f, k, iters)
            ctx.upload_blocks([X]); ctx.set_H(H0); ctx.init_W()
            ctx.run(10, True, -1e300)                      # warm-up
            t0 = time.perf_counter()
            errs, n_done, stopped = ctx.run(iters, True, -1e300)
            dt = time.perf_counter() - t0
        print('%6d x %5d k=%3d %5s: %7.1f us / iteration  (%d iterations, last loss %.6e)' % (n, f, k, mode, 1e6 * dt / iters, n_done, errs[-1]), flush=True)
