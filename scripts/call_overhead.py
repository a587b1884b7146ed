#!/usr/bin/env python3
"""What one call of the reference's API costs at its own data scale, beyond the iterations themselves: wall time of
fit_coefficients(data, dictionary, 50) (learner.py:11-15: what experiment.py:177-180 calls 2M..9 times per run) against
the time of the 50 iterations alone (klnmf_run on a live context).  KLNMF_NO_POOL=1 shows the cost without pooled
contexts.    python scripts/call_overhead.py [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from multimodal_amd import _native
from multimodal_amd.learner import fit_coefficients

prec = sys.argv[1] if len(sys.argv) > 1 else 'f64'
os.environ['KLNMF_PRECISION'] = prec
for (n, f, k) in [(200, 450, 10), (1000, 2000, 50)]:
    rs = np.random.RandomState(3)
    X = rs.random_sample((n, f)) + 0.01
    D = rs.random_sample((k, f)) + .01; D /= D.sum(axis=1, keepdims=True)
    fit_coefficients(X, D, 50)                                   # first call: library load, allocations
    t0 = time.perf_counter()
    for _ in range(20):
        fit_coefficients(X, D, 50)
    call_ms = 1e3 * (time.perf_counter() - t0) / 20
    with _native.Context(prec, device=0) as ctx:
        ctx.set_problem(n, f, k, 50)
        ctx.upload_blocks([X]); ctx.set_H(D); ctx.init_W()
        ctx.run(50, False, 0.0)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.run(50, False, 0.0)
        loop_ms = 1e3 * (time.perf_counter() - t0) / 20
    print('%5d x %4d k=%2d %4s %s: fit_coefficients(50 iterations) %.2f ms per call, the 50 iterations alone %.2f ms -> %.2f ms per call around them'
          % (n, f, k, prec, 'no pool' if os.environ.get('KLNMF_NO_POOL') == '1' else 'pooled ', call_ms, loop_ms, call_ms - loop_ms), flush=True)
