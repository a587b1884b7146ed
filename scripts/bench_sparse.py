"""Timing of the CSR branch (next-row N3) on an Acorns-shaped problem: n x f very sparse, k components.
Prints ms per fit iteration on the GPU (f64 / f32) and for the numpy/scipy oracle on a row sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from multimodal_amd import _native
from oracle import klnmf_oracle as orc

n, f, k, dens, iters = 20000, 110000, 50, 0.005, 10
rs = np.random.RandomState(0)
X = sp.random(n, f, density=dens, format='csr', random_state=rs, data_rvs=lambda s: rs.gamma(1.0, 1.0, s))
H0 = orc.synthetic_H0(3, f, k)
print('X %dx%d nnz %d (%.2f%%), k=%d' % (n, f, X.nnz, 100.0 * X.nnz / (n * f), k))
for prec in ('f64', 'f32'):
    with _native.Context(prec) as c:
        c.set_problem_sparse(X.astype(np.float32) if prec == 'f32' else X, k, iters + 2)
        c.set_H(H0); c.init_W()
        c.run(2, True, 0.0)
        c.set_H(H0); c.init_W()
        t0 = time.perf_counter()
        errs, nd, st = c.run(iters, True, 0.0)
        dt = time.perf_counter() - t0
        # bytes per iteration (algorithmic): SDDMM reads nnz*(k of W amortised per row + k of H^T) ...; report the simple count
        b = X.nnz * (8 + 8 + 2 * 8) + 2 * n * k * 8 + 3 * k * f * 8
        print('%s: %.2f ms / iteration   loss %.6e -> %.6e   (~%.1f GB/s of the minimal bytes)' % (
            prec, 1e3 * dt / iters, errs[0], errs[-1], b / (dt / iters) / 1e9))
rows = 2000
Xs = X[:rows]
t0 = time.perf_counter()
orc.sparse_fit_transform(Xs, k, H0, max_iter=3, tol=0)
dt = (time.perf_counter() - t0) / 3
print('oracle (scipy, fp64) on %d rows: %.1f ms / iteration -> %.1f ms scaled to %d rows' % (rows, 1e3 * dt, 1e3 * dt * n / rows, n))
