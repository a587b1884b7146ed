import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import klnmf_oracle as orc
from multimodal_amd import _native
n, f, k, iters = 70000, 256, 40, 8
rs = np.random.RandomState(5)
X = orc.synthetic_V(13, n, f, 12)
X[:, f // 2:] = 1e-4 * rs.random_sample((n, f - f // 2))
for (i, j) in [(100, f // 2 + 3), (7000, f - 1)]:
    X[i, j] = 100.0 * X.mean()
H0 = orc.synthetic_H0(13, f, k)
with _native.Context('f16', device=0) as ctx:
    ctx.set_problem(n, f, k, iters)
    print('tile bytes', ctx.query(_native.Q_RATIO_TILE_BYTES))
    ctx.upload_blocks([X])
    print('sumV', ctx.sum_V(), X.sum(), 'max', X.max(), 'mean', X.mean())
    ctx.set_H(H0); ctx.init_W()
    e, nd, st = ctx.run(iters, True, 0.0)
    print(nd, st, ctx.fp8_report())
