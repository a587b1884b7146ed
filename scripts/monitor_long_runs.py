#!/usr/bin/env python3
"""How the monitor's two measurements behave over LONG healthy runs (does the ratios' spread of well-fitted dense data ever come
near the dead-zone threshold?) and where the constant-columns class ends for several thresholds.
    python scripts/monitor_long_runs.py  (GPU box)"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
os.environ['KLNMF_DEV'] = '1'
from oracle import klnmf_oracle as orc              # noqa: E402
import monitor_calibration as mc                    # noqa: E402


def report(name, X, H0, k, iters, envs, with_oracle=True):
    fo = None
    if with_oracle:
        Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0, warn=False)
        fo = orc.kl_error(X, Wo, Ho)
    for tag, env in envs:
        for v in ('KLNMF_MON_THRESHOLD', 'KLNMF_MON_MIN_SPREAD', 'KLNMF_COL8'):
            os.environ.pop(v, None)
        W, H, e, rep = mc.fit(X, H0, k, iters, env)
        for v in ('KLNMF_MON_THRESHOLD', 'KLNMF_MON_MIN_SPREAD', 'KLNMF_COL8'):
            os.environ.pop(v, None)
        rel = abs(orc.kl_error(X, W.astype(np.float64), H.astype(np.float64)) - fo) / fo if (fo is not None and len(e) == iters) else float('nan')
        print('%-58s %-34s KL rel %9.2e len %3d fp8 %3d stat %.2e spread %.4f trips %3d gave up %s' % (
            name, tag, rel, len(e), rep['tile_iterations'], rep['monitor_statistic'], rep['monitor_min_spread'], rep['monitor_trips'],
            rep['gave_up']), flush=True)


OFF = {'KLNMF_MON_THRESHOLD': '1', 'KLNMF_MON_MIN_SPREAD': '0'}       # measure, never trip
report('dense 40 000 x 512, k = 50 (C2 kind), 200 it', orc.synthetic_V(1234, 40000, 512, 50), orc.synthetic_H0(11, 512, 50), 50, 200,
       [('never trip', OFF)])
report('dense 70 000 x 256, k = 200 (C4 kind), 200 it', orc.synthetic_V(1234, 70000, 256, 200), orc.synthetic_H0(11, 256, 200), 200, 200,
       [('never trip', OFF)])
report('dense 66 000 x 384, k = 300 (C5 kind), 100 it', orc.synthetic_V(1234, 66000, 384, 300), orc.synthetic_H0(11, 384, 300), 300, 100,
       [('never trip', OFF)])
X = mc.low_rank(1, 66000, 300, 130)
X[:, ::7] = 3.0
H0 = orc.synthetic_H0(11, 300, 130)
report('constant columns 66 000 x 300, k = 130, 100 it', X, H0, 130, 100,
       [('spread 0.03', {'KLNMF_MON_MIN_SPREAD': '0.03'}), ('spread 0.04', {'KLNMF_MON_MIN_SPREAD': '0.04'}),
        ('spread 0.05', {'KLNMF_MON_MIN_SPREAD': '0.05'}),
        ('spread 0.04, f16 W operand', {'KLNMF_MON_MIN_SPREAD': '0.04', 'KLNMF_COL8': '0'}),
        ('spread 0.03, f16 W operand', {'KLNMF_MON_MIN_SPREAD': '0.03', 'KLNMF_COL8': '0'}),
        ('threshold 5e-4', {'KLNMF_MON_THRESHOLD': '5e-4'})])
X = mc.low_rank(1, 40000, 500, 100)
X[:, ::7] = 3.0
report('constant columns 40 000 x 500, k = 100, 100 it', X, orc.synthetic_H0(11, 500, 100), 100, 100,
       [('default', {}), ('spread 0.04', {'KLNMF_MON_MIN_SPREAD': '0.04'}), ('never trip', OFF)])
