#!/usr/bin/env python3
"""How the recorded loss of a fit on fp8 ratio tiles drifts from the oracle's over many iterations, against the row count:
the e4m3 rounding of a nearly converged fit's ratios is FROZEN from one iteration to the next (the ratios barely move), so what
averages out over the rows within an iteration (0.036 sqrt(2 / rows)) does not average out over the iterations -- the slow modes
of the multiplicative update integrate it.  One fit per (rows, tiles), loss deviation from the oracle along the iterations.
    python scripts/fp8_drift_probe.py [f] [k]   (GPU box)"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
os.environ['KLNMF_DEV'] = '1'
from oracle import klnmf_oracle as orc              # noqa: E402
import monitor_calibration as mc                    # noqa: E402

f = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
iters = 200
marks = (10, 20, 30, 50, 75, 100, 150, 199)
OFF = {'KLNMF_MON_THRESHOLD': '1', 'KLNMF_MON_MIN_SPREAD': '0'}
for n in (40000, 160000):
    X = orc.synthetic_V(1234, n, f, k)
    H0 = orc.synthetic_H0(11, f, k)
    Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0, warn=False)
    eo = np.array(eo)
    for tag, env in (('16-bit tiles', {'KLNMF_QTILE': '16'}), ('fp8 tiles', OFF), ('fp8 tiles, f16 W', dict(OFF, KLNMF_COL8='0'))):
        for v in ('KLNMF_MON_THRESHOLD', 'KLNMF_MON_MIN_SPREAD', 'KLNMF_COL8'):
            os.environ.pop(v, None)
        W, H, e, rep = mc.fit(X, H0, k, iters, env)
        for v in ('KLNMF_MON_THRESHOLD', 'KLNMF_MON_MIN_SPREAD', 'KLNMF_COL8'):
            os.environ.pop(v, None)
        e = np.array(e)
        m = min(len(e), len(eo))
        dev = (e[:m] - eo[:m]) / eo[:m]
        fin = (orc.kl_error(X, W.astype(np.float64), H.astype(np.float64)) - orc.kl_error(X, Wo, Ho)) / orc.kl_error(X, Wo, Ho)
        print('%7d x %d, k = %d  %-18s stat %.2e col8 %3d | signed loss deviation at it %s: %s | final (exact evaluation) %+.2e' % (
            n, f, k, tag, rep['monitor_statistic'], rep['column_pass_iterations'], '/'.join(str(i) for i in marks),
            ' '.join('%+.1e' % dev[min(i, m - 1)] for i in marks), fin), flush=True)
