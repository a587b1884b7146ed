#!/usr/bin/env python3
"""Which e4m3 operand hurts the constant-columns class (66 000 x 300, k = 130, every 7th column constant)?  The same fit with fp8
ratio tiles + f16 W operand (KLNMF_COL8=0) and with fp8 on both sides, monitor off, against the oracle along the iterations."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
os.environ['KLNMF_DEV'] = '1'
from oracle import klnmf_oracle as orc              # noqa: E402
import monitor_calibration as mc                    # noqa: E402

n, f, k, iters = 66000, 300, 130, int(sys.argv[1]) if len(sys.argv) > 1 else 100
X = mc.low_rank(1, n, f, k)
X[:, ::7] = 3.0
H0 = orc.synthetic_H0(11, f, k)
Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0, warn=False)
eo = np.array(eo)
for name, env in (('16-bit tiles', {'KLNMF_QTILE': '16'}),
                  ('fp8 tiles, f16 W operand', {'KLNMF_Q8_MONITOR': '0', 'KLNMF_COL8': '0'}),
                  ('fp8 tiles, e4m3 W image', {'KLNMF_Q8_MONITOR': '0'}),
                  ('monitor', {})):
    for v in ('KLNMF_COL8',):
        os.environ.pop(v, None)
    W, H, e, rep = mc.fit(X, H0, k, iters, env)
    os.environ.pop('KLNMF_COL8', None)
    e = np.array(e)
    m = min(len(e), len(eo))
    dev = np.abs(e[:m] - eo[:m]) / eo[:m]
    print('%-28s len %3d  fp8 its %3d col8 %3d  stat %.2e spread %.4f trips %d | loss deviation at it 5/10/20/40/60/80/last: %s' % (
        name, len(e), rep['tile_iterations'], rep['column_pass_iterations'], rep['monitor_statistic'], rep['monitor_min_spread'], rep['monitor_trips'],
        ' '.join('%.1e' % dev[min(i, m - 1)] for i in (5, 10, 20, 40, 60, 80, m - 1))), flush=True)
