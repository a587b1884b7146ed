#!/usr/bin/env python3
"""Deviation profile of a 16-bit fit against a large reference fixture (tests/golden/g11*, g12*): per-iteration relative loss
deviation, and the same deviation expressed in iterations of lead / lag along the reference's own loss curve.
    python scripts/fixture_profile.py g12_c2shape_200it [mode]"""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_inputs as gi
from multimodal_amd.lib import nmf

name = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
g = gi.load(name)
n, f, k, iters = int(g['n']), int(g['f']), int(g['k']), int(g['iters'])
X, H0 = gi.synthetic_problem(int(g['seed']), n, f, k)
m = nmf.KLdivNMF(n_components=k, max_iter=iters, tol=0, precision=mode)
m._init_dictionary = H0
with contextlib.redirect_stderr(io.StringIO()):
    W, e = m.fit_transform(X, return_errors=True)
e = np.asarray(e); ref = np.asarray(g['errors'])
dev = (e - ref[:len(e)]) / ref[:len(e)]
descent = np.r_[np.nan, (ref[:-1] - ref[1:]) / ref[1:]]
print(name, mode, 'len', len(e), '/', len(ref))
for i in list(range(0, len(e), max(1, len(e) // 25))) + [len(e) - 1]:
    print('%4d  ref %.8e  dev %+.2e  descent/iter %.2e  lag %.2f iterations' % (i, ref[i], dev[i], descent[i], dev[i] / descent[i] if i else 0))
fin = m.error(X, W)
true = nmf.KLdivNMF(n_components=k, precision='f64').error(X, W, H=m.components_)
print('max |dev| %.2e at %d; final reported %.2e true %.2e vs reference final' % (np.abs(dev).max(), int(np.abs(dev).argmax()),
      abs(fin - float(g['final'])) / float(g['final']), abs(true - float(g['final'])) / float(g['final'])))
