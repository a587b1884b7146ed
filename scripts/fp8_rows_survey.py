#!/usr/bin/env python3
"""How few rows can a context have and still run fp8 ratio tiles within the north star's 1e-4?  The tiles' e4m3 rounding
(relative step 2^-4 .. 2^-3, zero-mean) only enters the H numerator, a sum over ALL rows: its relative error falls like
0.036 sqrt(2 / n).  For each row count: the SAME fit with 16-bit tiles (KLNMF_QTILE=16) and with fp8 tiles forced
(KLNMF_QTILE=8), both against the fp64 oracle.
    python scripts/fp8_rows_survey.py [k [iters [f]]]      (GPU box; the oracle runs on its CPU)
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import klnmf_oracle as orc  # noqa: E402
from tests import golden_inputs as gi  # noqa: E402
from multimodal_amd.lib import nmf  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 50
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = int(sys.argv[3]) if len(sys.argv) > 3 else 4096


def fit(X, H0, qtile):
    os.environ['KLNMF_QTILE'] = qtile
    m = nmf.KLdivNMF(n_components=k, max_iter=iters, tol=0, precision='f16')
    m._init_dictionary = H0
    with contextlib.redirect_stderr(io.StringIO()):
        W, e = m.fit_transform(X, return_errors=True)
    return m, W, np.asarray(e)


for n in [int(a) for a in os.environ.get('ROWS', '4096,8192,16384,32768,50000').split(',')]:
    X, H0 = gi.synthetic_problem(4321, n, f, k)
    Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0)
    fo = orc.kl_error(X, Wo, Ho)
    for q in ('16', '8'):
        m, W, e = fit(X, H0, q)
        rep = m.last_fp8_report
        true = nmf.KLdivNMF(n_components=k, precision='f64').error(X, W, H=m.components_)
        print('n %6d k %d f %d %d it  tiles %-6s (fp8 iterations %2d)  len %d/%d  max loss dev %.1e  true final KL dev %.1e  H dev %.1e of max   bound 0.036 sqrt(2/n) = %.1e'
              % (n, k, f, iters, 'fp8' if q == '8' else '16-bit', rep['tile_iterations'], len(e), len(eo),
                 np.max(np.abs(e - eo[:len(e)]) / eo[:len(e)]) if len(e) else float('nan'), abs(true - fo) / fo,
                 np.abs(m.components_ - Ho).max() / np.abs(Ho).max(), 0.036 * np.sqrt(2.0 / n)), flush=True)
