#!/usr/bin/env python3
"""The two reference fixtures at the edge of the 16-bit mode's envelope (G18: rank-12 data under k = 200; G19: a stop inside a
plateau escape), fitted in every tile regime: measured deviation of the recorded losses and of the final KL from the REFERENCE's,
the monitor's statistic, KL / sum(V).  What tests/test_gpu_parity.py bounds, as numbers (profiles/r06_envelope_fixtures.txt)."""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('KLNMF_DEV', '1')
from tests import golden_inputs as gi      # noqa: E402
from oracle import klnmf_oracle as orc      # noqa: E402
from multimodal_amd.lib import nmf      # noqa: E402

for name, make in (('g18_rank12_k200_150it', lambda g: gi.low_rank_problem(int(g['seed']), int(g['n']), int(g['f']), 12, int(g['k']))),
                   ('g19_plateau_escape_150it', lambda g: gi.steep_problem(int(g['n']), int(g['f']), int(g['k'])))):
    g = gi.load(name)
    X, H0 = make(g)
    k, iters, final = int(g['k']), int(g['iters']), float(g['final'])
    print('%s: %d x %d, k = %d, %d iterations; reference final KL %.6e, KL / sum(V) %.2e' % (name, X.shape[0], X.shape[1], k, iters, final, final / X.sum()))
    for label, prec, env in (('f16 default (fp8 tiles, monitored)', 'f16', {}), ('f16, KLNMF_QTILE=16 (16-bit tiles)', 'f16', {'KLNMF_QTILE': '16'}),
                             ('f16, fp8 tiles whatever the monitor says', 'f16', {'KLNMF_QTILE': '8', 'KLNMF_Q8_MONITOR': '0'}), ('f32', 'f32', {})):
        for a in ('KLNMF_QTILE', 'KLNMF_Q8_MONITOR'):
            os.environ.pop(a, None)
        os.environ.update(env)
        nmf._NOTED.clear()
        m = nmf.KLdivNMF(n_components=k, max_iter=iters, tol=0, precision=prec)
        m._init_dictionary = H0
        buf = io.StringIO()
        with contextlib.redirect_stderr(buf):
            W, e = m.fit_transform(X, return_errors=True)
        e = np.asarray(e)
        rep = m.last_fp8_report
        dev = abs(orc.kl_error(X, W.astype(np.float64), m.components_.astype(np.float64)) - final) / final
        c = min(len(e), len(g['errors']))
        print('   %-44s len %d  max loss dev %.2e  final KL dev %.2e  fp8 tile iterations %3d  gave up %-5s  monitor stat %.2e (threshold %.1e)  KL/sumV %.2e  noted: %s'
              % (label, len(e), np.abs(e[:c] / g['errors'][:c] - 1).max(), dev, rep['tile_iterations'], rep['gave_up'], rep['monitor_statistic'],
                 rep['monitor_threshold'], rep['kl_over_sum_v'], rep.get('outside_f16_envelope')))
