"""Bug hunt, part 6: the stop rule with the reference's tolerances (nmf.py:205-222: tol x n x f absolute, default 1e-6) -- where
does each mode stop, and is the "Iteration limit reached" warning (nmf.py:224-225) printed exactly when the oracle prints it?

    python3 scripts/tol_fuzz.py
"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    from multimodal_amd.lib import nmf
    from oracle import klnmf_oracle as orc
    rs = np.random.RandomState(1)
    bad = 0
    for (n, f, k) in [(120, 80, 6), (500, 1000, 10), (300, 64, 33), (2000, 300, 50), (40000, 64, 8), (70001, 64, 8)]:
        X = rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))
        H0 = orc.synthetic_H0(n, f, k)
        for tol in (1e-6, 1e-4, 1e-8, 1e-2):
            for max_iter in (200, 7):
                Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=max_iter, tol=tol)
                for prec in ('f64', 'f32', 'f16'):
                    m = nmf.KLdivNMF(n_components=k, max_iter=max_iter, tol=tol, precision=prec)
                    m._init_dictionary = H0
                    buf = io.StringIO()
                    with contextlib.redirect_stderr(buf):
                        W, errors = m.fit_transform(X, return_errors=True, scale_W=True)
                    warned = 'Iteration limit reached' in buf.getvalue()
                    # the oracle warns when its loop ran to max_iter with tol > 0 (nmf.py:224-225)
                    o_warned = len(eo) == max_iter and tol > 0
                    slack = 0 if prec == 'f64' else max(1, len(eo) // 50)
                    ok = abs(len(errors) - len(eo)) <= slack and (warned == o_warned or abs(len(errors) - len(eo)) > 0)
                    print('%-4s %6d x %4d k=%3d tol %g max_iter %3d  %s  len %d/%d  warned %d/%d' % (
                        prec, n, f, k, tol, max_iter, 'ok  ' if ok else 'FAIL', len(errors), len(eo), warned, o_warned), flush=True)
                    bad += 0 if ok else 1
    print('%d case(s) differ' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
