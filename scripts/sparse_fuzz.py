"""Bug hunt, part 3: the CSR branch (nmf.py:52-70, 301-308, 331-334; csrc/sparse.hip.h) against the oracle's scipy restatement over
shapes, densities and structure: empty rows / columns, one entry, a dense row, f beyond the segmented H rule's threshold
(16 384), k = 1 ... 130, f64 / f32 kernels and the 16-bit modes' route to them.

    python3 scripts/sparse_fuzz.py
"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp


def main():
    from multimodal_amd.lib import nmf
    from oracle import klnmf_oracle as orc
    rs = np.random.RandomState(0)
    cases = []
    for (n, f, k, dens) in [(1, 1, 1, 1.0), (5, 7, 2, 0.5), (40, 300, 5, 0.05), (300, 40, 33, 0.2), (1000, 2000, 10, 0.01),
                            (64, 20000, 8, 0.002), (3000, 17000, 50, 0.003), (200, 500, 130, 0.1), (5000, 33, 1, 0.3),
                            (2000, 1000, 64, 0.0005), (31, 1025, 17, 0.03)]:
        X = sp.random(n, f, density=dens, format='csr', random_state=rs, data_rvs=lambda s: rs.gamma(1.0, 1.0, s))
        cases.append(('random %g' % dens, X, k))
    X = sp.random(400, 600, density=0.02, format='lil', random_state=rs, data_rvs=lambda s: rs.gamma(1.0, 1.0, s))
    X[5, :] = rs.gamma(1.0, 1.0, 600)            # one dense row
    X[:, 7] = 0; X[9, :] = 0                      # an empty column, an empty row
    cases.append(('dense row + empty row/column', X.tocsr(), 12))
    X = sp.csr_matrix(([3.5], ([2], [4])), shape=(6, 9))
    cases.append(('one entry', X, 3))
    X = sp.random(500, 800, density=0.05, format='csr', random_state=rs, data_rvs=lambda s: rs.gamma(1.0, 1.0, s) * 1e-6)
    cases.append(('values x 1e-6', X, 9))
    X = sp.random(500, 800, density=0.05, format='csr', random_state=rs, data_rvs=lambda s: rs.gamma(1.0, 1.0, s) * 1e6)
    cases.append(('values x 1e+6', X, 9))
    bad = 0
    for prec in ('f64', 'f32', 'f16'):
        for name, X, k in cases:
            n, f = X.shape
            H0 = orc.synthetic_H0(3, f, k)
            iters = 5
            for fit in (True, False):
                Wo, Ho, eo = orc.sparse_fit_transform(X, k, H0, max_iter=iters, tol=0, fit=fit, components=H0)
                m = nmf.KLdivNMF(n_components=k, max_iter=iters, tol=0, precision=prec)
                m._init_dictionary = H0
                if not fit:
                    m.components_ = H0
                buf = io.StringIO()
                try:
                    with contextlib.redirect_stderr(buf):
                        W, errors = m.fit_transform(X, return_errors=True, scale_W=True, _fit=fit)
                except Exception as e:
                    print('%-4s %5d x %5d k=%3d nnz %7d %-30s fit=%d EXCEPTION %s' % (prec, n, f, k, X.nnz, name, fit, str(e)[:140]), flush=True)
                    bad += 1
                    continue
                errors = np.array(errors)
                H = m.components_
                m_ = min(len(errors), len(eo))
                sx = float(X.sum())
                lim = 1e-9 if prec == 'f64' else 3e-5
                floor_e = (1e-12 if prec == 'f64' else 1e-6) * max(sx, 1e-300)
                rel_e = float(np.max(np.abs(errors[:m_] - np.array(eo[:m_])) / np.maximum(np.abs(eo[:m_]), floor_e))) if m_ else 0.0
                same = len(errors) == len(eo)
                dW = float(np.abs(W - Wo).max() / max(np.abs(Wo).max(), 1e-300)) if same else float('nan')
                dH = float(np.abs(H - Ho).max() / max(np.abs(Ho).max(), 1e-300)) if same else float('nan')
                finite = bool(np.all(np.isfinite(W)) and np.all(np.isfinite(H)) and np.all(np.isfinite(errors)))
                plateau = len(eo) >= 2 and abs(eo[-1] - eo[-2]) <= 1e-6 * max(abs(eo[-1]), floor_e)
                ok = finite and (same or (plateau and abs(len(errors) - len(eo)) <= 2)) and rel_e <= lim and (not same or (dW <= 100 * lim and dH <= 100 * lim))
                print('%-4s %5d x %5d k=%3d nnz %7d %-30s fit=%d %s len %d/%d losses %.1e W %.1e H %.1e  %s' % (
                    prec, n, f, k, X.nnz, name, fit, 'ok  ' if ok else 'FAIL', len(errors), len(eo), rel_e, dW, dH,
                    buf.getvalue().strip().replace('\n', ' | ')[:60]), flush=True)
                bad += 0 if ok else 1
    print('%d case(s) outside their tolerance' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
