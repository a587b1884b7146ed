"""Bug hunt, part 9: degenerate inputs in every mode -- all-zero V, constant V, one non-zero entry, a zero dictionary row, k larger than
both dimensions, NaN / inf / negative entries (the reference refuses them: sklearn_utils / check_non_negative), big k (FUSED-order
kernels, 224 < k <= 512) on stress data.  The oracle (or the reference's exception type) is the expectation.

    python3 scripts/degenerate_fuzz.py
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
    rs = np.random.RandomState(0)
    bad = 0

    def base(n, f, k):
        return rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))
    cases = []
    cases.append(('all zeros', np.zeros((40, 30)), 4, None))
    cases.append(('all zeros, fp8 size', np.zeros((33000, 40)), 4, None))
    cases.append(('constant 2.5', np.full((50, 70), 2.5), 5, None))
    X = np.zeros((60, 45)); X[7, 9] = 3.0
    cases.append(('one entry', X, 3, None))
    cases.append(('k > n and f', base(6, 5, 3), 20, None))
    H0 = orc.synthetic_H0(1, 80, 6); H0[2] = 0.0
    cases.append(('zero dictionary row', base(90, 80, 6), 6, H0))
    H0 = orc.synthetic_H0(1, 80, 6); H0[:, 11] = 0.0
    cases.append(('zero dictionary column under mass', base(90, 80, 6), 6, H0))
    cases.append(('big k 300, sparse', base(66000, 400, 20) * (rs.random_sample((66000, 400)) < 0.3), 300, None))
    cases.append(('big k 500, x 1e-6', base(300, 600, 30) * 1e-6, 500, None))
    cases.append(('big k 257, x 1e6, padded rows', base(65537 + 30, 64, 10) * 1e6, 257, None))
    for prec in ('f64', 'f32', 'f16'):
        for name, X, k, H0 in cases:
            n, f = X.shape
            H0c = orc.synthetic_H0(3, f, k) if H0 is None else H0
            iters = 4
            with np.errstate(all='ignore'):
                Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0c, max_iter=iters, tol=0)
            m = nmf.KLdivNMF(n_components=k, max_iter=iters, tol=0, precision=prec)
            m._init_dictionary = H0c
            buf = io.StringIO()
            try:
                with contextlib.redirect_stderr(buf):
                    W, errors = m.fit_transform(X, return_errors=True, scale_W=True)
            except Exception as e:
                print('%-4s %-34s %6d x %4d k=%3d EXCEPTION %s: %s' % (prec, name, n, f, k, type(e).__name__, str(e)[:120]), flush=True)
                bad += 1
                continue
            errors = np.array(errors)
            H = m.components_
            m_ = min(len(errors), len(eo))
            lim_e, lim_w = {'f64': (1e-9, 1e-7), 'f32': (3e-5, 3e-4)}.get(prec, (1e-3, 6e-3))
            floor_e = {'f64': 1e-12, 'f32': 1e-6}.get(prec, 1e-4) * max(float(X.sum()), 1e-300) + 1e-300
            o_fin = bool(np.all(np.isfinite(Wo)) and np.all(np.isfinite(Ho)))
            g_fin = bool(np.all(np.isfinite(W)) and np.all(np.isfinite(H)))
            rel_e = float(np.max(np.abs(errors[:m_] - np.array(eo[:m_])) / np.maximum(np.abs(eo[:m_]), floor_e))) if m_ else 0.0
            same = len(errors) == len(eo)
            dW = float(np.abs(W - Wo).max() / max(np.abs(Wo).max(), 1e-300)) if same and o_fin and g_fin else float('nan')
            dH = float(np.abs(H - Ho).max() / max(np.abs(Ho).max(), 1e-300)) if same and o_fin and g_fin else float('nan')
            ok = (g_fin == o_fin) and abs(len(errors) - len(eo)) <= (0 if prec == 'f64' else 1) and (rel_e <= lim_e or not np.isfinite(rel_e) and not o_fin) and \
                (not (same and o_fin and g_fin) or (dW <= lim_w and dH <= lim_w))
            print('%-4s %-34s %6d x %4d k=%3d %s len %d/%d finite %d/%d losses %.1e W %.1e H %.1e %s' % (
                prec, name, n, f, k, 'ok  ' if ok else 'FAIL', len(errors), len(eo), g_fin, o_fin, rel_e, dW, dH, buf.getvalue().strip()[:60]), flush=True)
            bad += 0 if ok else 1
        # inputs the reference refuses
        for name, X in (('NaN entry', np.array([[1., np.nan], [2., 3.]])), ('inf entry', np.array([[1., np.inf], [2., 3.]])), ('negative entry', np.array([[1., -1e-9], [2., 3.]]))):
            m = nmf.KLdivNMF(n_components=1, max_iter=2, tol=0, precision=prec)
            try:
                m.fit_transform(X)
                print('%-4s %-34s NOT REFUSED' % (prec, name)); bad += 1
            except ValueError as e:
                print('%-4s %-34s ok   ValueError: %s' % (prec, name, str(e)[:70]), flush=True)
    print('%d case(s) outside their expectation' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
