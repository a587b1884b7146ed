"""Bug hunt, part 5: state carried from one problem to the next.  KLdivNMF keeps native contexts in a pool (stream, device blocks,
images, scales, the eps carrier choice, the ratio scale ...): a seeded random SEQUENCE of fits and transforms of changing shape,
magnitude, sparsity and precision in ONE process, every result checked against the oracle run on the same inputs.

    python3 scripts/sequence_fuzz.py [--steps 60] [--seed 0]
"""
import argparse
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    from multimodal_amd.lib import nmf
    from oracle import klnmf_oracle as orc
    rs = np.random.RandomState(args.seed)
    shapes = [(120, 80, 6), (120, 80, 6), (300, 2755, 1), (64, 33, 40), (33000, 40, 5), (40000, 64, 8), (70001, 64, 8), (500, 1000, 10), (260, 300, 256)]
    bad = 0
    for step in range(args.steps):
        n, f, k = shapes[rs.randint(len(shapes))]
        prec = ['f16', 'f16', 'f64', 'f32', 'f16'][rs.randint(5)]
        scale = [1.0, 1.0, 1e-6, 1e6, 1e-3, 1e3][rs.randint(6)]
        sparse = rs.rand() < 0.25
        fit = rs.rand() < 0.7
        iters = int(rs.randint(1, 6))
        X = (rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))) * scale
        if sparse:
            X = X * (rs.random_sample((n, f)) < 0.1)
        H0 = orc.synthetic_H0(int(rs.randint(1000)), f, k)
        if fit:
            Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0)
        else:
            Wo, eo = orc.transform(X, H0, max_iter=iters, tol=0)
            Ho = H0
        m = nmf.KLdivNMF(n_components=k, max_iter=iters, tol=0, precision=prec)
        m._init_dictionary = H0
        if not fit:
            m.components_ = H0
        buf = io.StringIO()
        try:
            with contextlib.redirect_stderr(buf):
                W, errors = m.fit_transform(X, return_errors=True, scale_W=True, _fit=fit)
        except Exception as e:
            print('step %2d %-7s %6d x %4d k=%3d scale %g sparse %d fit %d iters %d  EXCEPTION %s' % (step, prec, n, f, k, scale, sparse, fit, iters, str(e)[:150]), flush=True)
            bad += 1
            continue
        errors = np.array(errors)
        m_ = min(len(errors), len(eo))
        sx = float(X.sum())
        lim_e, lim_w = {'f64': (1e-9, 1e-7), 'f32': (3e-5, 3e-4)}.get(prec, (1e-3, 6e-3))
        floor_e = {'f64': 1e-12, 'f32': 1e-6}.get(prec, 1e-4) * sx
        rel_e = float(np.max(np.abs(errors[:m_] - np.array(eo[:m_])) / np.maximum(np.abs(eo[:m_]), floor_e))) if m_ else 0.0
        same = len(errors) == len(eo)
        dW = float(np.abs(W - Wo).max() / max(np.abs(Wo).max(), 1e-300)) if same else float('nan')
        dH = float(np.abs(m.components_ - Ho).max() / max(np.abs(Ho).max(), 1e-300)) if same else float('nan')
        ok = bool(np.isfinite(W).all()) and abs(len(errors) - len(eo)) <= (0 if prec == 'f64' else 1) and rel_e <= lim_e and (not same or (dW <= lim_w and dH <= lim_w))
        print('step %2d %-7s %6d x %4d k=%3d scale %-6g sparse %d fit %d iters %d  %s len %d/%d losses %.1e W %.1e H %.1e' % (
            step, prec, n, f, k, scale, sparse, fit, iters, 'ok  ' if ok else 'FAIL', len(errors), len(eo), rel_e, dW, dH), flush=True)
        bad += 0 if ok else 1
    print('%d step(s) outside their tolerance' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
