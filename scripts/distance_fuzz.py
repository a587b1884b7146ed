"""Bug hunt, part 7: the nearest-neighbour measures (metrics.py:58-86 through klnmf_all_distances) and klnmf_matmul against numpy:
shapes 1 ... 5000 x 1 ... 3000 vectors of length 1 ... 4100, zero vectors, zero entries, magnitudes 1e-9 ... 1e9, fp32 / fp64.

    python3 scripts/distance_fuzz.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    from multimodal_amd.lib import metrics
    from multimodal_amd import _native
    from oracle import klnmf_oracle as orc
    rs = np.random.RandomState(0)
    bad = 0
    names = ['kl_div', 'rev_kl_div', 'sym_kl_div', 'frobenius', 'cosine_diff']
    fns = {nm: getattr(metrics, nm) for nm in names}
    for (na, nb, d) in [(1, 1, 1), (3, 5, 7), (48, 10, 50), (1000, 10, 200), (10, 1000, 33), (257, 129, 1025), (5000, 17, 64), (2, 3000, 4100), (64, 64, 1)]:
        for scale in (1.0, 1e-9, 1e9):
            for dt in (np.float64, np.float32):
                A = (rs.gamma(0.5, 1.0, (na, d)) * scale).astype(dt)
                B = (rs.gamma(0.5, 1.0, (nb, d)) * scale).astype(dt)
                A[rs.random_sample(A.shape) < 0.2] = 0
                B[rs.random_sample(B.shape) < 0.2] = 0
                if na > 2:
                    A[1] = 0                       # a zero vector (cosine_diff: 0 by the reference's rule)
                for nm in names:
                    want = orc.pairwise_distances(A, B, nm)
                    got = fns[nm](A[:, np.newaxis, :], B[np.newaxis, :, :], axis=-1)
                    lim = 1e-11 if dt == np.float64 else 2e-4
                    # KL measures cancel (x log(x/y) - x + y): compare on the scale of the vectors' mass
                    ref = np.maximum(np.abs(want), 1e-3 * (np.abs(A).sum(axis=1)[:, None] + np.abs(B).sum(axis=1)[None, :]) if 'kl' in nm else 1e-300)
                    if nm == 'cosine_diff':
                        ref = np.maximum(np.abs(want), 1e-3)
                    if nm == 'frobenius':
                        ref = np.maximum(np.abs(want), 1e-6 * scale)
                    err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want) / np.maximum(ref, 1e-300)))
                    ok = np.isfinite(got).all() and err <= lim
                    if not ok:
                        print('%-11s %5d x %5d d=%4d scale %g %s  FAIL err %.2e' % (nm, na, nb, d, scale, dt.__name__, err), flush=True)
                        bad += 1
        print('distances %5d x %5d d=%4d done' % (na, nb, d), flush=True)
    for (m, n, kk) in [(1, 1, 1), (7, 5, 3), (64, 2450, 50), (1000, 450, 200), (33, 4097, 513), (5000, 64, 1)]:
        for dt in (np.float64, np.float32):
            A = rs.random_sample((m, kk)).astype(dt)
            B = rs.random_sample((kk, n)).astype(dt)
            got = _native.matmul(A, B)
            want = A.astype(np.float64).dot(B.astype(np.float64))
            err = float(np.abs(got - want).max() / np.abs(want).max())
            ok = err <= (1e-13 if dt == np.float64 else 3e-6)
            if not ok:
                print('matmul %d x %d x %d %s FAIL %.2e' % (m, n, kk, dt.__name__, err), flush=True)
                bad += 1
    print('%d case(s) outside their tolerance' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
