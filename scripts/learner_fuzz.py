"""Bug hunt, part 4: MultimodalLearner (learner.py:31-94) on the HIP path against the oracle -- modalities of very different
widths and magnitudes, the reference's coefficients (1 / mean row sum), training, coefficients from ONE modality (a column
slice of the dictionary: rows that no longer sum to 1), cross-modal reconstruction; KLNMF_PRECISION = f64 and f16.

    python3 scripts/learner_fuzz.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    from oracle import klnmf_oracle as orc
    bad = 0
    cases = [
        (200, (50, 30), (1.0, 1.0), 8),
        (300, (3, 2000), (1e-3, 50.0), 12),            # a 3-column modality of small numbers beside a wide one of large numbers
        (500, (450, 2000), (1.0, 1.0), 50),            # the reference's own two-modality shape
        (400, (64, 64, 64), (1e4, 1.0, 1e-4), 20),
        (40000, (40, 24), (1.0, 30.0), 10),            # fp8 sizes
        (70000, (96, 32), (1.0, 1.0), 40),
        (150, (500, 7), (1.0, 1.0), 130),
        (400, (4096, 16), (1.0, 1e3), 3),               # f / k in the thousands: the first update's ratio scale, on a slice too
        (35000, (2000, 48), (1e-5, 1.0), 10),           # fp8 sizes, one modality of tiny numbers
    ]
    for prec in ('f64', 'f16'):
        os.environ['KLNMF_PRECISION'] = prec
        import importlib
        import multimodal_amd.learner as L
        importlib.reload(L)
        for (n, dims, mags, k) in cases:
            rs = np.random.RandomState(n + sum(dims) + k)
            Wt = rs.gamma(1.0, 1.0, (n, k))
            blocks = [mag * (Wt.dot(rs.gamma(0.5, 1.0, (k, d))) / k + 0.05 * rs.random_sample((n, d))) for d, mag in zip(dims, mags)]
            coefs = [float(1. / np.mean(np.sum(b, axis=1))) for b in blocks]
            H0 = orc.synthetic_H0(5, sum(dims), k)
            iters = 10
            dico_o, W_o = orc.learner_train(blocks, coefs, k, iters, H0)
            names = ['m%d' % i for i in range(len(dims))]
            orig = L.NMF
            state = {'done': False}

            def factory(**kw):
                m = orig(**kw)
                if not state['done'] and kw.get('n_components') == k:
                    m._init_dictionary = H0.copy()
                    state['done'] = True
                return m
            L.NMF = factory
            try:
                lr = L.MultimodalLearner(names, list(dims), coefs, k)
                lr.train(blocks, iters)
                rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
                d_dico = rel(lr.dico, dico_o)
                nt = min(n, 64)
                test = [b[:nt] for b in blocks]
                out = []
                for mi in range(len(dims)):
                    a0, a1 = orc.axis_range(list(dims), mi)
                    Wi = lr.reconstruct_internal(names[mi], test[mi], 8)
                    Wio = orc.learner_internal([test[mi]], [coefs[mi]], [dico_o[:, a0:a1]], 8)
                    # against the oracle run on THIS learner's dictionary too (separates the transform's error from the training's)
                    Wio2 = orc.learner_internal([test[mi]], [coefs[mi]], [np.asarray(lr.dico)[:, a0:a1]], 8)
                    dest = (mi + 1) % len(dims)
                    b0, b1 = orc.axis_range(list(dims), dest)
                    rec = lr.modality_to_modality(names[mi], names[dest], test[mi], 8)
                    rec_o = Wio2.dot(np.asarray(lr.dico)[:, b0:b1])
                    out.append((rel(Wi, Wio), rel(Wi, Wio2), rel(rec, rec_o), bool(np.isfinite(Wi).all() and np.isfinite(rec).all())))
                lim_t, lim_s = (1e-7, 1e-7) if prec == 'f64' else (6e-3, 6e-3)
                ok = d_dico <= lim_t and all(o[3] and o[1] <= lim_s and o[2] <= lim_s for o in out)
                print('%-4s n=%-6d dims %-16s k=%-3d %s dico %.1e | per modality (W vs oracle-trained, W vs same dico, reconstruction): %s' % (
                    prec, n, dims, k, 'ok  ' if ok else 'FAIL', d_dico, ' '.join('(%.1e %.1e %.1e)' % o[:3] for o in out)), flush=True)
                bad += 0 if ok else 1
            except Exception as e:
                print('%-4s n=%-6d dims %-16s k=%-3d EXCEPTION %s' % (prec, n, dims, k, str(e)[:200]), flush=True)
                bad += 1
            finally:
                L.NMF = orig
    print('%d case(s) outside their tolerance' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
