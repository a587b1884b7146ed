#!/usr/bin/env python3
"""Hybrid update pass (whole rows for the full rounds of workgroups, column-split last partial round: DESIGN h18) against
the whole-row pass (KLNMF_ROW_TAIL=0) of the same library, and against the oracle on the smaller case."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multimodal_amd.lib.nmf import KLdivNMF  # noqa: E402
from oracle import klnmf_oracle as orc  # noqa: E402


def fit(X, H0, k, iters, tail):
    if tail is None:
        os.environ.pop('KLNMF_ROW_TAIL', None)
    else:
        os.environ['KLNMF_ROW_TAIL'] = str(tail)
    m = KLdivNMF(n_components=k, max_iter=iters, tol=0, precision='f16')
    m._init_dictionary = H0.copy()
    t = time.time()
    W, e = m.fit_transform(X, return_errors=True)
    return W, m.components_, np.array(e), time.time() - t


def main():
    ok = True
    for (n, f, k, iters, with_oracle) in [(70000, 256, 200, 6, True), (70000, 1024, 200, 5, False), (66000, 512, 72, 5, False)]:
        X = orc.synthetic_V(7, n, f, min(k, 32))
        H0 = orc.synthetic_H0(7, f, k)
        try:
            Wa, Ha, ea, ta = fit(X, H0, k, iters, None)
        except RuntimeError as err:      # a development build of the library holds only some k
            print('%d x %d k=%d: skipped (%s)' % (n, f, k, str(err)[:60]))
            continue
        Wb, Hb, eb, tb = fit(X, H0, k, iters, 0)
        dW = np.abs(Wa - Wb).max() / np.abs(Wb).max()
        dH = np.abs(Ha - Hb).max() / np.abs(Hb).max()
        de = np.abs(ea / eb - 1).max()
        line = '%d x %d k=%d: hybrid vs whole rows: W %.2e  H %.2e  losses %.2e' % (n, f, k, dW, dH, de)
        good = dW < 2e-4 and dH < 2e-4 and de < 1e-6 and len(ea) == len(eb) == iters
        if with_oracle:
            Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0)
            do = np.abs(ea / np.array(eo) - 1).max()
            fo = abs(orc.kl_error(X, Wa.astype(np.float64), Ha.astype(np.float64)) / orc.kl_error(X, Wo, Ho) - 1)
            line += '   vs oracle: losses %.2e final KL %.2e' % (do, fo)
            good = good and do < 1e-4 and fo < 1e-4
        print(line, 'ok' if good else 'FAIL', flush=True)
        ok = ok and good
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
