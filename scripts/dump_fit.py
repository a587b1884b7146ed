#!/usr/bin/env python3
"""Dump W, H and the loss record of a few seeded f16 fits to an .npz (to compare two builds of the library bit by bit:
KLNMF_LIB=ab/libklnmf_A.so scripts/dump_fit.py a.npz; KLNMF_LIB=... scripts/dump_fit.py b.npz; scripts/dump_fit.py --cmp a.npz b.npz)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

CASES = [(3000, 520, 200, 4), (70000, 256, 200, 5), (2100, 4096, 500, 3), (520, 1030, 200, 5)]


def main():
    if sys.argv[1] == '--cmp':
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = 0
        for key in a.files:
            same = np.array_equal(a[key], b[key])
            d = np.abs(a[key].astype(np.float64) - b[key].astype(np.float64)).max() / max(1e-300, np.abs(a[key]).max())
            print('%-28s %s  (max rel diff %.2e)' % (key, 'identical' if same else 'DIFFERENT', d))
            bad += 0 if same else 1
        sys.exit(1 if bad else 0)
    from multimodal_amd.lib.nmf import KLdivNMF
    from oracle import klnmf_oracle as orc
    out = {}
    for (n, f, k, iters) in CASES:
        X = orc.synthetic_V(5, n, f, min(k, 32))
        H0 = orc.synthetic_H0(5, f, k)
        try:
            m = KLdivNMF(n_components=k, max_iter=iters, tol=0, precision='f16')
            m._init_dictionary = H0
            W, e = m.fit_transform(X, return_errors=True)
        except RuntimeError as err:
            print('skipped %s: %s' % ((n, f, k), str(err)[:70]))
            continue
        tag = '%dx%dk%d' % (n, f, k)
        out[tag + '_W'] = W
        out[tag + '_H'] = m.components_
        out[tag + '_e'] = np.array(e)
        print(tag, 'loss', e[-1])
    np.savez(sys.argv[1], **out)


if __name__ == '__main__':
    main()
