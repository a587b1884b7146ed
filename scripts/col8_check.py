#!/usr/bin/env python3
"""fp8 x fp8 column pass (colq8x.hip.h) against the f16-operand forms of the same library and the oracle:
    scripts/col8_check.py [k] [f]        (70 000 rows; k = 200 -> KLNMF_COL8=0 is the comparison; k > 256 -> KLNMF_QTILE=16)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multimodal_amd.lib.nmf import KLdivNMF  # noqa: E402
from oracle import klnmf_oracle as orc  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
f = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n, iters = 70000, 7
X = orc.synthetic_V(5, n, f, 32)
H0 = orc.synthetic_H0(5, f, k)


def fit(env):
    for key in ('KLNMF_COL8', 'KLNMF_QTILE'):
        os.environ.pop(key, None)
    os.environ.update(env)
    m = KLdivNMF(n_components=k, max_iter=iters, tol=0, precision='f16')
    m._init_dictionary = H0.copy()
    W, e = m.fit_transform(X, return_errors=True)
    return W, m.components_, np.array(e)


Wa, Ha, ea = fit({'KLNMF_QTILE': '16'} if k > 256 else {'KLNMF_COL8': '0'})
Wb, Hb, eb = fit({})
Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0)
print('k = %d: losses fp8 x fp8 vs f16 operands %.2e   H rel-to-max %.2e   W %.2e   vs oracle: losses %.2e (f16 form: %.2e)' % (
    k, np.abs(ea / eb - 1).max(), np.abs(Ha - Hb).max() / np.abs(Ha).max(), np.abs(Wa - Wb).max() / np.abs(Wa).max(),
    np.abs(eb / np.array(eo) - 1).max(), np.abs(ea / np.array(eo) - 1).max()))
print('differs from the f16 form:', bool(np.abs(ea - eb).max() > 0))
