"""Numerical estimate for storing the ratio tiles in 8 bits (next-round item): the H rule of the
reference loop with the ratio rounded to 7 / 3 / 2 explicit mantissa bits ONLY in the product W_new^T . Q (the W rule and the
loss keep the exact ratio, as the row pass has it in registers).  Prints the relative deviation of the loss per iteration from
the unrounded loop.  numpy, CPU:  python experiments/fp8_ratio_tiles_precision.py"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import klnmf_oracle as orc
def round_mant(x, bits):
    m, e = np.frexp(x)                      # x = m * 2^e, m in [0.5, 1)
    s = 2.0 ** (bits + 1)
    return np.ldexp(np.round(m * s) / s, e)
n, f, k, iters = 4096, 2048, 100, 8
X = orc.synthetic_V(1234, n, f, k); H0 = orc.synthetic_H0(1234, f, k)
eps = 1e-8
def run(qbits_h, wbits=None):
    W = X @ H0.T; H = H0.copy(); errs = []
    for it in range(iters):
        WH = W @ H
        errs.append(float(np.sum(X * np.log((X + eps) / (WH + eps)) - X + WH)))
        Q = (X + eps) / (WH + eps)
        Wn = W * (Q @ H.T)
        Qh = Q if qbits_h is None else round_mant(Q, qbits_h)
        Hn = H * (Wn.T @ Qh)
        H = Hn / (1e-16 + Hn.sum(axis=1, keepdims=True)); W = Wn
    WH = W @ H
    errs.append(float(np.sum(X * np.log((X + eps) / (WH + eps)) - X + WH)))
    return np.array(errs)
e_ref = run(None)
for bits, name in [(7, 'bf16 (7 explicit bits)'), (3, 'fp8 e4m3 (3 bits)'), (2, 'fp8 e5m2 (2 bits)')]:
    e = run(bits)
    print('%-24s rel loss diff per iter: %s' % (name, np.array2string(np.abs(e - e_ref) / e_ref, precision=1)))
