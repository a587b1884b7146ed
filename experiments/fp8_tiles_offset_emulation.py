"""Precision experiment (round 5, CPU): fp8 ratio tiles that hold t = ratio - 1 (signed e4m3) instead of ratio x sqrt(2) / 8.

Round 4's tiles spend e4m3's 3-bit significand on the ratio itself: around ratio 1 -- where every entry of a converging fit
lives -- a cell is 8.8 % wide, wider than the spread of a well-fitted column's ratios.  The rounding error is then not noise that
averages out over the rows but a pattern that depends on where a column's ratios sit relative to the cell boundaries: a bias of
~4e-4 in the H numerator that does NOT fall with the row count and is frozen from one iteration to the next; the slow modes of the
multiplicative update integrate it (scripts/fp8_drift_probe.py: final KL 1.9e-4 / 2.3e-4 off the oracle's after 200 iterations at
40 000 / 160 000 rows), exactly fitted columns lose the H rule's feedback altogether (a limit cycle between two cells), few
components sit in a dead zone.  A floating-point number's precision is relative to its distance from ZERO: storing t = ratio - 1
moves the fine end of the grid to where the ratios are -- |t| < 2^-6 has an absolute step of 2^-9 = 0.2 % of the ratio, 44 times
finer than before -- while the far end keeps e4m3's 6 % (ratios up to 449, x = 0 -> t = -1 exactly).  The H numerator becomes
    W^T.Q = colsum(W) + W^T.T        (nmf.py:349)
Exact fp64 updates, only the ratio that enters the H numerator rounded:  python3 experiments/fp8_tiles_offset_emulation.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from oracle import klnmf_oracle as orc


def q_scaled(q):          # round 4: e4m3(ratio x sqrt(2) / 8)
    s = 2 ** 0.5 / 8
    return torch.from_numpy(q * s).clamp(max=448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64).numpy() / s


def q_offset(q):          # round 5: 1 + e4m3(ratio - 1)
    return 1.0 + torch.from_numpy(q - 1.0).clamp(max=448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64).numpy()


def run(X, H0, iters, rnd, marks):
    H = H0.copy(); W = X.dot(H.T); out = {}
    for it in range(iters):
        Q = orc.ratio_q(X, W, H)
        Wn = orc.updated_w(X, W, H, Q=Q)
        Qh = Q if (rnd is None or it < 2) else np.where(Q >= 256.0, Q, rnd(Q))
        Hn = H * (Wn.T.dot(Qh)) / Wn.sum(axis=0)[:, None]; Hn = Hn / Hn.sum(axis=1, keepdims=True)
        W, H = Wn, Hn
        if it + 1 in marks:
            out[it + 1] = orc.kl_error(X, W, H)
    return out


rs = np.random.RandomState(1)
n, f, k = 20000, 300, 130
Xc = rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f)); Xc[:, ::7] = 3.0
cases = [('constant columns 20000x300 k=130', Xc, orc.synthetic_H0(11, f, k), (8, 20, 40, 60, 100)),
         ('dense 20000x512 k=50 (C2 kind)', orc.synthetic_V(1234, 20000, 512, 50), orc.synthetic_H0(11, 512, 50), (30, 50, 100, 150, 200)),
         ('k=2 40000x64', orc.synthetic_V(7 + 40000 + 64 + 2, 40000, 64, 2), orc.synthetic_H0(7 + 40000 + 64 + 2, 64, 2), (8, 20, 40)),
         ('k=1 33118x424', orc.synthetic_V(7 + 33118 + 424 + 1, 33118, 424, 1), orc.synthetic_H0(7 + 33118 + 424 + 1, 424, 1), (3, 6))]
for name, X, H0, marks in cases:
    ref = run(X, H0, max(marks), None, marks)
    for tag, rnd in (('ratio x sqrt(2)/8', q_scaled), ('ratio - 1', q_offset)):
        r = run(X, H0, max(marks), rnd, marks)
        print('%-34s %-18s' % (name, tag), ' '.join('%d:%+.1e' % (m, (r[m] - ref[m]) / ref[m]) for m in marks), flush=True)
