"""Precision experiment (round 5, CPU): stochastic rounding of the fp8 ratio tiles.

Round-to-nearest e4m3 of ratio x sqrt(2) / 8 has cells 8.8 % wide around ratio 1; a column's ratios of a converging fit are a peak
of about that width, so the rounding error has a MEAN that depends on where the peak sits in the cell grid: a bias of some 1e-4 in
the H numerator that does not fall with the row count and stays from one iteration to the next (scripts/fp8_drift_probe.py, the
monitor's "centred" statistic).  With stochastic rounding every stored entry is unbiased, E[q8(r)] = r: the numerator's error is
noise of 2^-4 / sqrt(rows), fresh every iteration.
Exact fp64 updates, only the ratio that enters the H numerator rounded:  python3 experiments/fp8_tiles_stochastic_rounding_emulation.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from oracle import klnmf_oracle as orc

S = 2 ** 0.5 / 8


def q_rne(q, rs=None):
    return torch.from_numpy(q * S).clamp(max=448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64).numpy() / S


def q_sr(q, rs, bits=None):
    """e4m3 neighbours of x = q S: step 2^(e - 3), e = max(floor(log2 x), -6); up with probability (x - lo) / step
    (bits: the number of random bits the hardware conversion would use, None = exact)."""
    x = np.minimum(q * S, 448.0)
    e = np.maximum(np.floor(np.log2(np.maximum(x, 2.0 ** -20))), -6.0)
    step = 2.0 ** (e - 3)
    lo = np.floor(x / step) * step
    p = (x - lo) / step
    u = rs.random_sample(x.shape)
    if bits is not None:
        u = np.floor(u * 2 ** bits) / 2 ** bits
        p = np.floor(p * 2 ** bits) / 2 ** bits          # truncating add of the random bits below the kept significand
    return np.where(u < p, lo + step, lo) / S


def run(X, H0, iters, rnd, marks, seed=0):
    H = H0.copy(); W = X.dot(H.T); out = {}
    rs = np.random.RandomState(seed)
    for it in range(iters):
        Q = orc.ratio_q(X, W, H)
        Wn = orc.updated_w(X, W, H, Q=Q)
        Qh = Q if (rnd is None or it < 2) else np.where(Q >= 256.0, Q, rnd(Q, rs))
        Hn = H * (Wn.T.dot(Qh)) / Wn.sum(axis=0)[:, None]; Hn = Hn / Hn.sum(axis=1, keepdims=True)
        W, H = Wn, Hn
        if it + 1 in marks:
            out[it + 1] = orc.kl_error(X, W, H)
    return out


def steep(n, f, k):
    rs = np.random.RandomState(1)
    return rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))


if __name__ == '__main__':
    rs = np.random.RandomState(1)
    n, f, k = 20000, 300, 130
    Xc = rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f)); Xc[:, ::7] = 3.0
    cases = [('steep rank 8 40000x256 k=8', steep(40000, 256, 8), orc.synthetic_H0(40000, 256, 8), (10, 20, 37, 50)),
             ('steep rank 16 40000x512 k=16', steep(40000, 512, 16), orc.synthetic_H0(40000, 512, 16), (10, 20, 37, 50)),
             ('rank 8 data 40000x512 k=32', steep(40000, 512, 8), orc.synthetic_H0(40000, 512, 32), (10, 20, 37, 50)),
             ('k=2 40000x64', orc.synthetic_V(7 + 40000 + 64 + 2, 40000, 64, 2), orc.synthetic_H0(7 + 40000 + 64 + 2, 64, 2), (8, 20, 40)),
             ('k=1 33118x424', orc.synthetic_V(7 + 33118 + 424 + 1, 33118, 424, 1), orc.synthetic_H0(7 + 33118 + 424 + 1, 424, 1), (3, 6)),
             ('constant columns 20000x300 k=130', Xc, orc.synthetic_H0(11, f, k), (8, 20, 40, 60, 100)),
             ('dense 20000x512 k=50 (C2 kind)', orc.synthetic_V(1234, 20000, 512, 50), orc.synthetic_H0(11, 512, 50), (30, 50, 100, 150, 200))]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, X, H0, marks in cases:
        if only and only not in name:
            continue
        ref = run(X, H0, max(marks), None, marks)
        for tag, rnd in (('nearest', q_rne), ('stochastic', q_sr), ('stochastic, 7 bits', lambda q, r: q_sr(q, r, 7))):
            r = run(X, H0, max(marks), rnd, marks)
            print('%-34s %-20s' % (name, tag), ' '.join('%d:%+.1e' % (m, (r[m] - ref[m]) / ref[m]) for m in marks), flush=True)
