#!/usr/bin/env python3
"""numpy emulation of the 16-bit-operand KL-NMF iteration (the data flow of mfma4.hip.h / colq.hip.h) to see where
the noise of the REPORTED loss comes from and what it takes to keep `prev - err < 0` (tol = 0, learner.py:39-40)
from firing where the fp64 reference continues.

Per iteration, for operand type T in {bf16, fp16}:
    d  = T(W) . T(H)                      (fp32 accumulate ~ exact here)
    q  = (x + eps) / (d + eps);  reported loss = sum x log q - x + d
    G  = T(q) . T(H)^T;  W_new = W * G    (fp32 masters)
    N  = T(W_new)^T . T(q);  H_new = normalise(H * N)
and next to it: the TRUE loss of the fp32 masters (fp64 arithmetic), the first-order correction of the W rounding
that the W-rule tail could add for free (sum_ia dW_ia (hsum_a - G_ia)), and the fp64 reference trajectory.

    python experiments/loss_noise_emulation.py --n 8192 --f 2048 --k 100 --iters 80
"""
import argparse

import numpy as np


def r_bf16(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def r_fp16(a, scale):
    return (np.asarray(a, dtype=np.float32) * np.float32(scale)).astype(np.float16).astype(np.float32) / np.float32(scale)


def kl(x, d, eps=1e-8):
    return float((x * np.log((x + eps) / (d + eps)) - x + d).sum())


def synth(seed, n, f, k):
    rs = np.random.RandomState(seed)
    Ht = rs.gamma(0.5, 1.0, (k, f))
    Wt = rs.gamma(1.0, 1.0, (n, k))
    V = Wt @ Ht / k + 0.05 * rs.random_sample((n, f))
    H0 = rs.random_sample((k, f)) + .01
    H0 /= H0.sum(1, keepdims=True)
    return V, H0


def run(V, H0, iters, mode):
    eps = 1e-8
    x = V
    if mode == 'f64':
        rnd = lambda a, s=1.0: np.asarray(a, dtype=np.float64)
    elif mode == 'bf16':
        rnd = lambda a, s=1.0: r_bf16(a).astype(np.float64)
    else:
        rnd = lambda a, s=1.0: r_fp16(a, s).astype(np.float64)
    dt = np.float64 if mode == 'f64' else np.float32
    H = H0.astype(dt)
    W = (x @ H0.T).astype(dt)
    rep, true, corrW = [], [], []
    for _ in range(iters):
        sW = 2.0 ** (10 - np.ceil(np.log2(W.max())))
        sH = 2.0 ** (14 - np.ceil(np.log2(H.max())))
        Wr, Hr = rnd(W, sW), rnd(H, sH)
        d = Wr @ Hr
        q = (x + eps) / (d + eps)
        rep.append(float((x * np.log(q) - x + d).sum()))
        true.append(kl(x, W.astype(np.float64) @ H.astype(np.float64)))
        qr = rnd(q, 2.0 ** -8)
        G = qr @ Hr.T
        # first-order effect of the W rounding on the loss: sum_ia dW_ia * dL/dW_ia, dL/dW_ia = hsum_a - G_ia
        dW = W.astype(np.float64) - Wr
        corrW.append(float((dW * (Hr.sum(1)[None, :] - G)).sum()))
        Wn = (W.astype(np.float64) * G).astype(dt)
        N = rnd(Wn, sW).T @ qr
        Hn = H.astype(np.float64) * N
        Hn = Hn / (1e-16 + Hn.sum(1, keepdims=True))
        W, H = Wn, Hn.astype(dt)
    return np.array(rep), np.array(true), np.array(corrW), kl(x, W.astype(np.float64) @ H.astype(np.float64))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=8192)
    p.add_argument('--f', type=int, default=2048)
    p.add_argument('--k', type=int, default=100)
    p.add_argument('--iters', type=int, default=80)
    p.add_argument('--seed', type=int, default=3)
    a = p.parse_args()
    V, H0 = synth(a.seed, a.n, a.f, a.k)
    ref, _, _, ref_final = run(V, H0, a.iters, 'f64')
    print('fp64 reference: loss[0] %.6e  loss[-1] %.6e  final %.6e  monotone %s' % (ref[0], ref[-1], ref_final, bool((np.diff(ref) < 0).all())))
    print('relative descent per iteration (fp64):', ' '.join('%d:%.1e' % (i, (ref[i - 1] - ref[i]) / ref[i]) for i in (1, 2, 5, 10, 20, 40, a.iters - 1)))
    for mode in ('bf16', 'fp16'):
        rep, true, cw, fin = run(V, H0, a.iters, mode)
        up_rep = np.nonzero(np.diff(rep) >= 0)[0]
        up_true = np.nonzero(np.diff(true) >= 0)[0]
        up_cor = np.nonzero(np.diff(rep + cw) >= 0)[0]
        print('--- %s operands' % mode)
        print('  reported loss rises at iterations', (up_rep + 1).tolist()[:12], '(first stop: %s)' % (up_rep[0] + 1 if len(up_rep) else 'never'))
        print('  true loss of the masters rises at', (up_true + 1).tolist()[:12])
        print('  reported + W first-order rises at', (up_cor + 1).tolist()[:12])
        print('  (reported - true)/true   :', ' '.join('%d:%+.1e' % (i, (rep[i] - true[i]) / true[i]) for i in (0, 1, 2, 5, 10, 20, 40, a.iters - 1)))
        print('  (rep+corrW - true)/true  :', ' '.join('%d:%+.1e' % (i, (rep[i] + cw[i] - true[i]) / true[i]) for i in (0, 1, 2, 5, 10, 20, 40, a.iters - 1)))
        print('  (true - fp64 ref)/ref    :', ' '.join('%d:%+.1e' % (i, (true[i] - ref[i]) / ref[i]) for i in (0, 1, 2, 5, 10, 20, 40, a.iters - 1)))
        print('  final KL vs fp64 reference: %.2e relative' % (abs(fin - ref_final) / ref_final))


if __name__ == '__main__':
    main()
