#!/usr/bin/env python3
"""Calibration of the fp8 monitor (csrc/monitor.hip.h): for every data class round 4's fuzz campaign found outside the 1e-4 bar
on fp8 ratio tiles -- and for the healthy dense shapes of the BASELINE configurations -- three fits against the oracle:

    16-bit   KLNMF_QTILE=16            the tiles the loop falls back to
    monitor  (default)                 fp8 tiles from the third iteration, the monitor deciding whether they stay
    fp8      KLNMF_Q8_MONITOR=0        fp8 tiles whatever they do (what the monitor must prevent)

and per fit: final KL relative to the oracle's, fp8 iterations, the monitor's largest statistic, tripped rows, decision.
    python scripts/monitor_calibration.py [--quick] [--only NAME]   (GPU box; ~6 min)  ->  profiles/r05_monitor_calibration.txt"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['KLNMF_DEV'] = '1'
from oracle import klnmf_oracle as orc              # noqa: E402
from multimodal_amd.lib.nmf import KLdivNMF         # noqa: E402


def low_rank(seed, n, f, k, noise=0.05):
    rs = np.random.RandomState(seed)
    return rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + noise * rs.random_sample((n, f))


def classes(quick):
    out = []
    # ---- healthy: the configurations' own kind of data at sizes the oracle finishes
    out.append(('dense 40 000 x 512, k = 50 (C2 kind)', lambda: orc.synthetic_V(1234, 40000, 512, 50), 50, 30, 11))
    out.append(('dense 70 000 x 256, k = 200 (C4 kind, fp8 x fp8 pass)', lambda: orc.synthetic_V(1234, 70000, 256, 200), 200, 30, 11))
    out.append(('dense 66 000 x 384, k = 300 (C5 kind)', lambda: orc.synthetic_V(1234, 66000, 384, 300), 300, 12, 11))
    out.append(('uniform |U(0,1)| 50 000 x 300, k = 20 (tests/test_nmf_kl.py kind)',
                lambda: np.abs(np.random.RandomState(5).random_sample((50000, 300))), 20, 30, 11))
    # ---- round 4's defect classes
    def sparse():
        rs = np.random.RandomState(3)
        D = rs.gamma(1.0, 1.0, (70000, 40)).dot(rs.gamma(0.5, 1.0, (40, 96))) / 40 + 0.05 * rs.random_sample((70000, 96))
        return D * (rs.random_sample((70000, 96)) < 0.05)
    out.append(('sparse stored densely 70 000 x 96, 95 % zeros, k = 40', sparse, 40, 60 if not quick else 30, 11))
    out.append(('k = 1: 33 118 x 424', lambda: orc.synthetic_V(7 + 33118 + 424 + 1, 33118, 424, 1), 1, 8, 7 + 33118 + 424 + 1))
    out.append(('k = 2: 40 000 x 64', lambda: orc.synthetic_V(7 + 40000 + 64 + 2, 40000, 64, 2), 2, 8, 7 + 40000 + 64 + 2))
    out.append(('k = 2: 40 000 x 64, 40 iterations', lambda: orc.synthetic_V(7 + 40000 + 64 + 2, 40000, 64, 2), 2, 40, 7 + 40000 + 64 + 2))
    out.append(('f = 8: 103 431 x 8, k = 4', lambda: orc.synthetic_V(7 + 103431 + 8 + 4, 103431, 8, 4), 4, 20, 7 + 103431 + 8 + 4))
    out.append(('f = 3: 41 388 x 3, k = 10', lambda: orc.synthetic_V(7 + 41388 + 3 + 10, 41388, 3, 10), 10, 20, 7 + 41388 + 3 + 10))

    def const_cols(n, f, k):
        def make():
            X = low_rank(1, n, f, k)
            X[:, ::7] = 3.0
            return X
        return make
    out.append(('constant columns 40 000 x 500, k = 100', const_cols(40000, 500, 100), 100, 8, 11))
    out.append(('constant columns 66 000 x 300, k = 130 (fp8 x fp8 pass)', const_cols(66000, 300, 130), 130, 8, 11))
    def spiked(k):
        def make():
            rs = np.random.RandomState(5)
            X = orc.synthetic_V(13, 70000, 256, 12)
            X[:, 128:] = 1e-4 * rs.random_sample((70000, 128))
            for (i, j) in [(100, 131), (7000, 255), (30001, 192), (65999, 216)]:
                X[i, j] = 100.0 * X.mean()
            return X
        return make
    out.append(('rank 12 + empty half + spikes 70 000 x 256, k = 200 (the fix-up tests)', spiked(200), 200, 10, 13))
    out.append(('rank 12 + empty half + spikes 70 000 x 256, k = 50', spiked(50), 50, 10, 13))
    out.append(('rank 12 data, k = 200: 70 000 x 256 (run_more test)', lambda: orc.synthetic_V(13, 70000, 256, 12), 200, 7, 13))
    # ---- round 5: a steep transient of low-noise low-rank data (tol_fuzz's fifth matrix and wider ones), stopped mid-descent
    def steep(n, f, k):
        def make():
            rs = np.random.RandomState(1)
            return rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))
        return make
    for f_ in (64, 256, 1024):
        out.append(('steep: low-noise rank 8, 40 000 x %d, k = 8, 37 iterations' % f_, steep(40000, f_, 8), 8, 37, 40000))
    out.append(('steep: low-noise rank 50, 40 000 x 512, k = 50, 37 iterations', steep(40000, 512, 50), 50, 37, 40000))
    out.append(('steep: low-noise rank 16, 40 000 x 512, k = 16, 37 iterations', steep(40000, 512, 16), 16, 37, 40000))
    out.append(('steep: low-noise rank 32, 40 000 x 512, k = 32, 37 iterations', steep(40000, 512, 32), 32, 37, 40000))
    out.append(('steep: low-noise rank 8, 160 000 x 256, k = 8, 37 iterations', steep(160000, 256, 8), 8, 37, 40000))
    out.append(('steep: rank 8 data, 40 000 x 512, k = 32, 37 iterations', steep(40000, 512, 8), 32, 37, 40000))
    out.append(('steep: low-noise rank 8, 640 000 x 256, k = 8, 37 iterations', steep(640000, 256, 8), 8, 37, 40000))
    def sparse_wide(f_):
        def make():
            rs = np.random.RandomState(3)
            D = rs.gamma(1.0, 1.0, (70000, 40)).dot(rs.gamma(0.5, 1.0, (40, f_))) / 40 + 0.05 * rs.random_sample((70000, f_))
            return D * (rs.random_sample((70000, f_)) < 0.05)
        return make
    out.append(('sparse stored densely 70 000 x 256, 95 % zeros, k = 40', sparse_wide(256), 40, 30, 11))
    out.append(('k = 2: 40 000 x 256', lambda: orc.synthetic_V(7 + 40000 + 64 + 2, 40000, 256, 2), 2, 40, 7 + 40000 + 64 + 2))
    if not quick:
        out.append(('constant columns 66 000 x 300, k = 130, 100 iterations', const_cols(66000, 300, 130), 130, 100, 11))
    return out


def fit(X, H0, k, iters, env):
    saved = {n: os.environ.get(n) for n in ('KLNMF_QTILE', 'KLNMF_Q8_MONITOR')}
    for n in saved:
        os.environ.pop(n, None)
    os.environ.update(env)
    try:
        m = KLdivNMF(n_components=k, max_iter=iters, tol=0, precision='f16')
        m._init_dictionary = H0
        W, e = m.fit_transform(X, return_errors=True)
        return W, m.components_, e, m.last_fp8_report
    finally:
        for n, v in saved.items():
            os.environ.pop(n, None)
            if v is not None:
                os.environ[n] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--only', default=None)
    ap.add_argument('--iters', type=int, default=0, help='run every chosen class for this many iterations')
    args = ap.parse_args()
    print('%-66s %-8s %10s %5s %5s %10s %6s %-7s %s' % ('class', 'run', 'KL rel', 'len', 'fp8', 'statistic', 'trips', 'gave up',
                                                       '[uncentred bias, noise, common factor]  min spread  KL / sum(V)'))
    for name, make, k, iters, hseed in classes(args.quick):
        if args.only and args.only not in name:
            continue
        t0 = time.time()
        X = make()
        if args.iters:
            iters = args.iters
        n, f = X.shape
        H0 = orc.synthetic_H0(hseed, f, k)
        Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0, warn=False)
        for run, env in (('16-bit', {'KLNMF_QTILE': '16'}), ('monitor', {}), ('fp8', {'KLNMF_Q8_MONITOR': '0'})):
            W, H, e, rep = fit(X, H0, k, iters, env)
            # the oracle run to the same number of updates (tol = 0 stops on a rise: plateaus may end a run early)
            if len(e) != len(eo):
                Wr, Hr, er = orc.fit_transform(X, k=k, H0=H0, max_iter=max(1, len(e)), tol=0, warn=False)
            else:
                Wr, Hr = Wo, Ho
            fo = orc.kl_error(X, Wr, Hr)
            rel = abs(orc.kl_error(X, W.astype(np.float64), H.astype(np.float64)) - fo) / fo
            print('%-66s %-8s %10.2e %5d %5d %10.2e %6d %-7s [%s]  %.3f  %.2e' % (
                name, run, rel, len(e), rep['tile_iterations'], rep['monitor_statistic'], rep['monitor_trips'], rep['gave_up'],
                ', '.join('%.1e' % v for v in rep['monitor_parts']), rep['monitor_min_spread'], fo / X.sum()), flush=True)
        print('    (%d x %d, k = %d, %d iterations, oracle len %d; %.0f s)' % (n, f, k, iters, len(eo), time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
