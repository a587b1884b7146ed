"""Next-row N3 with a roofline: the reference's CSR branch (nmf.py:52-70, 301-308, 331-334, 342, 349 -- SDDMM + two SpMMs per
fit iteration, csrc/sparseb.hip.h) on an Acorns-shaped matrix (HAC histograms, f = 110 000 columns, very
sparse): 20 000 x 110 000, 0.5 % stored entries, k = 50.  Prints ONE JSON line in bench.py's shape.

    python3 scripts/bench_sparse.py [--precision f64|f32] [--steps 10] [--no-cpu-baseline]

Roofline (HBM-bound integer / gather work, no MFMA).  `roofline.frac` is quoted on the COMPULSORY bytes of one fit iteration,
    stored entries:  fused ratio + W rule pass  nnz (idx 4 + x es + q es written)   H pass  nnz (row 4 + perm 4 + q es)
    factors:         W: read by the fused pass, its partial sums written and read, W_new written, W read again, column sums: 6 n k es
                     H: transpose (2), dots (1), partial numerator written and read (2), numerator (1), H rule (3) = 9 k f es
and the figure that includes the GATHERS -- one k-vector per stored entry and pass, 2 nnz k es (rows of H^T once, rows of W
once), what a cache-less machine would move and what the Infinity Cache / L2 actually serve -- stands beside it
(`with_gathers`).  `traffic`: HBM bytes per iteration from the committed PMC passes (profiles/r05_pmc_traffic_sparse.json:
FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs).  CPU baseline: the scipy restatement of the same branch
(oracle.sparse_fit_transform), fp64, on a row sample, scaled linearly in n."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp

PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=20000)
    ap.add_argument('--f', type=int, default=110000)
    ap.add_argument('--k', type=int, default=50)
    ap.add_argument('--density', type=float, default=0.005)
    ap.add_argument('--precision', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--repeats', type=int, default=3)
    ap.add_argument('--cpu-rows', type=int, default=2000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    from multimodal_amd import _native
    from oracle import klnmf_oracle as orc
    import bench

    n, f, k = args.n, args.f, args.k
    rs = np.random.RandomState(0)
    # uniformly placed entries, Gamma(1, 1) values: `per_row` random columns per row, sorted, duplicates merged (scipy.sparse.random
    # needs a minute for this matrix; this takes two seconds and gives the same kind of structure)
    per_row = max(1, int(round(args.density * f)))
    cols = np.sort(rs.randint(0, f, size=(n, per_row)), axis=1)
    X = sp.csr_matrix((rs.gamma(1.0, 1.0, n * per_row), cols.ravel(), np.arange(0, n * per_row + 1, per_row)), shape=(n, f))
    X.sum_duplicates()
    X.sort_indices()
    del cols
    H0 = orc.synthetic_H0(3, f, k)
    es = 8 if args.precision == 'f64' else 4
    nnz = int(X.nnz)
    seg, prof = [], None
    with _native.Context(args.precision) as c:
        c.set_problem_sparse(X.astype(np.float32) if es == 4 else X, k, args.steps + args.warmup)
        for rep in range(args.repeats):
            c.set_H(H0)
            c.init_W()
            c.loop_begin()
            c.run_more(args.warmup, True, 0.0)
            c.profile_enable(True)
            c.synchronize()
            t0 = time.perf_counter()
            c.run_more(args.steps, True, 0.0)
            c.synchronize()
            seg.append(time.perf_counter() - t0)
            prof = c.profile_read(reset=True)
            c.profile_enable(False)
            errs, nd, st = c.loop_end(args.steps + args.warmup)
    t = sorted(seg)[len(seg) // 2] / args.steps
    entries = nnz * ((4 + 2 * es) + (8 + es))
    gathers = 2 * nnz * k * es
    factors = 6 * n * k * es + 9 * k * f * es
    alg = entries + gathers + factors
    compulsory = entries + factors
    traffic = None
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r05_pmc_traffic_sparse.json')))
        for w in d.get('workloads', []):
            if (w.get('n'), w.get('f'), w.get('k'), w.get('precision')) == (n, f, k, args.precision):
                traffic = w
    except Exception:
        pass
    out = {
        'metric': 'nmf_update_iterations_per_sec', 'value': 1.0 / t, 'unit': 'it/s', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * t, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': 'KL-NMF fit iteration on CSR input (reference sparse branch), V %dx%d, %.2f %% stored (%d entries), k=%d'
                               % (n, f, 100.0 * nnz / (n * f), nnz, k),
                   'n': n, 'f': f, 'k': k, 'nnz': nnz, 'precision': args.precision,
                   'generator': '%d uniformly random columns per row (RandomState(0), duplicates merged), Gamma(1, 1) values' % per_row,
                   'timing': 'median of %d segments of %d iterations after %d warm-up iterations of the same loop' % (args.repeats, args.steps, args.warmup)},
        'valid': bool(nd == args.steps + args.warmup and not st),
        'loss_first': errs[0], 'loss_last': errs[-1],
        'loss_finite_and_decreasing': bool(all(b < a for a, b in zip(errs, errs[1:]))),
        'roofline': {'bound': 'hbm', 'kernel': 'the whole iteration (k_spb_qw: ratio + loss + W rule fused; k_spb_n: W^T.Q over CSC; dense H rule)',
                     'achieved': compulsory / t / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': compulsory / t / 1e9 / PEAK_HBM_GBS,
                     'traffic': traffic['hbm_bytes_per_iteration'] if traffic else None,
                     'traffic_source': 'profiles/r05_pmc_traffic_sparse.json' if traffic else None,
                     'traffic_over_compulsory': traffic['hbm_bytes_per_iteration'] / compulsory if traffic else None,
                     'compulsory_bytes': compulsory, 'of_which': {'stored_entries': entries, 'factors': factors},
                     'with_gathers': {'bytes': alg, 'gathers': gathers, 'achieved_gbs': alg / t / 1e9,
                                      'what': 'one k-vector per stored entry and pass (served by the Infinity Cache / L2, not HBM)'}},
        'kernels': {'k_spb_qw (ratio + loss + W rule)': {'avg_launch_ms': prof['rowpass_ms'] / max(1, prof['rowpass_launches']),
                                                         'gather_gbs': nnz * k * es / (prof['rowpass_ms'] / max(1, prof['rowpass_launches']) * 1e-3) / 1e9},
                    'k_spb_n (W^T.Q over CSC)': {'avg_launch_ms': prof['colpass_ms'] / max(1, prof['colpass_launches']),
                                                 'gather_gbs': nnz * k * es / (prof['colpass_ms'] / max(1, prof['colpass_launches']) * 1e-3) / 1e9}},
        'device': _native.device_info(0),
    }
    if not args.no_cpu_baseline:
        host = bench.host_info()
        rows = min(args.cpu_rows, n)
        Xs = X[:rows]
        orc.sparse_fit_transform(Xs, k, H0, max_iter=1, tol=0)
        t0 = time.perf_counter()
        iters = 3
        orc.sparse_fit_transform(Xs, k, H0, max_iter=iters, tol=0)
        dt = (time.perf_counter() - t0) / iters
        out['cpu_baseline'] = {'value': (1.0 / dt) * rows / n, 'unit': 'it/s', 'cores': 1, 'kind': 'port',
                               'sample': 'rows [0, %d) of the same matrix: %d timed fp64 iterations of the scipy restatement of the '
                                         'sparse branch (%.2f s / iteration on the sample), scaled linearly in n' % (rows, iters, dt),
                               'cpu_model': host.get('cpu_model'), 'numpy': host.get('numpy')}
    print(json.dumps(out))
    sys.stderr.write('%s: %.3f ms / iteration  (%.2f GB compulsory = %.1f %% of the HBM roof; with the gathers %.2f GB = %.0f GB/s; fused pass %.3f ms, W^T.Q %.3f ms)\n' % (
        args.precision, 1e3 * t, compulsory / 1e9, 100 * compulsory / t / 1e9 / PEAK_HBM_GBS, alg / 1e9, alg / t / 1e9,
        out['kernels']['k_spb_qw (ratio + loss + W rule)']['avg_launch_ms'], out['kernels']['k_spb_n (W^T.Q over CSC)']['avg_launch_ms']))


if __name__ == '__main__':
    main()
