#!/usr/bin/env python3
"""One whole run of the two-modality experiment (split, train, 2 M transforms on test and example rows, cross-modal
reconstructions, nearest-example search under 4 measures in 3 comparison spaces: experiment.py:158-172, 233-277) at the
reference's own data scale (10^2..10^3 samples; motion histograms 450 columns; a dense stand-in of
2000 columns for the sound modality), timed on

    --reference   the REFERENCE itself (build container only: imports /root/reference, read-only)
    (default)     the GPU box: the oracle's CPU restatement of the same run (numpy, this host's cores), the HIP path with
                  host arrays between the steps (round 2's N2), and the HIP path with dictionary, coefficients and
                  reconstructions resident on the device (round 3's DeviceEvaluation)

    python scripts/time_experiment_run.py [--reference] [--k 50] [--runs 3]
"""
import argparse
import contextlib
import io
import os
import sys
import time
from itertools import product

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_inputs as gi  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('--reference', action='store_true')
p.add_argument('--k', type=int, default=50)
p.add_argument('--runs', type=int, default=3)
p.add_argument('--iters', type=int, default=50)
p.add_argument('--per-label', type=int, default=100)
args = p.parse_args()
DIMS = (450, 2000)
mods = gi.experiment_modalities(21, n_per_label=args.per_label, n_labels=10, dims=DIMS)
print('data: %d samples, modalities %s, k = %d, %d training / %d test iterations, test ratio 0.1' % (
    mods[0][0].shape[0], DIMS, args.k, args.iters, args.iters), flush=True)

if args.reference:
    sys.dont_write_bytecode = True
    np.Inf = np.inf
    sys.path.insert(0, '/root/reference')
    import random
    from collections import OrderedDict
    from multimodal.experiment import TwoModalitiesExperiment
    from multimodal.db.models.loader import Loader

    class ArrayLoader(Loader):
        dataset_name = 'synthetic'

        def __init__(self, data, labels):
            Loader.__init__(self)
            self._data, self._labels = data, labels

        def get_data(self):
            return self._data

        def get_labels(self):
            return list(self._labels)
    times = []
    for r in range(args.runs):
        np.random.seed(r)
        random.seed(r)
        exp = TwoModalitiesExperiment(OrderedDict([('motion', ArrayLoader(*mods[0])), ('sound', ArrayLoader(*mods[1]))]),
                                      args.k, args.iters, args.iters, run_mode='single')
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            exp.prepare()
            t0 = time.perf_counter()
            exp._perform_one_run()
            times.append(time.perf_counter() - t0)
    import threadpoolctl
    blas = [t.get('num_threads') for t in threadpoolctl.threadpool_info() if t.get('user_api') == 'blas']
    print('reference (numpy %s, BLAS threads %s, %d cores): %.3f s per run (runs: %s)' % (
        np.__version__, blas, len(os.sched_getaffinity(0)), float(np.median(times)), ' '.join('%.3f' % t for t in times)))
    sys.exit(0)

# ---- GPU box ----------------------------------------------------------------------------------------------------------
from oracle import klnmf_oracle as orc  # noqa: E402
from multimodal_amd.device_data import DeviceDataset  # noqa: E402
from multimodal_amd.device_experiment import perform_one_run  # noqa: E402

data = [m[0] for m in mods]
labels = mods[0][1]                        # (the two modalities list their samples in different orders in the experiment; the timing does not care)
n = data[0].shape[0]
examples = [labels.index(l) for l in range(10)]
others = [i for i in range(n) if i not in examples]
coefs = [float(1. / np.average(x.sum(axis=1))) for x in data]
names = ['motion', 'sound']


def split(r):
    rs = np.random.RandomState(r)
    perm = rs.permutation(len(others))
    nt = len(others) // 10
    return [others[i] for i in perm[nt:]], [others[i] for i in perm[:nt]], rs


def oracle_run(r):
    train, test, rs = split(r)
    H0 = rs.random_sample((args.k, sum(DIMS))) + .01
    H0 /= H0.sum(axis=1, keepdims=True)
    dico, _ = orc.learner_train([x[train] for x in data], coefs, args.k, args.iters, H0)
    sl = [dico[:, :DIMS[0]], dico[:, DIMS[0]:]]

    def transformations(rows):
        internals = [orc.learner_internal([data[m][rows]], [coefs[m]], [sl[m]], args.iters) for m in range(2)]
        out = [[None] * 2 for _ in range(2)]
        for i in range(2):
            out[i][i] = data[i][rows]
            out[i][1 - i] = internals[i].dot(sl[1 - i])
            out[i].append(internals[i])
        return out
    tt, te = transformations(test), transformations(examples)
    for m1, m2, cmp_ in product(range(2), range(2), [-1, 0, 1]):
        for name in ('kl_div', 'rev_kl_div', 'frobenius', 'cosine_diff'):
            np.argmin(orc.pairwise_distances(tt[m1][cmp_], te[m2][cmp_], name), axis=1)


def gpu_run(ds, r, on_device):
    train, test, rs = split(r)
    H0 = rs.random_sample((args.k, sum(DIMS))) + .01
    H0 /= H0.sum(axis=1, keepdims=True)
    perform_one_run(ds, names, coefs, args.k, args.iters, args.iters, train, test, examples, [labels[t] for t in test],
                    [labels[e] for e in examples], init_dictionary=H0, on_device=on_device)


def timed(fn, reps):
    out = []
    for r in range(reps):
        t0 = time.perf_counter()
        fn(r)
        out.append(time.perf_counter() - t0)
    return out


t = timed(oracle_run, args.runs)
print('oracle restatement on this host (%d cores, numpy %s): %.3f s per run (%s)' % (
    len(os.sched_getaffinity(0)), np.__version__, float(np.median(t)), ' '.join('%.3f' % v for v in t)), flush=True)
for prec in ('f64', 'f16'):
    os.environ['KLNMF_PRECISION'] = prec
    ds = DeviceDataset(data)
    for on_device in (False, True):
        gpu_run(ds, 0, on_device)                                     # warm-up: library load, contexts, block cache
        t = timed(lambda r: gpu_run(ds, r, on_device), max(args.runs, 5))
        print('HIP path, %s, %-52s %.4f s per run (%s)' % (prec, 'intermediates resident on the device:' if on_device else
                                                             'host arrays between the steps:', float(np.median(t)),
                                                             ' '.join('%.4f' % v for v in t)), flush=True)
