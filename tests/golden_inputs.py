"""Seeded inputs shared by the golden-vector generator and the tests.

`gen_inputs` must stay byte-for-byte equivalent in behaviour to the function of
the same name in tests/golden/make_golden.py (the fixtures store only seeds and
reference outputs; inputs are regenerated from the legacy RandomState stream).
"""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)


def _normalize_rows(a):
    return a / (1e-16 + a.sum(axis=1, keepdims=True))


def gen_inputs(seed, n, f, k):
    rs = np.random.RandomState(seed)
    X = np.abs(rs.random_sample((n, f)))
    H0 = _normalize_rows(np.abs(rs.random_sample((k, f))) + .01)
    return X, H0


def g2_inputs(g):
    seed, n, f, k = int(g['seed']), int(g['n']), int(g['f']), int(g['k'])
    X, H0 = gen_inputs(seed, n, f, k)
    Xt = np.abs(np.random.RandomState(seed + 1).random_sample((17, f)))
    return X, H0, Xt


def g3_inputs(g):
    seed, n, f, k = int(g['seed']), int(g['n']), int(g['f']), int(g['k'])
    rs = np.random.RandomState(seed)
    X = np.abs(rs.random_sample((n, f)))
    W = np.abs(rs.random_sample((n, k)))
    H = np.abs(rs.random_sample((k, f)))
    return X, W, H


def g5_inputs(g):
    seed, n, k = int(g['seed']), int(g['n']), int(g['k'])
    dims = [int(d) for d in g['dims']]
    rs = np.random.RandomState(seed)
    blocks = [np.abs(rs.random_sample((n, d))) for d in dims]
    f = sum(dims)
    H0 = _normalize_rows(np.abs(rs.random_sample((k, f))) + .01)
    test = [np.abs(rs.random_sample((7, d))) for d in dims]
    return blocks, dims, H0, test


def g6_inputs(g):
    seed, n, f, k = int(g['seed']), int(g['n']), int(g['f']), int(g['k'])
    rs = np.random.RandomState(seed)
    dense = np.abs(rs.random_sample((n, f))) * (rs.random_sample((n, f)) < .5)
    W = np.abs(rs.random_sample((n, k)))
    H = np.abs(rs.random_sample((k, f)))
    return dense, W, H


def g9_inputs(g):
    seed, n, f, k = int(g['seed']), int(g['n']), int(g['f']), int(g['k'])
    rs = np.random.RandomState(seed)
    dense = np.abs(rs.random_sample((n, f))) * (rs.random_sample((n, f)) < .25)
    dense[7, :] = 0
    dense[:, 11] = 0
    H0 = _normalize_rows(np.abs(rs.random_sample((k, f))) + .01)
    return dense, H0


def g10_inputs(g):
    seed, na, nb, d = int(g['seed']), int(g['na']), int(g['nb']), int(g['d'])
    rs = np.random.RandomState(seed)
    A = np.abs(rs.random_sample((na, d))) * (rs.random_sample((na, d)) < .8)
    B = np.abs(rs.random_sample((nb, d)))
    A[3, :] = 0
    return A, B


def g8_edge_inputs():
    X, H0 = gen_inputs(81, 12, 9, 3)
    X[4, :] = 0
    X[:, 2] = 0
    return X, H0


def synthetic_problem(seed, n, f, k, block=8192):
    """The seeded synthetic workload of SURVEY.md section 8d (factorisable + noise, per-row-block RandomState
    streams) and its H0 -- the inputs of the large fixtures (tests/golden/make_golden_large.py) and of bench.py
    (multimodal_amd/synthetic.py generates the same blocks; tests/test_host_cpu.py checks they agree)."""
    Ht = np.random.RandomState(seed).gamma(0.5, 1.0, (k, f))
    X = np.empty((n, f))
    for b, r0 in enumerate(range(0, n, block)):
        r1 = min(n, r0 + block)
        rs = np.random.RandomState(seed + 1 + b)
        Wt = rs.gamma(1.0, 1.0, (r1 - r0, k))
        X[r0:r1] = Wt.dot(Ht) / k + 0.05 * rs.random_sample((r1 - r0, f))
    H0 = _normalize_rows(np.random.RandomState(seed - 1).random_sample((k, f)) + .01)
    return X, H0


def synthetic_modalities(seed, n, dims, k, scales=(1.0, 7.0, 0.2), block=8192):
    """Three (or more) dense modalities of the section-8d family over SHARED latent coefficients (one Wt per row
    block, one Ht per modality), each in its own units (`scales`), with the per-modality coefficients of
    experiment.py:70-72 (1 / mean row sum) -- the inputs of fixture G14 (tests/golden/make_golden_large.py), stacked by
    the learner as learner.py:53-56 does.  Returns (blocks, coefs, H0) with H0 [k, sum(dims)]."""
    f = sum(dims)
    Hts = [np.random.RandomState(seed + 100 * (m + 1)).gamma(0.5, 1.0, (k, d)) for m, d in enumerate(dims)]
    blocks = [np.empty((n, d)) for d in dims]
    for b, r0 in enumerate(range(0, n, block)):
        r1 = min(n, r0 + block)
        rs = np.random.RandomState(seed + 1 + b)
        Wt = rs.gamma(1.0, 1.0, (r1 - r0, k))
        for m, d in enumerate(dims):
            blocks[m][r0:r1] = scales[m % len(scales)] * (Wt.dot(Hts[m]) / k + 0.05 * rs.random_sample((r1 - r0, d)))
    coefs = [float(1.0 / np.mean(np.sum(x, axis=1))) for x in blocks]
    H0 = _normalize_rows(np.random.RandomState(seed - 1).random_sample((k, f)) + .01)
    return blocks, coefs, H0


def experiment_modalities(seed, n_per_label=14, n_labels=10, dims=(48, 30)):
    """Two dense non-negative modalities with class structure (fixture G13, tests/golden/make_golden_experiment.py):
    a sample of label l is its label's template plus noise, histogram-like (rows of the first modality sum to 1 as the
    motion histograms do, Appendix B of SURVEY.md).  Labels 0..9, listed in a shuffled order per modality."""
    rs = np.random.RandomState(seed)
    out = []
    for d in dims:
        templates = rs.gamma(0.4, 1.0, (n_labels, d)) + 0.02
        labels = np.repeat(np.arange(n_labels), n_per_label)
        rs.shuffle(labels)
        X = templates[labels] * (0.6 + 0.8 * rs.random_sample((labels.size, d))) + 0.05 * rs.random_sample((labels.size, d))
        out.append((X, [int(v) for v in labels]))
    Xa, la = out[0]
    out[0] = (Xa / Xa.sum(axis=1, keepdims=True), la)
    return out


def constant_columns_problem(seed, n, f, k):
    """Low-rank data + noise in which every 7th column is CONSTANT (round 4's fuzz class: the model fits those columns exactly,
    all their ratios collapse into one e4m3 cell); H0 by the reference's init rule from its own stream."""
    rs = np.random.RandomState(seed)
    X = rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))
    X[:, ::7] = 3.0
    H0 = _normalize_rows(np.abs(np.random.RandomState(seed + 1).random_sample((k, f))) + .01)
    return X, H0


def low_rank_problem(seed, n, f, k_true, k, block=8192):
    """Section-8d data of rank `k_true` (as `synthetic_problem`) fitted with k > k_true components: the H0 of k rows from the
    same stream rule.  Fixture G18 (rank 12 under k = 200: the class whose fp8-tile noise sits AT the monitor's threshold)."""
    X, _ = synthetic_problem(seed, n, f, k_true, block=block)
    H0 = _normalize_rows(np.random.RandomState(seed - 1).random_sample((k, f)) + .01)
    return X, H0


def steep_problem(n, f, k):
    """Low-noise data of rank k from ONE legacy stream (round 5's "steep transient" class: the fit sits on a plateau for ~100
    iterations and then escapes; scripts/monitor_calibration.py `steep`), H0 from the stream n - 1.  Fixture G19."""
    rs = np.random.RandomState(1)
    X = rs.gamma(1.0, 1.0, (n, k)).dot(rs.gamma(0.5, 1.0, (k, f))) / k + 0.05 * rs.random_sample((n, f))
    H0 = _normalize_rows(np.random.RandomState(n - 1).random_sample((k, f)) + .01)
    return X, H0
