# encoding: utf-8
"""One run of a multimodal experiment on device-resident data (next-row N2 of SURVEY.md 8f).

Mirrors what the reference's driver does per run -- `MultimodalExperiment._perform_one_run` (experiment.py:158-172:
split, train, evaluate) and `TwoModalitiesExperiment._evaluate` / `_get_all_transformations` (experiment.py:233-277:
internal coefficients of the test set and of the examples from every modality, cross-modal reconstructions,
nearest-example classification under four measures in every comparison space) -- with the modalities uploaded ONCE
(`DeviceDataset`): a run hands over row indices, the rows are gathered by the upload kernel, the fits and transforms run
through the HIP path and the distance matrices through `klnmf_all_distances`.  Result keys are the reference's
(`found_<m1>2<m2>[_<space>][_bis|_frob|_cosine]`, `score_...`), so its `Logger` and result tables apply unchanged.

`sweep_assignment` spreads the runs of a k sweep (samples/launcher.py:68: Ks 5 .. 200 x 20 runs, one OS process each in
the reference) over the ranks of a node: independent fits, no communication -- replica parallelism.
"""
from itertools import product

import numpy as np

from . import _native
from .evaluation import classify_NN, found_labels_to_score
from .learner import MultimodalLearner
from .lib.metrics import kl_div, rev_kl_div, frobenius, cosine_diff

INTERNAL = -1
MEASURES = ((kl_div, ''), (rev_kl_div, '_bis'), (frobenius, '_frob'), (cosine_diff, '_cosine'))
DEVICE_MEASURES = ((_native.DIST_KL, ''), (_native.DIST_REV_KL, '_bis'), (_native.DIST_FROBENIUS, '_frob'),
                   (_native.DIST_COSINE_DIFF, '_cosine'))


def exp_key(modalities, mod1, mod2, mod_cmp, suffix):
    """experiment.py:279-283."""
    return "{}2{}{}{}".format(modalities[mod1], modalities[mod2],
                              '' if mod_cmp == INTERNAL else '_' + modalities[mod_cmp], suffix)


def all_transformations(dataset, learner, rows, iter_test):
    """result[input modality][output modality] (+ the internal coefficients as the last entry), experiment.py:259-277."""
    M = len(learner.mod)
    internals = [dataset.reconstruct_internal(learner, learner.mod[m], rows, iter_test) for m in range(M)]
    out = [[None] * M for _ in range(M)]
    for i in range(M):
        out[i][i] = dataset.rows_of(i, rows)                # nothing to do
        for o in range(M):
            if o != i:
                out[i][o] = learner.reconstruct_modality(learner.mod[o], internals[i])
        out[i].append(internals[i])
    return out


def evaluate(dataset, learner, rows_test, rows_ex, labels_test, labels_ex, iter_test):
    """TwoModalitiesExperiment._evaluate (experiment.py:233-257) -> {key: found labels / score}."""
    M = len(learner.mod)
    t_test = all_transformations(dataset, learner, rows_test, iter_test)
    t_ex = all_transformations(dataset, learner, rows_ex, iter_test)
    results = {}
    for mod1, mod2, mod_cmp in product(range(M), range(M), [INTERNAL] + list(range(M))):
        for measure, suffix in MEASURES:
            found = classify_NN(t_test[mod1][mod_cmp], t_ex[mod2][mod_cmp], labels_ex, measure)
            key = exp_key(learner.mod, mod1, mod2, mod_cmp, suffix)
            results['found_' + key] = found
            results['score_' + key] = found_labels_to_score(labels_test, found)
    return results


# ---- ThreeModalitiesExperiment (experiment.py:322-404): comparisons in the internal space only, between single
# modalities and between a pair of modalities and the third ------------------------------------------------------------
def tested_combinations(n_modalities):
    """experiment.py:352-361: (mods1, mods2) as tuples of modality indices, in the order `_evaluate` walks them."""
    M = n_modalities
    combos = [([m1], [m2]) for m1 in range(M) for m2 in range(M) if m1 != m2]
    two_to_one = [([m for m in range(M) if m != mod], [mod]) for mod in range(M)]
    combos += two_to_one
    combos += [(y, x) for (x, y) in two_to_one]
    return [(tuple(x), tuple(y)) for (x, y) in combos]


def combo_key(modalities, mods1, mods2, suffix):
    """experiment.py:373-377 without the 'score_' prefix."""
    return "{}2{}{}".format('_'.join(modalities[m] for m in mods1), '_'.join(modalities[m] for m in mods2), suffix)


def all_internals(dataset, learner, rows, iter_test):
    """experiment.py:363-371: internal coefficients of `rows` from every modality set that occurs in a combination -- one
    transform per distinct set (three singles and three pairs for three modalities)."""
    internals = {}
    for mods, _rest in tested_combinations(len(learner.mod)):
        if mods not in internals:
            internals[mods] = dataset.reconstruct_internal_multi(learner, [learner.mod[m] for m in mods], rows, iter_test)
    return internals


def evaluate_internal(dataset, learner, rows_test, rows_ex, labels_test, labels_ex, iter_test):
    """ThreeModalitiesExperiment._evaluate (experiment.py:332-350) -> {found_<key>: labels, score_<key>: score}."""
    t_test = all_internals(dataset, learner, rows_test, iter_test)
    t_ex = all_internals(dataset, learner, rows_ex, iter_test)
    results = {}
    for mods1, mods2 in tested_combinations(len(learner.mod)):
        for measure, suffix in MEASURES:
            found = classify_NN(t_test[mods1], t_ex[mods2], labels_ex, measure)
            key = combo_key(learner.mod, mods1, mods2, suffix)
            results['found_' + key] = found
            results['score_' + key] = found_labels_to_score(labels_test, found)
    return results


# ---- the same two evaluations with everything between the transforms and the label lists ON the device (DeviceEvaluation:
# dictionary uploaded once, coefficients / reconstructions / raw rows as device matrices, distances by one kernel) --------
def evaluate_on_device(dataset, learner, rows_test, rows_ex, labels_test, labels_ex, iter_test):
    """`evaluate` (TwoModalitiesExperiment._evaluate) without host round trips."""
    from .device_data import DeviceEvaluation
    ev = DeviceEvaluation(dataset, learner, iter_test)
    M = len(learner.mod)

    def transformations(rows):
        internals = [ev.internal([learner.mod[m]], rows) for m in range(M)]
        out = [[None] * M for _ in range(M)]
        for i in range(M):
            out[i][i] = ev.raw(i, rows)
            for o in range(M):
                if o != i:
                    out[i][o] = ev.reconstruct(internals[i], learner.mod[o])
            out[i].append(internals[i])
        return out
    t_test, t_ex = transformations(rows_test), transformations(rows_ex)
    results = {}
    for mod1, mod2, mod_cmp in product(range(M), range(M), [INTERNAL] + list(range(M))):
        for metric, suffix in DEVICE_MEASURES:
            found = ev.found_labels(t_test[mod1][mod_cmp], t_ex[mod2][mod_cmp], labels_ex, metric)
            key = exp_key(learner.mod, mod1, mod2, mod_cmp, suffix)
            results['found_' + key] = found
            results['score_' + key] = found_labels_to_score(labels_test, found)
    return results


def evaluate_internal_on_device(dataset, learner, rows_test, rows_ex, labels_test, labels_ex, iter_test):
    """`evaluate_internal` (ThreeModalitiesExperiment._evaluate) without host round trips."""
    from .device_data import DeviceEvaluation
    ev = DeviceEvaluation(dataset, learner, iter_test)
    combos = tested_combinations(len(learner.mod))

    def internals(rows):
        got = {}
        for mods, _rest in combos:
            if mods not in got:
                got[mods] = ev.internal([learner.mod[m] for m in mods], rows)
        return got
    t_test, t_ex = internals(rows_test), internals(rows_ex)
    results = {}
    for mods1, mods2 in combos:
        for metric, suffix in DEVICE_MEASURES:
            found = ev.found_labels(t_test[mods1], t_ex[mods2], labels_ex, metric)
            key = combo_key(learner.mod, mods1, mods2, suffix)
            results['found_' + key] = found
            results['score_' + key] = found_labels_to_score(labels_test, found)
    return results


def perform_one_run(dataset, modalities, coefs, k, iter_train, iter_test, rows_train, rows_test, rows_ex,
                    labels_test, labels_ex, init_dictionary=None, kind=None, on_device=True):
    """experiment.py:158-172 on a DeviceDataset: returns (learner, results) with results['dictionary'] as stored there.
    on_device: the evaluation keeps every intermediate on the GPU (default) or goes through host arrays.  kind: 'two' = TwoModalitiesExperiment._evaluate (every comparison space), 'internal' = ThreeModalitiesExperiment's
    (internal space, single and paired modalities); default by the number of modalities, as the reference's scripts pick
    the class (samples/two_modalities.py, samples/three_modalities.py)."""
    learner = MultimodalLearner(list(modalities), [b.shape[1] for b in dataset.blocks], list(coefs), k)
    dataset.train(learner, rows_train, iter_train, init_dictionary=init_dictionary)
    results = {'train': list(rows_train), 'test': list(rows_test), 'dictionary': learner.get_dico()}
    if kind is None:
        kind = 'two' if len(learner.mod) == 2 else 'internal'
    if on_device:      # dictionary, coefficients and reconstructions stay on the GPU (DeviceEvaluation)
        ev = evaluate_on_device if kind == 'two' else evaluate_internal_on_device
    else:              # the host-array path of round 2 (every transform returns numpy arrays)
        ev = evaluate if kind == 'two' else evaluate_internal
    results.update(ev(dataset, learner, rows_test, rows_ex, labels_test, labels_ex, iter_test))
    return learner, results


def _one_sweep_job(dataset, modalities, coefs, labels, examples, k, run, iter_train, iter_test, test_ratio, seed, kind):
    """One job of the reference's sweep (samples/launcher.py:71-99: an experiment with run_mode 'single' = one random
    test_ratio split, experiment.py:131-133): seeded per (k, run), so that it does not matter which rank executes it."""
    rs = np.random.RandomState([seed, k, run])
    n_all = dataset.n_samples
    others = [i for i in range(n_all) if i not in set(examples)]
    perm = rs.permutation(len(others))
    n_test = max(1, int(round(test_ratio * len(others))))
    test = [others[i] for i in perm[:n_test]]
    train = [others[i] for i in perm[n_test:]]
    f = sum(b.shape[1] for b in dataset.blocks)
    H0 = rs.random_sample((k, f)) + .01                      # the init rule of nmf.py:149-151 from the job's own stream
    H0 /= 1e-16 + H0.sum(axis=1, keepdims=True)
    _, res = perform_one_run(dataset, modalities, coefs, k, iter_train, iter_test, train, test, list(examples),
                             [labels[t] for t in test], [labels[e] for e in examples], init_dictionary=H0, kind=kind)
    return {kk: float(v) for kk, v in res.items() if kk.startswith('score_')}


def _sweep_worker(job):
    """One process per GPU: uploads the modalities once, runs its share of the (k, run) grid, returns the scores."""
    (device, rank, world, data, modalities, coefs, labels, examples, ks, n_runs, iter_train, iter_test, test_ratio, seed, kind,
     precision) = job
    import os
    import torch
    saved = {name: os.environ.get(name) for name in ('KLNMF_PRECISION', 'KLNMF_DEVICE')}
    try:       # (the learner's NMF objects read their mode and device from the environment, as experiment.py's would)
        if precision is not None:
            os.environ['KLNMF_PRECISION'] = precision
        os.environ['KLNMF_DEVICE'] = str(device)
        torch.cuda.set_device(device)
        from .device_data import DeviceDataset
        ds = DeviceDataset(data, device=device)
        out = []
        for k, run in sweep_assignment(ks, n_runs, rank, world):
            out.append((k, run, _one_sweep_job(ds, modalities, coefs, labels, examples, k, run, iter_train, iter_test,
                                               test_ratio, seed, kind)))
        return out
    finally:   # a one-device sweep runs in the caller's process: leave its environment as it was
        for name, value in saved.items():
            if value is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = value


def run_sweep(data_matrices, labels, modalities, ks, n_runs, iter_train=50, iter_test=50, coefs=None, examples=None,
              test_ratio=.1, seed=0, devices=None, precision=None, kind=None):
    """The k sweep of the reference's launcher (samples/launcher.py:68-99, 122-126: Ks x N_RUN independent experiments,
    one OS process each) on the GPUs of this node: ONE process per device, each with the modalities resident on its GPU,
    executing `sweep_assignment`'s share of the (k, run) grid -- replica parallelism, no communication.  Returns
    (table, raw): table[k][score key] = (mean, std) over the runs -- what `Logger.merge_experiments(...).get_stats(key)`
    gives the launcher's plots (launcher.py:137-149) -- and raw = [(k, run, {score key: value})].

    data_matrices: one [n_samples, d_m] array per modality, samples paired across modalities; labels: one per sample;
    examples: the row of the example of each label (default: the first sample of every label, evaluation happens on the
    rest); coefs: per-modality coefficients (default 1 / mean row sum, experiment.py:70-72); devices: GPU ordinals
    (default: all visible; the same ordinal may be listed more than once)."""
    import torch
    data = [np.asarray(m.toarray() if hasattr(m, 'toarray') else m) for m in data_matrices]
    labels = [int(v) for v in labels]
    if coefs is None:
        coefs = [float(1. / np.average(x.sum(axis=1))) for x in data]
    if examples is None:
        examples = [labels.index(l) for l in sorted(set(labels))]
    if devices is None:
        devices = list(range(max(1, torch.cuda.device_count())))
    world = len(devices)
    jobs = [(dev, r, world, data, list(modalities), list(coefs), labels, list(examples), list(ks), int(n_runs), iter_train,
             iter_test, test_ratio, seed, kind, precision) for r, dev in enumerate(devices)]
    if world == 1:
        parts = [_sweep_worker(jobs[0])]
    else:
        import multiprocessing as mp
        with mp.get_context('spawn').Pool(world) as pool:       # fresh processes: each initialises HIP on its own device
            parts = pool.map(_sweep_worker, jobs)
    raw = sorted((k, run, sc) for part in parts for (k, run, sc) in part)
    table = {}
    for k in ks:
        runs = [sc for kk, _, sc in raw if kk == k]
        table[k] = {key: (float(np.mean([r[key] for r in runs])), float(np.std([r[key] for r in runs]))) for key in runs[0]}
    return table, raw


def sweep_assignment(ks, n_runs, rank=0, world_size=1):
    """(k, run) pairs of a sweep that rank `rank` of `world_size` executes: round robin over the flattened grid, so that
    the expensive large-k fits spread evenly.  Every pair is assigned to exactly one rank."""
    grid = [(k, r) for k in ks for r in range(n_runs)]
    return grid[rank::world_size]
