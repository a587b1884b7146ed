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

from .evaluation import classify_NN, found_labels_to_score
from .learner import MultimodalLearner
from .lib.metrics import kl_div, rev_kl_div, frobenius, cosine_diff

INTERNAL = -1
MEASURES = ((kl_div, ''), (rev_kl_div, '_bis'), (frobenius, '_frob'), (cosine_diff, '_cosine'))


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


def perform_one_run(dataset, modalities, coefs, k, iter_train, iter_test, rows_train, rows_test, rows_ex,
                    labels_test, labels_ex, init_dictionary=None):
    """experiment.py:158-172 on a DeviceDataset: returns (learner, results) with results['dictionary'] as stored there."""
    learner = MultimodalLearner(list(modalities), [b.shape[1] for b in dataset.blocks], list(coefs), k)
    dataset.train(learner, rows_train, iter_train, init_dictionary=init_dictionary)
    results = {'train': list(rows_train), 'test': list(rows_test), 'dictionary': learner.get_dico()}
    results.update(evaluate(dataset, learner, rows_test, rows_ex, labels_test, labels_ex, iter_test))
    return learner, results


def sweep_assignment(ks, n_runs, rank=0, world_size=1):
    """(k, run) pairs of a sweep that rank `rank` of `world_size` executes: round robin over the flattened grid, so that
    the expensive large-k fits spread evenly.  Every pair is assigned to exactly one rank."""
    grid = [(k, r) for k in ks for r in range(n_runs)]
    return grid[rank::world_size]
