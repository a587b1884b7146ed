"""One whole run of the reference's experiment driver, on device-resident data, against fixture G13 -- the outputs of
the REFERENCE's TwoModalitiesExperiment.run() (tests/golden/make_golden_experiment.py): per run the trained
dictionary and every found-label list and score of `_evaluate` (48 keys: 2 x 2 modality pairs x 3 comparison spaces
x 4 measures)."""
import numpy as np
import pytest

from tests import golden_inputs as gi
from multimodal_amd.device_experiment import perform_one_run, sweep_assignment


def test_sweep_assignment_covers_the_grid_once():
    ks, runs = [5, 10, 15, 20, 30, 40, 50, 75, 100, 200], 20        # samples/launcher.py:68-69
    per_rank = [sweep_assignment(ks, runs, r, 8) for r in range(8)]
    flat = [p for part in per_rank for p in part]
    assert sorted(flat) == sorted((k, r) for k in ks for r in range(runs))
    assert max(len(p) for p in per_rank) - min(len(p) for p in per_rank) <= 1
    work = [sum(k for k, _ in part) for part in per_rank]             # cost of a fit ~ k: spread evenly
    assert max(work) <= 1.2 * min(work)


def _ordered_data(g, dims=None):
    mods = gi.experiment_modalities(int(g['seed'])) if dims is None else gi.experiment_modalities(int(g['seed']), dims=dims)
    pair = np.asarray(g['sample_pairing'])
    ordered = [X[pair[:, m]] for m, (X, _) in enumerate(mods)]
    examples = [int(i) for i in g['examples']]
    others = [i for i in range(pair.shape[0]) if i not in examples]
    # the experiment's `data` (others) first, its `data_ex` (examples) after them: one device-resident matrix per modality
    return [np.vstack([x[others], x[examples]]) for x in ordered], len(others), len(examples)


@pytest.mark.gpu
@pytest.mark.parametrize('on_device', [True, False])
@pytest.mark.parametrize('precision', ['f64', 'f16'])
def test_one_run_equals_the_reference_experiment(monkeypatch, precision, on_device):
    """on_device: dictionary, coefficients and reconstructions stay on the GPU between the transforms and the distance kernel
    (DeviceEvaluation); False: the host-array path.  Same fixture, same bar."""
    from multimodal_amd.device_data import DeviceDataset
    monkeypatch.setenv('KLNMF_PRECISION', precision)
    g = gi.load('g13_experiment')
    data, n_others, n_ex = _ordered_data(g)
    ds = DeviceDataset(data)
    keys = [str(k) for k in g['keys']]
    labels, labels_ex = [int(v) for v in g['labels']], [int(v) for v in g['labels_ex']]
    rows_ex = list(range(n_others, n_others + n_ex))
    flips = 0
    for r in range(int(g['n_runs'])):
        train, test = [int(i) for i in g['run%d_train' % r]], [int(i) for i in g['run%d_test' % r]]
        learner, res = perform_one_run(ds, [str(m) for m in g['modalities']], [float(c) for c in g['coefs']], int(g['k']),
                                       int(g['iter_train']), int(g['iter_test']), train, test, rows_ex,
                                       [labels[t] for t in test], labels_ex, init_dictionary=g['run%d_H0' % r],
                                       on_device=on_device)
        dico_ref = g['run%d_dictionary' % r]
        if precision == 'f64':
            # (round 2 kept only fp32 copies on the device: 2e-5; the f64 mode now reads a float64 device copy)
            np.testing.assert_allclose(res['dictionary'], dico_ref, rtol=1e-9, atol=1e-13)
        else:
            assert np.abs(res['dictionary'] - dico_ref).max() <= 5e-3 * np.abs(dico_ref).max()
        found = np.array([res[k] for k in keys])
        scores = np.array([res[k.replace('found_', 'score_', 1)] for k in keys])
        ref_found, ref_scores = g['run%d_found' % r], g['run%d_scores' % r]
        if precision == 'f64':
            np.testing.assert_array_equal(found, ref_found)          # every nearest-example decision of the reference
            np.testing.assert_allclose(scores, ref_scores, rtol=0, atol=1e-12)
        else:
            per_key = (found != ref_found).sum(axis=1)
            flips += int(per_key.sum())
            # the fp16-operand mode moves coefficients and reconstructions by ~5e-4 of their maximum: nearest-example
            # decisions between near-equidistant examples can flip, mostly under the KL measures on reconstructed
            # entries close to zero.  Most keys: no flip at all; no key more than 8 of the 44 decisions.
            assert np.median(per_key) == 0 and (per_key <= 1).mean() >= 0.85 and per_key.max() <= 8, \
                sorted(zip(per_key.tolist(), keys), reverse=True)[:5]
    if precision != 'f64':
        assert flips <= 0.02 * sum(g['run%d_found' % r].size for r in range(int(g['n_runs'])))


def test_tested_combinations_follow_the_reference_order():
    from multimodal_amd.device_experiment import tested_combinations, combo_key
    g = gi.load('g15_experiment3')
    mods = [str(m) for m in g['modalities']]
    combos = tested_combinations(3)
    assert [' '.join(map(str, a)) + '>' + ' '.join(map(str, b)) for a, b in combos] == [str(c) for c in g['combos']]
    keys = ['score_' + combo_key(mods, a, b, sfx) for a, b in combos for sfx in ('', '_bis', '_frob', '_cosine')]
    assert keys == [str(k) for k in g['keys']]


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f64', 'f16'])
def test_three_modalities_run_equals_the_reference_experiment(monkeypatch, precision):
    """Fixture G15 = the REFERENCE's ThreeModalitiesExperiment.run() (experiment.py:322-404): per run the trained dictionary
    and, for the 12 tested combinations (single -> single, pair -> single, single -> pair) x 4 measures, every found label
    and score in the internal space."""
    from multimodal_amd.device_data import DeviceDataset
    monkeypatch.setenv('KLNMF_PRECISION', precision)
    g = gi.load('g15_experiment3')
    data, n_others, n_ex = _ordered_data(g, dims=(48, 30, 22))
    ds = DeviceDataset(data)
    keys = [str(k) for k in g['keys']]
    labels, labels_ex = [int(v) for v in g['labels']], [int(v) for v in g['labels_ex']]
    rows_ex = list(range(n_others, n_others + n_ex))
    total = flips = 0
    for r in range(int(g['n_runs'])):
        train, test = [int(i) for i in g['run%d_train' % r]], [int(i) for i in g['run%d_test' % r]]
        learner, res = perform_one_run(ds, [str(m) for m in g['modalities']], [float(c) for c in g['coefs']], int(g['k']),
                                       int(g['iter_train']), int(g['iter_test']), train, test, rows_ex,
                                       [labels[t] for t in test], labels_ex, init_dictionary=g['run%d_H0' % r])
        dico_ref = g['run%d_dictionary' % r]
        found = np.array([res[k.replace('score_', 'found_', 1)] for k in keys])
        scores = np.array([res[k] for k in keys])
        ref_found, ref_scores = g['run%d_found' % r], g['run%d_scores' % r]
        if precision == 'f64':
            np.testing.assert_allclose(res['dictionary'], dico_ref, rtol=1e-9, atol=1e-13)
            np.testing.assert_array_equal(found, ref_found)
            np.testing.assert_allclose(scores, ref_scores, rtol=0, atol=1e-12)
        else:
            # 126 training rows x 100 columns: the fp16 operands' rounding is averaged over very few rows (measured 1.2e-2 of
            # the dictionary's maximum after the 40 training iterations; 5e-3 on the two-modality fixture)
            assert np.abs(res['dictionary'] - dico_ref).max() <= 2e-2 * np.abs(dico_ref).max()
            per_key = (found != ref_found).sum(axis=1)
            flips += int(per_key.sum())
            total += found.size
            assert np.median(per_key) == 0 and per_key.max() <= 8
    if precision != 'f64':
        assert flips <= 0.02 * total


@pytest.mark.gpu
def test_run_sweep_over_devices_is_partition_independent():
    """`run_sweep` = the launcher's k sweep (samples/launcher.py:68-99) as one process per device.  The jobs are seeded per
    (k, run): executed by one process, or split over two worker processes (both on device 0 here: one GPU), the table is
    the same; and it has the launcher's shape -- per k, per score key, (mean, std) over the runs."""
    from multimodal_amd.device_experiment import run_sweep
    g = gi.load('g13_experiment')
    data, n_others, n_ex = _ordered_data(g)
    labels = [int(v) for v in g['labels']] + [int(v) for v in g['labels_ex']]
    examples = list(range(n_others, n_others + n_ex))
    kw = dict(iter_train=20, iter_test=10, examples=examples, seed=3, precision='f64')
    mods = [str(m) for m in g['modalities']]
    t1, raw1 = run_sweep(data, labels, mods, [4, 8], 2, devices=[0], **kw)
    t2, raw2 = run_sweep(data, labels, mods, [4, 8], 2, devices=[0, 0], **kw)
    assert sorted(t1) == [4, 8] and len(raw1) == 4 and [(k, r) for k, r, _ in raw1] == [(4, 0), (4, 1), (8, 0), (8, 1)]
    keys = sorted(t1[4])
    assert len(keys) == 48 and all(k.startswith('score_') for k in keys) and 'score_motion2sound' in keys
    for k in (4, 8):
        for key in keys:
            assert t1[k][key] == t2[k][key]
            assert 0.0 <= t1[k][key][0] <= 1.0 and t1[k][key][1] >= 0.0
    assert np.mean([t1[8][key][0] for key in keys]) > 0.3          # it does classify (chance = 0.1)
