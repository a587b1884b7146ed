# encoding: utf-8
"""Nearest-neighbour evaluation on the GPU (next-row N4; reference multimodal/evaluation.py:70-140).

`all_distances` is the one heavy operation there: every reconstructed sample against every example, for
one of the measures of `lib/metrics.py`.  The reference broadcasts [n_test, 1, d] x [1, n_ex, d] through
numpy; here it is one kernel launch (`klnmf_all_distances`).  The label bookkeeping around it is index
arithmetic and stays on the host.
"""
import numpy as np
import scipy.sparse as sp

from .lib import metrics

_METRIC_OF = {'kl_div': metrics.kl_div, 'rev_kl_div': metrics.rev_kl_div, 'sym_kl_div': metrics.sym_kl_div,
              'frobenius': metrics.frobenius, 'cosine_diff': metrics.cosine_diff}


def todense(X):
    return np.asarray(X.todense()) if sp.issparse(X) else X


def _gpu_measure(measure):
    """The GPU implementation of `measure`: one of this package's measures, or the reference's function of
    the same name (so that `experiment.py` can keep importing its own `lib.metrics`)."""
    name = getattr(measure, '__name__', None)
    if name not in _METRIC_OF:
        raise ValueError("measure %r has no GPU implementation (known: %s)" % (name, ', '.join(sorted(_METRIC_OF))))
    return _METRIC_OF[name]


def all_distances(reco_data, ex_data, measure):
    """[len(reco_data), len(ex_data)] matrix of measure(reco, example) (reference evaluation.py:103-106)."""
    reco = np.asarray(todense(reco_data))[:, np.newaxis, :]
    ex = np.asarray(todense(ex_data))[np.newaxis, :, :]
    return _gpu_measure(measure)(reco, ex, axis=-1)


def dists_to_found_labels(dists, ex_labels):
    """Label of the nearest example of every row (reference evaluation.py:74-77)."""
    nearest = np.argmin(dists, axis=1)
    return [ex_labels[j] for j in nearest]


def found_labels_to_score(true, found):
    """Fraction of matching labels (reference evaluation.py:80-83)."""
    return np.average([f == t for f, t in zip(found, true)])


def found_labels_to_confusion(true, found, n_labels):
    """conf[i, j] = 1 where label i was classified as j at least once (reference evaluation.py:86-92: the
    fancy-index `+=` there does not accumulate repeated pairs, and neither does this)."""
    conf = np.zeros((n_labels, n_labels))
    conf[true, found] += 1
    return conf


def classify_NN(reco_data, ex_data, ex_labels, measure):
    """Nearest example's label for every reconstructed sample (reference evaluation.py:109-116)."""
    return dists_to_found_labels(all_distances(reco_data, ex_data, measure), ex_labels)


def scores_from_dists(dists, true_labels_0, true_labels_1=None, verbose=False):
    """Deprecated in the reference too (evaluation.py:61-71)."""
    if true_labels_1 is None:
        assert(dists.shape[0] == dists.shape[1])
        true_labels_1 = true_labels_0
    found = dists_to_found_labels(dists, true_labels_1)
    result = found_labels_to_score(true_labels_0, found)
    if verbose:
        print(result)
    return result


def evaluate_NN_label(reco_data, test_data, true_labels, test_labels, measure):
    """Score of nearest-neighbour labelling against `test_data` (reference evaluation.py:119-131)."""
    return scores_from_dists(all_distances(reco_data, test_data, measure), true_labels, test_labels)
