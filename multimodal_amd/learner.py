# encoding: utf-8
"""Multimodal dictionary learning on top of the GPU KL-NMF.

Interface-compatible with reference multimodal/learner.py (`MultimodalLearner`
with attributes mod/dim/coef/k/dico/nmf_train and its twelve methods, plus
`fit_coefficients`), so `experiment.py` and the sample scripts can drive it
unchanged.  What differs is where the work happens: the per-modality blocks and
their balancing coefficients go straight to the NMF upload, where scale, cast
and column placement are one kernel per block, instead of first materialising
`safe_hstack([c * m ...])` on the host (reference learner.py:53-56).
"""
from .lib.nmf import KLdivNMF as NMF
from .lib.nmf import check_non_negative
from .lib.array_utils import safe_hstack
from .lib.sklearn_utils import atleast2d_or_csr
from . import _native
from .lib.nmf import _default_device


def _checked(blocks):
    """Apply the input contract of NMF.fit to every modality block."""
    checked = []
    for block in blocks:
        block = atleast2d_or_csr(block)
        check_non_negative(block, "NMF.fit")
        checked.append(block)
    return checked


def _coefficients_on(dictionary, blocks, scales, iter_nmf):
    """tol=0 transform of the (virtually) stacked blocks on a fixed dictionary."""
    model = NMF(n_components=dictionary.shape[0], max_iter=iter_nmf, tol=0)
    model.components_ = dictionary
    return model._transform_blocks(_checked(blocks), scales)


def fit_coefficients(data_obs, dictionary, iter_nmf=100, verbose=False):
    """Non-negative coefficients of `data_obs` on `dictionary`
    (reference learner.py:11-15: tol=0 transform for iter_nmf iterations)."""
    return _coefficients_on(dictionary, [data_obs], [1.], iter_nmf)


class MultimodalLearner(object):
    """Holds one dictionary over several column-stacked modalities.

    modalities: names; dimensions: number of columns of each; coefficients:
    per-modality scale applied before stacking; k: number of atoms.
    `dico` ([k, sum(dimensions)] ndarray, None until trained) may also be
    assigned from outside (the sample scripts do).
    """

    def __init__(self, modalities, dimensions, coefficients, k,
                 sparseness=None, sp_coef=.1):
        self.mod = modalities
        self.dim = dimensions
        self.coef = coefficients
        self.k = k
        self.sparseness = sparseness
        self.sp_coef = sp_coef
        self.dico = None

    # ---- bookkeeping ----
    def get_index(self, modality):
        return self.mod.index(modality)

    def get_axis_range(self, modality):
        """(start, stop) columns of a modality in the stacked matrix
        (reference learner.py:58-62)."""
        idx = self.get_index(modality)
        start = sum(self.dim[:idx])
        return (start, start + self.dim[idx])

    def _scales(self, modalities):
        return [self.coef[self.get_index(m)] for m in modalities]

    def stack_data(self, modalities, data_matrices):
        """Host-side stacked matrix hstack(c_m * X_m) (reference learner.py:53-56).
        Kept for callers; train/reconstruct do not need it."""
        return safe_hstack([c * m for m, c in
                            zip(data_matrices, self._scales(modalities))])

    # ---- dictionary ----
    def train(self, data_matrices, iterations):
        """Fit the dictionary on all modalities: `iterations` multiplicative
        updates with tol=0 (reference learner.py:31-41)."""
        n_samples = data_matrices[0].shape[0]
        for m, d in zip(data_matrices, self.dim):
            assert(m.shape == (n_samples, d))
        if self.sparseness is not None:
            raise NotImplemented        # as the reference (learner.py:37-38)
        self.nmf_train = NMF(n_components=self.k, max_iter=iterations, tol=0)
        self.nmf_train._fit_blocks(_checked(data_matrices), self._scales(self.mod),
                                   _fit=True)
        self.dico = self.nmf_train.components_

    def get_dico(self, modality=None):
        """Whole dictionary, or the column slice (a view) of one modality."""
        if modality is None:
            return self.dico
        start, stop = self.get_axis_range(modality)
        return self.dico[:, start:stop]

    def get_stacked_dicos(self, modalities):
        return safe_hstack([self.get_dico(modality=m) for m in modalities])

    # ---- inference ----
    def reconstruct_internal_multi(self, orig_mods, test_data, iterations):
        """Internal coefficients from a subset of modalities
        (reference learner.py:71-78)."""
        for mod, data in zip(orig_mods, test_data):
            assert(data.shape[1] == self.dim[self.get_index(mod)])
        return _coefficients_on(self.get_stacked_dicos(orig_mods), test_data,
                                self._scales(orig_mods), iterations)

    def reconstruct_internal(self, orig_mod, test_data, iterations):
        return self.reconstruct_internal_multi([orig_mod], [test_data],
                                               iterations)

    def reconstruct_modalities(self, dest_mods, internal):
        """internal . stacked dictionaries of dest_mods (reference learner.py:83-84),
        on the device (klnmf_matmul), in the operands' arithmetic."""
        return _native.matmul(internal, self.get_stacked_dicos(dest_mods), _default_device())

    def reconstruct_modality(self, dest_mod, internal):
        """internal . dictionary of dest_mod (reference learner.py:80-81)."""
        return _native.matmul(internal, self.get_dico(dest_mod), _default_device())

    def modalities_to_modalities(self, orig_mods, dest_mods, test_data,
                                 iterations):
        internal = self.reconstruct_internal_multi(orig_mods, test_data,
                                                   iterations)
        return self.reconstruct_modalities(dest_mods, internal)

    def modality_to_modality(self, orig_mod, dest_mod, test_data, iterations):
        return self.modalities_to_modalities([orig_mod], [dest_mod],
                                             [test_data], iterations)
