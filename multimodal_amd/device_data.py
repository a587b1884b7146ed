# encoding: utf-8
"""Experiment data resident on the GPU across runs (next-row N2 of SURVEY.md 8f).

The reference's experiment loop slices every modality on the host for each run
(`data_train = [x[train, :] for x in self.data]`, experiment.py:163-164) and the learner re-uploads
what it is given.  Here the modalities are uploaded once (fp32); a run hands over row indices, and
the rows are gathered by the tiling upload kernel itself (`klnmf_upload_V_device_rows`), together
with the per-modality coefficient and the column placement (learner.py:53-56).
"""
import numpy as np

from . import _native
from .lib.nmf import KLdivNMF, check_non_negative
from .lib.sklearn_utils import atleast2d_or_csr


class DeviceDataset(object):
    """modalities: list of [n_samples, d_m] arrays (dense or scipy sparse), kept as fp32 on `device`."""

    def __init__(self, data_matrices, device=None):
        import torch
        self.torch = torch
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.blocks = []
        self.maxima = []
        self.host = []           # the caller's matrices (for comparisons on the raw data, experiment.py:266: no transformation)
        for m in data_matrices:
            m = atleast2d_or_csr(m)
            check_non_negative(m, "NMF.fit")
            if hasattr(m, 'toarray'):
                m = m.toarray()
            self.host.append(np.asarray(m))
            t = torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(self.device)
            self.blocks.append(t)
            self.maxima.append(float(t.max().item()) if t.numel() else 0.0)
        self.n_samples = self.blocks[0].shape[0]
        assert all(b.shape[0] == self.n_samples for b in self.blocks)

    def _uploader(self, which, rows, coefs):
        torch = self.torch
        idx = torch.as_tensor(np.asarray(rows, dtype=np.int64), device=self.device)
        assert idx.numel() == 0 or (int(idx.min()) >= 0 and int(idx.max()) < self.n_samples)

        def upload(ctx):
            # an upper bound of the stacked maximum fixes the 16-bit storage factor (any bound is valid)
            ctx.set_v_max(max([c * self.maxima[w] for w, c in zip(which, coefs)] + [0.0]))
            col = 0
            for w, c in zip(which, coefs):
                b = self.blocks[w]
                ctx.upload_V_device_rows(b.data_ptr(), idx.data_ptr(), idx.numel(), b.shape[1], b.stride(0),
                                         row0=0, col0=col, scale=c)
                col += b.shape[1]
            torch.cuda.synchronize(self.device)      # the context runs on its own stream; idx must outlive the kernel
        return upload, idx.numel()

    # ---- what experiment.py:_perform_one_run does with the sliced copies ----
    def rows_of(self, which, rows):
        """Host rows of modality `which` as the caller gave them (what experiment.py compares raw data with)."""
        return self.host[which][np.asarray(rows, dtype=np.int64), :]

    def train(self, learner, rows, iterations, init_dictionary=None):
        """learner.train([x[rows] for x in data], iterations) (learner.py:31-41) without the host slices.
        `init_dictionary`: the initial dictionary instead of a draw from the global numpy stream (nmf.py:149-155)."""
        if learner.sparseness is not None:
            raise NotImplemented
        which = list(range(len(self.blocks)))
        assert [b.shape[1] for b in self.blocks] == list(learner.dim)
        upload, n = self._uploader(which, rows, list(learner.coef))
        nmf = KLdivNMF(n_components=learner.k, max_iter=iterations, tol=0)
        if init_dictionary is not None:
            nmf._init_dictionary = np.asarray(init_dictionary)
        nmf._fit_uploaded(n, sum(learner.dim), upload, lambda H: np.float64, _fit=True)
        learner.nmf_train = nmf
        learner.dico = nmf.components_
        return learner

    def reconstruct_internal_multi(self, learner, orig_mods, rows, iterations):
        """learner.reconstruct_internal_multi(orig_mods, [x[rows] ...], iterations) (learner.py:71-78)."""
        which = [learner.get_index(m) for m in orig_mods]
        coefs = [learner.coef[w] for w in which]
        upload, n = self._uploader(which, rows, coefs)
        dico = learner.get_stacked_dicos(orig_mods)
        nmf = KLdivNMF(n_components=dico.shape[0], max_iter=iterations, tol=0)
        nmf.components_ = dico
        nmf._init_dictionary = dico
        return nmf._fit_uploaded(n, dico.shape[1], upload, lambda H: np.float64, _fit=False)

    def reconstruct_internal(self, learner, orig_mod, rows, iterations):
        return self.reconstruct_internal_multi(learner, [orig_mod], rows, iterations)
