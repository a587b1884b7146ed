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
from .lib.nmf import KLdivNMF, check_non_negative, _default_precision, resolve_precision
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
        self._blocks64 = {}

    def source(self, which, precision=None):
        """(device matrix, is_float64) the fits and transforms read modality `which` from: the float64 copy when the NMF runs
        in the reference's own arithmetic (f64: results then agree with the reference to summation order), the float32 one
        otherwise (the 16-bit modes store V in 16 bits anyway).  `precision`: what the context that reads it resolved to
        (`Context.precision_name`); None: the process default."""
        if precision is None:
            precision = _default_precision()
        if precision == 'auto' or _native.PRECISIONS[precision] == _native.PREC_F64:
            return self.block64(which), True           # ('auto' decides per fit: the float64 copy serves either outcome)
        return self.blocks[which], False

    def block64(self, which):
        """float64 device copy of modality `which` (made on first use): the evaluation compares raw rows with
        reconstructions in the caller's own precision (experiment.py:266)."""
        if which not in self._blocks64:
            self._blocks64[which] = self.torch.from_numpy(np.ascontiguousarray(self.host[which], dtype=np.float64)).to(self.device)
        return self._blocks64[which]

    def _uploader(self, which, rows, coefs):
        torch = self.torch
        idx = torch.as_tensor(np.asarray(rows, dtype=np.int64), device=self.device)
        assert idx.numel() == 0 or (int(idx.min()) >= 0 and int(idx.max()) < self.n_samples)

        def upload(ctx):
            # an upper bound of the stacked maximum fixes the 16-bit storage factor (any bound is valid)
            ctx.set_v_max(max([c * self.maxima[w] for w, c in zip(which, coefs)] + [0.0]))
            col = 0
            for w, c in zip(which, coefs):
                b, f64 = self.source(w, getattr(ctx, 'precision_name', None))
                ctx.upload_V_device_rows_dt(b.data_ptr(), f64, idx.data_ptr(), idx.numel(), b.shape[1], b.stride(0),
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


class DeviceEvaluation(object):
    """What one run's evaluation needs of a trained learner, kept on the GPU (next-row N1; experiment.py:233-277, 332-371):

      * the dictionary is uploaded ONCE and every transform takes its column blocks from there (klnmf_set_H_device;
        get_dico / get_stacked_dicos, learner.py:43-51);
      * the test / example rows are gathered from the device-resident modalities by the upload kernel;
      * the coefficients stay on the device (klnmf_get_W_device), the reconstructions are products of device matrices
        (klnmf_matmul_device; learner.py:80-84), and the nearest-example search reads both sides from device memory
        (klnmf_all_distances_device) -- only the [n_test, n_examples] distance matrix comes back.

    All intermediates are float64 (as the host path's); the loop itself runs in the learner's NMF precision."""

    def __init__(self, dataset, learner, iter_test):
        import torch
        self.torch, self.ds, self.learner, self.iter_test = torch, dataset, learner, int(iter_test)
        self.dev = dataset.device
        self.dico = torch.from_numpy(np.ascontiguousarray(learner.get_dico(), dtype=np.float64)).to(self.dev)
        self.k, self.F = self.dico.shape
        self.offsets = [sum(learner.dim[:i]) for i in range(len(learner.dim))]

    def _rows(self, rows):
        idx = self.torch.as_tensor(np.asarray(rows, dtype=np.int64), device=self.dev)
        assert idx.numel() == 0 or (int(idx.min()) >= 0 and int(idx.max()) < self.ds.n_samples)
        return idx

    def internal(self, mods, rows):
        """learner.reconstruct_internal_multi(mods, [x[rows] ...], iter_test) -> device tensor [len(rows), k]."""
        torch, lr = self.torch, self.learner
        which = [lr.get_index(m) for m in mods]
        idx = self._rows(rows)
        n, f = int(idx.numel()), sum(lr.dim[w] for w in which)
        out = torch.empty((n, self.k), dtype=torch.float64, device=self.dev)
        model = KLdivNMF(n_components=self.k, max_iter=self.iter_test, tol=0)
        # (the shape decides the arithmetic exactly as the host path's _fit_uploaded does: 'auto' by size, k beyond the MFMA
        # kernels' range on the fp32 kernels)
        with model._context(shape=(n, f, self.k)) as ctx:
            ctx.set_problem(n, f, self.k, self.iter_test)
            ctx.set_v_max(max([lr.coef[w] * self.ds.maxima[w] for w in which] + [0.0]))
            col = 0
            for w in which:
                b, f64 = self.ds.source(w, getattr(ctx, 'precision_name', None))
                ctx.upload_V_device_rows_dt(b.data_ptr(), f64, idx.data_ptr(), n, b.shape[1], b.stride(0), row0=0, col0=col,
                                            scale=lr.coef[w])
                col += b.shape[1]
            col = 0
            for i, w in enumerate(which):                  # the stacked dictionary of these modalities, block by block
                d = lr.dim[w]
                ctx.set_H_device(self.dico.data_ptr() + 8 * self.offsets[w], True, self.F, col, d, last=(i == len(which) - 1))
                col += d
            ctx.init_W()                                   # W0 = X . H^T with the dictionary itself (nmf.py:156, 283)
            ctx.run(self.iter_test, False, 0.0)
            ctx.get_W_device(out.data_ptr(), True, self.k)
        torch.cuda.synchronize(self.dev)
        return out

    def reconstruct(self, internal, dest_mod):
        """learner.reconstruct_modality(dest_mod, internal) (learner.py:80-81) between device matrices."""
        w = self.learner.get_index(dest_mod)
        d = self.learner.dim[w]
        out = self.torch.empty((internal.shape[0], d), dtype=self.torch.float64, device=self.dev)
        _native.matmul_device(internal.data_ptr(), internal.stride(0), self.dico.data_ptr() + 8 * self.offsets[w], self.F,
                              out.data_ptr(), d, internal.shape[0], d, self.k, f64=True, device=self.dev.index or 0)
        return out

    def raw(self, which, rows):
        """The rows of modality `which` as the experiment compares raw data (float64 device copy of the caller's matrix)."""
        return self.ds.block64(which).index_select(0, self._rows(rows))

    def found_labels(self, test, examples, labels_ex, metric):
        """classify_NN (evaluation.py:109-116): nearest example's label for every row of `test` -- both on the device."""
        out = self.torch.empty((test.shape[0], examples.shape[0]), dtype=self.torch.float64, device=self.dev)
        _native.all_distances_device(test.data_ptr(), test.stride(0), examples.data_ptr(), examples.stride(0), out.data_ptr(),
                                     test.shape[0], examples.shape[0], test.shape[1], metric, f64=True,
                                     device=self.dev.index or 0)
        nearest = np.argmin(out.cpu().numpy(), axis=1)
        return [labels_ex[j] for j in nearest]
