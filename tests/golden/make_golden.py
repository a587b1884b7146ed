#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

It imports omangin/multimodal from /root/reference (with the one-line
`np.Inf` shim the reference needs on NumPy >= 2, see reference
multimodal/lib/nmf.py:206) and writes small .npz fixtures next to this file.
Inputs are regenerated from `np.random.RandomState(seed)` (the legacy stream is
version-stable), so the fixtures hold seeds/shapes plus the reference OUTPUTS.
Nothing of the reference's source travels: only data.

Fixture families:
  G1 fit dense fp64            G2 transform (trained H, column-sliced H)
  G3 single-step rules         G4 default-tol early stop + warning case
  G5 learner (2 and 3 modalities)   G6 CSR input (error/_Q/one update)
  G7 float32 run               G8 error paths + known-answer helper values
  G9 CSR input: full fit, transform, default-tol stop (the reference's sparse branch)
  G10 nearest-neighbour evaluation: pairwise distances for 5 measures, classify_NN labels
"""
import io
import os
import sys
import contextlib

sys.dont_write_bytecode = True
import numpy as np

np.Inf = np.inf  # NumPy 2 removed the alias used at reference nmf.py:206
sys.path.insert(0, '/root/reference')

import scipy.sparse as sp  # noqa: E402
from multimodal.lib import nmf as rnmf  # noqa: E402
from multimodal.lib.metrics import generalized_KL  # noqa: E402
from multimodal.lib.array_utils import normalize_sum  # noqa: E402
from multimodal import learner as rlearner  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def gen_inputs(seed, n, f, k):
    """Seeded test problem shared with the tests (tests/golden_inputs.py)."""
    rs = np.random.RandomState(seed)
    X = np.abs(rs.random_sample((n, f)))
    H0 = normalize_sum(np.abs(rs.random_sample((k, f))) + .01, axis=1)
    return X, H0


def ref_fit(X, H0, k, max_iter, tol):
    m = rnmf.KLdivNMF(n_components=k, max_iter=max_iter, tol=tol)
    m._init_dictionary = H0.copy()
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        W, errors = m.fit_transform(X, return_errors=True, scale_W=True)
    return m, W, np.array(errors, dtype=np.float64), err.getvalue()


def save(name, **kw):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **kw)
    print('wrote', path, {k: np.shape(v) for k, v in kw.items()})


def g1():
    for tag, seed, n, f, k, iters in [
            ('g1_20x30_k3', 11, 20, 30, 3, (1, 5, 50)),
            ('g1_37x53_k7', 12, 37, 53, 7, (1, 5, 50)),
            ('g1_500x1000_k10', 13, 500, 1000, 10, (50,))]:
        X, H0 = gen_inputs(seed, n, f, k)
        out = dict(seed=seed, n=n, f=f, k=k, iters=np.array(iters))
        for it in iters:
            m, W, errors, _ = ref_fit(X, H0, k, it, 0)
            out['W_%d' % it] = W
            out['H_%d' % it] = m.components_
            out['errors_%d' % it] = errors
            out['final_%d' % it] = m.error(X, W)
        save(tag, **out)


def g2():
    seed, n, f, k = 21, 40, 64, 5
    X, H0 = gen_inputs(seed, n, f, k)
    m, W, _, _ = ref_fit(X, H0, k, 30, 0)
    H = m.components_
    # transform new data with the trained dictionary
    Xt = np.abs(np.random.RandomState(seed + 1).random_sample((17, f)))
    t = rnmf.KLdivNMF(n_components=k, max_iter=25, tol=0)
    t.components_ = H
    Wt, et = t.transform(Xt, return_errors=True, scale_W=True)
    # column-sliced dictionary (rows no longer sum to 1), as learner does
    sl = slice(10, 42)
    t2 = rnmf.KLdivNMF(n_components=k, max_iter=25, tol=0)
    t2.components_ = H[:, sl]
    Ws, es = t2.transform(Xt[:, sl], return_errors=True, scale_W=True)
    save('g2_transform', seed=seed, n=n, f=f, k=k, H=H, W_train=W,
         Wt=Wt, errors_t=np.array(et), sl=np.array([sl.start, sl.stop]),
         Ws=Ws, errors_s=np.array(es))


def g3():
    seed, n, f, k = 31, 20, 30, 3
    rs = np.random.RandomState(seed)
    X = np.abs(rs.random_sample((n, f)))
    W = np.abs(rs.random_sample((n, k)))
    H = np.abs(rs.random_sample((k, f)))
    m = rnmf.KLdivNMF(n_components=k)
    m.components_ = H.copy()
    Q = m._Q(X, W, H)
    Wn = m._updated_W(X, W, H, Q=Q)
    Hn = m._updated_H(X, Wn, H, Q=Q)      # new W, old Q (q2)
    Hn_noq = m._updated_H(X, W, H)        # Q recomputed from the W given
    err = m.error(X, W, H=H)
    W_upd = m._update(X, W, _fit=True)
    H_upd = m.components_.copy()
    m2 = rnmf.KLdivNMF(n_components=k)
    m2.components_ = H.copy()
    W_upd_scaled = m2._update(X, W, _fit=True, scale_W=True)
    save('g3_steps', seed=seed, n=n, f=f, k=k, Q=Q, Wn=Wn, Hn=Hn,
         Hn_noq=Hn_noq, err=err, W_upd=W_upd, H_upd=H_upd,
         W_upd_scaled=W_upd_scaled, H_upd_scaled=m2.components_)


def g4():
    seed, n, f, k = 41, 10, 5, 3
    X, H0 = gen_inputs(seed, n, f, k)
    # default tolerance: stops early
    m, W, errors, msg = ref_fit(X, H0, k, 200, 1e-6)
    # loose tolerance
    m2, W2, errors2, msg2 = ref_fit(X, H0, k, 200, 1e-3)
    # iteration limit with tol > 0 -> stderr warning
    m3, W3, errors3, msg3 = ref_fit(X, H0, k, 4, 1e-9)
    # exactly factorisable data (reference tests/test_nmf_kl.py:162-165)
    rs = np.random.RandomState(seed + 5)
    Xf = np.abs(rs.random_sample((5, 2))).dot(np.abs(rs.random_sample((2, 3))))
    H0f = normalize_sum(np.abs(rs.random_sample((3, 3))) + .01, axis=1)
    m4, W4, errors4, msg4 = ref_fit(Xf, H0f, 3, 200, 1e-6)
    save('g4_tol', seed=seed, n=n, f=f, k=k,
         W=W, H=m.components_, errors=errors, warned=bool(msg),
         W2=W2, H2=m2.components_, errors2=errors2, warned2=bool(msg2),
         W3=W3, H3=m3.components_, errors3=errors3, warned3=bool(msg3),
         msg3=msg3,
         Xf=Xf, H0f=H0f, W4=W4, H4=m4.components_, errors4=errors4,
         warned4=bool(msg4))


def g5():
    for tag, seed, n, dims, k in [('g5_learner2', 51, 30, (24, 16), 4),
                                  ('g5_learner3', 52, 26, (12, 20, 9), 5)]:
        rs = np.random.RandomState(seed)
        blocks = [np.abs(rs.random_sample((n, d))) for d in dims]
        # 1/mean row-sum, as reference experiment.py:70-72
        coefs = [float(1. / np.mean(np.sum(b, axis=1))) for b in blocks]
        mods = ['m%d' % i for i in range(len(dims))]
        f = sum(dims)
        H0 = normalize_sum(np.abs(rs.random_sample((k, f))) + .01, axis=1)
        lr = rlearner.MultimodalLearner(mods, list(dims), coefs, k)
        # train() draws H0 from the global RNG; inject ours the way the
        # reference allows: patch the NMF factory for this call only.
        orig = rlearner.NMF

        def factory(**kw):
            m = orig(**kw)
            m._init_dictionary = H0.copy()
            return m
        rlearner.NMF = factory
        try:
            lr.train(blocks, 20)
        finally:
            rlearner.NMF = orig
        out = dict(seed=seed, n=n, dims=np.array(dims), k=k,
                   coefs=np.array(coefs), dico=lr.dico,
                   stacked_sum=lr.stack_data(mods, blocks).sum(axis=0))
        test = [np.abs(rs.random_sample((7, d))) for d in dims]
        for i, mname in enumerate(mods):
            out['dico_%d' % i] = lr.get_dico(mname)
            out['internal_%d' % i] = lr.reconstruct_internal(mname, test[i], 15)
        out['internal_01'] = lr.reconstruct_internal_multi(
            mods[:2], test[:2], 15)
        out['m2m_0_to_1'] = lr.modality_to_modality(mods[0], mods[1],
                                                     test[0], 15)
        save(tag, **out)


def g6():
    seed, n, f, k = 61, 20, 30, 3
    rs = np.random.RandomState(seed)
    dense = np.abs(rs.random_sample((n, f))) * (rs.random_sample((n, f)) < .5)
    X = sp.csr_matrix(dense)
    W = np.abs(rs.random_sample((n, k)))
    H = np.abs(rs.random_sample((k, f)))
    m = rnmf.KLdivNMF(n_components=k)
    m.components_ = H.copy()
    err = m.error(X, W, H=H)
    Q = m._Q(X, W, H)
    Wn = m._update(X, W, _fit=True)
    save('g6_sparse', seed=seed, n=n, f=f, k=k, err=err,
         Q_dense=np.asarray(Q.todense()), Wn=Wn, Hn=m.components_)


def g9():
    """Full fit and transform through the reference's CSR branch (nmf.py:52-70, 301-308, 331-334):
    sparse X incl. an empty row and an empty column."""
    seed, n, f, k = 91, 60, 90, 5
    rs = np.random.RandomState(seed)
    dense = np.abs(rs.random_sample((n, f))) * (rs.random_sample((n, f)) < .25)
    dense[7, :] = 0
    dense[:, 11] = 0
    H0 = normalize_sum(np.abs(rs.random_sample((k, f))) + .01, axis=1)
    X = sp.csr_matrix(dense)
    m, W, errors, _ = ref_fit(X, H0, k, 12, 0)
    Wt = m.transform(X[:20])
    m2 = rnmf.KLdivNMF(n_components=k, max_iter=300, tol=1e-4)
    m2._init_dictionary = H0.copy()
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        W2, errors2 = m2.fit_transform(X, return_errors=True)
    save('g9_sparse_fit', seed=seed, n=n, f=f, k=k, W=W, H=m.components_, errors=errors, Wt=Wt,
         errors_tol=np.array(errors2, dtype=np.float64), W_tol=W2, H_tol=m2.components_)


def g10():
    """Nearest-neighbour evaluation: all_distances for the four measures experiment.py uses, and the labels
    classify_NN finds (evaluation.py:70-116, metrics.py:58-86)."""
    from multimodal import evaluation as rev
    from multimodal.lib import metrics as rm
    seed, na, nb, d = 101, 23, 7, 50
    rs = np.random.RandomState(seed)
    A = np.abs(rs.random_sample((na, d))) * (rs.random_sample((na, d)) < .8)
    B = np.abs(rs.random_sample((nb, d)))
    A[3, :] = 0                                  # a zero vector (the cosine measure's special case)
    out = dict(seed=seed, na=na, nb=nb, d=d)
    for name in ('kl_div', 'rev_kl_div', 'sym_kl_div', 'frobenius', 'cosine_diff'):
        out[name] = rev.all_distances(A, B, getattr(rm, name))
    labels = list(range(nb))
    out['found_frobenius'] = np.array(rev.classify_NN(A, B, labels, rm.frobenius))
    out['found_cosine'] = np.array(rev.classify_NN(A, B, labels, rm.cosine_diff))
    save('g10_distances', **out)


def g7():
    seed, n, f, k = 71, 48, 80, 6
    X, H0 = gen_inputs(seed, n, f, k)
    X32, H032 = X.astype(np.float32), H0.astype(np.float32)
    m, W, errors, _ = ref_fit(X32, H032, k, 40, 0)
    assert W.dtype == np.float32
    save('g7_float32', seed=seed, n=n, f=f, k=k, W=W, H=m.components_,
         errors=errors)


def g8():
    # known-answer values the reference's own tests pin
    x = np.array([[1., 2.], [3., 4.]])
    y = np.array([[2., 2.], [1., 4.]])
    a = np.array([[1., 2., 3.], [4., 5., 6.]])
    out = dict(
        gkl=generalized_KL(x, y), gkl_axis0=generalized_KL(x, y, axis=0),
        gkl_axis1=generalized_KL(x, y, axis=1),
        ns0=normalize_sum(a, axis=0), ns1=normalize_sum(a, axis=1),
        scale_lines=rnmf._scale(np.array([[1, 2, 3], [4, 5, 6]]),
                                np.array([2, 3]), axis=1),
        scale_cols=rnmf._scale(np.array([[1, 2, 3], [4, 5, 6]]),
                               np.array([3, 2, 1]), axis=0))
    # edge cases: all-zero row / all-zero column stay finite
    X, H0 = gen_inputs(81, 12, 9, 3)
    X[4, :] = 0
    X[:, 2] = 0
    m, W, errors, _ = ref_fit(X, H0, 3, 10, 0)
    out.update(edge_W=W, edge_H=m.components_, edge_errors=errors)
    # error messages
    msgs = {}
    for key, bad in [('neg', np.array([[1., -1.], [0., 1.]])),
                     ('nan', np.array([[1., np.nan], [0., 1.]]))]:
        try:
            rnmf.KLdivNMF(n_components=1).fit(bad)
        except ValueError as e:
            msgs[key] = str(e)
    out['msg_neg'] = msgs['neg']
    out['msg_nan'] = msgs['nan']
    save('g8_known', **out)


if __name__ == '__main__':
    for g in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10):
        g()
