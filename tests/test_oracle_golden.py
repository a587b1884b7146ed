"""Pin the CPU oracle (oracle/klnmf_oracle.py) to vectors produced by the
reference itself (tests/golden/make_golden.py) and to the known-answer values
the reference's own tests hold.  CPU only."""
import io
import contextlib

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import klnmf_oracle as orc
from tests import golden_inputs as gi

RTOL = 1e-10  # dgemm summation order is implementation-defined: not bitwise


@pytest.mark.parametrize('name', ['g1_20x30_k3', 'g1_37x53_k7',
                                  'g1_500x1000_k10'])
def test_g1_fit(name):
    g = gi.load(name)
    X, H0 = gi.gen_inputs(int(g['seed']), int(g['n']), int(g['f']), int(g['k']))
    for it in g['iters']:
        W, H, errors = orc.fit_transform(X, k=int(g['k']), H0=H0,
                                         max_iter=int(it), tol=0)
        assert len(errors) == len(g['errors_%d' % it])
        assert_allclose(errors, g['errors_%d' % it], rtol=RTOL)
        assert_allclose(W, g['W_%d' % it], rtol=1e-9, atol=1e-13)
        assert_allclose(H, g['H_%d' % it], rtol=1e-9, atol=1e-15)
        assert_allclose(orc.kl_error(X, W, H), g['final_%d' % it], rtol=RTOL)


def test_g2_transform():
    g = gi.load('g2_transform')
    X, H0, Xt = gi.g2_inputs(g)
    k = int(g['k'])
    W, H, _ = orc.fit_transform(X, k=k, H0=H0, max_iter=30, tol=0)
    assert_allclose(H, g['H'], rtol=1e-9)
    assert_allclose(W, g['W_train'], rtol=1e-9)
    Wt, et = orc.transform(Xt, g['H'], max_iter=25, tol=0)
    assert_allclose(Wt, g['Wt'], rtol=1e-9)
    assert_allclose(et, g['errors_t'], rtol=RTOL)
    a, b = g['sl']
    Ws, es = orc.transform(Xt[:, a:b], g['H'][:, a:b], max_iter=25, tol=0)
    assert_allclose(Ws, g['Ws'], rtol=1e-9)
    assert_allclose(es, g['errors_s'], rtol=RTOL)


def test_g3_single_steps():
    g = gi.load('g3_steps')
    X, W, H = gi.g3_inputs(g)
    Q = orc.ratio_q(X, W, H)
    assert_allclose(Q, g['Q'], rtol=1e-12)
    Wn = orc.updated_w(X, W, H, Q=Q)
    assert_allclose(Wn, g['Wn'], rtol=1e-12)
    assert_allclose(orc.updated_h(X, Wn, H, Q=Q), g['Hn'], rtol=1e-12)
    assert_allclose(orc.updated_h(X, W, H), g['Hn_noq'], rtol=1e-12)
    assert_allclose(orc.kl_error(X, W, H), g['err'], rtol=1e-12)
    W_upd, H_upd = orc.update_step(X, W, H, fit=True)
    assert_allclose(W_upd, g['W_upd'], rtol=1e-12)
    assert_allclose(H_upd, g['H_upd'], rtol=1e-12)
    W_s, H_s = orc.update_step(X, W, H, fit=True, scale_W=True)
    assert_allclose(W_s, g['W_upd_scaled'], rtol=1e-12)
    assert_allclose(H_s, g['H_upd_scaled'], rtol=1e-12)
    # _fit=False leaves the dictionary untouched (reference nmf.py:254)
    _, H_same = orc.update_step(X, W, H, fit=False)
    assert H_same is H


def _fit_capture(X, H0, k, max_iter, tol):
    buf = io.StringIO()
    with contextlib.redirect_stderr(buf):
        W, H, errors = orc.fit_transform(X, k=k, H0=H0, max_iter=max_iter,
                                         tol=tol)
    return W, H, errors, buf.getvalue()


def test_g4_stop_rule_and_warning():
    g = gi.load('g4_tol')
    k = int(g['k'])
    X, H0 = gi.gen_inputs(int(g['seed']), int(g['n']), int(g['f']), k)
    for sfx, it, tol in [('', 200, 1e-6), ('2', 200, 1e-3), ('3', 4, 1e-9)]:
        W, H, errors, msg = _fit_capture(X, H0, k, it, tol)
        assert len(errors) == len(g['errors' + sfx])       # early-stop count
        assert_allclose(errors, g['errors' + sfx], rtol=RTOL)
        assert_allclose(W, g['W' + sfx], rtol=1e-8)
        assert_allclose(H, g['H' + sfx], rtol=1e-8)
        assert bool(msg) == bool(g['warned' + sfx])
    assert str(g['msg3']) == "Warning: Iteration limit reached during fit\n"
    W4, H4, e4, msg4 = _fit_capture(g['Xf'], g['H0f'], 3, 200, 1e-6)
    assert len(e4) == len(g['errors4'])
    assert_allclose(e4, g['errors4'], rtol=1e-7, atol=1e-14)
    assert e4[-1] < e4[0] * 1e-3    # reference tests/test_nmf_kl.py:162-165


@pytest.mark.parametrize('name', ['g5_learner2', 'g5_learner3'])
def test_g5_learner(name):
    g = gi.load(name)
    blocks, dims, H0, test = gi.g5_inputs(g)
    coefs = list(g['coefs'])
    k = int(g['k'])
    V = orc.stack_modalities(blocks, coefs)
    assert_allclose(V.sum(axis=0), g['stacked_sum'], rtol=1e-12)
    dico, _ = orc.learner_train(blocks, coefs, k, 20, H0)
    assert_allclose(dico, g['dico'], rtol=1e-9)
    for i in range(len(dims)):
        a, b = orc.axis_range(dims, i)
        assert_allclose(dico[:, a:b], g['dico_%d' % i], rtol=1e-9)
        W = orc.learner_internal([test[i]], [coefs[i]], [g['dico'][:, a:b]], 15)
        assert_allclose(W, g['internal_%d' % i], rtol=1e-9)
    a0, b0 = orc.axis_range(dims, 0)
    a1, b1 = orc.axis_range(dims, 1)
    W01 = orc.learner_internal(test[:2], coefs[:2],
                               [g['dico'][:, a0:b0], g['dico'][:, a1:b1]], 15)
    assert_allclose(W01, g['internal_01'], rtol=1e-9)
    assert_allclose(g['internal_0'].dot(g['dico'][:, a1:b1]), g['m2m_0_to_1'],
                    rtol=1e-9)


def test_g6_sparse_matches_dense_restatement():
    """The reference's CSR branch evaluates Q only on nnz(X); the dense rules on
    the densified matrix agree to ~eps (the product densifies sparse input)."""
    g = gi.load('g6_sparse')
    dense, W, H = gi.g6_inputs(g)
    assert_allclose(orc.kl_error(dense, W, H), g['err'], rtol=1e-7)
    Q = orc.ratio_q(dense, W, H)
    mask = dense != 0
    assert_allclose(Q[mask], g['Q_dense'][mask], rtol=1e-12)
    assert np.all(Q[~mask] < 1e-6)
    Wn, Hn = orc.update_step(dense, W, H, fit=True)
    assert_allclose(Wn, g['Wn'], rtol=1e-5)
    assert_allclose(Hn, g['Hn'], rtol=1e-5)


def test_g9_sparse_branch_fit_transform_and_stop():
    """The CSR branch of the reference (Q only on the stored entries of X): full fit, transform on the
    learnt dictionary, and the default-tolerance stop, against outputs of the imported reference."""
    import scipy.sparse as sp
    g = gi.load('g9_sparse_fit')
    dense, H0 = gi.g9_inputs(g)
    X = sp.csr_matrix(dense)
    k = int(g['k'])
    W, H, errors = orc.sparse_fit_transform(X, k, H0, max_iter=12, tol=0)
    assert_allclose(errors, g['errors'], rtol=1e-11)
    assert_allclose(W, g['W'], rtol=1e-9, atol=1e-300)
    assert_allclose(H, g['H'], rtol=1e-9, atol=1e-300)
    Wt, _, _ = orc.sparse_fit_transform(X[:20], k, H, max_iter=12, tol=0, fit=False, components=H)
    assert_allclose(Wt, g['Wt'], rtol=1e-9, atol=1e-300)
    W2, H2, e2 = orc.sparse_fit_transform(X, k, H0, max_iter=300, tol=1e-4)
    assert len(e2) == len(g['errors_tol'])
    assert_allclose(e2, g['errors_tol'], rtol=1e-10)
    assert_allclose(H2, g['H_tol'], rtol=1e-8, atol=1e-300)
    # and it is NOT the dense rule on the densified matrix (structural zeros stay out of Q)
    Wd, Hd, ed = orc.fit_transform(dense, k=k, H0=H0, max_iter=12, tol=0)
    assert abs(ed[-1] - errors[-1]) > 1e-9 * abs(errors[-1])


def test_g10_pairwise_measures():
    g = gi.load('g10_distances')
    A, B = gi.g10_inputs(g)
    for name in ('kl_div', 'rev_kl_div', 'sym_kl_div', 'frobenius', 'cosine_diff'):
        assert_allclose(orc.pairwise_distances(A, B, name), g[name], rtol=1e-12, atol=1e-300)


def test_g7_float32_reference_run():
    g = gi.load('g7_float32')
    k = int(g['k'])
    X, H0 = gi.gen_inputs(int(g['seed']), int(g['n']), int(g['f']), k)
    W, H, errors = orc.fit_transform(X, k=k, H0=H0, max_iter=40, tol=0)
    # the reference ran in float32 here; fp64 oracle agrees to fp32 accuracy
    assert len(errors) == len(g['errors'])
    assert_allclose(errors, g['errors'], rtol=2e-5)
    assert_allclose(W, g['W'], rtol=2e-3, atol=1e-5)


def test_g8_known_answers_and_edges():
    g = gi.load('g8_known')
    x = np.array([[1., 2.], [3., 4.]])
    y = np.array([[2., 2.], [1., 4.]])
    a = np.array([[1., 2., 3.], [4., 5., 6.]])
    assert_allclose(orc.generalized_kl(x, y), g['gkl'], rtol=1e-14)
    assert_allclose(orc.generalized_kl(x, y, axis=0), g['gkl_axis0'], rtol=1e-14)
    assert_allclose(orc.generalized_kl(x, y, axis=1), g['gkl_axis1'], rtol=1e-14)
    assert_allclose(orc.normalize_sum(a, axis=0), g['ns0'], rtol=1e-15)
    assert_allclose(orc.normalize_sum(a, axis=1), g['ns1'], rtol=1e-15)
    assert_allclose(orc.scale_matrix(np.array([[1, 2, 3], [4, 5, 6]]),
                                     np.array([2, 3]), axis=1), g['scale_lines'])
    assert_allclose(orc.scale_matrix(np.array([[1, 2, 3], [4, 5, 6]]),
                                     np.array([3, 2, 1]), axis=0), g['scale_cols'])
    X, H0 = gi.g8_edge_inputs()
    W, H, errors = orc.fit_transform(X, k=3, H0=H0, max_iter=10, tol=0)
    assert_allclose(W, g['edge_W'], rtol=1e-9, atol=1e-300)
    assert_allclose(H, g['edge_H'], rtol=1e-9, atol=1e-300)
    assert np.all(W[4] == 0) and np.all(np.isfinite(H))
    with pytest.raises(ValueError) as e:
        orc.check_input(np.array([[1., -1.], [0., 1.]]))
    assert str(e.value) == str(g['msg_neg'])
    with pytest.raises(ValueError) as e:
        orc.check_input(np.array([[1., np.nan], [0., 1.]]))
    assert str(e.value) == str(g['msg_nan'])


def test_reference_known_answer_values():
    """Values pinned by the reference's own tests, restated as data:
    tests/test_metrics.py:48-54 (log 2 + 3), tests/test_array_utils.py:31-41,
    tests/test_nmf_kl.py:56-68."""
    x = np.array([[1., 0.], [0., 2.]])   # not the reference's array; the
    y = np.array([[1., 0.], [0., 2.]])   # property: KL(x, x) == 0
    assert orc.generalized_kl(x, y) == 0
    # 1-homogeneity (reference tests/test_metrics.py:42-46)
    rs = np.random.RandomState(0)
    a, b = rs.random_sample((5, 6)), rs.random_sample((5, 6))
    assert_allclose(orc.generalized_kl(3 * a, 3 * b, eps=0),
                    3 * orc.generalized_kl(a, b, eps=0), rtol=1e-12)
    with pytest.raises(ValueError):
        orc.normalize_sum(np.ones((2, 2)), axis=2)
    with pytest.raises(ValueError):
        orc.scale_matrix(np.zeros((3, 4)), np.zeros(4), axis=3)
    with pytest.raises(ValueError):
        orc.scale_matrix(np.zeros((3, 4, 6)), np.zeros(3), axis=1)
    with pytest.raises(ValueError):
        orc.scale_matrix(np.zeros((3, 4)), np.zeros(2), axis=1)


def test_reference_pinned_values():
    """The exact input/expected pairs the reference's tests pin."""
    # tests/test_metrics.py:48-54
    x = np.zeros((4, 2))
    x[1, 1] = 1
    y = .5 * np.ones((4, 2))
    assert_allclose(orc.generalized_kl(x, y), np.log(2.) + 3., rtol=0, atol=1e-6)
    # tests/test_array_utils.py:31-41
    a = np.array([[0., 1., 3.], [2., 3., 3.]])
    assert np.all(orc.normalize_sum(a, axis=0)
                  == np.array([[0., .25, .5], [1., .75, .5]]))
    assert np.all(orc.normalize_sum(a, axis=1)
                  == np.array([[0., .25, .75], [.25, .375, .375]]))
    # tests/test_array_utils.py:54-58
    z = np.abs(np.random.RandomState(3).random_sample((2, 4)))
    z[1, :] = 0
    assert not np.any(np.isnan(orc.normalize_sum(z, axis=1)))
    # tests/test_nmf_kl.py:56-68
    m = np.array([[1, 2, 3], [4, 5, 6]])
    assert np.all(orc.scale_matrix(m, np.array([2, 3]), axis=1)
                  == np.array([[2, 4, 6], [12, 15, 18]]))
    assert np.all(orc.scale_matrix(m, np.array([3, 2, 1]), axis=0)
                  == np.array([[3, 4, 3], [12, 10, 6]]))
