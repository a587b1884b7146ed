"""Debug aid: per-mode comparison of the ping-pong row pass against an fp64 numpy model
(init W = V.H^T, loss, one update) for one shape; prints where W deviates."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import klnmf_oracle as orc
from multimodal_amd import _native

n, f, k = [int(x) for x in sys.argv[1:4]]
X = orc.synthetic_V(5, n, f, k); H0 = orc.synthetic_H0(5, f, k)
c = _native.Context('bf16')
c.set_problem(n, f, k, 4)
c.set_v_max(X.max()); c.upload_V(X); c.set_H(H0); c.init_W()
W = c.get_W()
Wref = X.dot(H0.T)
rel = np.abs(W - Wref) / np.abs(Wref).max()
print('init W: max rel dev %.3e' % rel.max())
bad = np.argwhere(rel > 0.05)
if len(bad):
    rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
    print('  bad rows (count %d): %s' % (len(rows), rows[:40])); print('  bad comps:', cols[:40])
e = c.error()
print('loss (LOSS mode) %.6e  oracle %.6e' % (e, orc.kl_error(X, Wref, H0)))
c.update(True)
W1 = c.get_W()
W1ref = orc.updated_w(X, Wref, H0)
rel = np.abs(W1 - W1ref) / np.abs(W1ref).max()
print('updated W: max rel dev %.3e' % rel.max())
bad = np.argwhere(~(rel < 0.05))
if len(bad):
    rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
    print('  bad rows (count %d): %s' % (len(rows), rows[:64])); print('  bad comps:', cols[:40])
