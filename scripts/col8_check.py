import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from multimodal_amd.lib.nmf import KLdivNMF
from oracle import klnmf_oracle as orc
def fit(env):
    for k_, v in env.items(): os.environ[k_] = v
    X = orc.synthetic_V(5, 70000, 512, 32); H0 = orc.synthetic_H0(5, 512, 200)
    m = KLdivNMF(n_components=200, max_iter=6, tol=0, precision='f16'); m._init_dictionary = H0
    W, e = m.fit_transform(X, return_errors=True)
    return W, m.components_, np.array(e)
Wa, Ha, ea = fit({'KLNMF_COL8': '0'})
Wb, Hb, eb = fit({'KLNMF_COL8': '1'})
print('losses', ea, eb)
print('rel loss diff', np.abs(ea / eb - 1).max(), 'H rel-to-max', np.abs(Ha - Hb).max() / np.abs(Ha).max(), 'W', np.abs(Wa - Wb).max() / np.abs(Wa).max())
