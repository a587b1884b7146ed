"""Debug aid: run the same small fits with the generation-1 and ping-pong row pass
(KLNMF_ROWPASS=1|4, separate processes) and compare W / H / errors."""
import os, subprocess, sys, json
import numpy as np

CASES = [(2048, 256, 40, 2), (2048, 256, 96, 2), (2048, 256, 128, 2), (2048, 256, 32, 2)]

def child(gen):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import klnmf_oracle as orc
    from multimodal_amd.lib.nmf import KLdivNMF
    out = {}
    for (n, f, k, it) in CASES:
        X = orc.synthetic_V(5, n, f, k); H0 = orc.synthetic_H0(5, f, k)
        m = KLdivNMF(n_components=k, tol=0, max_iter=it, precision='bf16')
        m._init_dictionary = H0
        W, err = m.fit_transform(X, return_errors=True)
        out['%d_%d_%d' % (n, f, k)] = dict(W=W, H=m.components_, err=np.asarray(err))
    np.savez('/tmp/rowgen_%s.npz' % gen, **{k + '_' + kk: v for k, d in out.items() for kk, v in d.items()})

if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1]); sys.exit(0)
    for g in ('1', '4'):
        env = dict(os.environ, KLNMF_ROWPASS=g)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), g], env=env)
    a = np.load('/tmp/rowgen_1.npz'); b = np.load('/tmp/rowgen_4.npz')
    for key in a.files:
        x, y = a[key], b[key]
        den = np.abs(x).max() + 1e-300
        print('%-22s max|d|/max = %.3e' % (key, np.abs(x - y).max() / den), x.ravel()[:3], y.ravel()[:3])
