import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/oracle') else os.getcwd())
import numpy as np
from oracle import klnmf_oracle as orc
from multimodal_amd import _native
n, f, k = 8192, 4096, 200
X = orc.synthetic_V(1234, n, f, k); H0 = orc.synthetic_H0(1234, f, k)
c = _native.Context('bf16')
c.set_problem(n, f, k, 4); c.set_v_max(X.max()); c.upload_V(X); c.set_H(H0); c.init_W()
e0 = c.error()
errs, nd, st = c.run(3, True, 0.0)
e3 = c.error()
print(os.environ.get('KLNMF_ROWPASS'), 'error() at init %.9e  run errors %s  error() after %.9e' % (e0, ['%.9e' % e for e in errs], e3))
