"""Precision experiment (round 4, CPU): the fp8 ratio tiles' dead zone.  Exact fp64 updates, only the ratio that enters the H numerator
rounded to e4m3 of ratio / 8 from the third iteration on (large ratios exact, as the product's fix-ups leave them):

    python3 experiments/fp8_tiles_dead_zone_emulation.py
    33118 424 1  KL rel 1.22e-03 dH 5.48e-03      (the HIP path on fp8 tiles: 1.23e-3, 5.7e-3)
    33118 424 2  KL rel 1.33e-05 dH 4.11e-03
    33118 424 5  KL rel 9.81e-08 dH 2.43e-04
With one or two components on this low-rank data the ratios crowd around 1, where e4m3 steps by 6-12 %: the deviations from 1 --
all the H rule has to work with -- are rounded away together instead of averaging out.  klnmf_api.hip (begin_fp8_loop) keeps
16-bit tiles for k < 4."""
import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from oracle import klnmf_oracle as orc
def e4m3(q):
    t=torch.from_numpy(q/8.0).clamp(max=448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64).numpy()*8.0
    return t
def run(n,f,k,iters,quant):
    X=orc.synthetic_V(7+n+f+k,n,f,k); H=orc.synthetic_H0(7+n+f+k,f,k); W=X.dot(H.T)
    for it in range(iters):
        Q=orc.ratio_q(X,W,H)
        Wn=orc.updated_w(X,W,H,Q=Q)
        Qh=Q
        if quant and it>=2:
            Qh=e4m3(Q)
            big=Q>=256.0
            Qh=np.where(big,Q,Qh)      # exact fix-ups of large ratios
        num=Wn.T.dot(Qh)
        Hn=H*num/Wn.sum(axis=0)[:,None]; Hn=Hn/Hn.sum(axis=1,keepdims=True)
        W,H=Wn,Hn
    return orc.kl_error(X,W,H),H,Q
for (n,f,k) in [(33118,424,1),(33118,424,2),(33118,424,5)]:
    k0,H0,Q=run(n,f,k,3,False); k1,H1,_=run(n,f,k,3,True)
    print(n,f,k,'KL rel %.2e dH %.2e'%(abs(k1-k0)/k0,np.abs(H1-H0).max()/H0.max()),' q quantiles',np.quantile(Q,[0.01,0.1,0.5,0.9,0.99,0.9999]).round(3))
