#!/usr/bin/env python3
"""Where does the noise of the 16-bit mode's REPORTED loss come from?  Per iteration of a fit in a 16-bit mode:
  rep   the loss the mode reports for its current (W, H)            (ctx.error(): the row pass's loss arithmetic)
  true  the loss of the same fp32 masters evaluated by the f32 kernels (fp32 GEMM, fp64 loss sums)
  f32   the loss of the f32 mode's own trajectory from the same start
    python scripts/loss_noise_gpu.py --n 65536 --f 4096 --k 200 --iters 50 [--mode bf16]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=65536)
    p.add_argument('--f', type=int, default=4096)
    p.add_argument('--k', type=int, default=200)
    p.add_argument('--iters', type=int, default=50)
    p.add_argument('--mode', default='bf16')
    p.add_argument('--seed', type=int, default=1234)
    p.add_argument('--no-f32-trajectory', action='store_true')
    a = p.parse_args()
    import torch
    from multimodal_amd.distributed import ShardedKLNMF
    torch.cuda.set_device(0)
    H0 = bench.make_H0(a.seed, a.f, a.k)

    def make(mode):
        m = ShardedKLNMF(a.n, a.n, a.f, a.k, max_iter=a.iters, precision=mode)
        bench.fill_shard_device(torch, m, a.seed, 0, a.n, a.f, a.k)
        m.set_H(H0)
        m.init_W()
        return m
    fast = make(a.mode)
    ref = make('f32')
    rep, true = [], []
    for it in range(a.iters):
        rep.append(fast.ctx.error())
        ref.ctx.set_W(fast.ctx.get_W(dtype=np.float32))
        ref.ctx.set_H(fast.ctx.get_H(dtype=np.float32))
        true.append(ref.ctx.error())
        fast.ctx.update(True)
    traj = []
    if not a.no_f32_trajectory:
        ref.set_H(H0)
        ref.init_W()
        for it in range(a.iters):
            traj.append(ref.ctx.error())
            ref.ctx.update(True)
    fast.close()
    ref.close()
    rep, true, traj = np.array(rep), np.array(true), np.array(traj)
    print('%s  n=%d f=%d k=%d' % (a.mode, a.n, a.f, a.k))
    print(' it        reported            true(f32 eval)   (rep-true)/true   descent(true)/true   f32 trajectory   (true-f32)/f32')
    for i in range(a.iters):
        print('%3d  %.10e  %.10e  %+.2e  %s  %s' % (
            i, rep[i], true[i], (rep[i] - true[i]) / true[i],
            '%+.2e' % ((true[i - 1] - true[i]) / true[i]) if i else '    -    ',
            '%.10e  %+.2e' % (traj[i], (true[i] - traj[i]) / traj[i]) if len(traj) else ''))
    print('reported rises at', (np.nonzero(np.diff(rep) >= 0)[0] + 1).tolist())
    print('true rises at    ', (np.nonzero(np.diff(true) >= 0)[0] + 1).tolist())
    if len(traj):
        print('f32 traj rises at', (np.nonzero(np.diff(traj) >= 0)[0] + 1).tolist())


if __name__ == '__main__':
    main()
