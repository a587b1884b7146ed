#!/usr/bin/env python3
"""Loss per iteration of the same synthetic problem (bench.py's generator) in several precision modes, side by side.
    python scripts/loss_trace.py --n 250000 --f 12288 --k 500 --iters 12 --modes bf16,f32
Used to tell a precision effect from a kernel bug when the stop rule (tol = 0: "loss went up") fires in bf16 mode."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=250000)
    p.add_argument('--f', type=int, default=12288)
    p.add_argument('--k', type=int, default=500)
    p.add_argument('--iters', type=int, default=12)
    p.add_argument('--modes', default='bf16,f32')
    p.add_argument('--seed', type=int, default=1234)
    a = p.parse_args()
    import torch
    from multimodal_amd.distributed import ShardedKLNMF
    torch.cuda.set_device(0)
    out = {}
    for mode in a.modes.split(','):
        m = ShardedKLNMF(a.n, a.n, a.f, a.k, max_iter=a.iters, precision=mode)
        bench.fill_shard_device(torch, m, a.seed, 0, a.n, a.f, a.k)
        m.set_H(bench.make_H0(a.seed, a.f, a.k))
        m.init_W()
        m.begin()
        for _ in range(a.iters):
            m.iterate(fit=True, tol=-1e30)          # never stop: record every loss
        errs, n_done, stopped = m.end()
        out[mode] = list(errs)
        m.close()
        print(mode, 'n_done', n_done, 'stopped', stopped, flush=True)
    modes = list(out)
    for i in range(a.iters):
        row = ['%2d' % i]
        for mo in modes:
            e = out[mo]
            row.append('%s %.9e' % (mo, e[i]) if i < len(e) else '%s -' % mo)
        if len(modes) > 1 and all(i < len(out[mo]) for mo in modes):
            row.append('rel %.2e' % (abs(out[modes[0]][i] - out[modes[1]][i]) / abs(out[modes[1]][i])))
        print('  '.join(row))


if __name__ == '__main__':
    main()
