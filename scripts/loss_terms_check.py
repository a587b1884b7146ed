#!/usr/bin/env python3
"""Which of the three sums of the 16-bit mode's loss carries the noise?  klnmf_loss_terms gives the kernel's
sum x~ ln q, sum W.H and sum x~; the same sums are recomputed here in fp64 from EXACTLY the operands the kernel
uses (the 16-bit images of the fp32 masters, V as stored), so every difference is fp32 arithmetic inside the kernel,
not operand rounding.
    python scripts/loss_terms_check.py --n 16384 --f 4096 --k 200 --iters 10
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def r_bf16(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=16384)
    p.add_argument('--f', type=int, default=4096)
    p.add_argument('--k', type=int, default=200)
    p.add_argument('--iters', type=int, default=10)
    p.add_argument('--mode', default='bf16')
    p.add_argument('--opnd', default='bf16', help='operand rounding of the mode being checked: bf16 | fp16')
    a = p.parse_args()
    import torch
    from multimodal_amd import _native
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    Ht = torch.randn((a.k, a.f), device=dev, generator=g).square_().mul_(0.5)
    Wt = torch.rand((a.n, a.k), device=dev, generator=g).neg_().add_(1.0).log_().neg_()
    V = torch.rand((a.n, a.f), device=dev, generator=g).mul_(0.05)
    V.addmm_(Wt, Ht, alpha=1.0 / a.k)
    Vh = V.cpu().numpy()
    vmax = float(Vh.max())
    e = int(np.frexp(vmax)[1])
    c = 2.0 ** (15 - e)
    xs = (Vh.astype(np.float64) * c).astype(np.float16).astype(np.float64) / c      # V as stored
    H0 = bench.make_H0(1234, a.f, a.k)
    eps = 1e-8
    with _native.Context(a.mode, device=0) as ctx:
        ctx.set_problem(a.n, a.f, a.k, 1)
        ctx.set_v_max(vmax)
        ctx.upload_blocks([Vh])
        ctx.set_H(H0)
        ctx.init_W()
        print('  it   loss(kernel)      loss(fp64, same operands)   rel    |  d(sum x ln q)/sum x   d(sum WH)/sum x   d(sum x)/sum x')
        for it in range(a.iters):
            t = ctx.loss_terms()
            W = ctx.get_W(dtype=np.float32)
            H = ctx.get_H(dtype=np.float32)
            if a.opnd == 'bf16':
                Wb = r_bf16(W * np.float32(c)).astype(np.float64) / c
                Hb = r_bf16(H).astype(np.float64)
                eps_d = float(r_bf16(np.array([eps * c], dtype=np.float32))[0]) / c if a.k % 16 else eps
            else:
                raise SystemExit('fp16 operand emulation: fill in the scales of the mode')
            d = Wb @ Hb
            s_ln = float((xs * np.log((xs + eps) / (d + eps_d))).sum())
            s_d = float(d.sum())
            s_x = float(xs.sum())
            lk = t[0] + t[1] - t[2]
            le = s_ln + s_d - s_x
            print('%4d  %.9e  %.9e  %+.2e  |  %+.2e  %+.2e  %+.2e' % (
                it, lk, le, (lk - le) / le, (t[0] - s_ln) / s_x, (t[1] - s_d) / s_x, (t[2] - s_x) / s_x))
            ctx.update(True)


if __name__ == '__main__':
    main()
