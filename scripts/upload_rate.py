"""The boundary's host hand-over (klnmf_upload_V: pageable numpy block -> device staging -> tiled 16-bit / exact image) timed
beside what the link gives a pinned buffer.  C2-sized V (50 000 x 4096) by default.

    python3 scripts/upload_rate.py [--n 50000] [--f 4096] [--precision f16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=50000)
    ap.add_argument('--f', type=int, default=4096)
    ap.add_argument('--k', type=int, default=50)
    ap.add_argument('--precision', default='f16')
    ap.add_argument('--repeats', type=int, default=3)
    args = ap.parse_args()
    import torch
    from multimodal_amd import _native
    n, f = args.n, args.f
    rs = np.random.RandomState(0)
    V32 = rs.random_sample((n, f)).astype(np.float32)
    V64 = V32.astype(np.float64)
    print('V %d x %d: fp32 %.2f GB, fp64 %.2f GB; host threads available: %d' % (n, f, V32.nbytes / 1e9, V64.nbytes / 1e9,
                                                                                len(os.sched_getaffinity(0))))
    # the link itself: pinned host buffer -> device
    pin = torch.empty((n, f), dtype=torch.float32).pin_memory()
    dev = torch.empty((n, f), dtype=torch.float32, device='cuda')
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print('pinned fp32 buffer -> device (torch): %.1f ms  %.1f GB/s' % (1e3 * dt, V32.nbytes / dt / 1e9))
    t0 = time.perf_counter()
    pin.numpy()[:] = V32
    dt = time.perf_counter() - t0
    print('numpy copy into the pinned buffer (one thread): %.1f ms  %.1f GB/s' % (1e3 * dt, V32.nbytes / dt / 1e9))
    t0 = time.perf_counter()
    dev.copy_(torch.from_numpy(V32))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('pageable fp32 array -> device (torch): %.1f ms  %.1f GB/s' % (1e3 * dt, V32.nbytes / dt / 1e9))
    del pin, dev
    for name, V in (('fp32', V32), ('fp64', V64)):
        with _native.Context(args.precision) as c:
            c.set_problem(n, f, args.k, 10)
            c.set_v_max(1.0)
            for rep in range(args.repeats):
                c.synchronize()
                t0 = time.perf_counter()
                c.upload_V(V, row0=0, col0=0, scale=1.0)
                c.synchronize()
                dt = time.perf_counter() - t0
                print('klnmf_upload_V %s (%s storage) #%d: %.1f ms  %.1f GB/s of host bytes, %.2f Gelements/s' % (
                    name, args.precision, rep, 1e3 * dt, V.nbytes / dt / 1e9, n * f / dt / 1e9))


if __name__ == '__main__':
    main()
