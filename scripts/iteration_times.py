#!/usr/bin/env python3
"""Per-ITERATION GPU time of one fit loop, without a profiler: an event pair around every klnmf_run_more(1) on the loop's own
stream (2.8 us of dispatch gap per event record: the same for every iteration), several loops from the same start.  Shows what
a segment's mean hides: the monitored iterations (fp8 iterations 1, 2, 4, 8, 16, 32, ...: monitor launch + poll), the loop's
first two iterations on 16-bit tiles, and warm-up effects after an idle fence.

    python3 scripts/iteration_times.py --rows 125000 [--features 4096 --components 200 --iters 43 --loops 4] [--native]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=125000)
ap.add_argument('--features', type=int, default=4096)
ap.add_argument('--components', type=int, default=200)
ap.add_argument('--iters', type=int, default=43)
ap.add_argument('--loops', type=int, default=4)
ap.add_argument('--native', action='store_true', help='the collective branch on a one-rank RCCL communicator (KLNMF_COMM_SINGLE=1)')
args = ap.parse_args()
if args.native:
    os.environ['KLNMF_COMM_SINGLE'] = '1'
    os.environ['KLNMF_DEV'] = '1'
import torch      # noqa: E402
from multimodal_amd.distributed import ShardedKLNMF      # noqa: E402

n, f, k = args.rows, args.features, args.components
m = ShardedKLNMF(n, n, f, k, max_iter=args.iters, precision='f16', collective='native' if args.native else 'torch')
bench.fill_shard_device(torch, m, 1234, 0, n, f, k)
H0 = bench.make_H0(1234, f, k)
for loop in range(args.loops):
    m.set_H(H0)
    m.init_W()
    m.begin()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
    evs[0].record()
    for it in range(args.iters):
        m.iterate_many(1, fit=True, tol=0.0)
        evs[it + 1].record()
    torch.cuda.synchronize()
    t = [1e3 * evs[i].elapsed_time(evs[i + 1]) for i in range(args.iters)]
    m.end()
    rep = m.ctx.fp8_report()
    body = sorted(t[3:])
    print('loop %d: median of iterations 3.. %.0f us, mean %.0f us; checks %d' % (loop, body[len(body) // 2], sum(t[3:]) / len(t[3:]), rep['monitor_checks']))
    print('   ' + ' '.join('%.0f' % x for x in t))
m.close()
