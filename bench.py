#!/usr/bin/env python3
"""Headline benchmark: KL-NMF update-iterations/s at V = 1M x 4096, k = 200
(BASELINE.json `metric`, config 4), V row-sharded over N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one fit iteration of reference nmf.py:212-222: loss + ratio +
W rule (row pass), stop rule, H numerator (column pass), all-reduce of the
k x f numerator over the ranks, H rule + row normalisation.  Inputs (V tiles, W0,
H0) are resident in HBM before the timed region.  Rank 0 prints one JSON line.

Strong scaling: the total problem (n rows) is fixed, each rank holds n/N rows.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks (MI355X_MICROARCH.md): dense bf16 MFMA, HBM3E
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--n', type=int, default=1000000)
    p.add_argument('--f', type=int, default=4096)
    p.add_argument('--k', type=int, default=200)
    p.add_argument('--precision', default='bf16', choices=['bf16', 'bf16_v32', 'f32', 'f64'])
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-rows', type=int, default=8192)
    p.add_argument('--seed', type=int, default=1234)
    p.add_argument('--tol', type=float, default=None,
                   help='stop-rule tolerance of nmf.py:207,215 (relative; x n x f inside).  Default: the rule is evaluated every '
                        'iteration but can never fire, so that exactly --steps full iterations are timed: with tol = 0 '
                        '(MultimodalLearner.train) a loss that rises by bf16 rounding noise on the plateau of the synthetic '
                        'problem would stop the loop and the remaining timed launches would return at their first instruction')
    return p.parse_args()


def make_H0(seed, f, k):
    """Row-normalised |U(0,1)| + 0.01 (the reference's init rule, nmf.py:149-151)."""
    h = np.random.RandomState(seed - 1).random_sample((k, f)) + .01
    return h / (1e-16 + h.sum(axis=1, keepdims=True))


def fill_shard_device(torch, model, seed, rank, n_local, f, k, block=8192, vscale=1.0):
    """Synthetic non-negative V = Wt.Ht/k + 0.05*U (factorisable + noise, SURVEY 8d
    shape), generated block-wise on the GPU and fed to the tiling upload.  `vscale` multiplies the matrix on its way
    in (the upload's per-modality coefficient, learner.py:53-56): tests use it for the homogeneity property."""
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)                       # Ht identical on every rank
    # Gamma(1/2) = N(0,1)^2 / 2 and Gamma(1) = -log(U): built from randn / rand only, whose streams are reproduced
    # exactly by restoring the generator state (torch._standard_gamma's rejection sampler is not: a second pass
    # produced different values, the maximum of the first pass was exceeded and fp16 storage overflowed)
    Ht = torch.randn((k, f), device=dev, generator=g).square_().mul_(0.5)
    g.manual_seed(seed + 1000 * (rank + 1))

    def block_of(rows):
        Wt = torch.rand((rows, k), device=dev, generator=g).neg_().add_(1.0).log_().neg_()     # Exp(1)
        Vb = torch.rand((rows, f), device=dev, generator=g).mul_(0.05)
        Vb.addmm_(Wt, Ht, alpha=1.0 / k)
        return Vb
    # the 16-bit storage factor needs the global max before the first upload:
    # generate once for the max, then regenerate the identical stream for the upload
    state = g.get_state()
    vmax = 0.0
    for r0 in range(0, n_local, block):
        vmax = max(vmax, float(block_of(min(block, n_local - r0)).max().item()))
    model.set_v_max(vmax * vscale)
    g.set_state(state)
    for r0 in range(0, n_local, block):
        Vb = block_of(min(block, n_local - r0))
        model.upload_V_device(Vb.contiguous(), row0=r0, col0=0, scale=vscale)
    torch.cuda.synchronize()


def measured_traffic(args, n_local):
    """HBM bytes per row-pass launch from the committed PMC run of this exact workload
    (profiles/r01_pmc_traffic.json, produced by scripts/pmc_profile.sh); None if the
    workload differs -- the counters cannot be read from inside this process."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        d = json.load(open(path))
    except Exception:
        return None
    w = d.get('workload', {})
    if (w.get('n_local'), w.get('f'), w.get('k'), w.get('precision')) != (n_local, args.f, args.k, args.precision):
        return None
    for name, v in d.get('kernels', {}).items():
        if 'k_rowpass' in name:
            return v['hbm_bytes_per_launch']
    return None


def cpu_baseline(args):
    """The oracle (numpy restatement of the reference loop, fp64, same redundant
    work) timed on this host's cores on a bounded row sample of the workload."""
    from oracle import klnmf_oracle as orc
    rows, f, k = min(args.cpu_rows, args.n), args.f, args.k
    X = orc.synthetic_V(args.seed, rows, f, k)
    H0 = make_H0(args.seed, f, k)
    W, H = orc.init_factors(X, k, H0=H0)
    t_total, iters = 0.0, 0
    losses = []
    while iters < 2 or (t_total < 8.0 and iters < 6):
        t0 = time.perf_counter()
        losses.append(orc.kl_error(X, W, H))
        W, H = orc.update_step(X, W, H, fit=True)
        t_total += time.perf_counter() - t0
        iters += 1
    final = orc.kl_error(X, W, H)
    per_iter = t_total / iters
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    its_sample = 1.0 / per_iter
    return {
        'value': its_sample * rows / args.n,          # work is exactly linear in n
        'unit': 'it/s',
        'cores': cores,
        'kind': 'port',
        'sample': 'n=%d of %d rows at f=%d, k=%d, %d fp64 iterations of the numpy oracle '
                  '(%.2f s/iter on the sample), scaled linearly in n' % (rows, args.n, f, k, iters, per_iter),
    }, (X, H0, iters, losses, final)


def gpu_parity_on_sample(args, sample):
    """Same row sample through the HIP path: final KL relative to the oracle."""
    from multimodal_amd import _native
    X, H0, iters, losses, final = sample
    with _native.Context(args.precision, device=0) as ctx:
        ctx.set_problem(X.shape[0], X.shape[1], args.k, iters)
        ctx.upload_blocks([X])
        ctx.set_H(H0)
        ctx.init_W()
        errs, n_done, stopped = ctx.run(iters, True, 0.0)
        g_final = ctx.error()
    return {'final_kl_rel_err': abs(g_final - final) / abs(final),
            'first_loss_rel_err': abs(errs[0] - losses[0]) / abs(losses[0]),
            'iterations': iters, 'rows': int(X.shape[0])}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from multimodal_amd.distributed import ShardedKLNMF, row_partition
    from multimodal_amd import _native

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal of the N > 1 code path on a one-GPU box (scripts/README.md): KLNMF_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and exchanges over gloo.  The driver's runs use one GPU per rank and RCCL ("nccl").
    rehearsal = os.environ.get('KLNMF_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('gloo' if rehearsal else 'nccl', rank=rank, world_size=world)
    n_gpus = world if world > 1 else 1

    n, f, k = args.n, args.f, args.k
    tol = -1e300 / (float(n) * f) if args.tol is None else args.tol
    r0, r1 = row_partition(n, n_gpus)[rank]
    n_local = r1 - r0
    total_iters = args.warmup + args.steps
    model = ShardedKLNMF(n, n_local, f, k, max_iter=total_iters, precision=args.precision)
    fill_shard_device(torch, model, args.seed, rank, n_local, f, k)
    model.set_H(make_H0(args.seed, f, k))
    model.init_W()
    info = _native.device_info(local_rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model.begin()
    for _ in range(args.warmup):
        model.iterate(fit=True, tol=tol)
    model.ctx.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.iterate(fit=True, tol=tol)
    fence()
    elapsed = time.perf_counter() - t0
    prof = model.ctx.profile_read(reset=True)
    model.ctx.profile_enable(False)
    errors, n_done, stopped = model.end()

    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        its = args.steps / elapsed
        row_ms = prof['rowpass_ms'] / max(1, prof['rowpass_launches'])
        col_ms = prof['colpass_ms'] / max(1, prof['colpass_launches'])
        flops_row = 4.0 * n_local * f * k          # W.H and Q.H^T, unpadded k
        flops_col = 2.0 * n_local * f * k          # W_new^T.Q
        vbytes = 2 if args.precision == 'bf16' else 4
        pingpong = (args.precision == 'bf16' and (k <= 224 or 256 < k <= 512)
                    and os.environ.get('KLNMF_ROWPASS', '4') == '4')
        # the H rule runs on the ratios the row pass stores (2 B per element of V) unless the recomputing kernel is forced
        stored_q = pingpong and (k > 256 or os.environ.get('KLNMF_COLPASS', '2') in ('2', '3'))
        # bytes the row-pass launch must move by its contract: V once, W fp32 in and out, W bf16 in and out,
        # and -- stored-ratio schedule -- the ratio tiles out
        bytes_row = n_local * f * vbytes + n_local * k * (4 + 4 + 2 + 2) + (n_local * f * 2 if stored_q else 0)
        bytes_col = (n_local * f * 2 + n_local * k * 2) if stored_q else (n_local * f * vbytes + 2 * n_local * k * 2)
        row_tflops = flops_row / (row_ms * 1e-3) / 1e12 if row_ms > 0 else None
        row_gbs = bytes_row / (row_ms * 1e-3) / 1e9 if row_ms > 0 else None
        mfma_view = {'achieved': row_tflops, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': row_tflops / PEAK_BF16_TFLOPS if row_tflops else None}
        hbm_view = {'achieved': row_gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': row_gbs / 8000.0 if row_gbs else None}
        # binding roofline of the launch = the larger of the two lower bounds (t_min = max(flops/P, bytes/BW))
        hbm_bound = bytes_row / 8000e9 > flops_row / (PEAK_BF16_TFLOPS * 1e12)
        head, other = (hbm_view, mfma_view) if hbm_bound else (mfma_view, hbm_view)
        out = {
            'metric': 'nmf_update_iterations_per_sec',
            'value': its,
            'unit': 'it/s',
            'n_gpus': n_gpus,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'bf16' if args.precision.startswith('bf16') else args.precision,
            'data': 'synthetic',
            'config': {'workload': 'KL-NMF fit iteration, V %dx%d (row-sharded), k=%d' % (n, f, k),
                       'n': n, 'f': f, 'k': k, 'rows_per_gpu': n_local,
                       'precision': args.precision,
                       'parallelism': 'rows/%d' % n_gpus},
            'samples_per_sec': its * n,
            'iterations_done': n_done,
            'stopped_early': bool(stopped),
            'valid': bool(n_done == total_iters and not stopped),     # every timed launch did its full work
            'loss_first': errors[0] if errors else None,
            'loss_last': errors[-1] if errors else None,
            'stop_rule': 'evaluated, cannot fire' if args.tol is None else 'tol=%g' % args.tol,
            'loss_finite_and_decreasing': bool(len(errors) > 1 and all(e == e and abs(e) != float('inf') for e in errors)
                                               and all(b < a for a, b in zip(errors, errors[1:]))),
            'device': info,
            'roofline': dict(
                kernel=('k_rowpass4' if pingpong else 'k_rowpass') + ' (W.H -> ratio/loss -> Q.H^T -> W rule'
                       + (', ratio tiles stored for the H rule)' if stored_q else ')'),
                bound='hbm' if hbm_bound else 'mfma', achieved=head['achieved'], peak=head['peak'], unit=head['unit'],
                frac=head['frac'], traffic=measured_traffic(args, n_local),
                avg_launch_ms=row_ms, launches=prof['rowpass_launches'],
                algorithmic_flops_per_launch=flops_row, algorithmic_hbm_bytes_per_launch=bytes_row,
                t_min_ms={'mfma': flops_row / (PEAK_BF16_TFLOPS * 1e12) * 1e3, 'hbm': bytes_row / 8000e9 * 1e3},
                **{('mfma' if hbm_bound else 'hbm'): other}),
            'kernels': {
                ('k_colpass_q2' if stored_q else 'k_colpass'): {
                    'avg_launch_ms': col_ms, 'launches': prof['colpass_launches'],
                    'algorithmic_tflops': flops_col / (col_ms * 1e-3) / 1e12 if col_ms > 0 else None,
                    'executed_tflops': (1 if stored_q else 2) * flops_col / (col_ms * 1e-3) / 1e12 if col_ms > 0 else None,
                    'algorithmic_hbm_bytes_per_launch': bytes_col,
                    'hbm_gbs': bytes_col / (col_ms * 1e-3) / 1e9 if col_ms > 0 else None,
                    'hbm_frac': bytes_col / (col_ms * 1e-3) / 1e9 / 8000.0 if col_ms > 0 else None},
                'iteration_hbm_bytes_by_contract': bytes_row + bytes_col,
                'iteration_algorithmic_tflops': 6.0 * n * f * k / (ms_per_step * 1e-3) / 1e12,
                'iteration_frac_of_bf16_peak': 6.0 * n * f * k / (ms_per_step * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS * n_gpus),
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            base, sample = cpu_baseline(args)
            out['cpu_baseline'] = base
            try:
                out['parity'] = gpu_parity_on_sample(args, sample)
            except Exception as e:   # parity is reported, never hidden
                out['parity'] = {'error': str(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    model.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
