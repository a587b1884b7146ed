#!/usr/bin/env python3
"""Headline benchmark: KL-NMF update-iterations/s at V = 1M x 4096, k = 200
(BASELINE.json `metric`, configs[3]), V row-sharded over N GPUs of one node.

    python bench.py --gpus 1 --steps 40 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one fit iteration of reference nmf.py:212-222: loss + ratio + W rule (row pass), stop rule with
tol = 0 as MultimodalLearner.train sets it (learner.py:39-40), H numerator (column pass), all-reduce of the k x f
numerator over the ranks, H rule + row normalisation.  Inputs (V tiles, W0, H0) are resident in HBM before the
timed region.  Protocol: `--repeats` (5) independent fits from the same start; in each, W untimed
warm-up iterations, then EXACTLY K iterations timed between barrier + synchronize fences, the maximum over ranks
taken per segment; `value` = K / median segment time.  Rank 0 prints one JSON line.

Data: the seeded block-wise synthetic V of SURVEY.md 8d (multimodal_amd/synthetic.py: per-row-block RandomState
streams), the same bytes the CPU baseline sees.  Strong scaling: the problem (n rows) is fixed, rank r holds n/N rows.

Roofline accounting: the dominant kernel's ALGORITHMIC work per launch -- 4 n f k flops (W.H and
Q.H^T, unpadded k) and n f s_V + 2 n k 4 bytes (V once, fp32 W in and out) -- divided by its HIP-event time; what
the schedule additionally moves (ratio tiles, 16-bit W images) is reported beside it as `schedule_bytes_per_launch`
and the PMC-measured HBM bytes as `traffic`.
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks (MI355X_MICROARCH.md): dense bf16 / fp16 MFMA, dense fp8 (block-scaled e4m3) MFMA, HBM3E
PEAK_BF16_TFLOPS = 2500.0
PEAK_FP8_TFLOPS = 5000.0
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC_FILES = [os.path.join('profiles', 'r06_pmc_traffic.json'), os.path.join('profiles', 'r05_pmc_traffic.json'), os.path.join('profiles', 'r04_pmc_traffic.json'), os.path.join('profiles', 'r03_pmc_traffic.json'),
                     os.path.join('profiles', 'r02_pmc_traffic.json')]


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the library's sources: the PMC traffic figures are constants measured in a
    separate rocprofv3 run (counters cannot be read in-process) and go stale when the kernels change -- the line says
    at which source state they were measured and whether that is the state it ran."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'multimodal_amd', 'csrc')
    for path in sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.hip.h'))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


class Watchdog(object):
    """A rank that sits in a collective no other rank will ever join (a peer died, RCCL hangs) would block the whole job for
    ever: every timed or warm-up segment runs under this timer; when it expires the rank prints what it was doing and
    leaves with a non-zero status (os._exit: no clean-up that could block again, and never a re-exec)."""

    def __init__(self, seconds, what, rank):
        import threading
        self.what, self.rank, self.seconds = what, rank, seconds
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True

    def fire(self):
        sys.stderr.write('bench.py watchdog: rank %d spent more than %.0f s in "%s" -- a peer rank gone or a hung collective; '
                         'exiting with status 3\n' % (self.rank, self.seconds, self.what))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


SAMPLER_HELPER = r"""
import json, subprocess, sys, time
card = sys.argv[1]
samples, on = [], False
import select
def num(v):
    return float(str(v).strip('()').lower().replace('mhz', ''))
while True:
    r, _, _ = select.select([sys.stdin], [], [], 0.05 if on else 1.0)
    if r:
        cmd = sys.stdin.readline().strip()
        if cmd == 'start':
            on = True
        elif cmd == 'stop' or cmd == '':
            break
    if on:
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showuse', '--json'], capture_output=True,
                                 text=True, timeout=5).stdout
            c = json.loads(out[out.index('{'):]).get(card, {})
            power = [num(v) for kk, v in c.items() if 'power' in kk.lower()]
            samples.append((power[0] if power else None, num(c.get('sclk clock speed:')), num(c.get('GPU use (%)', 0))))
        except Exception:
            pass
print(json.dumps(samples))
"""


def under_profiler():
    """rocprofv3 preloads its tool library into this process (with --pmc it initialises the GPU before main): no helper
    processes may be started from here then (the pool refuses an exec from a process that has initialised the GPU)."""
    return 'rocprofiler' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)


class PowerClockSampler(object):
    """Socket power and shader clock of this rank's card while the timed segments run.  The samples come from a HELPER PROCESS
    (a loop around rocm-smi: sysfs reads, no GPU context of its own) that is started at the very top of main(), BEFORE this
    process touches the GPU, and told over a pipe when to sample: nothing is fork+exec'ed from a process that has initialised
    the GPU.  The 2.5 PFLOP/s the roofline divides by assume 2.4 GHz; under this workload the part sits at its board power
    limit and holds 1.5-1.6 GHz.  Purely informative (`held_clock`, outside `roofline`): any failure leaves the fields None."""
    NOMINAL_MHZ = 2400.0

    def __init__(self, card):
        self.samples, self.proc = [], None
        if under_profiler():
            return
        try:
            import subprocess
            self.proc = subprocess.Popen([sys.executable, '-c', SAMPLER_HELPER, 'card%d' % card], stdin=subprocess.PIPE,
                                         stdout=subprocess.PIPE, text=True)
        except Exception:
            self.proc = None

    def __enter__(self):
        try:
            if self.proc is not None:
                self.proc.stdin.write('start\n')
                self.proc.stdin.flush()
        except Exception:
            pass
        return self

    def __exit__(self, *exc):
        try:
            if self.proc is not None:
                out, _ = self.proc.communicate('stop\n', timeout=15)
                self.samples = [tuple(x) for x in json.loads(out.strip().splitlines()[-1])]
        except Exception:
            try:
                self.proc.kill()
            except Exception:
                pass
        self.proc = None
        return False

    def close(self):
        """(a run that never sampled: end the helper)"""
        if self.proc is not None:
            self.__exit__()

    def summary(self):
        busy = [(p, c) for p, c, u in self.samples if u >= 50 and p is not None and c is not None]
        if not busy:
            return None
        ps, cs = sorted(p for p, _ in busy), sorted(c for _, c in busy)
        return {'samples_used': len(busy), 'socket_power_w_median': ps[len(ps) // 2], 'socket_power_w_max': ps[-1],
                'sclk_mhz_median': cs[len(cs) // 2], 'sclk_mhz_min': cs[0], 'sclk_mhz_max': cs[-1],
                'nominal_sclk_mhz': self.NOMINAL_MHZ, 'samples_total': len(self.samples),
                'source': 'rocm-smi --showpower --showclocks --showuse from a helper process started before this one touched '
                          'the GPU, back to back during the timed segments (samples at >= 50 % use)'}


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=40)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--repeats', type=int, default=5, help='independent timed segments of --steps iterations (median reported)')
    # (long forms: torchrun's own parser takes a bare `--n` for an abbreviation of its options)
    p.add_argument('--n', '--rows', dest='n', type=int, default=1000000)
    p.add_argument('--f', '--features', dest='f', type=int, default=4096)
    p.add_argument('--k', '--components', dest='k', type=int, default=200)
    p.add_argument('--precision', default='f16', choices=['f16', 'bf16', 'f32', 'f64'],
                   help="f16: fp16 MFMA operands (scaled images), fp32 accumulate; 'bf16' is the round-1 name of the same mode")
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-rows', type=int, default=100000,
                   help='rows of the same V the CPU baseline is timed on (SURVEY 8d: n = 100 000 at the same f, k, scaled '
                        'linearly in n); more than 65 536, so the parity leg on the same sample runs fp8 ratio tiles and the '
                        'fp8 x fp8 column pass')
    p.add_argument('--cpu-iters', type=int, default=5,
                   help='timed CPU iterations after one warm-up (about 25 s of fp64 work at the default sample); the parity leg '
                        'runs the same 1 + N iterations on the GPU (6 by default: 4 of them on fp8 ratio tiles)')
    p.add_argument('--no-16bit-segment', action='store_true',
                   help='skip the extra segment that measures value_16bit (16-bit ratio tiles, f16 column pass)')
    p.add_argument('--segment-timeout', type=float, default=float(os.environ.get('KLNMF_BENCH_SEGMENT_TIMEOUT', '300')),
                   help='watchdog: seconds one warm-up + timed segment may take before the rank exits with status 3')
    p.add_argument('--seed', type=int, default=1234)
    p.add_argument('--event-every', type=int, default=4,
                   help='HIP events bracket the row-pass and column-pass launches of every N-th timed iteration (the per-kernel '
                        'durations of `roofline`): an event record is a stream packet of its own -- 5-6 us of dispatch gap each, '
                        'four per iteration = 3.5 %% of a 0.65 ms shard iteration when every launch is bracketed (N = 1)')
    p.add_argument('--data', default='blocks', choices=['blocks', 'device'],
                   help="blocks: the seeded RandomState blocks of SURVEY 8d (host-generated, identical to the CPU baseline's "
                        "data); device: a torch generator on the GPU (fast set-up for profiling runs, different values)")
    p.add_argument('--collective', default='native', choices=['native', 'torch'],
                   help='N > 1: native = one grouped RCCL all-reduce per iteration issued inside the C-ABI (klnmf_run_sharded); '
                        'torch = torch.distributed all-reduces sequenced in Python around the C-ABI pieces')
    p.add_argument('--workload', default='fit', choices=['fit', 'transform'],
                   help="fit: the training iteration of nmf.py:212-222 (the headline metric); transform: the iteration of "
                        "KLdivNMF.transform / fit_coefficients (nmf.py:275-291, learner.py:11-15: loss + ratio + W rule against a "
                        "FIXED trained dictionary -- no ratio tiles stored, no H rule, no numerator exchange: next-row N1)")
    p.add_argument('--train-iters', type=int, default=10, help='--workload transform: fit iterations that train the dictionary first')
    p.add_argument('--slice', type=int, default=0,
                   help='--workload transform: coefficients of the first SLICE columns only against the column-sliced dictionary '
                        '(learner.py:67-78: its rows no longer sum to 1); 0 = all columns')
    p.add_argument('--tol', type=float, default=0.0,
                   help='stop-rule tolerance of nmf.py:207,215 (relative; x n x f inside).  0 = MultimodalLearner.train')
    return p.parse_args(argv)


def make_H0(seed, f, k):
    """Row-normalised |U(0,1)| + 0.01 (the reference's init rule, nmf.py:149-151)."""
    from multimodal_amd import synthetic
    return synthetic.H0_of(seed, f, k)


def fill_shard_blocks(model, seed, r0, r1, n, f, k, vscale=1.0):
    """Global rows [r0, r1) of the seeded synthetic V into the model's shard.  Two passes over the block
    generator: the 16-bit storage factor needs the GLOBAL maximum before the first upload."""
    from multimodal_amd import synthetic
    vmax = synthetic.for_each_block(seed, r0, r1, n, f, k, None)
    model.set_v_max(vmax * vscale)
    synthetic.for_each_block(seed, r0, r1, n, f, k,
                             lambda lo, arr: model.upload_V(arr, row0=lo, col0=0, scale=vscale))


def fill_shard_device(torch, model, seed, rank, n_local, f, k, block=8192, vscale=1.0):
    """Synthetic non-negative V = Wt.Ht/k + 0.05*U of the same family, generated block-wise ON the GPU and fed to the
    tiling upload (seconds instead of a minute at 1M rows; used by profiling runs and the full-size property tests).
    `vscale` multiplies the matrix on its way in (the upload's per-modality coefficient, learner.py:53-56)."""
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)                       # Ht identical on every rank
    # Gamma(1/2) = N(0,1)^2 / 2 and Gamma(1) = -log(U): built from randn / rand only, whose streams are reproduced
    # exactly by restoring the generator state (torch._standard_gamma's rejection sampler is not)
    Ht = torch.randn((k, f), device=dev, generator=g).square_().mul_(0.5)
    g.manual_seed(seed + 1000 * (rank + 1))

    def block_of(rows):
        Wt = torch.rand((rows, k), device=dev, generator=g).neg_().add_(1.0).log_().neg_()     # Exp(1)
        Vb = torch.rand((rows, f), device=dev, generator=g).mul_(0.05)
        Vb.addmm_(Wt, Ht, alpha=1.0 / k)
        return Vb
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
    """HBM bytes per launch of the hot kernels from the committed PMC passes of this exact workload
    (profiles/r0N_pmc_traffic.json, produced by scripts/pmc_profile.sh: separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs, gfx950 correction applied); None if the workload differs -- counters cannot be read in-process.
    Returns (row, col, file, source_hash the file was measured at or None)."""
    for name in PMC_TRAFFIC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, name)))
        except Exception:
            continue
        for entry in d.get('workloads', []):
            w = entry.get('workload', {})
            if (w.get('n_local'), w.get('f'), w.get('k'), w.get('precision')) == (n_local, args.f, args.k, args.precision.replace('bf16', 'f16')):
                row = col = None
                for kname, v in entry.get('kernels', {}).items():
                    if 'k_rowpass' in kname and 'column-split' not in kname:      # the whole-row launch (the roofline entry)
                        row = v['hbm_bytes_per_launch']
                    elif 'k_colpass' in kname:      # (the guarded fallback launch behind the fp8 x fp8 pass moves nothing: the larger one)
                        col = max(col or 0.0, v['hbm_bytes_per_launch'])
                MEASURED_ITERATION['bytes'] = entry.get('hbm_bytes_per_iteration')
                return row, col, name, d.get('source_hash')
    return None, None, None, None


MEASURED_ITERATION = {'bytes': None}      # ... and of the whole iteration (every steady-state launch), where the file holds it


def measured_clock(args, n_local):
    """Shader clock the dominant kernel held in the committed PMC pass of this workload: GRBM_GUI_ACTIVE / 8 / duration (MHz)."""
    for name in PMC_TRAFFIC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, name)))
        except Exception:
            continue
        for entry in d.get('workloads', []):
            w = entry.get('workload', {})
            if (w.get('n_local'), w.get('f'), w.get('k'), w.get('precision')) == (n_local, args.f, args.k, args.precision.replace('bf16', 'f16')):
                for kname, v in entry.get('kernels', {}).items():
                    if 'k_rowpass' in kname and 'column-split' not in kname and v.get('sclk_mhz_from_grbm_gui_active'):
                        return v['sclk_mhz_from_grbm_gui_active']
    return None


def host_info():
    info = {'cpu_model': platform.processor() or 'unknown', 'numpy': np.__version__}
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                info['cpu_model'] = line.split(':', 1)[1].strip()
                break
    except Exception:
        pass
    try:
        info['cores'] = len(os.sched_getaffinity(0))
    except Exception:
        info['cores'] = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        info['blas'] = [{k: v for k, v in t.items() if k in ('internal_api', 'version', 'num_threads', 'threading_layer', 'architecture')}
                        for t in threadpool_info() if t.get('user_api') == 'blas']
    except Exception:
        info['blas'] = None
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                info['mem_available_gb'] = int(line.split()[1]) / 1e6
    except Exception:
        pass
    return info


def cpu_baseline(args):
    """The oracle (numpy restatement of the reference loop: fp64, a separate W.H for the loss and for the ratio, all
    temporaries) timed on this host's cores on the first `cpu_rows` rows of the SAME seeded V the GPU run uses, plus
    the "optimised CPU" variant (one W.H, float32).  Work is exactly linear in n, so the sample's rate scales to n."""
    from oracle import klnmf_oracle as orc
    from multimodal_amd import synthetic
    host = host_info()
    # BLAS threads: all the library allows.  The numpy wheel's OpenBLAS is BUILT for at most 64 threads (NUM_THREADS=64): on the
    # GPU box's 2 x 64-core host that is half the physical cores -- said in the line (`blas_threads_note`), not hidden.
    physical = host.get('cores')
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or physical
    except Exception:
        pass
    blas_threads = max([t.get('num_threads', 0) for t in (host.get('blas') or [])] + [0]) or None
    rows, f, k = min(args.cpu_rows, args.n), args.f, args.k
    # the faithful restatement holds ~6 n x f float64 arrays: shrink the sample rather than risk the box
    need_gb = 7 * rows * f * 8 / 1e9
    avail = host.get('mem_available_gb')
    if avail is not None and need_gb > 0.5 * avail:
        rows = max(8192, int(rows * 0.5 * avail / need_gb) // 8192 * 8192)
    X = synthetic.rows_of(args.seed, 0, rows, args.n, f, k)
    H0 = make_H0(args.seed, f, k)
    W, H = orc.init_factors(X, k, H0=H0)
    losses, times = [], []
    for it in range(1 + args.cpu_iters):              # the first iteration is the warm-up (page faults, BLAS threads)
        t0 = time.perf_counter()
        losses.append(orc.kl_error(X, W, H))
        W, H = orc.update_step(X, W, H, fit=True)
        times.append(time.perf_counter() - t0)
    final = orc.kl_error(X, W, H)
    per_iter = statistics.median(times[1:])
    X32 = X.astype(np.float32)
    W32, H32 = (a.astype(np.float32) for a in orc.init_factors(X, k, H0=H0))
    t32 = []
    for it in range(1 + args.cpu_iters):
        t0 = time.perf_counter()
        _, W32, H32 = orc.fit_iteration_lean32(X32, W32, H32)
        t32.append(time.perf_counter() - t0)
    lean = statistics.median(t32[1:])
    del X32, W32, H32
    base = {
        'value': (1.0 / per_iter) * rows / args.n,
        'unit': 'it/s',
        'cores': blas_threads or host.get('cores'),
        'cores_what': 'BLAS threads the timed run used (numpy element-wise passes: 1 thread); the host shows %s hardware threads, '
                      '%s physical cores' % (host.get('cores'), physical),
        'blas_threads_note': (None if not blas_threads or not physical or blas_threads >= physical else
                              'OpenBLAS of this numpy build caps at %d threads (its compile-time NUM_THREADS); the host has %d '
                              'physical cores' % (blas_threads, physical)),
        'kind': 'port',
        'sample': 'rows [0, %d) of the same seeded %d x %d V, k=%d: 1 warm-up + %d timed fp64 iterations of the numpy '
                  'restatement of nmf.py:212-222 (median %.2f s/iteration on the sample), scaled linearly in n (extrapolated)'
                  % (rows, args.n, f, k, args.cpu_iters, per_iter),
        'cpu_model': host.get('cpu_model'), 'numpy': host.get('numpy'), 'blas': host.get('blas'),
        'optimised_cpu': {'value': (1.0 / lean) * rows / args.n, 'unit': 'it/s',
                          'what': 'one W.H per iteration, float32, temporaries reused (oracle.fit_iteration_lean32), '
                                  'median %.2f s/iteration on the same sample' % lean},
    }
    return base, (X, H0, len(times), losses, final)


def gpu_parity_on_sample(args, sample):
    """The CPU baseline's row sample through the HIP path: every recorded loss and the final KL against the oracle."""
    from multimodal_amd import _native
    X, H0, iters, losses, final = sample
    with _native.Context(args.precision, device=0) as ctx:
        ctx.set_problem(X.shape[0], X.shape[1], args.k, iters)
        ctx.upload_blocks([X])
        ctx.set_H(H0)
        ctx.init_W()
        errs, n_done, stopped = ctx.run(iters, True, 0.0)
        fp8 = ctx.fp8_report()
        g_final = ctx.error()
    return {'final_kl_rel_err': abs(g_final - final) / abs(final),
            'max_loss_rel_err': float(max(abs(a - b) / abs(b) for a, b in zip(errs, losses))) if len(errs) == len(losses) else None,
            'len_errors': [int(len(errs)), int(len(losses))],
            'iterations': iters, 'rows': int(X.shape[0]), 'tolerance': 1e-4,
            'iterations_on_fp8_ratio_tiles': fp8['tile_iterations'], 'iterations_with_fp8_x_fp8_column_pass': fp8['column_pass_iterations']}


def run_transform(args, torch, dist, model, H0, world, rank, n_local, r0, r1, fence, info, t_setup, collective, sampler0):
    """--workload transform: `steps` iterations of KLdivNMF.transform's loop (nmf.py:275-291 -> 212-222 with _fit=False) on the
    resident V against a dictionary trained by `train_iters` fit iterations first.  One row pass per iteration (W.H -> ratio,
    loss -> Q.H^T -> W rule: 4 n f k flops, V read once, the fp32 master of W in and out) + the loss reduction; nothing is
    stored for an H rule and nothing but the loss scalar is exchanged between ranks."""
    from multimodal_amd.distributed import ShardedKLNMF
    n, f, k = args.n, args.f, args.k
    iters_per_fit = args.warmup + args.steps
    # ---- the dictionary: `train_iters` fit iterations from H0
    if args.train_iters > 0:
        model.set_H(H0)
        model.init_W()
        model.begin()
        model.iterate_many(args.train_iters, fit=True, tol=0.0)
        model.end()
        H = model.get_H()
    else:
        H = np.asarray(H0, dtype=np.float64)           # (PMC passes: no fit launches among the counted ones)
    tmodel, fs = model, f
    if args.slice and args.slice < f:
        # the first `slice` columns against the column-sliced dictionary (modality -> internal: learner.py:67-78)
        from multimodal_amd import synthetic
        fs = args.slice
        H = np.ascontiguousarray(H[:, :fs])
        tmodel = ShardedKLNMF(n, n_local, fs, k, max_iter=iters_per_fit, precision=args.precision, collective=collective)
        vmax = synthetic.for_each_block(args.seed, r0, r1, n, f, k, None)
        tmodel.set_v_max(vmax)
        synthetic.for_each_block(args.seed, r0, r1, n, f, k,
                                 lambda lo, arr: tmodel.upload_V(np.ascontiguousarray(arr[:, :fs]), row0=lo, col0=0))
    segments, prof_tot, last, tail_rows = [], {'rowpass_ms': 0.0, 'rowpass_launches': 0}, None, 0
    for rep in range(max(1, args.repeats)):
        with Watchdog(args.segment_timeout, 'transform segment %d' % rep, rank):
            tmodel.set_H(H)
            tmodel.init_W()
            tmodel.begin()
            tmodel.iterate_many(args.warmup, fit=False, tol=args.tol)
            tmodel.ctx.profile_enable(args.event_every)
            fence()
            t0 = time.perf_counter()
            tmodel.iterate_many(args.steps, fit=False, tol=args.tol)
            fence()
            elapsed = time.perf_counter() - t0
            prof = tmodel.ctx.profile_read(reset=True)
            tail_rows = prof.get('tail_rows', 0) if prof.get('tail_launches', 0) else 0
            tmodel.ctx.profile_enable(False)
            last = tmodel.end()
            t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            segments.append(float(t.item()))
        for key in prof_tot:
            prof_tot[key] += prof[key]
    if rank != 0:
        if tmodel is not model:
            tmodel.close()
        return None
    n_gpus = world if world > 1 else 1
    elapsed = statistics.median(segments)
    its = args.steps / elapsed
    row_ms = prof_tot['rowpass_ms'] / max(1, prof_tot['rowpass_launches'])
    fast16 = args.precision in ('f16', 'bf16')
    vbytes = 2 if fast16 else 4
    flops_row = 4.0 * n_local * fs * k
    alg_bytes_row = n_local * fs * vbytes + 2 * n_local * k * 4
    sched_bytes_row = alg_bytes_row + 2 * n_local * k * 2          # + the 16-bit W images in and out
    t_mfma, t_hbm = flops_row / (PEAK_BF16_TFLOPS * 1e12), alg_bytes_row / (PEAK_HBM_GBS * 1e9)
    mfma_bound = t_mfma >= t_hbm
    row_s = row_ms * 1e-3
    tfl, gbs = flops_row / row_s / 1e12, alg_bytes_row / row_s / 1e9
    traffic = None
    for name in [os.path.join('profiles', 'r05_pmc_traffic_transform.json'), os.path.join('profiles', 'r06_pmc_traffic_transform.json')]:      # (the later file wins)
        try:
            d = json.load(open(os.path.join(ROOT, name)))
            for entry in d.get('workloads', []):
                w = entry.get('workload', {})
                if (w.get('n_local'), w.get('f'), w.get('k')) == (n_local, fs, k):
                    traffic = {'bytes_per_launch': entry['hbm_bytes_per_launch'], 'source': name, 'source_hash': d.get('source_hash')}
        except Exception:
            pass
    errors, n_done, stopped = last
    out = {
        'metric': 'nmf_transform_iterations_per_sec',
        'value': its, 'unit': 'it/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f16 operands, fp32 accumulate' if fast16 else args.precision,
        'data': 'synthetic',
        'config': {'workload': 'KL-NMF transform iteration (fixed dictionary), V %dx%d (row-sharded), k=%d%s' % (
                       n, fs, k, ', first %d of %d columns against the column-sliced dictionary' % (fs, f) if fs != f else ''),
                   'n': n, 'f': fs, 'k': k, 'rows_per_gpu': n_local, 'precision': args.precision, 'parallelism': 'rows/%d' % n_gpus,
                   'dictionary': '%d fit iterations from the seeded H0%s' % (
                       args.train_iters, '; rows of the slice sum to %.3f .. %.3f' % (H.sum(axis=1).min(), H.sum(axis=1).max()) if fs != f else ''),
                   'timing': 'median of %d segments of %d iterations, each after a fresh W0 = V.H^T + %d warm-up iterations'
                             % (len(segments), args.steps, args.warmup)},
        'samples_per_sec': its * n,
        'segments_ms_per_step': [1e3 * sg / args.steps for sg in segments],
        'setup_s': t_setup, 'iterations_done': n_done, 'stopped_early': bool(stopped),
        'valid': bool(n_done == iters_per_fit and not stopped),
        'loss_first': errors[0] if errors else None, 'loss_last': errors[-1] if errors else None,
        'loss_finite_and_decreasing': bool(len(errors) > 1 and all(b < a for a, b in zip(errors, errors[1:]))),
        'device': info,
        'roofline': {
            'kernel': 'k_rowpass4 (W.H -> ratio/loss -> Q.H^T -> W rule; no ratio tiles stored)',
            'bound': 'mfma' if mfma_bound else 'hbm',
            'achieved': tfl if mfma_bound else gbs, 'peak': PEAK_BF16_TFLOPS if mfma_bound else PEAK_HBM_GBS,
            'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
            'frac': (tfl / PEAK_BF16_TFLOPS) if mfma_bound else (gbs / PEAK_HBM_GBS),
            'traffic': traffic['bytes_per_launch'] if traffic else None,
            'traffic_source': traffic['source'] if traffic else None,
            'traffic_is_current': bool(traffic and traffic.get('source_hash') == kernel_source_hash()),
            # (the PMC figure is the WHOLE-ROW launch's: where the last partial round of workgroups runs column-split, it covers
            # n_local - tail_rows rows, and so does the algorithmic figure it is divided by -- ADVICE round 5)
            'traffic_rows': n_local - tail_rows,
            'traffic_over_algorithmic': (traffic['bytes_per_launch'] / ((n_local - tail_rows) * fs * vbytes + 2 * (n_local - tail_rows) * k * 4))
                                        if traffic else None,
            'source_hash': kernel_source_hash(),
            'avg_launch_ms': row_ms, 'launches': prof_tot['rowpass_launches'], 'rows_per_launch': n_local,
            'algorithmic_flops_per_launch': flops_row, 'algorithmic_bytes_per_launch': alg_bytes_row,
            'schedule_bytes_per_launch': sched_bytes_row,
            't_min_ms': {'mfma': t_mfma * 1e3, 'hbm': t_hbm * 1e3},
            'other_roof': {'bound': 'hbm' if mfma_bound else 'mfma', 'achieved': gbs if mfma_bound else tfl,
                           'frac': (gbs / PEAK_HBM_GBS) if mfma_bound else (tfl / PEAK_BF16_TFLOPS)}},
        'kernels': {'iteration_algorithmic_tflops': 4.0 * n * fs * k / (1e3 * elapsed / args.steps * 1e-3) / 1e12,
                    'rest_of_the_iteration_ms': 1e3 * elapsed / args.steps - row_ms},
    }
    if n_gpus == 1 and not args.no_cpu_baseline:
        out.update(transform_cpu_baseline_and_parity(args, H, fs))
    if tmodel is not model:
        tmodel.close()
    return out


def transform_cpu_baseline_and_parity(args, H, fs):
    """The oracle's transform loop (numpy restatement of nmf.py:275-291: fp64, a separate W.H for the loss and for the ratio)
    timed on the first `cpu_rows` rows of the same V against the SAME dictionary, and the HIP path on that sample."""
    from oracle import klnmf_oracle as orc
    from multimodal_amd import synthetic, _native
    host = host_info()
    blas_threads = max([t.get('num_threads', 0) for t in (host.get('blas') or [])] + [0]) or None
    rows = min(args.cpu_rows, args.n)
    X = np.ascontiguousarray(synthetic.rows_of(args.seed, 0, rows, args.n, args.f, args.k)[:, :fs])
    H = np.asarray(H, dtype=np.float64)
    W = X.dot(H.T)
    losses, times = [], []
    for it in range(1 + args.cpu_iters):
        t0 = time.perf_counter()
        losses.append(orc.kl_error(X, W, H))
        W, _ = orc.update_step(X, W, H, fit=False)
        times.append(time.perf_counter() - t0)
    final = orc.kl_error(X, W, H)
    per_iter = statistics.median(times[1:])
    base = {'value': (1.0 / per_iter) * rows / args.n, 'unit': 'it/s', 'cores': blas_threads or host.get('cores'), 'kind': 'port',
            'sample': 'rows [0, %d) of the same seeded V (%d columns), k=%d, the trained dictionary: 1 warm-up + %d timed fp64 '
                      'iterations of the numpy restatement of the transform loop (median %.2f s/iteration on the sample), scaled '
                      'linearly in n (extrapolated)' % (rows, fs, args.k, args.cpu_iters, per_iter),
            'cpu_model': host.get('cpu_model'), 'numpy': host.get('numpy'), 'blas': host.get('blas')}
    iters = len(losses)
    with _native.Context(args.precision, device=0) as ctx:
        ctx.set_problem(rows, fs, args.k, iters)
        ctx.upload_blocks([X])
        ctx.set_H(H)
        ctx.init_W()
        errs, n_done, stopped = ctx.run(iters, False, 0.0)
        g_final = ctx.error()
        Wg = ctx.get_W()
    parity = {'final_kl_rel_err': abs(g_final - final) / abs(final),
              'max_loss_rel_err': float(max(abs(a - b) / abs(b) for a, b in zip(errs, losses))) if len(errs) == len(losses) else None,
              'W_max_abs_err_over_max': float(np.abs(Wg - W).max() / np.abs(W).max()),
              'len_errors': [int(len(errs)), int(len(losses))], 'iterations': iters, 'rows': int(rows), 'tolerance': 1e-4}
    return {'cpu_baseline': base, 'parity': parity}


def main():
    args = parse_args()
    # stdout carries ONE JSON line and nothing else: libraries write there too (RCCL prints a version banner at the first
    # communicator of a process, to fd 1), so everything before the line goes to stderr -- fd 1 is parked on fd 2 until rank 0
    # prints its result
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # (the power / clock helper is a child process: it must exist before anything here initialises the GPU)
    sampler0 = PowerClockSampler(int(os.environ.get('LOCAL_RANK', '0'))) if int(os.environ.get('RANK', '0')) == 0 else None
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
    r0, r1 = row_partition(n, n_gpus)[rank]
    n_local = r1 - r0
    iters_per_fit = args.warmup + args.steps
    t_setup = time.perf_counter()
    # (KLNMF_COMM_SINGLE=1: one process rehearses the native path on a one-rank communicator -- the library then runs the
    # collective branch, all-reduces included: tests/test_distributed_gpu.py)
    comm_single = world == 1 and os.environ.get('KLNMF_COMM_SINGLE') == '1'
    collective = args.collective if ((world > 1 and not rehearsal) or comm_single) else 'torch'
    native_error = None
    model = None
    with Watchdog(args.segment_timeout, 'set-up of the model and its communicator', rank):
        try:
            model = ShardedKLNMF(n, n_local, f, k, max_iter=iters_per_fit, precision=args.precision, collective=collective)
        except Exception as e:           # librccl not loadable, communicator refused, ...
            if collective != 'native':
                raise
            native_error = e
        if collective == 'native' and world > 1:
            # ALL ranks switch together or none does: a rank that fell back alone would wait in a torch collective while
            # the others wait in RCCL
            flag = torch.tensor([1.0 if native_error is not None else 0.0], device='cuda')
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag.item()) != 0.0:
                if rank == 0:
                    sys.stderr.write('bench.py: native collective path unavailable on at least one rank (%s); every rank uses '
                                     'torch.distributed\n' % (native_error if native_error is not None else 'another rank'))
                if model is not None:
                    model.close()
                collective = 'torch'
                model = ShardedKLNMF(n, n_local, f, k, max_iter=iters_per_fit, precision=args.precision, collective=collective)
        elif native_error is not None:
            raise native_error
    if args.data == 'blocks':
        fill_shard_blocks(model, args.seed, r0, r1, n, f, k)
    else:
        fill_shard_device(torch, model, args.seed, rank, n_local, f, k)
    H0 = make_H0(args.seed, f, k)
    info = _native.device_info(local_rank)
    t_setup = time.perf_counter() - t_setup

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == 'transform':
        out = run_transform(args, torch, dist, model, H0, world, rank, n_local, r0, r1, fence, info, t_setup, collective, sampler0)
        if rank == 0:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
            print(json.dumps(out))
            sys.stdout.flush()
            os.dup2(2, 1)
        model.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    def run_segments(repeats):
        """`repeats` independent fits from the same start; returns (segment times, fits, summed kernel profile, tail rows,
        what the last fit's loop ran on fp8)."""
        segments, fits = [], []
        prof_tot = {'rowpass_ms': 0.0, 'rowpass_launches': 0, 'colpass_ms': 0.0, 'colpass_launches': 0, 'tail_ms': 0.0,
                    'tail_launches': 0}
        tail_rows, fp8 = 0, None
        for rep in range(max(1, repeats)):
            with Watchdog(args.segment_timeout, 'segment %d (init + %d warm-up + %d timed iterations, collective=%s)'
                          % (rep, args.warmup, args.steps, collective), rank):
                model.set_H(H0)
                model.init_W()
                # ONE loop on every path: warm-up iterations | fence | timed iterations | fence.  One process and the native
                # collective path: klnmf_run_more (the launches klnmf_run / klnmf_run_sharded enqueue, the grouped RCCL
                # all-reduce of every iteration included); torch path: the loop's pieces around torch.distributed.
                model.begin()
                model.iterate_many(args.warmup, fit=True, tol=args.tol)
                model.ctx.profile_enable(args.event_every)
                fence()
                t0 = time.perf_counter()
                model.iterate_many(args.steps, fit=True, tol=args.tol)
                fence()
                elapsed = time.perf_counter() - t0
                prof = model.ctx.profile_read(reset=True)
                model.ctx.profile_enable(False)
                errors, n_done, stopped = model.end()
                t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                segments.append(float(t.item()))
            fits.append((list(errors), int(n_done), bool(stopped)))
            for key in prof_tot:
                prof_tot[key] += prof[key]
            tail_rows = prof.get('tail_rows', 0)
            fp8 = model.ctx.fp8_report()
        return segments, fits, prof_tot, tail_rows, fp8

    if rank == 0 and sampler0 is not None:          # (one sampler per job: rank 0's card)
        with sampler0 as sampler:
            segments, fits, prof_tot, tail_rows, fp8 = run_segments(args.repeats)
        power_clock = sampler.summary()
    else:
        segments, fits, prof_tot, tail_rows, fp8 = run_segments(args.repeats)
        power_clock = None
    # The same workload with 16-bit ratio tiles and the f16-operand column pass (KLNMF_QTILE=16 is read at every loop's
    # entry): the number the north star's "bf16/16-bit MFMA contractions" wording describes, beside the headline one.
    value_16bit = None
    fast16 = args.precision in ('f16', 'bf16')
    if fast16 and not args.no_16bit_segment and fp8 is not None and fp8['tile_iterations'] > 0:
        saved = os.environ.get('KLNMF_QTILE')
        os.environ['KLNMF_QTILE'] = '16'
        try:
            seg16, fits16, prof16, _, fp8_16 = run_segments(1)
        finally:
            if saved is None:
                del os.environ['KLNMF_QTILE']
            else:
                os.environ['KLNMF_QTILE'] = saved
        value_16bit = {
            'value': args.steps / seg16[0], 'unit': 'it/s', 'ms_per_step': 1e3 * seg16[0] / args.steps,
            'what': 'one extra segment of the same run with KLNMF_QTILE=16: 16-bit (fp16) ratio tiles and f16 operands in '
                    'every product, no e4m3 anywhere',
            'row_pass_ms': prof16['rowpass_ms'] / max(1, prof16['rowpass_launches']),
            'tail_ms': (prof16['tail_ms'] / prof16['tail_launches']) if prof16['tail_launches'] else 0.0,
            'col_pass_ms': prof16['colpass_ms'] / max(1, prof16['colpass_launches']),
            'fp8_tile_iterations': fp8_16['tile_iterations'],
            'valid': all(nd == iters_per_fit and not st for _, nd, st in fits16)}

    if rank == 0:
        elapsed = statistics.median(segments)
        ms_per_step = 1e3 * elapsed / args.steps
        its = args.steps / elapsed
        row_ms = prof_tot['rowpass_ms'] / max(1, prof_tot['rowpass_launches'])
        col_ms = prof_tot['colpass_ms'] / max(1, prof_tot['colpass_launches'])
        fast16 = args.precision in ('f16', 'bf16')
        vbytes = 2 if fast16 else 4
        stored_q = fast16 and k <= 512          # the 16-bit mode keeps the ratio tiles of its row pass for the column pass
        # what the timed loop ran, as the LIBRARY reports it (klnmf_query), not a host-side copy of its rules: the loop's
        # first two iterations keep 16-bit tiles; warm-up and timed iterations are ONE loop on every path, so the warm-up
        # absorbs them when --warmup >= 2
        fp8_iters = col8_iters = 0
        if fp8 and stored_q:      # the 16-bit iterations come first
            fp8_iters = max(0, min(args.steps, fp8['tile_iterations'] - max(0, args.warmup - (iters_per_fit - fp8['tile_iterations']))))
            col8_iters = max(0, min(args.steps, fp8['column_pass_iterations'] - max(0, args.warmup - (iters_per_fit - fp8['column_pass_iterations']))))
        frac8 = fp8_iters / float(args.steps)
        qbytes = (2.0 - frac8) if stored_q else 0          # bytes per element of V, averaged over the timed iterations
        col8 = col8_iters > 0                                # fp8 x fp8 column pass (colq8x.hip.h) in the timed region
        frac_col8 = col8_iters / float(args.steps)
        # ---- the row-pass launch (W.H -> ratio, loss -> Q.H^T -> W rule): SURVEY 8d per-launch figures.
        # Hybrid update pass: the dominant kernel is the whole-row k_rowpass4 over the full rounds of
        # workgroups (n_row rows); the column-split last partial round + its slab W rule are reported beside it.
        section_ms = row_ms
        hybrid = prof_tot['tail_launches'] > 0 and prof_tot['tail_launches'] == prof_tot['rowpass_launches']
        n_row = n_local
        tail_ms = 0.0
        if hybrid:
            n_pad = (n_local + 63) // 64 * 64
            n_row = n_pad - tail_rows                       # all of them valid rows: the padding sits in the tail
            tail_ms = prof_tot['tail_ms'] / prof_tot['tail_launches']
            row_ms = section_ms - tail_ms
        flops_row = 4.0 * n_row * f * k
        alg_bytes_row = n_row * f * vbytes + 2 * n_row * k * 4
        sched_bytes_row = alg_bytes_row + 2 * n_row * k * 2 + n_row * f * qbytes      # + 16-bit W images in/out + ratio tiles out
        sched_bytes_section = sched_bytes_row * n_local / n_row
        flops_col = 2.0 * n_local * f * k
        sched_bytes_col = (n_local * f * qbytes + n_local * k * (1 if col8 else 2)) if stored_q else (n_local * f * vbytes + 2 * n_local * k * 2)
        t_mfma = flops_row / (PEAK_BF16_TFLOPS * 1e12)
        t_hbm = alg_bytes_row / (PEAK_HBM_GBS * 1e9)
        mfma_bound = t_mfma >= t_hbm
        row_s = row_ms * 1e-3
        row_tflops = flops_row / row_s / 1e12 if row_s > 0 else None
        row_gbs = alg_bytes_row / row_s / 1e9 if row_s > 0 else None
        traffic_row, traffic_col, traffic_file, traffic_hash = measured_traffic(args, n_local)
        src_hash = kernel_source_hash()
        roofline = {
            'kernel': 'k_rowpass4' + ' (W.H -> ratio/loss -> Q.H^T -> W rule'
                      + (', ratio tiles stored for the H rule)' if stored_q else ')'),
            'bound': 'mfma' if mfma_bound else 'hbm',
            'achieved': row_tflops if mfma_bound else row_gbs,
            'peak': PEAK_BF16_TFLOPS if mfma_bound else PEAK_HBM_GBS,
            'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
            'frac': ((row_tflops / PEAK_BF16_TFLOPS) if mfma_bound else (row_gbs / PEAK_HBM_GBS)) if row_s > 0 else None,
            'traffic': traffic_row,
            'traffic_source': traffic_file if traffic_row else None,
            'traffic_measured_at_source_hash': traffic_hash if traffic_row else None,
            'source_hash': src_hash,
            'traffic_is_current': bool(traffic_row and traffic_hash == src_hash),
            'operands': 'f16 (v_mfma_f32_32x32x16_f16), fp32 accumulate' if fast16 else args.precision,
            'avg_launch_ms': row_ms, 'launches': prof_tot['rowpass_launches'], 'rows_per_launch': n_row,
            'launches_timed_by': 'HIP events on the kernel\'s stream around every %s launch of the timed region' % (
                'single' if args.event_every <= 1 else '%d-th' % args.event_every),
            'algorithmic_flops_per_launch': flops_row,
            'algorithmic_bytes_per_launch': alg_bytes_row,
            'schedule_bytes_per_launch': sched_bytes_row,
            'traffic_over_algorithmic': traffic_row / alg_bytes_row if traffic_row else None,
            't_min_ms': {'mfma': t_mfma * 1e3, 'hbm': t_hbm * 1e3},
            'other_roof': {'bound': 'hbm' if mfma_bound else 'mfma',
                           'achieved': row_gbs if mfma_bound else row_tflops,
                           'frac': (row_gbs / PEAK_HBM_GBS if mfma_bound else row_tflops / PEAK_BF16_TFLOPS) if row_s > 0 else None},
            'schedule_hbm_gbs': sched_bytes_row / row_s / 1e9 if row_s > 0 else None,
        }
        # `frac` divides by the nominal peak (2.4 GHz) and is the contract's number.  Informative, and kept OUT of `roofline`: the
        # same kernel against the MFMA peak at the clock the card held under its power limit -- (a) from the committed PMC pass of
        # this workload (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration: an in-run cycle count, what the microarchitecture guide
        # accepts), (b) from rocm-smi samples taken during the timed segments (sysfs clocks read up to 10 % high)
        held = {'nominal_sclk_mhz': PowerClockSampler.NOMINAL_MHZ}
        pmc_mhz = measured_clock(args, n_local)
        if pmc_mhz and mfma_bound and roofline['frac']:
            held['sclk_mhz_from_pmc_GRBM_GUI_ACTIVE'] = pmc_mhz
            held['row_pass_frac_at_pmc_clock'] = roofline['frac'] * PowerClockSampler.NOMINAL_MHZ / pmc_mhz
        if power_clock and mfma_bound and roofline['frac']:
            held['sclk_mhz_rocm_smi_median'] = power_clock['sclk_mhz_median']
            held['rocm_smi_samples'] = power_clock['samples_used']
            held['row_pass_frac_at_rocm_smi_clock'] = roofline['frac'] * PowerClockSampler.NOMINAL_MHZ / power_clock['sclk_mhz_median']
        if value_16bit is not None and value_16bit.get('row_pass_ms'):
            # the all-16-bit run's own roofline entry (its whole-row launch writes 2 B instead of 1 B per element of V and keeps
            # the numerator's eps): the figure that goes with `value_16bit`
            r16 = value_16bit['row_pass_ms'] - (value_16bit.get('tail_ms') or 0.0)
            value_16bit['roofline'] = {'bound': 'mfma' if mfma_bound else 'hbm', 'avg_launch_ms': r16, 'rows_per_launch': n_row,
                                       'achieved': (flops_row / (r16 * 1e-3) / 1e12) if mfma_bound else (alg_bytes_row / (r16 * 1e-3) / 1e9),
                                       'peak': PEAK_BF16_TFLOPS if mfma_bound else PEAK_HBM_GBS,
                                       'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
                                       'frac': ((flops_row / (r16 * 1e-3) / 1e12) / PEAK_BF16_TFLOPS) if mfma_bound
                                               else ((alg_bytes_row / (r16 * 1e-3) / 1e9) / PEAK_HBM_GBS)}
        it_t_mfma = (4.0 * n * f * k / (PEAK_BF16_TFLOPS * 1e12)
                     + 2.0 * n * f * k * (frac_col8 / (PEAK_FP8_TFLOPS * 1e12) + (1 - frac_col8) / (PEAK_BF16_TFLOPS * 1e12))) / n_gpus
        it_t_hbm = (n * f * vbytes + 2 * n * k * 4 + 3 * k * f * 4) / (PEAK_HBM_GBS * 1e9) / n_gpus
        errors, n_done, stopped = fits[-1]
        all_full = all(nd == iters_per_fit and not st for _, nd, st in fits)
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
            # the arithmetic types the path computed in (not a precision claim): the W.H and Q.H^T contractions, the W rule
            # and the loss always on f16 operands with fp32 accumulation; the H-numerator product W_new^T.Q on e4m3
            # operands (ratio x sqrt(2) / 8 tiles, scaled W image, both stochastically rounded) in the iterations the library reports
            'dtype': (('f16 operands, fp32 accumulate; e4m3 ratio tiles and e4m3 W image in the H-numerator product'
                       if col8 else ('f16 operands, fp32 accumulate; e4m3 ratio tiles (converted back to f16) in the H-numerator product'
                                     if frac8 > 0 else 'f16'))
                      if args.precision in ('f16', 'bf16') else args.precision),
            'data': 'synthetic',
            'config': {'workload': 'KL-NMF fit iteration, V %dx%d (row-sharded), k=%d' % (n, f, k),
                       'n': n, 'f': f, 'k': k, 'rows_per_gpu': n_local,
                       'precision': args.precision + (' + e4m3 H-numerator product' if col8 else ''),
                       'parallelism': 'rows/%d' % n_gpus,
                       'fp8': {'loop_allowed': bool(fp8 and fp8['allowed']),
                               'timed_iterations_with_fp8_ratio_tiles': fp8_iters,
                               'timed_iterations_with_fp8_x_fp8_column_pass': col8_iters,
                               # e4m3 saturation of the last timed loop: counted and kept out of the result (include/klnmf.h)
                               'ratio_without_numerator_eps': fp8.get('no_numerator_eps') if fp8 else None,
                               'w_image_saturated_entries': fp8['w_image_saturated'] if fp8 else None,
                               'w_image_fallback_iterations': fp8['w_image_fallback_iterations'] if fp8 else None,
                               'ratio_entries_saturated': fp8['ratio_saturated'] if fp8 else None,
                               'ratio_entries_unfixed': fp8['ratio_unfixed'] if fp8 else None,
                               # the in-loop monitor of the last timed loop (each check is followed by a poll = one host
                               # synchronisation inside the timed region)
                               'monitor_checks': fp8['monitor_checks'] if fp8 else None,
                               'monitor_trips': fp8['monitor_trips'] if fp8 else None,
                               'gave_up': fp8['gave_up'] if fp8 else None,
                               'source': 'klnmf_query'},
                       'rccl_ranks': model.rccl_ranks() if (n_gpus > 1 or comm_single) else None,
                       'collective_path': collective if (n_gpus > 1 or comm_single) else None,
                       'collective': ('one grouped RCCL all-reduce of the k x f numerator + the loss per iteration, issued inside '
                                      'the C-ABI (klnmf_run_sharded)' if collective == 'native' else
                                      'torch.distributed all-reduce of the k x f numerator + async all-reduce of the loss')
                                     if n_gpus > 1 else None,
                       'generator': 'seeded RandomState row blocks (SURVEY 8d), seed %d' % args.seed if args.data == 'blocks'
                                    else 'torch generator on the device, seed %d' % args.seed,
                       'timing': 'median of %d segments of %d iterations, each after a fresh init + %d warm-up iterations'
                                 % (len(segments), args.steps, args.warmup)},
            'samples_per_sec': its * n,
            'value_16bit': value_16bit,
            'segments_ms_per_step': [1e3 * s / args.steps for s in segments],
            'setup_s': t_setup,
            'iterations_done': n_done,
            'stopped_early': bool(stopped),
            'valid': bool(all_full),        # every timed launch of every segment did its full work
            'loss_first': errors[0] if errors else None,
            'loss_last': errors[-1] if errors else None,
            'stop_rule': 'tol=%g (nmf.py:207,215%s)' % (args.tol, '; learner.py:39-40' if args.tol == 0 else ''),
            'loss_finite_and_decreasing': bool(len(errors) > 1 and all(e == e and abs(e) != float('inf') for e in errors)
                                               and all(b < a for a, b in zip(errors, errors[1:]))),
            'device': info,
            'roofline': roofline,
            'power_clock': power_clock,
            'held_clock': held,
            'kernels': {
                'row_pass_section': {
                    'what': ('whole-row k_rowpass4 over %d rows (the roofline entry) + column-split k_rowpass4 over the last %d '
                             'rows + k_wrule_slabs on those rows' % (n_row, n_local - n_row)) if hybrid
                            else 'one launch (the roofline entry)',
                    'avg_ms': section_ms, 'tail_avg_ms': tail_ms if hybrid else None,
                    'tail_rows': n_local - n_row if hybrid else 0,
                    'algorithmic_tflops': 4.0 * n_local * f * k / (section_ms * 1e-3) / 1e12 if section_ms > 0 else None,
                    'frac_of_bf16_peak': 4.0 * n_local * f * k / (section_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if section_ms > 0 else None},
                (('k_colpass_q8x' if col8 else 'k_colpass_q2') if stored_q else 'k_colpass'): {
                    'avg_launch_ms': col_ms, 'launches': prof_tot['colpass_launches'],
                    'algorithmic_tflops': flops_col / (col_ms * 1e-3) / 1e12 if col_ms > 0 else None,
                    'operands': 'e4m3 x e4m3 (v_mfma_scale_f32_32x32x64_f8f6f4)' if col8 else 'f16',
                    'peak_tflops_of_its_operand_type': PEAK_FP8_TFLOPS if col8 else PEAK_BF16_TFLOPS,
                    'frac_of_own_peak': flops_col / (col_ms * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if col8 else PEAK_BF16_TFLOPS) if col_ms > 0 else None,
                    'schedule_bytes_per_launch': sched_bytes_col,
                    'schedule_hbm_gbs': sched_bytes_col / (col_ms * 1e-3) / 1e9 if col_ms > 0 else None,
                    'traffic': traffic_col},
                'iteration_algorithmic_bytes': n * f * vbytes + 2 * n * k * 4 + 3 * k * f * 4,
                'iteration_schedule_bytes': (sched_bytes_section + sched_bytes_col) * n_gpus,
                'iteration_algorithmic_tflops': 6.0 * n * f * k / (ms_per_step * 1e-3) / 1e12,
                # t_min of the iteration = max(MFMA, HBM).  MFMA, dtype-true: 4nfk on 16-bit operands + 2nfk on the operands
                # the column pass ran; HBM: the algorithmic minimum bytes above at the nominal 8 TB/s
                'iteration_t_min_ms': {'mfma': 1e3 * it_t_mfma, 'hbm': 1e3 * it_t_hbm},
                'iteration_bound': 'mfma' if it_t_mfma >= it_t_hbm else 'hbm',
                'iteration_frac': max(it_t_mfma, it_t_hbm) / (ms_per_step * 1e-3),
                # every steady-state launch of one iteration, PMC (FETCH_SIZE x 2 + WRITE_SIZE; the committed pass of this workload)
                'iteration_traffic': MEASURED_ITERATION['bytes'],
                'iteration_traffic_over_algorithmic': (MEASURED_ITERATION['bytes'] * n_gpus / (n * f * vbytes + 2 * n * k * 4 + 3 * k * f * 4))
                                                      if MEASURED_ITERATION['bytes'] else None,
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            base, sample = cpu_baseline(args)
            out['cpu_baseline'] = base
            try:
                out['parity'] = gpu_parity_on_sample(args, sample)
            except Exception as e:   # parity is reported, never hidden
                out['parity'] = {'error': str(e)}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)
    model.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
