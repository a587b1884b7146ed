"""Bug hunt, part 8: the row-sharded driver (torch path, gloo exchange, every rank on GPU 0) over shapes that shard badly: rows not
divisible by the ranks, shards of one row, shards around the 32-row tile and the fp8 thresholds, k = 1, wide f / k (ratio scale),
sparse shards (one rank's rows all zero).  Every case against the oracle on the whole matrix.  (Shards are whole 32-row tiles:
row_partition refuses more ranks than tiles with a ValueError before any process touches the GPU.)

    python3 scripts/shard_fuzz.py
"""
import os
import socket
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def make(case):
    from oracle import klnmf_oracle as orc
    n, f, k, kind = case
    X = orc.synthetic_V(3 + n + f + k, n, f, k)
    if kind == 'zero_shard':
        X[: n // 2] = 0.0                       # rank 0 of 2 holds nothing but zeros
    if kind == 'sparse':
        X = X * (np.random.RandomState(1).random_sample(X.shape) < 0.05)
    return X, orc.synthetic_H0(3 + n + f + k, f, k)


def _worker(rank, world, port, case, iters, precision, out_dir):
    import torch
    import torch.distributed as dist
    from multimodal_amd.distributed import ShardedKLNMF, row_partition
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        X, H0 = make(case)
        n, f, k, _ = case
        r0, r1 = row_partition(n, world)[rank]
        msg = ''
        try:
            m = ShardedKLNMF(n, r1 - r0, f, k, max_iter=iters, precision=precision)
            m.set_v_max(X.max())
            m.upload_V(X[r0:r1])
            m.set_H(H0)
            m.init_W()
            errors, n_done, stopped = m.run(iters, fit=True, tol=0.0)
            np.savez(os.path.join(out_dir, 'r%d.npz' % rank), W=m.gather_W(), H=m.get_H(), errors=np.array(errors))
            m.close()
        except Exception as e:
            msg = '%s: %s' % (type(e).__name__, str(e)[:160])
        with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fh:
            fh.write(msg)
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    from oracle import klnmf_oracle as orc
    cases = [((67, 40, 6, ''), 2), ((67, 40, 6, ''), 3), ((70, 9, 2, ''), 3), ((64, 33, 1, ''), 2), ((400, 2755, 2, ''), 3),
             ((4096 + 31, 96, 40, ''), 4), ((66000, 64, 8, ''), 2), ((70000, 96, 40, 'sparse'), 2), ((300, 50, 5, 'zero_shard'), 2),
             ((131072 + 5, 64, 8, ''), 4)]
    bad = 0
    for prec in ('f16', 'f64'):
        for case, world in cases:
            n, f, k, kind = case
            iters = 4
            X, H0 = make(case)
            Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=0)
            with tempfile.TemporaryDirectory() as d:
                mp.spawn(_worker, args=(world, _free_port(), case, iters, prec, d), nprocs=world, join=True)
                msgs = [open(os.path.join(d, 'r%d.txt' % r)).read() for r in range(world)]
                if any(msgs):
                    same_everywhere = all(bool(m_) for m_ in msgs)
                    print('%-4s %6d x %4d k=%2d %-10s world %d  %s  every rank raised: %s | %s' % (prec, n, f, k, kind, world, 'REFUSED', same_everywhere, msgs[0][:110]), flush=True)
                    bad += 0 if same_everywhere else 1
                    continue
                res = [np.load(os.path.join(d, 'r%d.npz' % r)) for r in range(world)]
            rep_ok = all(np.array_equal(res[0]['H'], r['H']) and np.array_equal(res[0]['errors'], r['errors']) for r in res[1:])
            e = res[0]['errors']
            m_ = min(len(e), len(eo))
            lim_e, lim_w = (1e-9, 1e-7) if prec == 'f64' else (1e-3, 6e-3)
            floor_e = (1e-12 if prec == 'f64' else 1e-4) * float(X.sum())
            rel_e = float(np.max(np.abs(e[:m_] - np.array(eo[:m_])) / np.maximum(np.abs(eo[:m_]), floor_e)))
            same = len(e) == len(eo)
            dW = float(np.abs(res[0]['W'] - Wo).max() / max(np.abs(Wo).max(), 1e-300)) if same else float('nan')
            dH = float(np.abs(res[0]['H'] - Ho).max() / max(np.abs(Ho).max(), 1e-300)) if same else float('nan')
            ok = rep_ok and same and rel_e <= lim_e and dW <= lim_w and dH <= lim_w
            print('%-4s %6d x %4d k=%2d %-10s world %d  %s  replicas identical %s  len %d/%d losses %.1e W %.1e H %.1e' % (
                prec, n, f, k, kind, world, 'ok  ' if ok else 'FAIL', rep_ok, len(e), len(eo), rel_e, dW, dH), flush=True)
            bad += 0 if ok else 1
    print('%d case(s) outside their tolerance' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
