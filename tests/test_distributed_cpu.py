"""Row-sharded driver (multimodal_amd/distributed.py) over gloo, world_size 2,
on CPU.  The per-rank arithmetic is an oracle-backed double with the same
interface as `_native.Context` (the HIP kernels need a GPU); what is tested is
the N>1 path itself: row partition, the sequencing of the klnmf_iter_* pieces
around the two all-reduces (loss, k x f numerator), the common stop decision,
and that the sharded result equals the single-process oracle fit."""
import ctypes
import os
import socket

import numpy as np
import pytest

from oracle import klnmf_oracle as orc
from multimodal_amd.distributed import ShardedKLNMF, row_partition


class OracleContext(object):
    """CPU double of _native.Context for one row shard (fp64)."""

    def __init__(self):
        self.loss = self.numer = None

    def set_problem(self, n, f, k, cap):
        self.n, self.f, self.k = n, f, k
        self.V = np.zeros((n, f))
        self.W = np.zeros((n, k))
        self.H = np.zeros((k, f))

    def exchange_buffers(self):
        return None, None, self.k * self.f, True

    def bind_exchange(self, loss_ptr, numer_ptr):
        self.loss = np.ctypeslib.as_array((ctypes.c_double * 2).from_address(loss_ptr))
        self.numer = np.ctypeslib.as_array(
            (ctypes.c_double * (self.k * self.f)).from_address(numer_ptr)).reshape(self.k, self.f)

    def set_v_max(self, vmax):
        self.vmax = vmax

    def upload_V(self, block, row0=0, col0=0, scale=1.0):
        b = np.asarray(block, dtype=np.float64)
        self.V[row0:row0 + b.shape[0], col0:col0 + b.shape[1]] = scale * b

    def set_H(self, H):
        self.H = np.array(H, dtype=np.float64)

    def init_W(self):
        self.W = self.V.dot(self.H.T)

    # ---- the loop in pieces (device semantics: no-ops once stopped) ----
    def loop_begin(self):
        self.prev, self.stop, self.errors = np.inf, False, []

    def iter_rowpass(self, fit):
        if self.stop:
            return
        self.loss[0] = orc.kl_error(self.V, self.W, self.H)
        self.loss[1] = 0.0
        self.Q = orc.ratio_q(self.V, self.W, self.H)
        self.W_new = orc.updated_w(self.V, self.W, self.H, Q=self.Q)

    def iter_decide(self, tol_abs):
        if self.stop:
            return
        err = float(self.loss[0])
        if self.prev - err < tol_abs:
            self.stop = True
            return
        self.prev = err
        self.errors.append(err)
        self.commit = True

    def iter_colpass(self):
        if self.stop:
            return
        self.numer[...] = self.W_new.T.dot(self.Q)

    def iter_update_H(self):
        if self.stop:
            return
        self.H = orc.normalize_sum(self.H * self.numer, axis=1)

    def iter_advance(self):
        if not self.stop:
            self.W = self.W_new

    def loop_end(self, cap):
        return list(self.errors), len(self.errors), self.stop

    def get_W(self, dtype=np.float64):
        return self.W.astype(dtype)

    def get_H(self, dtype=np.float64):
        return self.H.astype(dtype)

    def close(self):
        pass


def test_row_partition():
    assert row_partition(100, 1) == [(0, 100)]
    parts = row_partition(1000000, 8)
    assert parts[0][0] == 0 and parts[-1][1] == 1000000
    assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    sizes = [b - a for a, b in parts]
    assert max(sizes) - min(sizes) <= 32 and all(s % 32 == 0 for s in sizes[:-1])
    assert row_partition(37, 2) == [(0, 32), (32, 37)]
    with pytest.raises(ValueError):          # one 32-row tile cannot feed four ranks (a rank with no rows would hang the others)
        row_partition(5, 4)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class PartedOracleContext(OracleContext):
    """The column-part form of the exchange (klnmf_exchange_parts / klnmf_iter_colpass_part): the numerator is produced and
    exchanged in `nparts` column ranges, each one contiguous block [k][width] of the buffer."""

    def __init__(self, nparts):
        OracleContext.__init__(self)
        self.nparts = nparts

    def _ranges(self):
        edges = [self.f * p // self.nparts for p in range(self.nparts + 1)]
        return list(zip(edges[:-1], edges[1:]))

    def bind_exchange(self, loss_ptr, numer_ptr):
        self.loss = np.ctypeslib.as_array((ctypes.c_double * 2).from_address(loss_ptr))
        self.flat = np.ctypeslib.as_array((ctypes.c_double * (self.k * self.f)).from_address(numer_ptr))

    def exchange_parts(self):
        out, off = [], 0
        for a, b in self._ranges():
            out.append((off, self.k * (b - a), a, b - a))
            off += self.k * (b - a)
        return out

    def iter_colpass_part(self, p):
        if self.stop:
            return
        (off, cnt, a, w) = self.exchange_parts()[p]
        self.flat[off:off + cnt] = self.W_new.T.dot(self.Q[:, a:a + w]).ravel()

    def iter_colpass(self):
        raise AssertionError("a problem with column parts is driven part by part")

    def iter_update_H(self):
        if self.stop:
            return
        numer = np.hstack([self.flat[off:off + cnt].reshape(self.k, w) for off, cnt, a, w in self.exchange_parts()])
        self.H = orc.normalize_sum(self.H * numer, axis=1)


def _worker(rank, world, port, n, f, k, iters, tol, fit, out_dir, nparts=1):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        X = orc.synthetic_V(77, n, f, k)
        H0 = orc.synthetic_H0(77, f, k)
        r0, r1 = row_partition(n, world)[rank]
        m = ShardedKLNMF(n, r1 - r0, f, k, max_iter=iters, backend=OracleContext() if nparts == 1 else PartedOracleContext(nparts))
        assert (m.parts is None) == (nparts == 1)
        m.set_v_max(X[r0:r1].max())
        m.upload_V(X[r0:r1])
        m.set_H(H0)
        m.init_W()
        errors, n_done, stopped = m.run(iters, fit=fit, tol=tol)
        W = m.gather_W()
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), W=W, H=m.get_H(),
                 errors=np.array(errors), n_done=n_done, stopped=stopped)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('iters,tol,fit', [(8, 0.0, True), (200, 1e-4, True), (6, 0.0, False)])
def test_sharded_equals_single_process(tmp_path, iters, tol, fit):
    import torch.multiprocessing as mp
    n, f, k, world = 75, 40, 6, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, f, k, iters, tol, fit, str(tmp_path)), nprocs=world, join=True)
    X = orc.synthetic_V(77, n, f, k)
    H0 = orc.synthetic_H0(77, f, k)
    if fit:
        Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=tol, warn=False)
    else:
        Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=tol, fit=False,
                                       components=H0, warn=False)
    res = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    for r in res:
        assert int(r['n_done']) == len(eo)                       # same stop iteration on every rank
        np.testing.assert_allclose(r['errors'], eo, rtol=1e-11)
        np.testing.assert_allclose(r['W'], Wo, rtol=1e-9)
        np.testing.assert_allclose(r['H'], Ho, rtol=1e-9)
    np.testing.assert_array_equal(res[0]['H'], res[1]['H'])      # replicas stay bit-identical
    if tol > 0:
        assert bool(res[0]['stopped']) and len(eo) < iters


@pytest.mark.parametrize('world,nparts,iters,tol', [(2, 2, 8, 0.0), (4, 2, 6, 0.0), (2, 3, 200, 1e-4)])
def test_numerator_exchanged_in_column_parts_equals_single_process(tmp_path, world, nparts, iters, tol):
    """The column-range form of the exchange (round 4: the all-reduce of part p is started while part p + 1 computes):
    sequencing over gloo with 2 and 4 ranks, 2 and 3 parts, with and without a stop rule that fires -- the result is the
    single-process oracle fit, the replicas stay bit-identical."""
    import torch.multiprocessing as mp
    n, f, k = 140, 40, 6
    mp.spawn(_worker, args=(world, _free_port(), n, f, k, iters, tol, True, str(tmp_path), nparts), nprocs=world, join=True)
    X = orc.synthetic_V(77, n, f, k)
    H0 = orc.synthetic_H0(77, f, k)
    Wo, Ho, eo = orc.fit_transform(X, k=k, H0=H0, max_iter=iters, tol=tol, warn=False)
    res = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    for r in res:
        assert int(r['n_done']) == len(eo)
        np.testing.assert_allclose(r['errors'], eo, rtol=1e-11)
        np.testing.assert_allclose(r['W'], Wo, rtol=1e-9)
        np.testing.assert_allclose(r['H'], Ho, rtol=1e-9)
        np.testing.assert_array_equal(res[0]['H'], r['H'])


def _tiny_worker(rank, world, port, n, out_dir):
    """Every rank must fail the same way BEFORE any collective when there are fewer 32-row tiles than ranks."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        try:
            row_partition(n, world)
            msg = 'no error'
        except ValueError as e:
            msg = str(e)
        with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fh:
            fh.write(msg)
        dist.barrier()                       # and nobody is left waiting in a collective
    finally:
        dist.destroy_process_group()


def test_fewer_row_tiles_than_ranks_is_refused_on_every_rank(tmp_path):
    import torch.multiprocessing as mp
    assert [b - a for a, b in row_partition(100, 3)] == [64, 32, 4]      # 4 tiles of 32 rows over 3 ranks, the last one ragged
    with pytest.raises(ValueError):
        row_partition(100, 8)                # 4 tiles of 32 rows for 8 ranks (ranks with no rows)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_tiny_worker, args=(2, port, 20, str(tmp_path)), nprocs=2, join=True)
    msgs = [open(os.path.join(str(tmp_path), 'r%d.txt' % r)).read() for r in range(2)]
    assert msgs[0] == msgs[1] and 'cannot shard 20 rows' in msgs[0]


class RefusingContext(OracleContext):
    """A shard whose loop entry fails (as klnmf_loop_begin does for a V beyond the announced maximum)."""

    def loop_begin(self):
        raise RuntimeError("klnmf error -1: uploaded V exceeds the maximum given to klnmf_set_v_max")


def _refusal_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n, f, k = 75, 40, 6
        X = orc.synthetic_V(77, n, f, k)
        r0, r1 = row_partition(n, world)[rank]
        m = ShardedKLNMF(n, r1 - r0, f, k, max_iter=3, backend=RefusingContext() if rank == 1 else OracleContext())
        m.set_v_max(X[r0:r1].max()); m.upload_V(X[r0:r1]); m.set_H(orc.synthetic_H0(77, f, k)); m.init_W()
        try:
            m.run(3, fit=True, tol=0.0)
            msg = 'no error'
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fh:
            fh.write(msg)
        dist.barrier()                       # nobody is left waiting in a collective
    finally:
        dist.destroy_process_group()


class NoSumContext(OracleContext):
    """A shard whose sums cannot be read at the loop's entry (klnmf_query_f64 failing on one rank)."""
    def __init__(self, broken):
        OracleContext.__init__(self)
        self.broken = broken

    def sum_V(self):
        if self.broken:
            raise RuntimeError("klnmf error -4: hipMemcpyAsync: device lost")
        return float(self.V.sum())

    def loop_begin(self, sum_all=None, cells_all=None, nnz_all=None, fp8_shape_all=None):
        OracleContext.loop_begin(self)


def _nosum_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n, f, k = 75, 40, 6
        X = orc.synthetic_V(77, n, f, k)
        r0, r1 = row_partition(n, world)[rank]
        m = ShardedKLNMF(n, r1 - r0, f, k, max_iter=3, backend=NoSumContext(broken=(rank == 1)))
        m.set_v_max(X[r0:r1].max()); m.upload_V(X[r0:r1]); m.set_H(orc.synthetic_H0(77, f, k)); m.init_W()
        try:
            m.run(3, fit=True, tol=0.0)
            msg = 'no error'
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fh:
            fh.write(msg)
        dist.barrier()                       # nobody is left waiting in a collective of another shape
    finally:
        dist.destroy_process_group()


def test_a_failing_sum_on_one_rank_joins_the_same_collective(tmp_path):
    """an exception from ctx.sum_V() on one rank happened BEFORE the entry's first all-reduce, so that rank
    went straight to the 1-element flag all-reduce while its peers sat in the 2-element one (mismatched collectives).  The
    flag now rides in the same all-reduce (sums, cells, flag, entries > 0): every rank raises, nobody hangs."""
    import torch.multiprocessing as mp
    mp.spawn(_nosum_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(os.path.join(str(tmp_path), 'r%d.txt' % r)).read() for r in range(2)]
    assert 'device lost' in msgs[1]
    assert "another rank" in msgs[0]


def test_a_refusal_on_one_rank_stops_every_rank(tmp_path):
    """a rank-local refusal at the loop's entry (here rank 1) must fail the loop on EVERY rank instead of
    leaving the others in their first all-reduce for ever (torch-sequenced path; klnmf_run_sharded does the same natively)."""
    import torch.multiprocessing as mp
    mp.spawn(_refusal_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(os.path.join(str(tmp_path), 'r%d.txt' % r)).read() for r in range(2)]
    assert 'exceeds the maximum' in msgs[1]                  # the rank that was refused says why
    assert "another rank's shard was refused" in msgs[0]     # the other one stops too, and says so


class NoRcclContext(OracleContext):
    """A rank on which librccl cannot be opened (klnmf_comm_unique_id fails with KLNMF_ERR_RCCL)."""
    def __init__(self, broken):
        OracleContext.__init__(self)
        self.broken = broken

    def comm_unique_id(self):
        if self.broken:
            raise RuntimeError("klnmf error -6: librccl not found")
        return b'\0' * 128

    def comm_init(self, uid, rank, nranks):
        raise AssertionError("ncclCommInitRank must not be entered when a rank cannot load librccl")


def _preflight_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        try:
            ShardedKLNMF(64, 32, 16, 4, max_iter=2, backend=NoRcclContext(broken=(rank == 1)), collective='native')
            msg = 'constructed'
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fh:
            fh.write(msg)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_native_communicator_preflight_is_agreed(tmp_path):
    """A rank that cannot open librccl says so before anybody enters ncclCommInitRank: every rank raises (bench.py then falls
    back to the torch path on all of them together) instead of the healthy ranks waiting for the missing one for ever."""
    import torch.multiprocessing as mp
    mp.spawn(_preflight_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(os.path.join(str(tmp_path), 'r%d.txt' % r)).read() for r in range(2)]
    assert 'librccl not found' in msgs[1]
    assert 'not usable on another rank' in msgs[0]


class CountingContext(OracleContext):
    """A shard that can count its entries > 0 (klnmf_query_f64 KLNMF_QF_NNZ_V) and records what the loop's entry was given."""

    def sum_V(self):
        return float(self.V.sum())

    def nnz_V(self):
        return float((self.V > 0).sum())

    fp8_ok = True            # this shard's SHAPE allows fp8 ratio tiles (klnmf_query KLNMF_Q_RATIO_TILE_BYTES == 1)

    def fp8_shape_ok(self):
        return self.fp8_ok

    def loop_begin(self, sum_all=None, cells_all=None, nnz_all=None, fp8_shape_all=None):
        self.entry = (sum_all, cells_all, nnz_all, 1.0 if fp8_shape_all else 0.0)
        OracleContext.loop_begin(self)


def _nnz_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n, f, k = 96, 40, 6
        X = orc.synthetic_V(77, n, f, k)
        X[:40] *= (np.random.RandomState(3).random_sample((40, f)) < 0.1)      # the first shard is sparse, the second dense
        r0, r1 = row_partition(n, world)[rank]
        ctx = CountingContext()
        ctx.fp8_ok = rank != world - 1 or world == 1 or out_dir.endswith('all_ok')      # the last shard is "too short" unless told otherwise
        m = ShardedKLNMF(n, r1 - r0, f, k, max_iter=2, backend=ctx)
        m.set_v_max(X.max()); m.upload_V(X[r0:r1]); m.set_H(orc.synthetic_H0(77, f, k)); m.init_W()
        m.run(2, fit=True, tol=0.0)
        np.save(os.path.join(out_dir, 'r%d.npy' % rank), np.array(ctx.entry, dtype=np.float64))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_every_rank_enters_the_loop_with_the_global_count_of_entries(tmp_path):
    """Round 4: the fp8 decision of a loop needs enough ENTRIES > 0 per column (sparse data stored densely), so the loop's entry
    all-reduces the shards' counts with their sums (one 4-element all-reduce: sum, cells, refusal flag, entries) and hands
    every context the same three global numbers -- klnmf_loop_begin_sharded_nnz on the HIP path."""
    import torch.multiprocessing as mp
    mp.spawn(_nnz_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    n, f, k = 96, 40, 6
    X = orc.synthetic_V(77, n, f, k)
    X[:40] *= (np.random.RandomState(3).random_sample((40, f)) < 0.1)
    got = [np.load(os.path.join(str(tmp_path), 'r%d.npy' % r)) for r in range(2)]
    np.testing.assert_array_equal(got[0], got[1])
    # [3]: the conjunction of the ranks' "my shard's shape allows fp8 ratio tiles" (round 5, klnmf_loop_begin_agreed): shards can
    # straddle the row threshold, and one rank on 16-bit tiles while the others sum fp8-tile numerators (sqrt(2) larger) would
    # corrupt the H rule silently -- here the last rank lacks the shape, so NO rank may take the tiles
    np.testing.assert_allclose(got[0], [X.sum(), n * f, (X > 0).sum(), 0.0], rtol=1e-12)
    ok_dir = os.path.join(str(tmp_path), 'all_ok')
    os.makedirs(ok_dir)
    mp.spawn(_nnz_worker, args=(2, _free_port(), ok_dir), nprocs=2, join=True)
    got = [np.load(os.path.join(ok_dir, 'r%d.npy' % r)) for r in range(2)]
    np.testing.assert_array_equal(got[0], got[1])
    assert got[0][3] == 1.0


class TrippingContext(OracleContext):
    """A shard whose column pass publishes a rank-local count in loss[1] AFTER the loss exchange has started -- what k_post's
    last block does with the fp8 monitor's trips (csrc/post.hip.h) -- and whose `iter_advance` polls it on the iterations
    `fp8_poll_due` names (KLNMF_Q_FP8_POLL_DUE)."""

    def __init__(self, my_trips, due):
        OracleContext.__init__(self)
        self.my_trips, self.due, self.it, self.seen = my_trips, due, 0, []

    def fp8_poll_due(self):
        return self.it in self.due

    def iter_rowpass(self, fit):
        OracleContext.iter_rowpass(self, fit)
        self.loss[1] = -777.0                # (whatever the loss kernel left there: it must never reach a poll)

    def iter_colpass(self):
        OracleContext.iter_colpass(self)
        self.loss[1] = float(self.my_trips.get(self.it, 0))

    def iter_advance(self):
        if self.fp8_poll_due():
            self.seen.append((self.it, float(self.loss[1])))
        OracleContext.iter_advance(self)
        self.it += 1


def _trip_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n, f, k = 96, 40, 6
        X = orc.synthetic_V(77, n, f, k)
        r0, r1 = row_partition(n, world)[rank]
        # only rank 1's monitor trips, in iteration 2 (3 component rows) -- iterations 1, 2 and 4 poll
        ctx = TrippingContext({2: 3} if rank == 1 else {}, due=(1, 2, 4))
        m = ShardedKLNMF(n, r1 - r0, f, k, max_iter=6, backend=ctx)
        m.set_v_max(X.max()); m.upload_V(X[r0:r1]); m.set_H(orc.synthetic_H0(77, f, k)); m.init_W()
        errors, n_done, stopped = m.run(6, fit=True, tol=0.0)
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), seen=np.array(ctx.seen), errors=np.array(errors), H=m.get_H())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_the_trip_count_is_exchanged_behind_the_column_pass(tmp_path):
    """the torch path started the all-reduce of BOTH loss doubles right behind the row pass, and the
    column pass's last launch then wrote its rank-local trip count into loss[1] while (or after) that collective ran: each
    rank polled its own count and the ranks could leave the fp8 regime in different iterations.  Now loss[0] alone is
    exchanged early; loss[1] behind the column pass on the polling iterations: every rank reads the SUM of the counts the
    column passes of that iteration published, and the loss itself is untouched by it."""
    import torch.multiprocessing as mp
    mp.spawn(_trip_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(2)]
    np.testing.assert_array_equal(got[0]['seen'], got[1]['seen'])
    np.testing.assert_array_equal(got[0]['seen'], [[1, 0.0], [2, 3.0], [4, 0.0]])
    np.testing.assert_array_equal(got[0]['errors'], got[1]['errors'])
    np.testing.assert_array_equal(got[0]['H'], got[1]['H'])
    n, f, k = 96, 40, 6
    X = orc.synthetic_V(77, n, f, k)
    _, _, eo = orc.fit_transform(X, k=k, H0=orc.synthetic_H0(77, f, k), max_iter=6, tol=0)
    np.testing.assert_allclose(got[0]['errors'], eo, rtol=1e-10)
