"""Row-sharded KL-NMF across the GPUs of one node (one process per GPU).

The reference has no parallelism; this is the
one data-parallel strategy the algebra offers: samples (rows of V and W) are
partitioned, the dictionary H is replicated.  Per fit iteration the ranks
exchange exactly

  * the k x f numerator of the H rule, W_new^T . Q   (nmf.py:349)  -- one
    all-reduce(sum) over xGMI (RCCL; `torch.distributed` backend "nccl"), and
  * the scalar loss (2 doubles)                       (nmf.py:214)  -- a second,
    tiny all-reduce so every rank takes the same stop decision.

The W rule (nmf.py:342) is row-local and needs no communication; a transform
(`fit=False`) exchanges only the loss.  After the all-reduce every rank applies
the H rule and the row normalisation to identical inputs, so the replicas stay
bit-identical.

The per-rank arithmetic is a `_native.Context`; this module only sequences the
pieces of `include/klnmf.h` (klnmf_iter_*) around the collectives.  `backend` may
be any object with the same methods (the CPU tests drive the sequencing with an
oracle-backed double over gloo).
"""
import os

import numpy as np


def row_partition(n, world_size):
    """Contiguous row ranges [start, stop) per rank; sizes differ by at most 1
    block of 32 rows (the MFMA row-tile), the last rank takes the remainder.

    Every rank needs at least one 32-row tile: with fewer tiles than ranks this raises ValueError -- on every rank,
    before any collective (a rank with no rows would fail alone in klnmf_set_problem and leave the others waiting in
    their first all-reduce)."""
    tiles = (n + 31) // 32
    if world_size < 1 or tiles < world_size:
        raise ValueError("cannot shard %d rows (%d tiles of 32) over %d ranks: use at most %d ranks"
                         % (n, tiles, world_size, max(1, tiles)))
    base, extra = divmod(tiles, world_size)
    bounds = [0]
    for r in range(world_size):
        t = base + (1 if r < extra else 0)
        bounds.append(min(n, bounds[-1] + 32 * t))
    bounds[-1] = n
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


class ShardedKLNMF(object):
    """The loop of nmf.py:212-222 over row shards.

    Parameters
    ----------
    n_total, f, k : global problem shape (this rank holds `n_local` rows).
    precision     : 'f16' (= 'bf16') | 'f32' | 'f64'
    group         : torch.distributed process group or None (single process).
    backend       : context object; default = a HIP `_native.Context` on the
                    current torch device / stream.
    """

    def __init__(self, n_total, n_local, f, k, max_iter, precision='f16',
                 group=None, backend=None, device=None, collective='torch'):
        import torch
        self.torch = torch
        self.n_total, self.n_local, self.f, self.k = n_total, n_local, f, k
        self.max_iter = int(max_iter)
        self.group = group
        self.dist = None
        self.world_size = 1
        if group is not None or (torch.distributed.is_available()
                                 and torch.distributed.is_initialized()):
            self.dist = torch.distributed
            self.world_size = self.dist.get_world_size(group)
        if backend is None:
            from . import _native
            dev = torch.cuda.current_device() if device is None else device
            stream = torch.cuda.current_stream(dev).cuda_stream
            backend = _native.Context(precision=precision, device=dev, stream=stream)
            self.tensor_device = torch.device('cuda', dev)
        else:
            self.tensor_device = torch.device('cpu')
        self.ctx = backend
        self.ctx.set_problem(n_local, f, k, self.max_iter)
        _, _, count, is64 = self.ctx.exchange_buffers()
        # exchange buffers owned by torch so they can be handed to all_reduce
        self.numer_t = torch.zeros(count, dtype=torch.float64 if is64 else torch.float32,
                                   device=self.tensor_device)
        self.loss_t = torch.zeros(2, dtype=torch.float64, device=self.tensor_device)
        self.ctx.bind_exchange(self.loss_t.data_ptr(), self.numer_t.data_ptr())
        # only the k real component rows travel (the 16-bit modes lay the numerator out as [KP][f_pad]; the rows beyond
        # k are padding, one of them carries eps): the context says how many elements that is
        valid = count
        if hasattr(self.ctx, 'exchange_layout'):
            _, valid = self.ctx.exchange_layout()
        self.numer_xchg = self.numer_t[:valid]
        # column parts of the numerator (KLNMF_COMM_PARTS > 1, 16-bit modes): each part is one contiguous block of the buffer and
        # is exchanged while the next part's column pass computes (`iterate`)
        self.parts = None
        if hasattr(self.ctx, 'exchange_parts'):
            parts = self.ctx.exchange_parts()
            if len(parts) > 1:
                self.parts = [self.numer_t[off:off + cnt] for off, cnt, _, _ in parts]
        # collective path: 'torch' = torch.distributed all-reduces sequenced here around the C-ABI's pieces;
        # 'native' = klnmf_run_sharded: ONE grouped RCCL all-reduce per iteration issued inside the C-ABI
        self.collective = collective
        if collective == 'native' and self.dist is not None and self.world_size > 1:
            rank = self.dist.get_rank(group)
            # pre-flight: every rank opens librccl (klnmf_comm_unique_id does) and the outcome is agreed BEFORE anybody enters
            # ncclCommInitRank -- a rank that cannot load the library would otherwise leave the others waiting there for ever
            err, my_id = None, None
            try:
                my_id = self.ctx.comm_unique_id()
            except Exception as e:
                err = e
            ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64, device=self.tensor_device)
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=group)
            if float(ok.item()) == 0.0:
                raise err if err is not None else RuntimeError('librccl is not usable on another rank: no native communicator on any rank')
            box = [my_id if rank == 0 else None]
            self.dist.broadcast_object_list(box, src=0, group=group)
            self.ctx.comm_init(box[0], rank, self.world_size)
        elif collective == 'native' and self.world_size == 1 and os.environ.get('KLNMF_COMM_SINGLE') == '1':
            # rehearsal on one GPU: a one-rank communicator takes the collective branch of the library (tests, bench.py)
            self.ctx.comm_init(self.ctx.comm_unique_id(), 0, 1)
        self.iterations_enqueued = 0

    # ---- data ----
    def set_v_max(self, local_max):
        """Fix the 16-bit storage factor of V from the GLOBAL maximum (all ranks
        must use the same factor because they share H); call before uploading."""
        if self.collective == 'native':
            self.ctx.set_v_max(self.ctx.comm_max(float(local_max)))
            return
        t = self.torch.tensor([float(local_max)], dtype=self.torch.float64,
                              device=self.tensor_device)
        if self.dist is not None and self.world_size > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        self.ctx.set_v_max(float(t.item()))

    def upload_V(self, block, row0=0, col0=0, scale=1.0):
        self.ctx.upload_V(block, row0=row0, col0=col0, scale=scale)

    def upload_V_device(self, tensor, row0=0, col0=0, scale=1.0):
        """fp32 CUDA tensor [rows, cols] (row-major) already on this rank's GPU."""
        assert tensor.dtype == self.torch.float32 and tensor.is_contiguous()
        self.ctx.upload_V_device(tensor.data_ptr(), tensor.shape[0], tensor.shape[1],
                                 tensor.stride(0), row0, col0, scale)

    def set_H(self, H):
        self.ctx.set_H(H)

    def init_W(self):
        self.ctx.init_W()

    # ---- loop ----
    def _all_reduce(self, t):
        if self.dist is not None and self.world_size > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _all_reduce_async(self, t):
        """Start the collective now (it waits for what this rank's stream has enqueued so far) and return its
        handle; the caller's later kernels are NOT ordered behind it until `handle.wait()`."""
        if self.dist is not None and self.world_size > 1:
            return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def begin(self):
        """Open the loop on every rank -- or on none: a rank whose own shard is refused at the loop's entry (V beyond the
        announced maximum, factors beyond the fp16 operand range) must not leave the others waiting in their first
        all-reduce, so the outcome of the local entry is agreed (max) before anybody enqueues a collective."""
        if self.collective == 'native':
            # the agreement (refusal flags max, sums of V and cells) travels over the context's own RCCL communicator
            # inside klnmf_loop_begin: every rank raises together there
            self.ctx.loop_begin()
            self.iterations_enqueued = 0
            return
        multi = self.dist is not None and self.world_size > 1
        err = None
        if multi and hasattr(self.ctx, 'sum_V'):
            # every rank must take the same fp8 decision (16-bit modes): it is made from the sums over ALL shards, as
            # klnmf_run_sharded does on the native path.  ONE all-reduce carries [sum V, cells, refusal flag, entries > 0]: a rank whose
            # own sum cannot be read joins it with zeros and its flag set, so no rank is ever alone in a collective of
            # another shape
            # [4]: "this shard's shape does not allow fp8 ratio tiles" -- shards can straddle the row threshold, and ranks must
            # not mix tile formats (their numerators differ by sqrt(2)): any rank without them keeps all on 16-bit tiles
            vals = [0.0, 0.0, 0.0, 0.0, 0.0]
            try:
                cells = float(self.n_local) * float(self.f)
                # (entries > 0: a context that cannot count them reports a dense shard)
                nnz = self.ctx.nnz_V() if hasattr(self.ctx, 'nnz_V') else cells
                no_fp8 = 0.0 if (not hasattr(self.ctx, 'fp8_shape_ok') or self.ctx.fp8_shape_ok()) else 1.0
                vals = [self.ctx.sum_V(), cells, 0.0, nnz, no_fp8]
            except Exception as e:
                err, vals = e, [0.0, 0.0, 1.0, 0.0, 0.0]
            t = self.torch.tensor(vals, dtype=self.torch.float64, device=self.tensor_device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            if err is None and float(t[2].item()) == 0.0:
                try:
                    self.ctx.loop_begin(float(t[0].item()), float(t[1].item()), float(t[3].item()),
                                        fp8_shape_all=(float(t[4].item()) == 0.0))
                except Exception as e:         # (reported below, on every rank)
                    err = e
            elif err is None:
                err = RuntimeError("another rank could not read its shard's sums: the sharded loop is not started on any rank")
        else:
            try:
                self.ctx.loop_begin()
            except Exception as e:
                err = e
        if multi:
            flag = self.torch.tensor([1.0 if err is not None else 0.0], dtype=self.torch.float64, device=self.tensor_device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, group=self.group)
            if float(flag.item()) != 0.0:
                raise err if err is not None else RuntimeError(
                    "another rank's shard was refused at the loop's entry: the sharded loop is not started on any rank")
        elif err is not None:
            raise err
        self.iterations_enqueued = 0

    def rccl_ranks(self):
        """What RCCL reports for the native communicator (ncclCommCount); None on the torch path."""
        if self.collective == 'native' and hasattr(self.ctx, 'query'):
            from . import _native
            return self.ctx.query(_native.Q_COMM_RANKS)
        return None

    def iterate(self, fit=True, tol=0.0):
        """Enqueue one iteration (no host synchronisation)."""
        tol_abs = tol * self.n_total * self.f          # nmf.py:207 on the GLOBAL shape
        if self.collective == 'native':
            self.ctx.run_more(1, fit, tol_abs)
            self.iterations_enqueued += 1
            return
        self.ctx.iter_rowpass(fit)
        if fit:
            # The column pass does not depend on the stop decision (it needs W_new and the old ratio only), so
            # it runs before the numerator exchange: the GPUs do not idle between the two passes waiting for
            # a 16-byte collective.  If the stop rule fires, this
            # iteration's numerator is simply not applied (iter_update_H is a no-op once stopped).
            # The 16-byte loss exchange starts as soon as the row pass has left the local loss and runs on the
            # collective's own stream WHILE the column pass computes: only the numerator exchange is exposed.
            # ... the LOSS alone: loss_t[1] (monitor trips + unfixable fp8 ratio entries) is written by the column pass's last
            # launch (k_post, csrc/post.hip.h) and must not be in flight while that runs; it is exchanged behind the column
            # pass, on the iterations whose `iter_advance` polls it (the same iterations on every rank)
            pending = self._all_reduce_async(self.loss_t[:1])
            if self.parts is not None:
                # the numerator in column parts: the all-reduce of part p runs (on the collective's own stream) while the
                # column pass of part p + 1 computes; only the last part's exchange is exposed
                handles = []
                for p, buf in enumerate(self.parts):
                    self.ctx.iter_colpass_part(p)
                    handles.append(self._all_reduce_async(buf))
                for h in handles:
                    if h is not None:
                        h.wait()
            else:
                self.ctx.iter_colpass()
                self._all_reduce(self.numer_xchg)
            if pending is not None:
                pending.wait()
            if hasattr(self.ctx, 'fp8_poll_due') and self.ctx.fp8_poll_due():
                self._all_reduce(self.loss_t[1:])
            self.ctx.iter_decide(tol_abs)
            self.ctx.iter_update_H()
        else:
            self._all_reduce(self.loss_t)
            self.ctx.iter_decide(tol_abs)
        self.ctx.iter_advance()
        self.iterations_enqueued += 1

    def iterate_many(self, count, fit=True, tol=0.0):
        """`count` iterations of the open loop.  One process: the library enqueues them itself (klnmf_run_more -- the loss
        reduction and the stop rule then ride in the column pass's slab-sum launch, as in klnmf_run); several: `iterate`."""
        if (self.collective == 'native' or self.dist is None or self.world_size == 1) and hasattr(self.ctx, 'run_more'):
            # (native path, several ranks: klnmf_run_more carries the grouped RCCL all-reduce of every iteration itself --
            # klnmf_run_sharded in parts)
            self.ctx.run_more(count, fit, tol * self.n_total * self.f)
            self.iterations_enqueued += int(count)
            return
        for _ in range(int(count)):
            self.iterate(fit=fit, tol=tol)

    def end(self):
        """Synchronise; returns (errors, n_done, stopped) -- identical on every rank."""
        return self.ctx.loop_end(max(1, self.iterations_enqueued))

    def run(self, max_iter=None, fit=True, tol=0.0):
        max_iter = self.max_iter if max_iter is None else int(max_iter)
        if self.collective == 'native':          # the whole loop, collectives included, in one C-ABI call
            return self.ctx.run_sharded(self.n_total, max_iter, fit, tol)
        self.begin()
        for _ in range(max_iter):
            self.iterate(fit=fit, tol=tol)
        return self.end()

    # ---- results ----
    def get_W_local(self, dtype=np.float64):
        return self.ctx.get_W(dtype=dtype)

    def get_H(self, dtype=np.float64):
        return self.ctx.get_H(dtype=dtype)

    def gather_W(self, dtype=np.float64):
        """Full W on every rank (all-gather of the row shards; host side)."""
        W = self.get_W_local(dtype=dtype)
        if self.dist is None or self.world_size == 1:
            return W
        # one all-gather of equally sized (padded) blocks instead of pickled objects
        torch = self.torch
        sizes = [b - a for a, b in row_partition(self.n_total, self.world_size)]
        pad = max(sizes)
        mine = torch.zeros((pad, W.shape[1]), dtype=torch.float64 if np.dtype(dtype) == np.float64 else torch.float32,
                           device=self.tensor_device)
        mine[:W.shape[0]] = torch.from_numpy(np.ascontiguousarray(W)).to(self.tensor_device)
        parts = [torch.empty_like(mine) for _ in range(self.world_size)]
        self.dist.all_gather(parts, mine, group=self.group)
        return np.vstack([p[:s].cpu().numpy() for p, s in zip(parts, sizes)])

    def close(self):
        self.ctx.close()
