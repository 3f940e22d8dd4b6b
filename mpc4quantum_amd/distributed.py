"""Ensemble sharding over the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Ensemble members are independent closed loops
(mpc4quantum/mpc.py:128-304 has no cross-instance data flow), so the data path has NO collective: rank r
runs the contiguous block [r*B/G, (r+1)*B/G) and ONE gather of the results closes the job.

The results of a rank live in ONE contiguous byte buffer laid out by `ResultLayout`
    [ xs | us | exit_codes | steps_done | qp_solves ]          (each region sized for the largest shard)
On the GPU the session's output fields are bound INTO that buffer (m4q_session_bind_output): the kernel writes the
bytes RCCL sends, nothing is packed or staged through the host.  With "gloo" (CPU tests, rehearsals) the same layout
is filled from host arrays, so the offsets, padding and unpacking are exercised without a GPU.
torch is used for the process group and as the owner of the device buffer only."""
import numpy as np

from . import _lib


def shard_bounds(B, rank, world):
    """Contiguous block split; the first B % world ranks take one extra member."""
    base, extra = divmod(int(B), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _take(arr, lo, hi, B):
    """Slice the ensemble axis of an input that has one (leading axis of length B); pass shared inputs through."""
    arr = np.asarray(arr)
    return arr[lo:hi] if (arr.ndim > 0 and arr.shape[0] == B and B > 1) else arr


class ResultLayout:
    """Byte layout of one rank's results for `rows` members (time-major, as the C ABI holds them):
    xs [rows][xs_cols][n] complex128 | us [rows][ns][m] float64 | exit_codes [rows] i32 | steps_done [rows] i32 |
    qp_solves [rows][ns] i32.  xs_cols = ns + 1 (whole state history) or 1 (final state only)."""

    FIELDS = ("xs", "us", "exit_codes", "steps_done", "qp_solves")

    def __init__(self, rows, n, m, ns, final_state_only=False):
        self.rows, self.n, self.m, self.ns = int(rows), int(n), int(m), int(ns)
        self.xs_cols = 1 if final_state_only else self.ns + 1
        self.shape = {"xs": (self.rows, self.xs_cols, self.n), "us": (self.rows, self.ns, self.m), "exit_codes": (self.rows,),
                      "steps_done": (self.rows,), "qp_solves": (self.rows, self.ns)}
        self.dtype = {"xs": np.complex128, "us": np.float64, "exit_codes": np.int32, "steps_done": np.int32,
                      "qp_solves": np.int32}
        self.offset, pos = {}, 0
        for f in self.FIELDS:
            self.offset[f] = pos
            pos += int(np.prod(self.shape[f])) * np.dtype(self.dtype[f]).itemsize
            pos = (pos + 15) // 16 * 16                       # every region starts 16-byte aligned
        self.nbytes = pos

    def field_bytes(self, f):
        return int(np.prod(self.shape[f])) * np.dtype(self.dtype[f]).itemsize

    def view(self, buf, f):
        """NumPy view of field f inside a host byte buffer (uint8 ndarray of self.nbytes)."""
        o = self.offset[f]
        return buf[o:o + self.field_bytes(f)].view(self.dtype[f]).reshape(self.shape[f])

    def pack(self, res, buf):
        """Host arrays (k <= rows members, time-major xs [k][ns+1][n] / us [k][ns][m]) -> buf."""
        k = res["us"].shape[0]
        xs = res["xs"][:, -1:, :] if self.xs_cols == 1 else res["xs"]
        self.view(buf, "xs")[:k] = xs
        for f in self.FIELDS[1:]:
            self.view(buf, f)[:k] = res[f]

    def unpack(self, buf, k):
        """First k members of a host byte buffer as a dict of arrays (copies)."""
        return {f: self.view(buf, f)[:k].copy() for f in self.FIELDS}


class ShardedResults:
    """A session whose outputs live in one torch-owned device buffer, and the one collective that moves it.
    final_state_only: the buffer carries xs[:, -1] instead of the whole state history (the history stays in the session)."""

    _FIELD_ID = {"xs": _lib.F_XS, "us": _lib.F_US, "exit_codes": _lib.F_CODES, "steps_done": _lib.F_STEPS_DONE,
                 "qp_solves": _lib.F_QP_SOLVES}

    def __init__(self, sess, rows, group=None, dst=0, final_state_only=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.sess, self.group, self.dst = sess, group, dst
        p = sess.problem
        self.layout = ResultLayout(rows, p.dim_x, p.dim_u, p.n_steps, final_state_only)
        self.local = sess.B
        if sess.B > rows:
            raise ValueError("the buffer holds %d members, the session has %d" % (rows, sess.B))
        self.on_device = dist.get_backend(group) == "nccl"
        if self.on_device and not hasattr(sess, "bind_output"):
            raise NotImplementedError("host-solved blocks travel over gloo; nccl gathers a session's device buffer")
        dev = "cuda" if self.on_device else "cpu"
        self.buf = torch.zeros(self.layout.nbytes, dtype=torch.uint8, device=dev)
        self.xs_hist = None
        self.final_state_only = final_state_only
        self.pending = None
        if self.on_device and final_state_only:
            # the kernel writes the history into its own torch buffer; its last column is copied on the device
            self.xs_hist = torch.empty(sess.B * (p.n_steps + 1) * p.dim_x * 2, dtype=torch.float64, device=dev)
        self.bind()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.outs = [torch.empty_like(self.buf) for _ in range(self.world)] if self.rank == dst else None

    def bind(self):
        """Point the session's output fields at THIS buffer (two ShardedResults can alternate on one session, so that the
        gather of one run overlaps the next run's kernel)."""
        if not self.on_device:
            return
        sess = self.sess
        base = self.buf.data_ptr()
        for f in ResultLayout.FIELDS:
            if f == "xs" and self.final_state_only:
                sess.bind_output(_lib.F_XS, self.xs_hist.data_ptr(), self.xs_hist.numel() * 8)
                continue
            # the session's field covers sess.B members: a prefix of the region sized for `rows`
            sess.bind_output(self._FIELD_ID[f], base + self.layout.offset[f], sess.field_bytes(self._FIELD_ID[f]))

    def wait(self):
        """Block until a gather started with wait=False has completed (no-op otherwise)."""
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
            if self.on_device:
                self.torch.cuda.current_stream().synchronize()

    def gather(self, wait=True):
        """The one collective of the job.  Call after sess.run(); returns the per-rank buffers on dst (device tensors with
        nccl), None elsewhere.  The kernel runs on the session's own stream: it is drained first.  wait=False: return as
        soon as the collective is enqueued (call wait() before this buffer is bound and written again)."""
        torch = self.torch
        self.wait()
        self.sess.sync()
        if self.on_device:
            if self.xs_hist is not None:
                p = self.sess.problem
                o = self.layout.offset["xs"]
                dstv = self.buf[o:o + self.sess.B * p.dim_x * 16].view(torch.float64).view(self.sess.B, p.dim_x * 2)
                dstv.copy_(self.xs_hist.view(self.sess.B, p.n_steps + 1, p.dim_x * 2)[:, -1, :])
        else:
            res = self.sess.results()
            if res is not None:                                  # (a rank without members sends its zeroed buffer)
                self.layout.pack(res, self.buf.numpy())
        if not wait:
            self.pending = self.dist.gather(self.buf, self.outs, dst=self.dst, group=self.group, async_op=True)
            return self.outs
        self.dist.gather(self.buf, self.outs, dst=self.dst, group=self.group)
        if self.on_device:
            # RCCL runs on its own stream and the session on another that torch knows nothing about: without this the next
            # sess.run() could overwrite the buffer while it is still being sent
            torch.cuda.current_stream().synchronize()
        return self.outs

    def unpack(self, counts):
        """dst only: host dict of the whole ensemble from the gathered buffers; counts[r] = members of rank r."""
        parts = [self.layout.unpack(o.cpu().numpy(), k) for o, k in zip(self.outs, counts) if k > 0]
        return {f: np.concatenate([q[f] for q in parts], axis=0) for f in ResultLayout.FIELDS}


class _HostBlock:
    """Stand-in for a session when the block was solved by an injected host callable (CPU tests): same attributes
    ShardedResults reads."""

    class _P:
        pass

    def __init__(self, res, n, m, ns):
        self.problem = self._P()
        self.problem.dim_x, self.problem.dim_u, self.problem.n_steps = n, m, ns
        self.B = 0 if res is None else res["us"].shape[0]
        self._res = res

    def sync(self):
        pass

    def results(self):
        return self._res


def mpc_batch_sharded(x0, models, dim_u, order, X_targ, U_targ, clock, plant_op0, plant_ops, Q, R, Qf, sat, du=None,
                      group=None, dst=0, solver=None, final_state_only=False, **kw):
    """Same contract as mpc.mpc_batch, evaluated by every rank of `group` on its block; rank `dst` returns the
    full-ensemble dict (xs [B, n, cols], us [B, m, n_steps], exit_codes, steps_done, qp_solves), the others None.
    With backend "nccl" the block runs on the rank's GPU with its outputs bound into the gather buffer; `solver`
    (tests) is a host callable with mpc_batch's signature whose results take the same packed path over "gloo"."""
    import torch
    import torch.distributed as dist
    from .mpc import open_session
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    x0 = np.asarray(x0)
    B, n = x0.shape
    ns = clock.n_steps
    lo, hi = shard_bounds(B, rank, world)
    counts = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    rows = max(counts)
    models = np.asarray(models)
    if models.ndim == 2:
        models = models[None]
    op0 = np.asarray(plant_op0)
    ops = np.asarray(plant_ops)
    args = (x0[lo:hi], _take(models, lo, hi, B), dim_u, order,
            X_targ if np.ndim(X_targ) == 2 else _take(X_targ, lo, hi, B),
            U_targ if np.ndim(U_targ) == 2 else _take(U_targ, lo, hi, B), clock,
            _take(op0, lo, hi, B) if op0.ndim == 3 else op0, _take(ops, lo, hi, B) if ops.ndim == 4 else ops,
            Q, R, Qf, sat, du)
    sess = None
    try:
        if solver is not None or dist.get_backend(group) != "nccl":
            if solver is None:
                from .mpc import mpc_batch as solver
            res = None
            if hi > lo:
                res = dict(solver(*args, **kw))
                res["xs"], res["us"] = np.swapaxes(res["xs"], 1, 2), np.swapaxes(res["us"], 1, 2)     # time-major
            block = _HostBlock(res, n, dim_u, ns)
        else:
            if hi <= lo:
                raise ValueError("more ranks than ensemble members")
            kw.setdefault("device", torch.cuda.current_device())
            sess = block = open_session(*args, **kw)
        sr = ShardedResults(block, rows, group, dst, final_state_only)
        if sess is not None:
            sess.run(0, ns)
        sr.gather()
        if rank != dst:
            return None
        out = sr.unpack(counts)
    finally:
        if sess is not None:
            sess.close()
    out["xs"] = np.swapaxes(out["xs"], 1, 2)
    out["us"] = np.swapaxes(out["us"], 1, 2)
    return out
