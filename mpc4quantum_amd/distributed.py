"""Ensemble sharding over the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Ensemble members are independent closed loops
(mpc4quantum/mpc.py:128-304 has no cross-instance data flow), so the data path has NO collective: rank r
runs the contiguous block [r*B/G, (r+1)*B/G) and ONE gather of the results closes the job."""
import numpy as np


def shard_bounds(B, rank, world):
    """Contiguous block split; the first B % world ranks take one extra member."""
    base, extra = divmod(int(B), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _take(arr, lo, hi, B):
    """Slice the ensemble axis of an input that has one (leading axis of length B); pass shared inputs through."""
    arr = np.asarray(arr)
    return arr[lo:hi] if (arr.ndim > 0 and arr.shape[0] == B and B > 1) else arr


def mpc_batch_sharded(x0, models, dim_u, order, X_targ, U_targ, clock, plant_op0, plant_ops, Q, R, Qf, sat, du=None,
                      group=None, dst=0, solver=None, **kw):
    """Same contract as mpc.mpc_batch, evaluated by every rank of `group` on its block; rank `dst` returns the
    full-ensemble dict, the others return None.  `solver` defaults to the HIP path (mpc.mpc_batch) on the rank's
    own device; tests inject a CPU callable to exercise the partition/gather logic without a GPU."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    x0 = np.asarray(x0)
    B = x0.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    models = np.asarray(models)
    if models.ndim == 2:
        models = models[None]
    op0 = np.asarray(plant_op0)
    ops = np.asarray(plant_ops)
    local = None
    if hi > lo:
        if solver is None:
            from .mpc import mpc_batch as solver
            if torch.cuda.is_available():
                kw.setdefault("device", torch.cuda.current_device())
        local = solver(x0[lo:hi], _take(models, lo, hi, B), dim_u, order,
                       X_targ if np.ndim(X_targ) == 2 else _take(X_targ, lo, hi, B),
                       U_targ if np.ndim(U_targ) == 2 else _take(U_targ, lo, hi, B), clock,
                       _take(op0, lo, hi, B) if op0.ndim == 3 else op0, _take(ops, lo, hi, B) if ops.ndim == 4 else ops,
                       Q, R, Qf, sat, du, **kw)
    # the one collective of the job
    use_cuda = dist.get_backend(group) == "nccl"
    if use_cuda:
        # RCCL moves device buffers: pack the block's results into one tensor per rank
        keys = ["xs", "us", "exit_codes", "steps_done", "qp_solves"]
        n, ns, m = x0.shape[1], clock.n_steps, dim_u
        width = 2 * n * (ns + 1) + m * ns + 2 + ns
        rows = max(shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world))
        buf = torch.zeros(rows, width, dtype=torch.float64)
        if local is not None:
            k = hi - lo
            packed = np.concatenate([local["xs"].reshape(k, -1).view(np.float64), local["us"].reshape(k, -1),
                                     local["exit_codes"].reshape(k, 1).astype(np.float64),
                                     local["steps_done"].reshape(k, 1).astype(np.float64),
                                     local["qp_solves"].astype(np.float64)], axis=1)
            buf[:k] = torch.from_numpy(packed)
        buf = buf.cuda()
        outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, outs, dst=dst, group=group)
        if rank != dst:
            return None
        parts = []
        for r in range(world):
            a, b = shard_bounds(B, r, world)
            parts.append(outs[r][:b - a].cpu().numpy())
        allp = np.concatenate(parts, axis=0)
        o = 0
        xs = np.ascontiguousarray(allp[:, o:o + 2 * n * (ns + 1)]).view(np.complex128).reshape(B, n, ns + 1)
        o += 2 * n * (ns + 1)
        us = allp[:, o:o + m * ns].reshape(B, m, ns)
        o += m * ns
        return {"xs": xs, "us": us, "exit_codes": allp[:, o].astype(np.int32), "steps_done": allp[:, o + 1].astype(np.int32),
                "qp_solves": allp[:, o + 2:].astype(np.int32)}
    gathered = [None] * world if rank == dst else None
    dist.gather_object(local, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [g for g in gathered if g is not None]
    return {k: np.concatenate([g[k] for g in parts], axis=0) for k in parts[0] if isinstance(parts[0][k], np.ndarray)}
