"""World-size-2 test of the multi-GPU driver on CPU (gloo): partition, per-rank solve, one gather,
reassembly.  The per-rank solver is injected (the oracle, as the checker) because this box has no GPU; on
the GPU box the same driver calls the HIP path and gathers over RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_solver(x0, models, dim_u, order, X_targ, U_targ, clock, op0, ops, Q, R, Qf, sat, du, **kw):
    from oracle import m4q_oracle as orc
    xs, us, codes, solves = orc.mpc_batch(x0, models, dim_u, order, X_targ, U_targ, clock.dt, clock.horizon, clock.n_steps,
                                          op0, list(np.asarray(ops).reshape((-1,) + np.asarray(ops).shape[-3:])[0]), Q, R, Qf,
                                          sat, du)
    return {"xs": xs, "us": us, "exit_codes": codes, "steps_done": np.full(len(codes), clock.n_steps, dtype=np.int32),
            "qp_solves": solves}


def _worker(rank, world, port, out_path, final_only=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mpc4quantum_amd as m4q
    from mpc4quantum_amd import configs
    from mpc4quantum_amd.distributed import mpc_batch_sharded
    from gloo_transport import GlooTransport
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = configs.build(2, batch=5, horizon=6, n_steps=4)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = mpc_batch_sharded(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                            p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], solver=_oracle_solver,
                            transport=GlooTransport(), final_state_only=final_only)
    if rank == 0:
        np.savez(out_path, **res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("final_only", [False, True])
def test_sharded_driver_world2_matches_single_process(tmp_path, final_only):
    """5 members over 2 ranks (blocks of 3 and 2: the shorter block is padded inside the gather buffer).  Every rank packs
    its block into the ResultLayout byte buffer the GPU path binds its session outputs into, ONE gather moves it, rank 0
    unpacks.  final_only: the buffer carries the final state instead of the state history (what bench.py gathers)."""
    import torch.multiprocessing as mp
    import mpc4quantum_amd as m4q
    from mpc4quantum_amd import configs
    out = str(tmp_path / "gathered.npz")
    port = 29500 + (os.getpid() % 2000) + (7 if final_only else 0)
    mp.spawn(_worker, args=(2, port, out, final_only), nprocs=2, join=True)
    got = np.load(out)
    p = configs.build(2, batch=5, horizon=6, n_steps=4)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    ref = _oracle_solver(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                         p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"])
    if final_only:
        ref["xs"] = ref["xs"][:, :, -1:]
    for k in ref:
        assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k


def _failing_worker(rank, world, port, out_path):
    """rank 1's solve raises: it must still join the gather (status word), rank 0 must raise instead of returning a half
    ensemble, and neither may hang."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import mpc4quantum_amd as m4q
    from mpc4quantum_amd import configs, _lib
    from mpc4quantum_amd.distributed import mpc_batch_sharded
    from gloo_transport import GlooTransport
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def solver(*a, **kw):
        if rank == 1:
            raise RuntimeError("boom on rank 1")
        return _oracle_solver(*a, **kw)

    p = configs.build(2, batch=5, horizon=6, n_steps=4)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    what = "returned"
    try:
        mpc_batch_sharded(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                          p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], solver=solver, transport=GlooTransport())
    except _lib.M4qError as e:
        what = "M4qError: %s" % e
    except RuntimeError as e:
        what = "RuntimeError: %s" % e
    open("%s.%d" % (out_path, rank), "w").write(what)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_rank_local_failure_joins_the_gather(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "status")
    port = 29500 + (os.getpid() % 2000) + 13
    mp.spawn(_failing_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = open(out + ".0").read(), open(out + ".1").read()
    assert r0.startswith("M4qError") and "rank(s) [1]" in r0, r0
    assert r1.startswith("RuntimeError: boom"), r1


def test_more_ranks_than_members_raises_on_every_rank():
    """B < world on a device transport: decided on data every rank has, before any session or collective."""
    import mpc4quantum_amd as m4q
    from mpc4quantum_amd import configs
    from mpc4quantum_amd.distributed import mpc_batch_sharded

    class FakeDev:
        on_device = True
        world = 4

        def __init__(self, rank):
            self.rank = rank

    p = configs.build(2, batch=3, horizon=6, n_steps=4)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    for r in range(4):
        with pytest.raises(ValueError, match="more ranks"):
            mpc_batch_sharded(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                              p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], transport=FakeDev(r))


def test_unique_id_file_exchange(tmp_path, monkeypatch):
    """The launch key names a launch (port + the launcher's pid and start time); a reader waits for the whole file."""
    from mpc4quantum_amd import distributed as D
    monkeypatch.setenv("MASTER_PORT", "12345")
    k = D._launch_key()
    assert k.startswith("12345_%d_" % os.getppid()) and k == D._launch_key()
    path = str(tmp_path / "uid")
    open(path, "wb").write(b"x" * 10)                       # a torn / foreign file is not accepted
    with pytest.raises(TimeoutError):
        D.exchange_unique_id(1, 2, path=path, timeout=0.2)
    open(path, "wb").write(bytes(range(128)))
    data, _ = D.exchange_unique_id(1, 2, path=path, timeout=1.0)
    assert data == bytes(range(128))


def test_product_package_does_not_import_torch():
    """north_star: host code is Python over a ctypes C ABI, no PyTorch.  torch lives in the test harness only."""
    import re
    pkg = os.path.join(ROOT, "mpc4quantum_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(import|from)\s+torch", src, re.M), name
