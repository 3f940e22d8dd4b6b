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
    import torch.distributed as dist
    import mpc4quantum_amd as m4q
    from mpc4quantum_amd import configs
    from mpc4quantum_amd.distributed import mpc_batch_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = configs.build(2, batch=5, horizon=6, n_steps=4)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = mpc_batch_sharded(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                            p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], solver=_oracle_solver,
                            final_state_only=final_only)
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
