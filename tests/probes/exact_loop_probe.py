"""Exploration: closed loop with the exact box QP on the device against the oracle's (BVLS) loop, and timing against the
clipped loop.  Test infrastructure only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc4quantum_amd as m4q            # noqa: E402
from mpc4quantum_amd import configs      # noqa: E402
from oracle import m4q_oracle as orc     # noqa: E402


def gpu(p, idx, **kw):
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    return m4q.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                         p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], **kw)


def oracle(p, idx, **kw):
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    return orc.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"], p["horizon"],
                         p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"], p["R"], p["Qf"], p["sat"], p["du"], **kw)


def parity(cfg, order, batch, horizon):
    p = configs.build(cfg, batch=batch, order=order, horizon=horizon)
    idx = np.arange(batch)
    t0 = time.time()
    xs, us, codes, solves = oracle(p, idx, qp_mode="exact")
    t1 = time.time()
    for path in ("real", "complex"):
        res = gpu(p, idx, exact_qp=True, force_complex=(path == "complex"))
        resc = gpu(p, idx, force_complex=(path == "complex"))
        du_k = np.abs(res["us"] - us).max(axis=(0, 1))
        print("cfg %d order %d B %d T %s %s: codes %s solves equal %s  |du| first %.2e max %.2e  |dx| max %.2e   exact-vs-clip |du| %.2e  oracle %.1fs"
              % (cfg, order, batch, horizon, path, res["exit_codes"], np.array_equal(res["qp_solves"], solves), du_k[0], du_k.max(),
                 np.abs(res["xs"] - xs).max(), np.abs(res["us"] - resc["us"]).max(), t1 - t0), flush=True)
        print("   fidelity-like final err exact %.3e clip %.3e" % (np.abs(res["xs"][:, :, -1] - p["X_targ"][None, :, p["n_steps"]]).max(),
                                                               np.abs(resc["xs"][:, :, -1] - p["X_targ"][None, :, p["n_steps"]]).max()))


def timing(cfg, batch):
    p = configs.build(cfg, batch=batch)
    idx = np.arange(batch)
    for ex in (False, True):
        for path in ("real", "complex"):
            sess_kw = dict(exact_qp=ex, force_complex=(path == "complex"))
            res = gpu(p, idx, **sess_kw)
            res = gpu(p, idx, **sess_kw)
            print("cfg %d B %d exact %s %s: kernel %.2f ms  solves %d  codes %s  stats %s" % (
                cfg, batch, ex, path, res.get("kernel_ms", float("nan")), int(res["qp_solves"].sum()), np.bincount(res["exit_codes"]),
                res["qp_stats"]), flush=True)


if __name__ == "__main__":
    parity(1, 1, 2, None)
    parity(3, 1, 3, 16)
    parity(3, 2, 2, None)
    parity(4, 1, 2, 12)
    timing(3, 65536)
