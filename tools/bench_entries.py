#!/usr/bin/env python3
"""The step-by-step entry points of the C ABI (rows a7, a10-a12, a16, a17: what a host loop that keeps the reference's own mpc()
calls one at a time) on a config-3-shaped ensemble: wall time per call with host buffers (PCIe both ways, as the ABI hands them over)
and, under `rocprofv3 --kernel-trace --stats -- python3 tools/bench_entries.py`, the kernel's own duration in the stats file.
    python3 tools/bench_entries.py [--batch 16384] [--config 3]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
from mpc4quantum_amd.experiment import plant_step_batch
from mpc4quantum_amd.optimize import quad_program_batch
from mpc4quantum_amd.vectorize import discretize_homogeneous_batch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--config", type=int, default=3)
a = ap.parse_args()
B = a.batch
p = configs.build(a.config, batch=B, host_models=False)
n, m, T = p["dim_x"], p["dim_u"], p["horizon"]
rng = np.random.default_rng(0)


def timed(name, fn, units, unit_name, bytes_moved):
    fn()
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - t0)
    best = min(t)
    print("%-28s %8.2f ms per call (host buffers in and out)  %.3g %s/s  %.2f GB over PCIe" % (name, 1e3 * best, units / best, unit_name,
                                                                                             bytes_moved / 1e9), flush=True)
    return out


models = timed("m4q_discretize_batch", lambda: discretize_homogeneous_batch(list(p["generators"]), p["dt"], p["order"], scales=p["scales"]),
               B, "models", B * n * n * (1 + m) * 16)
wm = m4q.WrapModel(models[0][:, :n], models[0][:, n:], m, p["order"])
X = np.broadcast_to(p["x0"][:, None, :], (B, T, n)).copy() + 1e-3 * (rng.standard_normal((B, T, n)) + 0j)
U = 0.1 * p["sat"] * rng.standard_normal((B, T, m))
A_ls, B_ls, D_ls = timed("m4q_linearize_batch", lambda: wm.linearize_batch(X, U), B * T, "linearisations", B * T * (n * n + n * m + n) * 16)
Xb = np.broadcast_to(p["X_targ"][:, :T + 1].T[None], (1, T + 1, n))
Ub = np.zeros((1, T, m))
Q_ls = np.stack([p["Q"]] * T + [p["Qf"]])
R_ls = np.stack([p["R"]] * T)
timed("m4q_quad_program_batch", lambda: quad_program_batch(p["x0"], Xb, Ub, Q_ls, R_ls, A_ls, B_ls, D_ls, sat=p["sat"]), B * T, "horizon-steps",
      B * T * (n * n + n * m + n) * 16 + B * T * (n + 1) * m * 16)
timed("m4q_quad_program_batch exact", lambda: quad_program_batch(p["x0"], Xb, Ub, Q_ls, R_ls, A_ls, B_ls, D_ls, sat=p["sat"], exact=True), B * T,
      "horizon-steps", B * T * (n * n + n * m + n) * 16 + B * T * (n + 1) * m * 16)
u = 0.5 * p["sat"] * rng.standard_normal((B, m))
timed("m4q_plant_step_batch", lambda: plant_step_batch(p["x0"], u, p["plant_op0"], p["plant_ops"], p["dt"]), B, "plant steps", 2 * B * n * 16)
