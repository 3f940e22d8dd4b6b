"""Exploration: exact box QP on device (M4Q_QP_EXACT_BOX) against the BVLS oracle on linearisations of the
reference scenarios.  Prints control/cost gaps and timing; test infrastructure only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("M4Q_QP_TRACE", "1")
import mpc4quantum_amd as m4q            # noqa: E402
from mpc4quantum_amd import configs      # noqa: E402
from oracle import m4q_oracle as orc     # noqa: E402


def problem(config, order, T, Bn, seed, sat_scale, amp):
    rng = np.random.default_rng(seed)
    p = configs.build(config, batch=Bn, order=order, horizon=T)
    n, m = p["x0"].shape[1], p["U_targ"].shape[0]
    A, Bm, D, x0 = [], [], [], []
    for b in range(Bn):
        mod = p["models"][b if p["models"].shape[0] > 1 else 0]
        wm = orc.OracleWrapModel(mod[:, :n], mod[:, n:], m, order)
        xg = np.tile(p["x0"][b][:, None], (1, T + 1))
        ug = amp * p["sat"] * rng.uniform(-1, 1, (m, T))
        Ao, Bo, Do = wm.get_model_along_traj(xg, ug, np.arange(T))
        A.append(np.stack(Ao)); Bm.append(np.stack(Bo)); D.append(np.stack(Do).reshape(T, n)); x0.append(xg[:, 0])
    A, Bm, D, x0 = np.stack(A), np.stack(Bm), np.stack(D), np.stack(x0)
    Xb = p["X_targ"][:, :T + 1].T[None]
    Ub = p["U_targ"][:, :T].T[None].real
    Qs = np.stack([p["Q"]] * T + [p["Qf"]]).astype(complex)
    Rs = np.stack([p["R"]] * T).astype(complex)
    return dict(A=A, Bm=Bm, D=D, x0=x0, Xb=Xb, Ub=Ub, Qs=Qs, Rs=Rs, sat=p["sat"] * sat_scale, n=n, m=m, T=T)


def run(config, order, T, Bn, sat_scale, amp, du=None):
    q = problem(config, order, T, Bn, 11 + config, sat_scale, amp)
    up = np.zeros((Bn, q["m"])) if du else None
    t0 = time.time()
    X, U, cost, _ = m4q.quad_program_batch(q["x0"], q["Xb"], q["Ub"], q["Qs"], q["Rs"], q["A"], q["Bm"], q["D"], up, q["sat"], du,
                                           exact=True)
    t1 = time.time()
    Xc, Uc, costc, _ = m4q.quad_program_batch(q["x0"], q["Xb"], q["Ub"], q["Qs"], q["Rs"], q["A"], q["Bm"], q["D"], up, q["sat"],
                                              du)
    worst_u = worst_c = 0.0
    nact = 0
    for b in range(min(Bn, 6)):
        Xe, Ue, ce = orc.exact_quad_program(q["x0"][b], q["Xb"][0].T, q["Ub"][0].T, list(q["Qs"]), list(q["Rs"]), list(q["A"][b]),
                                            list(q["Bm"][b]), list(q["D"][b]), None if up is None else up[b], q["sat"], du)
        worst_u = max(worst_u, np.abs(U[b].T - Ue).max() / q["sat"])
        worst_c = max(worst_c, (cost[b] - ce) / max(1.0, abs(ce)))
        nact += int((np.abs(Ue) >= q["sat"] * (1 - 1e-9)).sum())
    print("config %d order %d T %d B %d sat*%.2f du %s: |U-Ubvls|/sat %.2e  (J-Jbvls)/J %.2e  active %d  clip-vs-exact cost gap %.2e  %.2fs"
          % (config, order, T, Bn, sat_scale, du, worst_u, worst_c, nact, float(np.max((costc - cost) / np.maximum(1, cost))), t1 - t0),
          flush=True)


if __name__ == "__main__":
    run(1, 1, 10, 8, 1.0, 0.3)
    run(1, 1, 10, 8, 0.3, 0.3)
    run(1, 2, 25, 8, 0.3, 0.3)
    run(3, 2, 10, 8, 0.3, 0.3)
    run(3, 2, 40, 8, 0.3, 0.3)
    run(3, 2, 40, 8, 0.15, 0.1, du=None)
    run(3, 2, 20, 8, 0.3, 0.3, du=0.05)
    run(4, 1, 20, 8, 0.3, 0.3)
