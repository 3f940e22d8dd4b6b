import sys, numpy as np
sys.path.insert(0, '/root/repo')
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
from oracle import m4q_oracle as orc

def run(cfg, B, T, ns, **kw):
    p = configs.build(cfg, batch=B, horizon=T, n_steps=ns)
    idx = np.arange(B)
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = m4q.mpc_batch(p["x0"], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], **kw)
    xs, us, codes, solves = orc.mpc_batch(p["x0"], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"], p["horizon"], p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"], p["R"], p["Qf"], p["sat"], p["du"], qp_mode="exact" if kw.get("exact_qp") else "qp")
    print("cfg %d B %d T %d ns %d %s: codes %s/%s solves eq %s  du %.2e dx %.2e" % (cfg, B, T, ns, kw, res["exit_codes"], codes, np.array_equal(res["qp_solves"], solves), np.abs(res["us"] - us).max(), np.abs(res["xs"] - xs).max()), flush=True)

for kw in ({}, {"exact_qp": True}, {"force_complex": True}, {"exact_qp": True, "force_complex": True}):
    run(1, 1, 1, 1, **kw)
    run(1, 3, 2, 1, **kw)
    run(3, 5, 1, 3, **kw)
    run(3, 1, 3, 2, **kw)
    run(4, 2, 2, 2, **kw)
