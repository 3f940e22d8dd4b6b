import sys, numpy as np
sys.path.insert(0, ".")
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
p = configs.build(3, batch=3, order=1, horizon=16)
clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
res = m4q.mpc_batch(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], force_complex=True, exact_qp=True)
print("codes", res["exit_codes"], "steps_done", res["steps_done"], "solves", res["qp_solves"].sum(axis=1), "stats", res["qp_stats"], "ms", res["kernel_ms"])
