import sys, numpy as np
sys.path.insert(0, '/root/repo')
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
from oracle import m4q_oracle as orc
p = configs.build(3, batch=1, horizon=3, n_steps=2)
clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
models = p["models"]
for kw in ({"exact_qp": True}, {}):
    res = m4q.mpc_batch(p["x0"], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], **kw)
    tr = []
    xs, us, codes, solves = orc.mpc_batch(p["x0"], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"], p["horizon"], p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"], p["R"], p["Qf"], p["sat"], p["du"], qp_mode="exact" if kw else "qp", trace=tr)
    print(kw, "gpu solves", res["qp_solves"], "oracle", solves, "us gpu", res["us"], "oracle", us, "stats", res["qp_stats"])
