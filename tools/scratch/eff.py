import os, sys
sys.path.insert(0, '/root/repo')
os.environ["M4Q_QP_TRACE"] = "1"
import numpy as np
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
for cfg in (3,):
    B = 65536
    p = configs.build(cfg, batch=B)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    for rep in range(2):
        res = m4q.mpc_batch(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], exact_qp=True)
    print(cfg, res["kernel_ms"], res["qp_stats"])
