"""Probe (GPU): how the exact box-QP solves of BASELINE config 5 (T = 80) end, by ensemble size."""
import sys
import numpy as np
sys.path.insert(0, ".")
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs

for B in (16, 256, 4096):
    p = configs.build(5, batch=B)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = m4q.mpc_batch(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"],
                        p["Q"], p["R"], p["Qf"], p["sat"], p["du"], exact_qp=True)
    codes = res["exit_codes"]
    print("B", B, "qp_stats (solves, sweeps, ratio, kkt, precision, cap)", res["qp_stats"], "codes", np.bincount(codes, minlength=4).tolist(),
          "first capped members", np.nonzero(codes == 2)[0][:8].tolist(), "steps_done of those", res["steps_done"][codes == 2][:8].tolist(),
          "ms", round(res["kernel_ms"], 1), flush=True)
