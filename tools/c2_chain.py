import sys
sys.path.insert(0, ".")
import numpy as np
import mpc4quantum_amd as m4q
from mpc4quantum_amd import _lib, configs
p = configs.build(2, batch=8192, host_models=False)
n, m, T, ns = p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
s = m4q.EnsembleSession(8192, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=True, target_cols=ns + T + 1)
s.build_models(p["dt"], p["generators"], p["scales"])
s.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
s.run(0, ns); s.sync()
q = s.download(_lib.F_QP_SOLVES, (8192, ns)).astype(int)
tot = q.sum(axis=1)
print("per member solves: mean %.1f max %d  p99 %d; head (steps 0-1) mean %.1f max %d; steps: %s" % (tot.mean(), tot.max(), np.percentile(tot, 99), q[:, :2].sum(1).mean(), q[:, :2].sum(1).max(), q.max(axis=0).tolist()))
print("members with > 100 solves:", int((tot > 100).sum()))
