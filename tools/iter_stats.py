import numpy as np, sys
sys.path.insert(0,'.')
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
p = configs.build(3)
clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
res = m4q.mpc_batch(p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"])
s = res["qp_solves"]
print("mean solves/inst", s.sum(1).mean(), "step0 mean/std/min/max", s[:,0].mean(), s[:,0].std(), s[:,0].min(), s[:,0].max(), "step1", s[:,1].mean(), s[:,1].std(), s[:,1].max())
q = s.reshape(-1,4,s.shape[1])
qmax = q.max(1)            # per quad per step iterations executed by the wave
print("wave-executed solves per quad", qmax.sum(1).mean(), " vs mean per instance", s.sum(1).mean(), " ratio", qmax.sum(1).mean()/s.sum(1).mean())
w = qmax.sum(1).reshape(-1)   # per quad cost
# static striding over 2048 waves
cost = w.reshape(-1,2048).sum(0) if w.size%2048==0 else None
print("static per-wave cost: mean %.1f max %.1f  imbalance %.3f" % (cost.mean(), cost.max(), cost.max()/cost.mean()))
print(np.bincount(s[:,0])[:60])
