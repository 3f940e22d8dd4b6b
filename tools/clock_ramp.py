#!/usr/bin/env python3
"""Launch time of the closed-loop kernel, launch by launch, from a cold (idle) GPU: how long the chip takes to settle at the clock it
holds under this load.    python3 tools/clock_ramp.py [--launches 300] [--complex]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=300)
ap.add_argument("--complex", action="store_true")
a = ap.parse_args()
B = 65536
p = configs.build(3, batch=B, host_models=False)
n, m, T, ns = p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
s = m4q.EnsembleSession(B, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=True, target_cols=ns + T + 1,
                        force_complex=a.complex)
s.build_models(p["dt"], p["generators"], p["scales"])
s.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
time.sleep(3.0)                                   # idle
ms = []
for i in range(a.launches):
    s.run(0, ns)
    s.sync()
    ms.append(s.kernel_ms()[0])
ms = np.array(ms)
t = np.cumsum(ms) / 1e3
print("path", s.path(), "launch ms: first 10:", " ".join("%.2f" % v for v in ms[:10]))
for lo in range(0, a.launches, 25):
    print("launches %3d-%3d (t = %5.1f s): mean %.3f ms" % (lo, min(lo + 25, a.launches) - 1, t[lo], ms[lo:lo + 25].mean()))
