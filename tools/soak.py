#!/usr/bin/env python3
"""Soak of the shipped binary: closed-loop launches of every mode, over and over, each compared bit for bit with the first of its
mode (states, controls, solve counts, final SQP guesses).  A progress line per round.
    python3 tools/soak.py --minutes 8 [--batch 16384]"""
import argparse
import hashlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mpc4quantum_amd as m4q
from mpc4quantum_amd import _lib, configs

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=8.0)
ap.add_argument("--batch", type=int, default=16384)
a = ap.parse_args()
MODES = [("config3 real (tile sweep: default)", 3, {}), ("config3 complex", 3, {"force_complex": True}), ("config3 exact real (pinned sweep on tiles: default)", 3, {"exact_qp": True}),
         ("config3 exact real, DPP sweeps", 3, {"exact_qp": True, "tile": False}),
         ("config3 exact complex", 3, {"exact_qp": True, "force_complex": True}), ("config4 real (shared generators: default)", 4, {}),
         ("config4 real, per-member models", 4, {"shared_generators": False}),
         ("config4 complex", 4, {"force_complex": True}), ("config2 real", 2, {}), ("config5 real (T=80)", 5, {}),
         ("config3 real, 9 coordinates", 3, {"traceless": False}), ("config3 real, DPP sweeps", 3, {"tile": False}),
         ("config4 exact real", 4, {"exact_qp": True}), ("config2 exact real", 2, {"exact_qp": True})]
sessions = []
for name, cfg, kw in MODES:
    B = a.batch if "exact complex" not in name and "config4 complex" not in name and "config4 exact" not in name else a.batch // 4
    p = configs.build(cfg, batch=B, host_models=False)
    n, m, T, ns = p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
    s = m4q.EnsembleSession(B, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=True, target_cols=ns + T + 1, **kw)
    s.build_models(p["dt"], p["generators"], p["scales"])
    s.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
    sessions.append((name, s, B, ns))


def digest(s, B, ns):
    st = s.state()
    h = hashlib.sha256()
    for key in ("xs", "us", "x_guess", "u_guess", "exit_codes"):
        h.update(np.ascontiguousarray(st[key]).tobytes())
    h.update(np.ascontiguousarray(s.download(_lib.F_QP_SOLVES, (B, ns))).tobytes())
    return h.hexdigest()


first = {}
t0 = time.time()
rounds = launches = 0
while time.time() - t0 < 60 * a.minutes:
    for name, s, B, ns in sessions:
        s.run(0, ns)
        d = digest(s, B, ns)
        launches += 1
        if name not in first:
            first[name] = d
        elif d != first[name]:
            print("MISMATCH in %s at round %d" % (name, rounds), flush=True)
            sys.exit(1)
    rounds += 1
    print("round %d: %d launches, all bit-identical to their first (%.0f s)" % (rounds, launches, time.time() - t0), flush=True)
for name, d in first.items():
    print("%-36s %s" % (name, d))
print("soak ok: %d rounds, %d launches, %d modes" % (rounds, launches, len(MODES)))
