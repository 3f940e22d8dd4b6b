"""The exact box-QP iteration of the device (csrc/m4q_mpc.h: box_qp_iterate), in its condensed dense CPU form
(tests/probes/active_set_proto.py, pdas_switch_proto.py), against the oracle's independent solver (SciPy BVLS) on the QPs of a
closed loop: the feasible active-set iteration alone, and with the primal-dual phase that takes over after the first clipped
trial that does not lower J.  Both must reach the same optimum (the saving of the second shows on the hard solves of the
full-horizon configurations: profiles/r03_exact_qp_log.txt; on this small case the two need about the same number of sweeps)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
import active_set_proto as ap            # noqa: E402
import pdas_switch_proto as sw           # noqa: E402


def _qps(cfg, order, members, horizon):
    ap.CAPTURE.clear()
    ap.capture_loop(cfg, order, members, horizon=horizon)
    out = []
    for q in ap.CAPTURE:
        x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls = q["args"]
        m, T = U_bm.shape
        H, f, c = ap.condense(x_init, np.asarray(X_bm, dtype=complex), np.asarray(U_bm, dtype=float), Q_ls, R_ls, A_ls, B_ls, D_ls)
        lo = -q["sat"] * np.ones(T * m)
        hi = q["sat"] * np.ones(T * m)
        if q["du"] is not None and q["u_prev"] is not None:
            up = np.reshape(q["u_prev"], -1).real
            lo[:m] = np.maximum(lo[:m], up - q["du"])
            hi[:m] = np.minimum(hi[:m], up + q["du"])
        out.append((H, f, c, lo, hi, q["U_guess"].T.reshape(-1), m, q["U"].T.reshape(-1)))
    return out


def test_active_set_iterations_reach_the_bvls_optimum():
    qps = _qps(3, 1, 1, 12)                       # config 3's model, T = 12: 20 MPC steps of one member, a few dozen QPs
    assert len(qps) >= 20
    sweeps = {0: 0, 40: 0}
    for H, f, c, lo, hi, u0, m, u_ref in qps:
        for cap in (0, 40):                       # 0: the feasible iteration alone; 40: with the primal-dual phase (the device's cap)
            u, s, ratios, pd, why = sw.solve_switch(H, f, c, lo, hi, u0, m, PCAP=cap, fail_thresh=1 if cap else 99)
            assert why in ("kkt", "kkt-pdas"), why
            assert np.abs(u - u_ref).max() <= 1e-7 * hi.max(), (cap, np.abs(u - u_ref).max())
            assert np.all(u <= hi + 1e-12) and np.all(u >= lo - 1e-12)
            sweeps[cap] += s
    assert sweeps[40] <= 1.15 * sweeps[0]
