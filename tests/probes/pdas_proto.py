import sys, os
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/probes')
import active_set_proto as ap

def pdas(H, f, lo, hi, u0, maxit=60, init="grad"):
    u = np.clip(u0, lo, hi)
    eps = 1e-12 * np.max(np.abs(hi))
    g = H @ u + f
    side = np.zeros(len(u))          # +1 pinned at hi, -1 at lo, 0 free
    if init == "grad":
        side[(u <= lo + eps) & (g > 0)] = -1
        side[(u >= hi - eps) & (g < 0)] = 1
    seen = set()
    for it in range(1, maxit + 1):
        pinned = side != 0
        un = np.where(side > 0, hi, np.where(side < 0, lo, 0.0))
        fr = ~pinned
        if fr.any():
            un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pinned)] @ un[pinned]))
        mu = H @ un + f
        new = side.copy()
        new[fr & (un > hi)] = 1
        new[fr & (un < lo)] = -1
        new[(side > 0) & ~(mu < 0)] = 0
        new[(side < 0) & ~(mu > 0)] = 0
        if np.array_equal(new, side):
            return un, it, "kkt"
        key = new.tobytes()
        if key in seen:
            return un, it, "cycle"
        seen.add(key)
        side = new
    return un, maxit, "cap"

cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ap.capture_loop(cfg, order, members)
tot = {}
its = 0; its_dev = 0
for q in ap.CAPTURE:
    x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls = q["args"]
    m, T = U_bm.shape
    H, f, c = ap.condense(x_init, np.asarray(X_bm, dtype=complex), np.asarray(U_bm, dtype=float), Q_ls, R_ls, A_ls, B_ls, D_ls)
    lo = -q["sat"] * np.ones(T * m); hi = q["sat"] * np.ones(T * m)
    if q["du"] is not None and q["u_prev"] is not None:
        up = np.reshape(q["u_prev"], -1).real
        lo[:m] = np.maximum(lo[:m], up - q["du"]); hi[:m] = np.minimum(hi[:m], up + q["du"])
    u, it, why = pdas(H, f, lo, hi, q["U_guess"].T.reshape(-1))
    ud, sw, rt, whyd = ap.solve(H, f, c, lo, hi, q["U_guess"].T.reshape(-1), m)
    err = np.abs(u - q["U"].T.reshape(-1)).max()
    tot[why] = tot.get(why, 0) + 1
    its += it; its_dev += sw
    if it > 3 or why != "kkt" or sw > 4:
        print("step %2d iter %d: pdas %3d (%s) err %.1e | device sweeps %3d ratios %3d" % (q["step"], q["n_iter"], it, why, err, sw, rt))
print("solves", len(ap.CAPTURE), tot, "pdas its/solve %.2f device sweeps/solve %.2f" % (its / len(ap.CAPTURE), its_dev / len(ap.CAPTURE)))
