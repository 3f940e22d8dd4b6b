"""CPU prototype (dense, condensed) of SURVEY f2's other half: ADMM on the Riccati factorisation, then an active-set polish.

    min 1/2 u'Hu + f'u,  lo <= u <= hi        u+ = argmin J(u) + rho/2 |u - z + lam|^2   (ONE factorisation of H + rho I for all
                                              iterations: on the device, the gains of the LQR with R + rho/2 I; an iteration is an
                                              affine-only backward pass and a rollout, ~1/6 of a full sweep + rollout at n = 15)
                                              z+ = clip(u+ + lam),  lam+ = lam + u+ - z+
Run to a tolerance, read the active set off z, hand it to the face solve (one full sweep) and test KKT; on failure the device's
feasible active-set iteration (active_set_proto / arc_proto: solve_arc) continues from the ADMM point.  Counts: ADMM iterations,
full sweeps of the polish, and whether the set ADMM identified was the optimal one.
    python tests/probes/admm_proto.py <config> <order> <members> [rho ...]
Development tool; not used by the package, the tests or the bench."""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import active_set_proto as ap  # noqa: E402
import arc_proto  # noqa: E402


def load_capture(cfg, order, members):
    cache = "/tmp/arc_capture_%d_%d_%d.pkl" % (cfg, order, members)
    if os.path.exists(cache):
        return pickle.load(open(cache, "rb"))
    ap.CAPTURE.clear()
    ap.capture_loop(cfg, order, members)
    for q in ap.CAPTURE:
        x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls = q["args"]
        q["H"], q["f"], q["c"] = ap.condense(x_init, np.asarray(X_bm, dtype=complex), np.asarray(U_bm, dtype=float), Q_ls, R_ls, A_ls, B_ls, D_ls)
        q["shape"] = U_bm.shape
        del q["args"]
    pickle.dump(list(ap.CAPTURE), open(cache, "wb"))
    return list(ap.CAPTURE)


def bounds(q):
    m, T = q["shape"]
    lo = -q["sat"] * np.ones(T * m)
    hi = q["sat"] * np.ones(T * m)
    if q["du"] is not None and q["u_prev"] is not None:
        up = np.reshape(q["u_prev"], -1).real
        lo[:m] = np.maximum(lo[:m], up - q["du"])
        hi[:m] = np.minimum(hi[:m], up + q["du"])
    return lo, hi


def admm(H, f, lo, hi, u0, rho, iters, tol, adapt=False):
    """Returns (z, lam, iterations, history of active-set changes)."""
    n = len(f)
    z = np.clip(u0, lo, hi)
    lam = np.zeros(n)
    L = np.linalg.cholesky(H + rho * np.eye(n))
    refactor = 0
    last_set = None
    stable = 0
    for k in range(1, iters + 1):
        rhs = -f + rho * (z - lam)
        u = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        zn = np.clip(u + lam, lo, hi)
        lam = lam + u - zn
        r_prim = np.abs(u - zn).max()
        r_dual = rho * np.abs(zn - z).max()
        z = zn
        cur = np.sign((z >= hi) * 1.0 - (z <= lo) * 1.0)
        if last_set is not None and np.array_equal(cur, last_set):
            stable += 1
        else:
            stable = 0
        last_set = cur
        if adapt and k % 10 == 0 and r_dual > 0 and r_prim > 0:
            ratio = np.sqrt(r_prim / (r_dual / rho) ) if False else np.sqrt((r_prim) / (r_dual))
            if ratio > 5 or ratio < 0.2:
                new = rho * min(max(ratio, 0.1), 10.0)
                lam = lam * rho / new
                rho = new
                L = np.linalg.cholesky(H + rho * np.eye(n))
                refactor += 1
        if r_prim <= tol * max(1.0, np.abs(hi).max()) and r_dual <= tol * max(1.0, np.abs(f).max()):
            return z, lam * rho, k, refactor, stable
    return z, lam * rho, iters, refactor, stable


def polish(H, f, lo, hi, z, mu):
    """Active set from the ADMM point: on a bound with a multiplier of the right sign.  One face solve; KKT test."""
    side = np.zeros(len(z))
    eps = 1e-9 * np.abs(hi).max()
    side[(z >= hi - eps)] = 1
    side[(z <= lo + eps)] = -1
    pinned = side != 0
    pv = np.where(side > 0, hi, lo)
    un = np.where(pinned, pv, 0.0)
    fr = ~pinned
    if fr.any():
        un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pinned)] @ un[pinned]))
    g = H @ un + f
    feas = ((un <= hi + 1e-12) & (un >= lo - 1e-12)).all()
    signs = not (((side > 0) & (g > 1e-12)) | ((side < 0) & (g < -1e-12))).any()
    return un, feas and signs


if __name__ == "__main__":
    cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rhos = [float(x) for x in sys.argv[4:]] or [1e-2, 1e-1, 1.0, 10.0]
    cap = load_capture(cfg, order, members)
    print("config %d order %d: %d QPs" % (cfg, order, len(cap)))
    for q in cap[:1]:
        w = np.linalg.eigvalsh(q["H"])
        print("H eigenvalues: min %.2e max %.2e cond %.1e" % (w.min(), w.max(), w.max() / w.min()))
    for adapt in (False, True):
        for rho in rhos:
            for tol in (1e-3, 1e-5):
                its, ok1, tot_sweeps, worst, hist = [], 0, 0, 0.0, []
                for q in cap:
                    m, T = q["shape"]
                    H, f, c = q["H"], q["f"], q["c"]
                    lo, hi = bounds(q)
                    u0 = q["U_guess"].T.reshape(-1)
                    z, mu, k, refac, stable = admm(H, f, lo, hi, u0, rho, 2000, tol, adapt)
                    un, ok = polish(H, f, lo, hi, z, mu)
                    sweeps = 1 + refac                      # the ADMM factorisation(s) + the polish's face solve
                    sweeps += 1
                    if ok:
                        ok1 += 1
                        u = un
                    else:
                        u, sw, rt, ev, why = arc_proto.solve_arc(H, f, c, lo, hi, z, m)
                        sweeps += sw
                    err = np.abs(u - q["U"].T.reshape(-1)).max()
                    worst = max(worst, err / q["sat"])
                    its.append(k)
                    tot_sweeps += sweeps
                    hist.append((q["step"], k, sweeps))
                its = np.array(its)
                n = len(cap)
                print("adapt %d rho %-6g tol %.0e: ADMM iterations mean %.0f median %.0f max %d | first polish optimal %d/%d | full sweeps/solve %.2f "
                      "| sweep-equivalents/solve (iteration = 1/6 sweep) %.2f | worst |u - bvls|/sat %.1e" % (
                          adapt, rho, tol, its.mean(), np.median(its), its.max(), ok1, n, tot_sweeps / n, tot_sweeps / n + its.mean() / 6.0, worst))
                if "-v" in sys.argv:
                    print("    (step, ADMM iterations, full sweeps):", hist)
