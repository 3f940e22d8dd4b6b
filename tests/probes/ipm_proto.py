"""CPU prototype (dense, condensed): a primal-dual interior-point iteration (Mehrotra predictor-corrector) for the box QP of the
exact mode - how many Newton systems (each ONE Riccati sweep with stage-varying R_t + diag(d_t) on the device) it takes to reach the
active set, on the QPs captured from an oracle closed loop; then the active-set polish (one face solve + KKT test, else the
device's feasible iteration from the IPM point clipped into the box).
    python tests/probes/ipm_proto.py <config> <order> <members>
Development tool; not used by the package, the tests or the bench."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import admm_proto as adp  # noqa: E402
import arc_proto  # noqa: E402


def ipm(H, f, lo, hi, u0, tol, iters=60):
    n = len(f)
    w = hi - lo
    fixed = w <= 1e-14
    u = np.clip(u0, lo + 0.1 * w, hi - 0.1 * w)
    sl, su = u - lo, hi - u                  # slacks
    sl[fixed] = su[fixed] = 1.0
    zl = np.ones(n); zu = np.ones(n)         # duals
    mu0 = None
    for k in range(1, iters + 1):
        g = H @ u + f - zl + zu
        mu = (sl @ zl + su @ zu) / (2 * n)
        if mu0 is None:
            mu0 = mu
        res = max(np.abs(g[~fixed]).max() if (~fixed).any() else 0.0, mu)
        if np.abs(g[~fixed]).max() <= tol * max(1.0, np.abs(f).max()) and mu <= tol * 1e-2:
            return u, k - 1
        D = zl / sl + zu / su
        M = H + np.diag(D)
        M[fixed, :] = 0; M[:, fixed] = 0; M[fixed, fixed] = 1.0
        L = np.linalg.cholesky(M)
        solve = lambda r: np.linalg.solve(L.T, np.linalg.solve(L, r))

        def step(sigma_mu, corr_l, corr_u):
            # complementarity: sl zl = sigma mu - corr  ->  dzl = (sigma mu - corr - sl zl - zl du) / sl ; su: dsu = -du
            rl = sigma_mu - corr_l - sl * zl
            ru = sigma_mu - corr_u - su * zu
            r = -g + rl / sl - ru / su
            r[fixed] = 0.0
            du = solve(r)
            dzl = (rl - zl * du) / sl
            dzu = (ru + zu * du) / su
            return du, dzl, dzu

        def maxstep(v, dv):
            neg = dv < 0
            return min(1.0, (-(v[neg]) / dv[neg]).min()) if neg.any() else 1.0
        du, dzl, dzu = step(0.0, 0.0, 0.0)
        ap_ = min(maxstep(sl, du), maxstep(su, -du)); ad = min(maxstep(zl, dzl), maxstep(zu, dzu))
        mu_aff = ((sl + ap_ * du) @ (zl + ad * dzl) + (su - ap_ * du) @ (zu + ad * dzu)) / (2 * n)
        sigma = (mu_aff / mu) ** 3
        du, dzl, dzu = step(sigma * mu, du * dzl, -du * dzu)
        ap_ = 0.995 * min(maxstep(sl, du), maxstep(su, -du)); ad = 0.995 * min(maxstep(zl, dzl), maxstep(zu, dzu))
        u = u + ap_ * du
        sl = sl + ap_ * du; su = su - ap_ * du
        zl = zl + ad * dzl; zu = zu + ad * dzu
        sl[fixed] = su[fixed] = 1.0
    return u, iters


if __name__ == "__main__":
    cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    cap = adp.load_capture(cfg, order, members)
    print("config %d order %d: %d QPs" % (cfg, order, len(cap)))
    for tol in (1e-4, 1e-6, 1e-8):
        its, ok1, tot, worst, hist = [], 0, 0, 0.0, []
        for q in cap:
            m, T = q["shape"]
            H, f, c = q["H"], q["f"], q["c"]
            lo, hi = adp.bounds(q)
            u, k = ipm(H, f, lo, hi, q["U_guess"].T.reshape(-1), tol)
            # active set: within 1e-6 of the width of a bound
            w = np.maximum(hi - lo, 1e-300)
            z = u.copy()
            z[(hi - u) / w < 1e-4] = hi[(hi - u) / w < 1e-4]
            z[(u - lo) / w < 1e-4] = lo[(u - lo) / w < 1e-4]
            un, ok = adp.polish(H, f, lo, hi, z, None)
            sweeps = k + 1
            if ok:
                ok1 += 1
                uf = un
            else:
                uf, sw, rt, ev, why = arc_proto.solve_arc(H, f, c, lo, hi, np.clip(z, lo, hi), m)
                sweeps += sw
            worst = max(worst, np.abs(uf - q["U"].T.reshape(-1)).max() / q["sat"])
            its.append(k); tot += sweeps; hist.append((q["step"], k, sweeps))
        its = np.array(its)
        print("tol %.0e: IPM iterations mean %.1f median %.0f max %d | first polish optimal %d/%d | full sweeps/solve (IPM + polish) %.2f, max %d | "
              "worst |u - bvls|/sat %.1e" % (tol, its.mean(), np.median(its), its.max(), ok1, len(cap), tot / len(cap), max(h[2] for h in hist), worst))
        if "-v" in sys.argv:
            print("   ", hist)
