"""CPU prototype: what to do when the clipped policy trial of the device's exact box-QP iteration fails to lower J.
Round 3: a primal-dual phase (cap 40), then ratio steps that add one or two constraints each - config 4 (three controls) needs
58-120 sweeps on its hard solves.  Here: a projection-arc search along the face-minimiser direction (Bertsekas' two-metric
projection: u(a) = clip(u + a (u_N - u)), a = 1, 1/2, 1/4, ...; accepted when J decreases) before any ratio step.
    python tests/probes/arc_proto.py <config> <order> <members>
Development tool; not used by the package, the tests or the bench."""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import active_set_proto as ap  # noqa: E402


def solve_arc(H, f, c, lo, hi, u0, m, arcs=(1.0, 0.5, 0.25, 0.125, 0.0625), use_pdas=0, trial=True):
    J = lambda u: 0.5 * u @ H @ u + f @ u + c
    u = np.clip(u0, lo, hi)
    Jk = J(u)
    eps = 1e-12 * np.max(np.abs(hi))
    need_adj, fmin = True, False
    side = np.zeros(len(u))
    sweeps = ratios = evals = 0

    def kkt_ok(un, side):
        mu = H @ un + f
        return not (((side > 0) & ~(mu < 0)) | ((side < 0) & ~(mu > 0))).any()
    for it in range(400):
        if need_adj:
            g = H @ u + f
            new = np.zeros(len(u))
            new[(u <= lo + eps) & (g > 0)] = -1
            new[(u >= hi - eps) & (g < 0)] = 1
            if fmin and np.array_equal(new, side):
                return u, sweeps, ratios, evals, "kkt"
            side = new
        need_adj, fmin = False, False
        pinned = side != 0
        pin_val = np.where(side > 0, hi, lo)
        sweeps += 1
        if trial:
            uc, clipped = ap.closed_loop_clipped(H, f, pinned, np.where(pinned, pin_val, 0.0), lo, hi, m)
            Jc = J(uc)
            if np.abs(uc - u).max() <= 1e-13 * np.max(hi):
                if not clipped and kkt_ok(uc, side):
                    return uc, sweeps, ratios, evals, "kkt"
                fmin, need_adj = True, True
                continue
            if Jc < Jk or (not clipped and Jc <= Jk + 1e-12 * abs(Jk)):
                u, Jk = uc, Jc
                if not clipped and kkt_ok(uc, side):
                    return uc, sweeps, ratios, evals, "kkt"
                fmin, need_adj = not clipped, True
                continue
            if not clipped:
                return u, sweeps, ratios, evals, "precision"
        # face minimiser (same sweep: the unclipped rollout)
        un = np.where(pinned, pin_val, 0.0)
        fr = ~pinned
        un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pinned)] @ un[pinned]))
        inside = ((un <= hi + eps) & (un >= lo - eps)).all()
        if inside:
            if np.abs(un - u).max() <= 1e-13 * np.max(hi):
                fmin, need_adj = True, True
                continue
            Jn = J(un)
            u, Jk = un, Jn
            if kkt_ok(un, side):
                return un, sweeps, ratios, evals, "kkt"
            fmin, need_adj = True, True
            continue
        d = un - u
        took = False
        for a in arcs:
            ut = np.clip(u + a * d, lo, hi)
            evals += 1
            Jt = J(ut)
            if Jt < Jk - 1e-14 * abs(Jk):
                u, Jk, took = ut, Jt, True
                break
        if took:
            need_adj = True
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            a_hi = np.where(un > hi, (hi - u) / d, np.inf)
            a_lo = np.where(un < lo, (lo - u) / d, np.inf)
        av = np.minimum(a_hi, a_lo)
        al = min(1.0, av.min())
        ratios += 1
        hit = av <= al + 1e-14
        u = u + al * d
        for i in np.flatnonzero(hit):
            u[i] = hi[i] if un[i] > hi[i] else lo[i]
            side[i] = 1 if un[i] > hi[i] else -1
        Jk = J(u)
    return u, sweeps, ratios, evals, "cap"


if __name__ == "__main__":
    cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    cache = "/tmp/arc_capture_%d_%d_%d.pkl" % (cfg, order, members)
    if os.path.exists(cache):
        ap.CAPTURE.extend(pickle.load(open(cache, "rb")))
    else:
        ap.capture_loop(cfg, order, members)
        for q in ap.CAPTURE:
            x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls = q["args"]
            q["H"], q["f"], q["c"] = ap.condense(x_init, np.asarray(X_bm, dtype=complex), np.asarray(U_bm, dtype=float), Q_ls, R_ls, A_ls, B_ls, D_ls)
            q["shape"] = U_bm.shape
            del q["args"]
        pickle.dump(ap.CAPTURE, open(cache, "wb"))
    variants = {"arc after failed trial": dict(), "arc, no clipped trial": dict(trial=False), "arc 1, 1/4, 1/16": dict(arcs=(1.0, 0.25, 0.0625)),
                "no arc (ratio only)": dict(arcs=())}
    for name, kw in variants.items():
        tot = {}; S = R = E = 0; worst = 0.0; hard = []
        for q in ap.CAPTURE:
            m, T = q["shape"]
            H, f, c = q["H"], q["f"], q["c"]
            lo = -q["sat"] * np.ones(T * m); hi = q["sat"] * np.ones(T * m)
            if q["du"] is not None and q["u_prev"] is not None:
                up = np.reshape(q["u_prev"], -1).real
                lo[:m] = np.maximum(lo[:m], up - q["du"]); hi[:m] = np.minimum(hi[:m], up + q["du"])
            u, sw, rt, ev, why = solve_arc(H, f, c, lo, hi, q["U_guess"].T.reshape(-1), m, **kw)
            err = np.abs(u - q["U"].T.reshape(-1)).max()
            worst = max(worst, err)
            tot[why] = tot.get(why, 0) + 1
            S += sw; R += rt; E += ev
            if sw > 12:
                hard.append((q["step"], sw, rt, ev))
        n = len(ap.CAPTURE)
        print("%-26s solves %d %s sweeps/solve %.2f ratios/solve %.2f J-evaluations/solve %.2f worst err %.1e   hard: %s" % (
            name, n, tot, S / n, R / n, E / n, worst, hard[:8]))
