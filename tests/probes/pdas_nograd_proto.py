import sys, os
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/probes')
import active_set_proto as ap

def clipped_rollout(H, f, side, lo, hi, m):
    """as ap.closed_loop_clipped, also returns which free controls were clipped (+1/-1)"""
    N = len(f)
    pinned = side != 0
    u = np.where(side > 0, hi, np.where(side < 0, lo, 0.0))
    decided = pinned.copy()
    clip = np.zeros(N)
    for t0 in range(0, N, m):
        rest = ~decided
        if not rest.any():
            break
        sol = np.linalg.solve(H[np.ix_(rest, rest)], -(f[rest] + H[np.ix_(rest, decided)] @ u[decided]))
        idx = np.flatnonzero(rest)
        for i, v in zip(idx, sol):
            if t0 <= i < t0 + m:
                if v > hi[i]: clip[i] = 1
                if v < lo[i]: clip[i] = -1
                u[i] = min(max(v, lo[i]), hi[i])
                decided[i] = True
    return u, clip

def solve(H, f, c, lo, hi, u0, m, mode, PCAP=40):
    """mode 'grad': device (gradient-derived sets) + pdas phase.  mode 'nograd': W0 = controls on a bound at the start; after an accepted
    clipped trial W' = W + clipped - pinned with wrong-sign gradient at the new iterate ... no adjoint pass anywhere"""
    J = lambda u: 0.5 * u @ H @ u + f @ u + c
    u = np.clip(u0, lo, hi)
    Jk = J(u)
    eps = 1e-12 * np.max(np.abs(hi))
    side = np.zeros(len(u))
    sweeps = ratios = pd = adj = 0
    def derive(u):
        g = H @ u + f
        new = np.zeros(len(u))
        new[(u <= lo + eps) & (g > 0)] = -1
        new[(u >= hi - eps) & (g < 0)] = 1
        return new
    if mode == 'grad':
        side = derive(u); adj += 1
    else:
        side[u <= lo + eps] = -1
        side[u >= hi - eps] = 1
    pdas_done = False
    fmin = False
    for it in range(400):
        uc, clip = clipped_rollout(H, f, side, lo, hi, m)
        clipped = (clip != 0).any()
        sweeps += 1
        Jc = J(uc)
        g = H @ uc + f                         # (device: multiplier rows along the policy; here the exact gradient at the trial)
        wrong = ((side > 0) & ~(g < 0)) | ((side < 0) & ~(g > 0))
        nomove = np.abs(uc - u).max() <= 1e-13 * np.max(hi)
        if not clipped and not wrong.any():
            return uc, sweeps, ratios, pd, adj, "kkt"
        accept = (not nomove) and (Jc < Jk or (not clipped and Jc <= Jk + 1e-12 * abs(Jk)))
        if accept or nomove:
            if accept:
                u, Jk = uc, Jc
            if mode == 'grad':
                new = derive(u); adj += 1
                if (not clipped) and np.array_equal(new, side):
                    return u, sweeps, ratios, pd, adj, "kkt"
            else:
                new = side.copy()
                new[clip > 0] = 1
                new[clip < 0] = -1
                for i in np.flatnonzero(wrong):
                    if not (clip[(i // m) * m:] != 0).any():
                        new[i] = 0
                if np.array_equal(new, side):
                    # nothing to change yet not optimal (clipped): fall back to the gradient rule
                    new = derive(u); adj += 1
                    if np.array_equal(new, side):
                        return u, sweeps, ratios, pd, adj, "stuck"
            side = new
            continue
        if not clipped:
            return u, sweeps, ratios, pd, adj, "precision"
        # failed trial -> primal-dual phase
        if not pdas_done:
            pdas_done = True
            s = side.copy()
            ok = False
            for k in range(PCAP):
                pn = s != 0
                un = np.where(s > 0, hi, np.where(s < 0, lo, 0.0))
                fr = ~pn
                if fr.any():
                    un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pn)] @ un[pn]))
                pd += 1
                if k > 0: sweeps += 1
                mu = H @ un + f
                new = s.copy()
                new[fr & (un > hi)] = 1
                new[fr & (un < lo)] = -1
                new[(s > 0) & ~(mu < 0)] = 0
                new[(s < 0) & ~(mu > 0)] = 0
                if np.array_equal(new, s):
                    return un, sweeps, ratios, pd, adj, "kkt-pdas"
                s = new
            side = derive(u); adj += 1
            continue
        pinned = side != 0
        un = np.where(side > 0, hi, np.where(side < 0, lo, 0.0))
        fr = ~pinned
        # classical ratio step needs the iterate ON the face: pinned controls at their bounds in u
        if np.abs(u[pinned] - un[pinned]).max(initial=0.0) > 0:
            side = derive(u); adj += 1
            pinned = side != 0
            un = np.where(side > 0, hi, np.where(side < 0, lo, 0.0)); fr = ~pinned
            sweeps += 1
        un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pinned)] @ un[pinned]))
        d = un - u
        with np.errstate(divide="ignore", invalid="ignore"):
            a_hi = np.where(un > hi, (hi - u) / d, np.inf)
            a_lo = np.where(un < lo, (lo - u) / d, np.inf)
        a = np.minimum(a_hi, a_lo)
        al = min(1.0, a.min())
        ratios += 1
        hit = a <= al + 1e-14
        u = u + al * d
        for i in np.flatnonzero(hit):
            u[i] = hi[i] if un[i] > hi[i] else lo[i]
            side[i] = 1 if un[i] > hi[i] else -1
        Jk = J(u)
    return u, sweeps, ratios, pd, adj, "cap"

if __name__ == "__main__":
    cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ap.capture_loop(cfg, order, members)
    for mode in ("grad", "nograd"):
        tot = {}; S = R = P = A = 0; worst = 0
        for q in ap.CAPTURE:
            x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls = q["args"]
            m, T = U_bm.shape
            if "H" not in q:
                q["H"], q["f"], q["c"] = ap.condense(x_init, np.asarray(X_bm, dtype=complex), np.asarray(U_bm, dtype=float), Q_ls, R_ls, A_ls, B_ls, D_ls)
            H, f, c = q["H"], q["f"], q["c"]
            lo = -q["sat"] * np.ones(T * m); hi = q["sat"] * np.ones(T * m)
            if q["du"] is not None and q["u_prev"] is not None:
                up = np.reshape(q["u_prev"], -1).real
                lo[:m] = np.maximum(lo[:m], up - q["du"]); hi[:m] = np.minimum(hi[:m], up + q["du"])
            u, sw, rt, pd, adj, why = solve(H, f, c, lo, hi, q["U_guess"].T.reshape(-1), m, mode)
            err = np.abs(u - q["U"].T.reshape(-1)).max()
            worst = max(worst, err)
            tot[why] = tot.get(why, 0) + 1
            S += sw; R += rt; P += pd; A += adj
        n = len(ap.CAPTURE)
        print("%s solves %d %s sweeps/solve %.2f ratios %.2f pdas-its %.2f adjoint passes %.2f worst err %.1e" % (mode, n, tot, S / n, R / n, P / n, A / n, worst))
