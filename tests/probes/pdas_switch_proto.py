import sys, os
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/probes'); sys.path.insert(0, '/tmp')
import active_set_proto as ap

def solve_switch(H, f, c, lo, hi, u0, m, PCAP=30, fail_thresh=1):
    J = lambda u: 0.5 * u @ H @ u + f @ u + c
    u = np.clip(u0, lo, hi)
    Jk = J(u)
    eps = 1e-12 * np.max(np.abs(hi))
    need_adj, fmin = True, False
    side = np.zeros(len(u))
    sweeps = ratios = pd = 0
    fails = 0
    pdas_done = False
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
                return u, sweeps, ratios, pd, "kkt"
            side = new
        need_adj, fmin = False, False
        pinned = side != 0
        pin_val = np.where(side > 0, hi, lo)
        uc, clipped = ap.closed_loop_clipped(H, f, pinned, np.where(pinned, pin_val, 0.0), lo, hi, m)
        sweeps += 1
        Jc = J(uc)
        if np.abs(uc - u).max() <= 1e-13 * np.max(hi):
            if not clipped and kkt_ok(uc, side):
                return uc, sweeps, ratios, pd, "kkt"
            fmin, need_adj = True, True
            continue
        if Jc < Jk or (not clipped and Jc <= Jk + 1e-12 * abs(Jk)):
            u, Jk = uc, Jc
            if not clipped and kkt_ok(uc, side):
                return uc, sweeps, ratios, pd, "kkt"
            fmin, need_adj = not clipped, True
            continue
        if not clipped:
            return u, sweeps, ratios, pd, "precision"
        fails += 1
        if fails >= fail_thresh and not pdas_done:
            # ---- primal-dual active set from the current working set; iterates need not be feasible
            pdas_done = True
            s = side.copy()
            for k in range(PCAP):
                pn = s != 0
                un = np.where(s > 0, hi, np.where(s < 0, lo, 0.0))
                fr = ~pn
                if fr.any():
                    un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pn)] @ un[pn]))
                pd += 1
                if k > 0:
                    sweeps += 1            # (the first one reuses the sweep just made)
                mu = H @ un + f
                new = s.copy()
                new[fr & (un > hi)] = 1
                new[fr & (un < lo)] = -1
                new[(s > 0) & ~(mu < 0)] = 0
                new[(s < 0) & ~(mu > 0)] = 0
                if np.array_equal(new, s):
                    return un, sweeps, ratios, pd, "kkt-pdas"
                s = new
            # gave up: classical iteration goes on from the feasible iterate
        un = np.where(pinned, pin_val, 0.0)
        fr = ~pinned
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
    return u, sweeps, ratios, pd, "cap"

if __name__ == "__main__":
    cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ap.capture_loop(cfg, order, members)
    for PCAP, ft in ((0, 99), (20, 1), (40, 1), (40, 2)):
        tot = {}; S = R = P = 0; worst = 0
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
            u, sw, rt, pd, why = solve_switch(H, f, c, lo, hi, q["U_guess"].T.reshape(-1), m, PCAP=PCAP, fail_thresh=ft)
            err = np.abs(u - q["U"].T.reshape(-1)).max()
            worst = max(worst, err)
            tot[why] = tot.get(why, 0) + 1
            S += sw; R += rt; P += pd
            if PCAP == 40 and ft == 1 and sw > 6:
                print("   step %2d iter %d: sweeps %3d ratios %3d pdas-its %3d %s err %.1e" % (q["step"], q["n_iter"], sw, rt, pd, why, err))
        n = len(ap.CAPTURE)
        print("PCAP=%d thresh=%d solves %d %s sweeps/solve %.2f ratios/solve %.2f pdas-its/solve %.2f worst err %.1e" % (PCAP, ft, n, tot, S / n, R / n, P / n, worst))
