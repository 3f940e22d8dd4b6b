"""CPU prototype (dense, condensed) of the device's exact box-QP iteration (csrc/m4q_mpc.h box_qp_iterate), to study its
iteration counts on the QPs of a closed loop.  Development tool; not used by the package, the tests or the bench."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc4quantum_amd import configs      # noqa: E402
from oracle import m4q_oracle as orc     # noqa: E402


def condense(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls):
    """J(u) = 1/2 u^T H u + f^T u + c over u = vec(U) (t-major), real."""
    m, T = U_bm.shape
    n = X_bm.shape[0]
    free = [np.asarray(x_init, dtype=complex).reshape(-1)]
    for t in range(T):
        free.append(A_ls[t] @ free[-1] + np.reshape(D_ls[t], -1))
    Phi = np.zeros((T + 1, n, T * m), dtype=complex)
    for s_ in range(T):
        for k in range(m):
            v = np.asarray(B_ls[s_], dtype=complex)[:, k]
            Phi[s_ + 1, :, s_ * m + k] = v
            for t in range(s_ + 1, T):
                v = A_ls[t] @ v
                Phi[t + 1, :, s_ * m + k] = v
    H = np.zeros((T * m, T * m))
    f = np.zeros(T * m)
    c = 0.0
    for t in range(T + 1):
        e0 = free[t] - X_bm[:, t]
        H += 2 * np.real(Phi[t].conj().T @ Q_ls[t] @ Phi[t])
        f += 2 * np.real(Phi[t].conj().T @ Q_ls[t] @ e0)
        c += np.real(e0.conj() @ Q_ls[t] @ e0)
    for t in range(T):
        sl = slice(t * m, (t + 1) * m)
        H[sl, sl] += 2 * np.real(R_ls[t])
        f[sl] += -2 * np.real(R_ls[t]) @ U_bm[:, t]
        c += U_bm[:, t] @ np.real(R_ls[t]) @ U_bm[:, t]
    return H, f, c


def face_min(H, f, u, pinned):
    fr = ~pinned
    un = u.copy()
    if fr.any():
        un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pinned)] @ u[pinned]))
    return un


def closed_loop_clipped(H, f, pinned, pin_val, lo, hi, m):
    """Stage by stage: the stage's free controls from the minimiser over everything not yet decided (pinned ones held),
    clipped; = the Riccati policy rollout with clipping."""
    N = len(f)
    u = np.where(pinned, pin_val, 0.0)
    decided = pinned.copy()
    clipped = False
    for t0 in range(0, N, m):
        rest = ~decided
        if not rest.any():
            break
        sol = np.linalg.solve(H[np.ix_(rest, rest)], -(f[rest] + H[np.ix_(rest, decided)] @ u[decided]))
        idx = np.flatnonzero(rest)
        for i, v in zip(idx, sol):
            if t0 <= i < t0 + m:
                if v > hi[i] or v < lo[i]:
                    clipped = True
                u[i] = min(max(v, lo[i]), hi[i])
                decided[i] = True
    return u, clipped


def solve(H, f, c, lo, hi, u0, m, log=None, variant="device"):
    J = lambda u: 0.5 * u @ H @ u + f @ u + c
    u = np.clip(u0, lo, hi)
    Jk = J(u)
    eps = 1e-12 * np.max(np.abs(hi))
    need_adj, fmin = True, False
    pinned = np.zeros(len(u), dtype=bool)
    sweeps = ratios = 0
    for it in range(200):
        if need_adj:
            g = H @ u + f
            new = ((u <= lo + eps) & (g > 0)) | ((u >= hi - eps) & (g < 0))
            if fmin and np.array_equal(new, pinned):
                return u, sweeps, ratios, "kkt"
            pinned = new
        need_adj, fmin = False, False
        pin_val = np.where(u >= hi - eps, hi, lo)
        uc, clipped = closed_loop_clipped(H, f, pinned, np.where(pinned, pin_val, 0.0), lo, hi, m)
        sweeps += 1
        Jc = J(uc)
        if log is not None:
            log.append((it, int(pinned.sum()), clipped, Jk, Jc))
        if np.abs(uc - u).max() <= 1e-13 * np.max(hi):
            fmin, need_adj = True, True
            continue
        if Jc < Jk or (not clipped and Jc <= Jk + 1e-12 * abs(Jk)):
            u, Jk, fmin, need_adj = uc, Jc, not clipped, True
            continue
        if not clipped:
            return u, sweeps, ratios, "precision"
        un = face_min(H, f, u, pinned)
        d = un - u
        with np.errstate(divide="ignore", invalid="ignore"):
            a_hi = np.where(un > hi, (hi - u) / d, np.inf)
            a_lo = np.where(un < lo, (lo - u) / d, np.inf)
        a = np.minimum(a_hi, a_lo)
        i = int(np.argmin(a))
        al = min(1.0, a[i])
        ratios += 1
        u = u + al * d
        u[i] = hi[i] if un[i] > hi[i] else lo[i]
        pinned = pinned.copy()
        pinned[i] = True
        Jk = J(u)
    return u, sweeps, ratios, "cap"


CAPTURE = []


def capture_loop(cfg, order, members, horizon=None):
    real = orc.exact_quad_program

    def wrap(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls, u_prev=None, sat=None, du=None):
        fr = sys._getframe(1).f_locals
        CAPTURE.append(dict(args=(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls), u_prev=u_prev, sat=sat, du=du,
                            U_guess=fr["U_guess"].copy(), step=fr["step"], n_iter=fr["n_iter"]))
        out = real(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls, u_prev, sat, du)
        CAPTURE[-1]["U"] = out[1]
        return out
    orc.exact_quad_program = wrap
    try:
        p = configs.build(cfg, batch=members, order=order, horizon=horizon)
        idx = np.arange(members)
        models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
        orc.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"], p["horizon"], p["n_steps"],
                      p["plant_op0"], list(p["plant_ops"][0]), p["Q"], p["R"], p["Qf"], p["sat"], p["du"], qp_mode="exact")
    finally:
        orc.exact_quad_program = real


if __name__ == "__main__":
    cfg, order, members = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    capture_loop(cfg, order, members)
    tot = {"kkt": 0, "precision": 0, "cap": 0}
    sw = rt = 0
    for q in CAPTURE:
        x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls = q["args"]
        m, T = U_bm.shape
        H, f, c = condense(x_init, np.asarray(X_bm, dtype=complex), np.asarray(U_bm, dtype=float), Q_ls, R_ls, A_ls, B_ls, D_ls)
        lo = -q["sat"] * np.ones(T * m)
        hi = q["sat"] * np.ones(T * m)
        if q["du"] is not None and q["u_prev"] is not None:
            up = np.reshape(q["u_prev"], -1).real
            lo[:m] = np.maximum(lo[:m], up - q["du"])
            hi[:m] = np.minimum(hi[:m], up + q["du"])
        log = []
        u, sweeps, ratios, why = solve(H, f, c, lo, hi, q["U_guess"].T.reshape(-1), m, log)
        err = np.abs(u - q["U"].T.reshape(-1)).max()
        tot[why] += 1
        sw += sweeps
        rt += ratios
        nact = int(((u <= lo + 1e-12) | (u >= hi - 1e-12)).sum())
        print("step %2d iter %d: sweeps %3d ratios %3d end %-9s active %2d/%d  |u-bvls| %.1e" % (q["step"], q["n_iter"], sweeps, ratios,
                                                                                               why, nact, len(u), err))
        if why == "cap" and "-v" in sys.argv:
            for row in log[:40]:
                print("      it %3d pinned %2d clipped %s Jk %.12e Jc %.12e" % row)
    print("solves %d: %s  sweeps/solve %.2f ratios/solve %.2f" % (len(CAPTURE), tot, sw / len(CAPTURE), rt / len(CAPTURE)))
