"""CPU prototype: gradient-projection phases between the face solves of the exact box-QP iteration (More-Toraldo's GPCG with the
face minimiser from one Riccati sweep instead of CG), on the QPs captured from config 4's closed loop (tests/probes/arc_proto.py
writes the capture).  Counts face solves (= pinned Riccati sweeps on the device, O(n^3) per index) and cheap passes (gradient or
objective evaluations = one rollout or adjoint pass, O(n^2) per index).
    python tests/probes/gpcg_proto.py [capture.pkl]
Development tool; not used by the package, the tests or the bench."""
import pickle
import sys

import numpy as np


def solve_gpcg(H, f, c, lo, hi, u0, k1=8, betas=(1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125), mu=1e-4, binding_only=True):
    J = lambda u: 0.5 * u @ H @ u + f @ u + c
    u = np.clip(u0, lo, hi)
    Jk = J(u)
    eps = 1e-12 * np.max(np.abs(hi))
    sweeps = cheap = 0
    for it in range(400):
        # ---- phase 1: projected gradient steps while the active set keeps changing
        act_prev = None
        for _ in range(k1):
            g = H @ u + f
            cheap += 1
            free = ~(((u <= lo + eps) & (g > 0)) | ((u >= hi - eps) & (g < 0)))
            gf = np.where(free, g, 0.0)
            if not gf.any():
                break
            a0 = (gf @ gf) / (gf @ H @ gf)                # Cauchy step on the free face
            took = False
            for s in (1.0, 0.25, 0.0625, 0.015625):
                ut = np.clip(u - s * a0 * g, lo, hi)
                Jt = J(ut)
                cheap += 1
                if Jt <= Jk + mu * (g @ (ut - u)):
                    took = True
                    break
            if not took:
                break
            act = (ut <= lo + eps) | (ut >= hi - eps)
            dec = Jk - Jt
            u, Jk = ut, Jt
            if act_prev is not None and np.array_equal(act, act_prev):
                break
            act_prev = act
        # ---- phase 2: face minimiser and projected search
        g = H @ u + f
        cheap += 1
        atlo, athi = u <= lo + eps, u >= hi - eps
        if binding_only:
            pin_lo, pin_hi = atlo & (g > 0), athi & (g < 0)
        else:
            pin_lo, pin_hi = atlo, athi
        pinned = pin_lo | pin_hi
        un = np.where(pin_hi, hi, np.where(pin_lo, lo, 0.0))
        fr = ~pinned
        if fr.any():
            un[fr] = np.linalg.solve(H[np.ix_(fr, fr)], -(f[fr] + H[np.ix_(fr, pinned)] @ un[pinned]))
        sweeps += 1
        inside = ((un <= hi + eps) & (un >= lo - eps)).all()
        if inside:
            mu_ = H @ un + f
            cheap += 1
            ok = not ((pin_hi & ~(mu_ < 0)) | (pin_lo & ~(mu_ > 0))).any()
            Jn = J(un)
            if ok:
                return un, sweeps, cheap, "kkt"
            if np.abs(un - u).max() <= 1e-13 * np.max(hi):
                # face minimiser, wrong multipliers: the next gradient phase releases them
                u, Jk = un, Jn
                continue
            u, Jk = un, Jn
            continue
        d = un - u
        took = False
        for b in betas:
            ut = np.clip(u + b * d, lo, hi)
            Jt = J(ut)
            cheap += 1
            if Jt < Jk - 1e-14 * abs(Jk):
                u, Jk, took = ut, Jt, True
                break
        if not took:
            # ratio step to the first bound
            with np.errstate(divide="ignore", invalid="ignore"):
                a_hi = np.where(un > hi, (hi - u) / d, np.inf)
                a_lo = np.where(un < lo, (lo - u) / d, np.inf)
            al = min(1.0, np.minimum(a_hi, a_lo).min())
            u = np.clip(u + al * d, lo, hi)
            Jk = J(u)
            cheap += 1
    return u, sweeps, cheap, "cap"


if __name__ == "__main__":
    cap = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/arc_capture_4_1_2.pkl", "rb"))
    variants = {"gpcg k1=8": dict(), "gpcg k1=3": dict(k1=3), "gpcg k1=20": dict(k1=20), "gpcg k1=8 all-active": dict(binding_only=False),
                "k1=0 (newton + arc only)": dict(k1=0)}
    for name, kw in variants.items():
        tot = {}; S = C = 0; worst = 0.0; hard = []
        for q in cap:
            m, T = q["shape"]
            H, f, c = q["H"], q["f"], q["c"]
            lo = -q["sat"] * np.ones(T * m); hi = q["sat"] * np.ones(T * m)
            if q["du"] is not None and q["u_prev"] is not None:
                up = np.reshape(q["u_prev"], -1).real
                lo[:m] = np.maximum(lo[:m], up - q["du"]); hi[:m] = np.minimum(hi[:m], up + q["du"])
            u, sw, ch, why = solve_gpcg(H, f, c, lo, hi, q["U_guess"].T.reshape(-1), **kw)
            err = np.abs(u - q["U"].T.reshape(-1)).max()
            worst = max(worst, err)
            tot[why] = tot.get(why, 0) + 1
            S += sw; C += ch
            if sw > 12:
                hard.append((q["step"], sw, ch))
        n = len(cap)
        print("%-28s solves %d %s face solves/solve %.2f cheap passes/solve %.1f worst err %.1e   hard: %s" % (
            name, n, tot, S / n, C / n, worst, hard[:8]))
