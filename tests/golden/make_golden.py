"""Generate tests/golden/*.npz from the REFERENCE's own files.

Run once in the build container (where /root/reference exists):
    python tests/golden/make_golden.py
(`python tests/golden/make_golden.py dmdc` regenerates only dmdc.npz.)
The reference modules are loaded by file path (the package __init__ needs qutip/cvxpy, which are
absent); only linearize.py, lqr.py, model.py and vectorize.py are executed.  Two harness-side
shims restore NumPy-1 names the reference uses (np.product, np.math); an inert module named
``qutip`` satisfies vectorize.py's import (only vectorize_me touches it, and is not called).
Nothing from the reference is copied: the fixtures hold inputs and the reference's outputs.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np

REF = "/root/reference/mpc4quantum/"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    np.product = np.prod
    np.math = math
    pkg = types.ModuleType("m4q_reference")
    pkg.__path__ = [REF]
    sys.modules["m4q_reference"] = pkg
    sys.modules.setdefault("qutip", types.ModuleType("qutip"))
    mods = {}
    for name in ("linearize", "lqr", "model", "vectorize"):
        spec = importlib.util.spec_from_file_location("m4q_reference." + name, REF + name + ".py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules["m4q_reference." + name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


# ---- physical systems of the reference's tests, as plain ndarrays (tests/util_qubits.py) ----
SX = np.array([[0, 1], [1, 0]], dtype=complex)
SY = np.array([[0, -1j], [1j, 0]], dtype=complex)
SZ = np.array([[1, 0], [0, -1]], dtype=complex)
I2 = np.identity(2, dtype=complex)


def liou(H):
    d = H.shape[0]
    return -1j * (np.kron(H, np.identity(d)) - np.kron(np.identity(d), H.T))


def systems():
    out = {}
    wq = 2 * np.pi * 4
    out["qubit"] = (1.0, [0.5 * (wq - wq) * SZ + 0.05 * SZ, 0.5 * SX])          # util_qubits.py:77-79 (+detuning)
    dt = 0.25
    alpha = -2 * np.pi * 0.1 / dt
    a = np.diag(np.sqrt(np.arange(1, 3)), 1).astype(complex)
    H0 = alpha * np.diag([0, 0, 1]).astype(complex)
    out["transmon"] = (dt, [H0, 0.5 * (a.conj().T + a), 0.5j * (a.conj().T - a)])  # util_qubits.py:104-107
    out["coupled"] = (dt, [np.kron(SZ, SZ), np.kron(SY, I2), np.kron(I2, SY), np.kron(SZ, I2)])  # :26-34
    return out


def rand_density(rng, d):
    M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = M @ M.conj().T
    return rho / np.trace(rho).real


def main():
    ref = load_reference()
    lin, lqr, mdl, vec = ref["linearize"], ref["lqr"], ref["model"], ref["vectorize"]
    rng = np.random.default_rng(20211013)

    # (1) library tables
    tab = {}
    for order in (1, 2, 3):
        for m in (1, 2, 3):
            key = "o%d_m%d" % (order, m)
            tab[key + "_powers"] = np.vstack(lin.create_power_list(order, m))
            tab[key + "_size"] = np.array(lin.size_of_library(order, m))
            _, coefs = lin.diff_library(order, m)
            tab[key + "_dcoef"] = np.stack([c.reshape(-1) for c in coefs])
            u = rng.standard_normal((m, 3))
            tab[key + "_u"] = u
            tab[key + "_lib"] = np.vstack([f(u) for f in lin.create_library(order, m)])
            dfns, _ = lin.diff_library(order, m)
            tab[key + "_dlib"] = np.stack([np.vstack([f(u) for f in fl]) for fl in dfns])
    a = rng.standard_normal((3, 4))
    b = rng.standard_normal((5, 4)) + 1j * rng.standard_normal((5, 4))
    tab["kr_a"], tab["kr_b"], tab["kr_out"] = a, b, lin.krtimes(a, b)
    np.savez(os.path.join(OUT, "library_tables.npz"), **tab)

    # (2) discretize_homogeneous
    disc = {}
    for name, (dt, Hs) in systems().items():
        A_cts = [liou(H) for H in Hs]
        disc[name + "_dt"] = np.array(dt)
        disc[name + "_A_cts"] = np.stack(A_cts)
        for order in (1, 2):
            if name == "coupled" and order == 2:
                continue
            disc["%s_o%d" % (name, order)] = vec.discretize_homogeneous(A_cts, dt, order)
    np.savez(os.path.join(OUT, "discretize.npz"), **disc)

    # (3) linearisation along seeded Hermitian trajectories, and (4) lqr.quad_program
    lz = {}
    qp = {}
    for name, (dt, Hs) in systems().items():
        d = Hs[0].shape[0]
        n, m = d * d, len(Hs) - 1
        A_cts = [liou(H) for H in Hs]
        for order in (1, 2):
            if name == "coupled" and order == 2:
                continue
            key = "%s_o%d" % (name, order)
            A_dst = vec.discretize_homogeneous(A_cts, dt, order)
            P = lin.size_of_library(order, m) - 1
            model = mdl.DMDc(n, n, n * P, A_dst)
            A_x, A_u = model.get_discrete()
            wm = lin.WrapModel(A_x, A_u, m, order)
            T = 6
            xs = np.stack([rand_density(rng, d).reshape(-1) for _ in range(T + 1)], axis=1)
            us = 0.4 * rng.standard_normal((m, T))
            A_ls, B_ls, D_ls = wm.get_model_along_traj(xs, us, np.arange(T) * dt)
            lz[key + "_model"] = A_dst
            lz[key + "_xs"], lz[key + "_us"] = xs, us
            lz[key + "_A"], lz[key + "_B"], lz[key + "_D"] = np.stack(A_ls), np.stack(B_ls), np.stack(D_ls)
            lz[key + "_liftu"] = wm.lift_u(us)
            lz[key + "_f"] = np.hstack([wm.f(xs[:, i], us[:, i], 0) for i in range(T)])
            ux = lin.krtimes(wm.lift_u(us[:, :1]), xs[:, :1])
            lz[key + "_predict"] = model.predict(xs[:, :1], ux)

            # lqr.quad_program with Delta = 0 semantics (it takes none), constant targets
            target = np.zeros(n, dtype=complex)
            target[d + 1] = 1.0                                  # |1><1|
            X_bm = np.tile(target.reshape(-1, 1), (1, T + 1))
            U_bm = np.zeros((m, T))
            Qm = np.diag((np.arange(n) % (d + 1) == 0).astype(float))
            Q_ls = [Qm] * T + [2.0 * Qm]
            for tag, sat, rval in (("free", 50.0, 1e-1), ("sat", 0.15, 1e-3)):
                R_ls = [rval * np.identity(m)] * T
                x0 = xs[:, 0]
                X, U, cost, gains = lqr.quad_program(x0, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, None, sat, None)
                k2 = key + "_" + tag
                qp[k2 + "_sat"], qp[k2 + "_r"] = np.array(sat), np.array(rval)
                qp[k2 + "_X"], qp[k2 + "_U"], qp[k2 + "_cost"] = X, U, np.array(cost)
                qp[k2 + "_gains"] = np.stack(gains)
            qp[key + "_x0"], qp[key + "_X_bm"], qp[key + "_U_bm"] = xs[:, 0], X_bm, U_bm
            qp[key + "_Q"], qp[key + "_Qf"] = Qm, 2.0 * Qm
            qp[key + "_A"], qp[key + "_B"] = np.stack(A_ls), np.stack(B_ls)
    np.savez(os.path.join(OUT, "linearize.npz"), **lz)
    np.savez(os.path.join(OUT, "lqr.npz"), **qp)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


def golden_dmdc():
    """(5) the streaming model refits of model.py:109-313 (DiscrepDMDc, OnlineDMDc) on seeded complex snapshot data:
    batch fits, then the model after each of a run of fit_iteration updates."""
    mdl = load_reference()["model"]
    rng = np.random.default_rng(20211014)
    n, k, N, steps = 4, 4, 24, 8

    def cplx(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    A_true = 0.4 * cplx(n, n + k)
    X, U = cplx(n, N), cplx(k, N)
    Y = A_true @ np.vstack([X, U]) + 1e-3 * cplx(n, N)
    Xs, Us = cplx(n, steps), cplx(k, steps)
    Ys = A_true @ np.vstack([Xs, Us]) + 1e-3 * cplx(n, steps)
    A_boot = A_true + 0.05 * cplx(n, n + k)
    out = dict(X=X, U=U, Y=Y, Xs=Xs, Us=Us, Ys=Ys, A_boot=A_boot)

    d = mdl.DiscrepDMDc.from_data(Y, X, U, rcond=1e-12)
    out["discrep_from_data_A"] = d.A
    d = mdl.DiscrepDMDc.from_bootstrap(n, n, k, A_boot.copy())
    d.discount = 0.9
    d.append(Y[:, :6], X[:, :6], U[:, :6])
    seq = []
    for i in range(steps):
        A_x, A_u = d.fit_iteration(Ys[:, i], Xs[:, i], Us[:, i])
        seq.append(np.hstack([A_x, A_u]))
    out["discrep_stream_A"] = np.stack(seq)
    out["discrep_stream_Y"] = d.Y

    o = mdl.OnlineDMDc.from_data(Y, X, U)
    out["online_from_data_A"], out["online_from_data_P"] = o.A, o.P
    o = mdl.OnlineDMDc.from_bootstrap(n, n, k, A_boot.copy(), alpha=1e2)
    o.discount = 0.95
    seqA, seqP = [], []
    for i in range(steps):
        o.fit_iteration(Ys[:, i], Xs[:, i], Us[:, i])
        seqA.append(o.A.copy())
        seqP.append(o.P.copy())
    out["online_stream_A"], out["online_stream_P"] = np.stack(seqA), np.stack(seqP)
    out["predict"] = o.predict(Xs, Us)
    np.savez(os.path.join(OUT, "dmdc.npz"), **out)
    print("wrote dmdc.npz")


if __name__ == "__main__":
    if sys.argv[1:] == ["dmdc"]:
        golden_dmdc()
    else:
        main()
        golden_dmdc()
