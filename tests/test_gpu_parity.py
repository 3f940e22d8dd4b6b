"""GPU parity tests (-m gpu): every HIP entry point, called through the C ABI, against
(a) outputs of the reference's own files (tests/golden/*.npz) and (b) the CPU oracle on the same
seeded inputs.  fp64 tolerances: 1e-10 relative for one QP solve / one MPC step,
1e-8 after a 20-step closed loop."""
import json
import os

import numpy as np
import pytest

import mpc4quantum_amd as m4q
from mpc4quantum_amd import _lib, configs
from mpc4quantum_amd import lqr as m4q_lqr
from oracle import m4q_oracle as orc

pytestmark = pytest.mark.gpu

SYSTEMS = [("qubit", 1), ("qubit", 2), ("transmon", 1), ("transmon", 2), ("coupled", 1)]
DIMS = {"qubit": (2, 1), "transmon": (3, 2), "coupled": (4, 3)}


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


# ---- the audit trail of the sensitivity clause ("make the parity claim auditable again") ------------------------------
# A teacher-forced MPC step that misses the fixed bounds (1e-10 on us[k], xs[k+1]; 1e-7 on the guesses left behind) may pass only
# against TEN TIMES what the oracle itself moves under a 1e-15 perturbation of the guess the step starts from - and every such
# admission is (i) recorded with its errors and the measured sensitivity, (ii) checked against the committed list
# profiles/r04_parity_admissions.json: a step that is not on that list FAILS the suite; configurations with no entry there admit
# nothing (the headline - config 3, order 1, T = 40 - and configs 1 and 2 on every arithmetic path among them).  Everything a run
# admitted is written to gpurun_out/parity_admissions_measured.json for comparison with the committed record.
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ADMISSIONS_FILE = os.path.join(ROOT, "profiles", "r04_parity_admissions.json")
_ADMISSIONS = []


def _expected_admissions(key):
    try:
        with open(_ADMISSIONS_FILE) as f:
            return set(json.load(f)["allowed"].get(key, []))
    except OSError:
        return set()


def _admit(key, step, errs, sens, tols):
    """Record one use of the sensitivity clause and check it against the committed list."""
    _ADMISSIONS.append({"case": key, "step": int(step), "errs": [float(e) for e in errs], "oracle_sensitivity": [float(v) for v in sens],
                        "fixed_bounds": list(tols)})
    if os.environ.get("M4Q_RECORD_ADMISSIONS"):         # the run that (re)generates the committed record: tools/parity_admissions.py
        return
    allowed = _expected_admissions(key)
    assert step in allowed, ("step %d of %s needs the sensitivity clause and is not on the committed list %s (errs %s, oracle "
                             "sensitivity %s)" % (step, key, sorted(allowed), errs, list(sens)))


_CASES_RUN = set()          # every teacher-forced case this run executed, with or without admissions


@pytest.fixture(scope="module", autouse=True)
def _dump_admissions():
    """A later, partial run (pytest -k ...) must not replace a fuller record with an emptier one: the cases THIS run executed
    replace their records in gpurun_out/parity_admissions_measured.json, every other case keeps what an earlier run measured."""
    yield
    out = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(out, "parity_admissions_measured.json")
    try:
        os.makedirs(out, exist_ok=True)
        kept, cases = [], set(_CASES_RUN)
        try:
            with open(dst) as f:
                old = json.load(f)
            kept = [r for r in old.get("admissions", []) if r["case"] not in _CASES_RUN]
            cases |= set(old.get("cases_run", []))
        except (OSError, ValueError):
            pass
        with open(dst, "w") as f:
            json.dump({"admissions": kept + _ADMISSIONS, "cases_run": sorted(cases)}, f, indent=1)
    except OSError:
        pass


def test_device_present():
    assert _lib.device_count() >= 1


# ---------------------------------------------------------------- linearisation
@pytest.mark.parametrize("name,order", SYSTEMS)
def test_linearize_vs_reference_golden(golden, name, order):
    g = golden("linearize")
    key = "%s_o%d" % (name, order)
    d, m = DIMS[name]
    n = d * d
    model = g[key + "_model"]
    dm = m4q.DMDc(n, n, model.shape[1] - n, model)
    wm = m4q.WrapModel(*dm.get_discrete(), m, order)
    xs, us = g[key + "_xs"], g[key + "_us"]
    A_ls, B_ls, D_ls = wm.get_model_along_traj(xs, us, np.arange(us.shape[1]))
    assert rel(np.stack(A_ls), g[key + "_A"]) <= 1e-13
    assert rel(np.stack(B_ls), g[key + "_B"]) <= 1e-13
    assert rel(np.stack(D_ls), g[key + "_D"]) <= 1e-13
    assert D_ls[0].shape == (n, 1)
    assert rel(wm.df_dx(xs[:, 2], us[:, 2], 0), g[key + "_A"][2]) <= 1e-13
    assert rel(wm.df_du(xs[:, 2], us[:, 2], 0), g[key + "_B"][2]) <= 1e-13


def test_linearize_batch_ragged_and_per_instance_models():
    # 7 trajectories (not a multiple of 4), per-instance models through the raw C entry point
    rng = np.random.default_rng(5)
    p = configs.build(3, batch=7, order=2)
    n, m, T = 9, 2, 5
    X = rng.standard_normal((7, T, n)) + 1j * rng.standard_normal((7, T, n))
    U = 0.3 * rng.standard_normal((7, T, m))
    A = np.empty((7, T, n, n), dtype=complex)
    Bm = np.empty((7, T, n, m), dtype=complex)
    D = np.empty((7, T, n), dtype=complex)
    L = _lib.lib()
    _lib.check(L.m4q_linearize_batch(7, n, m, 2, T, _lib.cbuf(p["models"])[1], 1, _lib.cbuf(X)[1], _lib.rbuf(U)[1],
                                     A.ctypes.data_as(_lib._dp), Bm.ctypes.data_as(_lib._dp), D.ctypes.data_as(_lib._dp)))
    for b in range(7):
        mod = p["models"][b]
        wm = orc.OracleWrapModel(mod[:, :n], mod[:, n:], m, 2)
        Ao, Bo, Do = wm.get_model_along_traj(X[b].T, U[b].T, np.arange(T))
        assert rel(A[b], np.stack(Ao)) <= 1e-13
        assert rel(Bm[b], np.stack(Bo)) <= 1e-13
        assert rel(D[b], np.stack(Do)[:, :, 0]) <= 1e-12


def test_unsupported_shape_raises():
    with pytest.raises(_lib.M4qError):
        m4q.WrapModel(np.eye(25), np.zeros((25, 25)), 1, 1).linearize_batch(np.zeros((1, 2, 25)), np.zeros((1, 2, 1)))
    with pytest.raises(ValueError):
        m4q.WrapModel(np.eye(4), np.zeros((4, 12)), 1, 1)


# ---------------------------------------------------------------- model construction
@pytest.mark.parametrize("d,m,order", [(2, 1, 1), (2, 1, 2), (3, 2, 1), (3, 2, 2), (4, 3, 1)])
def test_discretize_batch_vs_oracle(d, m, order):
    """vectorize.discretize_homogeneous on the device: shared operators, per-member operators, per-member scales."""
    rng = np.random.default_rng(40 + d)
    n = d * d
    Bn = 7
    Hs = [(lambda M: M + M.conj().T)(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) for _ in range(m + 1)]
    Ls = [m4q.liouvillian(H) + 0.1 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for H in Hs]
    dt = 0.3
    shared = m4q.discretize_homogeneous_batch(Ls, dt, order)
    assert shared.shape[0] == 1 and rel(shared[0], orc.discretize_homogeneous(Ls, dt, order)) <= 1e-13
    per = [np.stack([L * (1 + 0.1 * b) for b in range(Bn)]) for L in Ls]
    out = m4q.discretize_homogeneous_batch(per, dt, order)
    scales = 1 + 0.1 * rng.standard_normal((Bn, m + 1))
    out_s = m4q.discretize_homogeneous_batch(Ls, dt, order, scales=scales)
    for b in range(Bn):
        assert rel(out[b], orc.discretize_homogeneous([L[b] for L in per], dt, order)) <= 1e-13
        assert rel(out_s[b], orc.discretize_homogeneous([scales[b, k] * Ls[k] for k in range(m + 1)], dt, order)) <= 1e-13


def test_discretize_known_answer_on_device():
    """reference tests/test_mpc4quantum.py:147-188 shape of statement: order 1 gives [I + dt A | dt N_k]."""
    rng = np.random.default_rng(1)
    Ls = [rng.standard_normal((9, 9)) + 1j * rng.standard_normal((9, 9)) for _ in range(3)]
    out = m4q.discretize_homogeneous_batch(Ls, 1.0, 1)[0]
    assert np.allclose(out, np.hstack([Ls[0] + np.identity(9), Ls[1], Ls[2]]), rtol=0, atol=1e-14)


def test_session_build_models_matches_uploaded_models():
    """Models built on the device from generators and scales (no model crosses PCIe) give the same closed loop as
    host-built, uploaded models, on the real path (both take it) and on the complex path."""
    p = configs.build(3, batch=6, horizon=12, n_steps=6)
    ref = _gpu_batch(p, np.arange(6))
    for fc in (False, True):
        sess = _session(p, 6, force_complex=fc)
        try:
            sess.build_models(p["dt"], p["generators"], p["scales"])
            sess.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
            assert sess.path() == ("complex" if fc else "real")
            got = sess.download(_lib.F_MODELS, (6, 9, 27))
            assert rel(got, p["models"]) <= 1e-14
            sess.run()
            r = sess.results()
        finally:
            sess.close()
        assert np.array_equal(r["qp_solves"], ref["qp_solves"])
        assert rel(np.swapaxes(r["us"], 1, 2), ref["us"]) <= 1e-7 and rel(r["us"][:, 0], ref["us"][:, :, 0]) <= 1e-11


# ---------------------------------------------------------------- QP / Riccati
@pytest.mark.parametrize("name,order", SYSTEMS)
@pytest.mark.parametrize("tag", ["free", "sat"])
def test_lqr_mode_vs_reference_golden(golden, name, order, tag):
    """M4Q_QP_REF_LQR against outputs of the reference's lqr.quad_program."""
    g = golden("lqr")
    key = "%s_o%d" % (name, order)
    k2 = key + "_" + tag
    A_ls, B_ls = list(g[key + "_A"]), list(g[key + "_B"])
    T = len(A_ls)
    mdim = B_ls[0].shape[1]
    Q_ls = [g[key + "_Q"]] * T + [g[key + "_Qf"]]
    R_ls = [float(g[k2 + "_r"]) * np.identity(mdim)] * T
    X, U, cost, gains = m4q_lqr.quad_program(g[key + "_x0"], g[key + "_X_bm"], g[key + "_U_bm"], Q_ls, R_ls, A_ls, B_ls,
                                             None, float(g[k2 + "_sat"]), None)
    assert rel(np.stack(gains), g[k2 + "_gains"]) <= 1e-10
    assert rel(X, g[k2 + "_X"]) <= 1e-10
    assert rel(U, g[k2 + "_U"]) <= 1e-10
    assert abs(cost - float(g[k2 + "_cost"])) <= 1e-10 * max(1.0, abs(cost))


def _random_ltv(rng, n, m, T, Bn):
    A = np.eye(n) + 0.3 * (rng.standard_normal((Bn, T, n, n)) + 1j * rng.standard_normal((Bn, T, n, n))) / np.sqrt(n)
    Bm = 0.5 * (rng.standard_normal((Bn, T, n, m)) + 1j * rng.standard_normal((Bn, T, n, m)))
    D = 0.05 * (rng.standard_normal((Bn, T, n)) + 1j * rng.standard_normal((Bn, T, n)))
    x0 = rng.standard_normal((Bn, n)) + 1j * rng.standard_normal((Bn, n))
    Xb = 0.5 * (rng.standard_normal((Bn, T + 1, n)) + 1j * rng.standard_normal((Bn, T + 1, n)))
    Ub = 0.2 * rng.standard_normal((Bn, T, m))
    Qs = []
    for _ in range(T + 1):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        Qs.append(M @ M.conj().T / n)
    Rs = []
    for _ in range(T):
        M = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        Rs.append(M @ M.conj().T / m + 0.5 * np.eye(m))
    return A, Bm, D, x0, Xb, Ub, np.stack(Qs), np.stack(Rs)


@pytest.mark.parametrize("n,m", [(4, 1), (9, 2), (16, 3)])
def test_qp_mode_vs_oracle_general(n, m):
    """Dense complex A_t, time-varying Hermitian Q_t, R_t, ramped targets, Delta, per-instance benchmarks,
    ragged batch, both with the bounds inactive and active, with and without the du band."""
    rng = np.random.default_rng(100 + n)
    T, Bn = 7, 6
    A, Bm, D, x0, Xb, Ub, Qs, Rs = _random_ltv(rng, n, m, T, Bn)
    uprev = 0.1 * rng.standard_normal((Bn, m))
    for sat, du in ((1e3, None), (0.3, None), (0.6, 0.2)):
        X, U, cost, gains = m4q.quad_program_batch(x0, Xb, Ub, Qs, Rs, A, Bm, D, uprev if du else None, sat, du)
        for b in range(Bn):
            Xo, Uo, co, go = orc.quad_program(x0[b], Xb[b].T, Ub[b].T, list(Qs), list(Rs), list(A[b]), list(Bm[b]),
                                              list(D[b]), uprev[b] if du else None, sat, du)
            assert rel(gains[b], np.stack([gk.T for gk in go])) <= 1e-9
            assert rel(X[b].T, Xo) <= 1e-9
            assert rel(U[b].T, Uo) <= 1e-9
            assert abs(cost[b] - co) <= 1e-9 * max(1.0, abs(co))
        if du:
            assert np.all(np.abs(U[:, 0, :] - uprev) <= du + 1e-15)
        assert np.abs(U).max() <= sat + 1e-15


def test_qp_mode_vs_independent_kkt():
    """Unconstrained case against a dense KKT solve (real controls) of the QP stated at optimize.py:27-41,54.
    lqr.py takes the real part of a complex gain (lqr.py:75-76), which is the real-control optimum when the
    model preserves Hermiticity, as every vectorised-Liouvillian model does: use such a problem."""
    rng = np.random.default_rng(7)
    p = configs.build(3, batch=2, order=2)
    n, m, T = 9, 2, 6
    A, Bm, D, x0 = [], [], [], []
    for b in range(2):
        mod = p["models"][b]
        wm = orc.OracleWrapModel(mod[:, :n], mod[:, n:], m, 2)
        rhos = []
        for _ in range(T + 1):
            M = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
            r = M @ M.conj().T
            rhos.append((r / np.trace(r).real).reshape(-1))
        xg = np.stack(rhos, axis=1)
        ug = 0.4 * rng.standard_normal((m, T))
        Ao, Bo, Do = wm.get_model_along_traj(xg, ug, np.arange(T))
        A.append(np.stack(Ao)); Bm.append(np.stack(Bo)); D.append(np.stack(Do)[:, :, 0]); x0.append(xg[:, 0])
    A, Bm, D, x0 = np.stack(A), np.stack(Bm), np.stack(D), np.stack(x0)
    Xb = np.tile(p["X_targ"][:, :T + 1].T[None], (2, 1, 1))
    Ub = 0.05 * rng.standard_normal((2, T, m))
    Qs = np.stack([p["Q"]] * T + [3 * p["Q"]]).astype(complex)
    Rs = np.stack([0.05 * np.eye(m)] * T).astype(complex)
    X, U, cost, _ = m4q.quad_program_batch(x0, Xb, Ub, Qs, Rs, A, Bm, D, None, 1e6, None)
    for b in range(2):
        Xk, Uk = orc.kkt_quad_program(x0[b], Xb[b].T, Ub[b].T, list(Qs), list(Rs), list(A[b]), list(Bm[b]), list(D[b]))
        assert np.abs(Uk).max() > 1e-2
        assert rel(U[b].T, Uk) <= 1e-8
        assert rel(X[b].T, Xk) <= 1e-8


@pytest.mark.parametrize("r_scale", [1.0, 300.0])
def test_qp_mode_vs_independent_kkt_at_the_headline_horizon(r_scale):
    """The same pin at config 3's own T = 40 (order 1): Delta != 0, a target ramped over the window, a non-zero control target,
    bounds inactive - the device's affine Riccati solve against the dense KKT solve of optimize.py:27-41,54 (800 unknowns), and
    against the oracle's `qp` mode, which tests/test_oracle_golden.py holds to the same KKT solve on the CPU."""
    from tests.test_oracle_golden import _config3_ltv_with_delta
    rng = np.random.default_rng(17)
    T = 40
    for b in range(2):
        p, x0, Xb, Ub, A_ls, B_ls, D_ls = _config3_ltv_with_delta(b, T, rng)
        Qs = np.stack([p["Q"]] * T + [p["Qf"]]).astype(complex)
        Rs = np.stack([r_scale * p["R"]] * T).astype(complex)
        X, U, cost, _ = m4q.quad_program_batch(x0[None], Xb.T[None], Ub.T[None], Qs, Rs, np.stack(A_ls)[None], np.stack(B_ls)[None],
                                               np.stack(D_ls).reshape(1, T, -1), None, 1e6, None)
        Xk, Uk = orc.kkt_quad_program(x0, Xb, Ub, list(Qs), list(Rs), A_ls, B_ls, D_ls)
        Xo, Uo, _, _ = orc.quad_program(x0, Xb, Ub, list(Qs), list(Rs), A_ls, B_ls, D_ls, None, 1e6, None)
        assert np.abs(Uk).max() > 0.5
        assert rel(U[0].T, Uk) <= 1e-9 and rel(X[0].T, Xk) <= 1e-9
        assert rel(U[0].T, Uo) <= 1e-9 and rel(X[0].T, Xo) <= 1e-9


def _scenario_qp(config, order, T, Bn, seed, amp):
    """Linearisation of a reference scenario's model along a perturbed guess: a Hermiticity-preserving LTV problem."""
    rng = np.random.default_rng(seed)
    p = configs.build(config, batch=Bn, order=order, horizon=T)
    n, m = p["x0"].shape[1], p["U_targ"].shape[0]
    A, Bm, D, x0 = [], [], [], []
    for b in range(Bn):
        mod = p["models"][b if p["models"].shape[0] > 1 else 0]
        wm = orc.OracleWrapModel(mod[:, :n], mod[:, n:], m, order)
        xg = np.tile(p["x0"][b][:, None], (1, T + 1))
        ug = amp * p["sat"] * rng.uniform(-1, 1, (m, T))
        Ao, Bo, Do = wm.get_model_along_traj(xg, ug, np.arange(T))
        A.append(np.stack(Ao)); Bm.append(np.stack(Bo)); D.append(np.stack(Do).reshape(T, n)); x0.append(xg[:, 0])
    Qs = np.stack([p["Q"]] * T + [p["Qf"]]).astype(complex)
    Rs = np.stack([p["R"]] * T).astype(complex)
    return (np.stack(x0), p["X_targ"][:, :T + 1].T[None], p["U_targ"][:, :T].T[None].real, Qs, Rs, np.stack(A), np.stack(Bm),
            np.stack(D), p["sat"])


@pytest.mark.parametrize("config,order,T,sat_scale,du", [(1, 1, 10, 1.0, None), (1, 2, 25, 0.3, None), (3, 2, 40, 0.3, None),
                                                          (3, 2, 20, 0.3, 0.05), (3, 1, 12, 1e3, None), (4, 1, 20, 0.3, None)])
def test_exact_box_qp_vs_bvls_oracle(config, order, T, sat_scale, du):
    """M4Q_QP_EXACT_BOX: the box-constrained QP of optimize.py:27-54 solved to optimality on the device (projected Newton
    on the Riccati factorisation) against an independent solver of the same statement (scipy BVLS on the condensed
    problem) - tolerance 1e-9 of the bound on the controls, 1e-11 relative on the objective."""
    Bn = 5
    x0, Xb, Ub, Qs, Rs, A, Bm, D, sat = _scenario_qp(config, order, T, Bn, 40 + config, 0.3)
    sat = sat * sat_scale
    up = 0.02 * np.arange(Bn * Ub.shape[2]).reshape(Bn, -1) if du else None
    X, U, cost, _ = m4q.quad_program_batch(x0, Xb, Ub, Qs, Rs, A, Bm, D, up, sat, du, exact=True)
    Xc, Uc, costc, _ = m4q.quad_program_batch(x0, Xb, Ub, Qs, Rs, A, Bm, D, up, sat, du)
    assert np.abs(U).max() <= sat
    assert np.all(cost <= costc * (1 + 1e-14))                    # never worse than the clipped rollout it starts from
    active = 0
    for b in range(Bn):
        Xe, Ue, ce = orc.exact_quad_program(x0[b], Xb[0].T, Ub[0].T, list(Qs), list(Rs), list(A[b]), list(Bm[b]), list(D[b]),
                                            None if up is None else up[b], sat, du)
        assert np.abs(U[b].T - Ue).max() <= 1e-9 * min(sat, 1.0)
        assert rel(X[b].T, Xe) <= 1e-9
        assert abs(cost[b] - ce) <= 1e-11 * max(1.0, abs(ce))
        active += int((np.abs(Ue) >= sat * (1 - 1e-12)).sum())
        if du:
            assert np.all(np.abs(U[b, 0] - up[b]) <= du * (1 + 1e-14))
    if sat_scale < 100:
        assert active > 0 and np.abs(U - Uc).max() > 1e-4 * sat   # the bounds matter in these cases
    else:
        assert active == 0 and rel(U, Uc) <= 1e-12                # inactive: the clipped Riccati rollout is already optimal


def test_exact_box_qp_rejects_ref_lqr():
    rng = np.random.default_rng(1)
    A, Bm, D, x0, Xb, Ub, Qs, Rs = _random_ltv(rng, 4, 1, 3, 1)
    with pytest.raises(_lib.M4qError):
        m4q.quad_program_batch(x0, Xb, Ub, Qs, Rs, A, Bm, D, None, 1.0, None, flags=_lib.QP_REF_LQR, exact=True)


def test_quad_program_dropin_signature():
    """optimize.quad_program(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls, u_prev, sat, du, verbose)."""
    rng = np.random.default_rng(9)
    n, m, T = 9, 2, 4
    A, Bm, D, x0, Xb, Ub, Qs, Rs = _random_ltv(rng, n, m, T, 1)
    args = (x0[0], Xb[0].T, Ub[0].T, list(Qs), list(Rs), list(A[0]), list(Bm[0]), [d.reshape(-1, 1) for d in D[0]],
            np.zeros((m, 1)), 0.5, 0.25, False)
    X, U, obj, aux = m4q.quad_program(*args)
    Xo, Uo, co, go = orc.quad_program(*args)
    assert X.shape == (n, T + 1) and U.shape == (m, T) and aux[0].shape == (m, n + 1)
    assert rel(X, Xo) <= 1e-10 and rel(U, Uo) <= 1e-10 and abs(obj - co) <= 1e-10 * max(1, abs(co))
    with pytest.raises(TypeError):
        m4q.quad_program(*args[:9], None, None)


def test_nonfinite_inputs_give_nonfinite_cost():
    rng = np.random.default_rng(3)
    A, Bm, D, x0, Xb, Ub, Qs, Rs = _random_ltv(rng, 4, 1, 3, 1)
    A[0, 1, 0, 0] = np.nan
    _, _, cost, _ = m4q.quad_program_batch(x0, Xb, Ub, Qs, Rs, A, Bm, D, None, 1.0, None)
    assert not np.isfinite(cost[0])


# ---------------------------------------------------------------- plant
@pytest.mark.parametrize("cfg", [1, 3, 4])
def test_plant_hamiltonian_vs_expm(cfg):
    p = configs.build(cfg, batch=5)
    rng = np.random.default_rng(cfg)
    d, n, m = p["d"], p["dim_x"], p["dim_u"]
    rho = []
    for _ in range(5):
        M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        r = M @ M.conj().T
        rho.append((r / np.trace(r).real).reshape(-1))
    x = np.stack(rho)
    for scale in (1.0, 40.0):      # the large one forces scaling-and-squaring (norm > theta_13)
        u = scale * p["sat"] * rng.uniform(-1, 1, (5, m))
        out = m4q.plant_step_batch(x, u, p["plant_op0"], p["plant_ops"], p["dt"])
        for b in range(5):
            ref = orc.plant_step(x[b], u[b], p["plant_op0"][0], list(p["plant_ops"][0]), p["dt"])
            assert rel(out[b], ref) <= 1e-12
        tr = out.reshape(5, d, d).trace(axis1=1, axis2=2)
        assert np.abs(tr - 1).max() < 1e-12


@pytest.mark.parametrize("d,m", [(2, 1), (3, 2), (4, 3)])
def test_plant_generator_vs_expm_with_dissipation(d, m):
    """General generator: Hamiltonian part plus amplitude damping (not unitary), per-instance operators."""
    rng = np.random.default_rng(10 + d)
    n = d * d
    Bn = 5
    a = np.diag(np.sqrt(np.arange(1, d)), 1).astype(complex)

    def lindblad(c):
        cd = c.conj().T
        eye = np.eye(d)
        return np.kron(c, c.conj()) - 0.5 * (np.kron(cd @ c, eye) + np.kron(eye, (cd @ c).T))
    L0 = np.stack([m4q.liouvillian((lambda M: M + M.conj().T)(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))))
                   + 0.3 * lindblad(a) for _ in range(Bn)])
    Lk = np.stack([[m4q.liouvillian((lambda M: M + M.conj().T)(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))))
                    for _ in range(m)] for _ in range(Bn)])
    x = rng.standard_normal((Bn, n)) + 1j * rng.standard_normal((Bn, n))
    for dt in (0.1, 2.5):
        u = rng.uniform(-1, 1, (Bn, m))
        out = m4q.plant_step_batch(x, u, L0, Lk, dt, _lib.PLANT_GENERATOR)
        for b in range(Bn):
            ref = orc.plant_step_generator(x[b], u[b], L0[b], list(Lk[b]), dt)
            assert rel(out[b], ref) <= 1e-11


def _blackman_pulse(ts, t0, tf, dt):
    """The reference's test pulse (tests/util_qubits.py:9-17): a Blackman window sampled at dt, linearly interpolated."""
    M = int((tf - t0) / dt)
    return np.interp(ts, np.linspace(t0, tf, M), np.blackman(M), left=0, right=0)


@pytest.mark.parametrize("system", ["qubit", "transmon", "coupled"])
def test_vectorized_model_rollout_tracks_the_device_plant(system):
    """The reference's one model-vs-plant check (tests/test_mpc4quantum.py:215-274, test_vectorization) with the device on
    both sides: the bilinear model (`vectorize_me` on the |i><j| basis -> `discretize_homogeneous` on the GPU, order 2; order 1 at d = 4) rolled
    out under a Blackman pulse with `DMDc.predict` (mpc.py:267's call shape) against the plant kernel stepping the same held
    controls (`m4q_plant_step_batch`: Pade matrix exponential; the reference has qutip.mesolve there).  The reference's
    criterion: the plant state within 0.1 of the model's mid-step average on more than 90 % of the points - on its qubit
    (tests/util_qubits.py:60-91, wQ = wD = wR, dt = 0.5, 25 steps, unit amplitude), and on the transmon and coupled-qubit
    systems of configs 3 and 4 at their own dt.  Plus what the Taylor truncation promises: the one-step model error is
    O((dt |H|)^(order + 1)) and falls by ~2^(order + 1) when dt is halved."""
    from mpc4quantum_amd.configs import SX, SZ
    n_train = 25
    order = 1 if system == "coupled" else 2              # (the device builds the (16, 3) shape at order 1 only: m4q_shapes.inc)
    if system == "qubit":
        d, dt, amp = 2, 0.5, [1.0]
        H = [0.5 * (np.pi - np.pi) * SZ, 0.5 * SX]
    else:
        p = configs.build(3 if system == "transmon" else 4, batch=1)
        d, dt = p["d"], p["dt"]
        H = [p["plant_op0"][0]] + list(p["plant_ops"][0])
        amp = [0.8 * p["sat"], -0.5 * p["sat"], 0.3 * p["sat"]][:len(H) - 1]
    n, m = d * d, len(H) - 1
    basis = [np.outer(np.eye(d)[i], np.eye(d)[j]) for i in range(d) for j in range(d)]
    A_cts = [m4q.vectorize_me(h, basis) for h in H]
    assert all(np.abs(a - m4q.liouvillian(h)).max() <= 1e-14 for a, h in zip(A_cts, H))
    ts = np.arange(n_train) * dt
    us = np.stack([a * _blackman_pulse(ts, 0, n_train * dt, dt) for a in amp])             # (m, n_train)
    rho0 = np.zeros((d, d), dtype=complex)
    rho0[0, 0] = 1
    x0 = rho0.reshape(-1)

    def rollouts(dt_k, us_k):
        A = m4q.discretize_homogeneous_batch(A_cts, dt_k, order)[0]                         # device (discretize_kernel)
        assert rel(A, m4q.discretize_homogeneous(A_cts, dt_k, order)) <= 1e-13
        model = m4q.DMDc(n, n, A.shape[1] - n, A)
        lib = m4q.create_library(order, m)[1:]
        xs_lin, xs_me, one_step = [x0], [x0], []
        for i in range(us_k.shape[1]):
            lift_u = np.array([f(us_k[:, i:i + 1]) for f in lib]).reshape(-1, 1)
            xs_lin.append(model.predict(xs_lin[-1], m4q.krtimes(lift_u, xs_lin[-1].reshape(-1, 1))).reshape(-1))
            nxt = m4q.plant_step_batch(xs_me[-1][None], us_k[:, i][None], H[0], np.stack(H[1:]), dt_k)[0]
            one_step.append(np.abs(model.predict(xs_me[-1], m4q.krtimes(lift_u, xs_me[-1].reshape(-1, 1))).reshape(-1) - nxt).max())
            xs_me.append(nxt)
        return np.array(xs_lin).T, np.array(xs_me).T, max(one_step)

    xs_lin, xs_me, err1 = rollouts(dt, us)
    # like with like: model and plant at the same instants (the sample times of this plant are known)
    assert (np.abs(xs_me - xs_lin) < 0.1).mean() > 0.9, np.abs(xs_me - xs_lin).max()
    # the reference's own statement: the model's mid-step average against the plant's samples; it does not know which end of
    # the step mesolve reports ("Not sure where mesolve reports the value", :268) - one of the two alignments must hold
    mid = 0.5 * (xs_lin[:, 1:] + xs_lin[:, :-1])
    frac = max((np.abs(xs_me[:, :-1] - mid) < 0.1).mean(), (np.abs(xs_me[:, 1:] - mid) < 0.1).mean())
    assert frac > 0.9, frac
    # the plant is trace preserving and unitary; the order-2 model only approximately
    assert np.abs(np.trace(xs_me[:, -1].reshape(d, d)) - 1) <= 1e-12
    assert abs(np.vdot(xs_me[:, -1], xs_me[:, -1]).real - 1) <= 1e-12
    # truncation order: the same pulse on a grid twice as fine - the worst one-step error falls by about 2^(order + 1)
    ts2 = np.arange(2 * n_train) * dt / 2
    us2 = np.stack([a * _blackman_pulse(ts2, 0, n_train * dt, dt) for a in amp])
    _, _, err2 = rollouts(dt / 2, us2)
    assert err1 > 0 and 0.6 * 2 ** (order + 1) <= err1 / err2 <= 1.5 * 2 ** (order + 1), (err1, err2)


def test_qexperiment_simulate_shape_and_values():
    p = configs.build(3, batch=1)
    exp = m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    ts = np.array([0.0, 0.25, 0.5])
    us = np.array([[0.3, -0.2, 0.0], [0.1, 0.4, 0.0]])
    out = exp.simulate(p["x0"][0], ts, us)
    assert out.shape == (9, 3)
    x = p["x0"][0]
    for i in range(2):
        x = orc.plant_step(x, us[:, i], p["plant_op0"][0], list(p["plant_ops"][0]), 0.25)
        assert rel(out[:, i + 1], x) <= 1e-12


def test_qexperiment_collapse_and_expectation_operators():
    """QExperiment.set('c_ops', ...) / set('e_ops', ...) (experiment.py:196-210): the Lindblad plant over held-control
    intervals against scipy's expm of the generator; expectation values tr(E rho(t)) as mesolve's `expect`."""
    import scipy.linalg
    p = configs.build(3, batch=1)
    a = np.diag([1.0, np.sqrt(2.0)], 1).astype(complex)           # transmon lowering operator
    exp = m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    exp.set("c_ops", [0.2 * a, 0.1 * (a.conj().T @ a)])            # relaxation + dephasing
    ts = np.array([0.0, 0.25, 0.5, 0.75])
    us = np.array([[0.3, -0.2, 0.5, 0.0], [0.1, 0.4, -0.3, 0.0]])
    out = exp.simulate(p["x0"][0], ts, us)
    assert out.shape == (9, 4)
    L0, Lk = exp.operators()
    x = p["x0"][0]
    for i in range(3):
        x = scipy.linalg.expm(0.25 * (L0 + us[0, i] * Lk[0] + us[1, i] * Lk[1])) @ x
        assert rel(out[:, i + 1], x) <= 1e-12
    assert np.abs(out[0] + out[4] + out[8] - 1).max() <= 1e-12       # trace
    pops = [np.diag(np.eye(3)[k]).astype(complex) for k in range(3)]
    exp.set("e_ops", pops)
    ex = exp.simulate(p["x0"][0], ts, us)
    assert ex.shape == (3, 4) and np.abs(ex - out[[0, 4, 8]]).max() <= 1e-15


def test_mpc_with_collapse_operators_runs_fused_and_equals_the_generator_plant():
    """mpc() with a QExperiment carrying c_ops: the closed loop stays on the GPU (generator plant) and equals, bit for bit, the
    run with the same generators handed over as an LExperiment; a foreign-style host loop (one launch per step, simulate() on
    the host) agrees to rounding."""
    p = configs.build(1, batch=1)
    sm = np.array([[0, 1], [0, 0]], dtype=complex)
    qe = m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    qe.set("c_ops", [np.sqrt(0.02) * sm])
    L0, Lk = qe.operators()
    le = m4q.LExperiment(L0, list(Lk))
    outs = []
    for exp in (qe, le):
        clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
        model = m4q.DMDc(4, 4, 4, p["models"][0])
        (xs, us), _, code = m4q.mpc(p["x0"][0], 1, 1, p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"], p["Qf"],
                                    sat=p["sat"], du=p["du"], progress_bar=False)
        assert code == 0
        outs.append((xs, us))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert abs(outs[0][0][0, -1] + outs[0][0][3, -1] - 1) < 1e-10           # trace preserved, purity not
    assert abs(np.vdot(outs[0][0][:, -1], outs[0][0][:, -1]).real - 1) > 1e-4


# ---------------------------------------------------------------- closed loop
def _oracle_batch(p, idx, **kw):
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    return orc.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"], p["horizon"],
                         p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"], p["R"], p["Qf"], p["sat"], p["du"],
                         **kw)


def _gpu_batch(p, idx, **kw):
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    return m4q.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"],
                         p["plant_ops"], p["Q"], p["R"], p["Qf"], p["sat"], p["du"], **kw)


def _session(p, B, **kw):
    models = p["models"]
    return m4q.EnsembleSession(B, p["dim_x"], p["dim_u"], p["order"], p["horizon"], p["n_steps"], p["dt"], p["sat"], p["du"],
                               model_per_instance=models.shape[0] > 1, target_cols=p["n_steps"] + p["horizon"] + 1, **kw)


def _envelope(p, idx, xs, us, eps=1e-14, **kw):
    """How far the ORACLE itself moves when its inputs are perturbed in the last bits - x0 scaled by 1 +- eps, the model by
    1 +- 1e-15 (what a change of basis or of summation order does to it): the conditioning of the 20-step closed loop.
    Controls saturate and the Riccati gains are stiff (R ~ 1e-3/sat^2), so one ulp grows by many orders of magnitude over a
    run (the reference's own run at T = 40 is chaotic from MPC step 7 on: tests/golden/mpc_loop_long.npz, env_us); a
    free-running comparison can only be asked to stay within that envelope.  Step-by-step parity is checked separately
    with teacher forcing."""
    eu = np.zeros(us.shape[2])
    ex = np.zeros(xs.shape[2])
    variants = [("x0", 1 + eps), ("x0", 1 - eps), ("x0", 1 + 7 * eps), ("x0", 1 - 5 * eps), ("models", 1 + 1e-15), ("models", 1 - 1e-15),
                ("models", 1 + 3e-15)]
    for key, scale in variants:
        q = dict(p)
        q[key] = p[key] * scale
        xs2, us2, _, _ = _oracle_batch(q, idx, **kw)
        eu = np.maximum(eu, np.maximum.accumulate(np.abs(us2 - us).max(axis=(0, 1))))
        ex = np.maximum(ex, np.maximum.accumulate(np.abs(xs2 - xs).max(axis=(0, 1))))
    return eu, ex


@pytest.mark.parametrize("path", ["real", "complex"])
@pytest.mark.parametrize("cfg,order,batch,horizon", [(1, 1, 1, None), (1, 2, 1, None), (2, 1, 6, None), (3, 1, 5, 16),
                                                      (3, 2, 3, 12), (4, 1, 3, 10)])
def test_closed_loop_vs_oracle(cfg, order, batch, horizon, path):
    """Free-running closed loop, all MPC steps in one launch, on both arithmetic paths: the Hermitian-basis real
    path these (Liouvillian) models qualify for, and the general complex path forced with M4Q_OPT_FORCE_COMPLEX."""
    p = configs.build(cfg, batch=batch, order=order, horizon=horizon)
    idx = np.arange(batch)
    res = _gpu_batch(p, idx, force_complex=(path == "complex"))
    assert res["path"] == path
    xs, us, codes, solves = _oracle_batch(p, idx)
    assert np.array_equal(res["exit_codes"], codes)
    assert np.array_equal(res["qp_solves"], solves)
    assert np.all(res["steps_done"] == p["n_steps"])
    assert rel(res["us"][:, :, 0], us[:, :, 0]) <= 1e-10           # one MPC step (incl. all its SQP iterations)
    assert rel(res["xs"][:, :, 1], xs[:, :, 1]) <= 1e-10
    eu, ex = _envelope(p, idx, xs, us)
    du_k = np.abs(res["us"] - us).max(axis=(0, 1))
    dx_k = np.abs(res["xs"] - xs).max(axis=(0, 1))
    assert np.all(du_k <= 1e-9 + 100 * eu), (du_k, eu)
    assert np.all(dx_k[1:] <= 1e-9 + 100 * ex[1:]), (dx_k, ex)
    assert np.abs(res["us"]).max() <= p["sat"] * (1 + 1e-15)


@pytest.mark.parametrize("path", ["real", "complex"])
@pytest.mark.parametrize("cfg,order,batch,horizon", [(1, 1, 2, None), (3, 1, 3, 16), (3, 2, 2, None), (4, 1, 2, 12),
                                                      (2, 1, 3, None), (3, 1, 2, None), (4, 1, 2, None)])
def test_closed_loop_exact_qp_vs_oracle(cfg, order, batch, horizon, path):
    """M4Q_QP_EXACT_BOX in the closed loop: every QP solved to the box-constrained optimum on the device (active-set
    iteration on the Riccati factorisation) against the oracle's loop with the BVLS solve of the same QP.  Same SQP
    iteration counts and exit codes; the first MPC step (all its SQP iterations) to 1e-9, the free-running 20-step
    trajectory to 1e-4 of the bound (measured 1e-6 .. 1e-10: with exact solves the loop is far less sensitive than with
    clipped ones).  horizon None = the BASELINE configuration's own horizon: config 2 (T = 20), config 3 at orders 1 and 2
    (T = 40) and config 4 (T = 40) run at full size."""
    p = configs.build(cfg, batch=batch, order=order, horizon=horizon)
    idx = np.arange(batch)
    res = _gpu_batch(p, idx, force_complex=(path == "complex"), exact_qp=True)
    assert res["path"] == path
    xs, us, codes, solves = _oracle_batch(p, idx, qp_mode="exact")
    assert np.array_equal(res["exit_codes"], codes) and np.all(codes == 0)
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, 0], us[:, :, 0]) <= 1e-9
    assert rel(res["xs"][:, :, 1], xs[:, :, 1]) <= 1e-9
    if horizon is None and cfg in (3, 4) and order == 1:
        # order-1 models at T = 40 (mode growth 1.18 per step): the free-running loop is held to the ORACLE's own envelope
        # (what its trajectory moves by when x0 is perturbed by 1e-14), as the clipped loop is; step-by-step parity at these
        # sizes is test_closed_loop_exact_stepwise_teacher_forced
        eu, ex = _envelope(p, idx, xs, us, qp_mode="exact")
        assert np.all(np.abs(res["us"] - us).max(axis=(0, 1)) <= 1e-9 + 100 * eu), (np.abs(res["us"] - us).max(axis=(0, 1)), eu)
        assert np.all(np.abs(res["xs"] - xs).max(axis=(0, 1))[1:] <= 1e-9 + 100 * ex[1:])
    else:
        assert np.abs(res["us"] - us).max() <= 1e-4 * p["sat"]
        assert np.abs(res["xs"] - xs).max() <= 1e-4
    assert np.abs(res["us"]).max() <= p["sat"] * (1 + 1e-15)
    n_solves, sweeps, ratio_steps, end_kkt, end_precision, end_cap = res["qp_stats"]
    assert n_solves == solves.sum() and end_kkt + end_precision == n_solves and end_cap == 0
    assert sweeps >= n_solves
    # and it matters: the clipped-Riccati loop lands elsewhere
    clip = _gpu_batch(p, idx, force_complex=(path == "complex"))
    assert np.abs(clip["us"] - res["us"]).max() > 1e-3 * p["sat"]
    assert clip["qp_stats"] == (0, 0, 0, 0, 0, 0)


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("cfg,horizon", [(3, 5), (3, 6), (3, 7), (3, 13), (2, 3), (2, 9), (1, 7)])
def test_tile_sweeps_at_horizons_that_are_not_multiples_of_four(cfg, horizon, exact):
    """The tile sweeps work on blocks of four horizon indices (m4q_tile3.h) and take the (T - 1) % 4 + 1 top indices first: every
    residue of T, shorter than one block as well, on the clipped solve's backward sweep and on the exact solve's pinned sweep,
    against the oracle's loop (first MPC step to 1e-10 / 1e-9, the short free run to the usual bounds)."""
    p = configs.build(cfg, batch=3, order=1, horizon=horizon, n_steps=4)
    idx = np.arange(3)
    res = _gpu_batch(p, idx, exact_qp=exact)
    assert res["path"] == "real" and res["path_detail"] == "traceless-tile"
    xs, us, codes, solves = _oracle_batch(p, idx, **({"qp_mode": "exact"} if exact else {}))
    assert np.array_equal(res["exit_codes"], codes) and np.array_equal(res["qp_solves"], solves)
    tol = 1e-9 if exact else 1e-10
    assert rel(res["us"][:, :, 0], us[:, :, 0]) <= tol
    assert rel(res["xs"][:, :, 1], xs[:, :, 1]) <= tol
    eu, ex = _envelope(p, idx, xs, us, **({"qp_mode": "exact"} if exact else {}))
    assert np.all(np.abs(res["us"] - us).max(axis=(0, 1)) <= 1e-9 + 100 * eu)
    assert np.all(np.abs(res["xs"] - xs).max(axis=(0, 1))[1:] <= 1e-9 + 100 * ex[1:])
    # and the DPP sweeps of the same session agree with the tile sweeps
    dpp = _gpu_batch(p, idx, exact_qp=exact, tile=False)
    assert dpp["path_detail"] == "traceless"
    assert np.abs(dpp["us"] - res["us"]).max() <= 1e-8 * p["sat"] + 100 * eu.max()


def test_exact_qp_iteration_counts_config3():
    """What the exact mode costs on BASELINE config 3 (T = 40, 20 steps): with the primal-dual phase for the hard solves (the first
    warm steps) a solve takes 2.8 pinned sweeps on average and almost no ratio steps (round 2: 3.4 sweeps and 1.0 ratio step), and
    every solve ends by the KKT test.  A guard: a change that silently sends solves back to the slow path shows up here."""
    p = configs.build(3, batch=512)
    res = _gpu_batch(p, np.arange(512), exact_qp=True)
    n_solves, sweeps, ratio_steps, end_kkt, end_precision, end_cap = res["qp_stats"]
    assert np.all(res["exit_codes"] == 0) and end_cap == 0 and end_kkt + end_precision == n_solves and end_precision <= 2
    assert sweeps <= 3.1 * n_solves, (sweeps, n_solves)
    assert ratio_steps <= 0.15 * n_solves, (ratio_steps, n_solves)


def test_exact_qp_session_rejects_ref_lqr():
    p = configs.build(1, batch=1)
    with pytest.raises(_lib.M4qError):
        _session(p, 1, qp_flags=_lib.QP_REF_LQR, exact_qp=True)


def _oracle_step_sensitivity(p, models, b, k, xs, us, guess):
    """How far the ORACLE's outputs of MPC step k (us[k], xs[k+1], next guesses) move when the SQP guess it starts from is
    perturbed by a few ulp (relative 1e-15): the conditioning of that one step.  At T = 80 (BASELINE config 5) single
    steps exist where this is 0.36 rad/ns on the control - the horizon QP is not determined to fp64 there."""
    n = p["dim_x"]
    Am = models[b if models.shape[0] > 1 else 0]
    model = orc.OracleDMDc(n, n, Am.shape[1] - n, Am)
    exp = orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    outs = []
    for eps in (0.0, 1e-15, -1e-15, 3e-15):
        clock = orc.OracleClock(p["dt"], p["horizon"], p["n_steps"])
        tr = []
        st = dict(step=k, xs=xs[b], us=us[b], X_guess=guess[0] * (1 + eps), U_guess=guess[1])
        (x2, u2), _, _ = orc.mpc(p["x0"][b], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"],
                                 p["Qf"], sat=p["sat"], du=p["du"], start=st, stop=k + 1, trace=tr)
        outs.append((u2[:, k], x2[:, k + 1], tr[-1][0], tr[-1][1]))
    return [max(np.abs(o[i] - outs[0][i]).max() for o in outs[1:]) for i in range(4)]


# config 5's shape made well conditioned (configs.build: drift_scale, r_scale): T = 80 held to the FIXED bounds on every path
_VARIANTS = {"5w": (5, dict(drift_scale=0.125, r_scale=100.0)), "5d": (5, dict(drift_scale=0.125))}
_STRICT = ((1, 1), (1, 2), (2, 1), (3, 1), ("5w", 1), (5, 2))          # (config, order) that may admit NOTHING, on any path


def _build_cfg(cfg, **kw):
    if cfg in _VARIANTS:
        base, extra = _VARIANTS[cfg]
        return configs.build(base, **extra, **kw)
    return configs.build(cfg, **kw)


@pytest.mark.parametrize("path", ["real", "complex", "real9", "tile", "sg"])
@pytest.mark.parametrize("cfg,order,batch,horizon", [(1, 1, 1, None), (1, 2, 1, None), (2, 1, 3, None), (3, 1, 4, None),
                                                      (3, 2, 2, None), (4, 1, 2, 12), (4, 1, 2, None), (5, 1, 2, None),
                                                      ("5w", 1, 2, None), ("5d", 1, 2, None), (5, 2, 2, None)])
def test_closed_loop_stepwise_teacher_forced(cfg, order, batch, horizon, path):
    """Every MPC step of the run, started from the ORACLE's state (states, controls, SQP guesses) through the
    session's checkpoint/restore fields.  The outputs of the step - applied control us[k], next state xs[k+1],
    QP-solve count - must match to 1e-10 whatever the conditioning of the loop; the shifted SQP
    guesses (the far end of a stiff 40-step horizon) to 1e-7.  horizon None = the BASELINE config's own size: every config
    runs at its own T (config 2: 20, configs 3 and 4: 40, config 5: 80) for all of its 20 MPC steps.
    Paths: "sg" = the traceless clipped solve on shared generators and per-member scales (models from build_models; d = 4),
    "real" = the d*d - 1 traceless Hermitian coordinates on DPP rows (M4Q_OPT_NO_TILE), "real9" = the d*d Hermitian
    coordinates (M4Q_OPT_NO_TRACELESS), "tile" = traceless with the backward sweep on matrix-core tiles (what a Liouvillian
    model with a constant target gets by default at d = 2, 3),
    "complex" = the general path.  A step may exceed the fixed bounds only by ten times what the ORACLE itself moves when the
    guess the step starts from is perturbed by 1e-15, and only if profiles/r04_parity_admissions.json lists that step for that
    case (_admit); config 3 order 1 (the headline) and configs 1, 2 admit nothing on any path.
    T = 80 on the headline's arithmetic is pinned STRICTLY by two more cases (round 5): "5w" = config 5 with the anharmonicity
    scaled by 1/8 (the order-1 truncation then grows by 1.003 per step instead of 1.18) and R by 100 (controls mostly off their
    bounds) - the oracle's own sensitivity is <= 2e-9 on the guesses and 1e-14 on the outputs, all 20 steps - and config 5 with
    the order-2 model (real / complex; no tile sweep at order 2): both must admit nothing.  "5d" (drift scaled only, controls
    saturating) holds us[k], xs[k+1] to 1e-10 on every step; only the guesses it leaves behind may use the clause."""
    if path == "real9" and (cfg, order, horizon) not in ((2, 1, None), (3, 1, None), (4, 1, 12)):
        pytest.skip("the d*d-coordinate real path is exercised on one configuration per dimension")
    if path == "sg" and not (order == 1 and cfg == 4):
        pytest.skip("the shared-generator kernel exists at d = 4 with an order-1 model (config 4 at T = 12 and at its own T = 40)")
    if path == "tile" and not (order == 1 and cfg in (1, 2, 3, 5, "5w", "5d")):
        pytest.skip("the tile sweep exists at d = 2, 3 with an order-1 model (every such configuration runs it here)")
    p = _build_cfg(cfg, batch=max(batch, 4) if cfg == 3 else batch, order=order, horizon=horizon)
    idx = np.arange(batch)
    trace = []
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    xs, us, codes, solves = orc.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"],
                                          p["horizon"], p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"],
                                          p["R"], p["Qf"], p["sat"], p["du"], trace=trace)
    q = dict(p)
    q["models"] = models
    ns, T = p["n_steps"], p["horizon"]
    sess = _session(q, batch, force_complex=(path == "complex"), traceless=(path != "real9"), tile=(path == "tile"))
    try:
        if path == "sg":
            # the models built on the device from the shared generators and the members' scales: the kernel of path 4 works on
            # the generators themselves ("sg"); the oracle above on the host-built models of the same members
            sess.build_models(p["dt"], p["generators"], p["scales"][idx])
        sess.load_problem(None if path == "sg" else models, p["x0"][idx], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"],
                          p["plant_op0"], p["plant_ops"])
        assert sess.path_detail() == {"real": "traceless", "real9": "real", "tile": "traceless-tile", "complex": "complex",
                                      "sg": "traceless-sg"}[path]
        xs_t, us_t = np.swapaxes(xs, 1, 2), np.swapaxes(us, 1, 2)          # time-major, as the C ABI holds them
        key = "stepwise[cfg%s-o%d-B%d-T%d-%s]" % (cfg, order, batch, T, path)
        _CASES_RUN.add(key)
        admitted = []
        for k in range(ns):
            if k > 0:
                st = {"xs": np.zeros_like(xs_t), "us": np.zeros_like(us_t),
                      "x_guess": np.stack([trace[b][k][0].T for b in range(batch)]),
                      "u_guess": np.stack([trace[b][k][1].T for b in range(batch)]),
                      "exit_codes": np.zeros(batch, dtype=np.int32), "steps_done": np.full(batch, k, dtype=np.int32)}
                st["xs"][:, :k + 1] = xs_t[:, :k + 1]
                st["us"][:, :k] = us_t[:, :k]
                sess.restore(st)
            sess.run(k, k + 1)
            got = sess.state()
            assert np.array_equal(sess.download(_lib.F_QP_SOLVES, (batch, ns))[:, k], solves[:, k])
            errs = [rel(got["us"][:, k], us_t[:, k]), rel(got["xs"][:, k + 1], xs_t[:, k + 1]),
                    rel(got["x_guess"], np.stack([trace[b][k + 1][0].T for b in range(batch)])),
                    rel(got["u_guess"], np.stack([trace[b][k + 1][1].T for b in range(batch)]))]
            if cfg == "5d":
                assert max(errs[:2]) <= 1e-10, (k, errs)                  # the step's outputs: fixed bound, all 20 steps at T = 80
            if not (max(errs[:2]) <= 1e-10 and max(errs[2:]) <= 1e-7):
                # beyond the fixed bounds: admissible only where the oracle itself is that sensitive to its last bits, and only
                # on the steps the committed record lists for this case (_admit)
                sens = np.max([_oracle_step_sensitivity(p, models, b, k, xs, us, trace[b][k]) for b in range(batch)], axis=0)
                for e, s_k, tol in zip(errs, sens, (1e-10, 1e-10, 1e-7, 1e-7)):
                    assert e <= tol + 10 * s_k, (k, errs, sens.tolist())
                _admit(key, k, errs, sens, (1e-10, 1e-10, 1e-7, 1e-7))
                admitted.append(k)
            assert np.all(got["steps_done"] == k + 1) and np.all(got["exit_codes"] == 0)
        if (cfg, order) in _STRICT:
            # the headline configuration, configs 1, 2 and the two strict T = 80 cases: fixed bounds, every step
            assert admitted == [] or os.environ.get("M4Q_RECORD_ADMISSIONS"), (key, admitted)
    finally:
        sess.close()


def _oracle_exact_step_sensitivity(p, models, b, k, xs, us, guess):
    """As _oracle_step_sensitivity for the exact mode: how far us[k], xs[k+1] of the ORACLE (BVLS solve of the box QP) move when the
    SQP guess the step starts from is perturbed by a relative 1e-15."""
    n = p["dim_x"]
    Am = models[b if models.shape[0] > 1 else 0]
    model = orc.OracleDMDc(n, n, Am.shape[1] - n, Am)
    exp = orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    outs = []
    for eps in (0.0, 1e-15, -1e-15, 3e-15):
        clock = orc.OracleClock(p["dt"], p["horizon"], p["n_steps"])
        st = dict(step=k, xs=xs[b], us=us[b], X_guess=guess[0] * (1 + eps), U_guess=guess[1])
        (x2, u2), _, _ = orc.mpc(p["x0"][b], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"],
                                 p["Qf"], sat=p["sat"], du=p["du"], start=st, stop=k + 1, qp_mode="exact")
        outs.append((u2[:, k], x2[:, k + 1]))
    return [max(np.abs(o[i] - outs[0][i]).max() for o in outs[1:]) for i in range(2)]


@pytest.mark.parametrize("cfg,order,batch,path", [(2, 1, 3, "real"), (3, 1, 2, "real"), (3, 1, 2, "dpp"), (3, 1, 2, "complex"),
                                                  (4, 1, 2, "real")])
def test_closed_loop_exact_stepwise_teacher_forced(cfg, order, batch, path):
    """M4Q_QP_EXACT_BOX at the BASELINE sizes (config 2: T = 20, config 3 order 1: T = 40, config 4: T = 40), every MPC step of
    the run started from the ORACLE's state (exact mode: BVLS on the condensed box QP, the statement of optimize.py:27-54):
    the same number of SQP iterations, us[k] within 1e-9 of the bound and xs[k+1] within 1e-9 - plus, where a step is
    ill-conditioned, ten times what the oracle itself moves under a 1e-15 perturbation of the step's starting guess.
    "real" at d = 2, 3 runs the pinned sweep on matrix-core tiles (m4q_tile3.h, the default), "dpp" the same kernel's DPP sweep
    (M4Q_OPT_NO_TILE)."""
    p = configs.build(cfg, batch=max(batch, 4) if cfg == 3 else batch, order=order)
    idx = np.arange(batch)
    trace = []
    models = p["models"] if p["models"].shape[0] == 1 else p["models"][idx]
    xs, us, codes, solves = orc.mpc_batch(p["x0"][idx], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"],
                                          p["horizon"], p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"],
                                          p["R"], p["Qf"], p["sat"], p["du"], trace=trace, qp_mode="exact")
    assert np.all(codes == 0)
    q = dict(p)
    q["models"] = models
    ns = p["n_steps"]
    sess = _session(q, batch, force_complex=(path == "complex"), exact_qp=True, **({"tile": False} if path == "dpp" else {}))
    try:
        sess.load_problem(models, p["x0"][idx], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"],
                          p["plant_ops"])
        assert sess.path() == ("real" if path == "dpp" else path)
        xs_t, us_t = np.swapaxes(xs, 1, 2), np.swapaxes(us, 1, 2)
        admitted = []
        _CASES_RUN.add("exact_stepwise[cfg%d-o%d-B%d-T%d-%s]" % (cfg, order, batch, p["horizon"], path))
        for k in range(ns):
            if k > 0:
                st = {"xs": np.zeros_like(xs_t), "us": np.zeros_like(us_t),
                      "x_guess": np.stack([trace[b][k][0].T for b in range(batch)]),
                      "u_guess": np.stack([trace[b][k][1].T for b in range(batch)]),
                      "exit_codes": np.zeros(batch, dtype=np.int32), "steps_done": np.full(batch, k, dtype=np.int32)}
                st["xs"][:, :k + 1] = xs_t[:, :k + 1]
                st["us"][:, :k] = us_t[:, :k]
                sess.restore(st)
            sess.run(k, k + 1)
            got = sess.state()
            assert np.all(got["steps_done"] == k + 1) and np.all(got["exit_codes"] == 0), k
            assert np.array_equal(sess.download(_lib.F_QP_SOLVES, (batch, ns))[:, k], solves[:, k]), k
            eu = np.abs(got["us"][:, k] - us_t[:, k]).max() / p["sat"]
            ex = np.abs(got["xs"][:, k + 1] - xs_t[:, k + 1]).max()
            if not (eu <= 1e-9 and ex <= 1e-9):
                sens = np.max([_oracle_exact_step_sensitivity(p, models, b, k, xs, us, trace[b][k]) for b in range(batch)], axis=0)
                assert eu <= 1e-9 + 10 * sens[0] / p["sat"] and ex <= 1e-9 + 10 * sens[1], (k, eu, ex, sens.tolist())
                _admit("exact_stepwise[cfg%d-o%d-B%d-T%d-%s]" % (cfg, order, batch, p["horizon"], path), k, (eu, ex), sens, (1e-9, 1e-9))
                admitted.append(k)
        assert len(admitted) <= ns // 4, admitted          # the sensitivity clause is for the odd step, not the rule
    finally:
        sess.close()


@pytest.mark.parametrize("path", ["complex", "real"])
def test_checkpoint_resume(path):
    """state() after step 7, restore() into a fresh session, run the rest.  On the complex path the result has
    identical bits to an uninterrupted run; on the real path the checkpoint holds the guess in the original complex
    basis, and the two basis changes cost one rounding each."""
    p = configs.build(3, batch=6, horizon=16)
    idx = np.arange(6)
    fc = path == "complex"
    full = _gpu_batch(p, idx, force_complex=fc)
    s1 = _session(p, 6, force_complex=fc)
    s2 = _session(p, 6, force_complex=fc)
    try:
        for s in (s1, s2):
            s.load_problem(p["models"], p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"],
                           p["plant_ops"])
        s1.run(0, 7)
        s2.restore(s1.state())
        s2.run(7, p["n_steps"])
        r2 = s2.results()
    finally:
        s1.close()
        s2.close()
    if fc:
        assert np.array_equal(np.swapaxes(r2["xs"], 1, 2), full["xs"])
        assert np.array_equal(np.swapaxes(r2["us"], 1, 2), full["us"])
    else:
        assert rel(np.swapaxes(r2["xs"], 1, 2), full["xs"]) <= 1e-9 and rel(np.swapaxes(r2["us"], 1, 2), full["us"]) <= 1e-9
    assert np.all(r2["steps_done"] == p["n_steps"])


def test_closed_loop_full_horizon_config3_sample():
    """BASELINE config 3 at its own T = 40 on a 4-member sample of the 65,536 ensemble, free running."""
    p = configs.build(3, batch=64)
    idx = np.array([0, 17, 42, 63])
    res = _gpu_batch(p, idx)
    xs, us, codes, solves = _oracle_batch(p, idx)
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, 0], us[:, :, 0]) <= 1e-10 and rel(res["xs"][:, :, 1], xs[:, :, 1]) <= 1e-10
    eu, ex = _envelope(p, idx, xs, us)
    assert np.all(np.abs(res["us"] - us).max(axis=(0, 1)) <= 1e-9 + 100 * eu)
    assert np.all(np.abs(res["xs"] - xs).max(axis=(0, 1))[1:] <= 1e-9 + 100 * ex[1:])


def test_closed_loop_dense_costs_general_line_search():
    """Hermitian Q, Qf with complex off-diagonal entries and a non-diagonal R: exercises the dense-block line
    search (the diagonal fast path is what every reference scenario takes) and the full Q terms of the sweep."""
    p = configs.build(3, batch=3, horizon=10, n_steps=5)
    rng = np.random.default_rng(11)
    M = rng.standard_normal((9, 9)) + 1j * rng.standard_normal((9, 9))
    p["Q"] = p["Q"] + 0.05 * (M @ M.conj().T)
    M = rng.standard_normal((9, 9)) + 1j * rng.standard_normal((9, 9))
    p["Qf"] = 2 * p["Q"] + 0.05 * (M @ M.conj().T)
    p["R"] = p["R"] @ np.array([[1.0, 0.3], [0.3, 2.0]])
    p["R"] = 0.5 * (p["R"] + p["R"].T)
    idx = np.arange(3)
    res = _gpu_batch(p, idx)
    assert res["path"] == "complex"          # W^H Q W is not real: the problem does not qualify for the real path
    xs, us, codes, solves = _oracle_batch(p, idx)
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, 0], us[:, :, 0]) <= 1e-10 and rel(res["xs"][:, :, 1], xs[:, :, 1]) <= 1e-10
    assert rel(res["us"], us) <= 1e-6 and rel(res["xs"], xs) <= 1e-6


def test_closed_loop_no_warm_start_and_max_iter():
    p = configs.build(1, batch=1)
    res = _gpu_batch(p, np.arange(1), warm_start=False, max_iter=3)
    xs, us, codes, solves = _oracle_batch(p, np.arange(1), warm_start=False, max_iter=3)
    assert np.array_equal(res["qp_solves"], solves) and solves.max() == 3
    assert rel(res["us"][:, :, :3], us[:, :, :3]) <= 1e-9 and rel(res["xs"][:, :, :4], xs[:, :, :4]) <= 1e-9
    assert rel(res["us"], us) <= 1e-5 and rel(res["xs"], xs) <= 1e-5


def test_closed_loop_lqr_mode_vs_oracle():
    """The whole loop with quad_program = the reference's lqr.py arithmetic."""
    p = configs.build(1, batch=1)
    res = _gpu_batch(p, np.arange(1), qp_flags=_lib.QP_REF_LQR)
    xs, us, codes, solves = _oracle_batch(p, np.arange(1), qp_mode="lqr")
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, :3], us[:, :, :3]) <= 1e-9 and rel(res["xs"][:, :, :4], xs[:, :, :4]) <= 1e-9
    assert rel(res["us"], us) <= 1e-5 and rel(res["xs"], xs) <= 1e-5


@pytest.mark.parametrize("exact", [False, True])
def test_mpc_dropin_fused_equals_host_plant_and_oracle(exact):
    """mpc() with this package's QExperiment (fused, one launch) == mpc() with a foreign experiment object
    (one launch per step, plant on the host) == oracle; return shapes and clock mutation as mpc.py:294-304.
    exact: the QPs solved to the box-constrained optimum (exact_qp=True) against the oracle's BVLS loop."""
    p = configs.build(3, batch=1, horizon=12, n_steps=8)
    n, m = p["dim_x"], p["dim_u"]
    model = m4q.DMDc(n, n, p["models"].shape[2] - n, p["models"][0])

    def run(exp, **kw):
        clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
        out = m4q.mpc(p["x0"][0], m, p["order"], p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"], p["Qf"],
                      sat=p["sat"], du=p["du"], progress_bar=False, exact_qp=exact, **kw)
        return out, clock
    (d1, _, c1), clk1 = run(m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0])))
    (d2, _, c2), clk2 = run(orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0])))
    oclk = orc.OracleClock(p["dt"], p["horizon"], p["n_steps"])
    (xo, uo), _, co = orc.mpc(p["x0"][0], m, p["order"], p["X_targ"], p["U_targ"], oclk,
                              orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0])),
                              orc.OracleDMDc(n, n, p["models"].shape[2] - n, p["models"][0]), p["Q"], p["R"], p["Qf"],
                              sat=p["sat"], du=p["du"], qp_mode="exact" if exact else "qp")
    assert c1 == c2 == co == 0
    assert d1[0].shape == (n, 9) and d1[1].shape == (m, 8)
    assert len(clk1.ts_sim) == len(clk2.ts_sim) == len(oclk.ts_sim) == 8
    assert rel(d1[0], xo) <= 1e-8 and rel(d1[1], uo) <= 1e-8
    assert rel(d2[0], xo) <= 1e-8 and rel(d2[1], uo) <= 1e-8


def test_mpc_streaming_refits_the_model_object():
    """streaming=True (mpc.py:281-285): after every MPC step the model object is refitted with the measured transition
    (here OnlineDMDc, recursive least squares), while the loop keeps the operators extracted at entry (quirk Q6): the
    controls equal the non-streaming run, the returned model has moved towards the plant."""
    p = configs.build(1, batch=1)
    A0 = p["models"][0]

    def run(mod, clock_cls, exp, model, x0=None, **kw):
        clock = clock_cls(p["dt"], p["horizon"], p["n_steps"])
        return mod(p["x0"][0] if x0 is None else x0, 1, 1, p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"], p["Qf"],
                   sat=p["sat"], du=p["du"], streaming=True, **kw)
    exp = m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    (xs, us), mdl, code = run(m4q.mpc, m4q.StepClock, exp, m4q.OnlineDMDc.from_bootstrap(4, 4, 4, A0.copy(), alpha=1e-2),
                              progress_bar=False)
    oexp = orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    (xo, uo), omdl, co = run(orc.mpc, orc.OracleClock, oexp, m4q.OnlineDMDc.from_bootstrap(4, 4, 4, A0.copy(), alpha=1e-2))
    assert code == co == 0 and mdl._iteration == omdl._iteration == p["n_steps"]
    # steps 0 and 1 - every SQP iteration of them - to 1e-10; the free-running rest against the ORACLE's own measured envelope
    # (how far its run moves when x0 changes in the last bits), as the free-running closed-loop tests do - not a constant
    assert rel(us[:, :2], uo[:, :2]) <= 1e-10 and rel(xs[:, :3], xo[:, :3]) <= 1e-10
    eu, ex, eA = np.zeros(us.shape[1]), np.zeros(xs.shape[1]), 0.0
    for scale in (1 + 1e-14, 1 - 1e-14, 1 + 7e-14, 1 - 5e-14):
        (xv, uv), vmdl, _ = run(orc.mpc, orc.OracleClock, oexp, m4q.OnlineDMDc.from_bootstrap(4, 4, 4, A0.copy(), alpha=1e-2),
                                x0=p["x0"][0] * scale)
        eu = np.maximum(eu, np.maximum.accumulate(np.abs(uv - uo).max(axis=0)))
        ex = np.maximum(ex, np.maximum.accumulate(np.abs(xv - xo).max(axis=0)))
        eA = max(eA, np.abs(vmdl.A - omdl.A).max())
    assert np.all(np.abs(us - uo).max(axis=0) <= 1e-9 + 100 * eu), (np.abs(us - uo).max(axis=0), eu)
    assert np.all(np.abs(xs - xo).max(axis=0) <= 1e-9 + 100 * ex), (np.abs(xs - xo).max(axis=0), ex)
    assert rel(us, uo) <= 1e-6 and rel(xs, xo) <= 1e-6                               # (and never beyond the old constant)
    assert np.abs(mdl.A - omdl.A).max() <= 1e-9 + 100 * eA and np.abs(mdl.A - A0).max() > 1e-6   # refitted, identically
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    (x2, u2), _, _ = m4q.mpc(p["x0"][0], 1, 1, p["X_targ"], p["U_targ"], clock, exp, m4q.DMDc(4, 4, 4, A0), p["Q"], p["R"], p["Qf"],
                             sat=p["sat"], du=p["du"], progress_bar=False)
    assert rel(us, u2) <= 1e-8                                                    # Q6: the refit does not steer the loop


def test_mpc_exit_condition_and_exit_code_1():
    p = configs.build(1, batch=1)
    model = m4q.DMDc(4, 4, 4, p["models"][0])
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    exp = m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))
    (xs, us), _, code = m4q.mpc(p["x0"][0], 1, 1, p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"], p["Qf"],
                                sat=p["sat"], du=p["du"], progress_bar=False,
                                exit_condition=lambda xn, x, u: xn[3].real > 0.5)
    oclk = orc.OracleClock(p["dt"], p["horizon"], p["n_steps"])
    (xo, uo), _, co = orc.mpc(p["x0"][0], 1, 1, p["X_targ"], p["U_targ"], oclk,
                              orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0])),
                              orc.OracleDMDc(4, 4, 4, p["models"][0]), p["Q"], p["R"], p["Qf"], sat=p["sat"], du=p["du"],
                              exit_condition=lambda xn, x, u: xn[3].real > 0.5)
    assert code == co == 1
    assert xs.shape == xo.shape and us.shape == uo.shape and len(clock.ts_sim) == len(oclk.ts_sim)
    assert rel(xs, xo) <= 1e-8 and rel(us, uo) <= 1e-8


def test_real_and_complex_paths_agree_and_fallback_rules():
    """Same problem on both paths: identical QP-solve counts, results equal to rounding.  A model that does not
    preserve Hermiticity, or a non-Hermitian initial state, silently takes the complex path."""
    p = configs.build(3, batch=8, horizon=12, n_steps=6)
    idx = np.arange(8)
    r = _gpu_batch(p, idx)
    c = _gpu_batch(p, idx, force_complex=True)
    assert r["path"] == "real" and c["path"] == "complex"
    assert np.array_equal(r["qp_solves"], c["qp_solves"])
    assert rel(r["us"], c["us"]) <= 1e-7 and rel(r["xs"], c["xs"]) <= 1e-7
    assert rel(r["us"][:, :, 0], c["us"][:, :, 0]) <= 1e-11
    q = dict(p)
    q["models"] = p["models"].copy()
    q["models"][3, 0, 1] += 1e-3j
    assert _gpu_batch(q, idx)["path"] == "complex"
    q = dict(p)
    q["x0"] = p["x0"].copy()
    q["x0"][2, 1] += 1e-6
    assert _gpu_batch(q, idx)["path"] == "complex"


@pytest.mark.parametrize("path", ["real", "complex"])
def test_closed_loop_per_member_targets_and_plants(path):
    """target_per_instance and plant_per_instance: every member tracks its own (ramped) target through its own plant."""
    p = configs.build(3, batch=5, horizon=10, n_steps=6)
    rng = np.random.default_rng(21)
    cols = p["X_targ"].shape[1]
    ramp = np.minimum(1.0, (np.arange(cols) + 1) / 4.0)
    Xt = np.stack([p["X_targ"] * (ramp * (0.8 + 0.05 * b))[None, :] for b in range(5)])
    Ut = np.stack([0.01 * b * np.ones((2, cols - 1)) for b in range(5)])
    H0 = np.stack([p["plant_op0"][0] * (1 + 0.03 * b) for b in range(5)])
    Hk = np.stack([p["plant_ops"][0] * (1 - 0.02 * b) for b in range(5)])
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = m4q.mpc_batch(p["x0"], p["models"], 2, 1, Xt, Ut, clock, H0, Hk, p["Q"], p["R"], p["Qf"], p["sat"], p["du"],
                        force_complex=(path == "complex"))
    assert res["path"] == path
    xs, us, codes, solves = orc.mpc_batch(p["x0"], p["models"], 2, 1, Xt, Ut, p["dt"], p["horizon"], p["n_steps"], H0, Hk,
                                          p["Q"], p["R"], p["Qf"], p["sat"], p["du"])
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, :2], us[:, :, :2]) <= 1e-9 and rel(res["xs"][:, :, :3], xs[:, :, :3]) <= 1e-9
    assert rel(res["us"], us) <= 1e-6 and rel(res["xs"], xs) <= 1e-6


def test_closed_loop_generator_plant_with_dissipation():
    """M4Q_PLANT_GENERATOR in the loop: Liouvillian plus amplitude damping (the plant is not unitary)."""
    p = configs.build(1, batch=1)
    a = np.array([[0, 1], [0, 0]], dtype=complex)
    def lind(c):
        cd = c.conj().T
        return np.kron(c, c.conj()) - 0.5 * (np.kron(cd @ c, np.eye(2)) + np.kron(np.eye(2), (cd @ c).T))
    L0 = m4q.liouvillian(p["plant_op0"][0]) + 0.02 * lind(a)
    Lk = np.stack([m4q.liouvillian(h) for h in p["plant_ops"][0]])
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = m4q.mpc_batch(p["x0"], p["models"], 1, 1, p["X_targ"], p["U_targ"], clock, L0, Lk, p["Q"], p["R"], p["Qf"], p["sat"],
                        p["du"], plant_kind=_lib.PLANT_GENERATOR)
    xs, us, codes, solves = orc.mpc_batch(p["x0"], p["models"], 1, 1, p["X_targ"], p["U_targ"], p["dt"], p["horizon"],
                                          p["n_steps"], L0[None], list(Lk), p["Q"], p["R"], p["Qf"], p["sat"], p["du"],
                                          generator_plant=True)
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, :3], us[:, :, :3]) <= 1e-9 and rel(res["xs"][:, :, :4], xs[:, :, :4]) <= 1e-9
    assert rel(res["us"], us) <= 1e-4 and rel(res["xs"], xs) <= 1e-4         # free running: conditioning of the loop
    assert abs(res["xs"][0, 0, -1] + res["xs"][0, 3, -1] - 1) < 1e-10        # trace preserved by the Lindbladian


@pytest.mark.parametrize("kw", [{}, {"force_complex": True}, {"exact_qp": True}])
def test_closed_loop_qubit_with_two_quadrature_drives(kw):
    """Shape (4, 2, 1): a detuned qubit driven on both quadratures (sigma_x and sigma_y), the common single-qubit setup
    beyond the reference's one-control tests; model built by discretize_homogeneous from the Liouvillians."""
    from mpc4quantum_amd.configs import SX, SY, SZ, rx
    dt, T, ns = 0.5, 12, 10
    H0, Hk = 0.15 * SZ, [0.5 * SX, 0.5 * SY]
    model = m4q.discretize_homogeneous([m4q.liouvillian(H0)] + [m4q.liouvillian(h) for h in Hk], dt, 1)[None]
    sat = 2 * np.pi * 0.08
    r0 = rx(0.3)
    rho0 = (r0 @ np.diag([1.0, 0]).astype(complex) @ r0.conj().T).reshape(1, 4)
    x0 = np.concatenate([rho0, np.diag([1.0, 0]).astype(complex).reshape(1, 4)])
    target = np.array([0.5, -0.5j, 0.5j, 0.5])                    # |+i><+i|: needs both quadratures
    X_t = np.tile(target[:, None], (1, ns + T + 1))
    U_t = np.zeros((2, ns + T))
    Q = np.eye(4)
    R = 1e-2 / sat ** 2 * np.eye(2)
    clock = m4q.StepClock(dt, T, ns)
    res = m4q.mpc_batch(x0, model, 2, 1, X_t, U_t, clock, H0[None], np.stack(Hk)[None], Q, R, Q, sat, 0.5 * sat, **kw)
    xs, us, codes, solves = orc.mpc_batch(x0, model, 2, 1, X_t, U_t, dt, T, ns, H0[None], Hk, Q, R, Q, sat, 0.5 * sat,
                                          qp_mode="exact" if kw.get("exact_qp") else "qp")
    assert res["path"] == ("complex" if kw.get("force_complex") else "real")
    assert np.array_equal(res["exit_codes"], codes) and np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, :2], us[:, :, :2]) <= 1e-9 and rel(res["xs"][:, :, :3], xs[:, :, :3]) <= 1e-9
    assert rel(res["us"], us) <= 1e-5 and rel(res["xs"], xs) <= 1e-5
    assert np.abs(us[:, 1, :]).max() > 0.1 * sat                  # the sigma_y drive is really used
    fid = np.real(np.einsum("i,bi->b", target.conj(), res["xs"][:, :, -1]))
    assert np.all(fid > 0.9)                                       # and the loop gets there


def test_closed_loop_long_horizon_config5_shape():
    """BASELINE config 5's shape (T = 80) on two members, and the degenerate n_steps = 1 / T = 2 corner."""
    p = configs.build(5, batch=2, n_steps=4)
    res = _gpu_batch(p, np.arange(2))
    xs, us, codes, solves = _oracle_batch(p, np.arange(2))
    assert np.array_equal(res["qp_solves"], solves)
    assert rel(res["us"][:, :, 0], us[:, :, 0]) <= 1e-10 and rel(res["us"], us) <= 1e-6
    q = configs.build(2, batch=3, horizon=2, n_steps=1)
    res = _gpu_batch(q, np.arange(3))
    xs, us, codes, solves = _oracle_batch(q, np.arange(3))
    assert np.array_equal(res["qp_solves"], solves) and rel(res["us"], us) <= 1e-10 and rel(res["xs"], xs) <= 1e-10


def test_mpc_crosstalk_model_on_reduced_states():
    """The reference's crosstalk scenario (tests/test_mpc4quantum.py:281-397, tests/util_qubits.py:39-57): the model
    lives on the two reduced qubit states (n = 8, block diagonal), the plant on the joint 4-level state with a
    sigma_z sigma_z coupling the model does not know; lift = partial traces, proj = Kronecker product
    (experiment.py:238-306).  mpc() steps the device QP and evaluates plant and lift through the experiment object."""
    from mpc4quantum_amd.configs import I2, SX, SY, SZ, rx
    ct = 0.1
    H_joint = [0.5 * ct * np.kron(SZ, SZ), 0.5 * np.kron(SX, I2), 0.5 * np.kron(I2, SY)]
    L1 = [m4q.liouvillian(0 * SX), m4q.liouvillian(SX)]
    L2 = [m4q.liouvillian(0 * SY), m4q.liouvillian(SY)]
    z = np.zeros((4, 4))
    A_cts = [np.block([[L1[0], z], [z, L2[0]]]), np.block([[L1[1], z], [z, z]]), np.block([[z, z], [z, L2[1]]])]
    dt, T, ns = 0.5, 8, 6
    A_dst = m4q.discretize_homogeneous(A_cts, dt, 1)
    sat = 2 * np.pi * 0.1
    r1, r2 = rx(1e-2), rx(-1e-2)
    p0 = np.diag([1.0, 0]).astype(complex)
    p1 = np.diag([0, 1.0]).astype(complex)
    rho0 = np.kron(r1 @ p0 @ r1.conj().T, r2 @ p0 @ r2.conj().T)
    target = np.hstack([p1.flatten(), p1.flatten()])
    X_bm = np.tile(target[:, None], (1, ns + T + 1))
    U_bm = np.zeros((2, ns + T))
    Q = np.diag([1.0, 0, 0, 1, 1, 0, 0, 1])
    R = 1e-2 / sat ** 2 * np.eye(2)

    def run(mod, exp_cls, model_cls, clock_cls, **kw):
        clock = clock_cls(dt, T, ns)
        exp = exp_cls(H_joint[0], H_joint[1:])
        return mod(rho0.flatten(), 2, 1, X_bm, U_bm, clock, exp, model_cls(8, 8, 16, A_dst), Q, R, Q, sat=sat, du=0.5 * sat, **kw)
    (xs, us), _, code = run(m4q.mpc, m4q.QCoupledExperiment, m4q.DMDc, m4q.StepClock, progress_bar=False)
    (xo, uo), _, co = run(orc.mpc, orc.OracleQCoupledExperiment, orc.OracleDMDc, orc.OracleClock)
    assert code == co == 0 and xs.shape == (16, ns + 1) and us.shape == (2, ns)
    assert rel(us[:, :2], uo[:, :2]) <= 1e-9 and rel(xs[:, :3], xo[:, :3]) <= 1e-9
    assert rel(us, uo) <= 1e-5 and rel(xs, xo) <= 1e-5


@pytest.mark.parametrize("path", ["real", "complex", "real-exact"])
@pytest.mark.parametrize("cfg,mf", [(1, 5), (3, 3)])
def test_measure_freq(cfg, mf, path):
    """clock.measure_freq > 1 (reference test_NOT_state_freq, tests/test_mpc4quantum.py:705-804): the plant is measured
    every mf-th step by re-simulating from the last measured state with the held controls replayed in the order
    mpc.py:257 stacks them; the model closes the loop in between (mpc.py:252-267)."""
    p = configs.build(cfg, batch=3, horizon=10 if cfg == 3 else None, n_steps=11)
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    clock.measure_freq = mf
    models = p["models"]
    res = m4q.mpc_batch(p["x0"], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"],
                        p["Q"], p["R"], p["Qf"], p["sat"], p["du"], force_complex=(path == "complex"),
                        exact_qp=path.endswith("exact"))
    xs, us, codes, solves = orc.mpc_batch(p["x0"], models, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], p["dt"],
                                          p["horizon"], p["n_steps"], p["plant_op0"], list(p["plant_ops"][0]), p["Q"], p["R"],
                                          p["Qf"], p["sat"], p["du"], measure_freq=mf,
                                          qp_mode="exact" if path.endswith("exact") else "qp")
    assert np.array_equal(res["qp_solves"], solves)
    # steps 0 and 1 (every SQP iteration of them) to 1e-10; the free-running rest against the ORACLE's own measured envelope (what
    # its run moves by under last-bit perturbations of x0 and of the model), not against a constant; per-step parity is the
    # teacher-forced tests'
    assert rel(res["us"][:, :, :2], us[:, :, :2]) <= 1e-10 and rel(res["xs"][:, :, :3], xs[:, :, :3]) <= 1e-10
    eu, ex = _envelope(p, np.arange(3), xs, us, measure_freq=mf, qp_mode="exact" if path.endswith("exact") else "qp")
    du_k = np.abs(res["us"] - us).max(axis=(0, 1))
    dx_k = np.abs(res["xs"] - xs).max(axis=(0, 1))
    assert np.all(du_k <= 1e-9 + 100 * eu), (du_k, eu)
    assert np.all(dx_k[1:] <= 1e-9 + 100 * ex[1:]), (dx_k, ex)
    assert rel(res["us"], us) <= 1e-4 and rel(res["xs"], xs) <= 1e-4             # (and never beyond the old constant)


def test_mpc_dropin_measure_freq_host_plant_equals_fused():
    p = configs.build(1, batch=1)
    model = m4q.DMDc(4, 4, 4, p["models"][0])
    outs = []
    for exp in (m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0])), orc.OracleQExperiment(p["plant_op0"][0], list(p["plant_ops"][0]))):
        clock = m4q.StepClock(p["dt"], p["horizon"], 10)
        clock.measure_freq = 5
        (xs, us), _, code = m4q.mpc(p["x0"][0], 1, 1, p["X_targ"], p["U_targ"], clock, exp, model, p["Q"], p["R"], p["Qf"],
                                    sat=p["sat"], du=p["du"], progress_bar=False)
        outs.append((xs, us, code))
    assert outs[0][2] == outs[1][2] == 0
    assert rel(outs[0][0], outs[1][0]) <= 1e-8 and rel(outs[0][1], outs[1][1]) <= 1e-8


def test_exit_code_3_on_nonfinite_model():
    p = configs.build(2, batch=5)
    models = np.repeat(p["models"], 5, axis=0)
    models[3, 0, 0] = np.nan
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = m4q.mpc_batch(p["x0"], models, 1, 1, p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"],
                        p["R"], p["Qf"], p["sat"], p["du"])
    assert res["exit_codes"].tolist() == [0, 0, 0, 3, 0]
    assert res["steps_done"].tolist() == [20, 20, 20, 0, 20]
    assert res["qp_solves"][3, 0] == 1 and res["qp_solves"][3, 1:].sum() == 0


# ---------------------------------------------------------------- size-independent properties at scale
@pytest.mark.parametrize("cfg,batch", [(2, 8192), (3, 4096)])
def test_properties_at_scale(cfg, batch):
    """Full BASELINE sizes are beyond the oracle's reach; check what must hold for every member:
    unit trace and Hermiticity of rho_t (the plant is unitary), |u| <= sat, the du band on the first control
    of every step, completion, and bit-identical results for identical members placed in different wavefronts."""
    p = configs.build(cfg, batch=batch)
    if cfg == 3:
        p["models"][batch - 1] = p["models"][0]
        p["x0"][batch - 1] = p["x0"][0]
    else:
        p["x0"][batch - 1] = p["x0"][0]
        p["x0"][batch // 2 + 1] = p["x0"][0]
    res = _gpu_batch(p, np.arange(batch))
    d = p["d"]
    assert np.all(res["exit_codes"] == 0) and np.all(res["steps_done"] == p["n_steps"])
    rho = np.swapaxes(res["xs"], 1, 2).reshape(batch, p["n_steps"] + 1, d, d)
    assert np.abs(rho.trace(axis1=2, axis2=3) - 1).max() < 1e-10
    assert np.abs(rho - np.swapaxes(rho.conj(), 2, 3)).max() < 1e-10
    assert np.abs(res["us"]).max() <= p["sat"] * (1 + 1e-15)
    dus = np.abs(np.diff(res["us"], axis=2))[:, :, 1:]          # steps >= 2 are banded around us[step-1]
    assert dus.max() <= p["du"] * (1 + 1e-12)
    assert np.array_equal(res["xs"][0], res["xs"][batch - 1]) and np.array_equal(res["us"][0], res["us"][batch - 1])
    if cfg == 2:
        assert np.array_equal(res["us"][0], res["us"][batch // 2 + 1])
    assert res["qp_solves"][:, 2:].min() == 1 and res["qp_solves"][:, 2:].max() == 1


@pytest.mark.parametrize("cfg,batch", [(3, 65536), (4, 8192), (5, 131072)])
def test_properties_at_full_baseline_size(cfg, batch):
    """BASELINE config 3 at its full single-GPU size (65,536 per-member models, T = 40), config 4's per-GPU share of
    the 8-way sharding and config 5's (2^20 / 8 = 131,072 members, T = 80, all 20 MPC steps), set up the way bench.py
    does (models built on the device from generators and per-member scales).
    The oracle cannot follow; every member must still finish with exit code 0, keep rho_t Hermitian with unit trace,
    respect the box and the du band, spend exactly one QP solve per warm step - and members with identical inputs placed
    in different wavefronts must produce identical bits."""
    p = configs.build(cfg, batch=batch, host_models=False)
    twin = batch - 3
    p["scales"][twin] = p["scales"][1]
    p["x0"][twin] = p["x0"][1]
    n, m, T, ns, d = p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"], p["d"]
    sess = m4q.EnsembleSession(batch, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=True,
                               target_cols=ns + T + 1)
    try:
        sess.build_models(p["dt"], p["generators"], p["scales"])
        sess.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
        assert sess.path() == "real"
        sess.run(0, ns)
        res = sess.results()
    finally:
        sess.close()
    xs, us = res["xs"], res["us"]                                  # time-major: [B, ns+1, n], [B, ns, m]
    assert np.all(res["exit_codes"] == 0) and np.all(res["steps_done"] == ns)
    rho = xs.reshape(batch, ns + 1, d, d)
    assert np.abs(rho.trace(axis1=2, axis2=3) - 1).max() < 1e-10
    assert np.abs(rho - np.swapaxes(rho.conj(), 2, 3)).max() < 1e-10
    assert np.abs(us).max() <= p["sat"] * (1 + 1e-15)
    assert np.abs(np.diff(us, axis=1))[:, 1:].max() <= p["du"] * (1 + 1e-12)
    assert res["qp_solves"][:, 2:].min() == 1 and res["qp_solves"][:, 2:].max() == 1
    assert np.array_equal(xs[1], xs[twin]) and np.array_equal(us[1], us[twin])
    assert np.abs(us).max() > 0.5 * p["sat"]                       # (and the ensemble is really being driven)


@pytest.mark.parametrize("kw", [{}, {"tile": False}, {"exact_qp": True}, {"exact_qp": True, "tile": False}, {"force_complex": True},
                                {"traceless": False}, {"config": 4}, {"config": 4, "shared_generators": False}])
def test_repeated_launches_are_bit_identical(kw):
    """Rows pull their work from a device-wide queue, heads and tails of a run may land on different workgroups, and in the
    exact mode a solve spans a varying number of passes: none of that may reach the numbers.  Four launches of the same
    4,096-member problem must agree bit for bit (a soak of 116 full-size launches over all modes did)."""
    B = 4096
    kw = dict(kw)
    cfg = kw.pop("config", 3)               # (config 4: the shared-generator kernel of d = 4 and the per-member-model kernel beside it)
    p = configs.build(cfg, batch=B, host_models=False)
    n, m, T, ns = p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
    sess = m4q.EnsembleSession(B, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=True,
                               target_cols=ns + T + 1, **kw)
    try:
        sess.build_models(p["dt"], p["generators"], p["scales"])
        sess.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
        first = None
        for _ in range(4):
            sess.run(0, ns)
            res = sess.state()                                   # states, controls, codes and the final SQP guesses
            res["qp_solves"] = sess.download(_lib.F_QP_SOLVES, (B, ns))
            assert np.all(res["exit_codes"] == 0)
            if first is None:
                first = res
            else:
                for key in ("xs", "us", "qp_solves", "x_guess", "u_guess"):
                    assert np.array_equal(res[key], first[key]), key
    finally:
        sess.close()
    # ... and with the build that stored tests/golden/gpu_checksums.json: an optimisation that changes a rounding anywhere in the
    # loop shows up here and has to be accepted on purpose (parity tests green, then M4Q_STORE_CHECKSUMS=<file> to re-take).
    import hashlib
    import json
    h = hashlib.sha256()
    for key in ("xs", "us", "qp_solves"):
        h.update(np.ascontiguousarray(first[key]).tobytes())
    name = "config%d_B4096_" % cfg + ("exact" if kw.get("exact_qp") else "clip") + (
        "_complex" if kw.get("force_complex") else "_real9" if kw.get("traceless") is False else
        "_real" if kw.get("tile") is False or kw.get("shared_generators") is False else
        "_sg" if cfg == 4 else "_tile")                       # ({}: the default - backward / pinned sweep on tiles; d = 4: shared generators)
    store = os.environ.get("M4Q_STORE_CHECKSUMS")
    if store:
        have = json.load(open(store)) if os.path.exists(store) else {}
        have[name] = h.hexdigest()
        json.dump(have, open(store, "w"), indent=1, sort_keys=True)
        return
    stored = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpu_checksums.json")))
    assert stored[name] == h.hexdigest(), "results of %s differ bit-wise from the build that stored the checksum" % name


# ---------------------------------------------------------------- the closed loop against the REFERENCE's own mpc.py
# tests/golden/mpc_loop.npz: mpc4quantum/mpc.py:128-304 (loaded by path in the build container, tests/golden/make_golden.py)
# around the reference's own lqr.quad_program; the fused kernel runs the same loop with M4Q_QP_REF_LQR.
REF_LOOPS = ["qubit_o1", "qubit_o2_mf5", "qubit_o1_cold_cap4", "transmon_o1", "transmon_o2_mf2_cap5", "coupled_o1_cap6"]
# tests/golden/mpc_loop_long.npz: the same at the BASELINE horizons (config 3: T = 40, 20 steps, members 0 and 1; config 5: T = 80,
# 10 steps), with the reference's own per-step sensitivity to a 1e-15 perturbation of the linearisation point (sens_us, sens_xs)
REF_LOOPS_LONG = ["transmon_o1_T40_m0", "transmon_o1_T40_m1", "transmon_o1_T80_m0"]


def _which(name):
    return "mpc_loop_long" if name in REF_LOOPS_LONG else "mpc_loop"


def _ref_case(g, name):
    k = "loop_" + name + "_"
    c = {key[len(k):]: g[key] for key in g.files if key.startswith(k)}
    for key in ("d", "m", "T", "n_steps", "order", "measure_freq", "max_iter", "exit_index"):
        c[key] = int(c[key])
    for key in ("dt", "sat", "du", "growth", "exit_thr"):
        c[key] = float(c[key])
    c["warm_start"] = bool(c["warm_start"])
    return c


def _ref_plant(c):
    """The golden scenario's plant as this package's experiment object."""
    H = c["H_plant"]
    if c["growth"]:
        return m4q.LExperiment(m4q.liouvillian(H[0]) + c["growth"] * np.identity(c["d"] ** 2), [m4q.liouvillian(h) for h in H[1:]])
    return m4q.QExperiment(H[0], list(H[1:]))


def _ref_mpc(c, model=None, **kw):
    n = c["d"] ** 2
    if model is None:
        model = m4q.DMDc(n, n, c["model"].shape[1] - n, c["model"])
    clock = m4q.StepClock(c["dt"], c["T"], c["n_steps"])
    clock.measure_freq = c["measure_freq"]
    cond = None
    if c["exit_index"] >= 0:
        cond = lambda xn, x, u: abs(xn[c["exit_index"]]) > c["exit_thr"]      # noqa: E731
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        (xs, us), _, code = m4q.mpc(c["x0"], c["m"], c["order"], c["X_targ"], c["U_targ"], clock, _ref_plant(c),
                                    model, c["Q"], c["R"], c["Q"], sat=c["sat"],
                                    du=c["du"], max_iter=c["max_iter"], exit_condition=cond, warm_start=c["warm_start"],
                                    progress_bar=False, qp_flags=_lib.QP_REF_LQR, **kw)
    return xs, us, code, clock


@pytest.mark.parametrize("path", ["real", "complex"])
@pytest.mark.parametrize("name", REF_LOOPS + REF_LOOPS_LONG)
def test_mpc_loop_teacher_forced_vs_reference_mpc_py(golden, name, path):
    """Every MPC step of the REFERENCE's run (its own mpc.py around its own lqr.py), restarted on the device from the
    reference's state: states and controls so far and the SQP guess the reference handed to get_model_along_traj at the
    first QP solve of the step.  The step's outputs us[k], xs[k+1] and its QP-solve count must match to 1e-10; the guess
    the step leaves behind must be the one the reference starts step k+1 from.  Covers measure_freq in {1, 2, 5},
    warm_start off, small max_iter, orders 1 and 2, d = 2, 3, 4; and the BASELINE horizons T = 40 (config 3, 20 steps) and T = 80
    (config 5, 10 steps), where a step's bound is widened by ten times what the REFERENCE itself moves when its linearisation
    point is perturbed by 1e-15 (at most 3e-13: at these horizons, in the arithmetic of lqr.py, the reference determines every
    step to working precision)."""
    c = _ref_case(golden(_which(name)), name)
    n, m, T, ns = c["d"] ** 2, c["m"], c["T"], c["n_steps"]
    steps, Xg, Ug = c["solve_step"], c["solve_Xg"], c["solve_Ug"]
    su, sx, sg = c.get("sens_us", np.zeros(ns)), c.get("sens_xs", np.zeros(ns)), c.get("sens_xg", np.zeros(ns))
    sgu = c.get("sens_ug", np.zeros(ns))
    exp = _ref_plant(c)
    op0, ops = exp.operators()
    sess = m4q.EnsembleSession(1, n, m, c["order"], T, ns, c["dt"], c["sat"], c["du"], c["max_iter"], c["warm_start"],
                               qp_flags=_lib.QP_REF_LQR, plant_kind=exp.plant_kind, target_cols=ns + T + 1,
                               measure_freq=c["measure_freq"], force_complex=(path == "complex"))
    try:
        sess.load_problem(c["model"][None], c["x0"][None], c["X_targ"], c["U_targ"], c["Q"], c["R"], c["Q"], op0, ops)
        assert sess.path() == path
        xs_t, us_t = c["xs"].T[None], c["us"].T[None]
        for k in range(ns):
            idx = np.nonzero(steps == k)[0]
            st = {"xs": np.zeros((1, ns + 1, n), dtype=complex), "us": np.zeros((1, ns, m)),
                  "x_guess": Xg[idx[0]].T[None], "u_guess": Ug[idx[0]].T[None],
                  "exit_codes": np.zeros(1, dtype=np.int32), "steps_done": np.full(1, k, dtype=np.int32)}
            st["xs"][:, :k + 1] = xs_t[:, :k + 1]
            st["us"][:, :k] = us_t[:, :k]
            sess.restore(st)
            sess.run(k, k + 1)
            got = sess.state()
            assert sess.download(_lib.F_QP_SOLVES, (1, ns))[0, k] == len(idx), k
            assert rel(got["us"][:, k], us_t[:, k]) <= 1e-10 + 10 * su[k], (k, rel(got["us"][:, k], us_t[:, k]))
            assert rel(got["xs"][:, k + 1], xs_t[:, k + 1]) <= 1e-10 + 10 * sx[k], (k, rel(got["xs"][:, k + 1], xs_t[:, k + 1]))
            if k + 1 < ns:
                nxt = np.nonzero(steps == k + 1)[0][0]
                # (T = 80: the far end of the guess reaches 1e13 and the reference itself moves it by 2e-3, relative, under a 1e-15
                #  perturbation - sens_xg; the step's outputs above are what the loop applies)
                assert rel(got["x_guess"][0], Xg[nxt].T) <= 1e-7 + 10 * sg[k] and rel(got["u_guess"][0], Ug[nxt].T) <= 1e-7 + 10 * sgu[k], k
    finally:
        sess.close()


@pytest.mark.parametrize("name", REF_LOOPS + REF_LOOPS_LONG)
def test_mpc_dropin_free_running_vs_reference_mpc_py(golden, name):
    """The drop-in mpc() (one fused launch) against what the reference's mpc() returned: exit code, shapes, clock.ts_sim;
    MPC steps 0 and 1 with all their SQP iterations to 1e-10, the rest of the free-running trajectory within the loop's
    conditioning (the teacher-forced test above is the tight one)."""
    c = _ref_case(golden(_which(name)), name)
    xs, us, code, clock = _ref_mpc(c)
    assert code == int(c["exit_code"]) == 0 and xs.shape == c["xs"].shape and us.shape == c["us"].shape
    assert np.array_equal(clock.ts_sim, c["ts_sim"])
    assert rel(us[:, :2], c["us"][:, :2]) <= 1e-10 and rel(xs[:, :3], c["xs"][:, :3]) <= 1e-10
    if "env_us" in c:
        # BASELINE horizons: the REFERENCE's own free-running run moves by env (running maximum over steps) when x0 is scaled by
        # 1 +- 1e-14 - at T = 40 it is chaotic from step 7 on (a bang-bang switch flips: 3.1 = 2 sat by step 13)
        assert np.all(np.abs(us - c["us"]).max(axis=0) <= 1e-9 + 100 * c["env_us"]), (np.abs(us - c["us"]).max(axis=0), c["env_us"])
        assert np.all(np.abs(xs - c["xs"]).max(axis=0) <= 1e-9 + 100 * c["env_xs"])
    else:
        assert rel(us, c["us"]) <= 1e-5 and rel(xs, c["xs"]) <= 1e-5


def test_mpc_dropin_streaming_vs_reference_mpc_py(golden):
    """mpc(streaming=True) around OnlineDMDc with measure_freq = 2 against the reference's mpc() around ITS OnlineDMDc
    (mpc.py:261-267, 281-285): states, controls, and the model object the call hands back."""
    c = _ref_case(golden("mpc_loop"), "qubit_o1_streaming_mf2")
    n = c["d"] ** 2
    model = m4q.OnlineDMDc.from_bootstrap(n, n, c["model"].shape[1] - n, c["model"].copy(), alpha=float(c["alpha"]))
    xs, us, code, clock = _ref_mpc(c, model=model, streaming=True)
    assert code == int(c["exit_code"]) == 0 and xs.shape == c["xs"].shape and us.shape == c["us"].shape
    assert np.array_equal(clock.ts_sim, c["ts_sim"])
    assert rel(us[:, :2], c["us"][:, :2]) <= 1e-10 and rel(xs[:, :3], c["xs"][:, :3]) <= 1e-10
    if "env_us" in c:
        # BASELINE horizons: the REFERENCE's own free-running run moves by env (running maximum over steps) when x0 is scaled by
        # 1 +- 1e-14 - at T = 40 it is chaotic from step 7 on (a bang-bang switch flips: 3.1 = 2 sat by step 13)
        assert np.all(np.abs(us - c["us"]).max(axis=0) <= 1e-9 + 100 * c["env_us"]), (np.abs(us - c["us"]).max(axis=0), c["env_us"])
        assert np.all(np.abs(xs - c["xs"]).max(axis=0) <= 1e-9 + 100 * c["env_xs"])
    else:
        assert rel(us, c["us"]) <= 1e-5 and rel(xs, c["xs"]) <= 1e-5
    assert np.abs(c["model_final"] - c["model"]).max() > 1e-3 and np.abs(model.A - c["model_final"]).max() <= 1e-6


@pytest.mark.parametrize("name", ["qubit_o1_exit_step3", "qubit_o1_exit_step0"])
def test_mpc_dropin_exit_condition_vs_reference_mpc_py(golden, name):
    """exit_condition firing mid-run and at step 0 (mpc.py:289-304): code 1, last attempted entry dropped, us None at step 0."""
    c = _ref_case(golden("mpc_loop"), name)
    xs, us, code, clock = _ref_mpc(c)
    assert code == 1 == int(c["exit_code"]) and xs.shape == c["xs"].shape
    assert rel(xs, c["xs"]) <= 1e-9
    if bool(c["us_is_none"]):
        assert us is None
    else:
        assert us.shape == c["us"].shape and rel(us, c["us"]) <= 1e-9
    assert np.array_equal(clock.ts_sim, c["ts_sim"])


def test_mpc_loop_nonfinite_vs_reference_mpc_py(golden):
    """Exit code 3.  (a) An amplifying plant drives x^H Q x to overflow at MPC step 6: the reference returns code 3 with
    7 states and 6 controls (mpc.py:200-203,298-304) and so does the fused kernel.  (b) DOCUMENTED DIFFERENCE: NaN (or 1e200)
    in x0 makes the reference RAISE numpy.linalg.LinAlgError (pinv at lqr.py:61; mpc.py:200 tests isinf only); a batched
    engine cannot raise for one ensemble member, it ends that member with code 3 at step 0 (states (n, 1), controls None)."""
    g = golden("mpc_loop")
    c = _ref_case(g, "qubit_o1_inf_later")
    xs, us, code, clock = _ref_mpc(c)
    assert code == 3 == int(c["exit_code"])
    assert xs.shape == c["xs"].shape == (4, 7) and us.shape == c["us"].shape == (1, 6)
    mask = np.abs(c["xs"]) > 0
    assert np.abs(xs[mask] / c["xs"][mask] - 1).max() <= 1e-8 and rel(us[:, :2], c["us"][:, :2]) <= 1e-9
    assert np.array_equal(clock.ts_sim, c["ts_sim"])
    for name in ("qubit_o1_nan", "qubit_o1_inf_step0"):
        c = _ref_case(g, name)
        assert str(c["raised"]) == "LinAlgError"
        xs, us, code, clock = _ref_mpc(c)
        assert code == 3 and us is None and xs.shape == (4, 1) and len(clock.ts_sim) == 0


def test_m4q_mpc_batch_entry_point_equals_session_path():
    """The one-shot C entry point m4q_mpc_batch (include/m4q.h; SURVEY.md 8b's headline signature), called through ctypes
    with caller-owned host buffers, against the session API on the same inputs: bit for bit."""
    import ctypes as C
    p = configs.build(3, batch=9, horizon=12, n_steps=6)           # 9 members: a ragged last quad
    B, n, m, T, ns = 9, p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"]
    cols = ns + T + 1
    sess = _session(p, B)
    try:
        sess.load_problem(p["models"], p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
        sess.run(0, ns)
        ref = sess.results()
        prob = sess.problem
    finally:
        sess.close()
    Xt = np.ascontiguousarray(np.asarray(p["X_targ"], dtype=np.complex128)[:, :cols].T)
    Ut = np.zeros((cols, m))
    Ut[:p["U_targ"].shape[1]] = np.real(p["U_targ"]).T[:cols]
    xs = np.empty((B, ns + 1, n), dtype=np.complex128)
    us = np.empty((B, ns, m))
    codes = np.empty(B, dtype=np.int32)
    done = np.empty(B, dtype=np.int32)
    solves = np.empty((B, ns), dtype=np.int32)
    keep = [_lib.cbuf(p["models"]), _lib.cbuf(p["x0"]), _lib.cbuf(Xt), _lib.rbuf(Ut), _lib.cbuf(p["Q"]), _lib.cbuf(p["R"]),
            _lib.cbuf(p["Qf"]), _lib.cbuf(p["plant_op0"]), _lib.cbuf(p["plant_ops"])]
    _lib.check(_lib.lib().m4q_mpc_batch(C.byref(prob), B, *[k[1] for k in keep], xs.ctypes.data_as(_lib._dp),
                                        us.ctypes.data_as(_lib._dp), codes.ctypes.data_as(_lib._ip),
                                        done.ctypes.data_as(_lib._ip), solves.ctypes.data_as(_lib._ip)))
    assert np.array_equal(xs, ref["xs"]) and np.array_equal(us, ref["us"])
    assert np.array_equal(codes, ref["exit_codes"]) and np.array_equal(done, ref["steps_done"])
    assert np.array_equal(solves, ref["qp_solves"]) and np.all(codes == 0) and np.all(done == ns)
    # bad arguments come back as error codes, not crashes
    assert _lib.lib().m4q_mpc_batch(C.byref(prob), B, None, *[k[1] for k in keep[1:]], xs.ctypes.data_as(_lib._dp),
                                    us.ctypes.data_as(_lib._dp), None, None, None) == _lib.E_BADARG


def test_exact_qp_unconverged_solves_surface_as_exit_code_2():
    """BASELINE config 5's horizon (T = 80) is beyond what the exact box-QP iteration resolves in fp64:
    solves that stop at their iteration cap must not pass silently.  The reference's analogue is OSQP stopping at max_iter:
    cvxpy warns, mpc.py:183-197 turns the warning into exit code 2 and ends the run.  Here the member ends with exit code 2 at
    that step, the counters say how many solves ended that way, and mpc() warns."""
    p = configs.build(5, batch=256)          # (3 of these 256 members hit the cap; of 4,096: 48 - tests/probes/exact_cap_probe.py)
    res = _gpu_batch(p, np.arange(256), exact_qp=True)
    n_solves, sweeps, ratio_steps, end_kkt, end_precision, end_cap = res["qp_stats"]
    assert end_cap >= 1 and end_kkt + end_precision + end_cap == n_solves
    assert int((res["exit_codes"] == 2).sum()) == end_cap                       # every capped solve ended its member's run
    assert set(res["exit_codes"].tolist()) <= {0, 2}
    capped = res["exit_codes"] == 2
    assert np.all(res["steps_done"][capped] < p["n_steps"]) and np.all(res["steps_done"][~capped] == p["n_steps"])
    # the same horizon with clipped solves (the default mode) runs through
    clip = _gpu_batch(p, np.arange(256))
    assert np.all(clip["exit_codes"] == 0)
    # drop-in: warning + code 2 + trimmed returns
    b = int(np.nonzero(capped)[0][0])
    model = m4q.DMDc(9, 9, 18, p["models"][b])
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    with pytest.warns(UserWarning, match="iteration cap"):
        (xs, us), _, code = m4q.mpc(p["x0"][b], 2, 1, p["X_targ"], p["U_targ"], clock,
                                    m4q.QExperiment(p["plant_op0"][0], list(p["plant_ops"][0])), model, p["Q"], p["R"], p["Qf"],
                                    sat=p["sat"], du=p["du"], progress_bar=False, exact_qp=True)
    k = int(res["steps_done"][b])
    assert code == 2 and xs.shape == (9, k + 1) and (us is None if k == 0 else us.shape == (2, k)) and len(clock.ts_sim) == k


def test_arithmetic_path_selection():
    """Which arithmetic a session runs on: Liouvillian models with Hermitian states and diagonal costs take the traceless real path
    (d*d - 1 coordinates); M4Q_OPT_NO_TRACELESS keeps the d*d real coordinates; M4Q_QP_REF_LQR (whose cost terms are built on xbar
    itself) and a state / target pair of different trace stay on d*d coordinates too; a model that moves the identity component
    (an amplitude-damping generator: trace preserving, not unital) likewise; a dense complex cost goes to the complex path."""
    p = configs.build(3, batch=3, horizon=8, n_steps=3)

    def detail(mutate=None, **kw):
        q = dict(p)
        if mutate:
            mutate(q)
        sess = _session(q, 3, **kw)
        try:
            sess.load_problem(q["models"], q["x0"], q["X_targ"], q["U_targ"], q["Q"], q["R"], q["Qf"], q["plant_op0"], q["plant_ops"])
            sess.run(0, 1)
            sess.sync()
            return sess.path_detail()
        finally:
            sess.close()
    assert detail() == "traceless-tile"                               # d = 3, order 1, constant target: backward sweep on tiles
    assert detail(tile=False) == "traceless"                          # M4Q_OPT_NO_TILE
    assert detail(tile=True) == "traceless-tile"
    assert detail(traceless=False) == "real"
    assert detail(qp_flags=_lib.QP_REF_LQR) == "real"
    assert detail(force_complex=True) == "complex"
    assert detail(exact_qp=True) == "traceless-tile"                  # ... and the exact solve's pinned sweep
    assert detail(exact_qp=True, tile=False) == "traceless"

    def ramped_target(q):                                             # a target that moves over the window: DPP sweeps
        q["X_targ"] = q["X_targ"] * np.linspace(1.0, 1.0 - 1e-3, q["X_targ"].shape[1])[None, :]
    assert detail(ramped_target) in ("traceless", "real")

    def half_trace_target(q):
        q["X_targ"] = 0.5 * q["X_targ"]
    assert detail(half_trace_target) == "real"

    def non_unital_model(q):
        m = q["models"].copy()
        m[:, 4, 0] += 1e-3                   # rho_11 fed by rho_00: trace not preserved / identity component moved
        q["models"] = m
    assert detail(non_unital_model) == "real"

    def dense_complex_cost(q):
        rng = np.random.default_rng(11)
        M = rng.standard_normal((9, 9)) + 1j * rng.standard_normal((9, 9))
        q["Q"] = q["Q"] + 0.05 * (M @ M.conj().T)
    assert detail(dense_complex_cost) == "complex"


def test_shared_generator_path_selection_and_agreement():
    """Path 4 (M4Q_OPT_NO_SG): a d = 4 session whose models come from m4q_session_build_models with ONE generator set runs the
    clipped traceless solve on the shared generators - same results as the per-member-model kernel to the loop's own sensitivity,
    identical QP-solve counts, repeated launches bit-identical; uploaded models, the exact mode, lqr.py's mode and the opt-out stay
    on per-member models; shapes without that kernel (d = 3) are unaffected."""
    B = 64
    p = configs.build(4, batch=B, horizon=16, n_steps=8)
    # config 4 scales only the drift (J_i); give every member its own drive scales as well, so that the scales that ride on the
    # controls (u~_k = s_ik u_k, B_t = s_ik N_k xg) are exercised - models of the oracle and of the uploaded-model runs rebuilt to match
    rng = np.random.default_rng(44)
    p["scales"] = np.concatenate([p["scales"][:, :1], 1 + 0.05 * rng.standard_normal((B, 3))], axis=1)
    p["models"] = np.ascontiguousarray(m4q.discretize_homogeneous(
        [p["scales"][:, k, None, None] * p["generators"][k][None] for k in range(4)], p["dt"], 1))

    def run(build, **kw):
        sess = _session(p, B, **kw)
        try:
            if build:
                sess.build_models(p["dt"], p["generators"], p["scales"])
            sess.load_problem(None if build else p["models"], p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"],
                              p["plant_op0"], p["plant_ops"])
            sess.run(0, p["n_steps"])
            a = sess.results()
            sess.run(0, p["n_steps"])
            b = sess.results()
            for key in ("xs", "us", "qp_solves", "exit_codes"):
                assert np.array_equal(a[key], b[key]), key
            a["xs"], a["us"] = np.swapaxes(a["xs"], 1, 2), np.swapaxes(a["us"], 1, 2)        # (B, n, steps + 1), (B, m, steps)
            return sess.path_detail(), a, sess.info()
        finally:
            sess.close()
    d_sg, r_sg, i_sg = run(True)
    assert d_sg == "traceless-sg"
    d_pm, r_pm, i_pm = run(True, shared_generators=False)
    assert d_pm == "traceless"
    assert i_sg["lds_bytes"] < i_pm["lds_bytes"] and i_sg["lds_bytes"] <= 20480       # eight workgroups per CU: two wavefronts per SIMD
    assert run(False)[0] == "traceless"                                                 # uploaded models
    assert run(True, exact_qp=True)[0] == "traceless"
    assert run(True, qp_flags=_lib.QP_REF_LQR)[0] == "real"
    assert np.all(r_sg["exit_codes"] == 0) and np.array_equal(r_sg["qp_solves"], r_pm["qp_solves"])
    assert rel(r_sg["us"][:, :, 0], r_pm["us"][:, :, 0]) <= 1e-10 and rel(r_sg["xs"][:, :, 1], r_pm["xs"][:, :, 1]) <= 1e-10
    idx = np.arange(4)
    xs, us, codes, solves = _oracle_batch(p, idx)
    eu, ex = _envelope(p, idx, xs, us)
    assert np.array_equal(r_sg["qp_solves"][idx], solves)
    assert np.all(np.abs(r_sg["us"][idx] - us).max(axis=(0, 1)) <= 1e-9 + 100 * eu)
    assert np.all(np.abs(r_sg["xs"][idx] - xs).max(axis=(0, 1))[1:] <= 1e-9 + 100 * ex[1:])
    # the other code paths of the shared-generator kernel, each against the per-member-model kernel on the same device-built models
    # (first MPC step to 1e-10, solve counts, the run within the loop's own sensitivity): measure_freq = 2 (the model closes the loop
    # on the unmeasured steps: the provider's scaled controls in DMDc.predict's arithmetic), a target that moves over the window (the
    # sweep's general form instead of its constant-target form), a batch that does not fill its last wavefront
    def pair(Bn, mutate=None, **kw):
        q = {k: (v[:Bn] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in p.items()}
        if mutate:
            mutate(q)
        out = []
        for sg in (None, False):
            sess = _session(q, Bn, shared_generators=sg, **kw)
            try:
                sess.build_models(q["dt"], q["generators"], q["scales"])
                sess.load_problem(None, q["x0"], q["X_targ"], q["U_targ"], q["Q"], q["R"], q["Qf"], q["plant_op0"], q["plant_ops"])
                sess.run(0, q["n_steps"])
                out.append((sess.path_detail(), sess.results()))
            finally:
                sess.close()
        (da, a), (db, b) = out
        assert da == "traceless-sg" and db == "traceless", (da, db)
        assert np.array_equal(a["qp_solves"], b["qp_solves"]) and np.all(a["exit_codes"] == 0)
        assert rel(a["us"][:, 0], b["us"][:, 0]) <= 1e-10 and rel(a["xs"][:, 1], b["xs"][:, 1]) <= 1e-10
        return np.abs(a["us"] - b["us"]).max()

    def ramped(q):                     # a target that moves between two unit-trace states: still traceless, no longer constant
        lam = np.linspace(0.0, 0.05, q["X_targ"].shape[1])[None, :]
        q["X_targ"] = (1 - lam) * q["X_targ"] + lam * q["x0"][0][:, None]
    assert pair(16, measure_freq=2) <= 1e-6 * p["sat"]
    assert pair(16, ramped) <= 1e-6 * p["sat"]
    assert pair(5) <= 1e-6 * p["sat"]
    # the ensemble entry point with generators instead of models: same path, same numbers
    clock = m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
    res = m4q.mpc_batch(p["x0"], None, p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock, p["plant_op0"], p["plant_ops"], p["Q"],
                        p["R"], p["Qf"], p["sat"], p["du"], generators=p["generators"], scales=p["scales"])
    assert res["path_detail"] == "traceless-sg" and np.array_equal(res["us"], r_sg["us"]) and np.array_equal(res["xs"], r_sg["xs"])
    q3 = configs.build(3, batch=4, horizon=8, n_steps=3)
    s3 = _session(q3, 4)
    try:
        s3.build_models(q3["dt"], q3["generators"], q3["scales"])
        s3.load_problem(None, q3["x0"], q3["X_targ"], q3["U_targ"], q3["Q"], q3["R"], q3["Qf"], q3["plant_op0"], q3["plant_ops"])
        assert s3.path_detail() == "traceless-tile"
    finally:
        s3.close()


def test_mpc_batch_sharded_rccl_single_rank(tmp_path):
    """The product's multi-GPU function on its RCCL path, through the C ABI alone (m4q_comm_*: ncclCommInitRank, ncclGather on
    the communicator's stream behind the kernel) with one rank - all a one-GPU box allows: the session's outputs bound into the
    library-owned gather buffer, one gather of device memory, unpack.  Must equal mpc_batch bit for bit, whole state history and
    final-state-only.  The child process must never have imported torch."""
    import subprocess
    import sys
    script = tmp_path / "sharded_rccl.py"
    script.write_text('''
import os, sys
import numpy as np
sys.path.insert(0, %r)
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"; os.environ["LOCAL_RANK"] = "0"; os.environ.setdefault("MASTER_PORT", "29533")
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
from mpc4quantum_amd.distributed import mpc_batch_sharded, RcclComm
p = configs.build(3, batch=10, horizon=12, n_steps=6)
def clock(): return m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
args = lambda: (p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock(), p["plant_op0"], p["plant_ops"],
                p["Q"], p["R"], p["Qf"], p["sat"], p["du"])
ref = m4q.mpc_batch(*args())
for final_only in (False, True):
    got = mpc_batch_sharded(*args(), final_state_only=final_only)
    xs = ref["xs"][:, :, -1:] if final_only else ref["xs"]
    assert np.array_equal(got["xs"], xs), ("xs", final_only)
    for k in ("us", "exit_codes", "steps_done", "qp_solves"):
        assert np.array_equal(got[k], ref[k]), (k, final_only)
# an ensemble given by shared generators and per-member scales (d = 4: the shared-generator kernel), models built by every rank
q = configs.build(4, batch=9, horizon=10, n_steps=4)
clk = lambda: m4q.StepClock(q["dt"], q["horizon"], q["n_steps"])
gargs = lambda: (q["x0"], None, q["dim_u"], q["order"], q["X_targ"], q["U_targ"], clk(), q["plant_op0"], q["plant_ops"], q["Q"], q["R"],
                 q["Qf"], q["sat"], q["du"])
ref = m4q.mpc_batch(*gargs(), generators=q["generators"], scales=q["scales"])
assert ref["path_detail"] == "traceless-sg"
got = mpc_batch_sharded(*gargs(), generators=q["generators"], scales=q["scales"])
for k in ("xs", "us", "exit_codes", "steps_done", "qp_solves"):
    assert np.array_equal(got[k], ref[k]), ("generators", k)
comm = RcclComm.from_env()
assert comm.allreduce([1.5, 2.0], "sum").tolist() == [1.5, 2.0] and comm.allreduce([3.0], "max").tolist() == [3.0]
comm.barrier()
comm.close()
assert "torch" not in sys.modules
print("sharded rccl ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sharded rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_mpc_batch_sharded_two_ranks_on_one_gpu(tmp_path):
    """Two ranks (host transport of the test harness, both on the box's one GPU): each runs ITS contiguous block of an 11-member
    ensemble (6 + 5: the padded-row case) through the HIP kernels, one gather puts the ensemble together on rank 0.  Must equal
    one mpc_batch over all 11 members bit for bit.  (RCCL itself cannot put two ranks on one device; its one-rank case is the
    test above.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sharded_two.py"
    script.write_text('''
import os, sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import torch.distributed as dist
import mpc4quantum_amd as m4q
from mpc4quantum_amd import configs
from mpc4quantum_amd.distributed import mpc_batch_sharded, shard_bounds
from gloo_transport import GlooTransport
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
p = configs.build(3, batch=11, horizon=12, n_steps=6)
def clock(): return m4q.StepClock(p["dt"], p["horizon"], p["n_steps"])
args = lambda: (p["x0"], p["models"], p["dim_u"], p["order"], p["X_targ"], p["U_targ"], clock(), p["plant_op0"], p["plant_ops"],
                p["Q"], p["R"], p["Qf"], p["sat"], p["du"])
assert [shard_bounds(11, r, 2) for r in range(2)] == [(0, 6), (6, 11)]
for final_only in (False, True):
    got = mpc_batch_sharded(*args(), transport=GlooTransport(), final_state_only=final_only)
    if rank == 0:
        ref = m4q.mpc_batch(*args())
        xs = ref["xs"][:, :, -1:] if final_only else ref["xs"]
        assert got["xs"].shape == xs.shape and np.array_equal(got["xs"], xs), ("xs", final_only)
        for k in ("us", "exit_codes", "steps_done", "qp_solves"):
            assert np.array_equal(got[k], ref[k]), (k, final_only)
    else:
        assert got is None
dist.barrier()
dist.destroy_process_group()
print("rank %%d ok" %% rank)
''' % (root, root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=600))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and ("rank %d ok" % r) in so, so[-2000:] + se[-4000:]


def test_bench_line_schema():
    """bench.py prints ONE JSON line with the driver's contract fields, the roofline of the dominant kernel and the CPU baseline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "512", "--cpu-cores",
                          "2", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "MPC horizon-steps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["instances_ok"] == 512
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "valu_f64" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and "no counter record" in r["traffic_note"]          # 512 members: not a profiled configuration
    assert abs(d["value"] - d["config"]["qp_solves_per_step"] * d["config"]["horizon"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 2 and c["value"] > 0 and c["unit"] == d["unit"] and "sample" in c


def test_bench_gpus_flag_on_a_one_gpu_box():
    """`bench.py --gpus 2` on a box with one GPU: refused (exit 2) before any rank starts, by a device count taken in a child
    process - the launcher never opens the device; and `--gpus 1 --force-dist` (the N > 1 code path with one rank: RCCL
    communicator, bound gather buffers, one gather per run) reports the number of ranks RCCL joined."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    have = int(subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--device-count"], capture_output=True, text=True,
                              timeout=300, cwd=root, env=env).stdout.strip())
    assert have >= 1
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--batch", "256"],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 2 and "usable GPU" in out.stderr and out.stdout.strip() == ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                          "--batch", "512", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and "x1" in d["config"]["parallelism"] and d["config"]["instances_ok"] == 512


def test_config5_whole_ensemble_on_one_gpu():
    """Maximum size: BASELINE config 5's WHOLE ensemble (2^20 members, T = 80, 20 MPC steps; 8.9e8 horizon-steps) resident on one
    MI355X (25 GB of its 288 GB), models built on the device.  Every member must finish with exit code 0, respect the box and the
    du band, spend one QP solve per warm step; identical members at the two ends of the ensemble must produce identical bits; a
    strided sample of the states must be Hermitian with unit trace."""
    B = 1 << 20
    p = configs.build(5, batch=B, host_models=False)
    twin = B - 5
    p["scales"][twin] = p["scales"][3]
    p["x0"][twin] = p["x0"][3]
    n, m, T, ns, d = p["dim_x"], p["dim_u"], p["horizon"], p["n_steps"], p["d"]
    sess = m4q.EnsembleSession(B, n, m, p["order"], T, ns, p["dt"], p["sat"], p["du"], model_per_instance=True, target_cols=ns + T + 1)
    try:
        sess.build_models(p["dt"], p["generators"], p["scales"])
        sess.load_problem(None, p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
        assert sess.path() == "real"
        sess.run(0, ns)
        ms, launches = sess.kernel_ms()
        codes = sess.download(_lib.F_CODES, (B,))
        done = sess.download(_lib.F_STEPS_DONE, (B,))
        solves = sess.download(_lib.F_QP_SOLVES, (B, ns))
        us = sess.download(_lib.F_US, (B, ns, m))
        xs = sess.download(_lib.F_XS, (B, ns + 1, n))
        hbm = sess.info()["hbm_bytes"]
    finally:
        sess.close()
    assert launches == 1 and hbm > 20e9
    assert not codes.any() and np.all(done == ns)
    assert solves[:, 2:].min() == 1 and solves[:, 2:].max() == 1
    assert np.abs(us).max() <= p["sat"] * (1 + 1e-15) and np.abs(np.diff(us, axis=1))[:, 1:].max() <= p["du"] * (1 + 1e-12)
    assert np.array_equal(xs[3], xs[twin]) and np.array_equal(us[3], us[twin])
    rho = xs[::257].reshape(-1, ns + 1, d, d)
    assert np.abs(rho.trace(axis1=2, axis2=3) - 1).max() < 1e-10 and np.abs(rho - np.swapaxes(rho.conj(), 2, 3)).max() < 1e-10
    units = int(solves.astype(np.int64).sum()) * T
    print("config 5 whole ensemble on one GPU: %.1f ms, %.3g horizon-steps/s, %.1f GB resident" % (ms, units / (ms * 1e-3), hbm / 1e9))


def test_kernel_watchdog_turns_a_launch_that_does_not_end_into_an_error(monkeypatch):
    """The persistent kernel's exits depend on data (queue empty, every row finished, a tail waiting for its head's flag).  Every
    wavefront therefore also leaves when the device's constant clock has advanced M4Q_KERNEL_TIMEOUT_S (default 300 s) since it
    started, and the host reports M4Q_E_TIMEOUT instead of results.  Forced here with a deadline of 0.1 microseconds."""
    p = configs.build(3, batch=8, horizon=12, n_steps=6)
    sess = _session(p, 8)
    try:
        sess.load_problem(p["models"], p["x0"], p["X_targ"], p["U_targ"], p["Q"], p["R"], p["Qf"], p["plant_op0"], p["plant_ops"])
        monkeypatch.setenv("M4Q_KERNEL_TIMEOUT_S", "1e-7")
        sess.run()
        with pytest.raises(_lib.M4qError) as err:
            sess.sync()
        assert err.value.code == _lib.E_TIMEOUT and "watchdog" in str(err.value)
        monkeypatch.delenv("M4Q_KERNEL_TIMEOUT_S")
        sess.run()                                             # the session stays usable
        res = sess.results()
        assert np.all(res["exit_codes"] == 0) and np.all(res["steps_done"] == 6)
        # a launch queued behind the abandoned one must not wipe its flag before the host has looked
        monkeypatch.setenv("M4Q_KERNEL_TIMEOUT_S", "1e-7")
        sess.run()
        monkeypatch.delenv("M4Q_KERNEL_TIMEOUT_S")
        sess.run()
        with pytest.raises(_lib.M4qError) as err:
            sess.sync()
        assert err.value.code == _lib.E_TIMEOUT
        sess.run()
        assert np.all(sess.results()["steps_done"] == 6)
    finally:
        sess.close()
