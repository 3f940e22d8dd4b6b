"""Pins the CPU oracle (oracle/m4q_oracle.py) against outputs of the reference's own files
(tests/golden/*.npz, made by tests/golden/make_golden.py) and against the reference's one
known-answer test for this path (tests/test_mpc4quantum.py:147-188)."""
import numpy as np
import pytest

from oracle import m4q_oracle as orc

TOL = 1e-12
SYSTEMS = [("qubit", 1), ("qubit", 2), ("transmon", 1), ("transmon", 2), ("coupled", 1)]
DIMS = {"qubit": (2, 1), "transmon": (3, 2), "coupled": (4, 3)}


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("m", [1, 2, 3])
def test_library_tables(golden, order, m):
    g = golden("library_tables")
    key = "o%d_m%d" % (order, m)
    powers = np.vstack(orc.create_power_list(order, m))
    assert np.array_equal(powers, g[key + "_powers"])
    assert orc.size_of_library(order, m) == int(g[key + "_size"])
    dpow, dcoef = orc.diff_tables(order, m)
    assert np.array_equal(np.stack(dcoef), g[key + "_dcoef"])
    u = g[key + "_u"]
    assert np.allclose(orc.monomials(u, orc.create_power_list(order, m)), g[key + "_lib"], rtol=0, atol=TOL)
    dlib = np.stack([orc.monomials(u, dp) for dp in dpow])
    assert np.allclose(dlib, g[key + "_dlib"], rtol=0, atol=TOL)


def test_krtimes(golden):
    g = golden("library_tables")
    assert np.allclose(orc.krtimes(g["kr_a"], g["kr_b"]), g["kr_out"], rtol=0, atol=TOL)
    # probe recorded in SURVEY.md 8(a8)
    out = orc.krtimes(np.array([[10.], [20.]]), np.array([[1.], [2.], [3.]]))
    assert out.reshape(-1).tolist() == [10, 20, 30, 20, 40, 60]
    with pytest.raises(ValueError):
        orc.krtimes(np.ones((2, 3)), np.ones((2, 4)))


@pytest.mark.parametrize("name,order", SYSTEMS)
def test_discretize(golden, name, order):
    g = golden("discretize")
    out = orc.discretize_homogeneous(list(g[name + "_A_cts"]), float(g[name + "_dt"]), order)
    assert np.abs(out - g["%s_o%d" % (name, order)]).max() <= TOL


def test_discretize_known_answer():
    """Order 1, dt 1: [I + A | N_1 | N_2] (reference tests/test_mpc4quantum.py:147-188)."""
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    sy = np.array([[0, -1j], [1j, 0]], dtype=complex)
    basis = [np.outer(np.eye(2)[i], np.eye(2)[j]) for i in range(2) for j in range(2)]
    l1 = [orc.vectorize_me(op, basis) for op in (0 * sx, sx)]
    l2 = [orc.vectorize_me(op, basis) for op in (0 * sy, sy)]
    z = np.zeros((4, 4))
    ops = [np.block([[l1[0], z], [z, l2[0]]]), np.block([[l1[1], z], [z, z]]), np.block([[z, z], [z, l2[1]]])]
    out = orc.discretize_homogeneous(ops, 1, 1)
    expect = np.hstack([ops[0] + np.identity(8), ops[1], ops[2]])
    assert np.isclose(out, expect).all()


def test_vectorize_me_closed_form():
    rng = np.random.default_rng(0)
    for d in (2, 3, 4):
        M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        H = M + M.conj().T
        basis = [np.outer(np.eye(d)[i], np.eye(d)[j]) for i in range(d) for j in range(d)]
        assert np.abs(orc.vectorize_me(H, basis) - orc.liouvillian_ij(H)).max() < 1e-13


@pytest.mark.parametrize("name,order", SYSTEMS)
def test_linearize(golden, name, order):
    g = golden("linearize")
    key = "%s_o%d" % (name, order)
    d, m = DIMS[name]
    n = d * d
    model = g[key + "_model"]
    dm = orc.OracleDMDc(n, n, model.shape[1] - n, model)
    wm = orc.OracleWrapModel(*dm.get_discrete(), m, order)
    xs, us = g[key + "_xs"], g[key + "_us"]
    A_ls, B_ls, D_ls = wm.get_model_along_traj(xs, us, np.arange(us.shape[1]))
    assert np.abs(np.stack(A_ls) - g[key + "_A"]).max() <= TOL
    assert np.abs(np.stack(B_ls) - g[key + "_B"]).max() <= TOL
    assert np.abs(np.stack(D_ls) - g[key + "_D"]).max() <= TOL
    assert np.abs(wm.lift_u(us) - g[key + "_liftu"]).max() <= TOL
    f = np.hstack([wm.f(xs[:, i], us[:, i]) for i in range(us.shape[1])])
    assert np.abs(f - g[key + "_f"]).max() <= TOL
    ux = orc.krtimes(wm.lift_u(us[:, :1]), xs[:, :1])
    assert np.abs(dm.predict(xs[:, :1], ux) - g[key + "_predict"]).max() <= TOL
    # identities recorded in SURVEY.md 3.2: f == A_t x and Delta == -B_t u
    for i in range(us.shape[1]):
        assert np.abs(f[:, i] - A_ls[i] @ xs[:, i]).max() < 1e-13
        assert np.abs(D_ls[i][:, 0] + B_ls[i] @ us[:, i]).max() < 1e-13


def test_wrapmodel_dimension_check():
    with pytest.raises(ValueError):
        orc.OracleWrapModel(np.eye(4), np.zeros((4, 12)), 1, 1)


@pytest.mark.parametrize("name,order", SYSTEMS)
@pytest.mark.parametrize("tag", ["free", "sat"])
def test_lqr_quad_program(golden, name, order, tag):
    g = golden("lqr")
    key = "%s_o%d" % (name, order)
    k2 = key + "_" + tag
    A_ls, B_ls = list(g[key + "_A"]), list(g[key + "_B"])
    T = len(A_ls)
    m = B_ls[0].shape[1]
    Q_ls = [g[key + "_Q"]] * T + [g[key + "_Qf"]]
    R_ls = [float(g[k2 + "_r"]) * np.identity(m)] * T
    X, U, cost, gains = orc.lqr_quad_program(g[key + "_x0"], g[key + "_X_bm"], g[key + "_U_bm"], Q_ls, R_ls,
                                             A_ls, B_ls, None, float(g[k2 + "_sat"]), None)
    scale = max(1.0, np.abs(g[k2 + "_gains"]).max())
    assert np.abs(np.stack(gains) - g[k2 + "_gains"]).max() <= 1e-10 * scale
    assert np.abs(X - g[k2 + "_X"]).max() <= 1e-10
    assert np.abs(U - g[k2 + "_U"]).max() <= 1e-10
    assert abs(cost - float(g[k2 + "_cost"])) <= 1e-10 * max(1.0, abs(cost))
    if tag == "sat":
        assert np.isclose(np.abs(U).max(), float(g[k2 + "_sat"]))     # the bound is active in this fixture


def test_clipped_riccati_vs_exact_box_qp():
    """Quantifies the documented difference: with no bound active the Riccati path IS the QP solution;
    with bounds active it is feasible and its cost is above the exact optimum (scipy BVLS on the condensed problem)."""
    from mpc4quantum_amd import configs
    p = configs.build(1, batch=1)
    n, m, T = 4, 1, p["horizon"]
    mdl = p["models"][0]
    wm = orc.OracleWrapModel(mdl[:, :n], mdl[:, n:], m, 1)
    rng = np.random.default_rng(3)
    Xg = np.tile(p["x0"][0][:, None], (1, T + 1))
    Ug = 0.3 * p["sat"] * rng.uniform(-1, 1, (m, T))
    A_ls, B_ls, D_ls = wm.get_model_along_traj(Xg, Ug, np.arange(T))
    X_bm, U_bm = p["X_targ"][:, :T + 1], p["U_targ"][:, :T]
    Q_ls, R_ls = [p["Q"]] * T + [p["Qf"]], [p["R"]] * T
    for sat, active in ((1e3, False), (p["sat"], True)):
        Xc, Uc, cc, _ = orc.quad_program(Xg[:, 0], X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls, None, sat, None)
        Xe, Ue, ce = orc.exact_quad_program(Xg[:, 0], X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls, None, sat, None)
        assert np.abs(Ue).max() <= sat * (1 + 1e-12)
        if not active:
            assert np.abs(Uc - Ue).max() < 1e-8 and abs(cc - ce) < 1e-8 * max(1, ce)
        else:
            assert np.isclose(np.abs(Ue).max(), sat) and cc >= ce - 1e-9 and np.abs(Uc - Ue).max() > 1e-3


def _config3_ltv_with_delta(b, T, rng):
    """Config 3's model b (order 1) linearised along a model rollout under small random controls (Delta_t = -B_t u_t != 0,
    Hermitian states) with a target RAMPED from the initial state to |1><1| (xbar_{t+1} != xbar_t) and a non-zero control
    target: everything the Delta / xbar_{t+1} extension of lqr.py's sweep has to get right, at the headline's horizon."""
    from mpc4quantum_amd import configs
    p = configs.build(3, batch=b + 1, order=1, horizon=T)
    n, m = 9, 2
    mod = p["models"][b]
    wm = orc.OracleWrapModel(mod[:, :n], mod[:, n:], m, 1)
    ug = 0.1 * p["sat"] * rng.uniform(-1, 1, (m, T))
    xg = np.zeros((n, T + 1), dtype=complex)
    xg[:, 0] = p["x0"][b]
    for t in range(T):
        xg[:, t + 1] = wm.f(xg[:, t].reshape(-1, 1), ug[:, t]).reshape(-1)
    A_ls, B_ls, D_ls = wm.get_model_along_traj(xg, ug, np.arange(T))
    lam = np.linspace(0, 1, T + 1)
    Xb = (1 - lam)[None, :] * p["x0"][b][:, None] + lam[None, :] * p["X_targ"][:, :1]
    Ub = 0.02 * p["sat"] * rng.standard_normal((m, T))
    return p, xg[:, 0], Xb, Ub, A_ls, B_ls, D_ls


@pytest.mark.parametrize("r_scale", [1.0, 300.0])
def test_affine_riccati_vs_dense_kkt_at_the_headline_horizon(r_scale):
    """The oracle's `qp` mode - the checker of the headline arithmetic - against the independent dense KKT solve of the
    statement at optimize.py:27-41,54 at config 3's own T = 40 (n = 9: 720 + 80 real unknowns, 720 equalities), with
    Delta != 0, a ramped state target and a non-zero control target; bounds inactive.  (The small-T pins are
    test_qp_mode_vs_independent_kkt on the device and test_clipped_riccati_vs_exact_box_qp above.)"""
    rng = np.random.default_rng(17)
    T = 40
    for b in range(2):
        p, x0, Xb, Ub, A_ls, B_ls, D_ls = _config3_ltv_with_delta(b, T, rng)
        assert max(np.abs(d).max() for d in D_ls) > 1e-2 and np.abs(Xb[:, 1] - Xb[:, 0]).max() > 1e-2
        Q_ls, R_ls = [p["Q"]] * T + [p["Qf"]], [r_scale * p["R"]] * T
        X, U, cost, _ = orc.quad_program(x0, Xb, Ub, Q_ls, R_ls, A_ls, B_ls, D_ls, None, 1e6, None)
        Xk, Uk = orc.kkt_quad_program(x0, Xb, Ub, Q_ls, R_ls, A_ls, B_ls, D_ls)
        assert np.abs(Uk).max() > 0.5
        assert np.abs(U - Uk).max() <= 1e-9 * max(1.0, np.abs(Uk).max())
        assert np.abs(X - Xk).max() <= 1e-9 * max(1.0, np.abs(Xk).max())


def test_exact_box_qp_oracle_satisfies_kkt():
    """The BVLS reference solution of the box-constrained QP (optimize.py:27-54) satisfies the KKT conditions of that
    statement, evaluated independently through the adjoint recursion: zero gradient on interior controls, gradient
    pushing outward on controls at a bound.  Pins `exact_quad_program`, the checker of the device's exact mode."""
    from mpc4quantum_amd import configs
    p = configs.build(3, batch=1, order=2, horizon=12)
    n, m, T = 9, 2, 12
    mdl = p["models"][0]
    wm = orc.OracleWrapModel(mdl[:, :n], mdl[:, n:], m, 2)
    rng = np.random.default_rng(11)
    Xg = np.tile(p["x0"][0][:, None], (1, T + 1))
    Ug = 0.3 * p["sat"] * rng.uniform(-1, 1, (m, T))
    A_ls, B_ls, D_ls = wm.get_model_along_traj(Xg, Ug, np.arange(T))
    X_bm, U_bm = p["X_targ"][:, :T + 1], p["U_targ"][:, :T].real
    Q_ls, R_ls = [p["Q"]] * T + [p["Qf"]], [p["R"]] * T
    sat, du = p["sat"], 0.2 * p["sat"]
    u_prev = np.array([0.05, -0.02])
    Q_ls = [0.02 * q for q in Q_ls]                      # softer tracking: a mix of interior and saturated controls
    X, U, cost = orc.exact_quad_program(Xg[:, 0], X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, D_ls, u_prev, sat, du)
    lo, hi = -sat * np.ones((m, T)), sat * np.ones((m, T))
    lo[:, 0], hi[:, 0] = np.maximum(lo[:, 0], u_prev - du), np.minimum(hi[:, 0], u_prev + du)
    assert np.all(U >= lo - 1e-12) and np.all(U <= hi + 1e-12)
    lam = Q_ls[T] @ (X[:, T] - X_bm[:, T])
    g = np.zeros((m, T))
    for t in range(T - 1, -1, -1):
        g[:, t] = 2 * np.real(R_ls[t] @ (U[:, t] - U_bm[:, t]) + B_ls[t].conj().T @ lam)
        lam = Q_ls[t] @ (X[:, t] - X_bm[:, t]) + A_ls[t].conj().T @ lam
    at_lo, at_hi = U <= lo + 1e-10, U >= hi - 1e-10
    interior = ~(at_lo | at_hi)
    scale = np.abs(g).max()
    assert at_lo.sum() + at_hi.sum() > 3 and interior.sum() > 3
    assert np.abs(g[interior]).max() <= 1e-8 * scale
    assert np.all(g[at_lo] >= -1e-8 * scale) and np.all(g[at_hi] <= 1e-8 * scale)


# ---------------------------------------------------------------- closed-loop driver, pinned against the reference's mpc.py
# tests/golden/mpc_loop.npz: the reference's own mpc.py:101-125,128-304 run around its own lqr.quad_program (see
# tests/golden/make_golden.py).  Pins rows a1 / a13 / a14 / a19 of SURVEY.md section 8.
LOOPS_OK = ["qubit_o1", "qubit_o2_mf5", "qubit_o1_cold_cap4", "transmon_o1", "transmon_o2_mf2_cap5", "coupled_o1_cap6"]


def loop_case(g, name):
    k = "loop_" + name + "_"
    c = {key[len(k):]: g[key] for key in g.files if key.startswith(k)}
    for key in ("d", "m", "T", "n_steps", "order", "measure_freq", "max_iter", "exit_index"):
        c[key] = int(c[key])
    for key in ("dt", "sat", "du", "growth", "exit_thr"):
        c[key] = float(c[key])
    c["warm_start"] = bool(c["warm_start"])
    return c


def oracle_loop(c, model=None, **kw):
    """The oracle's loop (qp_mode "lqr" = the arithmetic of lqr.py) on a golden scenario."""
    n = c["d"] ** 2
    if model is None:
        model = orc.OracleDMDc(n, n, c["model"].shape[1] - n, c["model"])
    clock = orc.OracleClock(c["dt"], c["T"], c["n_steps"])
    clock.measure_freq = c["measure_freq"]
    H = c["H_plant"]
    if c["growth"]:
        exp = orc.OracleLExperiment(orc.liouvillian_ij(H[0]) + c["growth"] * np.identity(n),
                                    [orc.liouvillian_ij(h) for h in H[1:]])
    else:
        exp = orc.OracleQExperiment(H[0], list(H[1:]))
    cond = None
    if c["exit_index"] >= 0:
        cond = lambda xn, x, u: abs(xn[c["exit_index"]]) > c["exit_thr"]      # noqa: E731
    (xs, us), _, code = orc.mpc(c["x0"], c["m"], c["order"], c["X_targ"], c["U_targ"], clock, exp, model, c["Q"], c["R"],
                                c["Q"], sat=c["sat"], du=c["du"], max_iter=c["max_iter"], exit_condition=cond,
                                warm_start=c["warm_start"], qp_mode="lqr", **kw)
    return xs, us, code, clock


@pytest.mark.parametrize("d", [2, 3, 4])
@pytest.mark.parametrize("tag", ["diag", "dense"])
def test_iqp_line_search_vs_reference_mpc_py(golden, d, tag):
    g = golden("mpc_loop")
    k = "ls_d%d_%s_" % (d, tag)
    X, U = g[k + "X"], g[k + "U"]
    T = U.shape[2]
    Q_ls, R_ls = [g[k + "Q"]] * T + [g[k + "Qf"]], [g[k + "R"]] * T
    alpha, step = orc.iqp_line_search(Q_ls, R_ls, X[0], U[0], X[1], U[1], X[2], U[2])
    assert abs(alpha - float(g[k + "alpha"])) <= TOL * max(1.0, abs(float(g[k + "alpha"])))
    assert abs(step - float(g[k + "step"])) <= TOL * max(1.0, float(g[k + "step"]))


def test_shift_guess_and_clock_vs_reference_mpc_py(golden):
    g = golden("mpc_loop")
    assert np.array_equal(orc.shift_guess(g["shift_in"]), g["shift_out"])
    ck = orc.OracleClock(0.25, 7, 11)
    ck.measure_freq = 3
    assert np.array_equal(ck.ts, g["clock_ts"]) and np.array_equal(ck.ts_step(5), g["clock_ts_step5"])
    assert np.array_equal(ck.ts_horizon(4), g["clock_ts_horizon4"])
    ck.set_endsim(6)
    assert np.array_equal(ck.ts_sim, g["clock_ts_sim6"])


# tests/golden/mpc_loop_long.npz: the same harness at the BASELINE horizons - config 3 (T = 40, 20 steps, members 0 and 1 of the
# model ensemble) and config 5 (T = 80, 10 steps) - with the reference's OWN sensitivity per step (rerun with the linearisation
# point of the step's first QP solve scaled by 1 + 1e-15): sens_us / sens_xs.
LOOPS_LONG = ["transmon_o1_T40_m0", "transmon_o1_T40_m1", "transmon_o1_T80_m0"]


def _which(name):
    return "mpc_loop_long" if name in LOOPS_LONG else "mpc_loop"


@pytest.mark.parametrize("name", LOOPS_OK + LOOPS_LONG)
def test_mpc_loop_teacher_forced_vs_reference_mpc_py(golden, name):
    """Every MPC step of the reference's run, restarted from the REFERENCE's state (xs, us, SQP guess at the first QP
    solve of the step): the oracle must make the same number of QP solves with the same SQP iterates and produce the same
    us[k], xs[k+1].  Step by step the comparison is not polluted by the conditioning of the free-running loop.
    At the BASELINE horizons the bound of a step is widened by ten times what the REFERENCE itself moves under a 1e-15
    perturbation of that step's linearisation point (recorded in the fixture; at most 3e-13)."""
    c = loop_case(golden(_which(name)), name)
    steps, Xg, Ug = c["solve_step"], c["solve_Xg"], c["solve_Ug"]
    assert int(c["exit_code"]) == 0
    su = c.get("sens_us", np.zeros(c["n_steps"]))
    sx = c.get("sens_xs", np.zeros(c["n_steps"]))
    for k in range(c["n_steps"]):
        idx = np.nonzero(steps == k)[0]
        st = dict(step=k, xs=c["xs"], us=c["us"], X_guess=Xg[idx[0]], U_guess=Ug[idx[0]])
        tr = []
        xs, us, code, _ = oracle_loop(c, start=st, stop=k + 1, solve_trace=tr)
        assert code == 0 and len(tr) == len(idx), (k, len(tr), len(idx))
        for (s_k, X, U), i in zip(tr, idx):
            assert s_k == k
            assert np.abs(X - Xg[i]).max() <= 1e-10 and np.abs(U - Ug[i]).max() <= 1e-10, (k, i)
        assert np.abs(us[:, k] - c["us"][:, k]).max() <= 1e-11 + 10 * su[k], (k, np.abs(us[:, k] - c["us"][:, k]).max())
        assert np.abs(xs[:, k + 1] - c["xs"][:, k + 1]).max() <= 1e-11 + 10 * sx[k], k


@pytest.mark.parametrize("name", LOOPS_OK + LOOPS_LONG)
def test_mpc_loop_free_running_vs_reference_mpc_py(golden, name):
    """The whole run: same exit code, same returned shapes, same clock.ts_sim, same QP-solve schedule.  Steps 0 and 1
    (all their SQP iterations) to 1e-10; later steps within what the loop's conditioning allows (controls saturate, so
    differences in the last bits of a bang-bang switch grow) - bounded loosely here, tightly by the teacher-forced test."""
    c = loop_case(golden(_which(name)), name)
    tr = []
    xs, us, code, clock = oracle_loop(c, solve_trace=tr)
    assert code == int(c["exit_code"]) and xs.shape == c["xs"].shape and us.shape == c["us"].shape
    assert np.array_equal(clock.ts_sim, c["ts_sim"])
    assert np.array_equal(np.array([s for s, _, _ in tr]), c["solve_step"])
    assert np.abs(us[:, :2] - c["us"][:, :2]).max() <= 1e-10 and np.abs(xs[:, :3] - c["xs"][:, :3]).max() <= 1e-10
    if "env_us" in c:
        # BASELINE horizons: held to the REFERENCE's own free-running envelope (its run with x0 scaled by 1 +- 1e-14; at T = 40 the
        # reference's loop is chaotic from step 7 on: a bang-bang switch flips, 3.1 = 2 sat by step 13)
        assert np.all(np.abs(us - c["us"]).max(axis=0) <= 1e-9 + 100 * c["env_us"])
        assert np.all(np.abs(xs - c["xs"]).max(axis=0) <= 1e-9 + 100 * c["env_xs"])
    else:
        assert np.abs(us - c["us"]).max() <= 1e-6 * c["sat"] and np.abs(xs - c["xs"]).max() <= 1e-6


@pytest.mark.parametrize("name", ["qubit_o1_exit_step3", "qubit_o1_exit_step0"])
def test_mpc_loop_exit_condition_vs_reference_mpc_py(golden, name):
    """exit_condition firing (code 1) mid-run and at step 0: the last attempted entry is dropped, controls are None at
    step 0, clock.set_endsim(step) (mpc.py:289-304)."""
    c = loop_case(golden("mpc_loop"), name)
    xs, us, code, clock = oracle_loop(c)
    assert code == 1 == int(c["exit_code"])
    assert xs.shape == c["xs"].shape and np.abs(xs - c["xs"]).max() <= 1e-10
    if bool(c["us_is_none"]):
        assert us is None
    else:
        assert us.shape == c["us"].shape and np.abs(us - c["us"]).max() <= 1e-10
    assert np.array_equal(clock.ts_sim, c["ts_sim"])


def test_mpc_loop_streaming_vs_reference_mpc_py(golden):
    """streaming=True around the reference's OnlineDMDc with measure_freq = 2 (mpc.py:261-267, 281-285): the model object is
    refitted after every step and predicts the unmeasured states, while the loop keeps linearising the operators extracted at
    entry (fit_iteration rebinds model.A; the views WrapModel took at mpc.py:156 go stale).  The refit class used here is the
    package's host-side OnlineDMDc, itself pinned against the reference's by dmdc.npz."""
    from mpc4quantum_amd.model import OnlineDMDc
    c = loop_case(golden("mpc_loop"), "qubit_o1_streaming_mf2")
    assert bool(c["streaming"]) and c["measure_freq"] == 2
    n = c["d"] ** 2
    model = OnlineDMDc.from_bootstrap(n, n, c["model"].shape[1] - n, c["model"].copy(), alpha=float(c["alpha"]))
    tr = []
    xs, us, code, clock = oracle_loop(c, model=model, streaming=True, solve_trace=tr)
    assert code == int(c["exit_code"]) == 0 and xs.shape == c["xs"].shape and us.shape == c["us"].shape
    assert np.array_equal(np.array([s_k for s_k, _, _ in tr]), c["solve_step"])
    assert np.abs(us[:, :2] - c["us"][:, :2]).max() <= 1e-10 and np.abs(xs[:, :3] - c["xs"][:, :3]).max() <= 1e-10
    assert np.abs(us - c["us"]).max() <= 1e-6 * c["sat"] and np.abs(xs - c["xs"]).max() <= 1e-6
    assert np.abs(c["model_final"] - c["model"]).max() > 1e-3                      # the reference's model did move ...
    assert np.abs(model.A - c["model_final"]).max() <= 1e-6                          # ... and this one moved with it
    # the refit steers nothing but the predictions: the same run without streaming applies the same controls at measured steps
    xs2, us2, _, _ = oracle_loop(c)
    assert np.abs(us2[:, 0] - us[:, 0]).max() <= 1e-12 and np.abs(xs2[:, 1] - xs[:, 1]).max() <= 1e-12


def test_mpc_loop_infinite_objective_vs_reference_mpc_py(golden):
    """Exit code 3 (mpc.py:200-203): a plant that amplifies the state until x^H Q x overflows at MPC step 6."""
    c = loop_case(golden("mpc_loop"), "qubit_o1_inf_later")
    xs, us, code, clock = oracle_loop(c)
    assert code == 3 == int(c["exit_code"])
    assert xs.shape == c["xs"].shape == (4, 7) and us.shape == c["us"].shape
    assert np.abs(xs / c["xs"] - 1)[np.abs(c["xs"]) > 0].max() <= 1e-9
    assert np.array_equal(clock.ts_sim, c["ts_sim"])


def test_mpc_loop_nan_raises_like_the_reference(golden):
    """NaN (or 1e200) in x0: the reference does not return exit code 3 - mpc.py:200 tests np.isinf only, a NaN objective
    passes, and numpy.linalg.pinv raises LinAlgError at lqr.py:61 (1e200: at the first solve, B^H V B overflows).  The
    oracle follows.  The batched HIP engine cannot raise for one ensemble member: it ends that member with exit code 3
    (tests/test_gpu_parity.py::test_mpc_loop_nonfinite_vs_reference_mpc_py)."""
    g = golden("mpc_loop")
    for name in ("qubit_o1_nan", "qubit_o1_inf_step0"):
        c = loop_case(g, name)
        assert str(c["raised"]) == "LinAlgError" and int(c["exit_code"]) == -1
        with pytest.raises(np.linalg.LinAlgError):
            oracle_loop(c)


@pytest.mark.parametrize("d,m", [(2, 1), (3, 2), (4, 3)])
def test_plant_step_is_the_solution_of_the_ode_mesolve_integrates(d, m):
    """The reference's plant is qutip.mesolve on H = H0 + sum_k u_k(t) H_k (experiment.py:202-212) with the control held over each
    step (interp1d kind='previous', mpc.py:258): d rho/dt = -i [H, rho], rho(0) = x0.reshape(d, d), output flattened row-major.
    qutip is absent; an independent integrator of that very ODE (SciPy DOP853, rtol 1e-12) must land on the oracle's exact
    propagator - which is what the HIP plant is held to."""
    from scipy.integrate import solve_ivp
    rng = np.random.default_rng(60 + d)

    def herm():
        M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        return M + M.conj().T
    H0, Hk = herm(), [herm() for _ in range(m)]
    M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho0 = M @ M.conj().T
    rho0 /= np.trace(rho0).real
    u = rng.uniform(-1, 1, m)
    dt = 0.37
    H = H0 + sum(uk * h for uk, h in zip(u, Hk))

    def rhs(t, y):
        rho = y.reshape(d, d)
        return (-1j * (H @ rho - rho @ H)).reshape(-1)
    sol = solve_ivp(rhs, (0.0, dt), rho0.reshape(-1).astype(complex), method="DOP853", rtol=1e-12, atol=1e-14)
    assert np.abs(orc.plant_step(rho0.reshape(-1), u, H0, Hk, dt) - sol.y[:, -1]).max() <= 1e-10
