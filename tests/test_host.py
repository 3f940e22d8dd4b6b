"""CPU tests (-m "not gpu"): host-side logic of the product against the oracle and the reference-made
goldens, the C ABI surface, and loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

import mpc4quantum_amd as m4q
from mpc4quantum_amd import _lib
from mpc4quantum_amd.distributed import shard_bounds
from oracle import m4q_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("m", [1, 2, 3])
def test_library_tables_vs_reference_golden(golden, order, m):
    g = golden("library_tables")
    key = "o%d_m%d" % (order, m)
    assert np.array_equal(np.vstack(m4q.create_power_list(order, m)), g[key + "_powers"])
    assert m4q.size_of_library(order, m) == int(g[key + "_size"])
    u = g[key + "_u"]
    assert np.allclose(np.vstack([f(u) for f in m4q.create_library(order, m)]), g[key + "_lib"], rtol=0, atol=1e-12)
    fns, coefs = m4q.diff_library(order, m)
    assert np.array_equal(np.stack([c.reshape(-1) for c in coefs]), g[key + "_dcoef"])
    assert np.allclose(np.stack([np.vstack([f(u) for f in fl]) for fl in fns]), g[key + "_dlib"], rtol=0, atol=1e-12)
    assert [tuple(p) for p in m4q.multinomial_powers(order, m + 1)] == \
           [tuple(int(v) for v in p) for p in orc.multinomial_powers(order, m + 1)]


@pytest.mark.parametrize("order,m", [(1, 1), (2, 1), (1, 2), (2, 2), (1, 3)])
def test_device_power_table_matches_host(order, m):
    """The compile-time table inside the kernels (csrc/m4q_mpc.h PowTab) == linearize.create_power_list."""
    P = m4q.size_of_library(order, m)
    out = np.zeros(P * m, dtype=np.int32)
    got = _lib.lib().m4q_power_list(order, m, out.ctypes.data_as(_lib._ip))
    assert got == P == _lib.lib().m4q_library_size(order, m) + 1
    assert np.array_equal(out.reshape(P, m), np.vstack(m4q.create_power_list(order, m)))


def test_krtimes(golden):
    g = golden("library_tables")
    assert np.allclose(m4q.krtimes(g["kr_a"], g["kr_b"]), g["kr_out"], rtol=0, atol=1e-13)
    with pytest.raises(ValueError):
        m4q.krtimes(np.ones((2, 3)), np.ones((2, 4)))


@pytest.mark.parametrize("name,order", [("qubit", 1), ("qubit", 2), ("transmon", 1), ("transmon", 2), ("coupled", 1)])
def test_discretize_vs_reference_golden(golden, name, order):
    g = golden("discretize")
    out = m4q.discretize_homogeneous(list(g[name + "_A_cts"]), float(g[name + "_dt"]), order)
    assert np.abs(out - g["%s_o%d" % (name, order)]).max() <= 1e-12


def test_discretize_known_answer_and_batched():
    """reference tests/test_mpc4quantum.py:147-188: order 1, dt 1 -> [I + A | N_1 | N_2]."""
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    sy = np.array([[0, -1j], [1j, 0]], dtype=complex)
    basis = [np.outer(np.eye(2)[i], np.eye(2)[j]) for i in range(2) for j in range(2)]
    l1 = [m4q.vectorize_me(op, basis) for op in (0 * sx, sx)]
    l2 = [m4q.vectorize_me(op, basis) for op in (0 * sy, sy)]
    z = np.zeros((4, 4))
    ops = [np.block([[l1[0], z], [z, l2[0]]]), np.block([[l1[1], z], [z, z]]), np.block([[z, z], [z, l2[1]]])]
    out = m4q.discretize_homogeneous(ops, 1, 1)
    assert np.isclose(out, np.hstack([ops[0] + np.identity(8), ops[1], ops[2]])).all()
    assert np.abs(m4q.vectorize_me(sx + 0.3 * sy, basis) - m4q.liouvillian(sx + 0.3 * sy)).max() < 1e-14
    stacked = m4q.discretize_homogeneous([np.stack([o, 2 * o]) for o in ops], 0.5, 2)
    assert np.abs(stacked[1] - orc.discretize_homogeneous([2 * o for o in ops], 0.5, 2)).max() < 1e-12


def test_clock_and_small_helpers():
    c, o = m4q.StepClock(0.25, 16, 20), orc.OracleClock(0.25, 16, 20)
    for k in (0, 3, 19):
        assert np.array_equal(c.ts_step(k), o.ts_step(k)) and np.array_equal(c.ts_horizon(k), o.ts_horizon(k))
    assert np.array_equal(c.ts, o.ts)
    c.set_endsim(7)
    assert len(c.ts_sim) == 7
    assert m4q.val_to_str(1e-2) == "1d0em02" and m4q.val_to_str(0.25) == "2d5em01"
    assert c.to_string() == "mf_1d0e00_dt_2d5em01_h_1d6e01_n_2d0e01"
    a = np.arange(12).reshape(3, 4)
    assert np.array_equal(m4q.shift_guess(a), orc.shift_guess(a))
    assert np.array_equal(m4q.shift_guess(a)[:, -1], a[:, -1]) and np.array_equal(m4q.shift_guess(a)[:, 0], a[:, 1])


def test_line_search_host_form_vs_oracle():
    rng = np.random.default_rng(4)
    n, m, T = 4, 1, 5
    Q = np.diag([1.0, 0, 0, 1.0])
    Q_ls, R_ls = [Q] * T + [2 * Q], [0.3 * np.eye(m)] * T
    Xt, Xg, Xo = (rng.standard_normal((n, T + 1)) + 1j * rng.standard_normal((n, T + 1)) for _ in range(3))
    Ut, Ug, Uo = (rng.standard_normal((m, T)) for _ in range(3))
    a1, s1, _, _ = m4q.iqp_line_search(Q_ls, R_ls, Xt, Ut, Xg, Ug, Xo, Uo)
    a2, s2 = orc.iqp_line_search(Q_ls, R_ls, Xt, Ut, Xg, Ug, Xo, Uo)
    assert abs(a1 - a2) < 1e-12 and abs(s1 - s2) < 1e-12


@pytest.mark.parametrize("d", [2, 3, 4])
@pytest.mark.parametrize("tag", ["diag", "dense"])
def test_line_search_host_form_vs_reference_mpc_py(golden, d, tag):
    """The package's host iqp_line_search against the reference's own (mpc.py:101-125): all four returns."""
    g = golden("mpc_loop")
    k = "ls_d%d_%s_" % (d, tag)
    X, U = g[k + "X"], g[k + "U"]
    T = U.shape[2]
    Q_ls, R_ls = [g[k + "Q"]] * T + [g[k + "Qf"]], [g[k + "R"]] * T
    alpha, step, fval, slope = m4q.iqp_line_search(Q_ls, R_ls, X[0], U[0], X[1], U[1], X[2], U[2])
    for got, key in ((alpha, "alpha"), (step, "step"), (fval, "fval"), (slope, "slope")):
        ref = g[k + key]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), key


def test_clock_and_shift_guess_vs_reference_mpc_py(golden):
    g = golden("mpc_loop")
    assert np.array_equal(m4q.shift_guess(g["shift_in"]), g["shift_out"])
    ck = m4q.StepClock(0.25, 7, 11)
    ck.measure_freq = 3
    assert np.array_equal(ck.ts, g["clock_ts"]) and np.array_equal(ck.ts_step(5), g["clock_ts_step5"])
    assert np.array_equal(ck.ts_horizon(4), g["clock_ts_horizon4"])
    ck.set_endsim(6)
    assert np.array_equal(ck.ts_sim, g["clock_ts_sim6"])
    assert ck.to_string() == str(g["clock_string"])


@pytest.mark.parametrize("name,order", [("qubit", 1), ("qubit", 2), ("transmon", 1), ("transmon", 2), ("coupled", 1)])
def test_wrapmodel_f_and_lift_u_vs_reference_golden(golden, name, order):
    """Product WrapModel.f / lift_u / DMDc.predict (host NumPy, no GPU) against the reference's own outputs."""
    g = golden("linearize")
    key = "%s_o%d" % (name, order)
    m = {"qubit": 1, "transmon": 2, "coupled": 3}[name]
    model = g[key + "_model"]
    n = model.shape[0]
    dm = m4q.DMDc(n, n, model.shape[1] - n, model)
    wm = m4q.WrapModel(*dm.get_discrete(), m, order)
    xs, us = g[key + "_xs"], g[key + "_us"]
    f = np.hstack([np.reshape(wm.f(xs[:, i], us[:, i], 0), (n, 1)) for i in range(us.shape[1])])
    assert np.abs(f - g[key + "_f"]).max() <= 1e-13
    assert np.abs(wm.lift_u(us) - g[key + "_liftu"]).max() <= 1e-13
    ux = m4q.krtimes(wm.lift_u(us[:, :1]), xs[:, :1])
    assert np.abs(dm.predict(xs[:, :1], ux) - g[key + "_predict"]).max() <= 1e-13


def test_dmdc_container():
    A = np.arange(24).reshape(4, 6) + 1j
    d = m4q.DMDc(4, 4, 2, A)
    ax, au = d.get_discrete()
    assert ax.shape == (4, 4) and au.shape == (4, 2)
    x, u = np.ones(4), np.ones(2)
    assert np.allclose(d.predict(x, u), (ax @ x + au @ u).reshape(4, 1))


def test_wrapmodel_dimension_check_needs_no_gpu():
    with pytest.raises(ValueError):
        m4q.WrapModel(np.eye(4), np.zeros((4, 12)), 1, 1)
    wm = m4q.WrapModel(np.eye(4), np.zeros((4, 8)), 1, 2)
    assert wm.lift_u(np.array([[2.0]])).reshape(-1).tolist() == [2.0, 4.0]


def test_partial_trace_lift_proj_known_answer():
    """reference tests/test_mpc4quantum.py:190-213: lift(proj) round-trips product states and not entangled ones."""
    rng = np.random.default_rng(8)

    def rand_dm(d):
        M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        r = M @ M.conj().T
        return r / np.trace(r).real
    for nd in (2, 4):
        a, b = rand_dm(nd), rand_dm(nd)
        c = np.kron(a, b)
        ab = m4q.QCoupledExperiment.lift(c.flatten())
        assert np.isclose(np.hstack([a.flatten(), b.flatten()]), ab).all()
        assert np.isclose(ab, orc.OracleQCoupledExperiment.lift(c.flatten())).all()
        assert np.isclose(m4q.QCoupledExperiment.proj(ab).reshape(nd * nd, nd * nd), c).all()
        d = rand_dm(nd * nd)
        back = m4q.QCoupledExperiment.proj(m4q.QCoupledExperiment.lift(d.flatten())).reshape(nd * nd, nd * nd)
        assert not np.isclose(d, back).all()
    r3 = rand_dm(3)
    l2 = m4q.QExperiment32.lift(r3.flatten())
    assert l2.shape == (4,) and abs(l2[0] + l2[3] - 1) < 1e-14
    assert np.allclose(l2.reshape(2, 2), r3[:2, :2] / np.trace(r3[:2, :2]))
    # mpc.py:82-98 real/complex packing helpers and experiment.py:309-315 split_blocks
    z = np.array([1 + 2j, 3 - 1j, -0.5j])
    assert np.array_equal(m4q.real_to_complex(m4q.complex_to_real(z)), z)
    P = np.arange(9).reshape(3, 3) * (1 - 0.5j) + np.eye(3) * 2j
    assert np.array_equal(m4q.real_to_complex_op(m4q.complex_to_real_op(P)), P)
    assert np.allclose(m4q.complex_to_real_op(P) @ m4q.complex_to_real(z), m4q.complex_to_real(P @ z))
    blocks = m4q.split_blocks(np.arange(24).reshape(4, 6), 2, 3)
    assert blocks.shape == (4, 2, 3)
    assert np.array_equal(blocks[1], [[3, 4, 5], [9, 10, 11]]) and np.array_equal(blocks[2], [[12, 13, 14], [18, 19, 20]])
    with pytest.warns(UserWarning):
        m4q.isinf_warning()
    assert m4q.isqrt(16) == 4 and m4q.isqrt(17) == 4 and m4q.isqrt(0) == 0
    with pytest.raises(ValueError):
        m4q.isqrt(-1)


def test_shard_bounds_cover_everything():
    for B in (1, 7, 64, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


# ---------------------------------------------------------------- C ABI surface
@pytest.mark.parametrize("final_only", [False, True])
def test_result_layout_roundtrip_and_alignment(final_only):
    """The byte layout of the multi-GPU gather buffer: regions 16-byte aligned and disjoint, pack/unpack inverse of each
    other for a block shorter than the buffer."""
    from mpc4quantum_amd.distributed import ResultLayout
    rng = np.random.default_rng(3)
    rows, k, n, m, ns = 7, 5, 9, 2, 6
    lay = ResultLayout(rows, n, m, ns, final_only)
    ends = 0
    for f in ResultLayout.FIELDS:
        assert lay.offset[f] % 16 == 0 and lay.offset[f] >= ends
        ends = lay.offset[f] + lay.field_bytes(f)
    assert lay.nbytes >= ends
    res = {"xs": rng.standard_normal((k, ns + 1, n)) + 1j * rng.standard_normal((k, ns + 1, n)),
           "us": rng.standard_normal((k, ns, m)), "exit_codes": rng.integers(0, 4, k).astype(np.int32),
           "steps_done": rng.integers(0, ns + 1, k).astype(np.int32), "qp_solves": rng.integers(0, 100, (k, ns)).astype(np.int32)}
    buf = np.zeros(lay.nbytes, dtype=np.uint8)
    lay.pack(res, buf)
    back = lay.unpack(buf, k)
    assert np.array_equal(back["xs"], res["xs"][:, -1:] if final_only else res["xs"])
    for f in ResultLayout.FIELDS[1:]:
        assert np.array_equal(back[f], res[f]) and back[f].dtype == res[f].dtype
    assert not lay.view(buf, "us")[k:].any()                    # padding rows stay zero


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "m4q.h")).read()
    declared = set(re.findall(r"M4Q_API[^;(]*?\b(m4q_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 24
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert _lib.lib().m4q_version().startswith(b"m4q-hip")


def test_problem_struct_layout_matches_header():
    header = open(os.path.join(ROOT, "include", "m4q.h")).read()
    body = header[header.index("typedef struct m4q_problem {"):header.index("} m4q_problem;")]
    fields = re.findall(r"^\s*(int32_t|double)\s+(\w+);", body, re.M)
    assert [f for _, f in fields] == [f for f, _ in _lib.Problem._fields_]
    assert ctypes.sizeof(_lib.Problem) == 16 * 4 + 4 * 8


def test_supported_shapes():
    assert _lib.supported(4, 1, 1) and _lib.supported(9, 2, 1) and _lib.supported(9, 2, 2) and _lib.supported(16, 3, 1)
    assert _lib.supported(8, 2, 1) and _lib.supported(4, 2, 1) and not _lib.supported(25, 1, 1)


def test_fails_loudly_without_a_gpu():
    """The product has no CPU path: on a box with no device every compute entry point raises."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    wm = m4q.WrapModel(np.eye(4), np.zeros((4, 4)), 1, 1)
    with pytest.raises(_lib.M4qError):
        wm.get_model_along_traj(np.zeros((4, 3), dtype=complex), np.zeros((1, 3)), np.arange(3))
    with pytest.raises(_lib.M4qError):
        m4q.EnsembleSession(4, 4, 1, 1, 5, 3, 1.0, 0.5)
    with pytest.raises(_lib.M4qError):
        m4q.plant_step_batch(np.zeros((1, 4)), np.zeros((1, 1)), np.eye(2), np.eye(2)[None], 0.1)
    with pytest.raises(TypeError):
        m4q.EnsembleSession(4, 4, 1, 1, 5, 3, 1.0, None)


@pytest.mark.gpu
def test_missing_generator_library_is_an_error_not_a_crash():
    """A generator-plant session needs libm4q_hip_gen.so (loaded on first use, before any device call): when it does not load, the
    session is refused with the loader's reason.  (Round 4: the message was built from a second dlerror() call, which returns NULL.)"""
    import subprocess
    import sys
    code = ("import mpc4quantum_amd as m4q\nfrom mpc4quantum_amd import _lib\n"
            "try:\n    m4q.EnsembleSession(4, 9, 2, 1, 5, 3, 1.0, 0.5, plant_kind=_lib.PLANT_GENERATOR)\n"
            "except _lib.M4qError as e:\n    print('refused:', e)\n")
    env = dict(os.environ, M4Q_GEN_LIB="/nonexistent/libm4q_hip_gen.so")
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=120)
    assert res.returncode == 0, res.stderr.decode()[-400:]
    out = res.stdout.decode()
    assert "refused:" in out and "libm4q_hip_gen.so" in out and "/nonexistent" in out


def test_streaming_dmdc_refits_vs_reference_golden(golden):
    """DiscrepDMDc / OnlineDMDc (model.py:109-313), the models mpc(..., streaming=True) refits through fit_iteration:
    batch fits and the model after each of eight streaming updates against the reference's own classes."""
    g = golden("dmdc")
    X, U, Y, Xs, Us, Ys, A_boot = (g[k] for k in ("X", "U", "Y", "Xs", "Us", "Ys", "A_boot"))
    n, k = X.shape[0], U.shape[0]

    def close(a, b):
        return np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    d = m4q.DiscrepDMDc.from_data(Y, X, U, rcond=1e-12)
    assert (d.dim_y, d.dim_x, d.dim_u) == (n, n, k) and close(d.A, g["discrep_from_data_A"])
    d = m4q.DiscrepDMDc.from_bootstrap(n, n, k, A_boot.copy())
    d.discount = 0.9
    d.append(Y[:, :6], X[:, :6], U[:, :6])
    for i in range(Xs.shape[1]):
        A_x, A_u = d.fit_iteration(Ys[:, i], Xs[:, i], Us[:, i])
        assert close(np.hstack([A_x, A_u]), g["discrep_stream_A"][i])
    assert close(d.Y, g["discrep_stream_Y"])
    o = m4q.OnlineDMDc.from_data(Y, X, U)
    assert close(o.A, g["online_from_data_A"]) and close(o.P, g["online_from_data_P"])
    o = m4q.OnlineDMDc.from_bootstrap(n, n, k, A_boot.copy(), alpha=1e2)
    o.discount = 0.95
    for i in range(Xs.shape[1]):
        o.fit_iteration(Ys[:, i], Xs[:, i], Us[:, i])
        assert close(o.A, g["online_stream_A"][i]) and close(o.P, g["online_stream_P"][i])
    assert close(o.predict(Xs, Us), g["predict"])
    # the refit pulls the bootstrapped model towards the data
    A_true_fit = g["discrep_from_data_A"]
    assert np.abs(o.A - A_true_fit).max() < 0.5 * np.abs(A_boot - A_true_fit).max()
    # saved history and the no-control form
    o._save, o._isave = True, 1
    o.fit_iteration(Ys[:, 0], Xs[:, 0], Us[:, 0])
    assert len(o.iA) == 2 and len(o.iP) == 2
    d0 = m4q.DiscrepDMDc.from_data(Y, X, None, rcond=1e-12)
    assert d0.dim_u == 0 and d0.A.shape == (n, n)
    with pytest.raises(NotImplementedError):
        m4q.DMDc(n, n, k, A_boot).fit_iteration(Ys[:, 0], Xs[:, 0], Us[:, 0])


def test_qexperiment_set_collapse_operators_gives_the_lindblad_generator():
    """QExperiment.set (experiment.py:196-200): 'c_ops' turns the plant into the Lindblad generator on vec_r(rho); checked
    against the master equation written out, trace preservation, and the Hamiltonian case (no c_ops) left untouched."""
    from mpc4quantum_amd import _lib
    from mpc4quantum_amd.experiment import QExperiment
    rng = np.random.default_rng(11)
    d = 3
    def herm():
        a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        return a + a.conj().T
    H0, H1 = herm(), [herm(), herm()]
    exp = QExperiment(H0, H1)
    assert exp.plant_kind == _lib.PLANT_HAMILTONIAN and exp.operators()[0].shape == (d, d)
    cs = [0.3 * (rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))) for _ in range(2)]
    exp.set("c_ops", cs)
    exp.set("options", object())                              # integrator tuning: kept, unused
    assert exp.plant_kind == _lib.PLANT_GENERATOR
    L0, Lk = exp.operators()
    assert L0.shape == (d * d, d * d) and Lk.shape == (2, d * d, d * d)
    a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = a @ a.conj().T
    rho /= np.trace(rho)
    u = np.array([0.4, -0.7])
    H = H0 + u[0] * H1[0] + u[1] * H1[1]
    want = -1j * (H @ rho - rho @ H)
    for c in cs:
        cc = c.conj().T @ c
        want = want + c @ rho @ c.conj().T - 0.5 * (cc @ rho + rho @ cc)
    got = exp.f(0.0, rho.flatten(), u)
    assert np.abs(got - want.flatten()).max() <= 1e-13
    assert np.abs(np.identity(d).flatten() @ (L0 + u[0] * Lk[0] + u[1] * Lk[1])).max() <= 1e-13      # d/dt tr(rho) = 0
    exp.set("c_ops", [])
    assert exp.plant_kind == _lib.PLANT_HAMILTONIAN
