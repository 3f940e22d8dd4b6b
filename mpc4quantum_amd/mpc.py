"""Receding-horizon driver with the reference's interface (mpc4quantum/mpc.py): StepClock, mpc(),
shift_guess, iqp_line_search, val_to_str, plus the batched mpc_batch().

The loop body of the reference (mpc.py:161-292: linearise along the guess, solve the horizon QP,
line-search, apply U_opt[:,0], propagate the plant, shift) runs inside the persistent HIP kernel
`mpc_kernel` (csrc/m4q_kernels.hip) for every ensemble member at once.  Quirks kept on purpose:
u_prev from U_ref at steps 0 and 1 (:185), applied control U_opt[:,0] (:250), the one-step lag of
the target window (:276-277), exit codes and the dropped last entry (:294-304)."""
import numpy as np

from . import _lib
from .experiment import Experiment, QExperiment
from .library import krtimes
from .linearize import WrapModel
from .session import EnsembleSession


class StepClock:
    """mpc.py:14-35."""

    def __init__(self, dt, horizon, n_steps):
        self.dt = float(dt)
        self.horizon = horizon
        self.n_steps = n_steps
        self.measure_freq = 1
        self.ts = np.linspace(0, self.dt * self.n_steps, self.n_steps, endpoint=False)
        self.ts_sim = self.ts

    def set_endsim(self, index):
        self.ts_sim = self.ts[:index]

    def ts_step(self, a_step):
        return np.linspace(self.dt * (a_step + 1 - self.measure_freq), self.dt * (a_step + 1), self.measure_freq + 1)

    def ts_horizon(self, a_step):
        return np.linspace(self.dt * a_step, self.dt * (a_step + self.horizon), self.horizon, endpoint=False)

    def to_string(self):
        parts = ['mf', val_to_str(self.measure_freq), 'dt', val_to_str(self.dt), 'h', val_to_str(self.horizon), 'n',
                 val_to_str(self.n_steps)]
        return '_'.join(parts)


def val_to_str(val):
    """mpc.py:64-68: 1.0E-02 -> 1d0em02."""
    return f'{val:.1E}'.replace('E', 'e').replace('.', 'd').replace('-', 'm').replace('+', '')


def shift_guess(data):
    """mpc.py:71-73."""
    data = np.asarray(data)
    return np.hstack([data[:, 1:], data[:, -1:]])


def isinf_warning():
    """mpc.py:76-79: what the reference prints on exit code 3."""
    import warnings
    warnings.warn("Solution was infinite (failed to converge). Inspect the model for accuracy, check if control constraints "
                  "can regularize the problem, or run with verbose=True for more information.")


def solver_warning():
    """Exit code 2 (mpc.py:183-197: a solver warning ends the run).  Without OSQP the one solver that can give up is the exact
    box-QP iteration (exact_qp=True) stopping at its iteration cap."""
    import warnings
    warnings.warn("The exact box-QP solve stopped at its iteration cap before reaching the KKT point (exit code 2); "
                  "the horizon QP is too ill-conditioned for fp64 - shorten the horizon or use the clipped solve.")


def complex_to_real(z):
    """mpc.py:87-89: complex vector of length n -> [Re; Im] of length 2n."""
    return np.concatenate((np.real(z), np.imag(z)))


def real_to_complex(z):
    """mpc.py:82-84: the inverse, [Re; Im] -> complex."""
    z = np.asarray(z)
    half = len(z) // 2
    return z[:half] + 1j * z[half:]


def real_to_complex_op(P):
    """mpc.py:96-98: inverse of complex_to_real_op, read off the left block column."""
    P = np.asarray(P)
    r, c = P.shape[0] // 2, P.shape[1] // 2
    return P[:r, :c] + 1j * P[r:, :c]


def complex_to_real_op(P):
    P = np.asarray(P)
    return np.block([[P.real, -P.imag], [P.imag, P.real]])


def iqp_line_search(Q_ls, R_ls, X_htarg, U_htarg, X_guess, U_guess, X_opt, U_opt):
    """Host form of the line search the kernel performs (mpc.py:101-125), same return arity:
    (alpha, new_step, new_fval, new_slope).  Z is stacked and the cost blocks are laid out exactly as
    the reference does (see csrc/m4q_mpc.h line_search)."""
    def pack(X, U):
        xf, uf = np.asarray(X, dtype=complex).flatten(), np.asarray(U, dtype=complex).flatten()
        return np.concatenate((complex_to_real(xf), complex_to_real(uf)))
    Zt, Zg, Zo = pack(X_htarg, U_htarg), pack(X_guess, U_guess), pack(X_opt, U_opt)
    blocks = [complex_to_real_op(q) for q in Q_ls] + [complex_to_real_op(r) for r in R_ls]

    def apply(v):
        out = np.empty_like(v)
        pos = 0
        for b in blocks:
            k = b.shape[0]
            out[pos:pos + k] = 0.5 * (b + b.T) @ v[pos:pos + k]
            pos += k
        return out
    DZ = Zo - Zg
    alpha = -apply(Zg - Zt).dot(DZ) / (DZ @ apply(DZ))
    Zn = Zg + alpha * DZ
    new_fval = 0.5 * (Zn - Zt) @ apply(Zn - Zt)
    return alpha, np.linalg.norm(alpha * DZ), new_fval, apply(Zn - Zt)


def _native_plant(experiment):
    """True if the closed loop can stay on the GPU: one of this package's plants with identity lift/proj."""
    return (isinstance(experiment, QExperiment) and type(experiment).lift is Experiment.lift
            and type(experiment).proj is Experiment.proj and not getattr(experiment, "_sigma", 0)
            and getattr(experiment, "_me_args", {}).get("e_ops") is None)


def _trim(xs, us, code, done):
    """mpc.py:294-304: normal exit keeps done+1 states and done controls; an early exit drops the attempted entry."""
    if code == 0:
        return [xs[:, :done + 1], us[:, :done]]
    return [xs[:, :done + 1], us[:, :done] if done > 0 else None]


def mpc(x0, dim_u, order, X_targ, U_targ, clock, experiment, model, Q, R, Qf, sat=None, du=None, max_iter=100,
        exit_condition=None, streaming=False, warm_start=True, progress_bar=True, verbose=False, exact_qp=False,
        qp_flags=None):
    """Drop-in for mpc4quantum.mpc.mpc (mpc.py:128-304): returns ([xs, us], model, exit_code).
    exact_qp (extension): solve each QP to the box-constrained optimum, as the reference's OSQP call does, instead of
    clipping the Riccati rollout (identical whenever no bound is active).
    qp_flags (extension): M4Q_QP_* bits; _lib.QP_REF_LQR runs the loop around the arithmetic of the reference's lqr.py as
    written, which is what tests/golden/mpc_loop.npz (the reference's own mpc.py around its own lqr.py) pins."""
    mf = int(clock.measure_freq)
    x0 = np.asarray(x0, dtype=np.complex128).reshape(-1)
    lift_x0 = np.asarray(experiment.lift(x0), dtype=np.complex128).reshape(-1)
    A_x, A_u = model.get_discrete()
    wrapped = WrapModel(A_x, A_u, dim_u, order)          # validates the library size like mpc.py:156
    n = wrapped.dim_x
    T, ns = clock.horizon, clock.n_steps
    X_targ = np.atleast_2d(np.asarray(X_targ))
    U_targ = np.atleast_2d(np.asarray(U_targ))
    cols = min(X_targ.shape[1], ns + T + 1)
    fused = _native_plant(experiment) and exit_condition is None and not streaming
    kind = experiment.plant_kind if fused else _lib.PLANT_NONE
    sess = EnsembleSession(1, n, dim_u, order, T, ns, clock.dt, sat, du, max_iter, warm_start, qp_flags=qp_flags,
                           plant_kind=kind, target_cols=cols, measure_freq=mf, exact_qp=exact_qp)
    try:
        op0, ops = experiment.operators() if fused else (None, None)
        sess.load_problem(np.hstack([A_x, A_u])[None], lift_x0[None], X_targ, U_targ, Q, R, Qf, op0, ops)
        if fused:
            sess.run(0, ns)
            res = sess.results()
            code, done = int(res["exit_codes"][0]), int(res["steps_done"][0])
            if code == 3:
                isinf_warning()                                                                # mpc.py:200-203
            if code == 2:
                solver_warning()                                                               # mpc.py:193-196
            clock.set_endsim(done)
            return _trim(res["xs"][0].T, res["us"][0].T, code, done), model, code
        # host plant: one launch per MPC step, the plant (and lift/proj) evaluated by the caller's object
        xs = [x0]
        us = []
        code = 0
        step = 0
        it = range(ns)
        if progress_bar:
            try:
                from tqdm.auto import tqdm
                it = tqdm(it)
            except ImportError:
                pass
        for step in it:
            sess.run(step, step + 1)
            sess.sync()
            dev_code = int(sess.download(_lib.F_CODES, (1,))[0])
            if dev_code:
                code = dev_code
                if code == 3:
                    isinf_warning()
                if code == 2:
                    solver_warning()
                break
            u = sess.download(_lib.F_US, (1, ns, dim_u))[0, step]
            us.append(u)
            if (step + 1) % mf == 0:
                # measure: plant from the last measured state; controls stacked newest first as mpc.py:257 does
                ts_step = clock.ts_step(step)
                us_step = np.vstack([us[step - jq] for jq in range(mf)] + [us[step]]).T
                result = experiment.simulate(xs[step + 1 - mf], ts_step, _HeldControl(ts_step, us_step))   # mpc.py:256-260
                xs.append(np.asarray(result)[:, -1])
            else:
                lx = np.asarray(experiment.lift(xs[step])).reshape(-1, 1)                      # mpc.py:261-267
                lu = wrapped.lift_u(u.reshape(-1, 1))
                xs.append(np.asarray(experiment.proj(model.predict(lx, krtimes(lu, lx)))).flatten())
            sess.put_state(step + 1, np.asarray(experiment.lift(xs[step + 1]), dtype=np.complex128).reshape(1, -1))
            if streaming:                                                                      # mpc.py:281-285
                lu = wrapped.lift_u(u.reshape(-1, 1))
                lx = np.asarray(experiment.lift(xs[step])).reshape(-1, 1)
                model.fit_iteration(np.asarray(experiment.lift(xs[step + 1])).reshape(-1, 1), lx, krtimes(lu, lx))
            if exit_condition is not None and exit_condition(xs[step + 1], xs[step], us[step]):
                code = 1
                break
        if code == 0:
            clock.set_endsim(step + 1)
            return [np.vstack(xs[:step + 2]).T, np.vstack(us[:step + 1]).T], model, code
        clock.set_endsim(step)
        if step == 0:
            return [np.vstack(xs[:1]).T, None], model, code
        return [np.vstack(xs[:step + 1]).T, np.vstack(us[:step]).T], model, code
    finally:
        sess.close()


class _HeldControl:
    """interp1d(ts, us, kind='previous', fill_value='extrapolate') for the held control of one step."""

    def __init__(self, ts, us):
        self.x = np.asarray(ts)
        self.y = np.asarray(us)

    def __call__(self, t):
        idx = np.clip(np.searchsorted(self.x, t, side='right') - 1, 0, len(self.x) - 1)
        return self.y[..., idx]


def open_session(x0, models, dim_u, order, X_targ, U_targ, clock, plant_op0, plant_ops, Q, R, Qf, sat, du=None,
                 max_iter=100, warm_start=True, qp_flags=None, plant_kind=_lib.PLANT_HAMILTONIAN, device=-1,
                 force_complex=False, exact_qp=False, traceless=True, tile=None, generators=None, scales=None,
                 shared_generators=None):
    """An EnsembleSession loaded with mpc_batch's arguments (everything resident in HBM, nothing run yet).
    models = None with generators [1+m, n, n] (or [B, 1+m, n, n]) and optional scales [B, 1+m]: the members' models are built on the
    device (discretize_homogeneous of the scaled generators, vectorize.py:8-49), and a set of SHARED generators at order 1 lets the
    closed loop run on them directly where that kernel exists (d = 4; EnsembleSession(shared_generators=...))."""
    x0 = np.ascontiguousarray(x0, dtype=np.complex128)
    Bn, n = x0.shape
    if models is None:
        if generators is None:
            raise TypeError("models is None: pass generators (and scales) to have the models built on the device")
        per_model = True
    else:
        models = np.asarray(models, dtype=np.complex128)
        if models.ndim == 2:
            models = models[None]
        per_model = models.shape[0] > 1
    op0 = np.asarray(plant_op0, dtype=np.complex128)
    ops = np.asarray(plant_ops, dtype=np.complex128)
    if op0.ndim == 2:
        op0 = op0[None]
    if ops.ndim == 3:
        ops = ops[None]
    per_plant = op0.shape[0] > 1 or ops.shape[0] > 1
    if per_plant:
        op0 = np.broadcast_to(op0, (Bn,) + op0.shape[1:])
        ops = np.broadcast_to(ops, (Bn,) + ops.shape[1:])
    X_targ = np.asarray(X_targ)
    per_targ = X_targ.ndim == 3
    T, ns = clock.horizon, clock.n_steps
    cols = min(X_targ.shape[-1], ns + T + 1)
    sess = EnsembleSession(Bn, n, dim_u, order, T, ns, clock.dt, sat, du, max_iter, warm_start, qp_flags, plant_kind,
                           per_model, per_plant, per_targ, cols, device=device, force_complex=force_complex, traceless=traceless, tile=tile,
                           measure_freq=getattr(clock, "measure_freq", 1), exact_qp=exact_qp, shared_generators=shared_generators)
    try:
        if models is None:
            sess.build_models(clock.dt, generators, scales)
        sess.load_problem(models, x0, X_targ, U_targ, Q, R, Qf, op0, ops)
    except Exception:
        sess.close()
        raise
    return sess


def mpc_batch(x0, models, dim_u, order, X_targ, U_targ, clock, plant_op0, plant_ops, Q, R, Qf, sat, du=None,
              max_iter=100, warm_start=True, qp_flags=None, plant_kind=_lib.PLANT_HAMILTONIAN, device=-1,
              force_complex=False, exact_qp=False, traceless=True, tile=None, generators=None, scales=None, shared_generators=None):
    """B independent closed loops in one launch.
    x0 [B, n]; models [B|1, n, n(1+P)] (or None with generators / scales: built on the device, see open_session);
    X_targ (n, cols) / U_targ (m, cols) shared (or [B, ...] each);
    plant_op0 [B|1, k, k], plant_ops [B|1, m, k, k].  Returns a dict: xs [B, n, n_steps+1], us [B, m, n_steps]
    (entries beyond steps_done are not meaningful), exit_codes, steps_done, qp_solves [B, n_steps]."""
    sess = open_session(x0, models, dim_u, order, X_targ, U_targ, clock, plant_op0, plant_ops, Q, R, Qf, sat, du, max_iter,
                        warm_start, qp_flags, plant_kind, device, force_complex, exact_qp, traceless, tile, generators, scales,
                        shared_generators)
    try:
        sess.run(0, clock.n_steps)
        res = sess.results()
        res["path"] = sess.path()
        res["path_detail"] = sess.path_detail()
        res["kernel_ms"] = sess.kernel_ms()[0]
        res["qp_stats"] = sess.qp_stats()
    finally:
        sess.close()
    res["xs"] = np.swapaxes(res["xs"], 1, 2)
    res["us"] = np.swapaxes(res["us"], 1, 2)
    return res
