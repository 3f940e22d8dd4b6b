"""Resident ensemble session: thin object wrapper over the m4q_session_* C ABI (include/m4q.h).
All inputs live in HBM; `run` enqueues the fused closed-loop kernel on the session's stream."""
import ctypes as C

import numpy as np

from . import _lib

_DTYPES = {
    _lib.F_MODELS: np.complex128, _lib.F_X0: np.complex128, _lib.F_X_TARG: np.complex128, _lib.F_U_TARG: np.float64,
    _lib.F_Q: np.complex128, _lib.F_R: np.complex128, _lib.F_QF: np.complex128, _lib.F_OP0: np.complex128,
    _lib.F_OPS: np.complex128, _lib.F_XS: np.complex128, _lib.F_US: np.float64, _lib.F_CODES: np.int32,
    _lib.F_STEPS_DONE: np.int32, _lib.F_QP_SOLVES: np.int32, _lib.F_X_GUESS: np.complex128, _lib.F_U_GUESS: np.float64,
}


class EnsembleSession:
    def __init__(self, B, dim_x, dim_u, order, horizon, n_steps, dt, sat, du=None, max_iter=100, warm_start=True,
                 qp_flags=None, plant_kind=_lib.PLANT_HAMILTONIAN, model_per_instance=False, plant_per_instance=False,
                 target_per_instance=False, target_cols=None, ls_tol=1e-4, device=-1, force_complex=False, measure_freq=1,
                 exact_qp=False, traceless=True, tile=None, shared_generators=None):
        """exact_qp: solve every QP of the loop to the box-constrained optimum (M4Q_QP_EXACT_BOX, what the reference's OSQP
        call converges to) instead of clipping the Riccati rollout."""
        if sat is None:
            raise TypeError("sat is required (the reference negates it unconditionally, optimize.py:43 / lqr.py:76)")
        p = _lib.Problem()
        p.dim_x, p.dim_u, p.order, p.horizon, p.n_steps = dim_x, dim_u, order, horizon, n_steps
        p.max_iter, p.warm_start = int(max_iter), int(bool(warm_start))
        p.qp_flags = int(qp_flags if qp_flags is not None else (_lib.QP_DU_BAND if du is not None else 0))
        if exact_qp:
            p.qp_flags |= _lib.QP_EXACT_BOX
        p.plant_kind = int(plant_kind)
        p.model_per_instance, p.plant_per_instance = int(model_per_instance), int(plant_per_instance)
        p.target_per_instance = int(target_per_instance)
        p.target_cols = int(target_cols if target_cols is not None else n_steps + horizon + 1)
        # traceless=False keeps a real-path session on its d*d coordinates (M4Q_OPT_NO_TRACELESS) instead of the d*d - 1 traceless ones
        # tile: None / True = the library's choice (the backward sweep on matrix-core tiles where that form is built: d = 2, 3 with an
        # order-1 model, whenever the target is constant over the window), False = DPP sweeps (M4Q_OPT_NO_TILE)
        # shared_generators: None / True = the library's choice (models built by build_models from ONE generator set at order 1 run the
        # clipped traceless solve on the shared generators where that kernel is built: d = 4), False = per-member models (M4Q_OPT_NO_SG)
        p.reserved = (_lib.OPT_FORCE_COMPLEX if force_complex else 0) | (0 if traceless else _lib.OPT_NO_TRACELESS) | \
            (_lib.OPT_NO_TILE if tile is False else 0) | (_lib.OPT_NO_SG if shared_generators is False else 0)
        p.measure_freq = int(measure_freq)
        p.dt, p.sat, p.du, p.ls_tol = float(dt), float(sat), float(du if du is not None else 0.0), float(ls_tol)
        self.problem = p
        self.B = int(B)
        self._h = C.c_void_p()
        self._L = _lib.lib()
        _lib.check(self._L.m4q_session_create(C.byref(p), self.B, int(device), C.byref(self._h)))
        self._keep = {}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.m4q_session_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def field_bytes(self, field):
        return int(self._L.m4q_session_field_bytes(self._h, field))

    def upload(self, field, array):
        a = np.ascontiguousarray(array, dtype=_DTYPES[field])
        if a.nbytes != self.field_bytes(field):
            raise ValueError("field %d expects %d bytes, array has %d (shape %s)" % (field, self.field_bytes(field), a.nbytes,
                                                                                      a.shape))
        _lib.check(self._L.m4q_session_upload(self._h, field, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, field, shape):
        out = np.empty(shape, dtype=_DTYPES[field])
        _lib.check(self._L.m4q_session_download(self._h, field, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def put_state(self, step, x):
        a = np.ascontiguousarray(x, dtype=np.complex128).reshape(self.B, self.problem.dim_x)
        _lib.check(self._L.m4q_session_put_state(self._h, int(step), a.ctypes.data_as(C.c_void_p)))

    def get_state(self, step):
        out = np.empty((self.B, self.problem.dim_x), dtype=np.complex128)
        _lib.check(self._L.m4q_session_get_state(self._h, int(step), out.ctypes.data_as(C.c_void_p)))
        return out

    def device_ptr(self, field):
        return self._L.m4q_session_device_ptr(self._h, field)

    def bind_output(self, field, device_ptr, nbytes):
        _lib.check(self._L.m4q_session_bind_output(self._h, field, C.c_void_p(device_ptr), nbytes))

    def copy_final_state(self, device_ptr):
        """xs[:, n_steps, :] -> device memory [B][n] complex, on the session stream (final-state-only gather buffers)."""
        _lib.check(self._L.m4q_session_copy_final_state(self._h, C.c_void_p(device_ptr)))

    def copy_status(self, device_ptr):
        """the launch's watchdog flag -> one int32 of device memory, on the session stream (a gather buffer's status word)."""
        _lib.check(self._L.m4q_session_copy_status(self._h, C.c_void_p(device_ptr)))

    def build_models(self, dt, generators, scales=None):
        """Fill the MODELS field on the device: generators [1+m, n, n] (shared) or [B, 1+m, n, n], scales [B, 1+m]."""
        g = np.ascontiguousarray(generators, dtype=np.complex128)
        per = 1 if g.ndim == 4 and g.shape[0] > 1 else 0
        sc = None if scales is None else np.ascontiguousarray(scales, dtype=np.float64)
        _lib.check(self._L.m4q_session_build_models(self._h, float(dt), g.ctypes.data_as(_lib._dp), per,
                                                    sc.ctypes.data_as(_lib._dp) if sc is not None else None))

    def run(self, step_begin=0, step_end=None):
        _lib.check(self._L.m4q_session_run(self._h, int(step_begin), int(self.problem.n_steps if step_end is None else step_end)))

    def sync(self):
        _lib.check(self._L.m4q_session_sync(self._h))

    def set_codes(self, codes):
        a = np.ascontiguousarray(codes, dtype=np.int32)
        _lib.check(self._L.m4q_session_set_codes(self._h, a.ctypes.data_as(_lib._ip)))

    def kernel_ms(self):
        ms = C.c_double()
        n = C.c_int32()
        _lib.check(self._L.m4q_session_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def qp_stats(self):
        """exact_qp sessions: (QP solves, pinned Riccati sweeps, ratio-test steps, solves ended by KKT / at working
        precision / by the iteration cap) of the last launch."""
        out = (C.c_int64 * 6)()
        _lib.check(self._L.m4q_session_qp_stats(self._h, out))
        return tuple(int(v) for v in out)

    def path(self):
        """'real' if the uploaded problem runs in a Hermitian operator basis with real arithmetic, else 'complex'."""
        return "real" if _lib.check(self._L.m4q_session_path(self._h)) >= 1 else "complex"

    def path_detail(self):
        """'complex', 'real' (d*d Hermitian coordinates), 'traceless' (the d*d - 1 traceless Hermitian coordinates) or
        'traceless-tile' (the same with the backward sweep of the clipped solve / the pinned sweep of the exact solve on fp64
        matrix-core tiles), or 'traceless-sg' (the traceless clipped solve on shared generators and per-member scales)."""
        return ("complex", "real", "traceless", "traceless-tile", "traceless-sg")[_lib.check(self._L.m4q_session_path(self._h))]

    def info(self):
        hbm = C.c_int64()
        grid = C.c_int32()
        lds = C.c_int32()
        _lib.check(self._L.m4q_session_info(self._h, C.byref(hbm), C.byref(grid), C.byref(lds)))
        return {"hbm_bytes": hbm.value, "grid": grid.value, "lds_bytes": lds.value}

    # ---- convenience ----
    def load_problem(self, models, x0, X_targ, U_targ, Q, R, Qf, op0=None, ops=None):
        """Reference-shaped inputs: X_targ (n, cols) / U_targ (m, cols) (or with a leading ensemble
        axis), models [B|1, n, n(1+P)] (or [n, n(1+P)])."""
        p = self.problem
        n, m, cols = p.dim_x, p.dim_u, p.target_cols
        if models is not None:                       # None: already built on the device (build_models)
            self.upload(_lib.F_MODELS, models)
        self.upload(_lib.F_X0, x0)
        Xt = np.asarray(X_targ, dtype=np.complex128)
        Ut = np.real(np.asarray(U_targ)).astype(np.float64)
        Ut_full = np.zeros(Xt.shape[:-2] + (m, cols))
        Ut_full[..., :min(cols, Ut.shape[-1])] = Ut[..., :cols]
        self.upload(_lib.F_X_TARG, np.swapaxes(Xt[..., :cols], -1, -2))
        self.upload(_lib.F_U_TARG, np.swapaxes(Ut_full, -1, -2))
        self.upload(_lib.F_Q, np.asarray(Q, dtype=np.complex128).reshape(n, n))
        self.upload(_lib.F_R, np.asarray(R, dtype=np.complex128).reshape(m, m))
        self.upload(_lib.F_QF, np.asarray(Qf, dtype=np.complex128).reshape(n, n))
        if p.plant_kind != _lib.PLANT_NONE:
            self.upload(_lib.F_OP0, op0)
            self.upload(_lib.F_OPS, ops)

    def state(self):
        """Everything a run needs to resume at MPC step k (checkpoint): states, controls, SQP guesses, codes."""
        p = self.problem
        B, n, m, ns, T = self.B, p.dim_x, p.dim_u, p.n_steps, p.horizon
        return {
            "xs": self.download(_lib.F_XS, (B, ns + 1, n)), "us": self.download(_lib.F_US, (B, ns, m)),
            "x_guess": self.download(_lib.F_X_GUESS, (B, T + 1, n)), "u_guess": self.download(_lib.F_U_GUESS, (B, T, m)),
            "exit_codes": self.download(_lib.F_CODES, (B,)), "steps_done": self.download(_lib.F_STEPS_DONE, (B,)),
        }

    def restore(self, state):
        """Inverse of state(): afterwards run(k, ...) continues a run interrupted before step k."""
        self.upload(_lib.F_XS, state["xs"])
        self.upload(_lib.F_US, state["us"])
        self.upload(_lib.F_X_GUESS, state["x_guess"])
        self.upload(_lib.F_U_GUESS, state["u_guess"])
        self.upload(_lib.F_STEPS_DONE, state["steps_done"])
        self.set_codes(state["exit_codes"])

    def results(self):
        p = self.problem
        B, n, m, ns = self.B, p.dim_x, p.dim_u, p.n_steps
        return {
            "xs": self.download(_lib.F_XS, (B, ns + 1, n)),
            "us": self.download(_lib.F_US, (B, ns, m)),
            "exit_codes": self.download(_lib.F_CODES, (B,)),
            "steps_done": self.download(_lib.F_STEPS_DONE, (B,)),
            "qp_solves": self.download(_lib.F_QP_SOLVES, (B, ns)),
        }
