"""Plants.  `Experiment` keeps the reference's duck type (mpc4quantum/experiment.py:8-49: lift, proj,
simulate).  `QExperiment` is the state-preparation plant of experiment.py:175-212 with the ODE
  d rho/dt = -i [H0 + sum_k u_k(t) H_k, rho]
solved exactly over each held-control interval by the HIP Pade-13 kernel (m4q_plant_step_batch)
instead of qutip.mesolve.  `mpc()` recognises it and keeps the whole closed loop on the GPU."""
from abc import ABC, abstractmethod

import numpy as np

from . import _lib


class Experiment(ABC):
    def __init__(self):
        self.ts = None
        self.us = None
        self.xs = None

    @abstractmethod
    def f(self, t, x, u):
        """Time derivative of the state."""

    @staticmethod
    def lift(x):
        return x

    @staticmethod
    def proj(z):
        return z

    @abstractmethod
    def simulate(self, x0, ts, us):
        """States at all times in ts, shape (n, len(ts))."""


def _as_array(op):
    return np.asarray(op.full() if hasattr(op, "full") else op, dtype=np.complex128)


def plant_step_batch(x, u, op0, ops, dt, kind=_lib.PLANT_HAMILTONIAN):
    """x [B,n], u [B,m], op0 [B|1,k,k] (or [k,k]), ops [B|1,m,k,k] (or [m,k,k]) -> x_next [B,n]."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    u = np.ascontiguousarray(u, dtype=np.float64)
    Bn, n = x.shape
    m = u.shape[1]
    op0 = np.ascontiguousarray(op0, dtype=np.complex128)
    ops = np.ascontiguousarray(ops, dtype=np.complex128)
    per = 1 if (op0.ndim == 3 and op0.shape[0] > 1) else 0
    out = np.empty_like(x)
    L = _lib.lib()
    _lib.check(L.m4q_plant_step_batch(Bn, n, m, int(kind), float(dt), _lib.cbuf(x)[1], _lib.rbuf(u)[1], _lib.cbuf(op0)[1],
                                      _lib.cbuf(ops)[1], per, out.ctypes.data_as(_lib._dp)))
    return out


class QExperiment(Experiment):
    """Closed-system plant: H0 and H1_list are d x d Hermitian (ndarray or qutip.Qobj)."""

    def __init__(self, H0, H1_list):
        super().__init__()
        self.H0 = _as_array(H0)
        self.H1_list = [_as_array(h) for h in H1_list]
        self._me_args = {}
        self._sigma = 0

    def set(self, key, value):
        """Keyword argument of the reference's mesolve call (experiment.py:196-200).  The ones that change the ODE or its
        output are honoured: 'c_ops' (collapse operators: the plant becomes the Lindblad generator, still exact per held
        interval, still fused into the closed loop) and 'e_ops' (simulate returns expectation values, experiment.py:210).
        Anything else ('options', 'args', 'progress_bar', ...) only tunes qutip's integrator and is kept but unused."""
        self._me_args[key] = value

    def _c_ops(self):
        return [_as_array(c) for c in (self._me_args.get("c_ops") or [])]

    @property
    def plant_kind(self):
        return _lib.PLANT_GENERATOR if self._c_ops() else _lib.PLANT_HAMILTONIAN

    def operators(self):
        """(op0, ops) for the device plant: the Hamiltonians, or with collapse operators the generators on vec_r(rho)
        L0 = -i[H0, .] + sum_c (C . C^H - 1/2 {C^H C, .}),  L_k = -i[H_k, .]."""
        cs = self._c_ops()
        if not cs:
            return self.H0, np.stack(self.H1_list)
        from .vectorize import liouvillian
        d = self.H0.shape[0]
        eye = np.identity(d)
        L0 = liouvillian(self.H0)
        for c in cs:
            cc = c.conj().T @ c
            L0 = L0 + np.kron(c, c.conj()) - 0.5 * (np.kron(cc, eye) + np.kron(eye, cc.T))
        return L0, np.stack([liouvillian(h) for h in self.H1_list])

    def f(self, t, x, u):
        if self._c_ops():
            L0, Lk = self.operators()
            return (L0 + sum(l * uk for l, uk in zip(Lk, np.reshape(u, -1)))) @ np.reshape(x, -1)
        d = self.H0.shape[0]
        H = self.H0 + sum(h * uk for h, uk in zip(self.H1_list, np.reshape(u, -1)))
        rho = np.reshape(x, (d, d))
        return (-1j * (H @ rho - rho @ H)).reshape(-1)

    def set_sigma(self, sigma):
        self._sigma = sigma

    def simulate(self, x0, ts, us):
        """Piecewise-constant control (interp1d kind='previous', mpc.py:258): `us` is a callable of
        time or an (m, len(ts)) array whose column i is held on [ts[i], ts[i+1])."""
        ts = np.asarray(ts, dtype=float)
        m = len(self.H1_list)
        x = np.reshape(np.asarray(x0, dtype=np.complex128), -1)
        (op0, ops), kind = self.operators(), self.plant_kind
        cols = [x]
        for i in range(len(ts) - 1):
            u = np.reshape(us(ts[i]) if callable(us) else np.atleast_2d(us)[:, i], -1)[:m]
            x = plant_step_batch(x[None], np.real(u)[None], op0, ops, ts[i + 1] - ts[i], kind)[0]
            cols.append(x)
        self.ts, self.us = ts, us
        self.xs = np.stack(cols, axis=1)
        e_ops = self._me_args.get("e_ops")
        if e_ops is not None:                                # np.array(res.expect): tr(E rho(t)), one row per operator
            d = self.H0.shape[0]
            rho = self.xs.T.reshape(-1, d, d)
            self.xs = np.array([np.einsum('ij,tji->t', _as_array(e), rho) for e in e_ops])
        if self._sigma:
            noise = np.random.randn(*self.xs.shape) + 1j * np.random.randn(*self.xs.shape)
            return self.xs + noise * self._sigma
        return self.xs


class LExperiment(QExperiment):
    """Open-system plant: x' = (L0 + sum_k u_k L_k) x with n x n generators on vec_r(rho)."""

    plant_kind = _lib.PLANT_GENERATOR

    def operators(self):
        return self.H0, np.stack(self.H1_list)

    def f(self, t, x, u):
        L = self.H0 + sum(h * uk for h, uk in zip(self.H1_list, np.reshape(u, -1)))
        return L @ np.reshape(x, -1)

    def simulate(self, x0, ts, us):
        ts = np.asarray(ts, dtype=float)
        m = len(self.H1_list)
        x = np.reshape(np.asarray(x0, dtype=np.complex128), -1)
        cols = [x]
        for i in range(len(ts) - 1):
            u = np.reshape(us(ts[i]) if callable(us) else np.atleast_2d(us)[:, i], -1)[:m]
            x = plant_step_batch(x[None], np.real(u)[None], self.H0, np.stack(self.H1_list), ts[i + 1] - ts[i],
                                 _lib.PLANT_GENERATOR)[0]
            cols.append(x)
        self.ts, self.us = ts, us
        self.xs = np.stack(cols, axis=1)
        return self.xs


def split_blocks(bmatrix, nrows, ncols):
    """experiment.py:309-315: the (nrows x ncols) tiles of a block matrix, row-major over the tiles."""
    bmatrix = np.asarray(bmatrix)
    tiles_down, tiles_across = bmatrix.shape[0] // nrows, bmatrix.shape[1] // ncols
    return np.stack([bmatrix[a * nrows:(a + 1) * nrows, b * ncols:(b + 1) * ncols]
                     for a in range(tiles_down) for b in range(tiles_across)])


def isqrt(n):
    """Integer square root (experiment.py:311-327)."""
    import math
    if n < 0:
        raise ValueError("Square root not defined for negative numbers.")
    return math.isqrt(int(n))


class QExperiment32(QExperiment):
    """Three-level plant seen through its qubit block (experiment.py:215-235): the model lives on the 2x2 block."""

    @staticmethod
    def lift(rho33_vec):
        block = np.reshape(np.asarray(rho33_vec, dtype=np.complex128), (3, 3))[:2, :2]
        # Qobj.unit(): divide by the trace norm (sum of singular values)
        return (block / np.linalg.svd(block, compute_uv=False).sum()).flatten()

    @staticmethod
    def proj(rho22_vec):
        return np.asarray(rho22_vec).flatten()          # as the reference returns it (experiment.py:231-235)


class QCoupledExperiment(QExperiment):
    """Two identical subsystems: the model sees the two reduced states [vec(rho_A), vec(rho_B)] (partial traces),
    the plant the joint state (experiment.py:238-306)."""

    @staticmethod
    def lift(rhoAB_vec):
        v = np.asarray(rhoAB_vec, dtype=np.complex128).reshape(-1)
        dA = isqrt(isqrt(v.size))
        r = v.reshape(dA, dA, dA, dA)                    # r[a, b, a', b'] = <a b| rho |a' b'>
        return np.hstack([np.einsum('abcb->ac', r).flatten(), np.einsum('abad->bd', r).flatten()])

    @staticmethod
    def proj(rhoA_rhoB_vec):
        v = np.asarray(rhoA_rhoB_vec).reshape(-1)
        dA = isqrt(v.size // 2)
        return np.kron(v[:dA * dA].reshape(dA, dA), v[dA * dA:].reshape(dA, dA)).flatten()
