"""WrapModel with the reference's interface (mpc4quantum/linearize.py:8-77), its trajectory
linearisation running in the HIP kernel `linearize_kernel` through m4q_linearize_batch."""
import numpy as np

from . import _lib
from .library import (create_library, create_library_from_list, create_power_list, diff_library, krtimes,  # noqa: F401
                      multinomial_powers, size_of_library)


class WrapModel:
    """x+ = A x + N (polyu(u) (x) x) with a polynomial control library up to `order`."""

    def __init__(self, A_op, N_op, dim_u, order):
        self.A = np.asarray(A_op)
        self.N = np.asarray(N_op)
        self.dim_x = self.A.shape[1]
        self.dim_u = dim_u
        self.order = order
        self.polyu_dim = int(self.N.shape[1] / self.dim_x)
        if size_of_library(order, dim_u) - 1 != self.polyu_dim:
            raise ValueError("Dimension mismatch when wrapping a model operator.")
        self.fns = create_library(order, dim_u)[1:]
        self.deriv_fns, self.deriv_coefs = diff_library(order, dim_u)
        self.unpacked_N = self.N.reshape(self.dim_x, self.polyu_dim, self.dim_x)
        self._model = np.ascontiguousarray(np.hstack([self.A, self.N]), dtype=np.complex128)

    # -- small host-side pieces (no kernel needed) --
    def lift_u(self, u_shaped):
        return np.vstack([f(u_shaped) for f in self.fns])

    def f(self, x, u, t=None):
        x = np.reshape(x, (-1, 1))
        return self.A @ x + self.N @ krtimes(self.lift_u(np.reshape(u, (-1, 1))), x)

    # -- GPU --
    def linearize_batch(self, X, U):
        """X [B, T, n] complex, U [B, T, m] -> A [B,T,n,n], B [B,T,n,m], Delta [B,T,n]."""
        X = np.ascontiguousarray(X, dtype=np.complex128)
        U = np.ascontiguousarray(U, dtype=np.float64)
        Bn, T, n = X.shape
        m = self.dim_u
        A = np.empty((Bn, T, n, n), dtype=np.complex128)
        Bm = np.empty((Bn, T, n, m), dtype=np.complex128)
        D = np.empty((Bn, T, n), dtype=np.complex128)
        L = _lib.lib()
        _lib.check(L.m4q_linearize_batch(Bn, n, m, self.order, T, _lib.cbuf(self._model)[1], 0, _lib.cbuf(X)[1],
                                         _lib.rbuf(U)[1], A.ctypes.data_as(_lib._dp), Bm.ctypes.data_as(_lib._dp),
                                         D.ctypes.data_as(_lib._dp)))
        return A, Bm, D

    def df_dx(self, x, u, t=None):
        A, _, _ = self.linearize_batch(np.reshape(x, (1, 1, -1)), np.reshape(u, (1, 1, -1)))
        return A[0, 0]

    def df_du(self, x, u, t=None):
        _, Bm, _ = self.linearize_batch(np.reshape(x, (1, 1, -1)), np.reshape(u, (1, 1, -1)))
        return Bm[0, 0]

    def get_model_along_traj(self, xs, us, ts):
        """xs (n, >=len(ts)), us (m, >=len(ts)) -> lists A_ls, B_ls, Delta_ls (Delta (n,1)) of length len(ts)."""
        T = len(ts)
        X = np.asarray(xs)[:, :T].T[None]
        U = np.real(np.asarray(us))[:, :T].T[None]
        A, Bm, D = self.linearize_batch(X, U)
        return list(A[0]), list(Bm[0]), [d.reshape(-1, 1) for d in D[0]]

    def get_model_from_initial(self, xs, us, ts):
        A, Bm, D = self.get_model_along_traj(np.asarray(xs)[:, :1], np.asarray(us)[:, :1], ts[:1])
        return A * len(ts), Bm * len(ts), D * len(ts)
