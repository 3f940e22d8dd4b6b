"""quad_program with the signature and arithmetic of the reference's orphan mpc4quantum/lqr.py:14-79
(no Delta, ``du``/``u_prev`` ignored, absolute cost, augmented cost built on xbar_t), on the HIP
Riccati kernel with M4Q_QP_REF_LQR.  Used for parity against the reference's own file."""
import numpy as np

from . import _lib
from .optimize import _stack, quad_program_batch


def quad_program(x0, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, u_prev=None, sat=None, du=None, verbose=False):
    m, T = np.shape(U_bm)
    n = np.shape(X_bm)[0]
    X, U, cost, gains = quad_program_batch(
        np.reshape(x0, (1, n)), np.asarray(X_bm)[:, :T + 1].T[None], np.real(np.asarray(U_bm))[:, :T].T[None],
        _stack(Q_ls, (n, n)), _stack(R_ls, (m, m)), _stack(A_ls, (n, n))[None], _stack(B_ls, (n, m))[None],
        None, None, sat, None, flags=_lib.QP_REF_LQR)
    return X[0].T.copy(), U[0].T.copy(), float(cost[0]), [gains[0, t].T.copy() for t in range(T)]
