"""quad_program with the live reference signature (mpc4quantum/optimize.py:12), solved by the HIP
Riccati kernel (`qp_kernel`) through m4q_quad_program_batch instead of cvxpy + OSQP.

Same objective, dynamics (with Delta) and initial condition as optimize.py:27-41,54.  The box
|u| <= sat and the first-control band u_prev +- du are enforced by clipping in the forward
rollout: identical to the QP when no bound is active, feasible but sub-optimal when one is.
`exact=True` (M4Q_QP_EXACT_BOX) continues from that point with a projected-Newton iteration on the
same Riccati factorisation until the box-constrained optimum - what OSQP converges to - is reached."""
import numpy as np

from . import _lib


def _stack(ls, shape):
    return np.ascontiguousarray(np.stack([np.reshape(np.asarray(a), shape) for a in ls]), dtype=np.complex128)


def quad_program_batch(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls=None, u_prev=None, sat=None, du=None,
                       flags=None, exact=False):
    """Batched form.  x_init [B,n]; X_bm [B|1,T+1,n]; U_bm [B|1,T,m]; Q_ls [T+1,n,n]; R_ls [T,m,m];
    A_ls [B,T,n,n]; B_ls [B,T,n,m]; Delta_ls [B,T,n] or None; u_prev [B,m] or None.
    Returns X [B,T+1,n], U [B,T,m], cost [B], gains [B,T,n+1,m]."""
    if sat is None:
        raise TypeError("sat is required (the reference negates it unconditionally, optimize.py:43)")
    x_init = np.ascontiguousarray(x_init, dtype=np.complex128)
    A_ls = np.ascontiguousarray(A_ls, dtype=np.complex128)
    B_ls = np.ascontiguousarray(B_ls, dtype=np.complex128)
    Bn, T, n, m = B_ls.shape
    X_bm = np.ascontiguousarray(X_bm, dtype=np.complex128).reshape(-1, T + 1, n)
    U_bm = np.ascontiguousarray(np.real(U_bm), dtype=np.float64).reshape(-1, T, m)
    per = 1 if X_bm.shape[0] > 1 else 0
    if flags is None:
        flags = _lib.QP_DU_BAND if (u_prev is not None and du is not None) else 0
    if exact:
        flags |= _lib.QP_EXACT_BOX
    X = np.empty((Bn, T + 1, n), dtype=np.complex128)
    U = np.empty((Bn, T, m), dtype=np.float64)
    cost = np.empty(Bn, dtype=np.float64)
    gains = np.empty((Bn, T, n + 1, m), dtype=np.complex128)
    dptr = _lib._dp
    keep = [_lib.cbuf(x_init), _lib.cbuf(X_bm), _lib.rbuf(U_bm), _lib.cbuf(Q_ls), _lib.cbuf(R_ls), _lib.cbuf(A_ls),
            _lib.cbuf(B_ls)]
    d_ptr = _lib.cbuf(Delta_ls) if Delta_ls is not None else (None, None)
    up_ptr = _lib.rbuf(np.real(u_prev)) if u_prev is not None else (None, None)
    L = _lib.lib()
    _lib.check(L.m4q_quad_program_batch(Bn, n, m, T, int(flags), float(sat), float(du if du is not None else 0.0),
                                        keep[0][1], keep[1][1], keep[2][1], per, keep[3][1], keep[4][1], keep[5][1],
                                        keep[6][1], d_ptr[1], up_ptr[1], X.ctypes.data_as(dptr), U.ctypes.data_as(dptr),
                                        cost.ctypes.data_as(dptr), gains.ctypes.data_as(dptr)))
    return X, U, cost, gains


def quad_program(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls, u_prev=None, sat=None, du=None, verbose=False,
                 exact=False):
    """Drop-in for optimize.quad_program: lists of per-t arrays in, (X (n,T+1), U (m,T), obj_val, aux) out."""
    m, T = np.shape(U_bm)
    n = np.shape(X_bm)[0]
    X, U, cost, gains = quad_program_batch(
        np.reshape(x_init, (1, n)), np.asarray(X_bm)[:, :T + 1].T[None], np.real(np.asarray(U_bm))[:, :T].T[None],
        _stack(Q_ls, (n, n)), _stack(R_ls, (m, m)), _stack(A_ls, (n, n))[None], _stack(B_ls, (n, m))[None],
        _stack(Delta_ls, (n,))[None] if Delta_ls is not None else None,
        None if u_prev is None else np.reshape(np.real(u_prev), (1, m)), sat, du, exact=exact)
    aux = [gains[0, t].T.copy() for t in range(T)]           # Gains[t] is m x (n+1), as in lqr.py:61
    return X[0].T.copy(), U[0].T.copy(), float(cost[0]), aux
