"""CPU oracle for the MPC hot path (TEST INFRASTRUCTURE - never shipped, never measured as product).

This module is a NumPy/SciPy restatement of the arithmetic of the reference's per-step MPC
pipeline.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product (``mpc4quantum_amd``) never does and fails loudly when
its HIP library is missing.

Parity status (see tests/golden/make_golden.py and tests/test_oracle_golden.py):
  * library tables, ``krtimes``, ``discretize_homogeneous``, ``WrapModel.*`` and
    ``lqr.quad_program`` are PINNED against the reference's own files, loaded by path in the build
    container, through the committed ``tests/golden/*.npz`` vectors (<= 1e-12);
  * ``discretize_homogeneous`` is also pinned by the reference's known-answer test
    (tests/test_mpc4quantum.py:147-188, order 1 / dt 1  =>  [I + A | N_1 | ...]);
  * the ``Delta``/``X_bm[t+1]`` extension of the Riccati solve (mode "qp") is pinned by an
    independent dense KKT solve of the equality-constrained QP stated at optimize.py:27-41,54 -
    at small T and at the headline configuration's own T = 40 with Delta != 0, a ramped state
    target and a control target (tests/test_oracle_golden.py);
  * the driver (``mpc``: SQP loop, line search, shift, exit codes, clock) is pinned against the
    reference's own mpc.py run around its own lqr.py (tests/golden/mpc_loop*.npz);
  * the live cvxpy/OSQP arithmetic (optimize.py:58-59) and qutip.mesolve (experiment.py:209)
    are third-party and absent: those two boundaries are PARITY UNPINNED.  The plant is restated
    as the exact propagator of the ODE at experiment.py:190-191,208 under a held control.

All ``file:line`` citations are into /root/reference/.
"""
import itertools
import math

import numpy as np
from scipy.linalg import expm as _expm

__all__ = [
    "multinomial_powers", "create_power_list", "size_of_library", "monomials", "diff_tables",
    "krtimes", "discretize_homogeneous", "vectorize_me", "liouvillian_ij",
    "OracleWrapModel", "OracleDMDc", "lqr_quad_program", "quad_program", "kkt_quad_program", "exact_quad_program",
    "iqp_line_search", "shift_guess", "OracleClock", "OracleQExperiment", "plant_step",
    "mpc", "mpc_batch",
]


# --------------------------------------------------------------------------------------------
# Control-monomial library  (mpc4quantum/linearize.py:92-164)
# --------------------------------------------------------------------------------------------
def multinomial_powers(n, k):
    """Exponent tuples of (x_1+..+x_k)^n in dots-and-bars order (linearize.py:92-110)."""
    for bars in itertools.combinations(range(n + k - 1), k - 1):
        edges = (-1,) + tuple(bars) + (n + k - 1,)
        yield np.array([edges[i + 1] - edges[i] - 1 for i in range(k)])


def create_power_list(order, dimension):
    """All exponent vectors of total degree <= order; first entry is the constant
    (linearize.py:113-116: drop the slack variable, reverse the rest)."""
    return [p[:-1][::-1] for p in multinomial_powers(order, dimension + 1)]


def size_of_library(order, dimension):
    """linearize.py:119-120."""
    return len(create_power_list(order, dimension))


def monomials(u, powers):
    """Evaluate prod_i u_i**p_i for each exponent row; a negative exponent gives 0
    (linearize.py:123-128).  ``u`` is (m,) or (m, k); returns (len(powers),) or (len(powers), k)."""
    u = np.asarray(u, dtype=float)
    u2 = u.reshape(u.shape[0], -1)
    out = np.ones((len(powers), u2.shape[1]))
    for r, ps in enumerate(powers):
        for i, p in enumerate(ps):
            out[r] = out[r] * (np.zeros_like(u2[i]) if p < 0 else np.power(u2[i], p))
    return out.reshape((len(powers),) + u.shape[1:])


def diff_tables(order, dimension):
    """Derivative exponent tables and coefficients of the non-constant monomials
    (linearize.py:143-164).  Returns (deriv_powers[k] (P x m), deriv_coef[k] (P,)) per control k."""
    plist = np.vstack(create_power_list(order, dimension)[1:])
    unit = create_power_list(1, dimension)[1:]
    dpow, dcoef = [], []
    for d in unit:
        dpow.append(plist - d)
        dcoef.append(plist[:, d.astype(bool)].reshape(-1))
    return dpow, dcoef


def krtimes(A, B):
    """Column-wise Khatri-Rao product, row index = p * n + j (linearize.py:80-89)."""
    A = np.asarray(A)
    B = np.asarray(B)
    if A.shape[1] != B.shape[1]:
        raise ValueError("Cols of A =/ Cols of B")
    return (A[:, None, :] * B[None, :, :]).reshape(A.shape[0] * B.shape[0], A.shape[1])


# --------------------------------------------------------------------------------------------
# Model construction  (mpc4quantum/vectorize.py)
# --------------------------------------------------------------------------------------------
def discretize_homogeneous(A_cts_list, dt, order):
    """Taylor/Dyson-truncated exp(dt * (A_0 + sum_k u_k A_k)) binned by control monomial
    (vectorize.py:8-49).  Output is n x n(1+P) complex, blocks in create_power_list order."""
    ops = [np.asarray(a, dtype=complex) for a in A_cts_list]
    n = ops[0].shape[0]
    m = len(ops) - 1
    powers = [tuple(int(v) for v in p) for p in create_power_list(order, m)]
    blocks = [np.zeros((n, n), dtype=complex) for _ in powers]
    for k in range(order + 1):
        scale = dt ** k / math.factorial(k)
        for word in itertools.product(range(m + 1), repeat=k):
            term = np.identity(n, dtype=complex)
            for letter in word:
                term = term @ ops[letter]
            counts = tuple(sum(1 for w in word if w == c) for c in range(1, m + 1))
            hits = [i for i, p in enumerate(powers) if p == counts]
            if len(hits) != 1:
                raise ValueError("Error in discretization. Control powers should contribute uniquely.")
            blocks[hits[0]] = blocks[hits[0]] + scale * term
    return np.hstack(blocks)


def vectorize_me(H, measure_list):
    """Project -i[H, .] on a measurement basis through its structure constants
    (vectorize.py:52-75), with plain ndarrays in place of qutip.Qobj."""
    H = np.asarray(H, dtype=complex)
    basis = [np.asarray(s, dtype=complex) for s in measure_list]
    dm = len(basis)
    struct = np.zeros((dm, dm, dm), dtype=complex)
    for i, si in enumerate(basis):
        for j, sj in enumerate(basis):
            if i == j:
                continue
            comm_dag = (si @ sj - sj @ si).conj().T
            for k, sk in enumerate(basis):
                struct[i, j, k] = np.trace(comm_dag @ sk)
    h = np.array([np.trace(H.conj().T @ s) for s in basis])
    A = np.zeros((dm, dm), dtype=complex)
    for k in range(dm):
        for j in range(dm):
            # the reference fills a (k, j)-ordered list and reshapes row-major (vectorize.py:69-75)
            A[k, j] = -1j * np.sum(h * struct[:, k, j])
    return A


def liouvillian_ij(H):
    """Closed form of ``vectorize_me`` for the |i><j| basis ordered i-major: the generator of
    d/dt vec_r(rho) = -i [H, rho]  is  -i (H (x) I - I (x) H^T)."""
    H = np.asarray(H, dtype=complex)
    d = H.shape[0]
    eye = np.identity(d)
    return -1j * (np.kron(H, eye) - np.kron(eye, H.T))


class OracleDMDc:
    """Read-only DMDc container (model.py:7-103): A is (dim_y, dim_x + dim_u)."""

    def __init__(self, dim_y, dim_x, dim_u, A0):
        self.dim_y, self.dim_x, self.dim_u = dim_y, dim_x, dim_u
        self.A = np.asarray(A0)

    def get_discrete(self):
        return self.A[:self.dim_y, :self.dim_x], self.A[:self.dim_y, self.dim_x:]

    def predict(self, x, ux):
        A_x, A_u = self.get_discrete()
        return A_x @ np.reshape(x, (self.dim_x, -1)) + A_u @ np.reshape(ux, (self.dim_u, -1))


# --------------------------------------------------------------------------------------------
# Bilinear linearisation  (mpc4quantum/linearize.py:8-70)
# --------------------------------------------------------------------------------------------
class OracleWrapModel:
    """x+ = A x + N (polyu(u) (x) x); Jacobians along a trajectory."""

    def __init__(self, A_op, N_op, dim_u, order):
        self.A = np.asarray(A_op)
        self.N = np.asarray(N_op)
        self.dim_x = self.A.shape[1]
        self.dim_u = dim_u
        self.order = order
        self.polyu_dim = int(self.N.shape[1] / self.dim_x)
        if size_of_library(order, dim_u) - 1 != self.polyu_dim:
            raise ValueError("Dimension mismatch when wrapping a model operator.")
        self.powers = create_power_list(order, dim_u)[1:]
        self.dpow, self.dcoef = diff_tables(order, dim_u)
        # N[:, p, :] is the n x n block paired with monomial p (linearize.py:32)
        self.N3 = self.N.reshape(self.dim_x, self.polyu_dim, self.dim_x)

    def lift_u(self, u):
        return monomials(np.reshape(u, (self.dim_u, -1)), self.powers)

    def f(self, x, u, t=None):
        x = np.reshape(x, (-1, 1))
        pu = self.lift_u(np.reshape(u, (-1, 1)))
        return self.A @ x + self.N @ krtimes(pu, x)

    def df_dx(self, x, u, t=None):
        pu = self.lift_u(np.reshape(u, (-1, 1)))[:, 0]
        return self.A + np.einsum('ipj,p->ij', self.N3, pu)

    def df_du(self, x, u, t=None):
        x = np.reshape(x, (-1,))
        uu = np.reshape(u, (-1, 1))
        polyB = np.einsum('ipj,j->ip', self.N3, x)
        cols = []
        for k in range(self.dim_u):
            w = self.dcoef[k] * monomials(uu, self.dpow[k])[:, 0]
            cols.append(polyB @ w)
        return np.stack(cols, axis=1)

    def get_model_along_traj(self, xs, us, ts):
        A_ls, B_ls, D_ls = [], [], []
        for i in range(len(ts)):
            A_ls.append(self.df_dx(xs[:, i], us[:, i]))
            B_ls.append(self.df_du(xs[:, i], us[:, i]))
            pred = A_ls[-1] @ xs[:, i] + B_ls[-1] @ us[:, i]
            D_ls.append(self.f(xs[:, i], us[:, i]) - pred[:, None])
        return A_ls, B_ls, D_ls


# --------------------------------------------------------------------------------------------
# Finite-horizon LQR  (mpc4quantum/lqr.py:14-79, and the QP statement of optimize.py:12-60)
# --------------------------------------------------------------------------------------------
def _dag(M):
    return M.conj().T


def _riccati(x0, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, c_ls, r_ls, D_ls, lo0, hi0, sat, abs_cost):
    """Shared sweep.  Augmented state z = [x - xbar; 1].
    c_ls[t]: affine column of the augmented dynamics; r_ls[t]: offset used in the augmented cost;
    D_ls[t]: affine term added in the forward rollout; (lo0, hi0): extra box on the first control."""
    m, T = U_bm.shape
    n = X_bm.shape[0]

    def q_aug(Qt, r):
        qr = Qt @ r
        return np.block([[Qt, -qr], [-_dag(qr), _dag(r) @ qr]])

    V = q_aug(Q_ls[T], r_ls[T])
    gains = [None] * T
    for t in reversed(range(T)):
        A_aug = np.block([[A_ls[t], c_ls[t]], [np.zeros((1, n)), np.ones((1, 1))]])
        B_aug = np.vstack([B_ls[t], np.zeros((1, m))])
        G = R_ls[t] + _dag(B_aug) @ V @ B_aug
        gains[t] = -np.linalg.pinv(G) @ _dag(B_aug) @ V @ A_aug           # lqr.py:61-62
        S = A_aug + B_aug @ gains[t]
        V = q_aug(Q_ls[t], r_ls[t]) + _dag(gains[t]) @ R_ls[t] @ gains[t] + _dag(S) @ V @ S   # :64-65

    X = np.zeros((n, T + 1), dtype=complex)
    U = np.zeros((m, T))
    X[:, 0] = np.reshape(x0, -1)
    cost = 0.0
    for t in range(T):
        z = np.concatenate([X[:, t] - X_bm[:, t], [1.0]])
        u = (gains[t] @ z).real + U_bm[:, t].real                           # lqr.py:75-76
        lo = -sat * np.ones(m)
        hi = sat * np.ones(m)
        if t == 0 and lo0 is not None:
            lo, hi = np.maximum(lo, lo0), np.minimum(hi, hi0)
        u = np.minimum(np.maximum(u, lo), hi)
        U[:, t] = u
        X[:, t + 1] = A_ls[t] @ X[:, t] + B_ls[t] @ u + D_ls[t]
        if abs_cost:
            ex, eu = X[:, t + 1], u                                           # lqr.py:78
            cost += (ex.conj() @ Q_ls[t + 1] @ ex).real + (eu @ R_ls[t] @ eu).real
        else:
            ex, eu = X[:, t] - X_bm[:, t], u - U_bm[:, t].real               # optimize.py:33-34
            cost += (ex.conj() @ Q_ls[t] @ ex).real + (eu @ R_ls[t] @ eu).real
    if not abs_cost:
        ex = X[:, T] - X_bm[:, T]                                            # optimize.py:54
        cost += (ex.conj() @ Q_ls[T] @ ex).real
    return X, U, float(cost), gains


def lqr_quad_program(x0, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, u_prev=None, sat=None, du=None, verbose=False):
    """Restatement of lqr.quad_program AS WRITTEN (lqr.py:14-79), including its quirks:
    affine column (A_t - I) xbar_t + B_t ubar_t (assumes xbar_{t+1} == xbar_t, lqr.py:45);
    augmented cost built around xbar_t although z already is a deviation (lqr.py:28-34,54-58);
    no Delta, ``du``/``u_prev`` ignored, absolute cost (lqr.py:78)."""
    m, T = U_bm.shape
    n = X_bm.shape[0]
    X_bm = np.asarray(X_bm, dtype=complex)
    U_bm = np.asarray(U_bm, dtype=float)
    c_ls = [((A_ls[t] - np.identity(n)) @ X_bm[:, t] + B_ls[t] @ U_bm[:, t]).reshape(-1, 1) for t in range(T)]
    r_ls = [X_bm[:, t].reshape(-1, 1) for t in range(T)] + [X_bm[:, -1].reshape(-1, 1)]
    D_ls = [np.zeros(n)] * T
    return _riccati(x0, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, c_ls, r_ls, D_ls, None, None, sat, True)


def quad_program(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls, u_prev=None, sat=None, du=None,
                 verbose=False):
    """The QP of optimize.quad_program (optimize.py:12-60) solved by an affine Riccati sweep with
    clipping instead of OSQP.  Same 12-argument signature and return arity.
      objective   sum_t<T (x_t-xb_t)^H Q_t (x_t-xb_t) + (u_t-ub_t)^T R_t (u_t-ub_t) + terminal   (:33-34,54)
      dynamics    x_{t+1} = Delta_t + A_t x_t + B_t u_t                                            (:41)
      box         |u_t| <= sat  (:43), first control within +-du of u_prev (:29-30): by clipping.
    Exact when no bound is active; feasible but sub-optimal otherwise (documented difference)."""
    m, T = U_bm.shape
    n = X_bm.shape[0]
    X_bm = np.asarray(X_bm, dtype=complex)
    U_bm = np.asarray(U_bm, dtype=float)
    c_ls = []
    for t in range(T):
        c = A_ls[t] @ X_bm[:, t] + B_ls[t] @ U_bm[:, t] + np.reshape(Delta_ls[t], -1) - X_bm[:, t + 1]
        c_ls.append(c.reshape(-1, 1))
    r_ls = [np.zeros((n, 1))] * (T + 1)
    D_ls = [np.reshape(d, -1) for d in Delta_ls]
    lo0 = hi0 = None
    if u_prev is not None and du is not None:
        up = np.reshape(u_prev, -1).real
        lo0, hi0 = up - du, up + du
    return _riccati(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, c_ls, r_ls, D_ls, lo0, hi0, sat, False)


def kkt_quad_program(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls):
    """Independent check: dense KKT solve of the equality-constrained QP (optimize.py:27,33-41,54,
    inequalities dropped), real-ified.  Unknowns: Re/Im of x_1..x_T and u_0..u_{T-1}."""
    m, T = U_bm.shape
    n = X_bm.shape[0]
    nx, nu = 2 * n * T, m * T
    nz = nx + nu

    def c2r(M):
        M = np.asarray(M, dtype=complex)
        return np.block([[M.real, -M.imag], [M.imag, M.real]])

    def v2r(v):
        v = np.asarray(v, dtype=complex).reshape(-1)
        return np.concatenate([v.real, v.imag])

    Hm = np.zeros((nz, nz))
    g = np.zeros(nz)
    for t in range(1, T + 1):
        Qr = c2r(Q_ls[t])
        Qs = (Qr + Qr.T) / 2
        sl = slice(2 * n * (t - 1), 2 * n * t)
        Hm[sl, sl] = 2 * Qs
        g[sl] = -2 * Qs @ v2r(X_bm[:, t])
    for t in range(T):
        Rr = np.asarray(R_ls[t]).real
        Rs = (Rr + Rr.T) / 2
        sl = slice(nx + m * t, nx + m * (t + 1))
        Hm[sl, sl] = 2 * Rs
        g[sl] = -2 * Rs @ np.asarray(U_bm[:, t]).real
    E = np.zeros((2 * n * T, nz))
    rhs = np.zeros(2 * n * T)
    for t in range(T):
        rows = slice(2 * n * t, 2 * n * (t + 1))
        E[rows, 2 * n * t:2 * n * (t + 1)] = np.identity(2 * n)
        Br = np.asarray(B_ls[t], dtype=complex)
        E[rows, nx + m * t:nx + m * (t + 1)] = -np.vstack([Br.real, Br.imag])
        rhs[rows] = v2r(Delta_ls[t])
        if t == 0:
            rhs[rows] += c2r(A_ls[0]) @ v2r(x_init)
        else:
            E[rows, 2 * n * (t - 1):2 * n * t] = -c2r(A_ls[t])
    K = np.block([[Hm, E.T], [E, np.zeros((E.shape[0], E.shape[0]))]])
    sol = np.linalg.lstsq(K, np.concatenate([-g, rhs]), rcond=None)[0]
    X = np.zeros((n, T + 1), dtype=complex)
    X[:, 0] = np.reshape(x_init, -1)
    for t in range(1, T + 1):
        seg = sol[2 * n * (t - 1):2 * n * t]
        X[:, t] = seg[:n] + 1j * seg[n:]
    U = sol[nx:nx + nu].reshape(T, m).T
    return X, U


def exact_quad_program(x_init, X_bm, U_bm, Q_ls, R_ls, A_ls, B_ls, Delta_ls, u_prev=None, sat=None, du=None):
    """The box-constrained QP of optimize.quad_program (optimize.py:27-43,54) solved EXACTLY on the CPU: the
    states are condensed out and the bounded linear least-squares problem in the controls goes to
    scipy.optimize.lsq_linear (BVLS active set).  Reference point for the next-tier device solver:
    the shipped Riccati path clips instead, which is sub-optimal when a bound is active."""
    from scipy.linalg import sqrtm
    from scipy.optimize import lsq_linear
    m, T = U_bm.shape
    n = X_bm.shape[0]
    X_bm = np.asarray(X_bm, dtype=complex)
    U_bm = np.asarray(U_bm, dtype=float)
    lo = -sat * np.ones((T, m))
    hi = sat * np.ones((T, m))
    if u_prev is not None and du is not None:
        up = np.reshape(u_prev, -1).real
        lo[0], hi[0] = np.maximum(lo[0], up - du), np.minimum(hi[0], up + du)

    def c2r(M):
        M = np.asarray(M, dtype=complex)
        return np.block([[M.real, -M.imag], [M.imag, M.real]])

    def v2r(v):
        v = np.asarray(v, dtype=complex).reshape(-1)
        return np.concatenate([v.real, v.imag])
    free = [np.asarray(x_init, dtype=complex).reshape(-1)]
    for t in range(T):
        free.append(A_ls[t] @ free[-1] + np.reshape(Delta_ls[t], -1))
    Phi = np.zeros((T + 1, n, T * m), dtype=complex)
    for s_ in range(T):
        for k in range(m):
            v = np.asarray(B_ls[s_], dtype=complex)[:, k]
            Phi[s_ + 1, :, s_ * m + k] = v
            for t in range(s_ + 1, T):
                v = A_ls[t] @ v
                Phi[t + 1, :, s_ * m + k] = v
    rows, rhs = [], []
    for t in range(T + 1):
        Wq = np.real(sqrtm(c2r(Q_ls[t])))
        rows.append(Wq @ np.vstack([Phi[t].real, Phi[t].imag]))
        rhs.append(Wq @ (v2r(X_bm[:, t]) - v2r(free[t])))
    for t in range(T):
        Wr = np.real(sqrtm(np.asarray(R_ls[t]).real))
        E = np.zeros((m, T * m))
        E[:, t * m:(t + 1) * m] = np.identity(m)
        rows.append(Wr @ E)
        rhs.append(Wr @ U_bm[:, t])
    res = lsq_linear(np.vstack(rows), np.concatenate(rhs), bounds=(lo.reshape(-1), hi.reshape(-1)), method='bvls', tol=1e-15,
                     max_iter=5000)
    U = res.x.reshape(T, m).T
    X = np.zeros((n, T + 1), dtype=complex)
    X[:, 0] = free[0]
    cost = 0.0
    for t in range(T):
        X[:, t + 1] = A_ls[t] @ X[:, t] + B_ls[t] @ U[:, t] + np.reshape(Delta_ls[t], -1)
    for t in range(T + 1):
        e = X[:, t] - X_bm[:, t]
        cost += (e.conj() @ Q_ls[t] @ e).real
    for t in range(T):
        e = U[:, t] - U_bm[:, t]
        cost += (e @ R_ls[t] @ e).real
    return X, U, float(cost)


# --------------------------------------------------------------------------------------------
# Driver pieces  (mpc4quantum/mpc.py)
# --------------------------------------------------------------------------------------------
class OracleClock:
    """StepClock (mpc.py:14-35)."""

    def __init__(self, dt, horizon, n_steps):
        self.dt = float(dt)
        self.horizon = horizon
        self.n_steps = n_steps
        self.measure_freq = 1
        self.ts = np.linspace(0, self.dt * n_steps, n_steps, endpoint=False)
        self.ts_sim = self.ts

    def set_endsim(self, index):
        self.ts_sim = self.ts[:index]

    def ts_step(self, a_step):
        return np.linspace(self.dt * (a_step + 1 - self.measure_freq), self.dt * (a_step + 1), self.measure_freq + 1)

    def ts_horizon(self, a_step):
        return np.linspace(self.dt * a_step, self.dt * (a_step + self.horizon), self.horizon, endpoint=False)


def shift_guess(data):
    """Drop the first column, repeat the last (mpc.py:71-73)."""
    return np.hstack([data[:, 1:], data[:, -1:]])


def line_search_weights(Q_ls, R_ls):
    """Symmetrised block-diagonal cost of iqp_line_search as a list of dense real blocks
    (mpc.py:103-104,112-116): one 2n x 2n block per horizon node then one 2m x 2m per control."""
    def c2r(P):
        P = np.asarray(P, dtype=complex)
        return np.block([[P.real, -P.imag], [P.imag, P.real]])
    blocks = [c2r(q) for q in Q_ls] + [c2r(r) for r in R_ls]
    return [(b + b.T) / 2 for b in blocks]


def iqp_line_search(Q_ls, R_ls, X_htarg, U_htarg, X_guess, U_guess, X_opt, U_opt):
    """Exact minimiser of the tracking cost along (opt - guess) (mpc.py:101-125).
    NB the reference stacks Z = [Re X.flatten(), Im X.flatten(), Re U.flatten(), Im U.flatten()]
    (row-major over (n, T+1)) but lays the cost blocks out per horizon node, so block b multiplies
    Z[2n b : 2n (b+1)] whatever those entries are.  Restated as written."""
    def pack(X, U):
        xf = np.asarray(X, dtype=complex).flatten()
        uf = np.asarray(U, dtype=complex).flatten()
        return np.concatenate([xf.real, xf.imag, uf.real, uf.imag])
    Zt, Zg, Zo = pack(X_htarg, U_htarg), pack(X_guess, U_guess), pack(X_opt, U_opt)
    DZ = Zo - Zg
    E = Zg - Zt
    num = 0.0
    den = 0.0
    pos = 0
    for blk in line_search_weights(Q_ls, R_ls):
        k = blk.shape[0]
        num += (blk @ E[pos:pos + k]) @ DZ[pos:pos + k]
        den += DZ[pos:pos + k] @ (blk @ DZ[pos:pos + k])
        pos += k
    with np.errstate(divide='ignore', invalid='ignore'):
        alpha = -num / den
    return alpha, float(np.linalg.norm(alpha * DZ))


def plant_step(x, u, H0, H_list, dt):
    """Exact solution over one step of  d rho/dt = -i [H0 + sum_k u_k H_k, rho]  with u held
    (the ODE qutip.mesolve integrates at experiment.py:202-212 under interp1d(kind='previous'),
    mpc.py:258).  x is vec_r(rho) (row-major flatten, experiment.py:211)."""
    H = np.asarray(H0, dtype=complex)
    d = H.shape[0]
    for k, Hk in enumerate(H_list):
        H = H + float(u[k]) * np.asarray(Hk, dtype=complex)
    Um = _expm(-1j * dt * H)
    rho = np.reshape(x, (d, d))
    return (Um @ rho @ Um.conj().T).reshape(-1)


def plant_step_generator(x, u, L0, L_list, dt):
    """Same step for a general (Lindblad-type) generator on vec_r(rho): x+ = expm(dt L(u)) x."""
    L = np.asarray(L0, dtype=complex)
    for k, Lk in enumerate(L_list):
        L = L + float(u[k]) * np.asarray(Lk, dtype=complex)
    return _expm(dt * L) @ np.reshape(x, -1)


class OracleQExperiment:
    """Duck type of QExperiment (experiment.py:175-212) with the exact held-control propagator."""

    def __init__(self, H0, H1_list):
        self.H0 = np.asarray(H0, dtype=complex)
        self.H1_list = [np.asarray(h, dtype=complex) for h in H1_list]

    @staticmethod
    def lift(x):
        return x

    @staticmethod
    def proj(z):
        return z

    @staticmethod
    def _held(u_fn, i, t):
        """Control held on interval i starting at time t: a callable of time, an (m, len(ts)) array whose column i
        is held on [ts[i], ts[i+1]) (interp1d kind='previous', mpc.py:258), or one control vector."""
        if callable(u_fn):
            return np.reshape(u_fn(t), -1)
        u = np.asarray(u_fn)
        return u[:, i] if u.ndim == 2 else np.reshape(u, -1)

    def simulate(self, x0, ts, u_fn):
        xs = [np.reshape(x0, -1)]
        for i, (a, b) in enumerate(zip(ts[:-1], ts[1:])):
            xs.append(plant_step(xs[-1], self._held(u_fn, i, a), self.H0, self.H1_list, b - a))
        return np.stack(xs, axis=1)


class OracleLExperiment(OracleQExperiment):
    """Same duck type for a general generator on vec_r(rho): x' = (L0 + sum_k u_k L_k) x."""

    def simulate(self, x0, ts, u_fn):
        xs = [np.reshape(x0, -1)]
        for i, (a, b) in enumerate(zip(ts[:-1], ts[1:])):
            xs.append(plant_step_generator(xs[-1], self._held(u_fn, i, a), self.H0, self.H1_list, b - a))
        return np.stack(xs, axis=1)


class OracleQCoupledExperiment(OracleQExperiment):
    """lift = [partial trace over B, partial trace over A], proj = Kronecker product (experiment.py:247-306)."""

    @staticmethod
    def lift(rhoAB_vec):
        v = np.reshape(rhoAB_vec, -1)
        dAB = int(round(np.sqrt(v.size)))
        dA = int(round(np.sqrt(dAB)))
        rho = v.reshape(dAB, dAB)
        idA = np.identity(dA)
        rhoA = np.zeros((dA, dA), dtype=complex)
        rhoB = np.zeros((dA, dA), dtype=complex)
        for i in range(dA):
            ket = np.zeros((dA, 1))
            ket[i] = 1
            rhoA += np.kron(idA, ket.T) @ rho @ np.kron(idA, ket)
            rhoB += np.kron(ket.T, idA) @ rho @ np.kron(ket, idA)
        return np.hstack([rhoA.flatten(), rhoB.flatten()])

    @staticmethod
    def proj(v):
        v = np.reshape(v, -1)
        dA = int(round(np.sqrt(v.size // 2)))
        return np.kron(v[:dA * dA].reshape(dA, dA), v[dA * dA:].reshape(dA, dA)).flatten()


def mpc(x0, dim_u, order, X_targ, U_targ, clock, experiment, model, Q, R, Qf, sat=None, du=None, max_iter=100,
        exit_condition=None, warm_start=True, qp_mode="qp", count=None, trace=None, streaming=False, solve_trace=None,
        start=None, stop=None):
    """Receding-horizon loop restating mpc.py:128-304 for streaming == False (any clock.measure_freq),
    with ``quad_program`` being the Riccati solver above (qp_mode "qp"), the lqr.py restatement
    (qp_mode "lqr") or the exact box-constrained solve (qp_mode "exact", BVLS).  Keeps the quirks: u_prev from
    U_ref at steps 0 and 1 (:185), applied control U_opt[:,0] (:250), target window lag (:276-277), exit codes 0/1/3 and the dropped last entry
    (:294-304).  ``count`` (a list) receives the number of QP solves per MPC step; ``trace`` (a list) receives
    (X_guess, U_guess) as they stand when each MPC step starts; ``solve_trace`` (a list) receives (step, X_guess,
    U_guess) at EVERY QP solve (what the reference hands to get_model_along_traj, mpc.py:175).
    Teacher forcing (tests): ``start`` = dict(step, xs, us, X_guess, U_guess) resumes at MPC step ``step`` from the given
    history (xs (n, step+1), us (m, step)) and SQP guess; ``stop`` ends the run before MPC step ``stop``."""
    exit_code = 0
    T = clock.horizon
    lift_x0 = np.asarray(experiment.lift(x0), dtype=complex)
    xs = [None] * (clock.n_steps + 1)
    us = [None] * clock.n_steps
    X_guess = np.tile(lift_x0.reshape(-1, 1), (1, T + 1))
    U_guess = np.zeros((dim_u, T))
    X_ref = np.atleast_2d(X_targ[:, :T + 1])
    U_ref = np.atleast_2d(U_targ[:, :T])
    Q_ls = [Q] * T + [Qf]
    R_ls = [R] * T
    wm = OracleWrapModel(*model.get_discrete(), dim_u, order)
    xs[0] = np.asarray(x0, dtype=complex)
    step = 0
    first = 0
    if start is not None:
        first = int(start["step"])
        for i in range(first + 1):
            xs[i] = np.asarray(start["xs"][:, i], dtype=complex)
        for i in range(first):
            us[i] = np.asarray(start["us"][:, i], dtype=float)
        X_guess = np.array(start["X_guess"], dtype=complex)
        U_guess = np.array(start["U_guess"], dtype=float)
        if first > 0:                                            # the window left behind by step first-1 (mpc.py:276-277)
            X_ref = np.atleast_2d(X_targ[:, first - 1:first + T])
            U_ref = np.atleast_2d(U_targ[:, first - 1:first - 1 + T])
    for step in range(first, clock.n_steps if stop is None else min(int(stop), clock.n_steps)):
        n_iter = 0
        done = False
        if trace is not None:
            trace.append((X_guess.copy(), U_guess.copy()))
        while not done and n_iter < max_iter:
            if solve_trace is not None:
                solve_trace.append((step, X_guess.copy(), U_guess.copy()))
            A_ls, B_ls, D_ls = wm.get_model_along_traj(X_guess, U_guess, clock.ts_horizon(step))
            u_prev = us[step - 1] if step > 1 else U_ref[:, 0].reshape(-1, 1)
            x_now = experiment.lift(xs[step])
            if qp_mode == "lqr":
                X_opt, U_opt, obj, _ = lqr_quad_program(x_now, X_ref, U_ref, Q_ls, R_ls, A_ls, B_ls, u_prev, sat, du)
            elif qp_mode == "exact":
                X_opt, U_opt, obj = exact_quad_program(x_now, X_ref, U_ref, Q_ls, R_ls, A_ls, B_ls, D_ls, u_prev, sat, du)
            else:
                X_opt, U_opt, obj, _ = quad_program(x_now, X_ref, U_ref, Q_ls, R_ls, A_ls, B_ls, D_ls, u_prev, sat, du)
            if np.isinf(obj):                                    # mpc.py:200: isinf, NOT "not isfinite" - a NaN objective
                exit_code = 3                                    # goes on (and pinv raises LinAlgError at the next solve)
                break
            if step > (1 if warm_start else np.inf):
                alpha = 1
                done = True
            else:
                alpha, new_step = iqp_line_search(Q_ls, R_ls, X_ref, U_ref, X_guess, U_guess, X_opt, U_opt)
                if new_step < 1e-4:
                    done = True
            X_guess = X_guess + alpha * (X_opt - X_guess)
            U_guess = U_guess + alpha * (U_opt - U_guess)
            n_iter += 1
        if count is not None:
            count.append(n_iter + (1 if exit_code else 0))
        if exit_code > 0:
            break
        us[step] = U_opt[:, 0]
        mf = clock.measure_freq
        if (step + 1) % mf == 0:
            # measure: re-simulate from the last measured state over the last mf intervals (mpc.py:252-260).
            # NB the held controls are stacked newest first (:257) against an increasing time grid: replayed reversed.
            ts_step = clock.ts_step(step)
            us_step = np.vstack([us[step - jq] for jq in range(mf)] + [us[step]]).T
            res = experiment.simulate(xs[step + 1 - mf], ts_step, us_step)
            xs[step + 1] = res[:, -1]
        else:
            # close the loop with the model (mpc.py:261-267)
            lx = np.reshape(experiment.lift(xs[step]), (-1, 1))
            lu = wm.lift_u(us[step].reshape(-1, 1))
            xs[step + 1] = np.reshape(experiment.proj(model.predict(lx, krtimes(lu, lx))), -1)
        X_guess = shift_guess(X_guess)
        U_guess = shift_guess(U_guess)
        X_ref = np.atleast_2d(X_targ[:, step:step + T + 1])
        U_ref = np.atleast_2d(U_targ[:, step:step + T])
        if streaming:
            # mpc.py:281-285: the model object is refitted with this step's transition; the loop keeps linearising the
            # operators it extracted at entry (quirk Q6)
            lx = np.reshape(experiment.lift(xs[step]), (-1, 1))
            model.fit_iteration(np.reshape(experiment.lift(xs[step + 1]), (-1, 1)), lx,
                                krtimes(wm.lift_u(us[step].reshape(-1, 1)), lx))
        if exit_condition is not None and exit_condition(xs[step + 1], xs[step], us[step]):
            exit_code = 1
            break
    if trace is not None:
        trace.append((X_guess.copy(), U_guess.copy()))
    if exit_code == 0:
        clock.set_endsim(step + 1)
        return [np.vstack(xs[:step + 2]).T, np.vstack(us[:step + 1]).T], model, exit_code
    clock.set_endsim(step)
    if step == 0:
        return [np.vstack(xs[:step + 1]).T, None], model, exit_code
    return [np.vstack(xs[:step + 1]).T, np.vstack(us[:step]).T], model, exit_code


def mpc_batch(x0s, A_models, dim_u, order, X_targ, U_targ, dt, horizon, n_steps, H0s, H_list, Q, R, Qf,
              sat, du, max_iter=100, warm_start=True, qp_mode="qp", trace=None, generator_plant=False, measure_freq=1):
    """Loop ``mpc`` over an ensemble.  x0s (B, n); A_models (B or 1, n, n(1+P)); H0s (B or 1, d, d).
    Returns xs (B, n, n_steps+1) (NaN-padded after an early exit), us (B, m, n_steps),
    exit codes (B,), QP solves per MPC step (B, n_steps)."""
    x0s = np.asarray(x0s, dtype=complex)
    A_models = np.asarray(A_models, dtype=complex)
    H0s = np.asarray(H0s, dtype=complex)
    B, n = x0s.shape
    xs_out = np.full((B, n, n_steps + 1), np.nan + 0j)
    us_out = np.full((B, dim_u, n_steps), np.nan)
    codes = np.zeros(B, dtype=np.int32)
    solves = np.zeros((B, n_steps), dtype=np.int32)
    for b in range(B):
        Am = A_models[b if A_models.shape[0] > 1 else 0]
        model = OracleDMDc(n, n, Am.shape[1] - n, Am)
        Hl = np.asarray(H_list)
        Hl = Hl[b] if Hl.ndim == 4 else Hl                       # per-member plant operators [B, m, k, k]
        X_t = X_targ[b] if np.ndim(X_targ) == 3 else X_targ      # per-member targets [B, n, cols]
        U_t = U_targ[b] if np.ndim(U_targ) == 3 else U_targ
        exp = (OracleLExperiment if generator_plant else OracleQExperiment)(H0s[b if H0s.shape[0] > 1 else 0], list(Hl))
        clock = OracleClock(dt, horizon, n_steps)
        clock.measure_freq = measure_freq
        cnt = []
        tr = [] if trace is not None else None
        (xs, us), _, code = mpc(x0s[b], dim_u, order, X_t, U_t, clock, exp, model, Q, R, Qf, sat=sat, du=du,
                                max_iter=max_iter, warm_start=warm_start, qp_mode=qp_mode, count=cnt, trace=tr)
        if trace is not None:
            trace.append(tr)
        xs_out[b, :, :xs.shape[1]] = xs
        if us is not None:
            us_out[b, :, :us.shape[1]] = us
        codes[b] = code
        solves[b, :len(cnt)] = cnt
    return xs_out, us_out, codes, solves
