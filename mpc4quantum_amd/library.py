"""Control-monomial library helpers with the reference's names (mpc4quantum/linearize.py:80-164).

Host-side integer tables only; the kernels carry the same tables as compile-time constants
(csrc/m4q_mpc.h PowTab) and m4q_power_list() exposes them so tests can compare the two.
"""
import numpy as np


def create_power_list(order, dimension):
    """Exponent vectors of all monomials of total degree <= order, constant first.  Same order as
    linearize.create_power_list: the last variable's exponent varies slowest."""
    out = []

    def rec(k, left, cur):
        if k < 0:
            out.append(np.array(cur, dtype=int))
            return
        for e in range(left + 1):
            cur[k] = e
            rec(k - 1, left - e, cur)

    rec(dimension - 1, order, [0] * dimension)
    return out


def multinomial_powers(n, k):
    """Exponent vectors of (x_1+...+x_k)^n in the reference's iteration order (linearize.py:92-110)."""
    for p in create_power_list(n, k):
        if int(p.sum()) == n:
            yield p[::-1]


def size_of_library(order, dimension):
    return len(create_power_list(order, dimension))


def _monomial(powers):
    powers = np.asarray(powers, dtype=int)

    def fn(x, ps=powers):
        x = np.asarray(x)
        val = np.ones_like(x[0, :], dtype=float)
        for i, p in enumerate(ps):
            val = val * (np.zeros_like(x[i, :]) if p < 0 else np.power(x[i, :], p))
        return val
    return fn


def create_library_from_list(power_list):
    return [_monomial(p) for p in power_list]


def create_library(order, dimension):
    """List of callables x (dimension, k) -> (k,), one per monomial, constant first."""
    return create_library_from_list(create_power_list(order, dimension))


def diff_library(order, dimension):
    """(derivative libraries per variable, derivative coefficients per variable), linearize.py:143-164."""
    plist = np.vstack(create_power_list(order, dimension)[1:])
    fns, coefs = [], []
    for k in range(dimension):
        unit = np.zeros(dimension, dtype=int)
        unit[k] = 1
        fns.append(create_library_from_list(plist - unit))
        coefs.append(plist[:, [k]])
    return fns, coefs


def krtimes(A, B):
    """Column-wise Khatri-Rao product; row p*n + j pairs A[p] with B[j] (linearize.py:80-89)."""
    A = np.asarray(A)
    B = np.asarray(B)
    if A.shape[1] != B.shape[1]:
        raise ValueError("Cols of A =/ Cols of B")
    return np.einsum('pk,jk->pjk', A, B).reshape(A.shape[0] * B.shape[0], A.shape[1])
