"""Model construction (mpc4quantum/vectorize.py).  Host NumPy; the batched HIP version of this
row is listed as next-tier in SURVEY.md 8(f)."""
import itertools
import math

import numpy as np

from .library import create_power_list


def discretize_homogeneous(A_cts_list, dt, order):
    """Dyson/Taylor expansion of exp(dt (A_0 + sum u_k A_k)) to `order`, collected per control
    monomial: n x n(1+P) complex (vectorize.py:8-49).  Accepts a leading ensemble axis on every
    operator ([B, n, n]) and then returns [B, n, n(1+P)]."""
    ops = [np.asarray(a, dtype=complex) for a in A_cts_list]
    batched = ops[0].ndim == 3
    if not batched:
        ops = [a[None] for a in ops]
    Bn, n = ops[0].shape[0], ops[0].shape[-1]
    m = len(ops) - 1
    keys = {tuple(int(v) for v in p): i for i, p in enumerate(create_power_list(order, m))}
    out = np.zeros((len(keys), Bn, n, n), dtype=complex)
    eye = np.broadcast_to(np.identity(n, dtype=complex), (Bn, n, n))
    # words of length k grown from words of length k-1: prod(word + [c]) = prod(word) @ A_c
    level = {(): eye}
    for k in range(order + 1):
        scale = dt ** k / math.factorial(k)
        nxt = {}
        for word, prod in level.items():
            counts = tuple(sum(1 for w in word if w == c) for c in range(1, m + 1))
            if counts not in keys:
                raise ValueError('Error in discretization. Control powers should contribute uniquely.')
            out[keys[counts]] += scale * prod
            if k < order:
                for c in range(m + 1):
                    nxt[word + (c,)] = prod @ ops[c]
        level = nxt
    res = np.concatenate(list(out), axis=-1)
    return res if batched else res[0]


def discretize_homogeneous_batch(A_cts_list, dt, order, scales=None):
    """GPU form of discretize_homogeneous for an ensemble (HIP kernel `discretize_kernel` through
    m4q_discretize_batch).  Operators [n, n] (shared) or [B, n, n] (per member); ``scales`` [B, 1+m] multiplies
    operator k of member b (parameter ensembles over a few fixed operators).  Returns [B, n, n(1+P)]."""
    from . import _lib
    ops = [np.asarray(a, dtype=np.complex128) for a in A_cts_list]
    per = any(a.ndim == 3 for a in ops)
    n = ops[0].shape[-1]
    m = len(ops) - 1
    if per:
        Bn = max(a.shape[0] for a in ops if a.ndim == 3)
        gens = np.stack([np.broadcast_to(a, (Bn, n, n)) for a in ops], axis=1)
    else:
        gens = np.stack(ops)[None]
    if scales is not None:
        scales = np.ascontiguousarray(scales, dtype=np.float64)
        Bn = scales.shape[0]
    elif not per:
        Bn = 1
    L = _lib.lib()
    P = L.m4q_library_size(order, m)
    out = np.empty((Bn, n, n * (1 + P)), dtype=np.complex128)
    gens = np.ascontiguousarray(gens)
    _lib.check(L.m4q_discretize_batch(Bn, n, m, order, float(dt), gens.ctypes.data_as(_lib._dp), 1 if per else 0,
                                      scales.ctypes.data_as(_lib._dp) if scales is not None else None,
                                      out.ctypes.data_as(_lib._dp)))
    return out


def liouvillian(H):
    """Generator of d/dt vec_r(rho) = -i [H, rho] in the |i><j| basis, i-major:
    -i (H (x) I - I (x) H^T).  Equals vectorize_me(H, [|i><j|]) (vectorize.py:52-75)."""
    H = np.asarray(H, dtype=complex)
    d = H.shape[-1]
    eye = np.identity(d)
    if H.ndim == 2:
        return -1j * (np.kron(H, eye) - np.kron(eye, H.T))
    return -1j * (np.einsum('bij,kl->bikjl', H, eye) - np.einsum('ij,blk->bikjl', eye, H)).reshape(-1, d * d, d * d)


def vectorize_me(H, measure_list):
    """Projection of -i[H, .] on an operator basis through its structure constants
    (vectorize.py:52-75).  Operators are ndarrays or anything with .full() (qutip.Qobj)."""
    def arr(x):
        return np.asarray(x.full() if hasattr(x, "full") else x, dtype=complex)
    H = arr(H)
    basis = [arr(s) for s in measure_list]
    dm = len(basis)
    h = np.array([np.vdot(H, s) for s in basis])             # tr(H^dag s)
    A = np.zeros((dm, dm), dtype=complex)
    for i in range(dm):
        if h[i] == 0:
            continue
        for k in range(dm):
            if i == k:
                continue
            comm = basis[i] @ basis[k] - basis[k] @ basis[i]
            for j in range(dm):
                A[k, j] += -1j * h[i] * np.vdot(comm, basis[j])   # tr([s_i, s_k]^dag s_j)
    return A
