"""Synthetic ensembles for the BASELINE.json configurations (parameters from the reference's tests,
SURVEY.md section 8d: tests/test_mpc4quantum.py:399-466 (CNOT/d=4), :504-564 (DRAG/d=3), :607-670
(NOT/d=2) and tests/util_qubits.py).  Plain NumPy, no qutip."""
import numpy as np

from .vectorize import discretize_homogeneous, liouvillian

SX = np.array([[0, 1], [1, 0]], dtype=complex)
SY = np.array([[0, -1j], [1j, 0]], dtype=complex)
SZ = np.array([[1, 0], [0, -1]], dtype=complex)
I2 = np.identity(2, dtype=complex)


def rx(theta):
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return np.array([[c, -1j * s], [-1j * s, c]])


def _proj(d, i):
    p = np.zeros((d, d), dtype=complex)
    p[i, i] = 1
    return p


def _targets(target_state, n_steps, horizon, dim_u):
    cols = n_steps + horizon + 1
    X = np.tile(np.reshape(target_state, (-1, 1)), (1, cols))
    U = np.zeros((dim_u, cols - 1))
    return X, U


def build(config, batch=None, order=1, horizon=None, n_steps=None, offset=0, total=None, host_models=True, drift_scale=1.0,
          r_scale=1.0):
    """Returns a dict with: name, dim_x, dim_u, order, dt, horizon, n_steps, sat, du, Q, R, Qf, x0 [B,n],
    models [B|1,n,n(1+P)], X_targ (n,cols), U_targ (m,cols-1), plant_op0 [1|B,d,d], plant_ops [1|B,m,d,d],
    generators [1+m,n,n] and scales [B,1+m] (the continuous-time operators the models come from; models is None when
    host_models=False: build them on the device).  (offset, total): this call returns members [offset, offset+batch)
    of a `total`-member draw, so ranks of a sharded run see disjoint slices of ONE ensemble.
    (drift_scale, r_scale), configs 3 and 5 only: the anharmonicity alpha0 (model AND plant) and the control weight R are
    multiplied by them.  The order-1 truncation of exp(dt L) is not norm preserving (|1 + i dt alpha0| = 1.18 per step at the
    reference's alpha0, 6e5 over T = 80: the horizon QP is then beyond fp64); drift_scale = 1/8 gives 1.003 per step
    and r_scale = 100 keeps the controls off their bounds - the well-conditioned T = 80 case the parity tests pin strictly."""
    config = int(config)
    if config in (1, 2):
        d, m = 2, 1
        dt = 1.0
        T = 10 if config == 1 else 20
        ns = 20
        B = 1 if config == 1 else 8192
        sat = 2 * np.pi * 0.1
        du = 0.5 * sat
        wq = 2 * np.pi * 4
        H_model = [0.5 * (wq - wq) * SZ, 0.5 * SX]                        # util_qubits.py:77-79, wQ == wR
        H_plant0 = 0.5 * (0.99 * wq - wq) * SZ                           # test_mpc4quantum.py:638: detuned plant
        Qm = np.diag([1.0, 0, 0, 1.0])
        R = 1e-2 / sat ** 2 * np.identity(m)
        target = _proj(2, 1).reshape(-1)
        B = batch or B
        if config == 1:
            r = rx(1e-4)
            rho0 = r @ _proj(2, 0) @ r.conj().T
            x0 = np.tile(rho0.reshape(1, -1), (B, 1))
        else:
            rng = np.random.default_rng(2)
            psi = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
            psi /= np.linalg.norm(psi, axis=1, keepdims=True)
            x0 = np.einsum('bi,bj->bij', psi, psi.conj()).reshape(B, -1)
        gens = np.stack([liouvillian(h) for h in H_model])
        scales = None
        if config == 2:
            rng = np.random.default_rng(2)
            tot = total or B
            psi = rng.standard_normal((tot, 2)) + 1j * rng.standard_normal((tot, 2))
            psi = psi[offset:offset + B]
            psi /= np.linalg.norm(psi, axis=1, keepdims=True)
            x0 = np.einsum('bi,bj->bij', psi, psi.conj()).reshape(B, -1)
        models = discretize_homogeneous(list(gens), dt, order)[None]
        plant0, plantk = H_plant0[None], np.stack([H_model[1]])[None]
    elif config in (3, 5):
        d, m = 3, 2
        dt = 0.25
        T = 40 if config == 3 else 80
        ns = 20
        B = batch or (65536 if config == 3 else 2 ** 20)
        sat = 2 * np.pi * 0.25
        du = 0.5 * sat
        alpha0 = -2 * np.pi * 0.1 / dt * drift_scale
        a = np.diag(np.sqrt(np.arange(1, 3)), 1).astype(complex)
        HX = 0.5 * (a.conj().T + a)
        HY = 0.5j * (a.conj().T - a)
        P2 = _proj(3, 2)
        Qm = np.zeros((9, 9))
        Qm[0, 0] = Qm[4, 4] = 1.0
        R = r_scale * 1e-3 / sat ** 2 * np.identity(m)
        rho0 = _proj(3, 0)
        r = rx(1e-4)
        rho0[:2, :2] = r.conj().T @ rho0[:2, :2] @ r
        x0 = np.tile(rho0.reshape(1, -1), (B, 1))
        target = _proj(3, 1).reshape(-1)
        rng = np.random.default_rng(3 if config == 3 else 5)
        tot = total or B
        xi = rng.standard_normal(tot)[offset:offset + B]
        zeta = rng.standard_normal(tot)[offset:offset + B]
        gens = np.stack([alpha0 * liouvillian(P2), liouvillian(HX), liouvillian(HY)])
        scales = np.stack([1 + 0.05 * xi, 1 + 0.02 * zeta, 1 + 0.02 * zeta], axis=1)
        models = None
        if host_models:
            models = discretize_homogeneous([scales[:, k, None, None] * gens[k][None] for k in range(3)], dt, order)
        plant0, plantk = (alpha0 * P2)[None], np.stack([HX, HY])[None]
    elif config == 4:
        d, m = 4, 3
        dt = 0.25
        T, ns = 40, 20
        B = batch or 65536
        sat = 2 * np.pi * 0.05
        du = sat
        H0 = np.kron(SZ, SZ)
        Hk = [np.kron(SY, I2), np.kron(I2, SY), np.kron(SZ, I2)]
        Qm = np.zeros((16, 16))
        for i in (0, 5, 10, 15):
            Qm[i, i] = 1.0
        R = 1e-3 * np.identity(m)
        r1, r2 = rx(-1e-2), rx(1e-2)
        rho0 = np.kron(r1 @ _proj(2, 0) @ r1.conj().T, r2 @ _proj(2, 0) @ r2.conj().T)
        x0 = np.tile(rho0.reshape(1, -1), (B, 1))
        target = np.kron(_proj(2, 0), _proj(2, 1)).reshape(-1)
        rng = np.random.default_rng(4)
        tot = total or B
        J = 1 + 0.1 * rng.standard_normal(tot)[offset:offset + B]
        gens = np.stack([liouvillian(H0)] + [liouvillian(h) for h in Hk])
        scales = np.concatenate([J[:, None], np.ones((B, 3))], axis=1)
        models = None
        if host_models:
            models = discretize_homogeneous([scales[:, k, None, None] * gens[k][None] for k in range(4)], dt, order)
        plant0, plantk = H0[None], np.stack(Hk)[None]
    else:
        raise ValueError("config must be 1..5")
    if (drift_scale != 1.0 or r_scale != 1.0) and config not in (3, 5):
        raise ValueError("drift_scale / r_scale apply to configs 3 and 5")
    T = horizon or T
    ns = n_steps or ns
    X_targ, U_targ = _targets(target, ns, T, m)
    return dict(name="config%d" % config, dim_x=d * d, dim_u=m, d=d, order=order, dt=dt, horizon=T, n_steps=ns, sat=sat,
                du=du, Q=Qm, R=R, Qf=Qm.copy(), x0=np.ascontiguousarray(x0),
                models=None if models is None else np.ascontiguousarray(models), generators=gens, scales=scales,
                X_targ=X_targ, U_targ=U_targ, plant_op0=plant0, plant_ops=plantk, batch=B)
