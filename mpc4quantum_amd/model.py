"""DMDc linear control models with the reference's interface (mpc4quantum/model.py).

`DMDc` is the read-only container the MPC loop needs (`get_discrete`, `predict`).  `DiscrepDMDc` and `OnlineDMDc`
are the two streaming refits `mpc(..., streaming=True)` feeds through `fit_iteration` (mpc.py:281-285).  They are
host-side NumPy on (n x n(1+P)) matrices, as in the reference: the refit is outside the accelerated path (the loop
linearises the model it was handed at entry, mpc.py:156, quirk Q6), so nothing here touches the GPU."""
import numpy as np


class DMDc:
    """model.py:7-103: y = A_x x + A_u u with A = [A_x | A_u] of shape (dim_y, dim_x + dim_u)."""

    def __init__(self, dim_y, dim_x, dim_u, A0):
        self.dim_y = dim_y
        self.dim_x = dim_x
        self.dim_u = dim_u
        self.A = np.asarray(A0)
        self.discount = 1        # weight of old data per update; half-life k updates <=> 2 ** (-1 / k)
        self.rcond = 1e-15       # pinv cut-off

    @classmethod
    def from_data(cls, Y, X, U, **kwargs):
        raise NotImplementedError()

    @classmethod
    def from_bootstrap(cls, dim_y, dim_x, dim_u, A0, **kwargs):
        raise NotImplementedError()

    @classmethod
    def from_randn(cls, dim_y, dim_x, dim_u, **kwargs):
        raise NotImplementedError()

    def fit_iteration(self, next_y, next_x, next_u):
        raise NotImplementedError()

    def predict(self, current_x, current_u):
        A_x, A_u = self.get_discrete()
        return A_x @ np.reshape(current_x, (self.dim_x, -1)) + A_u @ np.reshape(current_u, (self.dim_u, -1))

    def get_discrete(self):
        return self.A[:self.dim_y, :self.dim_x], self.A[:self.dim_y, self.dim_x:]


def _stacked_inputs(X, U):
    """Z = [X; U] (U may be absent), and dim_u."""
    if U is None:
        return X, 0
    return np.vstack([X, U]), U.shape[0]


class _History:
    """Optional record of the model every `_isave` updates (`_save`), as the reference keeps in `iA` / `iP`."""

    def _init_history(self):
        self._save = False
        self._iteration = 0
        self._isave = 10

    def _tick(self):
        self._iteration += 1
        return self._save and self._iteration % self._isave == 0


class DiscrepDMDc(DMDc, _History):
    """model.py:109-213.  Keeps the (discounted) snapshot matrices and, at every update, adds a least-squares fit of
    the current residual Y - A [X; U] to the model - so it can start from any initial model (bootstrap)."""

    def __init__(self, dim_y, dim_x, dim_u, A0, **kwargs):
        super().__init__(dim_y, dim_x, dim_u, A0)
        self.initialization = kwargs
        self.Y = kwargs.get('Y')
        self.X = kwargs.get('X')
        self.U = kwargs.get('U')
        self.discount = kwargs.get('discount', self.discount)
        self.rcond = kwargs.get('rcond', self.rcond)
        self.min_rank = dim_x              # no refit until the state snapshots span the state space
        self.iA = [A0]
        self._init_history()

    @classmethod
    def from_randn(cls, dim_y, dim_x, dim_u, **kwargs):
        """Random real model of standard deviation kwargs['sigma']."""
        sigma = kwargs['sigma']
        return cls(dim_y, dim_x, dim_u, sigma * np.random.randn(dim_y, dim_x + dim_u), sigma=sigma)

    @classmethod
    def from_bootstrap(cls, dim_y, dim_x, dim_u, A0, **kwargs):
        return cls(dim_y, dim_x, dim_u, A0)

    @classmethod
    def from_data(cls, Y, X, U=None, **kwargs):
        """A0 = Y pinv([X; U]) with the cut-off kwargs['rcond'] (relative to the largest singular value)."""
        rcond = kwargs['rcond']
        Z, dim_u = _stacked_inputs(X, U)
        return cls(Y.shape[0], X.shape[0], dim_u, Y @ np.linalg.pinv(Z, rcond=rcond), Y=Y, X=X, U=U, rcond=rcond)

    @staticmethod
    def _update_stack(val, stack, discount, nadd=1):
        val = np.reshape(val, (-1, nadd))
        return val if stack is None else np.hstack([discount * stack, val])

    def fit_iteration(self, next_y, next_x, next_u=np.array([])):
        self.Y = self._update_stack(next_y, self.Y, self.discount)
        self.X = self._update_stack(next_x, self.X, self.discount)
        self.U = self._update_stack(next_u, self.U, self.discount)
        if np.linalg.matrix_rank(self.X) >= self.min_rank:
            residual = self.Y - self.predict(self.X, self.U)
            self.A = self.A + residual @ np.linalg.pinv(np.vstack([self.X, self.U]), rcond=self.rcond)
        if self._tick():
            self.iA.append(np.copy(self.A))
        return self.get_discrete()

    def append(self, Y, X, U):
        """Add snapshots without discounting or refitting."""
        nadd = Y.shape[1]
        self.Y = self._update_stack(Y, self.Y, 1, nadd)
        self.X = self._update_stack(X, self.X, 1, nadd)
        self.U = self._update_stack(U, self.U, 1, nadd)


class OnlineDMDc(DMDc, _History):
    """model.py:216-313: recursive least squares (Zhang et al., online DMD).  P tracks the inverse of the input Gram
    matrix; one update is a rank-one correction of A and P.  The products use plain (unconjugated) transposes on
    complex data, as the reference does."""

    def __init__(self, dim_y, dim_x, dim_u, P0, A0, **kwargs):
        super().__init__(dim_y, dim_x, dim_u, A0)
        self.initialization = kwargs
        self.P = P0
        self.iP = [P0]
        self.iA = [A0]
        self._init_history()

    @classmethod
    def from_randn(cls, dim_y, dim_x, dim_u, **kwargs):
        """Random real model (kwargs['sigma']) with P0 = kwargs['alpha'] * I."""
        dim_z = dim_x + dim_u
        return cls(dim_y, dim_x, dim_u, kwargs['alpha'] * np.identity(dim_z), kwargs['sigma'] * np.random.randn(dim_y, dim_z),
                   **kwargs)

    @classmethod
    def from_bootstrap(cls, dim_y, dim_x, dim_u, A0, **kwargs):
        """Start from A0 with P0 = kwargs['alpha'] * I (alpha = how fast new data overrides A0; try 1e2)."""
        return cls(dim_y, dim_x, dim_u, kwargs['alpha'] * np.identity(dim_x + dim_u), A0, **kwargs)

    @classmethod
    def from_data(cls, Y, X, U=None, **kwargs):
        Z, dim_u = _stacked_inputs(X, U)
        P0 = np.linalg.pinv(Z @ Z.T)
        return cls(Y.shape[0], X.shape[0], dim_u, P0, Y @ Z.T @ P0, Y=Y, X=X, U=U)

    def fit_iteration(self, next_y, next_x, next_u=np.array([])):
        y = np.reshape(next_y, (-1, 1))
        z = np.vstack([np.reshape(next_x, (-1, 1)), np.reshape(next_u, (-1, 1))])
        Pz = self.P @ z
        gamma = 1 / (1 + z.T @ Pz)
        self.A = self.A + gamma * (y - self.A @ z) @ Pz.T
        self.P = (self.P - gamma * Pz @ Pz.T) / self.discount
        if self._tick():
            self.iA.append(np.copy(self.A))
            self.iP.append(np.copy(self.P))
        return self.get_discrete()
