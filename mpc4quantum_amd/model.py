"""Read-only DMDc container (mpc4quantum/model.py:7-103).  Streaming refits (DiscrepDMDc,
OnlineDMDc) are outside the accelerated path (streaming=False everywhere)."""
import numpy as np


class DMDc:
    def __init__(self, dim_y, dim_x, dim_u, A0):
        self.dim_y = dim_y
        self.dim_x = dim_x
        self.dim_u = dim_u
        self.A = np.asarray(A0)
        self.discount = 1
        self.rcond = 1e-15

    def fit_iteration(self, next_y, next_x, next_u):
        raise NotImplementedError()

    def predict(self, current_x, current_u):
        A_x, A_u = self.get_discrete()
        return A_x @ np.reshape(current_x, (self.dim_x, -1)) + A_u @ np.reshape(current_u, (self.dim_u, -1))

    def get_discrete(self):
        return self.A[:self.dim_y, :self.dim_x], self.A[:self.dim_y, self.dim_x:]
