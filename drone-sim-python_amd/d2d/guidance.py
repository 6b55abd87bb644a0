"""Mirror of the guidance classes of src/d2d/guidance.py used by the full simulations:
WindField, DiffFlatness, DCFController, CircleTraj, GVFcontroller.  Numerics run through
libd2dhip.so.  (DFFFController and the pure-pursuit controllers of the reference are legacy
paths outside this engine's scope)"""
import numpy as np

import d2dhip
from d2d.dynamic import Aircraft


def norm_mpi_pi(v):
    return (v + np.pi) % (2 * np.pi) - np.pi


class WindField:
    def __init__(self, w=[0., 0.]):
        self.w = w

    def sample(self, t, loc):
        return self.w

    def summarize(self):
        return f'{self.w} m/s'


class DiffFlatness:
    def state_and_input_from_output(Ys, W, ac):
        """Ys[(derivative),(axis)] -> X(5), U(2), Xdot(5) (src/d2d/guidance.py:22-47)."""
        ctx = d2dhip.default_context()
        Ys = np.asarray(Ys, dtype=np.float64)
        Y = np.zeros((8, 1))
        for d in range(min(4, Ys.shape[0])):
            Y[2 * d, 0], Y[2 * d + 1, 0] = Ys[d, 0], Ys[d, 1]
        X, U, Xd = ctx.flatness(0, ctx.dev(Y), (float(W[0]), float(W[1])), ac.tau_phi, ac.tau_v)
        return X.cpu().numpy()[:, 0], U.cpu().numpy()[:, 0], Xd.cpu().numpy()[:, 0]


class DCFController:
    """Distributed circular-formation phase controller (src/d2d/guidance.py:99-126)."""

    def __init__(self):
        pass

    def get(self, n_ac, B, c, p, z_des, kr):
        z_des.shape = (len(z_des), 1)          # the reference reshapes the caller's array in place (:104)
        ctx = d2dhip.default_context()
        c = np.asarray(c, dtype=np.float64); p = np.asarray(p, dtype=np.float64)
        Ur, eth = ctx.dcf_eval(ctx.dev(np.ascontiguousarray(c.T)), ctx.dev(np.ascontiguousarray(p)), n_ac, B,
                               z_des[:, 0], float(kr))
        return Ur.cpu().numpy().reshape(n_ac, 1), eth.cpu().numpy()[:n_ac - 1].reshape(n_ac - 1, 1)


class CircleTraj:
    """Level set of a circle: e, grad, Hessian (src/d2d/guidance.py:133-146)."""

    def __init__(self, c=np.array([0, 0])):
        self.c = c

    def get(self, X, r=1):
        px, py = X[0] - self.c[0], X[1] - self.c[1]
        return np.asarray(px ** 2 + py ** 2 - r ** 2), np.asarray([2 * px, 2 * py]), np.asarray([[2, 0], [0, 2]])


class GVFcontroller:
    """Guiding-vector-field heading-rate controller (src/d2d/guidance.py:148-181)."""

    def __init__(self, traj, ac, wind):
        self.traj, self.ac, self.wind = traj, ac, wind

    def get(self, X, ke, kd, e, n, H):
        ctx = d2dhip.default_context()
        e = float(np.asarray(e).reshape(-1)[0])
        U = ctx.gvf_eval(ctx.dev(np.asarray(X, dtype=np.float64).reshape(5, 1)), ctx.dev(np.array([e])),
                         ctx.dev(np.asarray(n, dtype=np.float64).reshape(2, 1)),
                         ctx.dev(np.asarray(H, dtype=np.float64).reshape(4, 1)), float(ke), float(kd)).cpu().numpy()[:, 0]
        return U[0], U[1], U[2]
