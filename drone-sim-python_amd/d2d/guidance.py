"""Mirror of the guidance classes of src/d2d/guidance.py used by the full simulations:
WindField, DiffFlatness, DFFFController, DCFController, CircleTraj, GVFcontroller.  Numerics run
through libd2dhip.so.  (The pure-pursuit controllers of the reference are legacy paths outside this
engine's scope)"""
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


class DFFFController:
    """Differential-flatness feed-forward + 3-state LQR feedback (src/d2d/guidance.py:52-95).
    `get(X, t)` is one d2d_dfff_eval call; `get_batch` evaluates many aircraft at once."""

    def __init__(self, traj, ac, wind):
        self.traj, self.ac, self.wind = traj, ac, wind
        self.dt = 0.01
        self.time = np.arange(0, traj.duration, self.dt)
        self.carrot, self.ref_pos = [0, 0], [0, 0]
        self.Xref = []
        self.K = []

    def get(self, X, t):
        Yref = np.asarray(self.traj.get(t), dtype=np.float64)
        W = self.wind.sample(t, Yref[0])
        U, K, Xr = DFFFController.get_batch(np.asarray(X, float)[None], Yref[None], W, self.ac)
        self.Xref.append(Xr[0])
        self.K.append(K[0])
        return U[0]

    @staticmethod
    def get_batch(X, Yref, W, ac):
        """X (n,5), Yref (n,>=3,2) [derivative, axis], one wind W for the batch -> U (n,2), K (n,2,5), Xr (n,5)."""
        ctx = d2dhip.default_context()
        X = np.asarray(X, dtype=np.float64); Yref = np.asarray(Yref, dtype=np.float64)
        n = X.shape[0]
        Y = np.ascontiguousarray(Yref[:, :3, :].reshape(n, 6).T)          # rows x,y,xd,yd,xdd,ydd
        Xr, U, K1 = ctx.dfff_eval(ctx.dev(np.ascontiguousarray(X.T)), ctx.dev(Y), (float(W[0]), float(W[1])),
                                  ac.tau_phi, ac.tau_v)
        K = np.zeros((n, 2, 5))
        K[:, :, :3] = K1.cpu().numpy().T.reshape(n, 2, 3)
        return U.cpu().numpy().T.copy(), K, Xr.cpu().numpy().T.copy()

    def draw_debug(self, _f, _a, time):
        Xref = np.array(self.Xref)
        _a[0, 0].plot(time, Xref[:, 0])


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
