"""Mirror of src/Controllers.py: reference circle, flatness map with jerk terms and the
flatness feed-forward + LQR tracking controller.  ComputeFlatness / ComputeGain run on the
GPU (d2d_flatness variant 1 / d2d_ctrl_gain: flatness, linearisation, 5x5 Riccati solve and
saturations in one kernel)."""
import numpy as np

import d2dhip
import d2d.dynamic as ddyn


class CircleTraj:
    """Circular reference of radius r flown at speed v (src/Controllers.py:17-32)."""

    def __init__(self, v, r=40, c=[0, 0]):
        self.c, self.r, self.v = c, r, v
        self.omega = self.v / self.r

    def TrajPoints(self, t):
        th, r, om = self.omega * t, self.r, self.omega
        s, c = np.sin(th), np.cos(th)
        # the third derivative keeps the reference's (unscaled) expression, :31
        return [r * c, r * s], [-r * s * om, r * c * om], [-r * c * om ** 2, -r * s * om ** 2], [r * s, -r * c]


class DiffFlatness:
    def __init__(self, w=[0, 0]):
        self.w = w
        self.g = 9.81
        self.x_i, self.y_i, self.psi_i, self.phi_i, self.v_i = 0, 1, 2, 3, 4

    def ComputeFlatness(self, t, Y_ref, Yd_ref, Ydd_ref, Yddd_ref):
        ctx = d2dhip.default_context()
        ac = ddyn.Aircraft()
        Y = np.array([Y_ref[0], Y_ref[1], Yd_ref[0], Yd_ref[1], Ydd_ref[0], Ydd_ref[1], Yddd_ref[0], Yddd_ref[1]],
                     dtype=np.float64).reshape(8, 1)
        X, U, _ = ctx.flatness(1, ctx.dev(Y), (float(self.w[0]), float(self.w[1])), ac.tau_phi, ac.tau_v)
        return X.cpu().numpy()[:, 0], U.cpu().numpy()[:, 0]


class DiffController:
    def __init__(self, w=[0, 0]):
        self.w = w
        self.DF = DiffFlatness(self.w)
        self.psi_i, self.phi_i = 2, 3
        self.err_sats = np.array([20, 20, np.pi / 3, np.pi / 4, 1])
        self.v_min, self.v_max = 4, 20
        self.phi_lim = np.deg2rad(60)
        self.Q, self.R = [1, 1, 0.1, 0.01, 0.01], [8, 1]
        self.K = []

    def RestrictAngle(self, theta):
        return (theta + np.pi) % (2 * np.pi) - np.pi

    def ComputeGain(self, t, X, Y_ref, Yd_ref, Ydd_ref, Yddd_ref, ac):
        ctx = d2dhip.default_context()
        Y = np.array([Y_ref[0], Y_ref[1], Yd_ref[0], Yd_ref[1], Ydd_ref[0], Ydd_ref[1], Yddd_ref[0], Yddd_ref[1]],
                     dtype=np.float64).reshape(8, 1)
        # the reference's flatness map always uses a fresh Aircraft() for tau (src/Controllers.py:100);
        # the linearisation uses `ac` (:170).  Both default to the same constants.
        Xr, dX, U, K = ctx.ctrl_gain(ctx.dev(np.asarray(X, dtype=np.float64).reshape(5, 1)), ctx.dev(Y),
                                     w=(float(self.w[0]), float(self.w[1])), tau_phi=ac.tau_phi, tau_v=ac.tau_v,
                                     err_sats=tuple(self.err_sats), v_min=self.v_min, v_max=self.v_max,
                                     phi_lim=self.phi_lim, Q=tuple(self.Q), R=tuple(self.R))
        self.K.append(K.cpu().numpy()[:, 0].reshape(2, 5))
        return Xr.cpu().numpy()[:, 0], dX.cpu().numpy()[:, 0], U.cpu().numpy()[:, 0]
