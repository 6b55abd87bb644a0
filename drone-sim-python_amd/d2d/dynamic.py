"""Mirror of src/d2d/dynamic.py: the 5-state planar aircraft.  disc_dyn and cont_jac run
on the GPU (d2d_step / d2d_cont_jac); a whole simulation should use full_sim.* instead,
which keeps the time loop on the device."""
import numpy as np

import d2dhip


def _wind_of(W, t, loc):
    if hasattr(W, 'sample'):
        return W.sample(t, loc)
    return W if W is not None else (0.0, 0.0)


class Aircraft:
    i_phi, i_va, i_size = np.arange(3)
    s_x, s_y, s_psi, s_phi, s_va, s_size = np.arange(6)
    s_slice_pos = slice(s_x, s_y + 1)
    g = 9.81

    def __init__(self):
        self.tau_phi = 0.01      # src/d2d/dynamic.py:11 (0.9667 is the commented alternative)
        self.tau_v = 1.

    def cont_dyn(self, X, t, U, W):
        """Model right-hand side (src/d2d/dynamic.py:14-23); kept for callers that inspect it --
        the integration itself never calls back into Python."""
        wx, wy = _wind_of(W, t, X[:2])
        (x, y, psi, phi, v), (phi_c, v_c) = X, U
        return [v * np.cos(psi) + wx, v * np.sin(psi) + wy, self.g / v * np.tan(phi),
                -1 / self.tau_phi * (phi - phi_c), -1 / self.tau_v * (v - v_c)]

    def disc_dyn(self, Xk, Uk, W, t, dt):
        """One zero-order-hold step (src/d2d/dynamic.py:25-28) on the GPU."""
        ctx = d2dhip.default_context()
        wx, wy = _wind_of(W, t, Xk[:2])
        X = ctx.dev(np.asarray(Xk, dtype=np.float64).reshape(5, 1))
        U = ctx.dev(np.asarray(Uk, dtype=np.float64).reshape(2, 1))
        return ctx.step(X, U, (float(wx), float(wy)), self.tau_phi, self.tau_v, float(dt)).cpu().numpy()[:, 0]

    def cont_jac(self, Xr, Ur, t, W):
        """Linearisation A (5x5), B (5x2) with the reference's entries (src/d2d/dynamic.py:32-43)."""
        ctx = d2dhip.default_context()
        A, B = ctx.cont_jac(ctx.dev(np.asarray(Xr, dtype=np.float64).reshape(5, 1)), self.tau_phi, self.tau_v)
        return A.cpu().numpy()[:, 0].reshape(5, 5), B.cpu().numpy()[:, 0].reshape(5, 2)
