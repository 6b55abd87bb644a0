"""Mirror of the polynomial trajectory primitives of src/d2d/trajectory.py -- the basis the
fit is expressed in: PolynomialOne, MinSnapPoly, CompositeTraj (host-side, closed form; the
batched evaluation of whole fits is d2d_fit_sample / d2d_fit_coeffs on the GPU)."""
import math

import numpy as np


def arr(k, n):
    """n! / (n-k)!  (src/d2d/trajectory.py:41-45)."""
    a = 1
    for i in range(n, n - k, -1):
        a *= i
    return a


class PolynomialOne:
    """Degree-7 two-point boundary-value polynomial: 4 derivatives given at 0 and at `duration`
    (src/d2d/trajectory.py:47-82).  coefs[d, p] is the coefficient of t^p of derivative d."""

    def __init__(self, Y0, Y1, duration):
        self.duration = duration
        nd = len(Y0)
        self._der, self._order = nd, 2 * nd
        self.coefs = np.zeros((nd, 2 * nd))
        low = np.array([Y0[i] / arr(i, i) for i in range(nd)])
        M3 = np.array([[arr(i, j) * duration ** (j - i) if j >= i else 0. for j in range(nd)] for i in range(nd)])
        M4 = np.array([[arr(i, j + nd) * duration ** (j - i + nd) for j in range(nd)] for i in range(nd)])
        self.coefs[0, :nd] = low
        self.coefs[0, nd:] = np.linalg.solve(M4, np.asarray(Y1, dtype=float) - M3 @ low)
        for d in range(1, nd):
            for p in range(2 * nd - d):
                self.coefs[d, p] = arr(d, p + d) * self.coefs[0, p + d]

    @classmethod
    def from_coefs(cls, coefs0, duration):
        """Build from the 8 monomial coefficients (the layout d2d_fit_coeffs returns)."""
        self = cls.__new__(cls)
        nd = len(coefs0) // 2
        self.duration, self._der, self._order = duration, nd, 2 * nd
        self.coefs = np.zeros((nd, 2 * nd))
        self.coefs[0] = coefs0
        for d in range(1, nd):
            for p in range(2 * nd - d):
                self.coefs[d, p] = arr(d, p + d) * self.coefs[0, p + d]
        return self

    def get(self, t):
        Y = np.zeros(self._der)
        for d in range(self._der):
            v = self.coefs[d, -1]
            for j in range(self._order - 2, -1, -1):
                v = v * t + self.coefs[d, j]
            Y[d] = v
        return Y


class Trajectory:
    cx, cy, ncomp = 0, 1, 2
    nder = 3


class MinSnapPoly(Trajectory):
    """One PolynomialOne per axis (src/d2d/trajectory.py:166-187)."""

    def __init__(self, Y00=[0, 0], Y10=[1, 0], duration=1.):
        self.duration = duration

        def full(Y):
            Y = np.asarray(Y, dtype=float)
            if Y.ndim == 1:
                F = np.zeros((Trajectory.ncomp, Trajectory.nder + 1)); F[:, 0] = Y
                return F
            return Y
        Y0, Y1 = full(Y00), full(Y10)
        self._polys = [PolynomialOne(Y0[i], Y1[i], duration) for i in range(Trajectory.ncomp)]
        self.t0 = 0

    def reset(self, t0):
        self.t0 = t0

    def get(self, t):
        return np.array([p.get(t - self.t0) for p in self._polys]).T


class CompositeTraj(Trajectory):
    """Sequence of segments, periodic in its total duration (src/d2d/trajectory.py:190-208)."""

    def __init__(self, steps):
        self.steps = steps
        self.steps_dur = [s.duration for s in steps]
        self.steps_end = np.cumsum(self.steps_dur)
        self.duration = np.sum(self.steps_dur)
        for s, st in zip(steps[1:], self.steps_end):
            s.reset(st)
        self.t0 = 0.

    def reset(self, t0):
        self.t0 = t0

    def get(self, t):
        lapse = math.fmod(t - self.t0, self.duration)
        return self.steps[int(np.argmax(self.steps_end > lapse))].get(lapse)

    @classmethod
    def from_fit(cls, z, duration):
        """z (2, S, 8) monomial coefficients of one fitted trajectory -> CompositeTraj."""
        S = z.shape[1]
        steps = []
        for s in range(S):
            seg = MinSnapPoly.__new__(MinSnapPoly)
            seg.duration, seg.t0 = duration / S, 0
            seg._polys = [PolynomialOne.from_coefs(z[a, s], duration / S) for a in range(2)]
            steps.append(seg)
        return cls(steps)
