"""Mirror of src/d2d/trajectory.py: the 1-D primitives (CstOne, AffineOne, SinOne, PolynomialOne), the 2-D trajectories
(Trajectory, TrajectoryLine, TrajectoryCircle, MinSnapPoly, CompositeTraj, SpaceIndexedTraj) with the reference's
`get(t) -> [derivative, axis]` protocol, and `describe()`: the descriptor rows from which d2d_traj_sample evaluates lines,
circle arcs, slaloms, polynomial pairs and composites of those ON THE DEVICE for whole batches of aircraft (the reference
samples `traj.get(t)` in a Python loop per aircraft and time step, src/05_test_simulation.py:25).  The polynomial classes are
also the basis the fit is expressed in (d2d_fit_sample / d2d_fit_coeffs evaluate whole fits on the GPU)."""
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


class CstOne:
    def __init__(self, c=-1.):
        self.c = c

    def get(self, t):
        return np.array([self.c, 0, 0, 0])


class AffineOne:
    def __init__(self, c1=-1., c2=0, duration=1.):
        self.c1, self.c2, self.duration = c1, c2, duration

    def get(self, t):
        return np.array([self.c1 * t + self.c2, self.c1, 0, 0])


class SinOne:
    def __init__(self, c=0., a=1., om=1., duration=2 * np.pi):
        self.duration, self.c, self.a, self.om, self.t0 = duration, c, a, om, 0.

    def get(self, t):
        al = self.om * (t - self.t0)
        sa, ca = self.a * np.sin(al), self.a * np.cos(al)
        return np.array([self.c + sa, self.om * ca, -self.om ** 2 * sa, -self.om ** 3 * ca])


class Trajectory:
    """Base class (src/d2d/trajectory.py:88-122): get(t) -> (nder+1, ncomp) array [derivative, axis]."""
    desc = ''
    cx, cy, ncomp = 0, 1, 2
    nder = 3
    extends = (0, 100, 0, 100)

    def __init__(self):
        self.t0 = 0.

    def get(self, t):
        return np.zeros((self.nder + 1, self.ncomp))

    def reset(self, t0):
        self.t0 = t0

    def compute_extends(self, dt=0.1):
        Ys = np.array([self.get(t) for t in np.arange(self.t0, self.t0 + self.duration, dt)])
        p0, p1 = np.min(Ys[:, 0], axis=0).round(1) - 1, np.max(Ys[:, 0], axis=0).round(1) + 1
        self.extends = (p0[0], p1[0], p0[1], p1[1])

    def summarize(self):
        return f'{self.desc}\nduration: {self.duration:.2f}s\nextends: {self.extends}'

    def describe(self):
        """Descriptor segments for d2d_traj_sample: list of (type, params) or None if this trajectory has no device form."""
        return None


class TrajectoryLine(Trajectory):
    """Straight line at constant ground speed (src/d2d/trajectory.py:125-141)."""

    def __init__(self, p1, p2, v=10., t0=0.):
        self.p1, self.p2, self.v, self.t0 = np.asarray(p1, dtype=float), np.asarray(p2, dtype=float), v, t0
        dep = self.p2 - self.p1
        self.length = np.linalg.norm(dep)
        self.un = dep / self.length
        self.duration = self.length / self.v

    def get(self, t):
        Yc = np.zeros((Trajectory.nder + 1, Trajectory.ncomp))
        Yc[0] = self.p1 + self.un * self.v * (t - self.t0)
        Yc[1] = self.un * self.v
        return Yc

    def describe(self):
        return [(1, [self.p1[0], self.p1[1], self.un[0], self.un[1], self.v], self)]


class TrajectoryCircle(Trajectory):
    """Circle arc, sign of r = direction (src/d2d/trajectory.py:143-160)."""

    def __init__(self, c=[30., 30.], r=30., v=10., t0=0., alpha0=0, dalpha=2 * np.pi):
        self.c, self.r, self.v, self.t0 = np.asarray(c, dtype=float), r, v, t0
        self.alpha0, self.dalpha = alpha0, dalpha
        self.omega = self.v / self.r
        self.duration = np.abs(r) * dalpha / v

    def get(self, t):
        al = (t - self.t0) * self.omega + self.alpha0
        ca, sa = np.cos(al), np.sin(al)
        r, om = self.r, self.omega
        return np.array((self.c + r * np.array([ca, sa]), om * r * np.array([-sa, ca]), om ** 2 * r * np.array([-ca, -sa]),
                         om ** 3 * r * np.array([sa, -ca])))

    def describe(self):
        return [(2, [self.c[0], self.c[1], self.r, self.omega, self.alpha0], self)]


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

    def describe(self):
        if any(p._der != 4 for p in self._polys):
            return None
        return [(4, list(self._polys[0].coefs[0]) + list(self._polys[1].coefs[0]), self)]


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

    def describe(self):
        segs = []
        for st in self.steps:
            d = st.describe()
            if d is None or len(d) != 1:           # (composites of composites have no device form)
                return None
            segs += d
        return segs

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


class TabulatedTraj(Trajectory):
    """(a stub in the reference too, src/d2d/trajectory.py:212-217)"""

    def __init__(self, filename='/tmp/optyplan.npz'):
        pass


class SpaceIndexedTraj(Trajectory):
    """Geometry g(lambda) driven by a scalar dynamic lambda(t): chain rule up to the third derivative
    (src/d2d/trajectory.py:220-241).  Host-side only (no device descriptor)."""

    def __init__(self, geometry, dynamic):
        self.duration = dynamic.duration
        self.extends = geometry.extends
        self._geom, self._dyn = geometry, dynamic

    def set_dyn(self, dyn):
        self._dyn = dyn
        self.duration = dyn.duration

    def get(self, t):
        Yt = np.zeros((self._geom.nder + 1, self._geom.ncomp))
        lam = np.array(self._dyn.get(t), dtype=float)
        lam[0] = np.clip(lam[0], 0., 1.)
        g = self._geom.get(lam[0])
        Yt[0] = g[0]
        Yt[1] = lam[1] * g[1]
        Yt[2] = lam[2] * g[1] + lam[1] ** 2 * g[2]
        Yt[3] = lam[3] * g[1] + 3 * lam[1] * lam[2] * g[2] + lam[1] ** 3 * g[3]
        return Yt


def describe(traj):
    """One d2d_traj_sample descriptor row (numpy, d2dhip.TRAJ_STRIDE) of a trajectory, or None when it has no device form
    (space-indexed, spline, tabulated trajectories: sample those with traj.get on the host)."""
    import d2dhip
    segs = traj.describe() if hasattr(traj, 'describe') else None
    if segs is None or not 1 <= len(segs) <= d2dhip.TRAJ_MAX_SEG:
        return None
    row = np.zeros(d2dhip.TRAJ_STRIDE)
    comp = isinstance(traj, CompositeTraj)
    row[0], row[1], row[2], row[3] = len(segs), getattr(traj, 't0', 0.) if comp else 0., traj.duration, 1.0 if comp else 0.0
    end = 0.0
    for k, (typ, par, obj) in enumerate(segs):
        end += obj.duration
        o = 4 + k * d2dhip.TRAJ_SEG_STRIDE
        row[o], row[o + 1], row[o + 2] = typ, getattr(obj, 't0', 0.), end
        row[o + 3:o + 3 + len(par)] = par
    return row
