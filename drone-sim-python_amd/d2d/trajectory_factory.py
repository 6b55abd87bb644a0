"""Mirror of src/d2d/trajectory_factory.py: the named registry of demo trajectories that the legacy simulations
(src/02_test_traj.py, src/05_test_simulation.py through d2d.scenario) are driven with -- register / list_available / get(name) --
built on the primitives of d2d.trajectory.  Lines, circles, slaloms, min-snap polynomials and composites of them carry a
device descriptor (Trajectory.describe -> d2d_traj_sample); the spline / space-indexed / tabulated ones are sampled on the host."""
import numpy as np

import d2d.trajectory as ddt

trajectories = {}


def register(T):
    trajectories[T.name] = (T.desc, T)


def list_available():
    return ['{}: {}'.format(k, v[0]) for k, v in sorted(trajectories.items())]


class TrajCircle(ddt.TrajectoryCircle):
    name, desc = 'circle', '30m radius (30,30) centered circle'
    extends = (-10, 70, -10, 70)

    def __init__(self):
        ddt.TrajectoryCircle.__init__(self, c=[30., 30.], r=30., v=10., t0=0., alpha0=0, dalpha=2 * np.pi)


class _Polyline(ddt.CompositeTraj):
    """Constant-speed legs through `points` (how the reference builds its two-lines and square examples, :28-50)."""

    def __init__(self, points, v=10.):
        steps, t = [], 0.
        for a, b in zip(points[:-1], points[1:]):
            steps.append(ddt.TrajectoryLine(a, b, v=v, t0=t))
            t += steps[-1].duration
        ddt.CompositeTraj.__init__(self, steps)


class TrajTwoLines(_Polyline):
    name, desc = 'two_lines', 'example of composite trajectory'
    extends = (-20, 120, -20, 60)

    def __init__(self):
        _Polyline.__init__(self, [[0, 0], [50, 50], [100, 0]])


class TrajSquare(_Polyline):
    name, desc = 'square', 'example of composite trajectory'
    extends = (-10, 60, -10, 60)

    def __init__(self):
        _Polyline.__init__(self, [[0, 0], [50, 0], [50, 50], [0, 50], [0, 0]])


class TrajLineWithIntro(ddt.CompositeTraj):
    name, desc = 'line_with_intro', 'line with circle_intro'

    def __init__(self, Y0=[0, 0], Y1=[0, 50], Y2=[100, 50], r=-25.):
        half_turn = ddt.TrajectoryCircle(c=(np.asarray(Y0) + Y1) / 2, r=r, v=10., alpha0=np.pi / 2, dalpha=np.pi)
        ddt.CompositeTraj.__init__(self, [half_turn, ddt.TrajectoryLine(Y1, Y2, v=10., t0=half_turn.duration)])
        self.extends = (-50, 130, -30, 130)


class TrajWithIntro(ddt.CompositeTraj):
    """A straight run-in of `duration` seconds from Y0 to the start of `traj` (:65-87; not registered: it needs a scenario)."""
    name, desc = 'traj_with_intro', 'traj with circle_intro'

    def __init__(self, Y0, traj, v=10, duration=8.):
        Y0 = np.asarray(Y0, dtype=float)
        Y1 = traj.get(0)[0]
        intro = ddt.TrajectoryLine(Y0, Y1, v=np.linalg.norm(Y1 - Y0) / duration)
        ddt.CompositeTraj.__init__(self, [intro, traj])


class TrajMinSnapDemo(ddt.MinSnapPoly):
    name, desc = 'demo_minsnap', 'demo_minsnap'
    extends = (-10, 210, -10, 210)

    def __init__(self):
        ddt.MinSnapPoly.__init__(self, [[0, 10, 0, 0], [0, 0, 0, 0]], [[200, 0, 0, 0], [200, 10, 0, 0]], duration=33.65)


class TrajSlalom(ddt.Trajectory):
    """Line at speed v with a 10 m, 1 rad/s sine on y (:113-137)."""
    name, desc = 'slalom', 'slalom'
    extends = (-10, 100, -10, 50)
    A, OM = 10., 1.

    def __init__(self, p1=[0, 20], p2=[100, 20], v=10., t0=0., phi=0.):
        self.p1, self.p2, self.v, self.t0, self.phi = np.asarray(p1, dtype=float), np.asarray(p2, dtype=float), v, t0, phi
        dep = self.p2 - self.p1
        self.length = np.linalg.norm(dep)
        self.un = dep / self.length
        self.duration = self.length / self.v

    def get(self, t):
        Yc = np.zeros((self.nder + 1, self.ncomp))
        Yc[0] = self.p1 + self.un * self.v * (t - self.t0)
        Yc[1] = self.un * self.v
        a, om = self.A, self.OM
        al = om * (t - self.t0 + self.phi)
        s, c = np.sin(al), np.cos(al)
        Yc[0, 1] += a * s; Yc[1, 1] += a * om * c; Yc[2, 1] += -a * om ** 2 * s; Yc[3, 1] += -a * om ** 3 * c
        return Yc

    def describe(self):
        return [(3, [self.p1[0], self.p1[1], self.un[0], self.un[1], self.v, self.A, self.OM, self.phi], self)]


class TrajTabulated(ddt.Trajectory):
    """A saved plan (the planners' npz cache: sol_time, sol_x, ..., wind) replayed as a reference: nearest-sample lookup (:140-163)."""
    name, desc = 'tabulated', 'tabulated'
    extends = (-5, 25, -10, 20)

    def __init__(self, filename='./optyplan_exp0.npz'):
        d = np.load(filename)
        (self.sol_time, self.sol_x, self.sol_y, self.sol_psi, self.sol_phi, self.sol_v, self.wind) = (
            d[k] for k in ('sol_time', 'sol_x', 'sol_y', 'sol_psi', 'sol_phi', 'sol_v', 'wind'))
        print(f'loaded {filename}')
        self.t0, self.duration = 0., self.sol_time[-1]
        self.compute_extends()

    def get(self, t):
        Yc = np.zeros((self.nder + 1, self.ncomp))
        i = np.argmin(t > self.sol_time)
        (wx, wy), v, psi = self.wind[i], self.sol_v[i], self.sol_psi[i]
        Yc[0] = self.sol_x[i], self.sol_y[i]
        Yc[1] = v * np.cos(psi) + wx, v * np.sin(psi) + wy
        return Yc


class TrajSiDemo(ddt.SpaceIndexedTraj):
    name, desc = 'sidemo', 'space indexed trajectory demo'
    extends = (-10, 100, -10, 50)

    def __init__(self, p1=[0, 20], p2=[100, 20], duration=10., t0=0.):
        ddt.SpaceIndexedTraj.__init__(self, ddt.TrajectoryLine([0, 20], [100, 20], v=100),
                                      ddt.PolynomialOne([0, 0.05, 0, 0], [1, 0.05, 0, 0], duration=duration))


class TrajSpline(ddt.Trajectory):
    name, desc = 'spline', 'spline dev'

    def __init__(self, waypoints=None, duration=None):
        import scipy.interpolate as interpolate
        self.waypoints = waypoints or np.array([[0., 0.], [50, 50], [100, 0], [150, 50], [200, 0]])
        if duration is None:
            duration = np.sum(np.linalg.norm(self.waypoints[1:] - self.waypoints[:-1], axis=1)) / 10.
        self.duration = duration
        knots = np.linspace(0, self.duration, len(self.waypoints))
        self.splines = [interpolate.InterpolatedUnivariateSpline(knots, self.waypoints[:, i], k=4) for i in range(2)]
        self.extends = [0, 200, -20, 80]

    def get(self, t):
        t = np.fmod(t, self.duration)
        return np.array([self.splines[i].derivatives(t) for i in range(self.ncomp)])[:, :self.nder + 1].T


class SplineOne:
    def __init__(self, xs, ys):
        import scipy.interpolate as interpolate
        self.nder = 3
        self.dyn = interpolate.InterpolatedUnivariateSpline(xs, ys, k=4)
        self.duration = xs[-1]

    def get(self, t):
        return np.array(self.dyn.derivatives(t))[:self.nder + 1].T


class FooOne:
    """Piecewise-linear scalar dynamic (:221-231)."""

    def __init__(self, xs, ys):
        self.xs, self.ys = np.asarray(xs), np.asarray(ys)
        self.ds = (self.ys[1:] - self.ys[:-1]) / (self.xs[1:] - self.xs[:-1])
        self.duration = xs[-1]

    def get(self, t):
        i = np.where(t >= self.xs)[0][-1]
        i = min(i, len(self.ds) - 1)
        return np.array([self.ys[i] + (t - self.xs[i]) * self.ds[i], self.ds[i], 0, 0])


class TrajSiSpline(ddt.SpaceIndexedTraj):
    """Three quarters of a circle flown at constant AIR speed in a 5 m/s wind: the time law lambda(t) is fitted with
    scipy.optimize.minimize so that |ground velocity - wind| stays at 10 m/s (:241-284; the reference's one use of
    scipy.optimize).  Host-side."""
    name, desc = 'sispline', 'spline dev'
    extends = (-10, 100, -10, 50)

    def __init__(self, duration=30.):
        import scipy.optimize
        geometry = ddt.TrajectoryCircle(c=[30., 30.], r=30., v=2 * np.pi * 30., t0=0., alpha0=0, dalpha=3 * np.pi / 2)
        ddt.SpaceIndexedTraj.__init__(self, geometry, ddt.AffineOne(1. / duration, 0., duration=duration))
        self._dyn1 = self._dyn
        wind = np.array([5., 0.])
        self.ts = np.arange(0, self._dyn.duration, 0.5)
        npts, vtarget = 10, 10.
        xs = np.linspace(0, 30, npts)

        def err_fun(p):
            self.set_dyn(FooOne(xs, np.concatenate(([0.], np.cumsum(p)))))
            vel_gnd = np.array([self.get(t)[1] for t in self.ts])
            return np.mean(np.square(np.linalg.norm(vel_gnd - wind, axis=1) - vtarget))
        res = scipy.optimize.minimize(err_fun, [1. / npts] * (npts - 1))
        self._dyn4 = SplineOne(xs, np.concatenate(([0.], np.cumsum(res.x))))


for _T in (TrajCircle, TrajTwoLines, TrajSquare, TrajLineWithIntro, TrajMinSnapDemo, TrajSlalom, TrajTabulated, TrajSiDemo, TrajSpline,
           TrajSiSpline):
    register(_T)


def print_available():
    print('Available trajectories:')
    for i, n in enumerate(list_available()):
        print(f'{i} -> {n}')


def get(traj_name):
    return trajectories[traj_name][1](), trajectories[traj_name][0]
