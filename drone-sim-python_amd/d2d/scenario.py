"""Mirror of src/d2d/scenario.py: the simulation scenarios (which trajectories, wind, start states, perturbations) that
src/03_test_scenario.py and src/05_test_simulation.py run -- registry (register / list_available / get), the Scenario base class
with its defaults, and the reference's registered entries with the reference's parameters.  A scenario's aircraft are simulated
together on the device by full_sim.test_simulation (references sampled by d2d_traj_sample where the trajectory has a descriptor,
DFFFController loop by d2d_sim_dfff_run)."""
import numpy as np

import d2d.trajectory as ddt
import d2d.trajectory_factory as ddtf
import d2d.guidance as d2guid
import d2d.dynamic as d2dyn
from d2d.dynamic import Aircraft

_scenarios = {}
_default_dt = 0.01


def register(S):
    _scenarios[S.name] = (S.desc, S)


def list_available():
    return ['{}: {}'.format(k, v[0]) for k, v in sorted(_scenarios.items())]


class Scenario:
    """Children provide `trajs` (and whatever else they want to fix); everything missing gets the reference's default
    (src/d2d/scenario.py:24-56): time = 0 .. longest trajectory at 10 ms, one Aircraft per trajectory, zero perturbations, still
    air, start states on the references (flatness map), plot extends from the sampled references, DFFF control."""

    def __init__(self):
        nv = len(self.trajs)
        if not hasattr(self, 'time'):
            self.time = np.arange(0., np.max([traj.duration for traj in self.trajs]), _default_dt)
        if not hasattr(self, 'aircrafts'):
            self.aircrafts = [Aircraft() for _ in range(nv)]
        if not hasattr(self, 'perts'):
            self.perts = [np.zeros((len(self.time), Aircraft.s_size)) for _ in range(nv)]
        if not hasattr(self, 'windfield'):
            self.windfield = d2guid.WindField()
        if not hasattr(self, 'X0s'):
            self.X0s = []
            for ac, traj in zip(self.aircrafts, self.trajs):
                t0 = self.time[0]
                Yr = traj.get(t0)
                self.X0s.append(d2guid.DiffFlatness.state_and_input_from_output(Yr, self.windfield.sample(t0, Yr[0]), ac)[0])
        if not hasattr(self, 'extends'):
            self.extends = (0., 100., 0., 100.)
            self.autoscale()
        if not hasattr(self, 'ppctl'):
            self.ppctl = False

    def autoscale(self):
        P = np.array([traj.get(t)[0] for traj in self.trajs for t in self.time[::10]])
        pmin, pmax = P.min(0), P.max(0)
        margin = 0.05 * (pmax - pmin)
        pmin, pmax = pmin - margin, pmax + margin
        self.extends = (pmin[0], pmax[0], pmin[1], pmax[1])

    def summarize(self):
        ext = ''.join(f'{e:.1f} ' for e in self.extends)
        return (f'{len(self.trajs)} trajectories\nduration: {self.time[-1] - self.time[0]:.2f}s\n'
                f'wind: {self.windfield.summarize()}\nextends: {ext}')


class ScenLine(Scenario):
    name = desc = 'line'

    def __init__(self):
        self.trajs = [ddt.TrajectoryLine([0, 25], [100, 25], v=10., t0=0.)]
        self.extends = (-10, 110, 0, 50)
        self.windfield = d2guid.WindField()
        self.time = np.arange(0, 12., 0.01)
        self.X0s = [[10, 10, 0, 0, 10]]
        self.perts = [np.zeros((len(self.time), d2dyn.Aircraft.s_size))]
        self.perts[0][600, d2dyn.Aircraft.s_y] = 10          # a 10 m kick sideways at t = 6 s
        Scenario.__init__(self)


class ScenLine2(Scenario):
    name = desc = 'line2'

    def __init__(self):
        self.trajs = [ddtf.TrajTwoLines()]
        self.extends = self.trajs[0].extends
        self.windfield = d2guid.WindField()
        self.time = np.arange(0, 12., 0.01)
        self.X0s = [[0, 10, 0, 0, 10]]
        Scenario.__init__(self)


class ScenCircle(Scenario):
    name = desc = 'circle'

    def __init__(self, duration=None, cst_gvel=False):
        self.trajs = [ddt.TrajectoryCircle(alpha0=3 * np.pi / 2)] if cst_gvel else [ddtf.TrajSiSpline(duration=20.)]
        self.extends = (-10, 75, -10, 75)
        self.windfield = d2guid.WindField([5, 0])
        self.time = np.arange(0, self.trajs[0].duration, 0.01)
        Scenario.__init__(self)


class ScenSquare(Scenario):
    name = desc = 'square'

    def __init__(self):
        self.trajs = [ddtf.TrajSquare()]
        self.extends = self.trajs[0].extends
        self.windfield = d2guid.WindField()
        self.time = np.arange(0, 30., 0.01)
        self.X0s = [[0, 0, 0, 0, 10]]
        Scenario.__init__(self)


class ScenMultiCircle(Scenario):
    name, desc = 'mucir', '5 circles (30m radius, x offset)'

    def __init__(self, dx=0., dalpha=np.deg2rad(30.), nc=5, v=10.):
        self.trajs = [ddt.TrajectoryCircle(c=[40. + i * dx, 50.], alpha0=i * dalpha, v=v) for i in range(nc)]
        self.extends = (0, 80, 10, 90)
        self.X0s = [[75 - 5 * i, 60 + 5 * i, np.pi, 0, 10] for i in range(nc)]
        self.windfield = d2guid.WindField([5, 0])
        self.time = np.arange(0, 20, 0.01)
        Scenario.__init__(self)


class ScenMultiCircle2(Scenario):
    name, desc = 'mucir2', '2 circles (30m radius, x offset)'

    def __init__(self, dx=0):
        self.trajs = [ddt.TrajectoryCircle(c=[40., 50.], alpha0=0.), ddt.TrajectoryCircle(c=[40. + dx, 50.], alpha0=np.deg2rad(30.))]
        self.extends = (0, 100, 0, 100)
        self.X0s = [[75, 50, np.pi / 2, 0, 10], [85, 70, np.pi / 1.5, 0, 10]]
        self.windfield = d2guid.WindField([1., 0])
        self.time = np.arange(0, 18, 0.01)
        Scenario.__init__(self)


class ScenPatrol(Scenario):
    name, desc = 'patrol', "The original 'Line Patrol' scenario"

    def __init__(self):
        self.trajs = [ddtf.TrajLineWithIntro(Y0=[0., 100.], Y1=[0., 50.], Y2=[200., 50.], r=25.),
                      ddtf.TrajLineWithIntro(Y0=[0., 0.], Y1=[0., 50.], Y2=[200., 50.], r=-25.)]
        self.X0s = [[0, 100, -np.pi, 0, 10], [0, 0, -np.pi, 0, 10]]
        self.extends = (-30, 110, -10, 110)
        self.windfield = d2guid.WindField([0, 2.5])
        self.time = np.arange(0, 20, 0.01)
        Scenario.__init__(self)


class ScenPatrol2(Scenario):
    name, desc = 'patrol_2', 'dev patrol'

    def __init__(self):
        self.trajs = [ddtf.TrajLineWithIntro(Y0=[0., 100.], Y1=[0., 50.], Y2=[100., 50.], r=25.),
                      ddtf.TrajWithIntro([-20, 0], ddtf.TrajSlalom(p1=[0, 50], p2=[100, 50], v=10.), duration=8.),
                      ddtf.TrajWithIntro([0, 0], ddtf.TrajSlalom(p1=[0, 40], p2=[100, 40], v=10.), duration=8.)]
        self.X0s = [[0, 100, -np.pi, 0, 10], [-15, 0, np.pi / 2, 0, 10], [5, 0, np.pi / 2, 0, 10]]
        self.extends = (-30, 110, -10, 110)
        self.windfield = d2guid.WindField([0, 2.5])
        self.time = np.arange(0., 17.5, _default_dt)
        Scenario.__init__(self)


class ScenPatrol3(Scenario):
    name, desc = 'patrol_3', 'dev patrol'

    def __init__(self, nv=2):
        self.trajs = []
        for i in range(nv):
            dy = 5 * i
            dx = dy / 2
            out = ddt.TrajectoryLine([0, 10 + dy], [100 - dx, 10 + dy], v=10., t0=0.)
            turn = ddt.TrajectoryCircle(c=[100 - dx, 40], r=30. - dy, v=10., t0=0., alpha0=-np.pi / 2, dalpha=np.pi)
            back = ddtf.TrajSlalom(p1=[100, 60 - dy], p2=[0, 60 - dy], v=10., t0=0., phi=np.pi / 2)
            self.trajs.append(ddt.CompositeTraj([out, turn, back]))
        self.windfield = d2guid.WindField([0, 5.])
        Scenario.__init__(self)


class ScenCircularFormation(Scenario):
    name, desc = 'circForm', 'circular formation'

    def __init__(self):
        P0s = [[30, 10], [40, 10]]
        self.trajs = [ddt.TrajectoryCircle(alpha0=3 * np.pi / 2 + i * np.pi / 6) for i in range(len(P0s))]
        Scenario.__init__(self)
        for P0, X0 in zip(P0s, self.X0s):
            X0[:Aircraft.s_y + 1] = P0
        self.ppctl = False
        self.windfield = d2guid.WindField([0, 5.])


class ScenOval(Scenario):
    """(as in the reference, :275-286, this entry fills its attributes itself and does not run the base-class defaults)"""
    name = desc = 'oval'

    def __init__(self):
        self.trajs = [ddtf.TrajLineWithIntro(Y0=[0., 100.], Y1=[0., 50.], Y2=[200., 50.], r=25.)]
        self.X0s = [[0, 100, -np.pi, 0, 10], [0, 0, -np.pi, 0, 10]]
        self.extends = (-30, 110, -10, 110)
        self.windfield = d2guid.WindField([0, 2.5])
        self.time = np.arange(0, 20, 0.01)


class ScenDualOpty(Scenario):
    """Replays saved plans (./optyplan_exp6_<i>.npz, written by the planner's save_solution) as references."""
    name = desc = 'dual opty'

    def __init__(self):
        self.trajs = [ddtf.TrajTabulated(f'optyplan_exp6_{i}.npz') for i in range(3)]
        Scenario.__init__(self)


class ScenOpty2(Scenario):
    name = desc = 'opty2'

    def __init__(self):
        self.trajs = [ddtf.TrajTabulated(f'optyplan_exp7_{i}.npz') for i in range(5)]
        Scenario.__init__(self)


for _S in (ScenLine, ScenLine2, ScenCircle, ScenSquare, ScenMultiCircle, ScenMultiCircle2, ScenPatrol, ScenPatrol2, ScenPatrol3,
           ScenCircularFormation, ScenOval, ScenDualOpty, ScenOpty2):
    register(_S)


def print_available():
    print('Available scenarios:')
    for i, n in enumerate(list_available()):
        print(f'{i} -> {n}')


def get(_name):
    return _scenarios[_name][1](), _scenarios[_name][0]
