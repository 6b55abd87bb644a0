"""Scenario catalogue of the multi-aircraft planner in the reference's protocol -- the fifteen entries the reference keeps
inline in src/07_multioptyplan.py:170-435 (its newer src/multi_opt_planner.py:170-242 carries exp_0, exp_5 and trap_4 only;
those live in this package's multi_opt_planner.py).  Class attributes: name, desc, t0, t1, hz, p0s / p1s (one (x, y, psi, phi, v)
per aircraft), wind, initial_guess, tol, max_iter, vref, cost, obj_scale, x/y/phi/v_constraint, obstacles, ncases, set_case(idx),
label(idx); registry `scens`, desc_all_scens(), get_scen(idx), info_scen(idx).  Scenario constants are the reference's; set_case
mutating class attributes (exp_4_2 its own t1, inf_traj_4ac the t1 of exp_5) is part of the protocol and is kept."""
import numpy as np

import d2d.opty_utils as d2ou
import d2d.multiopty_utils as d2mou

_NAN = float('NaN')


class exp_0:
    name, desc = 'exp_0', 'single aircraft'
    t0, t1, hz = 0., 10., 50.
    dx, dy = 0., 50.
    p0s = ((0., 0., 0., 0., 10.), )
    p1s = ((dx, dy, np.pi / 2, 0., 10.), )
    wind = d2ou.WindField()
    initial_guess = 'tri'
    tol, max_iter = 1e-5, 5000
    vref = 12.
    cost, obj_scale = d2mou.CostInput(vsp=vref, kv=5., kphi=1.), 1.e-1
    x_constraint, y_constraint = (-5, 50), (-5, 50)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    v_constraint = (9., 15.)
    obstacles = []
    ncases = 1

    def set_case(idx): pass
    def label(idx): return ''


class exp_0_1(exp_0):
    name, desc = 'exp_0_1', 'single aircraft, varying weights'
    t0, p0s = 0., ((0., 0., 0., 0., 10.), )
    t1, p1s = 10., ((100, 0, 0., 0., 10.), )
    x_constraint, y_constraint = None, None
    Ks = [[1., 1.], [1., 20.], [1., 40.], [1., 60.]]
    ncases = len(Ks)

    def set_case(idx):
        exp_0_1.K = exp_0_1.Ks[idx]
        exp_0_1.cost = d2mou.CostInput(vsp=13., kv=exp_0_1.K[0], kphi=exp_0_1.K[1])

    def label(idx): return f'kvel, kbank {exp_0_1.K}'


class exp_1(exp_0):
    name, desc = 'exp_1', '2 aicraft face to face'
    t1 = 4.5
    vref = 12.
    dpsi = 0.01
    p0s = ((0., 0., 0., 0., 12.), (50., 0., np.pi - dpsi, 0., 12.))
    p1s = ((50., 0., 0., 0., 12.), (0., 0., np.pi + dpsi, 0., 12.))
    cost, obj_scale = d2mou.CostInput(vsp=vref, kv=5., kphi=1.), 1.e-1
    x_constraint, y_constraint = None, None
    obstacles = []
    initial_guess = 'rnd'


class exp_1_0(exp_1):
    name, desc = 'exp_1_0', '2 aicraft meeting'
    t1 = 4.5
    vref = 12.
    p0s = ((0., -20., np.pi / 2, 0., 12.), (7.5, -20., np.pi / 2, 0., 12.))
    p1s = ((40., 5., 0., 0., 12.), (40., 10., 0, 0., 12.))
    cost, obj_scale = d2mou.CostInput(vsp=vref, kv=1., kphi=1.), 1.e-1
    x_constraint, y_constraint = None, None


class exp_1_1(exp_1):
    name, desc = 'exp_1_1', '2 aicraft face to face, wind'
    initial_guess = 'tri'


class exp_2(exp_0):
    name, desc = 'exp_2', '4 aicraft'
    t1 = 5.5
    vref = 12.
    overtime = 1.5
    d = t1 * vref / 2 / overtime
    p0s = ((-d, 0., 0., 0., vref), (d, 0., np.pi, 0., vref), (0., d, -np.pi / 2, 0., vref), (0., -d, np.pi / 2, 0., vref))
    p1s = ((d, 0., 0., 0., vref), (-d, 0., np.pi, 0., vref), (0., -d, -np.pi / 2, 0., vref), (0., d, np.pi / 2, 0., vref))
    cost, obj_scale = d2mou.CostInput(vsp=vref, kv=1., kphi=1.), 1.


class exp_3(exp_0):
    name, desc = 'exp_3', 'single obstacle'
    t1 = 6.5
    vref = 12.
    p0s = ((0., 0., 0., 0., 10.), )
    p1s = ((50., 0., 0., 0., 10.), )
    obstacles = ((25, -20, 10), )
    cx, cy, r = obstacles[0]
    cost, obj_scale = d2mou.CostObstacle(c=(cx, cy), r=r, kind=0), 1.
    x_constraint, y_constraint = None, None
    v_constraint = (8., 18.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))


class exp_3_1(exp_3):
    name, desc = 'exp_3_1', 'single obstacle, size/location'
    obstacles = ((25, -20, 10), (25, -10, 10))
    ncases = len(obstacles)

    def set_case(idx):
        cx, cy, r = exp_3_1.obstacles[idx]
        exp_3_1.cost = d2mou.CostObstacle(c=(cx, cy), r=r, kind=0)

    def label(idx): return f'obstacle {exp_3_1.obstacles[idx]}'


class exp_4(exp_0):
    name, desc = 'exp_4', 'set of obstacle'
    t1 = 10.5
    vref = 12.
    p0s = ((0., 0., 0., 0., 10.), )
    p1s = ((100., 0., 0., 0., 10.), )
    obstacles = ((30, -10, 20), (70, 15, 20), )
    cost, obj_scale = d2mou.CostComposit(kvel=1., kbank=1., kobs=1., kcol=_NAN, vsp=vref, obss=obstacles, obs_kind=1, rcol=3.), 1.
    x_constraint, y_constraint = None, None
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    v_constraint = (9., 18.)
    initial_guess = 'rnd'


class exp_4_1(exp_4):
    name, desc = 'exp_4_1', 'set of obstacles, size'
    v_constraint = (9., 15.)
    _obstacles = (((30, -10, 15), (30, 25, 15)), ((50, -10, 15), (50, 25, 15)), ((70, -10, 15), (70, 25, 15)))
    obj_scale = 1e-2
    ncases = len(_obstacles)

    def set_case(idx):
        exp_4_1.obstacles = exp_4_1._obstacles[idx]
        exp_4_1.cost, exp_4_1.obj_scale = d2mou.CostComposit(kvel=1., kbank=1., kobs=1., kcol=_NAN, vsp=14., obss=exp_4_1.obstacles,
                                                              obs_kind=1, rcol=3.), 1.

    def label(idx): return f'obstacles {exp_4_1._obstacles[idx]}'


class exp_4_2(exp_4):
    name, desc = 'exp_4_2', 'set of obstacles, duration'
    _durations = [9, 10, 11, 12]
    ncases = len(_durations)
    initial_guess = 'rnd'

    def set_case(idx): exp_4_2.t1 = exp_4_2._durations[idx]
    def label(idx): return f'duration {exp_4_2._durations[idx]} s'


class exp_5(exp_0):
    name, desc = 'exp_5', '2 aicraft face to face'
    t1 = 4.2
    vref = 12.
    dpsi = 0.
    p0s = ((0., 0., 0., 0., 12.), (50., 0., np.pi - dpsi, 0., 12.))
    p1s = ((50., 0., 0., 0., 12.), (0., 0., np.pi + dpsi, 0., 12.))
    x_constraint, y_constraint = None, None
    obstacles = []
    initial_guess = 'tri'
    ncases = 2

    def set_case(idx):
        kcol, rcol = (_NAN, 3.) if idx == 0 else (10., 10.)
        exp_5.cost, exp_5.obj_scale = d2mou.CostComposit(kvel=70., kbank=1., kobs=_NAN, kcol=kcol, vsp=exp_5.vref, obss=[], obs_kind=0,
                                                          rcol=rcol), 1.e0

    def label(idx): return f'obj {["Ref", "AntiCol"][idx]}'


class exp_5_1(exp_5):
    name, desc = 'exp_5', '2 aicraft next to one another'
    t1 = 8.
    vref = 12.
    p0s = ((0., 0., 0., 0., 12.), (0., 5., 0, 0., 12.))
    p1s = ((50., 50., np.pi / 2, 0., 12.), (55., 50., np.pi / 2, 0., 12.))
    initial_guess = 'tri'


_BANK0 = np.deg2rad(-2.00691223e+01)


class gvf_trial_3ac(exp_5):
    name, desc = 'gvf_trial_3ac', 'Circular formation with 3 aircraft - trial'
    hz = 10
    t1 = 5.5
    vref = 12
    dpsi = 0
    p0s = ((0, 40, 0., _BANK0, 12), (25, 20, 0., _BANK0, 12), (25, -20, 0., _BANK0, 12), (0, -40, 0., _BANK0, 12))
    p1s = ((75, 40, 0, 0, 12), (100, 20, 0, 0, 12), (100, -20, 0, 0, 12), (75, -40, 0, 0, 12))
    x_constraint, y_constraint = (-150, 150), (-150, 150)
    initial_guess = 'tri'
    ncases = 1
    cost, obj_scale = d2mou.CostComposit(kvel=70., kbank=1., kobs=_NAN, kcol=10., vsp=vref, obss=[], obs_kind=0, rcol=10), 1.e0


class inf_traj_4ac(exp_5):
    name, desc = 'inf trajectory', 'attempting some fancy inf-like traj'
    hz = 10
    t = [10]
    vref = 12
    dpsi = 0
    _b0, _b1 = np.deg2rad(20), np.deg2rad(-39)
    p0s = ((75, 40, 0., _b0, 12), (100, 40, 0., _b0, 12), (100, -40, 0., _b0, 12), (75, -40, 0., _b0, 12))
    p1s = ((75, -40, 0., _b1, 12), (100, -40, 0., _b1, 12), (100, 40, 0., _b1, 12), (75, 40, 0., _b1, 12))
    x_constraint, y_constraint = None, None
    initial_guess = 'tri'
    ncases = len(t)
    # (cost and obj_scale are whatever exp_5 carries -- exp_0's CostInput / 0.1 until exp_5.set_case ran: the reference assigns
    # its CostComposit to LOCAL names inside set_case, :427-429, which has no effect)

    def set_case(idx):
        exp_5.t1 = inf_traj_4ac.t[idx]       # (the reference sets the t1 of exp_5, which this class inherits)

    def label(idx): return f't_flight_{inf_traj_4ac.t[idx]}'


scens = [exp_0, exp_0_1, exp_1, exp_1_0, exp_1_1, exp_2, exp_3, exp_3_1, exp_4, exp_4_1, exp_4_2, exp_5, exp_5_1, gvf_trial_3ac,
         inf_traj_4ac]


def desc_all_scens():
    return '\n'.join(f'{i}: {s.name} {s.desc}' for i, s in enumerate(scens))


def get_scen(idx):
    return scens[idx]


def info_scen(idx):
    s = scens[idx]
    return f'{s.name} {s.desc}\ninitial states {s.t0} {s.p0s}\nfinal states {s.t1} {s.p1s}\n'
