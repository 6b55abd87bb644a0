"""Planner scenario catalogue in the reference's scenario protocol (src/d2d/optyplan_scenarios.py:
classes with class attributes name, desc, t0/t1/hz, p0/p1, wind, cost, obj_scale, bounds,
obstacles, vref, tol, max_iter, ncases, set_case(idx), label(idx)).  Scenario parameters are
the reference's; what the HIP fit can and cannot lower is decided in single_opt_planner.lower_cost /
scen_row (up to d2dhip.MAX_OBS static obstacles of either kind; x/y boxes are verified, not enforced).
A subset of the reference's catalogue: the cases its planners and the full simulation drive, one per
kind of cost / obstacle count."""
import numpy as np

import d2d.opty_utils as d2ou


class exp_0:
    name, desc = 'exp0', 'Turn around - 12m/s objective'
    ncases = 1
    tol, max_iter = 1e-5, 1500
    vref = 12.
    cost, obj_scale = d2ou.CostAirVel(vref), 1.
    wind = d2ou.WindField(w=[0., 0.])
    obstacles = ()
    t0, p0 = 0., (0., 0., 0., 0., 10.)
    t1, p1 = 10., (0., 30., np.pi, 0., 10.)
    x_constraint, y_constraint = None, None
    phi_constraint = (-np.deg2rad(30.), np.deg2rad(30.))
    v_constraint = (9., 14.)
    hz = 10.
    initial_guess = 'tri'

    def set_case(idx): pass
    def label(idx): return ''


class exp_0_1(exp_0):
    name, desc = 'exp0_1', 'changing duration'
    tol, max_iter = 1e-5, 5000
    t1s = [7., 10., 15., 20, 30]
    ncases = len(t1s)

    def set_case(idx): exp_0.t1 = exp_0_1.t1s[idx]
    def label(idx): return f'{exp_0_1.t1s[idx]:.1f} s'


class exp_0_2(exp_0):
    name, desc = 'exp0_2', 'changing wind'
    tol, max_iter = 1e-5, 5000
    winds = [[0., 0.], [1., 0.], [2., 0.], [5., 0.]]
    ncases = len(winds)

    def set_case(idx): exp_0.wind = d2ou.WindField(w=exp_0_2.winds[idx])
    def label(idx): return f'wind {exp_0_2.winds[idx]} m/s'


class exp_1(exp_0):
    name, desc = 'exp1', 'obstacles, composite cost'
    tol, max_iter = 1e-5, 5000
    t1, p1 = 10., (100., 0., 0., 0., 10.)
    obstacles = ((33, 0, 15), (66, 0, 15))
    cost, obj_scale = d2ou.CostComposit(obstacles, vsp=exp_0.vref, kobs=1., kvel=1., kbank=1., obs_kind=1), 1.


def _maze(vsp, **kw):
    discs = ((25, 0, 15), (55, 7.5, 12), (80, -10, 12))
    return discs, d2ou.CostComposit(discs, vsp=vsp, **kw)


class exp_4_2(exp_0):
    name, desc = 'exp4', 'obstacles - maze'
    t1, p1 = 15., (100., 0., 0, 0., 10.)
    obstacles, cost = _maze(15., kobs=0.5, kvel=0.5, kbank=1.)
    obj_scale = 1.e-2
    x_constraint, y_constraint = (-5., 105.), (-15., 35.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    v_constraint = (9., 15.)


class exp_5(exp_0):
    name = 'exp5'
    t0, p0 = 0., (0., 40., 0, 0., 10.)
    t1, p1 = 12., (100., 40., 0, 0., 10.)
    obstacles = [(20. * i, 20. * j, 10.) for i in range(5) for j in range(5) if (i + j) % 2]     # 12 discs, checkerboard
    cost = d2ou.CostComposit(obstacles, vsp=15., kobs=0.5, kvel=10., kbank=1.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))


class exp_13:
    name, desc = 'exp13 - some traj', 'just going'
    ncases = 1
    tol, max_iter = 1e-5, 1500
    vref = 12.
    cost, obj_scale = d2ou.CostAirVel(vref), 1.
    wind = d2ou.WindField(w=[0., 0.])
    obstacles = ()
    t0, p0 = 0., (75, 40, np.deg2rad(0), 0, 12)
    t1, p1 = 3., (100, 20, np.deg2rad(-90), 0, 12)
    x_constraint, y_constraint = None, None
    phi_constraint = (-np.deg2rad(30.), np.deg2rad(30.))
    v_constraint = (9., 14.)
    hz = 10.
    initial_guess = 'tri'

    def set_case(idx): pass
    def label(idx): return ''


class exp_14(exp_0):
    name, desc = 'exp 14 - joining 2 points', 'single ac traj computation for test case 2 of full sim'
    ncases = 1
    tol, max_iter = 1e-5, 1500
    vref = 12
    cost, obj_scale = d2ou.CostAirVel(vref), 1
    wind = d2ou.WindField(w=[0, 0])
    obstacles = ()
    t0, p0 = 0, (-49.98, -58.14, 2.22, -0.35, 15.)
    t1, p1 = 12, (75, 40, 0, 0, 12)
    x_constraint, y_constraint = (-150, 150), (-150, 150)
    v_constraint = (9., 15.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    initial_guess = 'tri'
    hz = 10

    def set_case(idx): pass
    def label(idx): return ''


scens = [exp_0, exp_0_1, exp_0_2, exp_1, exp_4_2, exp_5, exp_13, exp_14]


def desc_all():
    return '\n'.join(f'{i}: {s.name} {s.desc}' for i, s in enumerate(scens))


def desc_one(idx):
    s = scens[idx]
    return f'{s.name} {s.desc}\ninitial state {s.t0} {s.p0}\nfinal state {s.t1} {s.p1}\n'
