"""Planner scenario catalogue in the reference's scenario protocol (src/d2d/optyplan_scenarios.py:9-264): classes with class
attributes name, desc, t0/t1/hz, p0/p1 (x, y, psi, phi, v), wind, cost, obj_scale, x/y/phi/v_constraint, obstacles, vref, tol,
max_iter, ncases, set_case(idx), label(idx); the registry `scens` and desc_all / desc_one.  All fifteen entries of the
reference, with the reference's parameters (scenario constants are configuration data; the class-body side effects the
reference has -- exp_0_1 / exp_0_2 / exp_6 mutate exp_0 in set_case, exp_6 sets exp_0.t1 = 15 when its class body runs -- are
part of the protocol and are kept).

What the backends do with them: single_opt_planner.lower_cost / scen_row lower the cost plug-in and the bounds to one
scenario row; up to d2dhip.MAX_OBS static obstacles of either kind (exp_5 holds 12).  With backend='fit' the bounds are soft
rows, with backend='nlp' (opty.direct_collocation.Problem on the GPU) they are hard."""
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


class exp_0_3(exp_0):
    name, desc = 'exp0_3', 'xy constraints'
    tol, max_iter = 1e-5, 5000
    cost, obj_scale = d2ou.CostBank(), 1.e-1
    x_constraint, y_constraint = (-5., 45.), (-1., 51.)
    t1 = 20.


class exp_1(exp_0):
    name, desc = 'exp_1', 'combined phi/vel objective'
    t0, p0 = 0., (0., 0., 0., 0., 12.)
    t1, p1 = 10., (100., 0., 0., 0., 12.)
    cost, obj_scale = d2ou.CostInput(vsp=12., kvel=1., kbank=50.), 1.


class exp_1_1(exp_1):
    name, desc = 'exp_1_1', 'combined phi/vel objective'
    Ks = [[1., 0.5], [1., 1.], [1., 10.], [1., 20.], [1., 30.], [1., 40.], [1., 50.]]
    ncases = len(Ks)

    def set_case(idx):
        exp_1_1.K = exp_1_1.Ks[idx]
        exp_1_1.cost = d2ou.CostInput(vsp=12., kvel=exp_1_1.K[0], kbank=exp_1_1.K[1])

    def label(idx): return f'kvel, kbank {exp_1_1.K}'


class exp_421(exp_0):
    name, desc = 'exp1', 'min mean bank objective'
    cost = d2ou.CostBank()


class exp_2(exp_0):
    name, desc = 'exp2', 'bank/vel obective'
    cost = d2ou.CostComposit(None, 11., kobs=0., kvel=0.1, kbank=10.)


class exp_3(exp_2):
    name, desc = 'exp3', 'bank/vel obective, xy constraints'
    cost, obj_scale = d2ou.CostComposit(None, 12., kobs=0., kvel=0.5, kbank=1.), 1.
    x_constraint, y_constraint = (-5., 35.), (-5., 35.)
    t1 = 20.


class exp_4(exp_0):
    name, desc = 'exp4', 'obstacle - simple case'
    t0, p0 = 0., (0., 0., 0, 0., 10.)
    t1, p1 = 6.5, (50., 0., 0, 0., 10.)
    obstacles = ((25, -20, 10), )
    cost, obj_scale = d2ou.CostComposit(obstacles, vsp=15., kobs=0.5, kvel=0.5, kbank=1.), 1.e-2
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    x_constraint, y_constraint = (-5., 105.), (-15., 35.)
    v_constraint = (9., 15.)


class exp_4_1(exp_0):
    name, desc = 'exp4_1', 'obstacle - simple case'
    t0, p0 = 0., (0., 0., 0, 0., 10.)
    t1, p1 = 8.5, (100., 0., 0, 0., 10.)
    obstacles = ((50, -10, 25), )
    cost, obj_scale = d2ou.CostInput(vsp=12., kvel=0.5, kbank=1.), 1.e-2
    x_constraint, y_constraint = (-5., 105.), (-10., 40.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    v_constraint = (9., 15.)


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


class exp_4_3(exp_4_2):
    """(the reference carries this second copy of the maze; it is not in its registry either)"""
    obstacles, cost = _maze(15., kobs=0.5, kvel=0.5, kbank=1.)


class exp_5(exp_0):
    name = 'exp5'
    t0, p0 = 0., (0., 40., 0, 0., 10.)
    t1, p1 = 12., (100., 40., 0, 0., 10.)
    obstacles = [(20. * i, 20. * j, 10.) for i in range(5) for j in range(5) if (i + j) % 2]     # 12 discs, checkerboard
    cost = d2ou.CostComposit(obstacles, vsp=15., kobs=0.5, kvel=10., kbank=1.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))


class exp_6(exp_0):
    name, desc = 'exp6', 'Rendez-vous'
    p0s = tuple((x, y, np.pi / 2, 0., 10.) for x in (0, 10, 20) for y in (10, 20))
    ncases = len(p0s)
    p1s = [(0, 50, np.pi, 0., 10.) for _ in range(ncases)]
    exp_0.t1 = 15.                      # (class-body side effect of the reference, :204: every exp_0-based scenario sees it)
    x_constraint, y_constraint = (-50., 105.), (0, 100)

    def set_case(idx): exp_0.p0 = exp_6.p0s[idx]; exp_0.p1 = exp_6.p1s[idx]
    def label(idx): return f'{idx}'


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


scens = [exp_0, exp_0_1, exp_0_2, exp_0_3, exp_1, exp_1_1, exp_2, exp_3, exp_4, exp_4_1, exp_4_2, exp_5, exp_6, exp_13, exp_14]


def desc_all():
    return '\n'.join(f'{i}: {s.name} {s.desc}' for i, s in enumerate(scens))


def desc_one(idx):
    s = scens[idx]
    return f'{s.name} {s.desc}\ninitial state {s.t0} {s.p0}\nfinal state {s.t1} {s.p1}\n'
