"""Mirror of src/single_opt_planner.py: the single-aircraft trajectory planner, solved on the
GPU.  Where the reference hands a direct-collocation NLP to opty/IPOPT
(src/single_opt_planner.py:62-71,124), this Planner fits a 6-segment C^3 degree-7 polynomial
to the same objective (flat outputs -> (psi, phi, v), the kinematic constraints hold
identically) with libd2dhip's batched Levenberg-Marquardt solver, then samples it at the
reference's nodes so that `solution`, `sol_*`, save/load keep the reference's layout."""
import os

import numpy as np

import d2dhip
import d2d.opty_utils as d2ou
import d2d.multiopty_utils as d2mou
import d2d.optyplan_scenarios as d2oscen

seed = None
# Which solver stands behind Planner.prob:
#   'fit' -- the batched polynomial fit (flat outputs, soft bounds; the north-star path)
#   'nlp' -- opty.direct_collocation.Problem on the GPU: the reference's own parameterisation (node values, backward-Euler
#            equalities, HARD bounds), built by the very call the reference makes (src/single_opt_planner.py:62-71)
#   'auto' (default) -- the fit first; if its plan overshoots a bound of the scenario (phi, v or the x / y box: the fit's bounds are
#            soft rows) by more than AUTO_TOL, the collocation problem is solved from that plan, so that what a script gets from
#            Planner(scen).run() never violates a bound IPOPT would have enforced; info['backend_used'] says which one answered
BACKEND = 'auto'
AUTO_TOL = 1e-6        # rad, m/s, m
N_SEG = 6
W_WAYPOINT = 0.02      # weight of the 'tri' waypoint rows (regulariser)
W_BOUND = 1.0          # weight of the soft phi / v bound rows

_plans = {}


def get_plan(K, duration, obj_scale_over_n, kv=5., kphi=1.):
    """Fit plans are cached per (K, duration, whitening metric): the basis block is shared by every trajectory that is
    fitted with the same node grid AND the same cost weights (the metric is the quadratic skeleton of the cost; a plan built
    for other weights spans the same space but is conditioned differently, so it is not reused)."""
    ctx = d2dhip.default_context()
    s = obj_scale_over_n
    wref = (W_WAYPOINT ** 2, s * max(kv, 1e-3), s * max(kphi, 1e-3) / 9.81 ** 2)
    key = (K, round(float(duration), 9)) + tuple(float(f'{w:.6e}') for w in wref)
    if key not in _plans:
        _plans[key] = d2dhip.FitPlan(ctx, N_SEG, K, duration, wref)
    return _plans[key]


def lower_cost(cost):
    """Recognise the reference's cost plug-ins structurally -> (vsp, kv, kphi, kobs, obstacles, kcol, rcol,
    okind, bankmax): okind = bit mask of the obstacles of CostObstacle kind 0, bankmax = CostBank max mode.
    Anything else has no kernel and raises (there is no CPU solver to fall back to)."""
    nan = float('nan')
    if isinstance(cost, d2ou.CostAirVel):
        return cost.vsp, 1., 0., 0., (), nan, 0., 0, 0
    if isinstance(cost, d2ou.CostBank):
        return 0., 0., 1., 0., (), nan, 0., 0, 0 if cost.use_mean else 1
    if isinstance(cost, (d2ou.CostInput, d2mou.CostInput)):
        return cost.vsp, cost.kv, cost.kphi, 0., (), nan, 0., 0, 0
    if isinstance(cost, d2mou.CostNull):
        return 0., 0., 0., 0., (), nan, 0., 0, 0
    if isinstance(cost, (d2ou.CostObstacle, d2mou.CostObstacle)):          # one disc on its own (07_multioptyplan exp_3)
        return 0., 0., 0., 1., ((cost.c[0], cost.c[1], cost.r),), nan, 0., 1 if cost.kind == 0 else 0, 0
    if isinstance(cost, (d2ou.CostObstacles, d2mou.CostObstacles)):
        obss = tuple((o.c[0], o.c[1], o.r) for o in cost.obss)
        return 0., 0., 0., 1., obss, nan, 0., sum(1 << i for i, o in enumerate(cost.obss) if o.kind == 0), 0
    if isinstance(cost, d2mou.CostCollision):                              # the collision term on its own
        return 0., 0., 0., 0., (), 1., cost.r, 0, 0
    if isinstance(cost, d2ou.CostComposit):
        obss = [(o.c[0], o.c[1], o.r) for o in cost.cobs.obss] if hasattr(cost, 'cobs') else []
        okind = sum(1 << i for i, o in enumerate(cost.cobs.obss) if o.kind == 0) if obss else 0
        return cost.cvel.vsp, cost.kvel, cost.kbank, cost.kobs, tuple(obss), nan, 0., okind, 0 if cost.cbank.use_mean else 1
    if isinstance(cost, d2mou.CostComposit):
        obss = ()
        kobs = 0.
        okind = 0
        if not np.isnan(cost.kobs):
            obss = tuple((o[0], o[1], o[2]) for o in cost.obss)
            kobs = cost.kobs
            okind = ((1 << len(obss)) - 1) if cost.obs_kind == 0 else 0
        return cost.vsp, cost.kvel, cost.kbank, kobs, obss, cost.kcol, cost.rcol, okind, 0
    raise NotImplementedError(f'cost plug-in {type(cost).__name__} has no HIP lowering')


def scen_row(p0, p1, vref, lowered, s, wind, phi_c, v_c, go_left=-1., x_c=None, y_c=None):
    """One d2dhip scenario row.  The reference's symbolic model adds the wind with the opposite
    sign to the plant (src/d2d/opty_utils.py:42-44 vs src/d2d/dynamic.py:18-19); the planner keeps
    that convention, hence -w."""
    vsp, kv, kphi, kobs, obss = lowered[:5]
    okind, bankmax = lowered[7], lowered[8]
    if len(obss) > d2dhip.MAX_OBS:
        raise NotImplementedError(f'at most {d2dhip.MAX_OBS} static obstacles per trajectory in this build')
    r = np.zeros(d2dhip.SCEN_STRIDE)
    r[d2dhip.SC_X0], r[d2dhip.SC_Y0], r[d2dhip.SC_PSI0] = p0[0], p0[1], p0[2]
    r[d2dhip.SC_X1], r[d2dhip.SC_Y1], r[d2dhip.SC_PSI1] = p1[0], p1[1], p1[2]
    r[d2dhip.SC_VREF], r[d2dhip.SC_VSP] = vref, vsp
    r[d2dhip.SC_KV], r[d2dhip.SC_KPHI], r[d2dhip.SC_KOBS], r[d2dhip.SC_S] = kv, kphi, kobs, s
    r[d2dhip.SC_WWP], r[d2dhip.SC_WBND], r[d2dhip.SC_GOLEFT] = W_WAYPOINT, W_BOUND, go_left
    r[d2dhip.SC_WX], r[d2dhip.SC_WY] = -wind[0], -wind[1]
    for i, o in enumerate(obss):
        c = d2dhip.obs_col(i)
        r[c:c + 3] = o
    r[d2dhip.SC_PHIMAX] = max(abs(phi_c[0]), abs(phi_c[1]))
    r[d2dhip.SC_VMIN], r[d2dhip.SC_VMAX] = v_c
    r[d2dhip.SC_OKIND], r[d2dhip.SC_BANKMAX] = okind, bankmax
    # x/y_constraint boxes (src/single_opt_planner.py:56-57): soft bound rows like phi / v
    if x_c is not None:
        r[d2dhip.SC_XMIN], r[d2dhip.SC_XMAX] = x_c
    if y_c is not None:
        r[d2dhip.SC_YMIN], r[d2dhip.SC_YMAX] = y_c
    return r


def box_violation(x, y, x_constraint, y_constraint):
    """Largest distance (m) by which the sampled plan leaves the x/y_constraint boxes.  The boxes are soft bound
    rows of the fit (weight W_BOUND, like phi / v), so a binding box is honoured up to a small overshoot that
    the planners report as info['box_violation'] instead of hiding it."""
    v = 0.0
    for val, box in ((x, x_constraint), (y, y_constraint)):
        if box is not None:
            v = max(v, float(box[0] - np.min(val)), float(np.max(val) - box[1]))
    return max(v, 0.0)


def bound_violation(phi, v, phi_constraint, v_constraint):
    """Largest overshoot of the sampled plan beyond phi_constraint (rad) and v_constraint (m/s): like the boxes these are soft
    bound rows of the fit, and a scenario whose bounds bind hard (or conflict) ends in a compromise that the planners report as
    info['phi_violation'] / info['v_violation'] instead of hiding it."""
    vp = max(float(np.max(phi) - phi_constraint[1]), float(phi_constraint[0] - np.min(phi)), 0.0)
    vv = max(float(np.max(v) - v_constraint[1]), float(v_constraint[0] - np.min(v)), 0.0)
    return vp, vv


class _FitProblem:
    """Stands where the reference keeps its opty Problem (`Planner.prob`): num_free, the two
    option spellings the reference uses, and solve(x0) -> (solution, info)."""

    def __init__(self, planner, n_ac):
        self._p = planner
        self.num_free = 5 * planner.num_nodes * n_ac
        self.options = {'tol': 1e-8, 'max_iter': 200}

    def addOption(self, k, v):
        self.options[k] = v
    add_option = addOption

    def solve(self, x0):
        return self._p._solve(np.asarray(x0, dtype=np.float64))

    # opty's plotting helpers (the reference's `if plot:` branches): out of scope, refused by name
    def _no_plot(self, *_a, **_k):
        raise NotImplementedError('the opty plotting helpers are not part of this backend: plot Planner.sol_* with your own matplotlib code')
    plot_objective_value = plot_trajectories = plot_constraint_violations = _no_plot


class Planner:
    def __init__(self, exp, initialize=True, backend=None):
        self.exp = exp
        self.backend = backend or BACKEND
        self.obj_scale = exp.obj_scale
        self.num_nodes, self.time_step, self.duration = d2ou.planner_timing(exp.t0, exp.t1, exp.hz)
        self.wind = exp.wind
        self.aircraft = d2ou.Aircraft()
        N = self.num_nodes
        self._slice_x, self._slice_y, self._slice_psi, self._slice_phi, self._slice_v = (
            slice(i * N, (i + 1) * N, 1) for i in range(5))
        self.obstacles = exp.obstacles
        if initialize and self.backend == 'nlp':
            import opty.direct_collocation
            _g = self.aircraft
            t0, (x0, y0, psi0, phi0, v0) = exp.t0, exp.p0
            self._instance_constraints = (_g._sx(t0) - x0, _g._sy(t0) - y0, _g._spsi(t0) - psi0)
            t1, (x1, y1, psi1, phi1, v1) = exp.t1, exp.p1
            self._instance_constraints += (_g._sx(t1) - x1, _g._sy(t1) - y1, _g._spsi(t1) - psi1)
            self._bounds = {_g._sphi(_g._st): exp.phi_constraint, _g._sv(_g._st): exp.v_constraint}
            if exp.x_constraint is not None:
                self._bounds[_g._sx(_g._st)] = exp.x_constraint
            if exp.y_constraint is not None:
                self._bounds[_g._sy(_g._st)] = exp.y_constraint
            obj = exp.cost
            self.prob = opty.direct_collocation.Problem(lambda _free: obj.cost(_free, self),
                                                        lambda _free: obj.cost_grad(_free, self),
                                                        _g.get_eom(self.wind), _g._state_symbols, self.num_nodes, self.time_step,
                                                        known_parameter_map={}, instance_constraints=self._instance_constraints,
                                                        bounds=self._bounds, parallel=False)
        elif initialize:
            self.prob = _FitProblem(self, 1)

    def _harden(self):
        """backend='auto': the fit's plan overshoots a bound -> the collocation problem (hard bounds), started from that plan."""
        viol = max(self.info.get('box_violation', 0.), self.info.get('phi_violation', 0.), self.info.get('v_violation', 0.))
        self.info['backend_used'] = 'fit'
        if self.backend != 'auto' or not (viol > AUTO_TOL):
            return
        try:
            hard = type(self)(getattr(self, 'exp', None) or self.scen, initialize=True, backend='nlp')
            for k, v in self.prob.options.items():
                if k in ('tol',):
                    hard.prob.addOption(k, v)
            sol, info = hard.prob.solve(self.solution)
        except NotImplementedError as e:     # a cost / bound variant without a collocation kernel: the soft-bound plan stands, and says so
            self.info['nlp_status'] = f'unsupported: {e}'
            return
        st = np.atleast_1d(info['status'])
        info['fit_info'] = self.info
        if (st == 1).all():
            info['backend_used'] = 'nlp'
            self.solution, self.info = sol, info
            # the polynomial of the rejected soft-bound fit no longer describes the plan that is returned (info['fit_info'] keeps
            # its record): fit_q / fit_plan / fit_scen / fit_coefs are valid only when backend_used == 'fit'
            self.fit_q = self.fit_plan = self.fit_scen = self.fit_coefs = None
        else:                              # (e.g. a scenario whose bounds cannot all hold: the soft-bound compromise is what there is)
            self.info['nlp_status'] = info['status']

    def configure(self, tol=1e-8, max_iter=3000):
        self.prob.addOption('tol', tol)
        self.prob.addOption('max_iter', max_iter)

    def get_initial_guess(self, kind='tri'):
        """Node-vector guess [x, y, psi, phi, v] (src/single_opt_planner.py:79-115)."""
        g = np.zeros(self.prob.num_free)
        N = self.num_nodes
        if kind == 'rnd':
            rng = np.random.default_rng(seed)
            cx = self.exp.x_constraint or [-100, 100]
            cy = self.exp.y_constraint or [-100, 100]
            g[self._slice_x] = rng.uniform(cx[0], cx[1], N)
            g[self._slice_y] = rng.uniform(cy[0], cy[1], N)
            g[self._slice_psi] = rng.uniform(-np.pi, np.pi, N)
        elif kind == 'tri':
            x, y, psi, phi, v = d2ou.triangle(self.exp.p0[:2], self.exp.p1[:2], self.exp.vref, self.duration, N, go_left=-1.)
            g[self._slice_x], g[self._slice_y], g[self._slice_psi], g[self._slice_phi], g[self._slice_v] = x, y, psi, phi, v
        else:
            g[self._slice_x] = np.linspace(self.exp.p0[0], self.exp.p1[0], N)
            g[self._slice_y] = np.linspace(self.exp.p0[1], self.exp.p1[1], N)
        return g

    def _solve(self, x0):
        ctx = d2dhip.default_context()
        N = self.num_nodes
        low = lower_cost(self.exp.cost)
        s = self.obj_scale / N
        plan = get_plan(N, self.duration, s, low[1], low[2])
        row = scen_row(self.exp.p0, self.exp.p1, self.exp.vref, low, s, self.wind.w, self.exp.phi_constraint,
                       self.exp.v_constraint, x_c=self.exp.x_constraint, y_c=self.exp.y_constraint)
        dsc = ctx.dev(row[None, :])
        xy = np.stack([x0[self._slice_x], x0[self._slice_y]])[None]
        q = plan.project(dsc, ctx.dev(xy))
        max_iter = int(min(max(self.prob.options.get('max_iter', 200), 1), 2000))
        cost, iters, status, stats = plan.solve(dsc, q, max_iter=max_iter)
        _, Xs = plan.sample(dsc, q)
        Xh = Xs.cpu().numpy()[0]
        self.fit_q, self.fit_plan, self.fit_scen = q, plan, dsc
        self.fit_coefs = plan.coeffs(dsc, q).cpu().numpy()[0]
        info = {'status': int(status.cpu().numpy()[0]), 'iters': int(iters.cpu().numpy()[0]),
                'obj_val': float(cost.cpu().numpy()[0]),
                'box_violation': box_violation(Xh[0], Xh[1], self.exp.x_constraint, self.exp.y_constraint),
                'phi_violation': bound_violation(Xh[3], Xh[4], self.exp.phi_constraint, self.exp.v_constraint)[0],
                'v_violation': bound_violation(Xh[3], Xh[4], self.exp.phi_constraint, self.exp.v_constraint)[1], 'status_msg': ('running', 'converged', 'max_iter', 'non-finite', 'stalled')[int(status.cpu().numpy()[0])]}
        return Xs.cpu().numpy()[0].reshape(-1), info

    def run(self, initial_guess=None):
        if initial_guess is None:
            initial_guess = self.get_initial_guess('tri')
        self.solution, self.info = self.prob.solve(initial_guess)
        if self.backend != 'nlp':
            self._harden()
        self.interpret_solution()

    def interpret_solution(self):
        self.sol_time = np.linspace(0.0, self.duration, num=self.num_nodes)
        self.sol_x, self.sol_y = self.solution[self._slice_x], self.solution[self._slice_y]
        self.sol_psi, self.sol_phi, self.sol_v = (self.solution[s] for s in (self._slice_psi, self._slice_phi, self._slice_v))

    def save_solution(self, filename):
        wind = np.array([self.wind.sample_num(t, x, y) for t, x, y in zip(self.sol_time, self.sol_x, self.sol_y)])
        np.savez(filename, sol_time=self.sol_time, sol_x=self.sol_x, sol_y=self.sol_y, sol_psi=self.sol_psi,
                 sol_phi=self.sol_phi, sol_v=self.sol_v, wind=wind)
        print('saved {}'.format(filename))

    def load_solution(self, filename):
        d = np.load(filename)
        self.sol_time, self.sol_x, self.sol_y, self.sol_psi, self.sol_phi, self.sol_v = (
            d[k] for k in ('sol_time', 'sol_x', 'sol_y', 'sol_psi', 'sol_phi', 'sol_v'))
        print(f'loaded {filename}')


def compute_or_load(_p, force_recompute=False, filename='/tmp/optyplan.npz', tol=1e-5, max_iter=1500, initial_guess=None):
    """Result cache keyed by file name (src/single_opt_planner.py:152-164)."""
    if force_recompute or not os.path.exists(filename):
        _p.configure(tol, max_iter)
        _p.run(_p.get_initial_guess())
        _p.save_solution(filename)
    else:
        _p.load_solution(filename)


def plot2d(_p, _f=None, _a=None, label=''):
    import matplotlib.pyplot as plt
    _f = _f or plt.figure()
    _a = _a or plt.gca()
    _a.plot(_p.sol_x, _p.sol_y, solid_capstyle='butt', label=label)
    for cx, cy, rm in _p.obstacles:
        _a.add_patch(plt.Circle((cx, cy), rm, color='r', alpha=0.1))
    _a.axis('equal')
    return _f, _a


def plot_chrono(_p, _f=None, _a=None):
    import matplotlib.pyplot as plt
    if _f is None:
        _f, _a = plt.subplots(5, 1)
    for ax, (v, lab) in zip(_a, ((_p.sol_x, 'x'), (_p.sol_y, 'y'), (np.rad2deg(_p.sol_psi), 'psi'),
                                  (np.rad2deg(_p.sol_phi), 'phi'), (_p.sol_v, 'v'))):
        ax.plot(_p.sol_time, v); ax.set_ylabel(lab)
    return _f, _a


exp_0 = d2oscen.exp_0


class exp_1:
    """src/single_opt_planner.py:227-248: the leg of full-sim case 3; p0 is injected by the caller (src/12_full_sim_case3.py:185-190)."""
    name, desc = 'exp 1 - joining 2 points', 'single ac traj computation for test case 2 of full sim'
    ncases = 1
    tol, max_iter = 1e-5, 1500
    vref = 12
    cost, obj_scale = d2ou.CostAirVel(vref), 1
    wind = d2ou.WindField(w=[0, 0])
    obstacles = ()
    t0 = 0
    t1, p1 = 12, (75, 40, 0, 0, 12)
    x_constraint, y_constraint = (-150, 150), (-150, 150)
    v_constraint = (9., 15.)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    initial_guess = 'tri'
    hz = 10

    def set_case(idx): pass
    def label(idx): return ''
