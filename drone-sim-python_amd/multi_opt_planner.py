"""Mirror of src/multi_opt_planner.py: the multi-aircraft planner.  All aircraft of a
scenario are fitted in one GPU batch (one wavefront per aircraft); see single_opt_planner
for how the reference's collocation NLP maps to the polynomial fit."""
import numpy as np

import d2dhip
import d2d.opty_utils as d2ou
import d2d.multiopty_utils as d2mou
import single_opt_planner as sop

seed = None


def scenario_rows(scen, p0s, p1s, N, duration, obj_scale, wind):
    """d2dhip scenario rows (n_ac, SCEN_STRIDE) of one multi-aircraft scenario, its fit plan and whether the aircraft
    are coupled by collision rows -- the lowering shared by Planner._solve and full_sim's on-device phase chain."""
    n = len(p0s)
    low = sop.lower_cost(scen.cost)
    coupled = not np.isnan(low[5]) and n >= 2
    if coupled and n > 8:
        raise NotImplementedError('collision coupling is built for groups of at most 8 aircraft')
    s = obj_scale / N / n                        # src/d2d/multiopty_utils.py:62
    plan = sop.get_plan(N, duration, s, low[1], low[2])
    rows = np.stack([sop.scen_row(p0, p1, scen.vref, low if i == 0 else low[:4] + ((),) + low[5:], s,
                                  wind, scen.phi_constraint, scen.v_constraint,
                                  x_c=scen.x_constraint, y_c=scen.y_constraint)
                     for i, (p0, p1) in enumerate(zip(p0s, p1s))])
    # (static obstacles act on aircraft 0 only, src/d2d/multiopty_utils.py:74)
    if low[4]:
        rows[:, d2dhip.SC_KOBS] *= n             # obstacle scale has no 1/n_ac (:91)
    if coupled:
        # CostCollision acts on the pair (aircraft 0, aircraft 1) only (src/d2d/multiopty_utils.py:124-125);
        # scale obj_scale/N without 1/n_ac (:132)
        rows[:, d2dhip.SC_KCOL], rows[:, d2dhip.SC_RCOL], rows[:, d2dhip.SC_SCOL] = low[5], low[6], obj_scale / N
        rows[0, d2dhip.SC_PMASK], rows[1, d2dhip.SC_PMASK] = 0b10, 0b01
    return rows, plan, coupled


class Planner:
    def __init__(self, scen, initialize=True, backend=None):
        self.scen = scen
        self.backend = backend or sop.BACKEND            # 'fit' | 'nlp' (single_opt_planner.BACKEND)
        self.obj_scale = scen.obj_scale
        self.wind = scen.wind
        self.acs = d2mou.AircraftSet(n=len(scen.p0s))
        self.num_nodes, self.time_step, self.duration = d2ou.planner_timing(scen.t0, scen.t1, scen.hz)
        N, n = self.num_nodes, self.acs.nb_aicraft
        # free-vector layout of the reference (src/multi_opt_planner.py:41-47)
        self._slice_x = [slice((0 + 3 * i) * N, (1 + 3 * i) * N, 1) for i in range(n)]
        self._slice_y = [slice((1 + 3 * i) * N, (2 + 3 * i) * N, 1) for i in range(n)]
        self._slice_psi = [slice((2 + 3 * i) * N, (3 + 3 * i) * N, 1) for i in range(n)]
        o = 3 * n * N
        self._slice_phi = [slice(o + i * N, o + (i + 1) * N, 1) for i in range(n)]
        o += n * N
        self._slice_v = [slice(o + i * N, o + (i + 1) * N, 1) for i in range(n)]
        if initialize and self.backend == 'nlp':
            import itertools
            import opty.direct_collocation

            def ic(_ac, _p, _t): return (_ac._sx(_t) - _p[0], _ac._sy(_t) - _p[1], _ac._spsi(_t) - _p[2])
            cons = [ic(_ac, _p, scen.t0) for _ac, _p in zip(self.acs.aircraft, scen.p0s)]
            cons += [ic(_ac, _p, scen.t1) for _ac, _p in zip(self.acs.aircraft, scen.p1s)]
            self._instance_constraints = tuple(itertools.chain(*cons))
            self._bounds = {}
            for _ac in self.acs.aircraft:
                self._bounds[_ac._sphi(_ac._st)] = scen.phi_constraint
                self._bounds[_ac._sv(_ac._st)] = scen.v_constraint
                if scen.x_constraint is not None:
                    self._bounds[_ac._sx(_ac._st)] = scen.x_constraint
                if scen.y_constraint is not None:
                    self._bounds[_ac._sy(_ac._st)] = scen.y_constraint
            obj = scen.cost
            self.prob = opty.direct_collocation.Problem(lambda _free: obj.cost(_free, self),
                                                        lambda _free: obj.cost_grad(_free, self),
                                                        self.acs.get_eom(self.wind), self.acs._state_symbols, self.num_nodes,
                                                        self.time_step, known_parameter_map={},
                                                        instance_constraints=self._instance_constraints, bounds=self._bounds, parallel=False)
        elif initialize:
            self.prob = sop._FitProblem(self, n)

    def get_initial_guess(self, what='rnd'):
        """(src/multi_opt_planner.py:94-112; 'rnd' draws y from the x range, as the reference does)."""
        g = np.zeros(self.prob.num_free)
        rng = np.random.default_rng(seed)
        N = self.num_nodes
        if what == 'rnd':
            for i in range(self.acs.nb_aicraft):
                cx = [-100, 100]
                g[self._slice_x[i]] = rng.uniform(cx[0], cx[1], N)
                g[self._slice_y[i]] = rng.uniform(cx[0], cx[1], N)
                g[self._slice_psi[i]] = rng.uniform(-np.pi, np.pi, N)
                g[self._slice_phi[i]] = rng.uniform(self.scen.phi_constraint[0], self.scen.phi_constraint[1], N)
                g[self._slice_v[i]] = rng.uniform(self.scen.v_constraint[0], self.scen.v_constraint[1], N)
        else:
            self.initial_guesses = [d2ou.triangle(np.array(p0)[:2], np.array(p1)[:2], self.scen.vref, self.duration, N, go_left=-1.)
                                    for p0, p1 in zip(self.scen.p0s, self.scen.p1s)]
            for i, ig in enumerate(self.initial_guesses):
                (g[self._slice_x[i]], g[self._slice_y[i]], g[self._slice_psi[i]], g[self._slice_phi[i]],
                 g[self._slice_v[i]]) = ig
        return g

    def _solve(self, x0):
        ctx = d2dhip.default_context()
        N, n = self.num_nodes, self.acs.nb_aicraft
        rows, plan, coupled = scenario_rows(self.scen, self.scen.p0s, self.scen.p1s, N, self.duration, self.obj_scale, self.wind.w)
        dsc = ctx.dev(rows)
        xy = np.stack([np.stack([x0[self._slice_x[i]], x0[self._slice_y[i]]]) for i in range(n)])
        q = plan.project(dsc, ctx.dev(xy))
        max_iter = int(min(max(self.prob.options.get('max_iter', 200), 1), 2000))
        if coupled:
            try:
                cost, sweeps, stats = plan.solve_groups(dsc, q, n, max_sweeps=max(1, max_iter // 8), inner_iters=8)
            finally:
                plan.set_groups(1)                   # the cached plan goes back to independent trajectories whatever happened
            torch = d2dhip._torch()
            iters = torch.full((n,), sweeps, dtype=torch.int32); status = torch.full((n,), 1 if stats[2] <= 1e-9 else 2, dtype=torch.int32)
        else:
            cost, iters, status, stats = plan.solve(dsc, q, max_iter=max_iter)
        _, Xs = plan.sample(dsc, q)
        Xs = Xs.cpu().numpy()                        # (n, 5, N)
        viol = sop.box_violation(Xs[:, 0], Xs[:, 1], self.scen.x_constraint, self.scen.y_constraint)
        self.fit_q, self.fit_plan, self.fit_scen = q, plan, dsc
        self.fit_coefs = plan.coeffs(dsc, q).cpu().numpy()
        sol = np.zeros(self.prob.num_free)
        for i in range(n):
            sol[self._slice_x[i]], sol[self._slice_y[i]], sol[self._slice_psi[i]] = Xs[i, 0], Xs[i, 1], Xs[i, 2]
            sol[self._slice_phi[i]], sol[self._slice_v[i]] = Xs[i, 3], Xs[i, 4]
        st = status.cpu().numpy()
        vphi, vv = sop.bound_violation(Xs[:, 3], Xs[:, 4], self.scen.phi_constraint, self.scen.v_constraint)
        info = {'status': st.tolist(), 'iters': iters.cpu().numpy().tolist(), 'obj_val': float(cost.sum().item()),
                'box_violation': viol, 'phi_violation': vphi, 'v_violation': vv}
        return sol, info

    def run(self, initial_guess=None, tol=1e-8, max_iter=500):
        if initial_guess is None:
            initial_guess = self.get_initial_guess('tri')
        self.prob.add_option('tol', tol)
        self.prob.addOption('max_iter', max_iter)
        self.solution, self.info = self.prob.solve(initial_guess)
        if self.backend == 'nlp':
            self.fit_q = self.fit_plan = self.fit_scen = self.fit_coefs = None
        else:
            sop.Planner._harden(self)      # backend='auto': a plan that overshoots a bound is re-planned by the collocation backend

    def interpret_solution(self):
        self.sol_time = np.linspace(0.0, self.duration, num=self.num_nodes)
        self.sol_x = [self.solution[s] for s in self._slice_x]
        self.sol_y = [self.solution[s] for s in self._slice_y]
        self.sol_psi = [self.solution[s] for s in self._slice_psi]
        self.sol_v = [self.solution[s] for s in self._slice_v]
        self.sol_phi = [self.solution[s] for s in self._slice_phi]


def plot2d(_p, _f=None, _a=None, label=''):
    import matplotlib.pyplot as plt
    _f = _f or plt.figure()
    _a = _a or plt.gca()
    for i in range(_p.acs.nb_aicraft):
        _a.plot(_p.sol_x[i], _p.sol_y[i], solid_capstyle='butt', label=label)
    for cx, cy, rm in _p.scen.obstacles:
        _a.add_patch(plt.Circle((cx, cy), rm, color='r', alpha=0.1))
    _a.axis('equal')
    return _f, _a


def plot_chrono(_p, _f=None, _a=None):
    import matplotlib.pyplot as plt
    if _f is None:
        _f, _a = plt.subplots(5, 1)
    for i in range(_p.acs.nb_aicraft):
        for ax, (v, lab) in zip(_a, ((_p.sol_x[i], 'x'), (_p.sol_y[i], 'y'), (np.rad2deg(_p.sol_psi[i]), 'psi'),
                                      (np.rad2deg(_p.sol_phi[i]), 'phi'), (_p.sol_v[i], 'v'))):
            ax.plot(_p.sol_time, v); ax.set_ylabel(lab)
    return _f, _a


# ---- scenarios (src/multi_opt_planner.py:170-242) ----------------------------------------
def export_csv(_p, filename):
    """The plan as the CSV the reference's scripts exchange (src/07_multioptyplan.py:476-489): columns time, then
    x_i, y_i, psi_i, phi_i, v_i per aircraft (1-based) -- what ExtractTrajData of 11_full_sim_case1.py reads back."""
    import pandas as pd
    states = {'time': _p.sol_time}
    for i in range(len(_p.sol_x)):
        for k, v in (('x', _p.sol_x), ('y', _p.sol_y), ('psi', _p.sol_psi), ('phi', _p.sol_phi), ('v', _p.sol_v)):
            states[f'{k}_{i + 1}'] = list(v[i])
    pd.DataFrame(states).to_csv(filename, index=False)


class exp_0:
    name, desc = 'exp_0', 'single aircraft'
    t0, t1, hz = 0., 10., 50.
    wind = d2ou.WindField()
    initial_guess = 'tri'
    tol, max_iter = 1e-5, 5000
    vref = 12.
    cost, obj_scale = d2mou.CostInput(vsp=vref, kv=5., kphi=1.), 1.e-1
    x_constraint, y_constraint = (-5, 50), (-5, 50)
    phi_constraint = (-np.deg2rad(40.), np.deg2rad(40.))
    v_constraint = (9., 15.)
    obstacles = []
    p0s, p1s = ((0., 0., 0., 0., 10.),), ((0., 50., 0., 0., 10.),)
    ncases = 1

    def set_case(idx): pass
    def label(idx): return ''


class exp_5(exp_0):
    name, desc = 'exp_5', '2 aicraft face to face'
    t1 = 4.2                                   # (hz = 50 inherited, as in the reference: 211 nodes)
    vref = 12.
    x_constraint, y_constraint = None, None
    p0s = ((0., 0., 0., 0., 12.), (50., 0., np.pi, 0., 12.))
    p1s = ((50., 0., 0., 0., 12.), (0., 0., np.pi, 0., 12.))
    ncases = 2

    def set_case(idx):
        nan = float('NaN')
        kcol = nan if idx == 0 else 10.
        exp_5.cost, exp_5.obj_scale = d2mou.CostComposit(kvel=70., kbank=1., kobs=nan, kcol=kcol, vsp=exp_5.vref, obss=[],
                                                          obs_kind=0, rcol=3. if idx == 0 else 10.), 1.e0

    def label(idx): return f'obj {["Ref", "AntiCol"][idx]}'


class trap_4(exp_5):
    """p0s / p1s / t1 are injected by the caller (src/11_full_sim_case1.py:444-447)."""
    name, desc = 'trap_4', 'trapezoidal formation with 4 aircraft'
    hz = 10
    vref = 12
    x_constraint, y_constraint = (-150, 150), (-150, 150)
    initial_guess = 'tri'
    ncases = 1
    cost, obj_scale = d2mou.CostComposit(kvel=70., kbank=1., kobs=float('NaN'), kcol=10., vsp=vref, obss=[], obs_kind=0, rcol=10), 1.e0

    def set_case(idx): pass
    def label(idx): return ''
