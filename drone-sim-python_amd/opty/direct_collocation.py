"""`opty.direct_collocation.Problem` with the constructor and the members the reference uses (src/single_opt_planner.py:62-71,
76-81,124; src/multi_opt_planner.py:69-78,84-86): the direct-collocation NLP in node variables, solved by d2d_nlp_solve
(csrc/nlp_kernels.hip) instead of sympy code generation + IPOPT.

    Problem(obj, obj_grad, eom, state_symbols, num_nodes, time_step, known_parameter_map=, instance_constraints=, bounds=,
            parallel=False)
    .num_free, .addOption(k, v) / .add_option(k, v), .solve(x0) -> (solution, info)

What is interpreted instead of compiled:
  * eom / state_symbols  -- the fixed kinematic model of d2d.opty_utils.Aircraft.get_eom (an `Eom`: wind, g, aircraft count);
  * instance_constraints -- `x(t) - value` objects: the end conditions of every aircraft;
  * bounds               -- {phi(t): (lo, hi), v(t): ..., x(t): ..., y(t): ..., psi(t): ...}: HARD boxes (primal-dual barrier); the phi
                            interval need not be symmetric (d2d_nlp_opts.bounds carries it and the psi box to the kernel);
  * obj / obj_grad       -- the reference passes closures over a cost plug-in and the planner
                            (`lambda _free: obj.cost(_free, self)`); the plug-in is taken from the closure (or from the explicit
                            `cost=` / `planner=` keywords) and lowered structurally (single_opt_planner.lower_cost): the known
                            classes of d2d.opty_utils / d2d.multiopty_utils have a kernel, anything else raises NotImplementedError.
                            The gradient the solver follows is the plug-in's cost_grad (reference quirks included, oracle/nlp.py).
The whole Problem is ONE launch (d2d_nlp_solve_groups): wavefront a of a workgroup solves aircraft a.  The aircraft are coupled
through the objective only -- CostCollision acts on the pair of aircraft 0 and 1, src/d2d/multiopty_utils.py:124-125 -- so the
pair alternates on the device (each turn a full solve against the partner's frozen node positions) until neither moves by more than
options['sweep_tol'] (1e-7 m) or options['max_sweeps'] (12): a fixed point of that alternation is a KKT point of the joint NLP.  A
pair that has not settled is reported 'max_iter'."""
import numpy as np

import d2dhip


def _closure_objects(fn):
    out = []
    for c in getattr(fn, '__closure__', None) or ():
        try:
            out.append(c.cell_contents)
        except ValueError:
            pass
    return out


class Problem:
    def __init__(self, obj, obj_grad, eom, state_symbols, num_nodes, time_step, known_parameter_map=None,
                 instance_constraints=(), bounds=None, parallel=False, cost=None, planner=None):
        self.obj, self.obj_grad = obj, obj_grad
        self.num_nodes, self.time_step = int(num_nodes), float(time_step)
        self.n_aircraft = len(state_symbols) // 3
        self.num_free = 5 * self.num_nodes * self.n_aircraft
        self.options = {'tol': 1e-8, 'max_iter': 3000}
        self.wind = tuple(getattr(eom, 'wind', (0., 0.)))
        if getattr(eom, 'n_aircraft', self.n_aircraft) != self.n_aircraft:
            raise ValueError('eom and state_symbols disagree on the number of aircraft')
        # the cost plug-in and the planner behind the obj closure (the reference's call sites close over both)
        if cost is None or planner is None:
            for o in _closure_objects(obj):
                if cost is None and hasattr(o, 'cost') and hasattr(o, 'cost_grad'):
                    cost = o
                if planner is None and hasattr(o, 'num_nodes') and hasattr(o, 'obj_scale'):
                    planner = o
        if cost is None or planner is None:
            raise NotImplementedError('Problem: pass the objective as a closure over a d2d cost plug-in and the planner (as the '
                                      'reference does) or give cost= / planner= explicitly: arbitrary Python callables have no kernel')
        self.cost, self.planner = cost, planner
        # end conditions per aircraft from the instance constraints: names x<i>, y<i>, psi<i>
        ids = [str(s.sym.name)[1:] for s in state_symbols[0::3]]
        t_all = sorted({c.t for c in instance_constraints})
        if len(t_all) != 2:
            raise NotImplementedError('instance constraints at exactly two times (t0 and t1) are supported')
        self.p0s = np.zeros((self.n_aircraft, 3)); self.p1s = np.zeros((self.n_aircraft, 3))
        seen = set()
        for c in instance_constraints:
            for k, nm in enumerate(('x', 'y', 'psi')):
                for a, i in enumerate(ids):
                    if c.name == nm + i:
                        (self.p0s if c.t == t_all[0] else self.p1s)[a, k] = c.value
                        seen.add((a, k, c.t == t_all[0]))
        if len(seen) != 6 * self.n_aircraft:
            raise NotImplementedError('every aircraft needs x, y, psi fixed at t0 and t1 (src/single_opt_planner.py:46-49)')
        # bounds per aircraft
        self.bounds = [{} for _ in range(self.n_aircraft)]
        for key, (lo, hi) in (bounds or {}).items():
            for nm in ('phi', 'psi', 'v', 'x', 'y'):
                for a, i in enumerate(ids):
                    if key.sym.name == nm + i:
                        self.bounds[a][nm] = (float(lo), float(hi))
        for bd in self.bounds:
            if 'phi' not in bd or 'v' not in bd:
                raise NotImplementedError('phi and v bounds are required (the model divides by v)')

    def addOption(self, k, v):
        self.options[k] = v
    add_option = addOption

    # opty's plotting helpers (called by the reference's `if plot:` branches, src/06_optyplan.py:159-161, src/07_multioptyplan.py:
    # 90-92): plotting is out of scope -- refuse by name instead of failing with an AttributeError
    def _no_plot(self, *_a, **_k):
        raise NotImplementedError('opty.direct_collocation.Problem plotting helpers are not part of this backend: plot Planner.sol_* '
                                  '(x, y, psi, phi, v over sol_time) with your own matplotlib code')
    plot_objective_value = plot_trajectories = plot_constraint_violations = _no_plot

    def _rows(self):
        import single_opt_planner as sop
        low = sop.lower_cost(self.cost)
        n, N = self.n_aircraft, self.num_nodes
        multi = hasattr(self.planner, 'acs')
        s = self.planner.obj_scale / N / (n if multi else 1)
        rows = []
        for a in range(n):
            la = low if (a == 0 or not multi) else low[:4] + ((),) + low[5:]     # static obstacles act on aircraft 0 only (multi, :74)
            bd = self.bounds[a]
            r = sop.scen_row(tuple(self.p0s[a]) + (0., 0.), tuple(self.p1s[a]) + (0., 0.), 0., la, s, self.wind, bd['phi'], bd['v'],
                             x_c=bd.get('x'), y_c=bd.get('y'))
            if multi and la[4]:
                r[d2dhip.SC_KOBS] *= n                                             # obstacle scale has no 1/n_ac (:91)
            rows.append(r)
        rows = np.stack(rows)
        coupled = multi and n >= 2 and not np.isnan(low[5]) and low[5] > 0
        if coupled:
            rows[:2, d2dhip.SC_KCOL], rows[:2, d2dhip.SC_RCOL], rows[:2, d2dhip.SC_SCOL] = low[5], low[6], self.planner.obj_scale / N
        # (low[8]: CostBank(use_mean=False) travels as D2D_SC_BANKMAX -- obj_scale kbank max phi^2 with the maximiser frozen for the
        # length of a Newton step, csrc/nlp_kernels.hip nlp_assemble / oracle/nlp.py)
        return rows, coupled

    def solve(self, x0):
        ctx = d2dhip.default_context()
        n, N = self.n_aircraft, self.num_nodes
        x0 = np.asarray(x0, dtype=np.float64)
        sl = self.planner
        single = not isinstance(sl._slice_x, (list, tuple))
        get = (lambda s, a: x0[s]) if single else (lambda s, a: x0[s[a]])
        W = np.stack([np.stack([get(sl._slice_x, a), get(sl._slice_y, a), get(sl._slice_psi, a), get(sl._slice_phi, a),
                                get(sl._slice_v, a)], 0) for a in range(n)], 0)                       # (n, 5, N)
        rows, coupled = self._rows()
        # IPOPT's max_iter counts Newton steps; here they are grouped as outer (multiplier / barrier updates) x inner (<= D2D_NLP_INNER_MAX):
        # between 720 and 3600 steps in all, as in rounds 2-3 (12 .. 60 batches of 60)
        _im = d2dhip.NLP_INNER_MAX
        kw = dict(inner_max=_im, outer_max=int(min(max(self.options.get('max_iter', 3000) // _im, 720 // _im), 3600 // _im)))
        # IPOPT's `tol` (the reference sets 1e-5 .. 1e-8) bounds its scaled KKT error.  This backend's own tolerances -- barrier KKT
        # error of the last inner problem 1e-7, collocation residual 1e-9 -- are at least as tight as every value the reference
        # uses, so a looser `tol` changes nothing; a tighter one tightens them with it.
        tol = float(self.options.get('tol', 1e-8))
        kw.update(opt_tol=min(tol, 1e-7), feas_tol=min(1e-2 * tol, 1e-9))
        dsc = ctx.dev(rows)
        dW = ctx.dev(np.ascontiguousarray(W))
        # an asymmetric phi interval / a box on psi travel beside the rows (d2d_nlp_opts.bounds; lo >= hi: not set)
        bnd = None
        if any('psi' in bd or abs(bd['phi'][0] + bd['phi'][1]) > 1e-12 for bd in self.bounds):
            bnd = ctx.dev(np.array([[bd['phi'][0], bd['phi'][1]] + list(bd.get('psi', (0.0, 0.0))) for bd in self.bounds], dtype=np.float64))
        # one launch for the whole Problem: wavefront a of a workgroup solves aircraft a; the pair coupled by CostCollision
        # alternates on the device (block Gauss-Seidel, d2d_nlp_solve_groups) until neither aircraft moves
        if not coupled:
            dsc[:, d2dhip.SC_KCOL] = 0.0
        out = ctx.nlp_solve_groups(dsc, dW, self.time_step, n, max_sweeps=int(self.options.get('max_sweeps', 12)),
                                   tol=float(self.options.get('sweep_tol', 1e-7)), bounds=bnd, **kw)
        sweeps = int(out['sweeps'][0].item())
        moved = float(out['moved'][0].item())
        ctx.sync()
        Wh = dW.cpu().numpy()
        sol = np.zeros(self.num_free)
        for a in range(n):
            for c, s in enumerate((sl._slice_x, sl._slice_y, sl._slice_psi, sl._slice_phi, sl._slice_v)):
                sol[s if single else s[a]] = Wh[a, c]
        st = out['status'].cpu().numpy()
        info = {'status': int(st.max()) if single else st.tolist(), 'feas': float(out['feas'].max().item()),
                'iters': out['iters'].cpu().numpy().tolist(), 'sweeps': sweeps, 'moved': moved,
                'obj_val': float(self.obj(sol)), 'status_msg': 'converged' if (st == 1).all() else 'max_iter',
                'box_violation': 0.0, 'phi_violation': 0.0, 'v_violation': 0.0}       # hard bounds: an interior-point iterate never leaves its box
        return sol, info
