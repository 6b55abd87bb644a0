"""Script-level simulation functions of src/11_full_sim_case1.py (that file runs main() at
import, so its functions are mirrored here under their own names) with the time loops on the
GPU, plus batched variants over many independent formations.

  CircularFormationGVF(c, r, v, n_ac, X0f, ...)      src/11_full_sim_case1.py:93-177
  implement_controller(n_ac, time, x_ref, y_ref, ..)  :241-291
  ConstructBMatrix, ComputeDerivatives, ExtractTrajData, ExtendTraj_symm   :81-91, :197-239
  run_simulation(time, aircraft, windfield, ctl, X0, perts)                src/05_test_simulation.py:21-34 (legacy DFFF loop)
"""
import numpy as np

import d2dhip
import d2d.dynamic as ddyn

KE, KD, KR = 0.0004, 25, 20            # src/11_full_sim_case1.py:108-110
X1_START = np.array([20, 30, -np.pi / 2, 0, 10])     # :113


def ConstructBMatrix(n_ac):
    B = np.zeros((n_ac, n_ac - 1))
    for j in range(n_ac - 1):
        B[j, j], B[j + 1, j] = -1, 1
    return B


def _planes(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)


def hold_steps(t_opt_comp, t_step):
    """How many further steps the phase-error rule must hold before the reference's case-3 loop breaks: it adds t_step to a
    running dt on every such step and breaks once dt >= t_opt_comp (src/12_full_sim_case3.py:163-178); same floating-point sum."""
    k, acc = 0, 0.0
    while not acc >= t_opt_comp:
        acc += t_step
        k += 1
    return k


def CircularFormationGVF_batch(c, r, v, n_ac, X0f=None, t_start=0, t_step=0.05, t_end=1000, X0=None,
                               tau_phi=None, rec_stride=1, record=('X', 'U', 'Rr', 'eth'), W=(0., 0.), etheta_tol_deg=None,
                               t_opt_comp=0.0):
    """Many formations at once.  c (n_form, n_ac, 2) centres; r scalar or (n_form, n_ac); X0
    (n_form, n_ac, 5) or None (every aircraft starts at the reference's X1); X0f (n_form, n_ac, >=3)
    or None.  etheta_tol_deg: stop by the phase-error rule of cases 2 / 3 instead of the state rule (after t_opt_comp more seconds on
    which it holds).  Returns the raw device dictionary of d2dhip.Context.gvf_run plus `time`."""
    ctx = d2dhip.default_context()
    c = np.asarray(c, dtype=np.float64).reshape(-1, n_ac, 2)
    n_form = c.shape[0]
    N = n_form * n_ac
    time = np.arange(t_start, t_end, t_step)
    X0 = np.tile(X1_START, (N, 1)) if X0 is None else np.asarray(X0, dtype=np.float64).reshape(N, 5)
    R = np.broadcast_to(np.asarray(r, dtype=np.float64), (n_form, n_ac)).reshape(N) if np.ndim(r) else np.full(N, float(r))
    ac = ddyn.Aircraft()
    x0f = None if X0f is None else ctx.dev(_planes(np.asarray(X0f, dtype=np.float64).reshape(N, -1)[:, :3]))
    out = ctx.gvf_run(ctx.dev(_planes(X0)), ctx.dev(_planes(c.reshape(N, 2))), ctx.dev(np.ascontiguousarray(R)), n_ac,
                      len(time), t_step, float(v), KE, KD, KR, B=ConstructBMatrix(n_ac), z_des=np.zeros(max(n_ac - 1, 0)),
                      tau_phi=ac.tau_phi if tau_phi is None else tau_phi, tau_v=ac.tau_v, W=W, X0f=x0f,
                      rec_stride=rec_stride, record=record, etheta_tol_deg=etheta_tol_deg,
                      stop_hold=hold_steps(t_opt_comp, t_step) if etheta_tol_deg is not None else 0)
    out['time'] = time
    return out


def CircularFormationGVF(c, r, v, n_ac, X0f, t_start=0, t_step=0.05, t_end=1000):
    """One formation, the reference's 8-tuple: X_array, U_array, U1_array, U2_array, Ur_array,
    e_theta_array, time, t_f -- trimmed at the stop row like the reference (its U2 slice quirk,
    `U2_array[i,:]`, is reproduced)."""
    out = CircularFormationGVF_batch(np.asarray(c)[None], r, v, n_ac, X0f=np.asarray(X0f, dtype=float)[None],
                                     t_start=t_start, t_step=t_step, t_end=t_end)
    d2dhip.default_context().sync()
    time = out['time']
    i = int(out['stop_row'].cpu().numpy()[0])
    X = out['X'].cpu().numpy().transpose(0, 2, 1)
    U = out['U'].cpu().numpy().transpose(0, 2, 1)
    Rr = out['Rr'].cpu().numpy(); eth = out['eth'].cpu().numpy()
    # U1/U2 (debug decomposition of the GVF command) are not kept by the fused loop
    U1 = np.zeros((len(time), n_ac)); U2 = np.zeros((len(time), n_ac))
    if i < len(time):
        t_f = time[i - 1]
        return X[:i], U[:i], U1[:i], U2[i, :], Rr[:i], eth[:i], time[:i], t_f
    return X, U, U1, U2, Rr, eth, time, t_end


def _gvf_trimmed(out, n_ac, t_end):
    d2dhip.default_context().sync()
    time = out['time']
    rows = int(out['stop_row'].cpu().numpy()[0])
    X = out['X'].cpu().numpy().transpose(0, 2, 1); U = out['U'].cpu().numpy().transpose(0, 2, 1)
    Rr = out['Rr'].cpu().numpy(); eth = out['eth'].cpu().numpy()
    U1 = np.zeros((len(time), n_ac)); U2 = np.zeros((len(time), n_ac))        # (debug decomposition, not kept by the fused loop)
    fired = rows < len(time) or bool(out['conv_row'].cpu().numpy()[0] >= 0 and rows == len(time))
    return X, U, U1, U2, Rr, eth, time, rows, fired


def CircularFormationGVF_case2(c, r, v, n_ac, t_start=0, t_step=0.05, t_end=1000, etheta_tol=0.5):
    """src/12_full_sim_case2.py:85-164: the circular-formation phase that ends when every inter-vehicle phase error is <= 0.5 deg.
    The reference's 8-tuple, trimmed to [:i+1] (its `U2_array[i+1,:]` slice quirk reproduced), t_f = time[i-1]."""
    out = CircularFormationGVF_batch(np.asarray(c)[None], r, v, n_ac, t_start=t_start, t_step=t_step, t_end=t_end,
                                     etheta_tol_deg=etheta_tol)
    X, U, U1, U2, Rr, eth, time, rows, fired = _gvf_trimmed(out, n_ac, t_end)
    if fired and rows < len(time):
        return X[:rows], U[:rows], U1[:rows], U2[rows, :], Rr[:rows], eth[:rows], time[:rows], time[rows - 2]
    return X, U, U1, U2, Rr, eth, time, t_end


def CircularFormationGVF_case3(c, r, v, n_ac, t_start=0, t_step=0.05, t_end=1000, etheta_tol=2., t_opt_comp=0.7):
    """src/12_full_sim_case3.py:85-181: as case 2 with a 2 deg tolerance, flying on for t_opt_comp seconds after the first
    convergence (the time the planner takes); the last element is the reference's convergence = [index, t_convergence, t_f]."""
    out = CircularFormationGVF_batch(np.asarray(c)[None], r, v, n_ac, t_start=t_start, t_step=t_step, t_end=t_end,
                                     etheta_tol_deg=etheta_tol, t_opt_comp=t_opt_comp)
    X, U, U1, U2, Rr, eth, time, rows, fired = _gvf_trimmed(out, n_ac, t_end)
    idx = int(out['conv_row'].cpu().numpy()[0])
    if rows < len(time):
        conv = [idx, time[idx], time[rows - 2]]
        return X[:rows], U[:rows], U1[:rows], U2[rows, :], Rr[:rows], eth[:rows], time[:rows], conv
    return X, U, U1, U2, Rr, eth, time, [idx, time[idx] if idx >= 0 else None, None]


def trajectory_gen_multi(p, delta):
    """A second aircraft flying the single-aircraft plan shifted by delta (src/12_full_sim_case3.py:195-201): sol_x, sol_y (N, 2)."""
    x, y = p.sol_x.reshape(-1, 1), p.sol_y.reshape(-1, 1)
    p.sol_x, p.sol_y = np.append(x, x + delta[0], axis=1), np.append(y, y + delta[1], axis=1)
    return p


def trajectory_optimization_single(scen, delta, backend=None):
    """src/12_full_sim_case3.py:184-193 without the plots: one single-aircraft plan (scen.p0 injected by the caller, :456-460),
    duplicated with an offset for the wingman."""
    import single_opt_planner as sop
    p = sop.Planner(scen, backend=backend)
    p.configure(tol=1e-5, max_iter=1500)
    p.run(initial_guess=p.get_initial_guess())
    return trajectory_gen_multi(p, delta)


def ComputeDerivatives(x_ref, y_ref, dt):
    """Two passes of second-order-edge central differences (src/11_full_sim_case1.py:197-204); the
    tracking kernel computes the same on the device -- this host version serves callers that only
    want the derivatives."""
    Fdx = np.gradient(x_ref, edge_order=2) / dt
    Fdy = np.gradient(y_ref, edge_order=2) / dt
    return Fdx, Fdy, np.gradient(Fdx, edge_order=2) / dt, np.gradient(Fdy, edge_order=2) / dt


def ExtractTrajData(df, n_ac):
    """CSV columns time, x_i, y_i, psi_i (1-based) -> arrays (src/11_full_sim_case1.py:206-217)."""
    t = np.array(df['time'])
    cols = lambda k: np.stack([np.array(df[f'{k}_{i + 1}']) for i in range(n_ac)], 1)   # noqa: E731
    return t, cols('x'), cols('y'), cols('psi')


def ExtendTraj_symm(n_ac, x_ref, y_ref, psi_ref, time):
    """Append the mirrored half taken from the aircraft whose start equals this one's end
    (src/11_full_sim_case1.py:219-239; psi is extended with y values, as the reference does)."""
    time = np.append(time, time + time[-1])
    x0, xf, y0, yf = x_ref[0, :], x_ref[-1, :], y_ref[0, :], y_ref[-1, :]
    ax = [int(np.nonzero((x0 == xf[i]) & (y0 == yf[i]))[0][0]) for i in range(n_ac)]
    xs, ys = x_ref[:, ax], y_ref[:, ax]
    return time, np.append(x_ref, xs, axis=0), np.append(y_ref, ys, axis=0), np.append(psi_ref, ys, axis=0)


def implement_controller_batch(time, x_ref, y_ref, w, X0s, record=('X', 'U', 'Xr', 'dX', 'Yd', 'Ydd')):
    """x_ref, y_ref (T, n) for n independent drones; X0s (n, 5).  Device dictionary out."""
    ctx = d2dhip.default_context()
    ac = ddyn.Aircraft()
    dt = time[1] - time[0]
    return ctx.track_run(ctx.dev(np.ascontiguousarray(x_ref, dtype=np.float64)), ctx.dev(np.ascontiguousarray(y_ref, dtype=np.float64)),
                         ctx.dev(_planes(np.asarray(X0s, dtype=np.float64))), float(dt), record=record,
                         w=(float(w[0]), float(w[1])), tau_phi=ac.tau_phi, tau_v=ac.tau_v)


def implement_controller(n_ac, time, x_ref, y_ref, v, w, X0s):
    """The reference's 6-tuple X_array, U_array, X_ref_array, Yd_ref_array, Ydd_ref_array, dX_array
    (src/11_full_sim_case1.py:241-291), each (T, n_ac, .)."""
    out = implement_controller_batch(time, x_ref, y_ref, w, X0s)
    d2dhip.default_context().sync()
    t = lambda k: out[k].cpu().numpy().transpose(0, 2, 1)      # noqa: E731
    return t('X'), t('U'), t('Xr'), t('Yd'), t('Ydd'), t('dX')


def plan_batch(scen_rows, K, duration, obj_scale_over_n, q0=None, backend='fit', W0=None, h=None, n_ac=1, **solve_kw):
    """Batched planning entry point: scen_rows (B, d2dhip.SCEN_STRIDE) in the d2dhip layout -> dict with device
    tensors q, cost, iters, status and host stats (polynomial fit, backend='fit').
    backend='nlp': the reference's direct-collocation Problem (hard bounds) for B / n_ac scenarios of n_ac aircraft in one launch
    (d2d_nlp_solve_groups; CostCollision couples aircraft 0 and 1 of a scenario whose rows carry KCOL > 0): W0 (B, 5, K) node
    values of the initial guess, h the time step -> dict with device tensors W (the solution), cost, feas, iters, status per
    aircraft and sweeps, moved per scenario."""
    import single_opt_planner as sop
    ctx = d2dhip.default_context()
    if backend == 'nlp':
        dsc = ctx.dev(np.ascontiguousarray(scen_rows, dtype=np.float64))
        W = ctx.dev(np.ascontiguousarray(W0, dtype=np.float64))
        assert W.shape == (dsc.shape[0], 5, K) and h is not None
        out = ctx.nlp_solve_groups(dsc, W, float(h), int(n_ac), **solve_kw)
        out.update(W=W, scen=dsc)
        return out
    plan = sop.get_plan(K, duration, obj_scale_over_n)
    dsc = ctx.dev(np.ascontiguousarray(scen_rows, dtype=np.float64))
    q = plan.init(dsc) if q0 is None else q0
    cost, iters, status, stats = plan.solve(dsc, q, **solve_kw)
    return dict(plan=plan, scen=dsc, q=q, cost=cost, iters=iters, status=status, stats=stats)


def full_sim_phases_batch(c, r, v, n_ac, X1_f, scen, X2_f, t_opt, ref3=None, t_sim_end=200., w=(0., 0.), t_step=0.05,
                          t_end_1=1000., X0=None, max_sweeps=250, record2=('X', 'U'), record3=('X', 'U')):
    """The three phases of src/11_full_sim_case1.py main() (:406-478) for many independent formations, chained ON THE
    DEVICE: the circular-formation phase hands its final states to the planner as a device tensor, the planner's sampled
    plan is the tracking reference of phase 2 without leaving HBM, and phase 3 restarts from phase 2's final states.

      c (n_form, n_ac, 2) centres, r radius, v flight speed, X1_f (n_ac, >=3) or (n_form, n_ac, >=3) the formation
      that ends phase 1 (:421)
      scen        multi_opt_planner scenario class (trap_4, :444); its p0s come from phase 1, p1s = X2_f ((n_ac, >=3) or
                  (n_form, n_ac, >=3)), t1 = t_opt
      ref3        (time_3, x_ref_3, y_ref_3) of phase 3 -- e.g. ExtendTraj_symm(ExtractTrajData(csv)) -- or None
    Returns a dict of device tensors (plane-major, drone index = formation * n_ac + aircraft):
      phase1 (gvf_run dict), plan (q, cost, Xs [N][5][K]), phase2 (track_run dict), phase3 (list of track_run dicts)."""
    import multi_opt_planner as mop
    import d2d.opty_utils as d2ou
    ctx = d2dhip.default_context()
    torch = d2dhip._torch()
    c = np.asarray(c, dtype=np.float64).reshape(-1, n_ac, 2)
    n_form = c.shape[0]
    X1f = np.broadcast_to(np.asarray(X1_f, dtype=np.float64).reshape(-1, n_ac, np.shape(X1_f)[-1])[:, :, :3], (n_form, n_ac, 3))
    X2f = np.broadcast_to(np.asarray(X2_f, dtype=np.float64).reshape(-1, n_ac, np.shape(X2_f)[-1])[:, :, :3], (n_form, n_ac, 3))
    ph1 = CircularFormationGVF_batch(c, r, v, n_ac, X0f=X1f, t_step=t_step, t_end=t_end_1, X0=X0, record=())
    Xs1 = ph1['X_final']                                            # dev [5][N]: state at each formation's stop row
    # ---- phase 2: plan from where phase 1 ended (scenario rows finished on the device) ----
    scen.t1 = t_opt
    N2, dt2, dur2 = d2ou.planner_timing(scen.t0, scen.t1, scen.hz)
    rows, plan, coupled = mop.scenario_rows(scen, [(0., 0., 0., 0., 0.)] * n_ac, X2f[0], N2, dur2, scen.obj_scale, scen.wind.w)
    rows = np.tile(rows, (n_form, 1))
    rows[:, [d2dhip.SC_X1, d2dhip.SC_Y1, d2dhip.SC_PSI1]] = X2f.reshape(-1, 3)
    dsc = ctx.dev(rows)
    dsc[:, d2dhip.SC_X0], dsc[:, d2dhip.SC_Y0], dsc[:, d2dhip.SC_PSI0] = Xs1[0], Xs1[1], Xs1[2]
    q = plan.init(dsc)
    if coupled:
        try:
            cost, sweeps, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=max_sweeps, inner_iters=8)
        finally:
            plan.set_groups(1)
    else:
        cost, iters, status, stats = plan.solve(dsc, q)
    _, Xs = plan.sample(dsc, q)                                     # dev [N][5][K]
    x_ref2 = Xs[:, 0, :].t().contiguous(); y_ref2 = Xs[:, 1, :].t().contiguous()     # dev [K][N]
    ac = ddyn.Aircraft()
    kw = dict(w=(float(w[0]), float(w[1])), tau_phi=ac.tau_phi, tau_v=ac.tau_v)
    ph2 = ctx.track_run(x_ref2, y_ref2, Xs1, float(dt2), record=record2, **kw)
    out = dict(phase1=ph1, plan=dict(q=q, cost=cost, Xs=Xs, scen=dsc, stats=stats), phase2=ph2, phase3=[])
    # ---- phase 3: the periodic formation-flight reference, restarted from the last state until t_sim_end (:466-474) ----
    if ref3 is not None:
        time_3, x3, y3 = ref3
        x3 = ctx.dev(np.tile(np.ascontiguousarray(x3, dtype=np.float64), (1, n_form)))
        y3 = ctx.dev(np.tile(np.ascontiguousarray(y3, dtype=np.float64), (1, n_form)))
        dt3 = float(time_3[1] - time_3[0])
        # elapsed time: phase 1 ends per formation at its own stop row; the loop count follows the slowest formation
        stop = ph1['stop_row'].cpu().numpy()
        t_final = float((np.max(np.minimum(stop, len(ph1['time']))) - 1) * t_step + dur2)
        X_last = ph2['X_final']
        while t_final <= t_sim_end:
            ph3 = ctx.track_run(x3, y3, X_last, dt3, record=record3, **kw)
            out['phase3'].append(ph3)
            X_last = ph3['X_final']
            t_final += float(time_3[-1])
    return out


def run_simulation_batch(time, Yrefs, X0s, perts=None, w=(0., 0.), record=('X', 'U', 'Xr')):
    """The legacy DFFFController loop for n independent aircraft.  Yrefs (T, n, >=3, 2): each trajectory's traj.get(t) at
    the sample times; X0s (n, 5); perts (T, n, 5) or None.  Device dictionary out (plane-major [T][.][n])."""
    ctx = d2dhip.default_context()
    ac = ddyn.Aircraft()
    Y = np.asarray(Yrefs, dtype=np.float64)
    T, n = Y.shape[:2]
    Yd = np.ascontiguousarray(Y[:, :, :3, :].transpose(0, 2, 3, 1).reshape(T, 6, n))      # rows x, y, xd, yd, xdd, ydd
    dP = None if perts is None else ctx.dev(np.ascontiguousarray(np.asarray(perts, dtype=np.float64).transpose(0, 2, 1)))
    return ctx.dfff_run(ctx.dev(Yd), ctx.dev(_planes(np.asarray(X0s, dtype=np.float64))), float(time[1] - time[0]), perts=dP,
                        record=record, w=(float(w[0]), float(w[1])), tau_phi=ac.tau_phi, tau_v=ac.tau_v)


def run_simulation(time, aircraft, windfield, ctl, X0, perts):
    """One aircraft, the reference's triple X (T,5), U (T,2), Yref (T,4,2) (src/05_test_simulation.py:21-34); ctl is a
    d2d.guidance.DFFFController (its trajectory is sampled on the host, the time loop runs on the GPU)."""
    Yref = np.array([ctl.traj.get(t) for t in time])
    w = windfield.sample(time[0], Yref[0, 0])
    out = run_simulation_batch(time, Yref[:, None], np.asarray(X0, dtype=np.float64)[None], None if perts is None else np.asarray(perts)[:, None],
                               w=w, record=('X', 'U'))
    d2dhip.default_context().sync()
    return out['X'].cpu().numpy()[:, :, 0], out['U'].cpu().numpy()[:, :, 0], Yref


def sample_references_batch(trajs, time):
    """Flat outputs of many reference trajectories at the sample times, as run_simulation_batch wants them: device tensor
    [T][6][n].  Trajectories with a descriptor (lines, circle arcs, slaloms, min-snap polynomials, composites of them:
    d2d.trajectory.describe) are evaluated by d2d_traj_sample on the GPU -- the reference calls traj.get(t) in a Python loop per
    aircraft and step (src/05_test_simulation.py:25); the others (splines, space-indexed, tabulated) are sampled on the host and
    uploaded into their columns."""
    import d2d.trajectory as ddt
    ctx = d2dhip.default_context()
    time = np.asarray(time, dtype=np.float64)
    T, n = len(time), len(trajs)
    dt = float(time[1] - time[0])
    rows = [ddt.describe(tr) for tr in trajs]
    dev_ok = [r is not None for r in rows]
    desc = np.stack([r if r is not None else np.zeros(d2dhip.TRAJ_STRIDE) for r in rows])
    Y = ctx.traj_sample(ctx.dev(desc), T, float(time[0]), dt)
    for j, tr in enumerate(trajs):
        if not dev_ok[j]:
            Yh = np.array([np.asarray(tr.get(t))[:3].reshape(-1) for t in time])       # rows x, y, xd, yd, xdd, ydd
            Y[:, :, j] = ctx.dev(np.ascontiguousarray(Yh))
    return Y


def test_simulation(scen, record=('X', 'U')):
    """src/05_test_simulation.py:37-54 without the plots: every aircraft of a d2d.scenario.Scenario flown with the DFFFController
    along its reference, all of them in ONE device loop.  Returns Xs, Us (lists of (T, 5) / (T, 2) arrays, one per aircraft)
    and Yrefs (T, n, 3, 2)."""
    ctx = d2dhip.default_context()
    time = np.asarray(scen.time, dtype=np.float64)
    n = len(scen.trajs)
    Y = sample_references_batch(scen.trajs, time)                                        # dev [T][6][n]
    ac = scen.aircrafts[0]
    w = scen.windfield.sample(time[0], None)
    perts = ctx.dev(np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float64) for p in scen.perts[:n]], 2)))   # [T][5][n]
    X0 = np.stack([np.asarray(x, dtype=np.float64) for x in scen.X0s[:n]])
    out = ctx.dfff_run(Y, ctx.dev(_planes(X0)), float(time[1] - time[0]), perts=perts, record=record,
                       w=(float(w[0]), float(w[1])), tau_phi=ac.tau_phi, tau_v=ac.tau_v)
    ctx.sync()
    Xh, Uh, Yh = out['X'].cpu().numpy(), out['U'].cpu().numpy(), Y.cpu().numpy()
    Yrefs = Yh.transpose(0, 2, 1).reshape(len(time), n, 3, 2)
    return [Xh[:, :, j] for j in range(n)], [Uh[:, :, j] for j in range(n)], Yrefs
