"""GPU parity: the direct-collocation NLP backend (d2d_nlp_solve, csrc/nlp_kernels.hip) -- the reference's own parameterisation
(node values, backward-Euler equalities, end conditions, hard boxes: src/single_opt_planner.py:35-71) -- against
  * the committed IPOPT output of the reference for exp_14 (tests/golden/planner_goldens.npz: cost 5.02972817),
  * the oracle's solver (oracle/nlp.py: the same algorithm in numpy with a banded LAPACK factorisation),
  * KKT conditions evaluated by the oracle with the reference's cost_grad."""
import numpy as np
import pytest

from oracle import nlp, costs as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import d2dhip
    c = d2dhip.Context(0)
    yield c
    c.close()


def _row(pb, obstacles=(), kobs=0.0, okind=0):
    """d2dhip scenario row of an oracle Problem."""
    import d2dhip as D
    r = np.zeros(D.SCEN_STRIDE)
    r[D.SC_X0:D.SC_X0 + 3] = pb.p0; r[D.SC_X1:D.SC_X1 + 3] = pb.p1
    r[D.SC_VSP], r[D.SC_KV], r[D.SC_KPHI], r[D.SC_S], r[D.SC_KOBS] = pb.vsp, pb.kv, pb.kphi, pb.s, kobs
    r[D.SC_WX], r[D.SC_WY] = -pb.wind[0], -pb.wind[1]
    r[D.SC_PHIMAX] = pb.hi[1, 3]; r[D.SC_VMIN], r[D.SC_VMAX] = pb.lo[1, 4], pb.hi[1, 4]
    if np.isfinite(pb.lo[1, 0]):
        r[D.SC_XMIN], r[D.SC_XMAX] = pb.lo[1, 0], pb.hi[1, 0]
    if np.isfinite(pb.lo[1, 1]):
        r[D.SC_YMIN], r[D.SC_YMAX] = pb.lo[1, 1], pb.hi[1, 1]
    for i, o in enumerate(obstacles):
        c = D.obs_col(i)
        r[c:c + 3] = o
    r[D.SC_OKIND] = okind
    return r


def _solve(ctx, pbs, W0s, rows, **kw):
    """W0s: (N, 5) node values per problem -> device [B][5][N]; returns the solution as (N, 5, B) and mult as (N, 3, B)."""
    W = ctx.dev(np.ascontiguousarray(np.stack([w.T for w in W0s], 0)))               # (B, 5, N)
    out = ctx.nlp_solve(ctx.dev(np.stack(rows)), W, pbs[0].h, want_mult=True, **kw)
    ctx.sync()
    res = {k: v.cpu().numpy() for k, v in out.items() if k != 'work'}
    res['mult'] = np.ascontiguousarray(res['mult'].transpose(2, 1, 0))
    return np.ascontiguousarray(W.cpu().numpy().transpose(2, 1, 0)), res


def test_exp14_reproduces_the_reference_ipopt_cost(ctx, gold):
    """optyplan_scenarios.exp_14 (src/d2d/optyplan_scenarios.py:219-253): 121 nodes, CostAirVel(12), phi in +-40 deg, v in [9, 15],
    boxes +-150, from the reference's 'tri' initial guess.  The committed IPOPT solution has cost 5.02972817 and satisfies the
    collocation to 2e-8; ours: the same cost to 1e-6 relative, feasibility <= 1e-8, KKT residual <= 1e-5 -- hard bounds held exactly."""
    g = gold('planner_goldens')
    N, h = 121, 0.1
    p0 = (-49.98, -58.14, 2.22, -0.35, 15.); p1 = (75, 40, 0, 0, 12)
    pb = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=0., obj_scale=1., phi_max=np.deg2rad(40.), v_min=9., v_max=15.,
                     x_box=(-150, 150), y_box=(-150, 150))
    Wg = nlp.from_free(g['exp14_free'], N)
    assert abs(nlp.cost(pb, Wg) - float(g['exp14_cost_airvel12'])) < 1e-12
    W0 = nlp.from_free(C.single_guess('tri', p0, p1, 12., 12.0, N), N)
    W, out = _solve(ctx, [pb], [W0], [_row(pb)])
    W = W[:, :, 0]
    assert out['status'][0] == 1
    assert abs(out['cost'][0] - 5.02972817) <= 1e-6 * 5.02972817, out['cost'][0]
    assert abs(out['cost'][0] - nlp.cost(pb, W)) <= 1e-12 and out['feas'][0] <= 1e-8
    assert np.abs(nlp.constraints(pb, W)).max() <= 1e-8
    # hard bounds: never violated, and they DO bind
    assert (W[:, 3] >= -np.deg2rad(40.)).all() and (W[:, 3] <= np.deg2rad(40.)).all() and (W[:, 4] <= 15.).all() and (W[:, 4] >= 9.).all()
    assert (W[:, 4] < 9. + 1e-5).sum() + (np.abs(W[:, 3]) > np.deg2rad(40.) - 1e-5).sum() >= 3
    np.testing.assert_allclose(W[0, :3], p0[:3], atol=0); np.testing.assert_allclose(W[-1, :3], p1[:3], atol=0)
    # KKT with the multipliers the solver returns (lambda = 2 rho mu; rho ends at its start value unless feasibility stalled)
    Wo, info = nlp.solve(pb, W0)
    assert abs(info['cost'] - out['cost'][0]) <= 1e-8 * info['cost']
    assert np.abs(W - Wo).max() <= 1e-5                                   # same algorithm, same path (banded LAPACK vs block Cholesky)
    assert abs(int(out['iters'][0]) - info['inner']) <= 10, (out['iters'][0], info['inner'])
    # the committed IPOPT node values: same trajectory to a few millimetres (IPOPT stopped at tol 1e-5; flat directions)
    assert np.abs(W[:, :2] - Wg[:, :2]).max() < 5e-3 and np.abs(W[:, 4] - Wg[:, 4]).max() < 5e-3


def test_batch_with_obstacles_wind_and_boxes_vs_oracle(ctx):
    """A ragged batch of different problems in one launch (each wavefront its own scenario): obstacles of both kinds, wind, a
    binding y box; every one against the oracle's solve and against the KKT conditions."""
    N, h = 41, 0.1
    pbs, rows, W0s, obs = [], [], [], []
    rng = np.random.default_rng(4)
    for i in range(7):
        p0 = (0., 0., rng.uniform(-0.5, 0.5), 0., 12.); p1 = (48. + rng.uniform(-4, 4), rng.uniform(-6, 6), rng.uniform(-0.4, 0.4), 0., 12.)
        ob = [(24. + rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(4, 7))] if i % 2 else []
        kind = 0 if i == 3 else 1
        # (i == 3: a kind-0 disc of 4-7 m radius ON the straight line: the dog-leg guess starts inside its 1e3 clip, where the
        # reference's cost_grad is the gradient of the paraboloid the solver continues the term with)
        pb = nlp.Problem(N, h, p0, p1, vsp=12., kv=5., kphi=1., obj_scale=0.1 if i < 4 else 1.0, wind=(1.0, -0.5) if i == 2 else (0., 0.),
                         phi_max=np.deg2rad(35.), v_min=9., v_max=15., y_box=(-6.5, 9.) if i == 5 else None, obstacles=ob,
                         kobs=1.0 if ob else 0.0, obs_kind=kind)
        pbs.append(pb); rows.append(_row(pb, ob, 1.0 if ob else 0.0, 1 if kind == 0 and ob else 0))
        W0s.append(nlp.from_free(C.single_guess('tri', p0, p1, 12., (N - 1) * h, N), N)); obs.append(ob)
    W, out = _solve(ctx, pbs, W0s, rows)
    for i, pb in enumerate(pbs):
        Wi = W[:, :, i]
        assert out['status'][i] == 1, (i, out)
        assert out['feas'][i] <= 1e-8 and np.abs(nlp.constraints(pb, Wi)).max() <= 1e-8
        assert abs(out['cost'][i] - nlp.cost(pb, Wi)) <= 1e-11 * max(1.0, out['cost'][i])
        assert (Wi >= pb.lo - 1e-15).all() and (Wi <= pb.hi + 1e-15).all()
        Wo, info = nlp.solve(pb, W0s[i])
        assert info['status'] == 1
        assert abs(info['cost'] - out['cost'][i]) <= 1e-7 * max(info['cost'], 1e-3), (i, info['cost'], out['cost'][i])
        assert np.abs(Wi - Wo).max() <= 1e-4, (i, np.abs(Wi - Wo).max())
        # stationarity of the Lagrangian with the reference's cost_grad: the kernel's point and multipliers, the bound duals of the
        # oracle's run (the kernel keeps its duals in its workspace)
        lam = 2 * info['rho'] * out['mult'][1:, :, i]
        kkt, feas = nlp.kkt_residual(pb, Wi, lam, info['zL'], info['zU'])
        assert kkt <= 1e-5 and feas <= 1e-8, (i, kkt)


@pytest.mark.parametrize('N', [3, 5, 63, 64, 65, 129, 200])
def test_ragged_node_counts_vs_oracle(ctx, N):
    """Node counts around the chunk size of the node-parallel phases (64) and the unroll of the serial recursions (4), down to the
    smallest problem the entry point takes (3 nodes: one free position): a side-step at 12 m/s, against the oracle's solve."""
    h = 0.1
    # (cost of order one: with obj_scale = 1 the 1/N-scaled cost of a gentle 200-node leg is 1e-3 and its minimiser is not
    # determined to the solver's tolerances -- the oracle itself moves by 0.5 m when the guess moves by 1e-9)
    p0 = (0., 0., 0., 0., 12.); p1 = (12. * h * (N - 1) * 0.995, (0.01 if N <= 5 else 0.15) * (N - 1), 0., 0., 12.)
    pb = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=5., obj_scale=float(N), phi_max=np.deg2rad(30.), v_min=9., v_max=15.)
    W0 = np.stack([np.linspace(p0[0], p1[0], N), np.linspace(p0[1], p1[1], N), np.zeros(N), np.zeros(N), np.full(N, 12.)], 1)
    W, out = _solve(ctx, [pb], [W0], [_row(pb)])
    W = W[:, :, 0]
    Wo, info = nlp.solve(pb, W0)
    assert info['status'] == 1 and out['status'][0] == 1, (info['status'], out['status'])
    assert out['feas'][0] <= 1e-8 and np.abs(nlp.constraints(pb, W)).max() <= 1e-8
    assert abs(info['cost'] - out['cost'][0]) <= 1e-7 * max(info['cost'], 1e-6) and np.abs(W - Wo).max() <= 1e-5
    np.testing.assert_allclose(W[0, :3], p0[:3], atol=0); np.testing.assert_allclose(W[-1, :3], p1[:3], atol=0)


def test_infeasible_problem_gives_up_like_the_oracle(ctx):
    """Three nodes and a side-step that needs 39 deg of bank against a 30 deg bound: no feasible point.  Kernel and oracle both stop
    with status 4 (stalled at the largest penalty) instead of running outer_max x inner_max steps, bounds never violated."""
    N, h = 3, 0.1
    p0 = (0., 0., 0., 0., 12.); p1 = (12. * h * 2 * 0.995, 0.08, 0., 0., 12.)
    pb = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=0.5, obj_scale=1., phi_max=np.deg2rad(30.), v_min=9., v_max=15.)
    W0 = np.stack([np.linspace(p0[0], p1[0], N), np.linspace(p0[1], p1[1], N), np.zeros(N), np.zeros(N), np.full(N, 12.)], 1)
    W, out = _solve(ctx, [pb], [W0], [_row(pb)])
    Wo, info = nlp.solve(pb, W0)
    assert info['status'] == 4 and out['status'][0] == 4, (info['status'], out['status'])
    assert out['feas'][0] > 1e-3 and (np.abs(W[:, 3, 0]) <= np.deg2rad(30.)).all()
    assert abs(int(out['iters'][0]) - info['inner']) <= 5


def _feas(sol_x, sol_y, sol_psi, sol_phi, sol_v, h, wind=(0., 0.)):
    free = np.concatenate([sol_x, sol_y, sol_psi, sol_phi, sol_v])
    return np.abs(C.collocation_residual(free, len(sol_x), h, wind)).max()


def test_single_planner_on_the_collocation_backend():
    """single_opt_planner.Planner(exp_14, backend='nlp'): Planner.prob IS opty.direct_collocation.Problem, built by the call the
    reference makes (cost closures, eom, state symbols, instance constraints, bounds: src/single_opt_planner.py:62-71), and
    .solve(x0) returns the reference's free vector: cost = the committed IPOPT run's, collocation feasible, bounds hard."""
    import d2d.optyplan_scenarios as d2oscen
    import opty.direct_collocation
    import single_opt_planner as sop
    p = sop.Planner(d2oscen.exp_14, initialize=True, backend='nlp')
    assert isinstance(p.prob, opty.direct_collocation.Problem) and p.prob.num_free == 5 * 121
    p.configure(1e-5, 1500)
    p.run(p.get_initial_guess('tri'))
    assert p.solution.shape == (605,) and p.info['status'] == 1, p.info
    c = d2oscen.exp_14.cost.cost(p.solution, p)
    assert abs(c - 5.02972817) <= 1e-6 * 5.02972817 and abs(p.info['obj_val'] - c) < 1e-14
    assert _feas(p.sol_x, p.sol_y, p.sol_psi, p.sol_phi, p.sol_v, p.time_step) <= 1e-8
    assert np.abs(p.sol_phi).max() <= np.deg2rad(40.) and p.sol_v.min() >= 9. and p.sol_v.max() <= 15.
    np.testing.assert_allclose([p.sol_x[0], p.sol_y[0], p.sol_psi[0], p.sol_x[-1], p.sol_y[-1], p.sol_psi[-1]],
                               list(d2oscen.exp_14.p0[:3]) + list(d2oscen.exp_14.p1[:3]), atol=0)
    # wind: the symbolic model's +w convention (src/d2d/opty_utils.py:42-44)
    import d2d.opty_utils as d2ou

    class windy(d2oscen.exp_14):                       # (a head wind of 2 m/s would need v > 15 on this leg: infeasible)
        wind = d2ou.WindField(w=[-1., 0.5])
    pw = sop.Planner(windy, initialize=True, backend='nlp')
    pw.run()
    assert pw.info['status'] == 1 and _feas(pw.sol_x, pw.sol_y, pw.sol_psi, pw.sol_phi, pw.sol_v, pw.time_step, (-1., 0.5)) <= 1e-8
    assert abs(pw.info['obj_val'] - 2.90645965) < 1e-6                      # (oracle/nlp.py on the same problem)
    # obstacles: CostComposit with two discs (kind 1) on the straight line
    class discs(d2oscen.exp_0):
        t1, p1 = 10., (100., 0., 0., 0., 10.)
        obstacles = ((33, 0, 15), (66, 0, 15))
        cost, obj_scale = d2ou.CostComposit(obstacles, vsp=12., kobs=1., kvel=1., kbank=1., obs_kind=1), 1.
    po = sop.Planner(discs, initialize=True, backend='nlp')
    po.run()
    assert po.info['status'] == 1 and _feas(po.sol_x, po.sol_y, po.sol_psi, po.sol_phi, po.sol_v, po.time_step) <= 1e-8
    d = np.minimum(np.hypot(po.sol_x - 33., po.sol_y), np.hypot(po.sol_x - 66., po.sol_y))
    assert d.min() > 5.0, d.min()                      # the straight line through both discs is left


def test_multi_planner_on_the_collocation_backend_like_11_full_sim():
    """multi_opt_planner.Planner(trap_4, backend='nlp') as src/11_full_sim_case1.py:444-447 drives it (4 aircraft, 6 s, 61 nodes,
    CostComposit with the collision term on the pair (0, 1)): every aircraft collocation-feasible with hard bounds; the
    reference's cost of the coupled plan is not worse than the uncoupled plan's."""
    import multi_opt_planner as mop
    import d2d.multiopty_utils as d2mou
    scen = mop.trap_4
    scen.t1 = 6
    scen.p0s = ((0, 40, 0, 0, 12), (25, 40, 0, 0, 12), (25, -40, 0, 0, 12), (0, -40, 0, 0, 12))
    scen.p1s = ((75, 40, 0, 0, 12), (100, 40, 0, 0, 12), (100, -40, 0, 0, 12), (75, -40, 0, 0, 12))
    keep = scen.cost
    try:
        _p = mop.Planner(scen, initialize=True, backend='nlp')
        _p.run(initial_guess=_p.get_initial_guess(scen.initial_guess), tol=scen.tol, max_iter=scen.max_iter)
        _p.interpret_solution()
        assert _p.solution.shape == (5 * 4 * 61,) and all(s == 1 for s in _p.info['status']), _p.info
        for i in range(4):
            assert _feas(_p.sol_x[i], _p.sol_y[i], _p.sol_psi[i], _p.sol_phi[i], _p.sol_v[i], _p.time_step) <= 1e-8
            assert np.abs(_p.sol_phi[i]).max() <= np.deg2rad(40.) and _p.sol_v[i].min() >= 9. and _p.sol_v[i].max() <= 15.
            np.testing.assert_allclose([_p.sol_x[i][0], _p.sol_y[i][0], _p.sol_x[i][-1], _p.sol_y[i][-1]],
                                       [scen.p0s[i][0], scen.p0s[i][1], scen.p1s[i][0], scen.p1s[i][1]], atol=0)
        c_coupled = keep.cost(_p.solution, _p)
        # 75 m in 6 s: 12.5 m/s on average -> 70 * mean (v - 12)^2 ~ 17.5 over the four aircraft
        assert 10.0 < c_coupled < 25.0, c_coupled
        scen.cost = d2mou.CostComposit(kvel=70., kbank=1., kobs=float('NaN'), kcol=float('NaN'), vsp=12., obss=[], obs_kind=0, rcol=10)
        _q = mop.Planner(scen, initialize=True, backend='nlp')
        _q.run(initial_guess=_q.get_initial_guess(scen.initial_guess), tol=scen.tol, max_iter=scen.max_iter)
        assert c_coupled <= keep.cost(_q.solution, _q) + 1e-6
    finally:
        scen.cost = keep


def test_catalogue_scenarios_on_the_collocation_backend():
    """A cross-section of the reference's single-aircraft catalogue (src/d2d/optyplan_scenarios.py) through
    Planner(scen, backend='nlp'): turn-around (exp_0), its 5 m/s wind case (exp_0_2[3]), bank/velocity composite (exp_2), kind-0
    obstacles whose clip region the path has to cross or skirt (exp_4: a 6.5 s leg that needs a detour; exp_5: 12-disc checkerboard),
    exp_4_1, rendez-vous (exp_6[0]).  Converged, collocation feasible to 1e-8, hard bounds held, end conditions exact.
    (tools/nlp_catalogue.py surveys all 33 cases: 30 converge; the others are listed in DESIGN.md 5.8.)"""
    import d2d.optyplan_scenarios as sc
    import single_opt_planner as sop
    keep = {k: getattr(sc.exp_0, k) for k in ('t1', 'wind', 'p0', 'p1')}
    try:
        for s, case in ((sc.exp_0, 0), (sc.exp_0_2, 3), (sc.exp_2, 0), (sc.exp_4, 0), (sc.exp_4_1, 0), (sc.exp_5, 0), (sc.exp_6, 0)):
            for k, v in keep.items():
                setattr(sc.exp_0, k, v)
            s.set_case(case)
            p = sop.Planner(s, initialize=True, backend='nlp')
            p.run(p.get_initial_guess('tri'))
            assert p.info['status'] == 1, (s.__name__, p.info)
            w = tuple(np.asarray(s.wind.w, float)[:2])
            assert _feas(p.sol_x, p.sol_y, p.sol_psi, p.sol_phi, p.sol_v, p.time_step, w) <= 1e-8, s.__name__
            assert np.abs(p.sol_phi).max() <= s.phi_constraint[1] and p.sol_v.min() >= s.v_constraint[0] and p.sol_v.max() <= s.v_constraint[1]
            np.testing.assert_allclose([p.sol_x[0], p.sol_y[0], p.sol_psi[0], p.sol_x[-1], p.sol_y[-1], p.sol_psi[-1]],
                                       list(s.p0[:3]) + list(s.p1[:3]), atol=0)
    finally:
        for k, v in keep.items():
            setattr(sc.exp_0, k, v)


def _trap4_like_the_golden(gold, tag='st_line'):
    """multi_opt_planner.trap_4 with the end poses and horizon of a committed four-aircraft plan (tests/golden/
    planner_feasibility_goldens.npz, from the reference's src/*.csv): the scenario src/11_full_sim_case1.py:444-447 builds."""
    import multi_opt_planner as mop
    g = gold('planner_feasibility_goldens')
    W = g[tag + '_W']                                    # (4, 5, N)
    t = g[tag + '_time']
    import d2d.multiopty_utils as d2mou
    scen = mop.trap_4
    keep = (scen.t1, scen.p0s, scen.p1s, scen.cost)
    scen.cost = d2mou.CostComposit(kvel=70., kbank=1., kobs=float('NaN'), kcol=10., vsp=12., obss=[], obs_kind=0, rcol=10)
    scen.t1 = float(t[-1])
    scen.p0s = tuple((W[a, 0, 0], W[a, 1, 0], W[a, 2, 0], 0., 12.) for a in range(4))
    scen.p1s = tuple((W[a, 0, -1], W[a, 1, -1], W[a, 2, -1], 0., 12.) for a in range(4))
    return scen, keep, W, g


def test_joint_multi_aircraft_problem_reproduces_the_committed_plan(gold):
    """The reference's multi-aircraft Problem (src/multi_opt_planner.py:69-78,86: all aircraft in ONE NLP, CostCollision between
    aircraft 0 and 1) on the device in one launch (d2d_nlp_solve_groups), started from the reference's committed IPOPT output
    src/opt_states_st_line.csv (4 aircraft, 71 nodes): the solve stays at that plan -- the reference's cost of our solution is
    its cost of the committed one, 0.2024378405, within 1e-5 and not above it (IPOPT stopped at tol 1e-5), every aircraft collocation-
    feasible to 1e-8 with its hard bounds held, the pair settled."""
    import multi_opt_planner as mop
    scen, keep, Wg, g = _trap4_like_the_golden(gold)
    try:
        _p = mop.Planner(scen, initialize=True, backend='nlp')
        assert _p.num_nodes == 71 and _p.prob.num_free == 5 * 4 * 71
        x0 = np.zeros(_p.prob.num_free)
        for a in range(4):
            for c, s in enumerate((_p._slice_x, _p._slice_y, _p._slice_psi, _p._slice_phi, _p._slice_v)):
                x0[s[a]] = Wg[a, c]
        c_gold = scen.cost.cost(x0, _p)
        assert abs(c_gold - 0.2024378405) <= 1e-9                    # the mirrored cost plug-in on the committed plan (the known answer)
        _p.run(initial_guess=x0, tol=scen.tol, max_iter=scen.max_iter)
        _p.interpret_solution()
        assert all(s == 1 for s in _p.info['status']), _p.info
        assert _p.info['sweeps'] <= 12 and _p.info['moved'] <= 1e-7
        c = scen.cost.cost(_p.solution, _p)
        # (measured: 0.20243398 -- 3.9e-6 BELOW the committed run's 0.20243784: IPOPT stopped on its tol = 1e-5, this solve goes on to 1e-7)
        assert abs(c - 0.2024378405) <= 1e-5 and c <= 0.2024378405, c
        for i in range(4):
            assert _feas(_p.sol_x[i], _p.sol_y[i], _p.sol_psi[i], _p.sol_phi[i], _p.sol_v[i], _p.time_step) <= 1e-8
            assert np.abs(_p.sol_phi[i]).max() <= scen.phi_constraint[1] and _p.sol_v[i].min() >= scen.v_constraint[0] and _p.sol_v[i].max() <= scen.v_constraint[1]
            # the plan itself: node positions within centimetres of the committed ones (the cost sees v and phi only: flat directions)
            assert np.abs(_p.sol_x[i] - Wg[i, 0]).max() <= 0.05 and np.abs(_p.sol_y[i] - Wg[i, 1]).max() <= 0.05
        # from the reference's own 'tri' guess instead: the same cost level (a minimum at least as good)
        _q = mop.Planner(scen, initialize=True, backend='nlp')
        _q.run(initial_guess=_q.get_initial_guess('tri'), tol=scen.tol, max_iter=scen.max_iter)
        assert all(s == 1 for s in _q.info['status']), _q.info
        assert scen.cost.cost(_q.solution, _q) <= c * (1 + 1e-3)
    finally:
        scen.t1, scen.p0s, scen.p1s, scen.cost = keep


def test_joint_problems_in_batches_equal_the_single_launches(gold):
    """d2d_nlp_solve_groups over R scenarios = R separate launches (the scenarios are independent), coupled pair and uncoupled aircraft
    alike; a scenario whose first row carries KCOL = 0 is uncoupled; full_sim.plan_batch(backend='nlp') is the batched entry point."""
    import d2dhip
    import full_sim
    import multi_opt_planner as mop
    scen, keep, Wg, g = _trap4_like_the_golden(gold)
    try:
        _p = mop.Planner(scen, initialize=True, backend='nlp')
        rows, coupled = _p.prob._rows()
        assert coupled and rows.shape == (4, d2dhip.SCEN_STRIDE)
        x0 = _p.get_initial_guess('tri')
        W0 = np.stack([np.stack([x0[s[a]] for s in (_p._slice_x, _p._slice_y, _p._slice_psi, _p._slice_phi, _p._slice_v)]) for a in range(4)])
        R = 6
        rng = np.random.default_rng(5)
        allrows = np.tile(rows, (R, 1)); allW = np.tile(W0, (R, 1, 1))
        for r in range(R):                       # move aircraft 1 towards aircraft 0 so that the collision term matters more or less
            allrows[4 * r + 1, [d2dhip.SC_Y0, d2dhip.SC_Y1]] -= rng.uniform(0.0, 25.0)
            allW[4 * r + 1, 1] = np.linspace(allrows[4 * r + 1, d2dhip.SC_Y0], allrows[4 * r + 1, d2dhip.SC_Y1], W0.shape[2])
        allrows[4 * (R - 1):, d2dhip.SC_KCOL] = 0.0                       # the last scenario: no coupling
        ctx = d2dhip.default_context()
        out = full_sim.plan_batch(allrows, W0.shape[2], None, None, backend='nlp', W0=allW, h=_p.time_step, n_ac=4)
        Wb = out['W'].cpu().numpy()
        st = out['status'].cpu().numpy(); sw = out['sweeps'].cpu().numpy()
        assert (st == 1).all(), st
        assert sw[-1] == 0 and (sw[:-1] >= 1).all() and (out['moved'].cpu().numpy() <= 1e-7).all()
        for r in (0, 3, R - 1):
            d1 = ctx.dev(allrows[4 * r:4 * r + 4].copy()); W1 = ctx.dev(allW[4 * r:4 * r + 4].copy())
            o1 = ctx.nlp_solve_groups(d1, W1, _p.time_step, 4)
            ctx.sync()
            assert np.array_equal(W1.cpu().numpy(), Wb[4 * r:4 * r + 4])             # same kernel, same inputs: same bits
            assert np.array_equal(o1['cost'].cpu().numpy(), out['cost'].cpu().numpy()[4 * r:4 * r + 4])
        # the coupling acts: with aircraft 1 flown close to aircraft 0 the pair keeps more distance than the uncoupled plans would
        from oracle import nlp as ON
        for r in range(R - 1):
            dmin = np.hypot(Wb[4 * r, 0] - Wb[4 * r + 1, 0], Wb[4 * r, 1] - Wb[4 * r + 1, 1]).min()
            assert dmin > 1.0
            # each aircraft of the pair is a KKT point of ITS sub-problem against the partner's final positions (= joint KKT):
            # the oracle's solver started at the kernel's answer with the partner frozen stays there
            for a, o in ((0, 1), (1, 0)):
                pb = ON.problem_from_row(allrows[4 * r + a], W0.shape[2], _p.time_step)
                pb.partner = Wb[4 * r + o, :2].T.copy()
                Wo, info = ON.solve(pb, Wb[4 * r + a].T.copy())
                assert info['status'] == 1 and abs(info['cost'] - out['cost'].cpu().numpy()[4 * r + a]) <= 1e-7 * max(1.0, info['cost'])
                assert np.abs(Wo[:, :2] - Wb[4 * r + a, :2].T).max() <= 1e-4
            if r >= 1:
                break
    finally:
        scen.t1, scen.p0s, scen.p1s, scen.cost = keep


def test_unusable_rows_are_refused_at_once():
    """PHIMAX = 0 (an unset row), VMIN >= VMAX or VMIN <= 0 gave a zero-width box and NaN pivots for outer_max x 30
    assemblies; such a problem now reports D2D_ST_NONFINITE with NaN cost immediately, its neighbours in the batch are unaffected."""
    import d2dhip
    ctx = d2dhip.default_context()
    N, h = 31, 0.1
    row = np.zeros(d2dhip.SCEN_STRIDE)
    row[d2dhip.SC_X1] = 35.0; row[d2dhip.SC_VSP], row[d2dhip.SC_KV], row[d2dhip.SC_S] = 12., 1., 1. / N
    row[d2dhip.SC_PHIMAX], row[d2dhip.SC_VMIN], row[d2dhip.SC_VMAX] = np.deg2rad(30.), 9., 15.
    rows = np.tile(row, (4, 1))
    rows[1, d2dhip.SC_PHIMAX] = 0.0
    rows[2, d2dhip.SC_VMIN] = 16.0
    rows[3, d2dhip.SC_VMIN] = 0.0
    W0 = np.stack([np.linspace(0, 35, N), np.zeros(N), np.zeros(N), np.zeros(N), np.full(N, 12.)])
    W = ctx.dev(np.tile(W0, (4, 1, 1)))
    out = ctx.nlp_solve(ctx.dev(rows), W, h)
    ctx.sync()
    st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy(); c = out['cost'].cpu().numpy()
    assert st[0] == 1 and np.isfinite(c[0])
    assert (st[1:] == d2dhip.ST_NONFINITE).all() and (it[1:] == 0).all() and np.isnan(c[1:]).all()


def test_cyclic_reduction_equals_the_serial_recursion():
    """d2d_nlp_opts.serial: the reduced block-tridiagonal system of a Newton step by block cyclic reduction (default; records in the LDS up
    to 121 nodes, in global memory beyond) and by round 2's twisted serial recursion are the same Cholesky solve in another elimination
    order: same verdicts, costs and node values to rounding on a batch with converged and infeasible problems, at horizons on both sides of
    the LDS limit and at the smallest ones (3 .. 6 nodes: levels with a single eliminated node, a missing right neighbour)."""
    import d2dhip
    from d2dhip import synth
    ctx = d2dhip.default_context()
    for N in (121, 150, 64, 65, 6, 5, 4, 3):
        rows, W0, h = synth.nlp_problems(48, N=max(N, 8))
        if N < 8:                                   # a short, gently curved leg of N nodes at h = 0.1 s
            W0 = np.ascontiguousarray(W0[:, :, :N]); h = 0.1
            leg = 11.0 * h * (N - 1)
            rows[:, d2dhip.SC_X1] = rows[:, d2dhip.SC_X0] + leg * np.cos(rows[:, d2dhip.SC_PSI0] + 0.02)
            rows[:, d2dhip.SC_Y1] = rows[:, d2dhip.SC_Y0] + leg * np.sin(rows[:, d2dhip.SC_PSI0] + 0.02)
            rows[:, d2dhip.SC_PSI1] = rows[:, d2dhip.SC_PSI0] + 0.04
            rows[:, d2dhip.SC_S] = 1.0 / N
            for b in range(len(rows)):
                W0[b, 0] = np.linspace(rows[b, d2dhip.SC_X0], rows[b, d2dhip.SC_X1], N); W0[b, 1] = np.linspace(rows[b, d2dhip.SC_Y0], rows[b, d2dhip.SC_Y1], N)
                W0[b, 2] = np.linspace(rows[b, d2dhip.SC_PSI0], rows[b, d2dhip.SC_PSI1], N); W0[b, 3] = 0.0; W0[b, 4] = 11.0
        dsc = ctx.dev(rows)
        Wa, Wb = ctx.dev(np.ascontiguousarray(W0)), ctx.dev(np.ascontiguousarray(W0))
        oa = ctx.nlp_solve(dsc, Wa, h, serial=0)
        ob = ctx.nlp_solve(dsc, Wb, h, serial=1)
        ctx.sync()
        sa, sb = oa['status'].cpu().numpy(), ob['status'].cpu().numpy()
        assert (sa == sb).all(), (N, sa, sb)
        ca, cb = oa['cost'].cpu().numpy(), ob['cost'].cpu().numpy()
        conv = sa == 1
        assert conv.sum() >= len(sa) // 2, (N, sa)
        np.testing.assert_allclose(ca[conv], cb[conv], rtol=1e-9, atol=1e-12)
        # (node values along the objective's flat directions -- the cost sees v only -- differ at the 1e-5 level between two rounding paths)
        assert np.abs(Wa.cpu().numpy()[conv] - Wb.cpu().numpy()[conv]).max() <= 1e-3
        assert float(oa['feas'].cpu().numpy()[conv].max()) <= 1e-8


def test_asymmetric_phi_interval_and_psi_box_vs_oracle(ctx):
    """opty's bounds dict may hold any interval (src/single_opt_planner.py:53): an asymmetric bank interval and a box on the heading
    travel beside the scenario rows (d2d_nlp_opts.bounds).  A left turn-around with phi in [-5, +35] deg (the right bank nearly
    forbidden) and psi in [-0.2, pi + 0.2]: same cost and nodes as the oracle with the same lo / hi, both bounds held, the upper
    bank bound binding; problem 1 of the batch has no override and stays on its row's symmetric interval."""
    N, h = 71, 0.1
    p0 = (0., 0., 0., 0., 12.); p1 = (0., 40., np.pi, 0., 12.)
    pb = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=0.5, obj_scale=1., phi_max=np.deg2rad(35.), v_min=9., v_max=15.)
    pa = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=0.5, obj_scale=1., phi_max=np.deg2rad(35.), v_min=9., v_max=15.)
    philo, phihi, psilo, psihi = -np.deg2rad(5.), np.deg2rad(35.), -0.2, np.pi + 0.2
    pa.lo[:, 3], pa.hi[:, 3] = philo, phihi
    pa.lo[1:-1, 2], pa.hi[1:-1, 2] = psilo, psihi
    W0 = nlp.from_free(C.single_guess('tri', p0, p1, 12., (N - 1) * h, N), N)
    bnd = ctx.dev(np.array([[philo, phihi, psilo, psihi], [0., 0., 0., 0.]]))
    W, out = _solve(ctx, [pa, pb], [W0, W0], [_row(pb), _row(pb)], bounds=bnd)
    assert (out['status'] == 1).all(), out['status']
    Wa, Wb = W[:, :, 0], W[:, :, 1]
    assert Wa[:, 3].min() >= philo and Wa[:, 3].max() <= phihi and Wa[1:-1, 2].min() >= psilo and Wa[1:-1, 2].max() <= psihi
    assert (Wa[:, 3] > phihi - 1e-4).sum() >= 3                         # the bank limit binds in the turn
    for p_, W_, k in ((pa, Wa, 0), (pb, Wb, 1)):
        Wo, info = nlp.solve(p_, W0)
        assert info['status'] == 1
        assert abs(info['cost'] - out['cost'][k]) <= 1e-7 * info['cost'], (k, info['cost'], out['cost'][k])
        assert np.abs(W_ - Wo).max() <= 1e-4
        assert np.abs(nlp.constraints(p_, W_)).max() <= 1e-8
    assert Wb[:, 3].min() < philo - 1e-3 or abs(out['cost'][0] - out['cost'][1]) < 1e-9       # the override changed problem 0 only
    # ... and through the reference's plug-point: Problem(bounds={phi: (lo, hi), psi: (lo, hi)})
    import contextlib, io
    import d2d.optyplan_scenarios as sc
    import single_opt_planner as sop

    class turn(sc.exp_0):
        t1, p1 = 7.0, (0., 40., np.pi, 0., 12.)
        p0 = (0., 0., 0., 0., 12.)
        phi_constraint = (philo, phihi)
    with contextlib.redirect_stdout(io.StringIO()):
        p = sop.Planner(turn, initialize=True, backend='nlp')
        p.run(p.get_initial_guess('tri'))
    assert p.sol_phi.min() >= philo - 1e-12 and p.sol_phi.max() <= phihi + 1e-12 and (p.sol_phi > phihi - 1e-4).sum() >= 3


def test_costbank_max_mode_vs_oracle(ctx):
    """CostBank(use_mean=False), src/d2d/opty_utils.py:68-82 -- cost obj_scale * kbank * max phi^2, cost_grad one-hot at the maximiser --
    on the collocation backend (D2D_SC_BANKMAX rows): the maximiser of an iterate carries the whole term for the length of a Newton
    step, the merit function is the true max, the solve ends converged in value (the one-hot gradient has no zero where two nodes
    share the maximum).  Kernel = oracle: same cost, same largest bank angle, feasible; the largest bank angle is clearly below the
    mean-mode plan's, whose cost functional it does not minimise; and the reference's plug-point accepts the cost."""
    import d2dhip as D
    N, h = 41, 0.1
    p0, p1 = (0., 0., 0.), (45., 12., 0.6)
    pm = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=1., obj_scale=1., phi_max=np.deg2rad(30.), v_min=9., v_max=14., bank_max=True)
    pa = nlp.Problem(N, h, p0, p1, vsp=12., kv=1., kphi=1., obj_scale=1., phi_max=np.deg2rad(30.), v_min=9., v_max=14.)
    t = np.linspace(0, 1, N)
    W0 = np.stack([45 * t, 12 * t, 0.6 * t, np.zeros(N), 12 * np.ones(N)], 1)
    rm = _row(pm); rm[D.SC_BANKMAX] = 1.0
    W, out = _solve(ctx, [pm, pa], [W0, W0], [rm, _row(pa)])
    assert (out['status'] == 1).all(), out['status']
    Wm, Wa = W[:, :, 0], W[:, :, 1]
    Wo, info = nlp.solve(pm, W0)
    assert info['status'] == 1
    assert abs(out['cost'][0] - nlp.cost(pm, Wm)) <= 1e-10 * out['cost'][0]          # the cost reported is the reference's cost()
    assert abs(info['cost'] - out['cost'][0]) <= 1e-5 * info['cost'], (info['cost'], out['cost'][0])
    assert abs(np.abs(Wm[:, 3]).max() - np.abs(Wo[:, 3]).max()) <= 1e-4
    assert np.abs(nlp.constraints(pm, Wm)).max() <= 1e-8
    assert np.abs(Wm[:, 3]).max() < 0.9 * np.abs(Wa[:, 3]).max()                      # min-max flattens the bank profile
    assert (np.abs(Wm[:, 3]) > 0.999 * np.abs(Wm[:, 3]).max()).sum() >= 2              # ... at least two nodes share the maximum
    # through opty.direct_collocation.Problem: a scenario whose cost is CostBank(use_mean=False) no longer raises
    import contextlib, io
    import d2d.opty_utils as d2ou
    import d2d.optyplan_scenarios as sc
    import single_opt_planner as sop

    cb = d2ou.CostBank()
    cb.use_mean = False                                  # (a class attribute in the reference, :69)

    class bankmax(sc.exp_0):
        t1, p0, p1 = 4.0, (0., 0., 0., 0., 12.), (45., 12., 0.6, 0., 12.)
        cost = cb
    with contextlib.redirect_stdout(io.StringIO()):
        p = sop.Planner(bankmax, initialize=True, backend='nlp')
        p.run(p.get_initial_guess('tri'))
    assert np.isfinite(p.sol_phi).all() and p.info['feas'] <= 1e-8
    assert abs(p.info['obj_val'] - bankmax.obj_scale * np.max(p.sol_phi ** 2)) <= 1e-12 + 1e-9 * p.info['obj_val']


def test_persistent_handout_is_independent_of_the_batch(ctx):
    """nlp_solve_kernel is persistent (one wavefront per wave slot, problems from a device counter) and a wavefront keeps ONE workspace for
    every problem it takes: nothing of a solve may leak into the next one through it.  2560 problems in one launch -- more than the 2048
    slots of an MI355X, so the slots that finish first take a second problem -- against the same problems solved in small launches of
    their own (first, last and a middle block): status, Newton steps, cost and node values bit-identical."""
    import d2dhip
    from d2dhip import synth
    B = 2560
    rows, W0, h = synth.nlp_problems(B, seed=3)
    dsc = ctx.dev(rows)
    W = ctx.dev(np.ascontiguousarray(W0))
    big = ctx.nlp_solve(dsc, W, h)
    ctx.sync()
    st, it, co, Wb = big['status'].cpu().numpy(), big['iters'].cpu().numpy(), big['cost'].cpu().numpy(), W.cpu().numpy()
    assert (st == 1).mean() > 0.9
    for lo in (0, 1200, B - 40):
        sl = slice(lo, lo + 40)
        Ws = ctx.dev(np.ascontiguousarray(W0[sl]))
        small = ctx.nlp_solve(ctx.dev(np.ascontiguousarray(rows[sl])), Ws, h)
        ctx.sync()
        assert (small['status'].cpu().numpy() == st[sl]).all()
        assert (small['iters'].cpu().numpy() == it[sl]).all()
        np.testing.assert_array_equal(small['cost'].cpu().numpy(), co[sl])
        np.testing.assert_array_equal(Ws.cpu().numpy(), Wb[sl])


def test_handout_order_only_schedules(ctx):
    """d2d_nlp_opts.order (ABI 109): the persistent launch hands the problems out in the caller's order -- here longest first by the
    step counts of a first solve, and a reversed order -- and every status, step count, cost and node value stays bit-identical.  An
    entry that is no problem index is skipped, not dereferenced: the problems it should have named keep their initial guess."""
    import torch
    from d2dhip import synth
    B = 2304                                        # more than the 2048 wave slots of an MI355X: the order decides who shares a slot
    rows, W0, h = synth.nlp_problems(B, seed=5)
    dsc = ctx.dev(rows)
    W = ctx.dev(np.ascontiguousarray(W0))
    ref = ctx.nlp_solve(dsc, W, h)
    ctx.sync()
    it = ref['iters'].cpu().numpy()
    for perm in (np.argsort(-it, kind='stable'), np.arange(B)[::-1].copy()):
        order = torch.from_numpy(perm.astype(np.int32)).to(ctx.device)
        W2 = ctx.dev(np.ascontiguousarray(W0))
        got = ctx.nlp_solve(dsc, W2, h, order=order)
        ctx.sync()
        for k in ('status', 'iters', 'cost', 'feas'):
            assert torch.equal(got[k], ref[k]), k
        assert torch.equal(W2, W)
    bad = np.arange(64, dtype=np.int32); bad[5] = 64; bad[9] = -1
    W3 = ctx.dev(np.ascontiguousarray(W0[:64]))
    got = ctx.nlp_solve(ctx.dev(np.ascontiguousarray(rows[:64])), W3, h, order=torch.from_numpy(bad).to(ctx.device))
    ctx.sync()
    keep = np.ones(64, bool); keep[[5, 9]] = False
    assert torch.equal(W3[torch.from_numpy(keep).to(ctx.device)], W[:64][torch.from_numpy(keep).to(ctx.device)])
    np.testing.assert_array_equal(W3[[5, 9]].cpu().numpy(), W0[[5, 9]])
