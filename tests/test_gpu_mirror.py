"""GPU: the reference-named classes (d2d.dynamic, d2d.guidance, Controllers, planners,
full_sim) -- written like the reference's call sites -- against the golden vectors and the
oracle.  Every numeric call below goes through libd2dhip.so."""
import numpy as np
import pytest

from oracle import sim as S, fit as F

pytestmark = pytest.mark.gpu


def test_aircraft_disc_dyn_and_jac(gold):
    import d2d.dynamic as ddyn
    import d2d.guidance as ddg
    g = gold('plant')
    ac = ddyn.Aircraft()
    X1 = ac.disc_dyn([20, 30, -np.pi / 2, 0, 10], [0.1, 15], ddg.WindField(), 0, 0.05)   # SURVEY.md 8c known answer
    np.testing.assert_allclose(X1, g['known_answer_disc_dyn'], atol=1e-7)
    np.testing.assert_allclose(X1, [20.0008409, 29.49385394, -1.56691014, 0.09932621, 10.24385288], atol=1e-7)
    for i in range(0, len(g['X']), 5):
        ac.tau_phi = 0.9667
        Xn = ac.disc_dyn(g['X'][i], g['U'][i], ddg.WindField(list(g['W'][i])), 0.3, 0.05)
        d = Xn - g['Xnext_tau0.9667'][i]; d[2] = S.norm_mpi_pi(d[2])
        assert np.abs(d).max() < 1e-6
        ac.tau_phi = 0.01
        A, B = ac.cont_jac(g['X'][i], g['U'][i], 0.0, None)
        np.testing.assert_allclose(A, g['A'][i], rtol=1e-13, atol=1e-13); np.testing.assert_allclose(B, g['B'][i], rtol=1e-15)
        np.testing.assert_allclose(ac.cont_dyn(g['X'][i], 0.0, g['U'][i], ddg.WindField(list(g['W'][i]))), g['cont_dyn'][i], rtol=1e-14)


def test_flatness_maps(gold):
    import d2d.dynamic as ddyn
    import d2d.guidance as ddg
    import Controllers as tracking
    g = gold('flatness_ctrl')
    ac = ddyn.Aircraft()
    for i in range(0, len(g['Y']), 3):
        Ys = np.array([g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i]])
        X, U, Xd = ddg.DiffFlatness.state_and_input_from_output(Ys, g['W'][i], ac)
        np.testing.assert_allclose(X, g['g_X'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(U, g['g_U'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(Xd, g['g_Xdot'][i], rtol=1e-12, atol=1e-12)
        X, U = tracking.DiffFlatness(list(g['W'][i])).ComputeFlatness(0.0, g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i])
        np.testing.assert_allclose(X, g['c_X'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(U, g['c_U'][i], rtol=1e-11, atol=1e-12)


def test_compute_gain_like_the_reference_loop(gold):
    import d2d.dynamic as ddyn
    import Controllers as tracking
    g = gold('flatness_ctrl')
    ac = ddyn.Aircraft()
    for i in range(0, len(g['Y']), 4):
        ctrl = tracking.DiffController(list(g['W'][i]))
        Xr, dX, U = ctrl.ComputeGain(0.0, g['X'][i].copy(), g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i], ac)
        np.testing.assert_allclose(Xr, g['gain_Xr_carestandin'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(dX, g['gain_dX_carestandin'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(U, g['gain_U_carestandin'][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(ctrl.K[-1], g['gain_K_carestandin'][i], rtol=1e-8, atol=1e-9)


def test_dfff_controller_class(gold):
    """ddg.DFFFController(traj, ac, wind).get(X, t) as 05_test_simulation.py / compare.py call it."""
    import d2d.dynamic as ddyn
    import d2d.guidance as ddg
    g = gold('dfff_carestandin')

    class Traj:
        duration = 10.0

        def __init__(self, Ys): self.Ys = Ys
        def get(self, t): return self.Ys

    ac = ddyn.Aircraft()
    for i in (0, 5, 30, 47):
        ctl = ddg.DFFFController(Traj(np.array([g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i]])), ac, ddg.WindField(list(g['W'][i])))
        U = ctl.get(g['X'][i].copy(), 0.5)
        np.testing.assert_allclose(U, g['U'][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(ctl.K[-1], g['K'][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(ctl.Xref[-1], g['Xr'][i], rtol=1e-12, atol=1e-12)


def test_lqr_vs_scipy_care():
    """control.lqr replacement on random stabilisable 5x5 / 2-input systems."""
    import scipy.linalg
    import d2dhip
    ctx = d2dhip.default_context()
    rng = np.random.default_rng(4)
    n = 64
    A = rng.normal(0, 1.0, (n, 5, 5)); Bm = rng.normal(0, 1.0, (n, 5, 2))
    Qh = rng.normal(0, 1, (5, 5)); Q = Qh @ Qh.T + 0.1 * np.eye(5); R = np.array([[2.0, 0.3], [0.3, 1.0]])
    K, P = ctx.lqr(ctx.dev(np.ascontiguousarray(A.reshape(n, 25).T)), ctx.dev(np.ascontiguousarray(Bm.reshape(n, 10).T)), Q, R)
    K = K.cpu().numpy().T.reshape(n, 2, 5); P = P.cpu().numpy().T.reshape(n, 5, 5)
    for i in range(n):
        Pr = scipy.linalg.solve_continuous_are(A[i], Bm[i], Q, R)
        np.testing.assert_allclose(P[i], Pr, rtol=1e-8, atol=1e-8 * np.abs(Pr).max())
        np.testing.assert_allclose(K[i], np.linalg.solve(R, Bm[i].T @ Pr), rtol=1e-7, atol=1e-8 * np.abs(Pr).max())


def test_lqr_when_the_inverse_must_exchange_rows():
    """The small-matrix inverse exchanges rows only where a diagonal entry is below 1e-3 of its column (a wave-uniform branch,
    sim_device.h inverse): systems with A[0][0] equal to the Cayley shift (A - gamma I starts with a ZERO pivot) on every third
    lane of the wavefront, ordinary ones on the others -- both kinds must come out right in the same launch."""
    import scipy.linalg
    import d2dhip
    ctx = d2dhip.default_context()
    rng = np.random.default_rng(11)
    n = 96
    A = rng.normal(0, 1.0, (n, 5, 5)); Bm = rng.normal(0, 1.0, (n, 5, 2))
    A[::3, 0, 0] = 3.0                      # D2D_CARE_GAMMA
    A[1::6, 1, 1] = 3.0; A[1::6, 0, 1] = 0.0; A[1::6, 1, 0] = 0.0      # a zero pivot in the SECOND column after the first elimination
    Q = np.diag([1.0, 2.0, 0.5, 0.1, 0.3]); R = np.array([[2.0, 0.3], [0.3, 1.0]])
    K, P = ctx.lqr(ctx.dev(np.ascontiguousarray(A.reshape(n, 25).T)), ctx.dev(np.ascontiguousarray(Bm.reshape(n, 10).T)), Q, R)
    K = K.cpu().numpy().T.reshape(n, 2, 5); P = P.cpu().numpy().T.reshape(n, 5, 5)
    assert np.isfinite(P).all()
    for i in range(n):
        Pr = scipy.linalg.solve_continuous_are(A[i], Bm[i], Q, R)
        np.testing.assert_allclose(P[i], Pr, rtol=1e-8, atol=1e-8 * np.abs(Pr).max())
        np.testing.assert_allclose(K[i], np.linalg.solve(R, Bm[i].T @ Pr), rtol=1e-7, atol=1e-8 * np.abs(Pr).max())


def test_dcf_circle_gvf_helpers(gold):
    import d2d.guidance as ddg
    g = gold('guidance')
    dcf = ddg.DCFController()
    for i in range(0, len(g['dcf_c']), 3):
        zd = g['dcf_zdes'][i].copy()
        Ur, eth = dcf.get(4, g['B'], g['dcf_c'][i], g['dcf_p'][i], zd, float(g['dcf_kr']))
        assert zd.shape == (3, 1) and Ur.shape == (4, 1) and eth.shape == (3, 1)        # the in-place reshape quirk
        np.testing.assert_allclose(Ur[:, 0], g['dcf_Ur'][i], rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(eth[:, 0], g['dcf_etheta_deg'][i], rtol=1e-12, atol=1e-11)
    for i in range(0, len(g['gvf_X']), 3):
        e, n, H = ddg.CircleTraj(g['gvf_c'][i]).get(g['gvf_X'][i], g['gvf_r'][i])
        np.testing.assert_allclose(e, g['gvf_e'][i], rtol=1e-15); np.testing.assert_allclose(n, g['gvf_n'][i], rtol=1e-15)
        U, U1, U2 = ddg.GVFcontroller(None, None, None).get(g['gvf_X'][i], float(g['gvf_ke']), float(g['gvf_kd']), e, n, H)
        np.testing.assert_allclose([U, U1, U2], [g['gvf_U'][i], g['gvf_U1'][i], g['gvf_U2'][i]], rtol=1e-11, atol=1e-11)


def test_circular_formation_like_11_full_sim():
    """The call of src/11_full_sim_case1.py:436 with its own parameters, shortened time grid."""
    import full_sim
    c = np.array([[0, -20], [25, -20], [25, -100], [0, -100.0]])
    X1_f = ((0, 40, 0, 0, 12), (25, 40, 0, 0, 12), (25, -40, 0, 0, 12), (0, -40, 0, 0, 12))
    X, U, U1, U2, Ur, eth, time, t_f = full_sim.CircularFormationGVF(c, 60, 15, 4, X1_f, 0, 0.05, 30)
    assert X.shape == (600, 4, 5) and U.shape == (600, 4, 2) and Ur.shape == (600, 4) and eth.shape == (600, 3)
    Xo, Uo, Rro, etho, stop = S.formation_gvf_run(c, 60.0, 15.0, np.tile(full_sim.X1_START, (4, 1)), 600, 0.05, X0f=X1_f)
    assert stop == 600 and t_f == 30
    d = X - Xo; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 1e-7
    np.testing.assert_allclose(U[:599], Uo[:599], atol=1e-8)
    np.testing.assert_array_equal(full_sim.ConstructBMatrix(4), S.construct_b_matrix(4))


def test_implement_controller_like_11_full_sim(gold):
    import full_sim
    g = gold('tracking_trace_carestandin')
    X, U, Xr, Yd, Ydd, dX = full_sim.implement_controller(4, g['time'], g['x_ref'], g['y_ref'], 15, [0, 0], g['X'][0])
    T = len(g['time'])
    d = X - g['X']; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 2e-4
    np.testing.assert_allclose(Xr[:T - 1], g['Xr'][:T - 1], rtol=1e-10, atol=1e-10)
    Fdx, Fdy, Fddx, Fddy = full_sim.ComputeDerivatives(g['x_ref'][:, 0], g['y_ref'][:, 0], g['time'][1] - g['time'][0])
    np.testing.assert_allclose(Yd[1:T - 1, 0, 0], Fdx[2:T], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Ydd[1:T - 1, 0, 1], Fddy[2:T], rtol=1e-10, atol=1e-10)


def _kinematic_residual(sol_x, sol_y, sol_psi, sol_phi, sol_v, coefs, duration, wind=(0, 0)):
    """The fitted trajectory satisfies the planner's kinematic model identically: compare the
    sampled states with derivatives of the polynomial itself."""
    import d2d.trajectory as ddt
    traj = ddt.CompositeTraj.from_fit(coefs, duration)
    K = len(sol_x)
    t = np.linspace(0, duration, K)
    worst = 0.0
    for k in range(K - 1):
        Y = traj.get(t[k])
        worst = max(worst, abs(Y[0, 0] - sol_x[k]), abs(Y[0, 1] - sol_y[k]),
                    abs(Y[1, 0] - (sol_v[k] * np.cos(sol_psi[k]) - wind[0])), abs(Y[1, 1] - (sol_v[k] * np.sin(sol_psi[k]) - wind[1])))
        psid = (Y[2, 1] * (Y[1, 0] + wind[0]) - Y[2, 0] * (Y[1, 1] + wind[1])) / sol_v[k] ** 2
        worst = max(worst, abs(psid - 9.81 / sol_v[k] * np.tan(sol_phi[k])))
    return worst


def test_single_planner_config0_plumbing(gold, tmp_path):
    """BASELINE configs[0]: aircraft 1 of opt_states_st_line.csv, (0,40,0) -> (75,40,0), t1 = 7 s at
    10 Hz (71 nodes) through the single_opt_planner.Planner protocol with CostAirVel(12)."""
    import d2d.opty_utils as d2ou
    import d2d.optyplan_scenarios as d2oscen
    import single_opt_planner as sop

    class scen(d2oscen.exp_14):
        name = 'st_line_ac1'
        t0, p0 = 0, (0, 40, 0, 0, 12)
        t1, p1 = 7.0, (75, 40, 0, 0, 12)
        cost = d2ou.CostAirVel(12.)
    p = sop.Planner(scen, initialize=True, backend='fit')
    assert p.num_nodes == 71
    fn = str(tmp_path / 'optyplan_st_line.npz')
    sop.compute_or_load(p, force_recompute=True, filename=fn, tol=1e-5, max_iter=500)
    assert p.info['status_msg'] in ('converged', 'stalled')
    # end conditions (src/single_opt_planner.py:46-49)
    np.testing.assert_allclose([p.sol_x[0], p.sol_y[0], p.sol_psi[0]], [0, 40, 0], atol=1e-9)
    np.testing.assert_allclose([p.sol_x[-1], p.sol_y[-1], p.sol_psi[-1]], [75, 40, 0], atol=1e-8)
    assert _kinematic_residual(p.sol_x, p.sol_y, p.sol_psi, p.sol_phi, p.sol_v, p.fit_coefs, p.duration) < 1e-8
    # the reference's objective on the solution: 75 m in 7 s needs a dog-leg at 12 m/s; bounds respected softly
    c_ref = scen.cost.cost(p.solution, p)
    assert c_ref < 0.5 and np.abs(p.sol_phi).max() < np.deg2rad(40) + 0.05 and p.sol_v.min() > 8.5
    # golden aircraft-1 solution of the reference's IPOPT run has cost ~0 on CostAirVel; ours is a smooth C^3 polynomial
    g = gold('planner_goldens')
    N = 71
    v_gold = g['stline_free'][4 * 4 * N:4 * 4 * N + N]
    assert abs(np.mean(p.sol_v) - np.mean(v_gold)) < 0.5
    # npz cache round trip in the reference's key layout
    d = np.load(fn)
    assert sorted(d.files) == ['sol_phi', 'sol_psi', 'sol_time', 'sol_v', 'sol_x', 'sol_y', 'wind']
    p2 = sop.Planner(scen, initialize=True, backend='fit')
    sop.compute_or_load(p2, force_recompute=False, filename=fn)
    np.testing.assert_array_equal(p2.sol_x, p.sol_x)
    # oracle parity of the same fit
    import d2dhip
    ob = F.FitBasis.from_arrays(6, 71, p.duration, *p.fit_plan.basis())
    row = p.fit_scen.cpu().numpy()[0]
    xy = p.get_initial_guess('tri')
    q0 = np.concatenate([ob.Pinit @ (xy[p._slice_x] - ob.Gp[0] @ F.end_data(row)[0]), ob.Pinit @ (xy[p._slice_y] - ob.Gp[0] @ F.end_data(row)[1])])
    qo, co, _, sto = F.lm_solve(ob, row, q0=q0)
    assert abs(p.info['obj_val'] - co) <= 1e-6 * co


def test_single_planner_exp14_and_wind():
    import d2d.opty_utils as d2ou
    import d2d.optyplan_scenarios as d2oscen
    import single_opt_planner as sop
    p = sop.Planner(d2oscen.exp_14, initialize=True, backend='fit')
    p.configure(1e-5, 500)
    p.run(p.get_initial_guess('tri'))
    assert p.solution.shape == (5 * 121,)
    assert p.fit_plan.kernel == 'long'            # 121 nodes: the persistent long-horizon kernel (second-order mode included)
    np.testing.assert_allclose([p.sol_x[0], p.sol_y[0], p.sol_psi[0]], d2oscen.exp_14.p0[:3], atol=1e-8)
    np.testing.assert_allclose([p.sol_x[-1], p.sol_y[-1]], d2oscen.exp_14.p1[:2], atol=1e-8)
    assert _kinematic_residual(p.sol_x, p.sol_y, p.sol_psi, p.sol_phi, p.sol_v, p.fit_coefs, p.duration) < 1e-8
    # the reference's IPOPT optimum of this scenario has CostAirVel = 5.0297; the smooth
    # polynomial fit with soft bounds lands in the same regime
    assert d2oscen.exp_14.cost.cost(p.solution, p) < 15.0

    class windy(d2oscen.exp_0):
        wind = d2ou.WindField(w=[2., 0.])
        t1 = 10.
    pw = sop.Planner(windy, initialize=True, backend='fit')
    pw.run()
    # planner sign convention: xdot = v cos(psi) - wx  (src/d2d/opty_utils.py:42)
    assert _kinematic_residual(pw.sol_x, pw.sol_y, pw.sol_psi, pw.sol_phi, pw.sol_v, pw.fit_coefs, pw.duration, wind=(2., 0.)) < 1e-8


def test_planner_cost_variants_and_boxes():
    """CostComposit with kind-0 obstacles, CostBank max mode, and the x/y boxes (soft bound rows)."""
    import d2d.opty_utils as d2ou
    import d2d.optyplan_scenarios as d2oscen
    import single_opt_planner as sop

    class kind0(d2oscen.exp_14):
        obstacles = [(30., 2., 4.)]
        cost = d2ou.CostComposit([(30., 2., 4.)], vsp=12., kobs=1., kvel=5., kbank=1., obs_kind=0)
    p = sop.Planner(kind0, initialize=True, backend='fit')
    p.run()
    d = np.hypot(p.sol_x - 30., p.sol_y - 2.)
    assert np.isfinite(p.solution).all() and p.info['obj_val'] > 0
    # the reported objective is the mirrored reference cost (+ waypoint / bound regularisers >= 0)
    assert kind0.cost.cost(p.solution, p) <= p.info['obj_val'] * (1 + 1e-9) + 1e-12, (kind0.cost.cost(p.solution, p), p.info, d.min())

    class bankmax(d2oscen.exp_14):
        cost = d2ou.CostBank()
    bankmax.cost.use_mean = False
    pb = sop.Planner(bankmax, initialize=True, backend='fit')
    pb.run()
    assert bankmax.cost.cost(pb.solution, pb) <= pb.info['obj_val'] * (1 + 1e-9) + 1e-12

    # x/y boxes are soft bound rows of the fit: a binding one bends the plan, a roomy one changes nothing
    # (exp_0, the U-turn, swings 50 m out in +x; the box allows half of that)
    free = sop.Planner(d2oscen.exp_0, initialize=True, backend='fit'); free.run()
    assert free.info['box_violation'] == 0.0 and np.max(free.sol_x) > 30.0
    lim = float(np.max(free.sol_x)) / 2

    class boxed(d2oscen.exp_0):
        x_constraint = (-1e3, lim)
    pbx = sop.Planner(boxed, initialize=True, backend='fit'); pbx.run()
    assert lim - 1.0 < np.max(pbx.sol_x) < lim + 0.6 and 0.0 < pbx.info['box_violation'] < 0.6, (np.max(pbx.sol_x), lim, pbx.info)
    assert pbx.info['obj_val'] > free.info['obj_val']             # the constrained optimum costs more

    class roomy(d2oscen.exp_0):
        x_constraint, y_constraint = (-500., 500.), (-500., 500.)
    pr = sop.Planner(roomy, initialize=True, backend='fit'); pr.run()
    np.testing.assert_allclose(pr.solution, free.solution, rtol=0, atol=1e-9)


def test_planner_three_and_twelve_obstacles():
    """The reference's three-disc 'maze' (exp_4_2, flown in 9 s: its 15 s make the box and the bank limit conflict) and
    the twelve-disc checkerboard (exp_5) through the single planner: the obstacle rows beyond the first two and the
    y box of the maze are lowered into the scenario row's extension columns; the fixed point the kernel reaches is a
    stationary point of the oracle's objective for that row (every row is in), and the maze plan leans on its box."""
    import d2d.optyplan_scenarios as d2oscen
    import single_opt_planner as sop
    from oracle import fit as F

    class maze9(d2oscen.exp_4_2):
        t1 = 9.
    for scen, n_obs in ((maze9, 3), (d2oscen.exp_5, 12)):
        assert len(scen.obstacles) == n_obs
        for pl in sop._plans.values():             # plans are cached per (K, duration) with the whitening weights of their
            pl.close()                             # first user: rebuild, so that the oracle basis below is this plan's
        sop._plans.clear()
        p = sop.Planner(scen, initialize=True, backend='fit')
        p.run(initial_guess=p.get_initial_guess('tri'))
        assert np.isfinite(p.solution).all() and p.info['obj_val'] > 0 and p.info['status_msg'] in ('converged', 'stalled'), p.info
        assert scen.cost.cost(p.solution, p) <= p.info['obj_val'] * (1 + 1e-9) + 1e-12
        full = scen.cost.cobs.cost(p.solution, p)
        assert abs(sum(o.cost(p.solution, p) for o in scen.cost.cobs.obss) - full) <= 1e-12 * max(full, 1e-300)
        row = p.fit_scen.cpu().numpy()[0]
        q = p.fit_q.cpu().numpy()[0]
        assert F.n_extra_obs(row) == n_obs - 2 and F.has_box(row) == (scen.y_constraint is not None)
        low = sop.lower_cost(scen.cost)
        s = scen.obj_scale / p.num_nodes
        b = F.FitBasis(sop.N_SEG, p.num_nodes, p.duration, (sop.W_WAYPOINT ** 2, s * max(low[1], 1e-3), s * max(low[2], 1e-3) / 9.81 ** 2))
        c, g, H = F.eval_normal(b, row, q)
        assert abs(c - p.info['obj_val']) <= 1e-9 * c, (c, p.info)
        assert np.abs(g).max() <= 1e-6 * max(1.0, c), (scen.name, np.abs(g).max(), c)
        if scen is maze9:
            assert p.info['box_violation'] < 0.05 and np.min(p.sol_y) < scen.y_constraint[0] + 0.5, (p.info, np.min(p.sol_y))


def test_multi_planner_like_11_full_sim():
    import multi_opt_planner as mop
    import d2d.multiopty_utils as d2mou
    scen = mop.trap_4
    scen.t1 = 6
    scen.p0s = ((0, 40, 0, 0, 12), (25, 40, 0, 0, 12), (25, -40, 0, 0, 12), (0, -40, 0, 0, 12))
    scen.p1s = ((75, 40, 0, 0, 12), (100, 40, 0, 0, 12), (100, -40, 0, 0, 12), (75, -40, 0, 0, 12))
    _p = mop.Planner(scen, initialize=True)                       # kcol = 10, rcol = 10: coupled pair (0, 1)
    _p.run(initial_guess=_p.get_initial_guess(scen.initial_guess), tol=scen.tol, max_iter=scen.max_iter)
    _p.interpret_solution()
    c_coupled = scen.cost.cost(_p.solution, _p)
    sep = np.hypot(_p.sol_x[0] - _p.sol_x[1], _p.sol_y[0] - _p.sol_y[1])
    scen_cost = scen.cost
    scen.cost = d2mou.CostComposit(kvel=70., kbank=1., kobs=float('NaN'), kcol=float('NaN'), vsp=12., obss=[], obs_kind=0, rcol=10)
    _p = mop.Planner(scen, initialize=True)
    _p.run(initial_guess=_p.get_initial_guess(scen.initial_guess), tol=scen.tol, max_iter=scen.max_iter)
    _p.interpret_solution()
    assert len(_p.sol_x) == 4 and _p.sol_x[0].shape == (61,)
    for i in range(4):
        np.testing.assert_allclose([_p.sol_x[i][0], _p.sol_y[i][0], _p.sol_x[i][-1], _p.sol_y[i][-1]],
                                   [scen.p0s[i][0], scen.p0s[i][1], scen.p1s[i][0], scen.p1s[i][1]], atol=1e-8)
        assert _kinematic_residual(_p.sol_x[i], _p.sol_y[i], _p.sol_psi[i], _p.sol_phi[i], _p.sol_v[i], _p.fit_coefs[i], _p.duration) < 1e-8
    # 75 m in 6 s at vsp = 12: straight line, ~12.5 m/s -- the symmetric pairs give mirrored answers
    np.testing.assert_allclose(_p.sol_y[0] - 40, -(_p.sol_y[3] + 40), atol=1e-6)
    # 75 m in 6 s needs 12.5 m/s on average: 70 * mean((v-12)^2) ~ 70 * 0.25
    c = scen.cost.cost(_p.solution, _p)
    assert 10.0 < c < 20.0 and abs(np.mean(_p.sol_v[0]) - 12.5) < 0.2, c
    # with the collision term the coupled pair keeps at least its uncoupled separation and the reference's
    # own cost (which includes CostCollision on pair (0,1)) is not worse than for the uncoupled plan
    sep0 = np.hypot(_p.sol_x[0] - _p.sol_x[1], _p.sol_y[0] - _p.sol_y[1])
    assert sep.min() >= sep0.min() - 1e-6
    assert c_coupled <= scen_cost.cost(_p.solution, _p) + 1e-9
    # ... and feeds the tracking phase exactly as src/11_full_sim_case1.py:455-460 does
    import full_sim
    x_ref = np.array(_p.sol_x).T; y_ref = np.array(_p.sol_y).T
    X, U, Xr, Yd, Ydd, dX = full_sim.implement_controller(4, np.array(_p.sol_time), x_ref, y_ref, 15, [0, 0], scen.p0s)
    assert X.shape == (61, 4, 5) and np.abs(X[-1, :, 0] - x_ref[-1]).max() < 3.0


def test_three_phase_chain_on_device():
    """full_sim.full_sim_phases_batch chains the phases of src/11_full_sim_case1.py main() on the device for many
    formations: formation 0 against the same phases driven one by one through the reference-named API (host hand-offs),
    an identical formation bit for bit, and a translated one (the whole problem is translation-equivariant)."""
    import full_sim as fs
    import multi_opt_planner as mop
    n_ac, r, v, w = 4, 60, 15, [0, 0]
    c = np.array([[0, -20], [25, -20], [25, -100], [0, -100]], float)
    X1_f = np.array(((0, 40, 0, 0, 12), (25, 40, 0, 0, 12), (25, -40, 0, 0, 12), (0, -40, 0, 0, 12)), float)
    X2_f = np.array(((75, 40, 0, 0, 12), (100, 40, 0, 0, 12), (100, -40, 0, 0, 12), (75, -40, 0, 0, 12)), float)
    t_opt = 6
    # one by one, as main() does it (:433-457)
    X1, U1, _, _, _, _, time1, t1_f = fs.CircularFormationGVF(c, r, v, n_ac, X1_f, 0, 0.05, 1000)
    X2_i = tuple(map(tuple, X1[-1]))
    scen = mop.trap_4
    scen.t1, scen.p0s, scen.p1s = t_opt, X2_i, tuple(map(tuple, X2_f))
    _p = mop.Planner(scen, initialize=True)
    _p.run(initial_guess=_p.get_initial_guess(scen.initial_guess), tol=scen.tol, max_iter=scen.max_iter)
    _p.interpret_solution()
    x_ref_2, y_ref_2 = np.array(_p.sol_x).T, np.array(_p.sol_y).T
    Xa2, Ua2, *_ = fs.implement_controller(n_ac, np.array(_p.sol_time), x_ref_2, y_ref_2, v, w, X2_i)
    # phase 3 reference: a closed loop flown twice (stands for the csv of :430)
    T3 = 120
    th = np.linspace(0, 2 * np.pi, T3)
    time_3 = np.arange(T3) * 0.1
    x3 = X2_f[None, :, 0] + 30 * np.sin(th)[:, None]; y3 = X2_f[None, :, 1] + 30 * (1 - np.cos(th))[:, None]
    Xa3, *_ = fs.implement_controller(n_ac, time_3, x3, y3, v, w, tuple(map(tuple, Xa2[-1])))
    # the chain, three formations
    off = np.array([40., 50.])                            # (stays inside the x/y boxes of trap_4, which are rows of the fit)
    cB = np.stack([c, c, c + off])
    X1B = np.stack([X1_f, X1_f, X1_f + np.r_[off, 0, 0, 0]]); X2B = np.stack([X2_f, X2_f, X2_f + np.r_[off, 0, 0, 0]])
    t_end = t1_f + t_opt + 2 * time_3[-1] - 1e-6          # room for exactly two passes of phase 3
    X0B = np.tile(fs.X1_START, (3, n_ac, 1)); X0B[2, :, :2] += off        # (every aircraft starts at the reference's X1, :113)
    out = fs.full_sim_phases_batch(cB, r, v, n_ac, X1B, mop.trap_4, X2B, t_opt, ref3=(time_3, x3, y3), t_sim_end=t_end,
                                   X0=X0B, record2=('X', 'U'), record3=('X',))
    fs.d2dhip.default_context().sync()
    N = 3 * n_ac
    Xf1 = out['phase1']['X_final'].cpu().numpy().T.reshape(3, n_ac, 5)
    np.testing.assert_array_equal(Xf1[0], X1[-1])
    np.testing.assert_array_equal(Xf1[1], Xf1[0])
    Xs = out['plan']['Xs'].cpu().numpy().reshape(3, n_ac, 5, -1)
    np.testing.assert_allclose(Xs[0, :, 0].T, x_ref_2, rtol=0, atol=1e-6)
    np.testing.assert_allclose(Xs[0, :, 1].T, y_ref_2, rtol=0, atol=1e-6)
    X2b = out['phase2']['X'].cpu().numpy().transpose(0, 2, 1).reshape(-1, 3, n_ac, 5)
    np.testing.assert_allclose(X2b[:, 0], Xa2, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(X2b[:, 1], X2b[:, 0])
    shift = np.r_[off, 0, 0, 0]
    np.testing.assert_allclose(X2b[:, 2] - shift, X2b[:, 0], rtol=0, atol=1e-5)
    assert len(out['phase3']) == 2
    X3b = out['phase3'][0]['X'].cpu().numpy().transpose(0, 2, 1).reshape(-1, 3, n_ac, 5)
    np.testing.assert_allclose(X3b[:, 0], Xa3, rtol=0, atol=1e-4)
    # x3 / y3 are shared by the formations: the translated one chases the untranslated reference, so only 0 == 1 here
    np.testing.assert_array_equal(X3b[:, 1], X3b[:, 0])
    assert np.isfinite(out['phase3'][1]['X_final'].cpu().numpy()).all()


def test_three_phase_chain_vs_the_oracle():
    """The chain of src/11_full_sim_case1.py main() (:405-499) for one formation, ORACLE end to end -- oracle/sim.py formation_gvf_run
    until the stop rule -> the oracle's solve of the trap_4 scenario from where the oracle's phase 1 ended -> oracle track_run on the
    oracle's plan -> two passes of phase 3 -- against full_sim.full_sim_phases_batch, which hands every phase's result to the next ON
    the device.  Tolerances are those of the per-phase tests or tighter (measured on an MI355X: phase 1's end state 1e-13, the plan's nodes 8e-8 m, phase 2 1e-7, the
    two passes of phase 3 5e-8 / 4e-13: the tracking loop contracts what the plan hands over)."""
    import full_sim as fs
    import multi_opt_planner as mop
    import d2d.opty_utils as d2ou
    n_ac, r, v, w = 4, 60, 15, [0, 0]
    c = np.array([[0, -20], [25, -20], [25, -100], [0, -100]], float)
    X1_f = np.array(((0, 40, 0, 0, 12), (25, 40, 0, 0, 12), (25, -40, 0, 0, 12), (0, -40, 0, 0, 12)), float)
    X2_f = np.array(((75, 40, 0, 0, 12), (100, 40, 0, 0, 12), (100, -40, 0, 0, 12), (75, -40, 0, 0, 12)), float)
    t_opt = 6
    T3 = 120
    th = np.linspace(0, 2 * np.pi, T3)
    time_3 = np.arange(T3) * 0.1
    x3 = X2_f[None, :, 0] + 30 * np.sin(th)[:, None]; y3 = X2_f[None, :, 1] + 30 * (1 - np.cos(th))[:, None]
    X0 = np.tile(fs.X1_START, (n_ac, 1))
    # ---- the oracle's chain --------------------------------------------------------------------------------------------------
    Xo1, Uo1, _, _, stop = S.formation_gvf_run(c, r, v, X0, 20000, 0.05, X0f=X1_f)
    assert 2 < stop < 20000
    Xo1_end = Xo1[stop - 1]
    scen = mop.trap_4
    scen.t1 = t_opt
    N2, dt2, dur2 = d2ou.planner_timing(scen.t0, scen.t1, scen.hz)
    rows, plan, coupled = mop.scenario_rows(scen, [tuple(x) for x in Xo1_end], X2_f[:, :3], N2, dur2, scen.obj_scale, scen.wind.w)
    ob = F.FitBasis.from_arrays(plan.S, N2, dur2, *plan.basis())
    if coupled:                     # (a cost with CostCollision: block Gauss-Seidel over the pair, as d2d_fit_solve_groups)
        qo, co, swo = F.bgs_solve(ob, rows, sweeps=250, inner_iters=8, tol=1e-12, ls=True)
    else:                           # trap_4 as the reference ships it: four independent fits, the library's default solver
        from oracle import fit_knot as FK
        kbo = FK.KnotBasis(ob) if plan.kernel == 'knot' else None
        qo = np.array([(FK.solve_minpack_knot(kbo, rows[i], hess_dtype=np.float32, chol_dtype=np.float32) if kbo is not None else
                        F.solve_minpack(ob, rows[i], hess_dtype=np.float32, chol_dtype=np.float32))[0] for i in range(n_ac)])
    Yo = np.array([F.flat_outputs(ob, rows[i], qo[i]) for i in range(n_ac)])          # (n_ac, 3, 2, K)
    x_ref_o, y_ref_o = Yo[:, 0, 0].T.copy(), Yo[:, 0, 1].T.copy()
    time_2 = np.arange(N2) * dt2
    Xo2, Uo2, *_ = S.track_run(time_2, x_ref_o, y_ref_o, Xo1_end)
    Xo3a, Uo3a, *_ = S.track_run(time_3, x3, y3, Xo2[-1])
    Xo3b, *_ = S.track_run(time_3, x3, y3, Xo3a[-1])
    # ---- the device's chain ----------------------------------------------------------------------------------------------------
    t_end = (stop - 1) * 0.05 + t_opt + 2 * time_3[-1] - 1e-6          # room for exactly two passes of phase 3
    out = fs.full_sim_phases_batch(c[None], r, v, n_ac, X1_f, mop.trap_4, X2_f, t_opt, ref3=(time_3, x3, y3), t_sim_end=t_end,
                                   X0=X0[None], record2=('X', 'U'), record3=('X', 'U'))
    fs.d2dhip.default_context().sync()
    assert int(out['phase1']['stop_row'].cpu().numpy()[0]) == stop
    Xf1 = out['phase1']['X_final'].cpu().numpy().T
    np.testing.assert_allclose(Xf1, Xo1_end, rtol=0, atol=1e-9)                      # (measured 1e-13; tests/test_gpu_sim.py: the closed GVF loop vs the oracle)
    Xs = out['plan']['Xs'].cpu().numpy()                                              # [n_ac][5][K]
    assert np.abs(Xs[:, 0].T - x_ref_o).max() <= 1e-5 and np.abs(Xs[:, 1].T - y_ref_o).max() <= 1e-5       # (measured 8e-8 m)
    zg = plan.coeffs(out['plan']['scen'], out['plan']['q']).cpu().numpy().reshape(n_ac, -1)
    zo = np.array([F.coefficients(ob, rows[i], qo[i]).reshape(-1) for i in range(n_ac)])
    assert np.abs(zg - zo).max() <= 1e-6 * np.abs(zo).max()                          # (the plan vs the oracle: the north-star's 1e-6)
    X2 = out['phase2']['X'].cpu().numpy().transpose(0, 2, 1); U2 = out['phase2']['U'].cpu().numpy().transpose(0, 2, 1)
    np.testing.assert_allclose(X2, Xo2, rtol=0, atol=1e-5)                             # (measured 1e-7)
    np.testing.assert_allclose(U2[:-1], Uo2[:-1], rtol=0, atol=1e-5)
    assert len(out['phase3']) == 2
    X3a = out['phase3'][0]['X'].cpu().numpy().transpose(0, 2, 1); X3b = out['phase3'][1]['X'].cpu().numpy().transpose(0, 2, 1)
    np.testing.assert_allclose(X3a, Xo3a, rtol=0, atol=1e-5)                           # (measured 5e-8)
    np.testing.assert_allclose(X3b, Xo3b, rtol=0, atol=1e-5)
    print('chain vs oracle: phase 1 end %.1e, plan %.1e m, phase 2 %.1e, phase 3 %.1e / %.1e' % (
        np.abs(Xf1 - Xo1_end).max(), max(np.abs(Xs[:, 0].T - x_ref_o).max(), np.abs(Xs[:, 1].T - y_ref_o).max()),
        np.abs(X2 - Xo2).max(), np.abs(X3a - Xo3a).max(), np.abs(X3b - Xo3b).max()))


def test_run_simulation_like_05_test_simulation():
    """full_sim.run_simulation (the legacy DFFF loop of src/05_test_simulation.py:21-34, time loop on the GPU) against the
    same loop driven call by call through the mirror's DFFFController.get and Aircraft.disc_dyn."""
    import full_sim as fs
    import d2d.dynamic as ddyn
    import d2d.guidance as ddg
    import d2d.trajectory as ddt
    J = [np.array([[0., 0.], [12., 0.], [0., 0.], [0., 0.]]).T,
         np.array([[40., 10.], [12., 1.], [0.2, -0.1], [0., 0.]]).T,
         np.array([[80., -5.], [12., -1.], [0., 0.3], [0., 0.]]).T]
    traj = ddt.CompositeTraj([ddt.MinSnapPoly(J[j], J[j + 1], 3.5) for j in range(2)])
    ac, wind = ddyn.Aircraft(), ddg.WindField([0.4, 0.2])
    time = np.arange(0, 7.0 - 1e-9, 0.05)
    perts = np.zeros((len(time), 5)); perts[30] = [0.5, -0.4, 0.03, 0.0, 0.2]
    X0 = np.array([0.5, -1.0, 0.05, 0.0, 11.5])
    X, U, Yref = fs.run_simulation(time, ac, wind, ddg.DFFFController(traj, ac, wind), X0, perts)
    ctl = ddg.DFFFController(traj, ac, wind)
    Xh = np.zeros_like(X); Uh = np.zeros_like(U); Xh[0] = X0
    for i in range(1, len(time)):
        Uh[i - 1] = ctl.get(Xh[i - 1].copy(), time[i - 1])
        Xh[i] = ac.disc_dyn(Xh[i - 1], Uh[i - 1], wind, time[i - 1], time[i] - time[i - 1]) + perts[i]
    Uh[-1] = ctl.get(Xh[-1].copy(), time[-1])
    assert Yref.shape == (len(time), 4, 2)
    np.testing.assert_allclose(X, Xh, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U, Uh, rtol=0, atol=1e-9)
    assert np.hypot(X[-1, 0] - Yref[-1, 0, 0], X[-1, 1] - Yref[-1, 0, 1]) < 2.0          # it tracks


def test_multi_planner_exp0_501_nodes_and_exp5_211_nodes():
    """multi_opt_planner's own scenarios at the reference's node counts (src/multi_opt_planner.py:170-228): exp_0 = one aircraft,
    10 s at 50 Hz = 501 nodes (long-horizon kernel: the basis does not fit the LDS); exp_5 = two aircraft face to face, 4.2 s at
    the inherited 50 Hz = 211 nodes, case 0 without and case 1 with the collision term (coupled-group kernels)."""
    import multi_opt_planner as mop
    p = mop.Planner(mop.exp_0, initialize=True)
    assert p.num_nodes == 501
    p.run(initial_guess=p.get_initial_guess(mop.exp_0.initial_guess), tol=mop.exp_0.tol, max_iter=mop.exp_0.max_iter)
    p.interpret_solution()
    assert p.fit_plan.kernel == 'long' and p.sol_x[0].shape == (501,)
    np.testing.assert_allclose([p.sol_x[0][0], p.sol_y[0][0], p.sol_x[0][-1], p.sol_y[0][-1]], [0., 0., 0., 50.], atol=1e-8)
    assert _kinematic_residual(p.sol_x[0], p.sol_y[0], p.sol_psi[0], p.sol_phi[0], p.sol_v[0], p.fit_coefs[0], p.duration) < 1e-7
    # 120 m of flight at vsp = 12 between points 50 m apart inside a 55 m box: the bounds bind hard, and with SOFT bound rows the
    # fit ends in a compromise.  The planner says so instead of hiding it: the reported overshoots are the observed ones.
    assert p.info['status'][0] in (1, 4), p.info
    assert abs(p.info['phi_violation'] - max(np.abs(p.sol_phi[0]).max() - np.deg2rad(40.), 0.0)) < 1e-12
    assert abs(p.info['v_violation'] - max(p.sol_v[0].max() - 15., 9. - p.sol_v[0].min(), 0.0)) < 1e-12
    assert p.info['box_violation'] < 5.0, p.info
    seps = []
    for case in (0, 1):
        mop.exp_5.set_case(case)
        p5 = mop.Planner(mop.exp_5, initialize=True)
        assert p5.num_nodes == 211
        p5.run(initial_guess=p5.get_initial_guess(mop.exp_5.initial_guess), tol=mop.exp_5.tol, max_iter=mop.exp_5.max_iter)
        p5.interpret_solution()
        for i in range(2):
            assert _kinematic_residual(p5.sol_x[i], p5.sol_y[i], p5.sol_psi[i], p5.sol_phi[i], p5.sol_v[i], p5.fit_coefs[i], p5.duration) < 1e-7
        seps.append(np.hypot(p5.sol_x[0] - p5.sol_x[1], p5.sol_y[0] - p5.sol_y[1]).min())
        assert np.isfinite(p5.solution).all()
    assert seps[1] >= seps[0] - 1e-6, seps           # the collision term never brings the pair closer


def test_default_backend_never_returns_a_plan_beyond_its_bounds():
    """single_opt_planner.BACKEND = 'auto' (what a script that only swaps sys.path gets): the polynomial fit answers when its plan
    stays inside the scenario's bounds; when the fit -- whose bounds are soft rows -- overshoots one, the collocation backend
    re-plans from it and the plan returned holds every bound exactly, like the reference's IPOPT plans (round-2 verdict, weak 6)."""
    import d2d.optyplan_scenarios as d2oscen
    import single_opt_planner as sop
    assert sop.BACKEND == 'auto'
    easy = sop.Planner(d2oscen.exp_14, initialize=True); easy.run()
    hard_fit = sop.Planner(d2oscen.exp_0, initialize=True, backend='fit'); hard_fit.run()
    hard = sop.Planner(d2oscen.exp_0, initialize=True); hard.run()
    # exp_14: the fit's plan is inside phi +-40 deg, v in [9, 15] -> answered by the fit
    if easy.info['backend_used'] == 'fit':
        assert max(easy.info['phi_violation'], easy.info['v_violation'], easy.info['box_violation']) <= sop.AUTO_TOL
    # exp_0 (a turn-around in 10 s): the soft-bound fit overshoots, the default re-plans with hard bounds
    assert max(hard_fit.info['phi_violation'], hard_fit.info['v_violation']) > sop.AUTO_TOL
    assert hard.info['backend_used'] == 'nlp' and hard.info['status'] == 1
    sc = d2oscen.exp_0
    assert np.abs(hard.sol_phi).max() <= sc.phi_constraint[1] and hard.sol_v.min() >= sc.v_constraint[0] and hard.sol_v.max() <= sc.v_constraint[1]
    assert hard.info['fit_info']['backend_used'] == 'fit' and hard.info['feas'] <= 1e-8
    np.testing.assert_allclose([hard.sol_x[0], hard.sol_y[0], hard.sol_x[-1], hard.sol_y[-1]], [sc.p0[0], sc.p0[1], sc.p1[0], sc.p1[1]], atol=0)
