"""GPU parity: plant / guidance / tracking kernels (through the C-ABI) against the oracle
and the golden vectors captured from the reference."""
import numpy as np
import pytest

from oracle import sim as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    import d2dhip
    c = d2dhip.Context(0)
    yield c
    c.close()


def _planes(a):          # (n, c) -> plane-major (c, n)
    return np.ascontiguousarray(np.asarray(a, float).T)


def test_step_vs_reference_odeint_and_oracle(ctx, gold):
    g = gold('plant')
    X, U, W = g['X'], g['U'], g['W']
    for tau in (0.01, 0.9667):
        out = np.empty_like(X)
        for i in range(len(X)):   # wind differs per row in the fixture -> one call per row
            o = ctx.step(ctx.dev(_planes(X[i:i + 1])), ctx.dev(_planes(U[i:i + 1])), W[i], tau, 1.0, 0.05)
            out[i] = o.cpu().numpy()[:, 0]
        ora = np.array([S.disc_dyn_glrk(X[i], U[i], W[i], 0.05, tau) for i in range(len(X))])
        d = out - ora; d[:, 2] = S.norm_mpi_pi(d[:, 2])
        assert np.abs(d).max() < 1e-11                          # same algorithm, fp64
        d = out - g[f'Xnext_tau{tau}']; d[:, 2] = S.norm_mpi_pi(d[:, 2])
        assert np.abs(d).max() < 1e-6                           # vs the reference's LSODA output


def test_step_large_batch_matches_oracle(ctx):
    rng = np.random.default_rng(3)
    n = 10007
    X = np.stack([rng.uniform(-100, 100, n), rng.uniform(-100, 100, n), rng.uniform(-np.pi, np.pi, n),
                  rng.uniform(-0.7, 0.7, n), rng.uniform(8, 16, n)], 1)
    U = np.stack([rng.uniform(-1, 1, n), rng.uniform(9, 16, n)], 1)
    o = ctx.step(ctx.dev(_planes(X)), ctx.dev(_planes(U)), (0.7, -0.4), 0.01, 1.0, 0.05).cpu().numpy().T
    ora = S.disc_dyn_glrk(X, U, (0.7, -0.4), 0.05, 0.01, 1.0)
    d = o - ora; d[:, 2] = S.norm_mpi_pi(d[:, 2])
    assert np.abs(d).max() < 1e-10


def test_step_one_panel_and_graded_meshes_match_oracle(ctx):
    """plant_step picks its mesh per drone by |phi - phi_c| (include/d2d.h D2D_GL_FAST_*): a batch that mixes both branches inside
    every wavefront, with rows at and just beyond the threshold, against the oracle's statement of the same rule."""
    rng = np.random.default_rng(5)
    n = 4099
    X = np.stack([rng.uniform(-100, 100, n), rng.uniform(-100, 100, n), rng.uniform(-np.pi, np.pi, n),
                  rng.uniform(-0.9, 0.9, n), rng.uniform(8, 16, n)], 1)
    dphi = np.where(rng.random(n) < 0.5, rng.uniform(-S.GL_FAST_DPHI, S.GL_FAST_DPHI, n), rng.uniform(-0.5, 0.5, n))
    dphi[:8] = [S.GL_FAST_DPHI, -S.GL_FAST_DPHI, np.nextafter(S.GL_FAST_DPHI, 1), -np.nextafter(S.GL_FAST_DPHI, 1), 0.0, 1e-9, 0.3, -0.3]
    U = np.stack([X[:, 3] - dphi, rng.uniform(9, 16, n)], 1)
    fast = np.abs(X[:, 3] - U[:, 0]) <= S.GL_FAST_DPHI
    assert 0.3 < fast.mean() < 0.8
    for tau, dt in ((0.01, 0.05), (0.9667, 0.05), (0.01, 0.2), (0.01, 0.06), (0.01, 0.061)):      # (dt > GL_FAST_RATIO tau: graded panels only)
        o = ctx.step(ctx.dev(_planes(X)), ctx.dev(_planes(U)), (0.7, -0.4), tau, 1.0, dt).cpu().numpy().T
        ora = S.disc_dyn_glrk(X, U, (0.7, -0.4), dt, tau, 1.0)
        d = o - ora; d[:, 2] = S.norm_mpi_pi(d[:, 2])
        assert np.abs(d).max() < 1e-10, (tau, dt, np.abs(d).max(), np.abs(d[fast]).max(), np.abs(d[~fast]).max())


def _gvf(ctx, g, n_rows, X0, n_form=1, **kw):
    c = np.tile(g['centres'], (n_form, 1)); N = 4 * n_form
    out = ctx.gvf_run(ctx.dev(_planes(np.tile(X0, (n_form, 1)))), ctx.dev(_planes(c)), ctx.dev(np.full(N, float(g['r']))),
                      4, n_rows, float(g['dt']), float(g['v_c']), float(g['ke']), float(g['kd']), float(g['kr']),
                      tau_phi=float(g['tau_phi']), **kw)
    ctx.sync()
    return out


def test_gvf_one_step_ahead_vs_reference_log(ctx, gold):
    """Every consecutive pair of rows kept from src/states_over_time.csv."""
    g = gold('states_over_time_sub')
    rows, X = g['rows'], g['X']
    worst = np.zeros(5)
    for a in range(len(rows) - 1):
        if rows[a + 1] != rows[a] + 1:
            continue
        out = _gvf(ctx, g, 2, X[a])
        Xn = out['X'].cpu().numpy()[1].T
        d = Xn - X[a + 1]; d[:, 2] = S.norm_mpi_pi(d[:, 2])
        worst = np.maximum(worst, np.abs(d).max(0))
    assert (worst[:2] < 5e-6).all() and (worst[2:] < 1e-6).all(), worst


def test_gvf_closed_loop_vs_oracle_and_log(ctx, gold):
    g = gold('states_over_time_sub')
    T = 401
    out = _gvf(ctx, g, T, g['X'][0], n_form=3)
    Xh = out['X'].cpu().numpy().transpose(0, 2, 1).reshape(T, 3, 4, 5)
    assert np.array_equal(Xh[:, 0], Xh[:, 1]) and np.array_equal(Xh[:, 0], Xh[:, 2])     # identical formations
    c = g['centres']; kw = dict(ke=float(g['ke']), kd=float(g['kd']), kr=float(g['kr']), tau_phi=float(g['tau_phi']))
    Xo, Uo, Rro, etho, _ = S.formation_gvf_run(c, float(g['r']), float(g['v_c']), g['X'][0], T, float(g['dt']), **kw)
    d = Xh[:, 0] - Xo; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 1e-8                                    # fp64 same algorithm over 400 steps
    d = Xh[:, 0] - g['X'][:T]; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 2e-4                                    # drift of the reference's LSODA tolerance
    U = out['U'].cpu().numpy().transpose(0, 2, 1)[:, :4]
    np.testing.assert_allclose(U[:T - 1], Uo[:T - 1], atol=1e-9)
    Rr = out['Rr'].cpu().numpy()[:, :4]
    np.testing.assert_allclose(Rr, Rro, atol=1e-7)
    eth = out['eth'].cpu().numpy()[:, :3]
    np.testing.assert_allclose(eth, etho, atol=1e-7)
    assert (out['stop_row'].cpu().numpy() == T).all()


def test_gvf_stop_rule_and_general_B(ctx):
    rng = np.random.default_rng(5)
    n_ac, n_form, T = 3, 5, 300
    c = rng.uniform(-20, 20, (n_form, n_ac, 2)); X0 = np.zeros((n_form, n_ac, 5))
    X0[..., 0] = rng.uniform(10, 40, (n_form, n_ac)); X0[..., 1] = rng.uniform(10, 40, (n_form, n_ac))
    X0[..., 2] = rng.uniform(-3, 3, (n_form, n_ac)); X0[..., 4] = 12.0
    B = np.array([[-1.0, 0.5], [1.0, -1.0], [0.0, 0.5]]); zd = np.array([0.3, -0.2])
    # oracle first, to pick stop targets that are actually reached
    ref = [S.formation_gvf_run(c[f], 50.0, 13.0, X0[f], T, 0.05, ke=4e-4, kd=25.0, kr=5.0, z_des=zd, B=B) for f in range(n_form)]
    X0f = np.array([ref[f][0][60 + 20 * f] for f in range(n_form)])          # state at a known step
    tol = (0.5, 0.5, 0.05)
    refs = [S.formation_gvf_run(c[f], 50.0, 13.0, X0[f], T, 0.05, ke=4e-4, kd=25.0, kr=5.0, z_des=zd, B=B,
                                X0f=X0f[f], stop_tol=tol) for f in range(n_form)]
    N = n_form * n_ac
    out = ctx.gvf_run(ctx.dev(_planes(X0.reshape(N, 5))), ctx.dev(_planes(c.reshape(N, 2))), ctx.dev(np.full(N, 50.0)),
                      n_ac, T, 0.05, 13.0, 4e-4, 25.0, 5.0, B=B, z_des=zd, X0f=ctx.dev(_planes(X0f.reshape(N, 5)[:, :3])),
                      stop_tol=tol)
    ctx.sync()
    stop = out['stop_row'].cpu().numpy()
    Xh = out['X'].cpu().numpy().transpose(0, 2, 1).reshape(T, n_form, n_ac, 5)
    for f in range(n_form):
        assert stop[f] == refs[f][4], (f, stop[f], refs[f][4])
        s = stop[f]
        d = Xh[:s, f] - refs[f][0][:s]; d[..., 2] = S.norm_mpi_pi(d[..., 2])
        assert np.abs(d).max() < 1e-8
        assert stop[f] < T


def test_ctrl_gain_vs_golden_and_oracle(ctx, gold):
    g = gold('flatness_ctrl')
    n = len(g['Y'])
    Yref = np.concatenate([g['Y'], g['Yd'], g['Ydd'], g['Yddd']], 1)       # (n, 8)
    for i in range(n):        # wind differs per row
        Xr, dX, U, K = ctx.ctrl_gain(ctx.dev(_planes(g['X'][i:i + 1])), ctx.dev(_planes(Yref[i:i + 1])), w=tuple(g['W'][i]))
        Xr, dX, U, K = (t.cpu().numpy()[:, 0] for t in (Xr, dX, U, K))
        np.testing.assert_allclose(Xr, g['gain_Xr_carestandin'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(dX, g['gain_dX_carestandin'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(K.reshape(2, 5), g['gain_K_carestandin'][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(U, g['gain_U_carestandin'][i], rtol=1e-8, atol=1e-9)


def test_dfff_vs_golden_and_oracle(ctx, gold):
    """d2d_dfff_eval (legacy DFFFController: flatness feed-forward + 3x3 Riccati feedback on the device)."""
    g = gold('dfff_carestandin')
    n = len(g['X'])
    Yref = np.concatenate([g['Y'], g['Yd'], g['Ydd']], 1)                    # (n, 6)
    tau_phi, tau_v = float(g['tau_phi']), float(g['tau_v'])
    for i in range(n):        # wind differs per row
        Xr, U, K = ctx.dfff_eval(ctx.dev(_planes(g['X'][i:i + 1])), ctx.dev(_planes(Yref[i:i + 1])), tuple(g['W'][i]), tau_phi, tau_v)
        Xr, U, K = (t.cpu().numpy()[:, 0] for t in (Xr, U, K))
        np.testing.assert_allclose(Xr, g['Xr'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(K.reshape(2, 3), g['K'][i][:, :3], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(U, g['U'][i], rtol=1e-8, atol=1e-9)
    # one batched call (rows without wind) and another lag constant against the oracle
    idx = np.where(np.abs(g['W']).sum(1) == 0)[0]
    for tau in (0.01, 0.9667):
        Xr, U, K = ctx.dfff_eval(ctx.dev(_planes(g['X'][idx])), ctx.dev(_planes(Yref[idx])), (0.0, 0.0), tau, 1.0)
        U = U.cpu().numpy().T; K = K.cpu().numpy().T.reshape(len(idx), 2, 3)
        for j, i in enumerate(idx):
            Ys = np.array([g['Y'][i], g['Yd'][i], g['Ydd'][i]])
            Uo, Ko, _ = S.dfff_get(g['X'][i], Ys, (0, 0), tau, 1.0)
            np.testing.assert_allclose(K[j], Ko[:, :3], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(U[j], Uo, rtol=1e-8, atol=1e-9)


def test_ctrl_gain_batch_and_tau(ctx):
    rng = np.random.default_rng(11)
    n = 777
    psi = rng.uniform(-np.pi, np.pi, n); sp = rng.uniform(9, 15, n)
    Y = np.stack([rng.uniform(-80, 80, n), rng.uniform(-80, 80, n), sp * np.cos(psi), sp * np.sin(psi),
                  rng.uniform(-4, 4, n), rng.uniform(-4, 4, n), np.zeros(n), np.zeros(n)], 1)
    X = np.stack([Y[:, 0] + rng.uniform(-5, 5, n), Y[:, 1] + rng.uniform(-5, 5, n), psi + rng.uniform(-0.5, 0.5, n),
                  rng.uniform(-0.3, 0.3, n), sp + rng.uniform(-1, 1, n)], 1)
    for tau in (0.01, 0.9667):
        Xr, dX, U, K = ctx.ctrl_gain(ctx.dev(_planes(X)), ctx.dev(_planes(Y)), tau_phi=tau)
        K = K.cpu().numpy().T.reshape(n, 2, 5); U = U.cpu().numpy().T
        for i in range(0, n, 37):
            _, _, Uo, Ko = S.compute_gain(X[i], Y[i, 0:2], Y[i, 2:4], Y[i, 4:6], Y[i, 6:8], (0, 0), tau, 1.0)
            np.testing.assert_allclose(K[i], Ko, rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(U[i], Uo, rtol=1e-8, atol=1e-9)


def test_track_run_vs_reference_trace(ctx, gold):
    """100-step phase-2/3 loop body executed by the reference's classes (CARE stand-in)."""
    g = gold('tracking_trace_carestandin')
    T, n = g['x_ref'].shape
    dt = g['time'][1] - g['time'][0]
    out = ctx.track_run(ctx.dev(g['x_ref']), ctx.dev(g['y_ref']), ctx.dev(_planes(g['X'][0])), dt,
                        tau_phi=float(g['tau_phi']), tau_v=float(g['tau_v']))
    ctx.sync()
    X = out['X'].cpu().numpy().transpose(0, 2, 1); U = out['U'].cpu().numpy().transpose(0, 2, 1)
    Xr = out['Xr'].cpu().numpy().transpose(0, 2, 1); dX = out['dX'].cpu().numpy().transpose(0, 2, 1)
    # closed loop over 100 steps against the LSODA-integrated reference: tolerance = accumulated
    # integrator difference (reference local error ~1e-6 m per step on |x| ~ 100 m)
    d = X - g['X']; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 2e-4, np.abs(d).max()
    np.testing.assert_allclose(Xr[:T - 1], g['Xr'][:T - 1], rtol=1e-10, atol=1e-10)
    assert np.abs(U[:T - 1] - g['U'][:T - 1]).max() < 2e-3
    # and against the oracle running the same integrator: tight
    Xo, Uo, Xro, Ydo, Yddo, dXo, Ko = S.track_run(g['time'], g['x_ref'], g['y_ref'], g['X'][0], (0, 0),
                                                  float(g['tau_phi']), float(g['tau_v']))
    d = X - Xo; d[..., 2] = S.norm_mpi_pi(d[..., 2])
    assert np.abs(d).max() < 1e-7
    np.testing.assert_allclose(U[:T - 1], Uo[:T - 1], atol=1e-6)
    np.testing.assert_allclose(dX[:T - 1], dXo[:T - 1], atol=1e-7)
    Yd = out['Yd'].cpu().numpy().transpose(0, 2, 1); Ydd = out['Ydd'].cpu().numpy().transpose(0, 2, 1)
    np.testing.assert_allclose(Yd[:T - 1], Ydo[:T - 1], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Ydd[:T - 1], Yddo[:T - 1], rtol=1e-11, atol=1e-10)


def test_error_paths(ctx):
    import d2dhip
    with pytest.raises(d2dhip.D2DError):
        ctx.step(ctx.zeros(5, 4), ctx.zeros(2, 4), (0, 0), -1.0, 1.0, 0.05)
    with pytest.raises(d2dhip.D2DError):
        ctx.gvf_run(ctx.zeros(5, 130), ctx.zeros(2, 130), ctx.zeros(130), 65, 3, 0.05, 12.0)


@pytest.mark.gpu
def test_dfff_run_loop(gold):
    """d2d_sim_dfff_run (run_simulation of 05_test_simulation.py with the legacy DFFFController): against the oracle's loop
    with the same integrator (tight), against the reference's own trace (bounded by its LSODA tolerances), and through the
    reference-named host function with the mirror's trajectory classes."""
    import d2dhip
    from oracle import sim as S
    g = gold('dfff_run_carestandin')
    ctx = d2dhip.default_context()
    T = len(g['time'])
    n = 70                                                     # more than one wave; copies with different start states
    rng = np.random.default_rng(1)
    X0 = np.tile(g['X'][0], (n, 1)); X0[1:, :2] += rng.uniform(-3, 3, (n - 1, 2)); X0[1:, 4] += rng.uniform(-1, 1, n - 1)
    Yd = np.ascontiguousarray(np.repeat(g['Yref'][:, None, :3, :], n, 1).transpose(0, 2, 3, 1).reshape(T, 6, n))
    P = np.ascontiguousarray(np.repeat(g['perts'][:, None], n, 1).transpose(0, 2, 1))
    kw = dict(w=tuple(g['W']), tau_phi=float(g['tau_phi']), tau_v=float(g['tau_v']))
    out = ctx.dfff_run(ctx.dev(Yd), ctx.dev(np.ascontiguousarray(X0.T)), float(g['time'][1] - g['time'][0]), perts=ctx.dev(P), **kw)
    ctx.sync()
    X = out['X'].cpu().numpy().transpose(0, 2, 1); U = out['U'].cpu().numpy().transpose(0, 2, 1); Xr = out['Xr'].cpu().numpy().transpose(0, 2, 1)
    for j in (0, 1, 69):
        Xo, Uo, Xro = S.dfff_run(g['time'], g['Yref'], X0[j], perts=g['perts'], W=tuple(g['W']), tau_phi=float(g['tau_phi']),
                                 tau_v=float(g['tau_v']), integrator='glrk')
        np.testing.assert_allclose(Xr[:, j], Xro, rtol=0, atol=1e-12)
        np.testing.assert_allclose(X[:, j], Xo, rtol=0, atol=2e-8)
        np.testing.assert_allclose(U[:, j], Uo, rtol=0, atol=2e-7)
    np.testing.assert_array_equal(out['X_final'].cpu().numpy().T, X[-1])
    # the reference's trace
    np.testing.assert_allclose(X[:, 0, 2:], g['X'][:, 2:], rtol=0, atol=2e-5)
    np.testing.assert_allclose(X[:, 0, :2], g['X'][:, :2], rtol=0, atol=1e-4)
    assert np.abs(U[:, 0] - g['U']).max() < 1e-3


def test_gvf_phase_error_stop_rules_of_cases_2_and_3():
    """The stop rules of src/12_full_sim_case2.py:156-164 (every phase error <= 0.5 deg: break at once) and
    src/12_full_sim_case3.py:163-178 (<= 2 deg, then 0.7 s more while the planner works) inside the device loop, against the
    oracle's restatement of those loops: same break row, same convergence index, same trimmed arrays."""
    import full_sim
    from oracle import sim as S
    c = np.array([[0, -20], [25, -40], [25, -80], [0, -100.0]])
    X0 = np.tile([20, 30, -np.pi / 2, 0, 10.0], (4, 1))
    n_steps = len(np.arange(0, 120, 0.05))
    Xo, Uo, Rro, etho, stop2, conv2 = S.formation_gvf_run(c, 60.0, 15.0, X0, n_steps, 0.05, etheta_tol_deg=0.5)
    X, U, U1, U2, Rr, eth, time, t_f = full_sim.CircularFormationGVF_case2(c, 60.0, 15.0, 4, 0, 0.05, 120)
    assert 10 < stop2 < n_steps and len(X) == stop2 == len(time) and abs(t_f - conv2[2]) < 1e-12, (stop2, len(X))
    assert np.abs(X - Xo[:stop2]).max() < 1e-7 and np.abs(eth - etho[:stop2]).max() < 1e-6
    Xo, Uo, Rro, etho, stop3, conv3 = S.formation_gvf_run(c, 60.0, 15.0, X0, n_steps, 0.05, etheta_tol_deg=2.0, t_opt_comp=0.7)
    out = full_sim.CircularFormationGVF_case3(c, 60.0, 15.0, 4, 0, 0.05, 120)
    X3, conv = out[0], out[-1]
    assert len(X3) == stop3 and conv[0] == conv3[0] and abs(conv[1] - conv3[1]) < 1e-12 and abs(conv[2] - conv3[2]) < 1e-12, (conv, conv3, stop3)
    assert stop3 - 1 - conv3[0] == full_sim.hold_steps(0.7, 0.05) + 1           # first convergence, then the planner's 0.7 s
    assert np.abs(X3 - Xo[:stop3]).max() < 1e-7
    # case 3's planning step: one single-aircraft plan + the wingman's offset copy
    import single_opt_planner as sop
    sop.exp_1.p0 = tuple(X3[-1, 0])
    p = full_sim.trajectory_optimization_single(sop.exp_1, (0., -20.))
    assert p.sol_x.shape == (121, 2) and np.allclose(p.sol_y[:, 1] - p.sol_y[:, 0], -20.) and np.allclose(p.sol_x[:, 1], p.sol_x[:, 0])
