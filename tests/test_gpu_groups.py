"""GPU parity: coupled groups (BASELINE configs[2] -- multi_opt_planner with collision rows between the
aircraft of one scenario) against the oracle's block Gauss-Seidel and the scipy joint arbiter."""
import numpy as np
import pytest

from oracle import fit as F

pytestmark = pytest.mark.gpu
K, S_ = 50, 6
DUR = F.planner_timing(0, 4.9, 10)[2]


def circle_scenarios(n_ac, R, seed=0, sigma=2.0, pair01_only=False):
    """BASELINE configs[2] scenario rows (d2dhip.synth.circle_group_scenarios, obj_scale = 1)."""
    from d2dhip import synth
    return synth.circle_group_scenarios(n_ac, R, DUR, K, seed=seed, sigma=sigma, pair01_only=pair01_only, obj_scale=1.0)


@pytest.fixture(scope='module')
def env():
    import d2dhip
    ctx = d2dhip.Context(0)
    s = 1.0 / K
    plan = d2dhip.FitPlan(ctx, S_, K, DUR, (0.02 ** 2, s * 5.0 / 8, s / 8 / F.G_ACC ** 2))
    ob = F.FitBasis.from_arrays(S_, K, DUR, *plan.basis())
    yield ctx, plan, ob
    plan.close(); ctx.close()


@pytest.mark.parametrize('n_ac,pair01', [(8, False), (4, False), (4, True), (2, False)])
def test_groups_vs_oracle_bgs_and_joint_arbiter(env, n_ac, pair01):
    from scipy.optimize import least_squares
    ctx, plan, ob = env
    R = 6
    sc = circle_scenarios(n_ac, R, seed=n_ac, pair01_only=pair01)
    dsc = ctx.dev(sc.reshape(R * n_ac, -1))
    q = plan.init(dsc)
    cost, sweeps, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=80, inner_iters=8, tol=1e-12)
    plan.set_groups(1)
    qh = q.cpu().numpy().reshape(R, n_ac, -1)
    ch = cost.cpu().numpy().reshape(R, n_ac)
    assert sweeps < 80, (sweeps, stats)
    for r in range(0, R, 2):
        # (a) joint arbiter: scipy LM on the stacked joint residual, started at the GPU solution, must not move
        fun = lambda x: F.group_residuals(ob, sc[r], x)       # noqa: E731
        pol = least_squares(fun, qh[r].reshape(-1), method='lm', xtol=1e-14, ftol=1e-14, gtol=1e-14)
        zg = np.array([F.coefficients(ob, sc[r, i], qh[r, i]) for i in range(n_ac)])
        zp = np.array([F.coefficients(ob, sc[r, i], pol.x.reshape(n_ac, -1)[i]) for i in range(n_ac)])
        assert np.abs(zg - zp).max() <= 1e-6 * np.abs(zp).max(), np.abs(zg - zp).max() / np.abs(zp).max()
        cj = F.group_cost(ob, sc[r], qh[r])
        assert abs(2 * pol.cost - cj) <= 1e-6 * cj
        # (b) the oracle's block Gauss-Seidel (same algorithm, fp64 Hessian)
        # (the persistent kernel follows its slow sweeps by a line search on the joint cost: ls=True)
        qo, co, swo = F.bgs_solve(ob, sc[r], sweeps=80, inner_iters=8, tol=1e-12, ls=True)
        zo = np.array([F.coefficients(ob, sc[r, i], qo[i]) for i in range(n_ac)])
        assert np.abs(zg - zo).max() <= 1e-6 * np.abs(zo).max()
        assert abs(co - cj) <= 1e-6 * cj
        # (c) per-aircraft sub-problem costs reported by the library
        pos = F.group_positions(ob, sc[r], qh[r])
        for i in range(n_ac):
            oth = [pos[j] for j in F.partners(sc[r, i], i, n_ac)]
            assert abs(ch[r, i] - F.cost(ob, sc[r, i], qh[r, i], others=oth)) <= 1e-9 * max(ch[r, i], 1e-3)
    # coupling did something: aircraft do not fly through the centre together
    if not pair01 and n_ac >= 4:       # (two aircraft already pass 30 m apart on their dog-legs)
        pos = F.group_positions(ob, sc[0], qh[0])
        d = min(np.hypot(*(pos[i] - pos[j])).min() for i in range(n_ac) for j in range(i + 1, n_ac))
        q0 = plan.init(dsc); plan.solve(dsc, q0)
        pos0 = F.group_positions(ob, sc[0], q0.cpu().numpy().reshape(R, n_ac, -1)[0])
        d0 = min(np.hypot(*(pos0[i] - pos0[j])).min() for i in range(n_ac) for j in range(i + 1, n_ac))
        # the uncoupled fits may already keep clear of each other (which minimum they reach depends on the LM variant):
        # coupling must push apart the ones that come within the collision radius and never pull anybody closer
        rcol = sc[0, 0, F.SC_RCOL]
        assert d > d0 + 1.0 if d0 < 0.8 * rcol else d >= d0 - 1e-3, (d, d0)


def test_line_search_on_slow_scenarios_vs_oracle(monkeypatch):
    """The slow scenarios of BASELINE configs[2] (seed 1 of the bench: 96 and 90 plain sweeps to 1e-6): with the line search on the
    joint cost (include/d2d.h D2D_GS_LS_*) the kernel settles them in a fraction of the sweeps, at the fixed point of the plain
    sweeps, and does what the oracle's statement of the same rule does (sweep counts, coefficients, joint cost)."""
    import d2dhip
    from d2dhip import synth
    ctx = d2dhip.Context(0)
    plan = d2dhip.FitPlan(ctx, S_, K, DUR, synth.default_wref(1.0, K))
    ob = F.FitBasis.from_arrays(S_, K, DUR, *plan.basis())
    n_ac, ids = 8, [2171, 308]
    sc = synth.circle_group_scenarios(n_ac, 8192, DUR, K, seed=1)[ids]
    dsc = ctx.dev(sc.reshape(len(ids) * n_ac, -1))
    try:
        q0 = plan.init(dsc)
        plan.solve_groups(dsc, q0, n_ac, max_sweeps=200, inner_iters=8, tol=1e-6, gs_ls=0)      # d2d_fit_opts.gs_ls = 0: plain sweeps
        sw_plain, _ = plan.group_report(len(ids))
        q = plan.init(dsc)
        plan.solve_groups(dsc, q, n_ac, max_sweeps=200, inner_iters=8, tol=1e-6)
        sw_ls, mv_ls = plan.group_report(len(ids))
    finally:
        plan.set_groups(1)
    assert (sw_plain >= 80).all() and (sw_ls <= 40).all() and (mv_ls <= 1e-6).all(), (sw_plain, sw_ls, mv_ls)
    qh, qp = q.cpu().numpy().reshape(len(ids), n_ac, -1), q0.cpu().numpy().reshape(len(ids), n_ac, -1)
    for r in range(len(ids)):
        cj, cp = F.group_cost(ob, sc[r], qh[r]), F.group_cost(ob, sc[r], qp[r])
        assert abs(cj - cp) <= 1e-6 * cp, (cj, cp)                        # the same minimum as the plain sweeps ...
        assert abs(F.group_merit(ob, sc[r], qh[r]) - cj) <= 1e-12 * cj     # (the merit the line search adds up IS the joint cost)
        tr = []
        qo, co, swo = F.bgs_solve(ob, sc[r], sweeps=200, inner_iters=8, tol=1e-6, ls=True, trace=tr)
        assert abs(swo - sw_ls[r]) <= 2, (swo, sw_ls[r], tr)                # ... reached the way the oracle reaches it
        assert any(al > 0 for _, _, al in tr)
        zg = np.array([F.coefficients(ob, sc[r, i], qh[r, i]) for i in range(n_ac)])
        zo = np.array([F.coefficients(ob, sc[r, i], qo[i]) for i in range(n_ac)])
        assert np.abs(zg - zo).max() <= 1e-4 * np.abs(zo).max()             # (both stopped at a sweep that moved <= 1e-6)
        assert abs(co - cj) <= 1e-6 * cj
    plan.close(); ctx.close()


def test_groups_config2_batch_properties(env):
    """configs[2] shape at reduced replica count: 8 aircraft x 256 replicas in one call; every group ends
    stationary (largest relative move of the last sweep <= tol) and identical replicas give identical answers."""
    ctx, plan, ob = env
    n_ac, R = 8, 256
    sc = circle_scenarios(n_ac, R, seed=1)
    sc[1] = sc[0]                                             # two identical scenarios
    dsc = ctx.dev(sc.reshape(R * n_ac, -1))
    q = plan.init(dsc)
    cost, sweeps, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=300, inner_iters=8, tol=1e-10)
    assert stats[2] <= 1e-10 and sweeps < 300, (sweeps, stats)
    qh = q.cpu().numpy().reshape(R, n_ac, -1)
    np.testing.assert_array_equal(qh[0], qh[1])
    assert np.isfinite(cost.cpu().numpy()).all()
    # the scheduling hint (longest-sweeping scenarios first) changes the hand-out order, never a result
    plan.group_order_from_last(R)
    q2 = plan.init(dsc)
    cost2, sweeps2, stats2 = plan.solve_groups(dsc, q2, n_ac, max_sweeps=300, inner_iters=8, tol=1e-10)
    assert sweeps2 == sweeps
    np.testing.assert_array_equal(q2.cpu().numpy(), q.cpu().numpy())
    np.testing.assert_array_equal(cost2.cpu().numpy(), cost.cpu().numpy())
    plan.group_order_from_last(R, False)
    import d2dhip
    with pytest.raises(d2dhip.D2DError):
        plan.group_order_from_last(R + 1)          # no sweep counts of a solve over R + 1 scenarios
    plan.set_groups(1)


def test_long_horizon_groups_on_the_chunked_kernel(monkeypatch):
    """Coupled groups beyond the LDS image of the group kernels (07_multioptyplan's 50 Hz scenarios: exp_2 276 nodes, exp_5 401):
    every visit of the block Gauss-Seidel is one launch of the chunked persistent kernel with the collision rows in its phase 1.
    At K = 71 both paths exist: the same fixed points as the launch-pair kernels and as the oracle's bgs_solve; at K = 301 (only
    the chunked path) against the oracle and the joint scipy arbiter."""
    import d2dhip
    from d2dhip import synth
    from scipy.optimize import least_squares
    ctx = d2dhip.Context(0)
    try:
        for K2, hz, both in ((71, 10.0, True), (301, 50.0, False)):
            dur = F.planner_timing(0, (K2 - 1) / hz, hz)[2]
            n_ac, R = 4, 3
            s = 1.0 / K2
            plan = d2dhip.FitPlan(ctx, S_, K2, dur, (0.02 ** 2, s * 5.0 / n_ac, s / n_ac / F.G_ACC ** 2))
            ob = F.FitBasis.from_arrays(S_, K2, dur, *plan.basis())
            sc = synth.circle_group_scenarios(n_ac, R, dur, K2, seed=3, obj_scale=1.0)
            dsc = ctx.dev(sc.reshape(R * n_ac, -1))
            try:
                assert plan.kernel == 'long'
                qa = plan.init(dsc)
                ca, swa, sta = plan.solve_groups(dsc, qa, n_ac, max_sweeps=80, inner_iters=8, tol=1e-12)      # (the default beyond 64 nodes)
                assert swa < 80 and sta[2] <= 1e-12, (swa, sta)
                qh = qa.cpu().numpy().reshape(R, n_ac, -1)
                if both:
                    qb = plan.init(dsc)         # d2d_fit_opts.gs_pairs = 1: the launch pairs of round 1, where their LDS image holds K
                    cb, swb, stb = plan.solve_groups(dsc, qb, n_ac, max_sweeps=80, inner_iters=8, tol=1e-12, gs_pairs=1)
                    assert np.abs(qa.cpu().numpy() - qb.cpu().numpy()).max() <= 1e-6 * np.abs(qb.cpu().numpy()).max()
                    np.testing.assert_allclose(ca.cpu().numpy(), cb.cpu().numpy(), rtol=1e-8)
                r = 0
                qo, co, swo = F.bgs_solve(ob, sc[r], sweeps=80, inner_iters=8, tol=1e-12)
                zg = np.array([F.coefficients(ob, sc[r, i], qh[r, i]) for i in range(n_ac)])
                zo = np.array([F.coefficients(ob, sc[r, i], qo[i]) for i in range(n_ac)])
                assert np.abs(zg - zo).max() <= 1e-6 * np.abs(zo).max()
                cj = F.group_cost(ob, sc[r], qh[r])
                assert abs(co - cj) <= 1e-6 * cj
                pol = least_squares(lambda x: F.group_residuals(ob, sc[r], x), qh[r].reshape(-1), method='lm', xtol=1e-14, ftol=1e-14, gtol=1e-14)
                assert abs(2 * pol.cost - cj) <= 1e-6 * cj
                pos = F.group_positions(ob, sc[r], qh[r])
                chh = ca.cpu().numpy().reshape(R, n_ac)
                for i in range(n_ac):
                    oth = [pos[j] for j in F.partners(sc[r, i], i, n_ac)]
                    assert abs(chh[r, i] - F.cost(ob, sc[r, i], qh[r, i], others=oth)) <= 1e-9 * max(chh[r, i], 1e-3)
            finally:
                plan.set_groups(1)
                plan.close()
    finally:
        ctx.close()
