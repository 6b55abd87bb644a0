"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the oracle cannot solve 32 768 fits in
seconds), plus the contraction-only launch pair d2d_fit_rows / d2d_fit_jtj and the scheduling hint of the LM kernel.

  configs[1]/[3]  32 768 independent fits: descent, stationarity (|J^T r| small at every converged fit), cost re-evaluated by a
                  second kernel, a seeded sample checked against the oracle's cost function, invariance under the hand-out order
  configs[2]      8 aircraft x 8192 replicas, collision rows: the replicas that share a scenario row agree, a sample is checked
                  against the oracle's joint cost, stationarity of the sub-problems
  configs[4]      65 536 drones x 10 000 GVF steps with rec_stride (history subsampled): formation 0 against the oracle's loop on
                  the kept rows, translation invariance across formations, idempotence of the frozen final state
"""
import numpy as np
import pytest

from oracle import fit as F, sim as S

pytestmark = pytest.mark.gpu

K, S_ = 50, 6
DUR = F.planner_timing(0, 4.9, 10)[2]
SS = 0.1 / K
WREF = (0.02 ** 2, SS * 5.0, SS / F.G_ACC ** 2)


@pytest.fixture(scope='module')
def ctx():
    import d2dhip
    c = d2dhip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope='module')
def plan(ctx):
    import d2dhip
    p = d2dhip.FitPlan(ctx, S_, K, DUR, WREF)
    yield p
    p.close()


@pytest.fixture(scope='module')
def obasis(plan):
    G, Gp, Z, Zp, P = plan.basis()
    return F.FitBasis.from_arrays(S_, K, DUR, G, Gp, Z, Zp, P)


def test_rows_then_jtj_equals_eval(ctx, plan, obasis):
    """The contraction-only kernel (records from HBM) gives the SAME J^T J as the one-launch evaluation, bit for bit
    (same operands, same MFMA order), and the oracle's to fp32 accuracy; ragged batch; wrong call order is refused."""
    import d2dhip
    B = 173
    sc = F.set_scale(F.synth_scenarios(B, seed=5), 0.1, K)
    dsc = ctx.dev(sc)
    rng = np.random.default_rng(1)
    qh = plan.init(dsc).cpu().numpy() + rng.normal(0, 0.4, (B, 48))
    dq = ctx.dev(qh)
    c1, g1, H1 = plan.eval(dsc, dq)
    c2, g2 = plan.rows(dsc, dq)
    H2 = plan.jtj(B)
    ctx.sync()
    assert np.array_equal(c1.cpu().numpy(), c2.cpu().numpy())
    assert np.array_equal(g1.cpu().numpy(), g2.cpu().numpy())
    assert np.array_equal(H1.cpu().numpy(), H2.cpu().numpy())
    Hh = H2.cpu().numpy()
    for i in (0, 77, 172):
        _, _, Ho = F.eval_normal(obasis, sc[i], qh[i])
        assert np.abs(Hh[i] - Ho).max() <= 2e-5 * np.abs(Ho).max()
    with pytest.raises(d2dhip.D2DError):
        plan.jtj(B - 1)                               # no records for that batch size


def test_32768_fits_properties_and_order_invariance(ctx, plan, obasis):
    """BASELINE configs[3], one GPU's share: 32 768 fits in one solve."""
    import d2dhip
    import torch
    B = 32768
    sc = F.set_scale(F.synth_scenarios(B, seed=20241008), 0.1, K)
    dsc = ctx.dev(sc)
    q0 = plan.init(dsc)
    c0, _, _ = plan.eval(dsc, q0, want_H=False)
    q = q0.clone()
    cost, iters, status, stats = plan.solve(dsc, q)
    st = status.cpu().numpy()
    conv = np.isin(st, (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED))
    assert conv.mean() >= 0.998, conv.mean()
    assert not (st == d2dhip.ST_NONFINITE).any()
    # descent everywhere, and the cost the solver reports is the cost a second kernel evaluates at the returned point
    c1, g1, _ = plan.eval(dsc, q, want_H=False)
    c0h, c1h, ch = c0.cpu().numpy(), c1.cpu().numpy(), cost.cpu().numpy()
    assert (ch <= c0h * (1 + 1e-12)).all()
    assert np.abs(c1h - ch).max() <= 2e-11 * np.abs(ch).max()        # (knot / Hermite evaluation in the solver, dense basis in d2d_fit_eval)
    # stationarity: |J^T r|_inf tiny relative to the scale of the gradient at the start, at every converged fit
    _, g0, _ = plan.eval(dsc, q0, want_H=False)
    gn1 = g1.abs().max(1).values.cpu().numpy(); gn0 = g0.abs().max(1).values.cpu().numpy()
    assert (gn1[conv] <= 1e-6 * np.maximum(gn0[conv], 1e-3)).mean() >= 0.999
    assert abs(stats[0] - ch.sum()) <= 1e-9 * ch.sum() and stats[2] == (~conv).sum()
    # a seeded sample against the oracle's cost function at the GPU's solution
    qh = q.cpu().numpy()
    for i in np.random.default_rng(0).integers(0, B, 12):
        co = F.cost(obasis, sc[i], qh[i])
        assert abs(ch[i] - co) <= 1e-10 * co
    # the scheduling hint changes which wave solves which fit, and nothing else: same bits
    plan.order_from_iters(iters)
    q2 = q0.clone()
    cost2, iters2, status2, _ = plan.solve(dsc, q2)
    plan.clear_order()
    assert torch.equal(q, q2) and torch.equal(iters, iters2) and torch.equal(cost, cost2) and torch.equal(status, status2)


@pytest.mark.parametrize('B', [1, 3, 5, 300, 1027, 2049])
def test_order_hint_small_and_ragged_batches(ctx, plan, B):
    """The ordered hand-out with its exclusive SIMDs (fit_lm_kernel: wave 4 of the first workgroups gives its position to the queue
    and waits for its SIMD-mate) on batches smaller than the grid, around the first queue positions (1024 + excl) and just past
    one round of static positions (2048): same bits as the index-order solve, every fit handled exactly once."""
    import torch
    from d2dhip import synth
    sc = synth.synth_scenarios(B, seed=77, obj_scale=0.1, K=K)
    dsc = ctx.dev(sc)
    q0 = plan.init(dsc)
    plan.clear_order()
    qa = q0.clone()
    ca, ia, sa, _ = plan.solve(dsc, qa)
    plan.order_from_iters(ia)
    qb = q0.clone()
    cb, ib, sb, _ = plan.solve(dsc, qb)
    plan.clear_order()
    assert torch.equal(qa, qb) and torch.equal(ia, ib) and torch.equal(ca, cb) and torch.equal(sa, sb)
    assert (sa.cpu().numpy() != 0).all()                    # nobody left RUNNING


def test_8x8192_groups_properties(ctx, obasis):
    """BASELINE configs[2]: 8-drone circular formation x 8192 replicas with collision rows (block Gauss-Seidel)."""
    import d2dhip
    from d2dhip import synth
    n_ac, R = 8, 8192
    s = 1.0 / K
    wref = (0.02 ** 2, s / n_ac * 5.0, s / n_ac / F.G_ACC ** 2)
    p = d2dhip.FitPlan(ctx, S_, K, DUR, wref)
    try:
        sc = synth.circle_group_scenarios(n_ac, R, DUR, K=K, seed=3, sigma=2.0)
        sc[R // 2:] = sc[:R // 2]                     # the second half repeats the first: replicas of identical scenarios
        rows = sc.reshape(R * n_ac, -1)
        dsc = ctx.dev(rows)
        q = p.init(dsc)
        cost, sweeps, stats = p.solve_groups(dsc, q, n_ac, max_sweeps=150, inner_iters=8, tol=1e-9)
        qh = q.cpu().numpy().reshape(R, n_ac, 48); ch = cost.cpu().numpy().reshape(R, n_ac)
        assert np.isfinite(qh).all() and stats[2] <= 1e-6
        # EVERY scenario settled to the north-star's tolerance: its own last sweep moved nothing by more than 1e-6
        sw, mv = p.group_report(R)
        assert (mv <= 1e-6).all(), (int((mv > 1e-6).sum()), float(mv.max()))
        assert sw.max() == sweeps and np.percentile(sw, 99) <= 60, (sw.max(), np.percentile(sw, 99))
        # identical scenarios, identical answers (determinism across workgroups / positions in the batch)
        assert np.array_equal(qh[:R // 2], qh[R // 2:])
        ob = F.FitBasis.from_arrays(S_, K, DUR, *p.basis())
        for r in (0, 1234, 4095):
            scs = sc[r]
            # sub-problem costs: own rows + collision rows against the others' final positions
            pos = F.group_positions(ob, scs, qh[r])
            for i in (0, 5):
                oth = [pos[j] for j in F.partners(scs[i], i, n_ac)]
                co = F.cost(ob, scs[i], qh[r, i], others=oth)
                assert abs(ch[r, i] - co) <= 1e-9 * co
                # stationarity of the sub-problem
                _, go, _ = F.eval_normal(ob, scs[i], qh[r, i], others=oth)
                assert np.abs(go).max() <= 1e-5
        # the collision rows did something: every pair keeps apart around the centre crossing
        pos = F.group_positions(ob, sc[0], qh[0])
        dmin = min(np.hypot(pos[i][0] - pos[j][0], pos[i][1] - pos[j][1]).min() for i in range(n_ac) for j in range(i))
        assert dmin > 1.0
    finally:
        p.close()


def test_65536_drones_10000_steps_gvf(ctx):
    """BASELINE configs[4] at full size; history kept every 500th row so that the check reads 21 rows, not 36 GB."""
    n_ac, N, rows, rs = 4, 65536, 10001, 500
    n_form = N // n_ac
    base_c = np.array([[0, -20], [25, -40], [25, -80], [0, -100.0]])
    rng = np.random.default_rng(7)
    shift = rng.uniform(-300, 300, (n_form, 2)); shift[0] = 0.0; shift[1] = (128.0, -64.0)   # (exactly representable shift)
    centres = (base_c[None] + shift[:, None, :]).reshape(N, 2)
    X0 = np.tile([20, 30, -np.pi / 2, 0, 10.0], (N, 1)); X0[:, :2] += np.repeat(shift, n_ac, 0)
    out = ctx.gvf_run(ctx.dev(np.ascontiguousarray(X0.T)), ctx.dev(np.ascontiguousarray(centres.T)), ctx.dev(np.full(N, 60.0)),
                      n_ac, rows, 0.05, 15.0, rec_stride=rs, record=('X',))
    ctx.sync()
    X = out['X'].cpu().numpy()                        # [21][5][N]
    assert X.shape == (21, 5, N) and np.isfinite(X).all()
    # formation 0 against the oracle's loop on a prefix (the oracle runs 1000 steps in seconds)
    Xo, *_ = S.formation_gvf_run(base_c, 60.0, 15.0, X0[:n_ac], 1001, 0.05)
    for r in (1, 2):
        assert np.abs(X[r, :, :n_ac].T - Xo[r * rs]).max() < 1e-6, r
    # translation invariance: formation 1 is formation 0 shifted by an exactly representable vector
    d = X[:, :, n_ac:2 * n_ac] - X[:, :, :n_ac]
    assert np.abs(d[:, 0] - 128.0).max() < 1e-6 and np.abs(d[:, 1] + 64.0).max() < 1e-6 and np.abs(d[:, 2:]).max() < 1e-7
    # every formation converges to its circle (|p - c| -> r up to the DCF radius modulation) and to the commanded speed
    xf = out['X_final'].cpu().numpy()
    rad = np.hypot(xf[0] - centres[:, 0], xf[1] - centres[:, 1])
    assert np.abs(rad - 60.0).max() < 8.0 and np.abs(xf[4] - 15.0).max() < 1e-6
    assert np.array_equal(X[-1], xf)                  # row 10000 is the final state


def test_collocation_batch_16384_properties(ctx):
    """SURVEY 8 f-1 at batch scale: 16 384 perturbed copies of the reference's exp_14 (121 nodes, hard bounds) in one launch.
    Every problem ends CONVERGED or STALLED (infeasible end poses); the converged ones satisfy the collocation equations to 1e-8
    (recomputed on the host in the reference's form), hold every bound and end condition exactly and report the reference's
    cost() of their own node values; identical problems give identical answers; a seeded sample equals the oracle's solve."""
    import d2dhip
    from d2dhip import synth
    from oracle import nlp as ON
    B, N = 16384, 121
    rows, W0, h = synth.nlp_problems(B)
    rows[1] = rows[0]; W0[1] = W0[0]                                    # two identical problems
    W = ctx.dev(np.ascontiguousarray(W0))
    out = ctx.nlp_solve(ctx.dev(rows), W, h)
    ctx.sync()
    st, feas, cost = out['status'].cpu().numpy(), out['feas'].cpu().numpy(), out['cost'].cpu().numpy()
    Wh = W.cpu().numpy()                                               # (B, 5, N)
    assert set(np.unique(st)) <= {1, 4}, np.unique(st)
    ok = st == 1
    assert 0.93 < ok.mean() < 0.98                                     # (the rest: end poses more than 15 m/s x 12 s apart along any flyable path)
    x, y, psi, phi, v = (Wh[:, c, :] for c in range(5))
    c1 = (x[:, 1:] - x[:, :-1]) / h - v[:, 1:] * np.cos(psi[:, 1:])
    c2 = (y[:, 1:] - y[:, :-1]) / h - v[:, 1:] * np.sin(psi[:, 1:])
    c3 = (psi[:, 1:] - psi[:, :-1]) / h - 9.81 / v[:, 1:] * np.tan(phi[:, 1:])
    viol = np.maximum(np.abs(c1).max(1), np.maximum(np.abs(c2).max(1), np.abs(c3).max(1)))
    assert np.abs(viol - feas).max() <= 1e-12 * max(1.0, viol.max())   # the kernel's feasibility number is this one
    assert viol[ok].max() <= 1e-8 and viol[~ok].min() > 1e-6
    assert np.abs(phi).max() <= np.deg2rad(40.) and v.min() >= 9. and v.max() <= 15. and np.abs(x).max() <= 150. and np.abs(y).max() <= 150.
    for c, col in ((0, d2dhip.SC_X0), (1, d2dhip.SC_Y0), (2, d2dhip.SC_PSI0)):
        np.testing.assert_array_equal(Wh[:, c, 0], rows[:, col]); np.testing.assert_array_equal(Wh[:, c, -1], rows[:, col + (d2dhip.SC_X1 - d2dhip.SC_X0)])
    np.testing.assert_allclose(cost, ((v - 12.0) ** 2).sum(1) / N, rtol=1e-12)
    np.testing.assert_array_equal(Wh[0], Wh[1])
    for b in np.random.default_rng(5).choice(np.nonzero(ok)[0], 3, replace=False):
        pb = ON.problem_from_row(rows[b], N, h)
        Wo, info = ON.solve(pb, W0[b].T.copy())
        assert info['status'] == 1 and abs(info['cost'] - cost[b]) <= 1e-7 * cost[b], (b, info['cost'], cost[b])


@pytest.fixture(scope='module')
def plan_q(ctx):
    """the same plan with the default solver on the q-coordinate kernel (kernel='fused'): the time-sliced hand-out lives there"""
    import d2dhip
    p = d2dhip.FitPlan(ctx, S_, K, DUR, WREF, kernel='fused')
    yield p
    p.close()


@pytest.mark.parametrize('mode', ['minpack', 'fast'])
def test_time_sliced_handout_is_bit_identical(ctx, plan_q, mode):
    """d2d_fit_opts.slice > 0: fits that have run `slice` iterations while others wait go to the back of the device-wide ring and are
    resumed by whichever wavefront pops them (state through device-scope atomics, fetch-add tickets).  Scheduling only: every result
    bit for bit what the run-to-completion hand-out gives, every fit handled exactly once, on a batch that makes the ring work
    (more fits than wave slots) and on ragged / small ones."""
    import torch
    import d2dhip
    from d2dhip import synth
    plan = plan_q
    kw = {} if mode == 'minpack' else {'mode': d2dhip.MODE_FAST}
    for B, sl in ((4500, 8), (2049, 4), (300, 4), (1, 4)):
        dsc = ctx.dev(synth.synth_scenarios(B, seed=31, obj_scale=0.1, K=K))
        q0 = plan.init(dsc)
        qa, qb = q0.clone(), q0.clone()
        ca, ia, sa, sta = plan.solve(dsc, qa, max_iter=300, **kw)
        cb, ib, sb, stb = plan.solve(dsc, qb, max_iter=300, slice=sl, **kw)
        assert torch.equal(qa, qb) and torch.equal(ia, ib) and torch.equal(ca, cb) and torch.equal(sa, sb), (B, sl)
        assert (sa.cpu().numpy() != 0).all() and stb[3] >= sta[3]        # (a resumed fit re-evaluates its rows: more evaluations, same answers)
    # ... and across launches: the host's convergence poll (check_every) stops and restarts fits with the same answers
    dsc = ctx.dev(synth.synth_scenarios(3000, seed=32, obj_scale=0.1, K=K))
    q0 = plan.init(dsc)
    qa, qb = q0.clone(), q0.clone()
    ca, ia, sa, _ = plan.solve(dsc, qa, max_iter=300, **kw)
    plan.begin(3000)
    while plan.iterate(dsc, qb, 7, max_iter=300, slice=5, **kw) > 0:
        pass
    cb, ib, sb, _ = plan.finish(dsc, qb)
    assert torch.equal(qa, qb) and torch.equal(ia, ib) and torch.equal(ca, cb) and torch.equal(sa, sb)


def test_32768_long_horizon_fits_properties(ctx):
    """The reference's 50 Hz horizon at the per-GPU batch of configs[3]: 32 768 fits of 301 nodes on the long-horizon kernel (segment
    formulation).  Descent, the reported cost re-evaluated by the evaluation kernel (d2d_fit_eval in the segment formulation: K > 229),
    stationarity, a seeded sample against the oracle's cost, invariance under the hand-out order."""
    import d2dhip
    import torch
    from d2dhip import synth
    B, K2, t2 = 32768, 301, 30.0
    dur = synth.planner_timing(0, t2, 10)[2]
    p = d2dhip.FitPlan(ctx, S_, K2, dur, synth.default_wref(0.1, K2))
    try:
        assert p.kernel == 'long'
        sc = synth.synth_scenarios(B, seed=20241008, obj_scale=0.1, K=K2, dist_range=(250., 375.))
        dsc = ctx.dev(sc)
        q0 = p.init(dsc)
        c0, g0, _ = p.eval(dsc, q0, want_H=False)
        q = q0.clone()
        cost, iters, status, stats = p.solve(dsc, q, max_iter=300)
        st = status.cpu().numpy()
        conv = np.isin(st, (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED))
        assert conv.mean() >= 0.998 and not (st == d2dhip.ST_NONFINITE).any()
        c1, g1, _ = p.eval(dsc, q, want_H=False)
        c0h, c1h, ch = c0.cpu().numpy(), c1.cpu().numpy(), cost.cpu().numpy()
        assert (ch <= c0h * (1 + 1e-12)).all()
        assert np.abs(c1h - ch).max() <= 1e-11 * np.abs(ch).max()
        gn1 = g1.abs().max(1).values.cpu().numpy(); gn0 = g0.abs().max(1).values.cpu().numpy()
        assert (gn1[conv] <= 1e-6 * np.maximum(gn0[conv], 1e-3)).mean() >= 0.999
        ob = F.FitBasis.from_arrays(S_, K2, dur, *p.basis())
        qh = q.cpu().numpy()
        for i in np.random.default_rng(1).integers(0, B, 6):
            co = F.cost(ob, sc[i], qh[i])
            assert abs(ch[i] - co) <= 1e-10 * co
        p.order_from_iters(iters)
        q2 = q0.clone()
        cost2, iters2, status2, _ = p.solve(dsc, q2, max_iter=300)
        p.clear_order()
        assert torch.equal(q, q2) and torch.equal(iters, iters2) and torch.equal(cost, cost2) and torch.equal(status, status2)
    finally:
        p.close()
