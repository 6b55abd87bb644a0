"""The predicted hand-out of the persistent fit kernels (d2d_fit_opts.handout = D2D_HANDOUT_PREDICTED, the default; include/d2d.h
d2d_fit_plan_set_handout_prior) and the ownership rules of a solve in parts.

The hand-out only schedules: whatever the prior says, every fit must end bit-identically to index order.  The device's keys are
checked against the host statement of the same features (d2dhip/handout.py), and the prior against the trial counts of THIS batch
(a batch the prior was not regressed on: tools/data/handout_calib.npz holds other seeds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, S_ = 50, 6


@pytest.fixture(scope='module')
def ctx():
    import d2dhip
    c = d2dhip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope='module')
def setup(ctx):
    import bench
    import d2dhip
    plan = d2dhip.FitPlan(ctx, S_, K, *bench._plan_consts())
    B = 4096
    sc = bench.bench_scenarios(B)
    dsc = ctx.dev(sc)
    yield plan, sc, dsc, B
    plan.close()


def _solve(plan, dsc, **kw):
    q = plan.init(dsc)
    cost, iters, status, stats = plan.solve(dsc, q, max_iter=150, **kw)
    return q.cpu().numpy(), cost.cpu().numpy(), iters.cpu().numpy(), status.cpu().numpy()


def test_predicted_handout_changes_nothing_but_the_order(ctx, setup):
    import d2dhip
    from d2dhip import handout
    from scipy.stats import spearmanr
    plan, sc, dsc, B = setup
    assert plan.kernel == 'knot'
    qi, ci, ii, si = _solve(plan, dsc, handout=d2dhip.HANDOUT_INDEX)
    with pytest.raises(d2dhip.D2DError):          # index order: there is no order to read back
        plan.last_order(B)
    qp, cp, ip, sp = _solve(plan, dsc)            # the default: predicted
    order = plan.last_order(B)
    for a, b in ((qi, qp), (ci, cp), (ii, ip), (si, sp)):
        assert np.array_equal(a, b)
    # the order is a permutation, sorted by the host statement of the device's key (bins at an edge may fall either way)
    assert np.array_equal(np.sort(order), np.arange(B))
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'drone-sim-python_amd', 'csrc', 'fit_handout_prior.h')).read()
    table = np.array([float(v) for v in re.findall(r'(-?[0-9]+\.[0-9]+)f', hdr)], dtype=np.float32).reshape(2, handout.NB, handout.ND)
    k, ki = handout.key(sc, plan.duration, table)
    along = ki[order]
    assert (np.diff(along) > 0).mean() <= 2e-3, (np.diff(along) > 0).sum()
    # ... and the prior knows something about THIS batch (not one it was regressed on): rank correlation with the measured trial
    # counts, and most of the longest fits are handed out in the first half
    assert spearmanr(k, ii).statistic >= 0.5
    longest = np.argsort(-ii, kind='stable')[:B // 20]
    pos = np.empty(B, np.int64); pos[order] = np.arange(B)
    assert (pos[longest] < B // 2).mean() >= 0.85


def test_a_batch_that_fits_the_wave_slots_is_not_ordered(ctx, setup):
    import d2dhip
    plan, sc, dsc, B = setup
    _solve(plan, dsc[:1024].contiguous())
    with pytest.raises(d2dhip.D2DError):
        plan.last_order(1024)


def test_custom_prior_and_explicit_hint(ctx, setup):
    """A caller's prior replaces the built-in one (here: one that ranks by chord length alone), None restores it; an explicit
    d2d_fit_plan_set_order hint overrides the prediction.  Results stay bit-identical throughout."""
    import torch
    from d2dhip import handout
    plan, sc, dsc, B = setup
    ref = _solve(plan, dsc)
    order_builtin = plan.last_order(B)
    t = np.zeros((2, handout.NB, handout.ND), np.float32)
    t[0] += np.arange(handout.ND, dtype=np.float32)[None, :] * 5.0
    plan.set_handout_prior(t)
    try:
        got = _solve(plan, dsc)
        order = plan.last_order(B)
        bx = handout.bins(sc, plan.duration)[2]
        assert (np.diff(bx[order]) > 0).mean() <= 2e-3          # longest chord bins first
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
    finally:
        plan.set_handout_prior(None)
    _solve(plan, dsc)
    assert np.array_equal(np.sort(plan.last_order(B)), np.arange(B))
    # learn_handout_prior: regress on this solve's own counts, install, solve again
    plan.learn_handout_prior(dsc, torch.from_numpy(ref[2]))
    try:
        got = _solve(plan, dsc)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
    finally:
        plan.set_handout_prior(None)
    # explicit hint: exactly the counting sort of the given counts
    plan.order_from_iters(ctx.dev(ref[2].astype(np.int32)))
    try:
        got = _solve(plan, dsc)
        o = plan.last_order(B)
        assert (np.diff(ref[2][o]) > 0).sum() == 0
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
    finally:
        plan.clear_order()
    assert not np.array_equal(order_builtin, o)


def test_a_solve_in_parts_owns_its_q_and_its_kernel(ctx, setup):
    """the knot kernel keeps the unknowns of the running fits in its own coordinates between launches.  Another q buffer,
    or options that would move the solve to another kernel mid-way (FAST mode / a time slice run on the q kernel), are refused
    instead of silently resuming from stale state; a fresh d2d_fit_begin accepts them."""
    import d2dhip
    plan, sc, dsc, B = setup
    d = dsc[:512].contiguous()
    q = plan.init(d)
    plan.begin(512)
    n = plan.iterate(d, q, 6, max_iter=150)
    assert n > 0
    q2 = q.clone()
    with pytest.raises(d2dhip.D2DError, match='owns q'):
        plan.iterate(d, q2, 6, max_iter=150)
    with pytest.raises(d2dhip.D2DError, match='another kernel'):
        plan.iterate(d, q, 6, max_iter=150, mode=d2dhip.MODE_FAST)
    with pytest.raises(d2dhip.D2DError, match='another kernel'):
        plan.iterate(d, q, 6, max_iter=150, slice=4)
    with pytest.raises(d2dhip.D2DError, match='differs from the buffer'):
        plan.finish(d, q2)
    while plan.iterate(d, q, 50, max_iter=150) > 0:
        pass
    cost, iters, status, _ = plan.finish(d, q)
    # the same fits in one piece: bit-identical (the budgeted launches resumed exactly)
    qa = plan.init(d)
    ca, ia, sa, _ = plan.solve(d, qa, max_iter=150)
    assert np.array_equal(q.cpu().numpy(), qa.cpu().numpy()) and np.array_equal(cost.cpu().numpy(), ca.cpu().numpy())
    assert np.array_equal(iters.cpu().numpy(), ia.cpu().numpy())
    # a fresh solve may use the other kernel (FAST mode of a knot plan runs on the q kernel)
    qf = plan.init(d)
    cf, itf, sf, _ = plan.solve(d, qf, max_iter=150, mode=d2dhip.MODE_FAST)
    assert np.isin(sf.cpu().numpy(), (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED)).mean() >= 0.99


def test_plan_kernel_is_an_argument(ctx):
    import bench
    import d2dhip
    dur, wref = bench._plan_consts()
    for kern, want in (('auto', 'knot'), ('knot', 'knot'), ('fused', 'fused'), ('long', 'long'), ('split', 'split')):
        p = d2dhip.FitPlan(ctx, S_, K, dur, wref, kernel=kern)
        assert p.kernel == want
        p.close()
    with pytest.raises(d2dhip.D2DError):          # the fused / knot kernels serve K <= 64 only
        d2dhip.FitPlan(ctx, S_, 121, 12.0, wref, kernel='knot')
    with pytest.raises(d2dhip.D2DError):          # ... and S = 6
        d2dhip.FitPlan(ctx, 4, K, dur, wref, kernel='fused')
