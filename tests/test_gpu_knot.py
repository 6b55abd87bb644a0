"""GPU parity of the knot-coordinate LM kernel (csrc/fit_knot.hip, selected with D2D_FIT_KNOT=1 at plan creation; oracle:
oracle/fit_knot.py) -- the same default solve as fit_lm_kernel in the reference's own local parameterisation (knot data of
CompositeTraj([MinSnapPoly...]), src/d2d/trajectory.py:166-208), where J^T J is block tridiagonal:
  * every fit converges; the cost reported is the oracle's cost at the returned q; J^T r (returned in q) is the oracle's
  * the same minimum as the oracle's knot-space solver with the kernel's precision split, trial counts close
  * the same minimum as the q-coordinate kernel and as the exact minimiser of scipy's basin (golden) on the bench scenarios
  * a budgeted solve (launch after launch) is bit-identical to one launch (the knot vector is kept between launches)"""
import os

import numpy as np
import pytest

from oracle import fit as F, fit_knot as FK

pytestmark = pytest.mark.gpu
K, S_ = 50, 6
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fit_scipy_bench1024.npz')


@pytest.fixture(scope='module')
def ctx():
    import d2dhip
    c = d2dhip.Context(0)
    yield c
    c.close()


def _plan(ctx, knot):
    import bench
    import d2dhip
    return d2dhip.FitPlan(ctx, S_, K, *bench._plan_consts(), kernel='knot' if knot else 'fused')


def test_knot_kernel_vs_oracle_and_q_kernel_and_golden(ctx):
    import bench
    import d2dhip
    import torch
    g = np.load(GOLD)
    B = 256
    sc = bench.bench_scenarios(4096)[:B]
    dsc = ctx.dev(sc)
    pk, pq = _plan(ctx, True), _plan(ctx, False)
    try:
        q0 = pk.init(dsc)
        qk, qq = q0.clone(), q0.clone()
        ck, ik, sk, stk = pk.solve(dsc, qk, max_iter=150)
        cq, iq, sq, _ = pq.solve(dsc, qq, max_iter=150)
        assert (sk == d2dhip.ST_CONVERGED).all() and (sq == d2dhip.ST_CONVERGED).all()
        ckh, qkh, cqh, qqh = ck.cpu().numpy(), qk.cpu().numpy(), cq.cpu().numpy(), qq.cpu().numpy()
        # the same minimum as the q-coordinate kernel (paths differ at rounding level: another basin is allowed on 1 %)
        same = (np.abs(ckh - cqh) <= 1e-6 * cqh) & (np.abs(qkh - qqh).max(1) <= 1e-6 * np.abs(qqh).max(1))
        assert same.mean() >= 0.99, same.mean()
        # cost and J^T r at the returned point, by the evaluation kernel and by the oracle
        c1, g1, _ = pk.eval(dsc, qk, want_H=False)
        assert np.abs(c1.cpu().numpy() - ckh).max() <= 1e-10 * ckh.max()
        assert float(g1.abs().max().item()) <= 1e-6 and stk[1] <= 1e-6
        ob = F.FitBasis.from_arrays(S_, K, bench._plan_consts()[0], *pk.basis())
        kb = FK.KnotBasis(ob)
        near = 0
        dit = []
        for i in range(24):
            assert abs(F.cost(ob, sc[i], qkh[i]) - ckh[i]) <= 1e-10 * ckh[i]
            qo, co, ito, sto, _ = FK.solve_minpack_knot(kb, sc[i], hess_dtype=np.float32, chol_dtype=np.float32, max_iter=150)
            ok = abs(co - ckh[i]) <= 1e-6 * co and np.abs(qo - qkh[i]).max() <= 1e-6 * np.abs(qo).max()
            near += int(ok)
            if ok:
                dit.append(abs(int(ik[i].item()) - ito))
        assert near >= 23 and np.median(dit) <= 3, (near, dit)
        # the exact minimiser of scipy's basin (tests/golden/make_fit_scipy_golden.py)
        z = pk.coeffs(dsc, qk).cpu().numpy().reshape(B, -1)
        zs = np.array([F.coefficients(ob, sc[i], g['k50_qstar'][i]).reshape(-1) for i in range(B)])
        ex = (np.abs(z - zs).max(1) <= 1e-6 * np.abs(zs).max(1)) & (np.abs(ckh - g['k50_cstar'][:B]) <= 1e-6 * g['k50_cstar'][:B])
        assert ex.mean() >= 0.995, np.nonzero(~ex)[0]
        # budgeted launches: 7 trials per launch until nothing runs == one launch
        q2 = q0.clone()
        pk.begin(B)
        for _ in range(40):
            if pk.iterate(dsc, q2, 7, max_iter=150) == 0:
                break
        c2, i2, s2, _ = pk.finish(dsc, q2)
        assert torch.equal(q2, qk) and torch.equal(c2, ck) and torch.equal(i2, ik)
    finally:
        pk.close(); pq.close()


def test_knot_kernel_wind_bankmax_box_and_extra_obstacles(ctx):
    """the rows that reach the kernel through sample_terms' rarer branches: wind, CostBank max mode, a position box, a third obstacle"""
    import bench
    import d2dhip
    sc = bench.bench_scenarios(64).copy()
    sc[1::4, F.SC_WX], sc[1::4, F.SC_WY] = 1.0, -0.5
    sc[2::4, F.SC_BANKMAX] = 1.0
    sc[3::4, F.SC_XMIN], sc[3::4, F.SC_XMAX] = sc[3::4, F.SC_X0] - 40.0, sc[3::4, F.SC_X0] + 40.0
    sc[3::4, F.SC_O2X], sc[3::4, F.SC_O2Y], sc[3::4, F.SC_O2R] = sc[3::4, F.SC_X0] + 10.0, sc[3::4, F.SC_Y0] + 10.0, 6.0
    dsc = ctx.dev(sc)
    pk, pq = _plan(ctx, True), _plan(ctx, False)
    try:
        q0 = pk.init(dsc)
        qk, qq = q0.clone(), q0.clone()
        ck, ik, sk, _ = pk.solve(dsc, qk, max_iter=200)
        cq, iq, sq, _ = pq.solve(dsc, qq, max_iter=200)
        ckh, cqh = ck.cpu().numpy(), cq.cpu().numpy()
        assert np.isin(sk.cpu().numpy(), (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED)).all()
        ob = F.FitBasis.from_arrays(S_, K, bench._plan_consts()[0], *pk.basis())
        for i in range(16):
            assert abs(F.cost(ob, sc[i], qk.cpu().numpy()[i]) - ckh[i]) <= 1e-10 * ckh[i]
        # (bank-max rows, boxes and third obstacles make more local minima: the two kernels' finishes -- damping lam Mu here,
        # lam diag|H| in q -- part ways more often than on the plain bench scenarios; both end in stationary points)
        assert ((np.abs(ckh - cqh) <= 1e-6 * cqh).mean()) >= 0.8
        c1, g1, _ = pk.eval(dsc, qk, want_H=False)
        conv = sk.cpu().numpy() == d2dhip.ST_CONVERGED
        # (CostBank max mode keeps one phi row at argmax: the cost is only piecewise smooth there, its minimum can sit on a kink)
        smooth = conv & (sc[:, F.SC_BANKMAX] == 0)
        assert g1.abs().max(1).values.cpu().numpy()[smooth].max() <= 1e-5
    finally:
        pk.close(); pq.close()


@pytest.mark.parametrize('K2', [64, 40, 57])
def test_knot_kernel_other_sample_counts(ctx, K2):
    """K = 64 (eleven samples in the longest segment: the kernel's general instantiation), K = 40 and an odd K: the same minima as the
    q-coordinate kernel, the reported cost is the oracle's at the returned point."""
    import d2dhip
    from d2dhip import synth
    dur = synth.planner_timing(0, (K2 - 1) / 10.0, 10)[2]
    wref = synth.default_wref(0.1, K2)
    sc = synth.synth_scenarios(96, seed=5, obj_scale=0.1, K=K2, dist_range=(30. * dur / 4.9, 55. * dur / 4.9))
    dsc = ctx.dev(sc)
    pk, pq = d2dhip.FitPlan(ctx, S_, K2, dur, wref, kernel='knot'), d2dhip.FitPlan(ctx, S_, K2, dur, wref, kernel='fused')
    try:
        assert pk.kernel == 'knot' and pq.kernel == 'fused'
        q0 = pk.init(dsc)
        qk, qq = q0.clone(), q0.clone()
        ck, ik, sk, _ = pk.solve(dsc, qk)
        cq, iq, sq, _ = pq.solve(dsc, qq)
        assert np.isin(sk.cpu().numpy(), (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED)).all()
        ckh, cqh = ck.cpu().numpy(), cq.cpu().numpy()
        assert (np.abs(ckh - cqh) <= 1e-6 * cqh).mean() >= 0.97
        ob = F.FitBasis.from_arrays(S_, K2, dur, *pk.basis())
        for i in range(8):
            assert abs(F.cost(ob, sc[i], qk.cpu().numpy()[i]) - ckh[i]) <= 1e-10 * ckh[i]
    finally:
        pk.close(); pq.close()
