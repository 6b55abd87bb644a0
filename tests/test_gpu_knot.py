"""GPU parity of the knot-coordinate LM kernel (csrc/fit_knot.hip, the default of S = 6, K <= 64 plans; oracle:
oracle/fit_knot.py) -- the same default solve as fit_lm_kernel in the reference's own local parameterisation (knot data of
CompositeTraj([MinSnapPoly...]), src/d2d/trajectory.py:166-208), where J^T J is block tridiagonal:
  * every fit converges; the cost reported is the oracle's cost at the returned q; J^T r (returned in q) is the oracle's
  * the same minimum as the oracle's knot-space solver with the kernel's precision split, trial counts close
  * the same minimum as the q-coordinate kernel and as the exact minimiser of scipy's basin (golden) on the bench scenarios
  * a budgeted solve (launch after launch) is bit-identical to one launch (the knot vector is kept between launches)
  * the rarer branches of the sample evaluation (wind, CostBank max mode, a position box, a third obstacle) and the other sample counts
    (K = 64: the general instantiation; 40; 57) are pinned to the SAME two references as the plain scenarios: oracle/fit_knot.py
    solve_minpack_knot fit by fit (cost, unknowns, trial counts) and the scipy golden of tests/golden/fit_scipy_variants.npz, every fit
    that ends elsewhere than scipy certified as a stationary point of the arbiter -- no comparison whose only reference is another
    HIP kernel"""
import os

import numpy as np
import pytest

from oracle import fit as F, fit_knot as FK

pytestmark = pytest.mark.gpu
K, S_ = 50, 6
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fit_scipy_bench1024.npz')
GOLD_VAR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fit_scipy_variants.npz')
TOL = 1e-6


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


def _scipy_from(ob, sc, q0):
    """the CPU arbiter started from q0 (tests/test_gpu_parity_scipy.py): cost and unknowns where it stops"""
    from scipy.optimize import least_squares
    wp = F.waypoints(sc, ob.K, ob.duration)
    fun = lambda qq: F.residuals(ob, sc, qq, wp).reshape(-1)                        # noqa: E731
    jac = lambda qq: F.jacobian(ob, F.residuals(ob, sc, qq, wp, True)[1])           # noqa: E731
    r = least_squares(fun, q0, jac=jac, method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return 2 * r.cost, r.x


def _pin_family(ctx, name, K2, dur, wref, sc, n_oracle, near_min, smooth=True):
    """One scenario family through the knot kernel, against (i) oracle/fit_knot.py fit by fit, (ii) the scipy golden `name`_*.
    Returns what the caller still wants to look at."""
    import hashlib
    import d2dhip
    g = np.load(GOLD_VAR)
    assert hashlib.sha256(np.ascontiguousarray(sc, dtype=np.float64).tobytes()).hexdigest() == str(g[name + '_scen_sha256'])
    n = len(sc)
    dsc = ctx.dev(sc)
    plan = d2dhip.FitPlan(ctx, S_, K2, dur, wref, kernel='knot')
    try:
        assert plan.kernel == 'knot'
        q = plan.init(dsc)
        cost, iters, status, stats = plan.solve(dsc, q, max_iter=200)
        z = plan.coeffs(dsc, q).cpu().numpy().reshape(n, -1)
        c1, g1, _ = plan.eval(dsc, q, want_H=False)
        ob = F.FitBasis.from_arrays(S_, K2, dur, *plan.basis())
    finally:
        plan.close()
    c, qh, it, st = cost.cpu().numpy(), q.cpu().numpy(), iters.cpu().numpy(), status.cpu().numpy()
    assert np.isin(st, (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED)).all()
    assert np.abs(c1.cpu().numpy() - c).max() <= 1e-10 * c.max()                    # the cost reported = a second kernel's at the returned q
    # (i) the kernel's CPU statement, same precision split: cost and unknowns within 1e-6 on all but at most one, trial counts close
    # (CostBank's max mode: where the minimum sits on a kink of the arg-max row -- `star_ok` False in the golden -- the stopping point
    # depends on rounding: the oracle with an fp64 Hessian and the oracle with the kernel's fp32 one agree on ALL smooth minima and on
    # NO kink, stopping 2e-7 .. 5e-3 apart in cost (the kernel: up to 1.4e-2); there the kernel is held to the oracle's cost within
    # 5e-2 -- and, by the caller, to a cost not above the arbiter's -- not to its point)
    kb = FK.KnotBasis(ob)
    near_o, n_smooth, dit = 0, 0, []
    for i in range(n_oracle):
        assert abs(F.cost(ob, sc[i], qh[i]) - c[i]) <= 1e-10 * c[i]
        qo, co, ito, sto, _ = FK.solve_minpack_knot(kb, sc[i], hess_dtype=np.float32, chol_dtype=np.float32, max_iter=200)
        if not smooth and not g[name + '_star_ok'][i]:
            assert abs(co - c[i]) <= 5e-2 * co, (name, i, c[i], co)
            continue
        n_smooth += 1
        ok = abs(co - c[i]) <= TOL * co and np.abs(qo - qh[i]).max() <= TOL * np.abs(qo).max()
        near_o += int(ok)
        if ok:
            dit.append(abs(int(it[i]) - ito))
    assert n_smooth >= n_oracle // 2 and near_o >= n_smooth - 1 and np.median(dit) <= 3, (name, near_o, n_smooth, dit)
    # (ii) the arbiter: within 1e-6 (cost and the 96 monomial coefficients) of the exact minimiser of the basin scipy stopped in ...
    zstar = np.array([F.coefficients(ob, sc[i], g[name + '_qstar'][i]).reshape(-1) for i in range(n)])
    near = (np.abs(z - zstar).max(1) <= TOL * np.abs(zstar).max(1)) & (np.abs(c - g[name + '_cstar']) <= TOL * g[name + '_cstar'])
    ok_star = g[name + '_star_ok']
    if smooth:
        assert ok_star.all()
        assert near.mean() >= near_min, (name, near.mean(), np.nonzero(~near)[0])
        # ... and every fit that is not must be ANOTHER stationary point: the arbiter started from the kernel's point stays there
        for i in np.nonzero(~near)[0]:
            c3, q3 = _scipy_from(ob, sc[i], qh[i])
            assert abs(c3 - c[i]) <= TOL * c[i] and np.abs(q3 - qh[i]).max() <= TOL * np.abs(qh[i]).max(), (name, i, c[i], c3, g[name + '_cost'][i])
        conv = st == d2dhip.ST_CONVERGED
        assert g1.abs().max(1).values.cpu().numpy()[conv].max() <= 1e-5
    return c, near, ok_star, g


@pytest.mark.parametrize('kind,near_min', [('wind', 0.99), ('box3', 0.995)])
def test_knot_kernel_wind_box_and_third_obstacle_vs_oracle_and_scipy(ctx, kind, near_min):
    """the rows that reach the kernel through sample_terms' rarer branches (src/d2d/opty_utils.py:68-82, 99-134; wind:
    src/d2d/optyplan_scenarios.py:44-53): 256 scenarios each, 32 of them also through the oracle's solver"""
    import bench
    from d2dhip import synth
    dur, wref = bench._plan_consts()
    _pin_family(ctx, kind, K, dur, wref, synth.variant_scenarios(kind, 256, K=K), 32, near_min)


def test_knot_kernel_costbank_max_mode_vs_oracle_and_scipy(ctx):
    """CostBank(use_mean=False) keeps ONE phi row, at the arg-max sample: the cost is only piecewise smooth and its minimum may sit on a
    kink, where scipy's lm stops early (up to 0.27 of gradient left in the golden) -- the kernel is pinned to its oracle fit by fit like
    every family; against the arbiter: on the scenarios whose minimum is smooth (`star_ok`) the same point within 1e-6 on >= 0.97, and
    on all of them a cost that is not higher than the arbiter's on >= 0.95 (lower on a third)."""
    import bench
    from d2dhip import synth
    dur, wref = bench._plan_consts()
    c, near, ok_star, g = _pin_family(ctx, 'bankmax', K, dur, wref, synth.variant_scenarios('bankmax', 256, K=K), 32, 0.0, smooth=False)
    assert 0.5 <= ok_star.mean() < 1.0
    assert near[ok_star].mean() >= 0.97, near[ok_star].mean()
    assert (c <= g['bankmax_cost'] * (1 + TOL)).mean() >= 0.95


@pytest.mark.parametrize('K2,near_min', [(64, 0.99), (40, 0.99), (57, 0.99)])
def test_knot_kernel_other_sample_counts_vs_oracle_and_scipy(ctx, K2, near_min):
    """K = 64 (eleven samples in the longest segment: the kernel's general instantiation), K = 40 and an odd K: 128 scenarios each,
    24 of them also through the oracle's solver"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_fit_scipy_variants_golden as MV
    dur, wref, sc = MV.other_k_scenarios(K2, 128)
    _pin_family(ctx, f'k{K2}', K2, dur, wref, sc, 24, near_min)
