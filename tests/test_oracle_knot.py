"""CPU: the knot-space statement of the fit (oracle/fit_knot.py -- what the round-5 kernel fit_lm_knot_kernel runs) against the
q-space statement it restates (oracle/fit.py) and against the scipy golden of the bench scenarios."""
import os

import numpy as np
import pytest

from oracle import fit as F, fit_knot as FK

K, S_ = 50, 6
DUR = F.planner_timing(0, 4.9, 10)[2]
SS = 0.1 / K
WREF = (0.02 ** 2, SS * 5.0, SS / F.G_ACC ** 2)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fit_scipy_bench1024.npz')


@pytest.fixture(scope='module')
def kb():
    return FK.KnotBasis(F.FitBasis(S_, K, DUR, WREF))


@pytest.fixture(scope='module')
def scen():
    return F.set_scale(F.synth_scenarios(24, seed=20241008), 0.1, K)


def test_knot_coordinates_are_the_same_curve(kb, scen):
    """q <-> u is an affine bijection, the Hermite tables reproduce the flat outputs of the dense basis, the metric is banded."""
    b = kb.basis
    rng = np.random.default_rng(0)
    for i in range(4):
        q = F.initial_guess(b, scen[i]) + rng.normal(0, 0.3, 2 * b.nq)
        u = kb.to_u(scen[i], q)
        assert np.abs(kb.to_q(scen[i], u) - q).max() <= 1e-12 * np.abs(q).max()
        Y1, Y2 = F.flat_outputs(b, scen[i], q), kb.flat_outputs(scen[i], u)
        assert np.abs(Y1 - Y2).max() <= 1e-10 * np.abs(Y1).max()
        # the end conditions sit in the knot vector as they are (scaled): position and velocity at both ends
        w = kb.full_knots(scen[i], u).reshape(S_ + 1, 2, 4)
        dx, dy = F.end_data(scen[i])
        assert np.allclose(w[0, 0, :2], [dx[0], dx[1] * kb.T]) and np.allclose(w[S_, 1, :2], [dy[2], dy[3] * kb.T])
        c, g, H = kb.eval_normal(scen[i], u)
        n = len(u)
        bw = max(abs(r - s) for r in range(n) for s in range(n) if abs(H[r, s]) > 1e-10 * np.abs(H).max())
        assert bw <= 15                                         # block tridiagonal in 8 x 8 blocks
    n = 2 * b.nq
    assert max(abs(r - s) for r in range(n) for s in range(n) if abs(kb.Mu[r, s]) > 1e-13 * np.abs(kb.Mu).max()) <= 11
    assert np.abs(kb.B.T @ kb.B - kb.Mu).max() <= 1e-12 * np.abs(kb.Mu).max()


def test_lmder_in_knot_coordinates_is_lmder_in_q(kb, scen):
    """MINPACK's lmder restated with the metric Mu takes the same trial points as lmder in q (unit scaling): same number of trials
    up to the hand-over, the same point there, on every scenario (fp64)."""
    b = kb.basis
    for i in range(12):
        q1, c1, n1, s1, i1 = F.lmder_solve(b, scen[i], finish=F.MP_FINISH, slow=F.MP_SLOW)
        u2, c2, n2, s2, i2 = FK.lmder_knot(kb, scen[i], finish=F.MP_FINISH, slow=F.MP_SLOW)
        assert n1 == n2 and i1['nfac'] == i2['nfac'] and i1['handover'] == i2['handover']
        assert abs(c1 - c2) <= 1e-9 * c1
        assert np.abs(kb.to_q(scen[i], u2) - q1).max() <= 1e-7 * np.abs(q1).max()


def test_default_solver_in_knot_coordinates_vs_q_and_vs_the_scipy_golden(kb):
    """lmder + second-order finish with the precision split of the kernel (fp32 Hessian, fp32 Cholesky): the same minimum as the q
    statement and as the exact minimiser of scipy's basin on the first bench scenarios."""
    import bench
    g = np.load(GOLD)
    sc = bench.bench_scenarios(4096)[:32]
    b = F.FitBasis(S_, K, *bench._plan_consts())
    kbb = FK.KnotBasis(b)
    near = 0
    for i in range(32):
        q1, c1, it1, st1, _ = F.solve_minpack(b, sc[i], hess_dtype=np.float32, chol_dtype=np.float32)
        q2, c2, it2, st2, _ = FK.solve_minpack_knot(kbb, sc[i], hess_dtype=np.float32, chol_dtype=np.float32)
        assert st1 == st2 == F.ST_CONVERGED
        assert abs(c1 - c2) <= 1e-6 * c1 and np.abs(q1 - q2).max() <= 1e-6 * np.abs(q1).max()
        z2 = F.coefficients(b, sc[i], q2).reshape(-1)
        zs = F.coefficients(b, sc[i], g['k50_qstar'][i]).reshape(-1)
        near += int(np.abs(z2 - zs).max() <= 1e-6 * np.abs(zs).max() and abs(c2 - g['k50_cstar'][i]) <= 1e-6 * g['k50_cstar'][i])
    assert near == 32
