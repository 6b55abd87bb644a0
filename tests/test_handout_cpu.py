"""CPU: the host side of the predicted hand-out (d2dhip/handout.py -- the statement the device's key kernel is compared with in
tests/test_gpu_handout.py): features from plain geometry, the regression recovers a planted prior, keys are quantised as the kernel does."""
import numpy as np

from d2dhip import synth, handout

DUR = synth.planner_timing(0, 4.9, 10)[2]


def test_features_are_the_headings_against_the_tri_dogleg():
    """t0 / t1 = end headings minus the direction of the first / second leg of the 'tri' polyline (src/d2d/opty_utils.py:171-187),
    computed here from the polyline itself (oracle/fit.py triangle)."""
    from oracle import fit as F
    sc = synth.synth_scenarios(64, seed=3, obj_scale=0.1, K=50)
    t0, t1, x = handout.features(sc, DUR)
    for i in range(64):
        px, py = F.triangle((sc[i, F.SC_X0], sc[i, F.SC_Y0]), (sc[i, F.SC_X1], sc[i, F.SC_Y1]), sc[i, F.SC_VREF], DUR, 50, sc[i, F.SC_GOLEFT])
        a0 = np.arctan2(py[1] - py[0], px[1] - px[0]); a1 = np.arctan2(py[-1] - py[-2], px[-1] - px[-2])
        w = lambda v: (v + np.pi) % (2 * np.pi) - np.pi          # noqa: E731
        assert abs(w(t0[i] - w(sc[i, F.SC_PSI0] - a0))) < 1e-9 and abs(w(t1[i] - w(sc[i, F.SC_PSI1] - a1))) < 1e-9
        d = np.hypot(sc[i, F.SC_X1] - sc[i, F.SC_X0], sc[i, F.SC_Y1] - sc[i, F.SC_Y0])
        assert abs(x[i] - d / (12.0 * DUR)) < 1e-12


def test_regression_recovers_a_planted_additive_prior():
    rng = np.random.default_rng(0)
    sc = synth.synth_scenarios(60000, seed=11, obj_scale=0.1, K=50)
    A = rng.normal(0, 5, (handout.NB, handout.ND)); B = rng.normal(0, 5, (handout.NB, handout.ND))
    b0, b1, bx = handout.bins(sc, DUR)
    n = 30.0 + A[b0, bx] + B[b1, bx] + rng.normal(0, 3, len(sc))
    t = handout.fit_prior(sc, DUR, n)
    k, ki = handout.key(sc, DUR, t)
    assert np.corrcoef(k, 30.0 + A[b0, bx] + B[b1, bx])[0, 1] > 0.98
    assert ki.min() >= 0 and ki.max() < handout.ORDER_BINS
    assert np.array_equal(ki, np.clip((np.float32(8) * (k + np.float32(40))).astype(np.int64), 0, handout.ORDER_BINS - 1))
    # cells without data are neutral (the mean in A, zero in B): no NaN reaches the device
    assert np.isfinite(t).all()
    occupied = np.bincount(bx, minlength=handout.ND) > 0
    assert np.allclose(t[0][:, ~occupied], n.mean(), atol=1e-3) and np.allclose(t[1][:, ~occupied], 0.0)
