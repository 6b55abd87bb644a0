"""Host side of the predicted hand-out (include/d2d.h d2d_fit_plan_set_handout_prior): the features of a scenario row exactly as
fit_handout_key_kernel (csrc/fit_kernels.hip) forms them, the regression that turns measured trial counts into a prior table, and
the sort key -- so that a user can calibrate the prior on solves of their own workload (FitPlan.learn_handout_prior) and tests
can check the device's order.  Scheduling only: no result of a solve depends on anything here."""
import numpy as np

NB, ND = 48, 12                 # D2D_HANDOUT_NB, D2D_HANDOUT_ND
X_LO, X_HI = 0.4, 1.0           # D2D_HANDOUT_X_LO, D2D_HANDOUT_X_HI
ORDER_BINS = 2048               # csrc/fit_kernels.hip ORDER_BINS
SC_X0, SC_Y0, SC_PSI0, SC_X1, SC_Y1, SC_PSI1, SC_VREF = range(7)
SC_GOLEFT = 15


def features(sc, duration):
    """(t0, t1, x) per scenario row: the end headings against the legs of the 'tri' dog-leg (src/d2d/opty_utils.py:171-187), wrapped
    to [-pi, pi), and the chord length over vref * duration."""
    sc = np.atleast_2d(np.asarray(sc, dtype=np.float64))
    dx = sc[:, SC_X1] - sc[:, SC_X0]; dy = sc[:, SC_Y1] - sc[:, SC_Y0]
    d = np.sqrt(dx * dx + dy * dy); D = sc[:, SC_VREF] * duration
    h = 0.5 * np.sqrt(np.maximum(D * D - d * d, 0.0))
    beta = np.arctan2(dy, dx); a = np.arctan2(np.sign(sc[:, SC_GOLEFT]) * h, 0.5 * d)
    wrap = lambda v: np.mod(v + np.pi, 2 * np.pi) - np.pi
    return wrap(sc[:, SC_PSI0] - (beta + a)), wrap(sc[:, SC_PSI1] - (beta - a)), np.where(D > 0, d / np.where(D > 0, D, 1.0), 0.0)


def bins(sc, duration):
    t0, t1, x = features(sc, duration)
    ab = lambda t: np.clip(((t + np.pi) * (NB / (2 * np.pi))).astype(np.int64), 0, NB - 1)
    bx = np.clip(np.floor((x - X_LO) * (ND / (X_HI - X_LO))).astype(np.int64), 0, ND - 1)
    return ab(t0), ab(t1), bx


def key(sc, duration, table):
    """The prior's trial count per row (float) and the integer sort key of the device (1/8 trial per bin, offset 40, clamped)."""
    b0, b1, bx = bins(sc, duration)
    table = np.asarray(table, dtype=np.float32).reshape(2, NB, ND)
    k = table[0, b0, bx] + table[1, b1, bx]
    k = np.where(np.isfinite(k), k, np.float32(0))
    ki = np.clip((np.float32(8) * (k + np.float32(40))).astype(np.int64), 0, ORDER_BINS - 1)
    return k, ki


def fit_prior(sc, duration, iters, sweeps=12, quantile=None, tail=None, min_count=20):
    """Additive model iters ~ mean + A[bin(t0), bin(x)] + B[bin(t1), bin(x)] by backfitting (a cell without data is neutral: 0).
    Returns table float32 [2][NB][ND] with the mean folded into A, so that the key is an expected trial count.
    quantile = q in (0, 1): instead, A and B hold the q-quantile of the counts of their cell (a RISK-aware key: what ends a launch is
    a long fit that starts late, so a cell with a heavy tail goes first whatever its mean); tail = T: 100 x the fraction of the cell's
    fits with >= T trials.  Cells with fewer than min_count samples take the value of all samples."""
    n = np.asarray(iters, dtype=np.float64)
    b0, b1, bx = bins(sc, duration)
    c0, c1 = b0 * ND + bx, b1 * ND + bx
    if quantile is not None or tail is not None:
        stat = (lambda v: np.percentile(v, 100.0 * quantile)) if quantile is not None else (lambda v: 100.0 * np.mean(v >= tail))
        out = np.empty((2, NB * ND))
        for t, c in enumerate((c0, c1)):
            order = np.argsort(c, kind='stable'); cs = c[order]; ns = n[order]
            edges = np.searchsorted(cs, np.arange(NB * ND + 1))
            glob = stat(n)
            for cell in range(NB * ND):
                lo, hi = edges[cell], edges[cell + 1]
                out[t, cell] = stat(ns[lo:hi]) if hi - lo >= min_count else glob
        return (0.5 * out).reshape(2, NB, ND).astype(np.float32)      # (halved: the key is the sum of two such entries)
    mu = n.mean()
    A = np.zeros(NB * ND); Bt = np.zeros(NB * ND)
    n0 = np.bincount(c0, minlength=NB * ND); n1 = np.bincount(c1, minlength=NB * ND)
    for _ in range(sweeps):
        A = np.bincount(c0, weights=n - mu - Bt[c1], minlength=NB * ND) / np.maximum(n0, 1)
        Bt = np.bincount(c1, weights=n - mu - A[c0], minlength=NB * ND) / np.maximum(n1, 1)
    A = A + mu
    return np.stack([A, Bt]).reshape(2, NB, ND).astype(np.float32)
