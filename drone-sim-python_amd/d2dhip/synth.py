"""Synthetic fit scenarios of SURVEY.md 8d ("synthetic inputs") and the default plan
constants -- host-side input generation for benchmarks and batch planning (numpy only;
no numerics of the hot path live here)."""
import numpy as np

from . import (SC_WX, SC_WY, SC_BANKMAX, obs_col,
               SCEN_STRIDE, SC_X0, SC_Y0, SC_PSI0, SC_X1, SC_Y1, SC_PSI1, SC_VREF, SC_VSP, SC_KV, SC_KPHI,
               SC_KOBS, SC_S, SC_WWP, SC_GOLEFT, SC_O0X, SC_O0Y, SC_O0R, SC_O1X, SC_O1Y, SC_O1R, SC_WBND,
               SC_PHIMAX, SC_VMIN, SC_VMAX, SC_KCOL, SC_RCOL, SC_SCOL, SC_PMASK, SC_XMIN, SC_XMAX, SC_YMIN, SC_YMAX)

G_ACC = 9.81


def planner_timing(t0, t1, hz):
    """Node count / step / rounded duration, as the reference's planner_timing
    (src/d2d/opty_utils.py:8-14) computes them."""
    num_nodes = int((t1 - t0) * hz) + 1
    time_step = 1.0 / hz
    return num_nodes, time_step, (num_nodes - 1) * time_step


def default_wref(obj_scale, K, kv=5.0, kphi=1.0, wwp=0.02):
    """Weights of the whitening metric: the quadratic skeleton of the cost."""
    s = obj_scale / K
    return (wwp ** 2, s * kv, s * kphi / G_ACC ** 2)


def synth_scenarios(B, seed=20241008, rank=0, n_obs=2, wbnd=1.0, wwp=0.02, obj_scale=0.1, K=50, dist_range=(30., 55.)):
    """B scenario rows (float64 [B][SCEN_STRIDE]): start/end poses dist_range (30-55 m) apart with random
    headings, vref = vsp = 12, kv = 5, kphi = 1 (src/multi_opt_planner.py:187-188), two
    circular obstacles beside the straight line, soft bounds phi in +-40 deg, v in [9,15]."""
    rng = np.random.default_rng(seed + rank)
    sc = np.zeros((B, SCEN_STRIDE))
    p0 = rng.uniform(-100, 100, (B, 2)); psi0 = rng.uniform(-np.pi, np.pi, B)
    dist = rng.uniform(dist_range[0], dist_range[1], B); beta = rng.uniform(-np.pi, np.pi, B)
    p1 = p0 + dist[:, None] * np.stack([np.cos(beta), np.sin(beta)], 1)
    psi1 = rng.uniform(-np.pi, np.pi, B)
    sc[:, SC_X0], sc[:, SC_Y0], sc[:, SC_PSI0] = p0[:, 0], p0[:, 1], psi0
    sc[:, SC_X1], sc[:, SC_Y1], sc[:, SC_PSI1] = p1[:, 0], p1[:, 1], psi1
    sc[:, SC_VREF] = 12.0; sc[:, SC_VSP] = 12.0
    sc[:, SC_KV] = 5.0; sc[:, SC_KPHI] = 1.0; sc[:, SC_KOBS] = 1.0
    sc[:, SC_WWP] = wwp; sc[:, SC_GOLEFT] = -1.0; sc[:, SC_WBND] = wbnd
    sc[:, SC_PHIMAX] = np.deg2rad(40.0); sc[:, SC_VMIN] = 9.0; sc[:, SC_VMAX] = 15.0
    along = rng.uniform(0.2, 0.8, (B, 2)); lat = rng.uniform(5, 15, (B, 2)) * rng.choice([-1.0, 1.0], (B, 2))
    rad = rng.uniform(5, 15, (B, 2))
    u = (p1 - p0) / dist[:, None]; nrm = np.stack([-u[:, 1], u[:, 0]], 1)
    for i, (ox, oy, orr) in enumerate(((SC_O0X, SC_O0Y, SC_O0R), (SC_O1X, SC_O1Y, SC_O1R))):
        c = p0 + along[:, i, None] * (p1 - p0) + lat[:, i, None] * nrm
        sc[:, ox], sc[:, oy] = c[:, 0], c[:, 1]
        sc[:, orr] = rad[:, i] if i < n_obs else 0.0
    sc[:, SC_S] = obj_scale / K
    return sc


VARIANTS = ('wind', 'bankmax', 'box3')


def variant_scenarios(kind, B, seed=20241008, rank=0, obj_scale=0.1, K=50):
    """The rarer rows of the cost catalogue on bench-style scenarios -- the branches of the sample evaluation the plain
    bench batch never takes:  'wind'  a constant wind (1.0, -0.5) m/s (exp_0_2, src/d2d/optyplan_scenarios.py:44-53);
    'bankmax'  CostBank(use_mean=False) (src/d2d/opty_utils.py:72-73);  'box3'  an x box of +-40 m around the start (x_constraint,
    src/single_opt_planner.py:56) and a third obstacle of radius 6 at (+10, +10) from it (CostObstacles, :136-144)."""
    sc = synth_scenarios(B, seed=seed, rank=rank, obj_scale=obj_scale, K=K)
    if kind == 'wind':
        sc[:, SC_WX], sc[:, SC_WY] = 1.0, -0.5
    elif kind == 'bankmax':
        sc[:, SC_BANKMAX] = 1.0
    elif kind == 'box3':
        sc[:, SC_XMIN], sc[:, SC_XMAX] = sc[:, SC_X0] - 40.0, sc[:, SC_X0] + 40.0
        c = obs_col(2)
        sc[:, c], sc[:, c + 1], sc[:, c + 2] = sc[:, SC_X0] + 10.0, sc[:, SC_Y0] + 10.0, 6.0
    else:
        raise ValueError(kind)
    return sc


def circle_group_scenarios(n_ac, R, duration, K=50, seed=0, sigma=2.0, pair01_only=False, obj_scale=1.0):
    """BASELINE configs[2]: R replicas of n_ac aircraft on a circle of radius t1*vref/2/1.5 heading to their
    antipodes (generalises exp_2, src/07_multioptyplan.py:251-260), collision rows kcol = rcol = 10 (:403);
    replicas differ by a sigma-metre perturbation of the start / end points.  Rows [R][n_ac][32]: the
    aircraft of one scenario are consecutive, as d2d_fit_solve_groups expects."""
    rng = np.random.default_rng(seed)
    rad = duration * 12.0 / 2 / 1.5
    s = obj_scale / K
    sc = np.zeros((R, n_ac, SCEN_STRIDE))
    for i in range(n_ac):
        a = 2 * np.pi * i / n_ac
        sc[:, i, SC_X0], sc[:, i, SC_Y0], sc[:, i, SC_PSI0] = rad * np.cos(a), rad * np.sin(a), a + np.pi
        sc[:, i, SC_X1], sc[:, i, SC_Y1], sc[:, i, SC_PSI1] = -rad * np.cos(a), -rad * np.sin(a), a + np.pi
    sc[..., [SC_X0, SC_Y0, SC_X1, SC_Y1]] += rng.normal(0, sigma, (R, n_ac, 4))
    sc[..., SC_VREF] = 12; sc[..., SC_VSP] = 12; sc[..., SC_KV] = 5; sc[..., SC_KPHI] = 1
    sc[..., SC_S] = s / n_ac; sc[..., SC_WWP] = 0.02; sc[..., SC_GOLEFT] = -1; sc[..., SC_WBND] = 1
    sc[..., SC_PHIMAX] = np.deg2rad(40.0); sc[..., SC_VMIN] = 9; sc[..., SC_VMAX] = 15
    sc[..., SC_KCOL] = 10; sc[..., SC_RCOL] = 10; sc[..., SC_SCOL] = s
    if pair01_only:
        sc[:, 0, SC_PMASK] = 0b10; sc[:, 1, SC_PMASK] = 0b01
    else:
        sc[..., SC_PMASK] = (1 << n_ac) - 1
    return sc


def nlp_problems(B, N=121, seed=0):
    """B perturbed copies of the reference's exp_14 (src/d2d/optyplan_scenarios.py:219-253: 121 nodes, 12 s, CostAirVel(12), phi in
    +-40 deg, v in [9, 15], boxes +-150) for the collocation backend: end poses moved by N(0, [3 m, 3 m, 0.1 rad]), the
    reference's 'tri' initial guess.  -> scenario rows [B][SCEN_STRIDE], node values W0 [B][5][N], time step."""
    from d2d.opty_utils import triangle
    rng = np.random.default_rng(seed)
    rows = np.zeros((B, SCEN_STRIDE)); W = np.zeros((B, 5, N))
    for b in range(B):
        p0 = np.array([-49.98, -58.14, 2.22]) + rng.normal(0, [3., 3., 0.1]); p1 = np.array([75., 40., 0.]) + rng.normal(0, [3., 3., 0.1])
        r = rows[b]
        r[SC_X0:SC_X0 + 3] = p0; r[SC_X1:SC_X1 + 3] = p1
        r[SC_VSP], r[SC_KV], r[SC_KPHI], r[SC_S] = 12., 1., 0., 1. / N
        r[SC_PHIMAX] = np.deg2rad(40.); r[SC_VMIN], r[SC_VMAX] = 9., 15.
        r[SC_XMIN], r[SC_XMAX], r[SC_YMIN], r[SC_YMAX] = -150, 150, -150, 150
        W[b] = np.stack(triangle(p0[:2], p1[:2], 12., 12.0, N, go_left=-1.), 0)
    return rows, W, 12.0 / (N - 1)
