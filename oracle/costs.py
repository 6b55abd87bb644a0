"""ORACLE (test infrastructure only) -- numpy restatement of the planner cost plug-ins,
free-vector layouts and initial guesses of the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Functional (array in / array out) form; every function cites the lines it follows
(relative to /root/reference).  Pinned by tests/golden/costs.npz, guess_poly.npz and
planner_goldens.npz.
"""
import numpy as np


# ---- free-vector layouts -----------------------------------------------------------
def single_slices(N):
    """src/single_opt_planner.py:35-39 -> dict of slices x,y,psi,phi,v."""
    return {k: slice(i * N, (i + 1) * N) for i, k in enumerate(('x', 'y', 'psi', 'phi', 'v'))}


def multi_slices(N, n):
    """src/multi_opt_planner.py:41-47 -> dict of lists of slices."""
    s = {'x': [slice((0 + 3 * i) * N, (1 + 3 * i) * N) for i in range(n)],
         'y': [slice((1 + 3 * i) * N, (2 + 3 * i) * N) for i in range(n)],
         'psi': [slice((2 + 3 * i) * N, (3 + 3 * i) * N) for i in range(n)]}
    o = 3 * n * N
    s['phi'] = [slice(o + i * N, o + (i + 1) * N) for i in range(n)]
    o += n * N
    s['v'] = [slice(o + i * N, o + (i + 1) * N) for i in range(n)]
    return s


# ---- single-aircraft costs (src/d2d/opty_utils.py) ----------------------------------
def airvel(free, N, scale, vsp):
    """:55-66."""
    sl = single_slices(N)
    dv = free[sl['v']] - vsp
    g = np.zeros_like(free); g[sl['v']] = scale / N * 2 * dv
    return scale * np.sum(dv ** 2) / N, g


def bank(free, N, scale, use_mean=True):
    """:68-82."""
    sl = single_slices(N)
    ph = free[sl['phi']]
    g = np.zeros_like(free)
    if use_mean:
        g[sl['phi']] = scale / N * 2 * ph
        return scale * np.sum(ph ** 2) / N, g
    i = int(np.argmax(ph ** 2))
    g[sl['phi'].start + i] = scale * 2 * ph[i]
    return scale * np.max(ph ** 2), g


def cost_input(free, N, scale, vsp, kv, kphi):
    """:85-97."""
    sl = single_slices(N)
    g = np.zeros_like(free)
    g[sl['phi']] = kphi * 2 * free[sl['phi']]
    g[sl['v']] = kv * 2 * (free[sl['v']] - vsp)
    g *= scale / N
    c = scale / N * (kv * np.sum((free[sl['v']] - vsp) ** 2) + kphi * np.sum(free[sl['phi']] ** 2))
    return c, g


def _obst_e(dx, dy, r, kind, k=2.0):
    if kind == 0:
        return np.clip(np.exp(r ** 2 - (dx ** 2 + dy ** 2)), 0.0, 1e3)
    return np.exp(-((dx / r * k) ** 2 + (dy / r * k) ** 2))


def obstacle(free, N, scale, c, r, kind, sx=None, sy=None):
    """:99-134 (gradient quirks reproduced: no (k/r)^2 factor, clip ignored)."""
    sl = single_slices(N)
    sx = sl['x'] if sx is None else sx; sy = sl['y'] if sy is None else sy
    dx, dy = free[sx] - c[0], free[sy] - c[1]
    e = _obst_e(dx, dy, r, kind)
    g = np.zeros_like(free)
    g[sx] = scale / N * -2.0 * dx * e
    g[sy] = scale / N * -2.0 * dy * e
    return scale / N * np.sum(e), g


def obstacles(free, N, scale, obss, kind, sx=None, sy=None):
    """:136-144."""
    c, g = 0.0, np.zeros_like(free)
    for o in obss:
        ci, gi = obstacle(free, N, scale, (o[0], o[1]), o[2], kind, sx, sy)
        c += ci; g += gi
    return c, g


def composit(free, N, scale, obss, vsp, kobs, kvel, kbank, kind):
    """:147-165 (obss=None drops the obstacle term)."""
    cv, gv = airvel(free, N, scale, vsp); cb, gb = bank(free, N, scale, True)
    c, g = kvel * cv + kbank * cb, kvel * gv + kbank * gb
    if obss is not None:
        co, go = obstacles(free, N, scale, obss, kind)
        c += kobs * co; g += kobs * go
    return c, g


# ---- multi-aircraft costs (src/d2d/multiopty_utils.py) ------------------------------
def m_input(free, N, n, scale, vsp, kv, kphi):
    """:56-71 (CostAirvel = kv 1,kphi 0 :33-43 ; CostBank = kv 0,kphi 1 :45-54)."""
    sl = multi_slices(N, n)
    g = np.zeros_like(free); sp = 0.0; sv = 0.0
    for s in sl['phi']:
        g[s] = kphi * 2 * free[s]; sp += np.sum(free[s] ** 2)
    for s in sl['v']:
        g[s] = kv * 2 * (free[s] - vsp); sv += np.sum((free[s] - vsp) ** 2)
    f = scale / N / n
    return f * (kv * sv + kphi * sp), g * f


def m_obstacles(free, N, n, scale, obss, kind):
    """:74-116 -- aircraft 0 only, scale without /n."""
    sl = multi_slices(N, n)
    return obstacles(free, N, scale, obss, kind, sl['x'][0], sl['y'][0])


def m_collision(free, N, n, scale, r, k=2.0):
    """:120-153 -- pair (0,1) only."""
    sl = multi_slices(N, n)
    dx = free[sl['x'][0]] - free[sl['x'][1]]; dy = free[sl['y'][0]] - free[sl['y'][1]]
    e = np.exp(-((dx / r * k) ** 2 + (dy / r * k) ** 2))
    g = np.zeros_like(free)
    g[sl['x'][0]] = scale / N * -2.0 * dx * e; g[sl['y'][0]] = scale / N * -2.0 * dy * e
    g[sl['x'][1]] = scale / N * 2.0 * dx * e; g[sl['y'][1]] = scale / N * 2.0 * dy * e
    return scale / N * np.sum(e), g


def m_composit(free, N, n, scale, kvel, kbank, kobs, kcol, vsp, obss, kind, rcol):
    """:156-174 (NaN gates)."""
    c, g = m_input(free, N, n, scale, vsp, kvel, kbank)
    if not np.isnan(kobs):
        co, go = m_obstacles(free, N, n, scale, obss, kind); c += kobs * co; g = g + kobs * go
    if not np.isnan(kcol):
        cc, gc = m_collision(free, N, n, scale, rcol); c += kcol * cc; g = g + kcol * gc
    return c, g


# ---- initial guesses ----------------------------------------------------------------
def triangle(p0, p1, va, duration, num_nodes, go_left=1.0):
    """src/d2d/opty_utils.py:171-187 -> x, y, psi, phi, v."""
    p0 = np.asarray(p0, float); p1 = np.asarray(p1, float)
    p0p1 = p1 - p0
    d = np.linalg.norm(p0p1)
    u = p0p1 / d; v = np.array([-u[1], u[0]])
    D = va * duration
    p2 = p0 + p0p1 / 2
    if D > d:
        p2 = p2 + np.sign(go_left) * np.sqrt(D ** 2 - d ** 2) / 2 * v
    n1 = int(num_nodes / 2); n2 = num_nodes - n1
    pts = np.vstack((np.linspace(p0, p2, n1), np.linspace(p2, p1, n2)))
    a = p2 - p0; b = p1 - p2
    psis = np.hstack((np.arctan2(a[1], a[0]) * np.ones(n1), np.arctan2(b[1], b[0]) * np.ones(n2)))
    return pts[:, 0], pts[:, 1], psis, np.zeros(num_nodes), va * np.ones(num_nodes)


def single_guess(kind, p0, p1, vref, duration, N):
    """src/single_opt_planner.py:79-115 ('tri' goes to -normal; else straight line, psi/phi/v = 0)."""
    g = np.zeros(5 * N); sl = single_slices(N)
    if kind == 'tri':
        x, y, psi, phi, v = triangle(p0[:2], p1[:2], vref, duration, N, -1.0)
        g[sl['x']], g[sl['y']], g[sl['psi']], g[sl['phi']], g[sl['v']] = x, y, psi, phi, v
    else:
        g[sl['x']] = np.linspace(p0[0], p1[0], N); g[sl['y']] = np.linspace(p0[1], p1[1], N)
    return g


def multi_guess_tri(p0s, p1s, vref, duration, N):
    """src/multi_opt_planner.py:107-110."""
    n = len(p0s); g = np.zeros(5 * n * N); sl = multi_slices(N, n)
    for i, (p0, p1) in enumerate(zip(p0s, p1s)):
        x, y, psi, phi, v = triangle(np.array(p0)[:2], np.array(p1)[:2], vref, duration, N, -1.0)
        g[sl['x'][i]], g[sl['y'][i]], g[sl['psi'][i]], g[sl['phi'][i]], g[sl['v'][i]] = x, y, psi, phi, v
    return g


def collocation_residual(free, N, h, wind=(0.0, 0.0)):
    """Backward-Euler collocation of src/d2d/opty_utils.py:38-50 as opty discretises it
    (SURVEY.md appendix A); note the +wind sign quirk of the symbolic model."""
    sl = single_slices(N)
    x, y, psi, phi, v = (free[sl[k]] for k in ('x', 'y', 'psi', 'phi', 'v'))
    r1 = (x[1:] - x[:-1]) / h - v[1:] * np.cos(psi[1:]) + wind[0]
    r2 = (y[1:] - y[:-1]) / h - v[1:] * np.sin(psi[1:]) + wind[1]
    r3 = (psi[1:] - psi[:-1]) / h - 9.81 / v[1:] * np.tan(phi[1:])
    return np.concatenate([r1, r2, r3])
