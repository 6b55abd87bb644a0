#!/usr/bin/env python3
"""Development (CPU only): is a box-constrained turn-around of the single-aircraft catalogue (exp_0_3, exp_3) feasible at all?
A flight at constant speed v along K segments of constant curvature (|kappa| <= g tan(phi_max) / v^2, length >= 0) from p0 must
reach p1 -- position AND heading, the heading without a 2 pi ambiguity: the NLP's end condition is psi(t1) = psi1 as a number --
inside the box, with total length v (t1 - t0), v in [v_min, v_max].  Multi-start least squares on (end-pose error, box excess,
curvature excess) over (v, kappa_j, l_j).  Usage: dev_arcpaths.py [scenario] [K] [starts]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
from scipy.optimize import least_squares

G = 9.81


def path(p0, kap, ln, ds=0.5):
    """sample points (x, y, psi) along the arc sequence, every <= ds metres, plus the end pose"""
    x, y, psi = p0
    xs, ys, ps = [x], [y], [psi]
    for k, l in zip(kap, ln):
        n = ds if isinstance(ds, int) else max(1, int(np.ceil(l / ds)))
        s = np.linspace(0, l, n + 1)[1:]
        if abs(k) < 1e-9:
            xs.extend(x + s * np.cos(psi)); ys.extend(y + s * np.sin(psi)); ps.extend(psi + 0 * s)
            x, y = x + l * np.cos(psi), y + l * np.sin(psi)
        else:
            xs.extend(x + (np.sin(psi + k * s) - np.sin(psi)) / k); ys.extend(y - (np.cos(psi + k * s) - np.cos(psi)) / k)
            ps.extend(psi + k * s)
            x, y, psi = x + (np.sin(psi + k * l) - np.sin(psi)) / k, y - (np.cos(psi + k * l) - np.cos(psi)) / k, psi + k * l
    return np.array(xs), np.array(ys), np.array(ps)


def residual(z, K, p0, p1, T, phimax, box, margin):
    v = z[0]; kap = z[1:1 + K]; frac = np.abs(z[1 + K:1 + 2 * K])
    ln = frac / max(frac.sum(), 1e-9) * v * T                       # the lengths always add up to v T
    xs, ys, ps = path(p0, kap, ln, ds=48)          # (a fixed number of points per segment)
    kmax = G * np.tan(phimax) / v ** 2
    r = [xs[-1] - p1[0], ys[-1] - p1[1], 10.0 * (ps[-1] - p1[2])]
    r.extend(10.0 * np.maximum(np.abs(kap) - kmax * (1 - margin), 0.0) / kmax)
    r.extend(np.maximum(box[0][0] + margin * 5 - xs, 0.0)); r.extend(np.maximum(xs - box[0][1] + margin * 5, 0.0))
    r.extend(np.maximum(box[1][0] + margin * 5 - ys, 0.0)); r.extend(np.maximum(ys - box[1][1] + margin * 5, 0.0))
    return np.array(r)


def search(scen, K=6, starts=400, seed=0, margin=0.0, verbose=True):
    rng = np.random.default_rng(seed)
    p0, p1 = np.array(scen.p0[:3], float), np.array(scen.p1[:3], float)
    T = scen.t1 - scen.t0
    phimax = scen.phi_constraint[1]; vlo, vhi = scen.v_constraint
    box = (scen.x_constraint, scen.y_constraint)
    best = []
    for s in range(starts):
        v0 = rng.uniform(vlo, min(vhi, vlo + 2.0))
        kmax = G * np.tan(phimax) / v0 ** 2
        z0 = np.concatenate([[v0], rng.choice([-1, 0, 1], K) * kmax * rng.uniform(0.5, 1.0, K), rng.uniform(0.05, 1.0, K)])
        lo = np.concatenate([[vlo], -np.full(K, np.inf), np.zeros(K)]); hi = np.concatenate([[vhi], np.full(K, np.inf), np.full(K, np.inf)])
        try:
            res = least_squares(residual, z0, bounds=(lo, hi), args=(K, p0, p1, T, phimax, box, margin), xtol=1e-12, ftol=1e-12, max_nfev=300)
        except ValueError:
            continue
        best.append((float(np.sqrt(2 * res.cost)), res.x))
    best.sort(key=lambda t: t[0])
    if verbose:
        print(f'{scen.__name__}: K={K}, {starts} starts, margin {margin}: smallest violations', [round(b[0], 4) for b in best[:8]])
        for viol, z in best[:3]:
            v = z[0]; kap = z[1:1 + K]; frac = np.abs(z[1 + K:]); ln = frac / frac.sum() * v * T
            print(f'   viol {viol:.2e} v {v:.3f} kmax {G * np.tan(phimax) / v ** 2:.4f} segments',
                  [(round(k * v * v / G / np.tan(phimax), 2), round(l, 1)) for k, l in zip(kap, ln)])
    return best


if __name__ == '__main__':
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        import d2d.optyplan_scenarios as sc
    name = sys.argv[1] if len(sys.argv) > 1 else 'exp_0_3'
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    starts = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    search(getattr(sc, name), K, starts)
