#!/usr/bin/env python3
"""Development: multi-start of the collocation backend on the two catalogue cases the dog-leg guess does not solve (exp_0_3, exp_3):
a family of loop-shaped initial guesses (straight leg, n + 1/2 loops of radius R, straight in), all solved in ONE launch."""
import os, sys, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import d2dhip
import d2d.optyplan_scenarios as sc
import single_opt_planner as sop


def loop_guess(p0, p1, N, dur, R, n_loops, direction, lead, box):
    """(5, N): fly `lead` metres along the start heading, turn (n_loops + the heading change) at radius R in `direction`, fly straight
    to a point behind p1 on its heading line, arrive.  Sampled at constant speed over the duration."""
    x0, y0, psi0 = p0[:3]; x1, y1, psi1 = p1[:3]
    pts = [(x0, y0)]
    a = np.array([x0 + lead * np.cos(psi0), y0 + lead * np.sin(psi0)])
    pts.append(tuple(a))
    # circle centre to the side of the heading
    c = a + R * direction * np.array([-np.sin(psi0), np.cos(psi0)])
    th0 = np.arctan2(a[1] - c[1], a[0] - c[0])
    dpsi = (psi1 - psi0) * direction
    dpsi = dpsi % (2 * np.pi)
    tot = dpsi + 2 * np.pi * n_loops
    for th in np.linspace(0, tot, max(8, int(tot / 0.15)))[1:]:
        pts.append((c[0] + R * np.cos(th0 + direction * th), c[1] + R * np.sin(th0 + direction * th)))
    # straight in along psi1
    e = np.array(pts[-1])
    back = np.array([x1, y1]) - 10.0 * np.array([np.cos(psi1), np.sin(psi1)])
    pts.append(tuple(back)); pts.append((x1, y1))
    P = np.array(pts)
    seg = np.hypot(*np.diff(P, axis=0).T)
    s = np.concatenate([[0], np.cumsum(seg)])
    si = np.linspace(0, s[-1], N)
    x = np.interp(si, s, P[:, 0]); y = np.interp(si, s, P[:, 1])
    if box[0] is not None:
        x = np.clip(x, box[0][0] + 0.5, box[0][1] - 0.5)
    if box[1] is not None:
        y = np.clip(y, box[1][0] + 0.5, box[1][1] - 0.5)
    x[0], y[0], x[-1], y[-1] = x0, y0, x1, y1
    psi = np.unwrap(np.arctan2(np.gradient(y), np.gradient(x)))
    psi += psi0 - psi[0]
    v = np.full(N, s[-1] / dur)
    phi = np.arctan(v ** 2 / (9.81 * R)) * direction * np.ones(N)
    return np.stack([x, y, psi, phi, v]), s[-1]


for scen in (sc.exp_3, sc.exp_0_3):
    with contextlib.redirect_stdout(io.StringIO()):
        p = sop.Planner(scen, initialize=True, backend='nlp')
        rows, _ = p.prob._rows()
    N = p.num_nodes
    guesses, tags = [], []
    for R in (14.5, 15., 15.5, 16., 18., 20.):
        for n_loops in (0, 1, 2):
            for d in (1, -1):
                for lead in (2., 10., 19.6, 22., 25., 28.):
                    g, L = loop_guess(scen.p0, scen.p1, N, p.duration, R, n_loops, d, lead, (scen.x_constraint, scen.y_constraint))
                    v = L / p.duration
                    if scen.v_constraint[0] * 0.8 <= v <= scen.v_constraint[1] * 1.2:
                        g[4] = np.clip(g[4], scen.v_constraint[0] + 0.2, scen.v_constraint[1] - 0.2)
                        g[3] = np.clip(g[3], scen.phi_constraint[0] * 0.95, scen.phi_constraint[1] * 0.95)
                        guesses.append(g); tags.append((R, n_loops, d, lead, round(L, 1)))
    B = len(guesses)
    ctx = d2dhip.default_context()
    W = ctx.dev(np.ascontiguousarray(np.stack(guesses)))
    out = ctx.nlp_solve(ctx.dev(np.tile(rows[:1], (B, 1))), W, p.time_step, outer_max=50)
    ctx.sync()
    st = out['status'].cpu().numpy(); cost = out['cost'].cpu().numpy(); feas = out['feas'].cpu().numpy(); it = out['iters'].cpu().numpy()
    print(scen.__name__, 'guesses', B, 'status counts', np.bincount(st, minlength=5), 'converged:', [(tags[i], round(float(cost[i]), 5), float(feas[i]), int(it[i])) for i in np.nonzero(st == 1)[0][:8]])
    print('   best feas of the rest', float(feas[st != 1].min()) if (st != 1).any() else None)
