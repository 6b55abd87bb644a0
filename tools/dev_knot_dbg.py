#!/usr/bin/env python3
"""Development: first evaluation of the knot kernel (H_u, g_u, u) against the oracle's knot-space statement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import d2dhip
import bench
from oracle import fit as F, fit_knot as FK
np.set_printoptions(linewidth=220, precision=4)
ctx = d2dhip.Context(0)
dur, wref = bench._plan_consts()
B = 8
sc = bench.bench_scenarios(B)
dsc = ctx.dev(sc)
os.environ['D2D_KNOT_DEBUG'] = '/tmp/knot_dbg.bin'
plan = d2dhip.FitPlan(ctx, 6, 50, dur, wref, kernel='knot')
q0 = plan.init(dsc)
q = q0.clone()
cost, iters, status, stats = plan.solve(dsc, q, max_iter=1)
d = np.fromfile('/tmp/knot_dbg.bin', dtype=np.float32).reshape(B, 48 * 48 + 192)
ob = F.FitBasis.from_arrays(6, 50, dur, *plan.basis())
kb = FK.KnotBasis(ob)
for i in range(2):
    H = d[i, :2304].reshape(48, 48).astype(np.float64); gu = d[i, 2304:2352]; uu = d[i, 2352:2400]
    u_or = kb.to_u(sc[i], q0.cpu().numpy()[i])
    c, g_or, H_or = kb.eval_normal(sc[i], u_or)
    print('fit', i, 'u err', np.abs(uu - u_or).max() / np.abs(u_or).max(), 'g err', np.abs(gu - g_or).max() / np.abs(g_or).max(),
          'H err', np.abs(H - H_or).max() / np.abs(H_or).max(), 'H sym', np.abs(H - H.T).max() / np.abs(H).max())
    dl = d[i, 2400:2448].astype(np.float64); x = d[i, 2448:2456]
    s_or = np.linalg.solve(H_or, -g_or)
    dxn = np.sqrt(s_or @ kb.Mu @ s_or); w = kb.Mu @ s_or / dxn; t2 = w @ np.linalg.solve(H_or, w)
    gn = np.sqrt(g_or @ kb.Mu_inv @ g_or); xn = np.sqrt((u_or - kb.u0(sc[i])) @ kb.Mu @ (u_or - kb.u0(sc[i])))
    print('   step err', np.abs(dl - s_or).max() / np.abs(s_or).max(), 'dxn', x[0], dxn, 't2', x[1], t2, 'gnrm', x[2], gn, 'ok', x[3], 'delta', x[4], 100 * xn, 'lam', x[5])
    if np.abs(gu - g_or).max() > 1e-4 * np.abs(g_or).max():
        print(' g kernel', gu[:16]); print(' g oracle', g_or[:16])
    E = np.abs(H - H_or) / np.abs(H_or).max()
    if E.max() > 1e-4:
        bad = np.argwhere(E > 1e-4)
        print(' bad H entries', len(bad), bad[:12].tolist())
        print(' H kernel[0:8,0:8]\n', H[:8, :8]); print(' H oracle\n', H_or[:8, :8])
