#!/usr/bin/env python3
"""Development: Newton steps per problem of the collocation batch (4096 perturbed exp_14) by final status -- what the launch waits for."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import numpy as np, torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows, W0, h = synth.nlp_problems(B)
dsc = ctx.dev(rows)
for kw in ({}, dict(inner_max=50), dict(inner_max=40), dict(inner_max=30, outer_max=80), dict(inner_max=25, outer_max=96), dict(inner_max=20, outer_max=120), dict(inner_max=15, outer_max=160)):
    best = 1e9
    for rep in range(2):
        W = ctx.dev(np.ascontiguousarray(W0)); torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ctx.nlp_solve(dsc, W, h, **kw); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy(); feas = out['feas'].cpu().numpy(); cost = out['cost'].cpu().numpy()
    print(f'{kw}: {best * 1e3:.1f} ms ({B / best / 1e3:.1f} k problems/s)')
    for s_, name in ((1, 'converged'), (2, 'maxiter'), (3, 'nonfinite'), (4, 'stalled')):
        m = st == s_
        if m.any():
            print(f'   {name:10s} {m.sum():5d}  Newton steps mean {it[m].mean():6.1f} p50 {np.percentile(it[m], 50):5.0f} p90 {np.percentile(it[m], 90):5.0f} p99 {np.percentile(it[m], 99):5.0f} max {it[m].max():4d}   feas median {np.median(feas[m]):.1e} max {feas[m].max():.1e}')
    if not kw: ref = (st.copy(), cost.copy())
    else:
        both = (st == 1) & (ref[0] == 1)
        print(f'   converged in both: {both.sum()}, cost rel diff max {np.abs(cost[both] - ref[1][both]).max() / np.abs(ref[1][both]).max():.1e}; converged only in the default: {((ref[0] == 1) & (st != 1)).sum()}, only here: {((ref[0] != 1) & (st == 1)).sum()}')
