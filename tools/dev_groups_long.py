#!/usr/bin/env python3
"""Development: coupled groups at 65 .. 229 nodes -- the visits of the block Gauss-Seidel as launch pairs (default there) or as one
launch of the persistent segment kernel each (the default beyond 64 nodes; PAIRS=1 in the tool's environment -> d2d_fit_opts.gs_pairs: the launch pairs).  python tools/dev_groups_long.py [K ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
from oracle import fit as F
ctx = d2dhip.Context(0)
for K in [int(x) for x in sys.argv[1:]] or [71, 121, 201]:
    hz = 10.0
    dur = F.planner_timing(0, (K - 1) / hz, hz)[2]
    n_ac = 4
    s = 1.0 / K
    plan = d2dhip.FitPlan(ctx, 6, K, dur, (0.02 ** 2, s * 5.0 / n_ac, s / n_ac / F.G_ACC ** 2))
    for R in (1, 256):
        sc = synth.circle_group_scenarios(n_ac, R, dur, K, seed=3, obj_scale=1.0)
        dsc = ctx.dev(sc.reshape(R * n_ac, -1))
        best = 1e9
        for rep in range(3):
            q = plan.init(dsc); torch.cuda.synchronize(); t0 = time.perf_counter()
            c, sw, st = plan.solve_groups(dsc, q, n_ac, max_sweeps=80, inner_iters=8, tol=1e-9, gs_pairs=int(os.environ.get('PAIRS', '0')))
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f'K={K} R={R} pairs={os.environ.get("PAIRS", "0")}: {best * 1e3:8.2f} ms, {sw} sweeps, cost sum {c.sum().item():.8f}', flush=True)
    plan.set_groups(1); plan.close()
