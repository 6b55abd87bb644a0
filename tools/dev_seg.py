#!/usr/bin/env python3
"""Development: the segment formulation of the long-horizon kernel (fit_seg.h, default) against the table kernels
(LONG_TABLES=3 in the tool's environment -> d2d_fit_plan_opts.long_tables): same minima, fits/s by node count, both solvers.
  python tools/dev_seg.py out.npz [K ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import torch, d2dhip
from d2dhip import synth

out = sys.argv[1]
Ks = [int(x) for x in sys.argv[2:]] or [65, 101, 121, 151, 201, 301, 501]
ctx = d2dhip.Context(0)
res = {}
B = int(os.environ.get('BATCH', '4096'))
MAXIT = int(os.environ.get('MAXIT', '300'))
for K in Ks:
    t1 = (K - 1) / 10.0
    dur = synth.planner_timing(0, t1, 10)[2]
    plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(0.1, K), long_tables=int(os.environ.get('LONG_TABLES', '-1')))
    dsc = ctx.dev(synth.synth_scenarios(B, seed=20241008, obj_scale=0.1, K=K, dist_range=(100. * t1 / 12, 150. * t1 / 12)))
    q0 = plan.init(dsc)
    for name, kw in (('minpack', {}), ('fast', dict(mode=d2dhip.MODE_FAST))):
        best = 1e9
        for rep in range(3):
            q = q0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            cost, iters, status, stats = plan.solve(dsc, q, max_iter=MAXIT, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        plan.order_from_iters(iters)
        q = q0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.solve(dsc, q, max_iter=MAXIT, **kw)
        torch.cuda.synchronize(); hinted = time.perf_counter() - t0
        plan.clear_order()
        st = status.cpu().numpy()
        print(f'K={K} {name:8s} long_tables={os.environ.get("LONG_TABLES", "-1")}: {best * 1e3:8.2f} ms  {B / best / 1e3:7.1f} k fits/s (hinted {B / hinted / 1e3:7.1f} k)  '
              f'mean iters {iters.float().mean().item():.1f} max {iters.max().item()} conv {(st == 1).mean():.4f} mean cost {cost.mean().item():.8f}', flush=True)
        res[f'cost_{K}_{name}'] = cost.cpu().numpy(); res[f'iters_{K}_{name}'] = iters.cpu().numpy(); res[f'q_{K}_{name}'] = q.cpu().numpy()
    plan.close()
np.savez(out, **res)
