#!/usr/bin/env python3
"""Development: the bench batch (S = 6, K = 50) on the fused kernel and, with KERNEL=long in the tool's environment (d2d_fit_plan_opts.kernel), on the persistent segment kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
K = 50
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(0.1, K), kernel=os.environ.get('KERNEL', 'auto'))
for B in (4096, 32768):
    dsc = ctx.dev(synth.synth_scenarios(B, seed=20241008, obj_scale=0.1, K=K))
    q0 = plan.init(dsc)
    for name, kw in (('minpack', {}), ('fast', dict(mode=d2dhip.MODE_FAST))):
        best = 1e9
        for rep in range(4):
            q = q0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            cost, iters, status, stats = plan.solve(dsc, q, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f'{plan.kernel:6s} B={B} {name:8s}: {best * 1e3:7.3f} ms {B / best / 1e6:.3f} M fits/s mean iters {iters.float().mean().item():.1f} max {iters.max().item()} '
              f'conv {(status == 1).float().mean().item():.4f} mean cost {cost.mean().item():.8f}', flush=True)
