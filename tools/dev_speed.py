#!/usr/bin/env python3
"""Development timing of the fused solver kernel (GPU box): 4096 and 32 768 fits, default and fast mode, index order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import torch, d2dhip
from d2dhip import synth
K, S_ = 50, 6
dur = synth.planner_timing(0, 4.9, 10)[2]; wref = synth.default_wref(0.1, K)
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, S_, K, dur, wref)
for name, kw in [('minpack', dict(mode=0)), ('fast', dict(mode=1))]:
    for B in (4096, 32768):
        d = ctx.dev(synth.synth_scenarios(B, seed=20241008, rank=0, obj_scale=0.1, K=K))
        qq0 = plan.init(d)
        best = 1e9
        for rep in range(5):
            q = qq0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            cost, iters, status, stats = plan.solve(d, q, max_iter=150, check_every=200, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f'{name:8s} B={B}: {best * 1e3:.3f} ms  {B / best / 1e6:.3f} M fits/s  mean iters {iters.float().mean().item():.2f} max {iters.max().item()} conv {(status == 1).float().mean().item():.4f} mean cost {cost.mean().item():.9f}', flush=True)
