#!/usr/bin/env python3
"""Development: the time-sliced hand-out (d2d_fit_opts.slice) against run-to-completion on the bench batch sizes, default solver."""
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
for B in (2048, 3072, 4096, 6144, 8192, 16384, 32768):
    for rank in (0, 1):
        d = ctx.dev(synth.synth_scenarios(B, seed=20241008, rank=rank, obj_scale=0.1, K=K))
        qq0 = plan.init(d)
        line = f'B={B} rank {rank}:'
        for sl in (0, 8, 12, 16, 24, 32):
            ts = []
            for rep in range(8):
                q = qq0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
                plan.solve(d, q, max_iter=150, check_every=200, slice=sl)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            line += f' slice {sl}: {np.median(ts) * 1e3:.3f} ms;'
        print(line, flush=True)
