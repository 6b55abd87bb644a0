#!/usr/bin/env python3
"""configs[2] wall time against the sweep cap: separates the throughput part from the tail of slow scenarios (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import numpy as np
import torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0)
K, S = 50, 6
dur = synth.planner_timing(0, 4.9, 10)[2]
plan = d2dhip.FitPlan(ctx, S, K, dur, synth.default_wref(1.0, K))
R, n_ac = 8192, 8
sc = synth.circle_group_scenarios(n_ac, R, dur, K, seed=1)
dsc = ctx.dev(sc.reshape(R * n_ac, -1))
q0 = plan.init(dsc)
for hint in (False, True):
  inner = 8
  if hint:
    q = q0.clone(); plan.solve_groups(dsc, q, n_ac, max_sweeps=120, inner_iters=inner, tol=1e-10); plan.group_order_from_last(R)
    qa = q.clone()
  if True:
    for cap in ((120,) if hint else (40, 120)):
        best = 1e9
        for rep in range(3):
            q = q0.clone()
            torch.cuda.synchronize(); t = time.perf_counter()
            res = plan.solve_groups(dsc, q, n_ac, max_sweeps=cap, inner_iters=inner, tol=1e-10)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        if hint: print('same result with the hint:', bool((q == qa).all()))
        print(f'hint {hint} inner {inner} cap {cap:4d}  {best*1e3:8.2f} ms  evals {res[2][3]:.3e}  stats {[float(x) for x in res[2]]}', flush=True)
