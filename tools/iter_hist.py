import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/drone-sim-python_amd')
import numpy as np, torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0); K = 50
plan = d2dhip.FitPlan(ctx, 6, K, synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(0.1, K))
sc = ctx.dev(synth.synth_scenarios(4096)); q = plan.init(sc)
cost, iters, status, stats = plan.solve(sc, q, max_iter=400, check_every=400)
it = iters.cpu().numpy(); st = status.cpu().numpy()
print('status counts', np.bincount(st, minlength=5))
for p in (50, 75, 90, 95, 99, 99.5, 99.9, 100): print('pct', p, np.percentile(it, p))
for m in (60, 80, 100, 120, 150, 200, 300): print('iters <=', m, (it <= m).mean())
