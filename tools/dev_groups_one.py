#!/usr/bin/env python3
"""Development: ONE scenario of BASELINE configs[2] (index argv[1] of the 8192) through fit_groups_kernel (debug build prints the sweeps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
idx = int(sys.argv[1]); K, n_ac = 50, 8
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
if os.environ.get('TESTSET'):
    s_ = 1.0 / K
    plan = d2dhip.FitPlan(ctx, 6, K, dur, (0.02 ** 2, s_ / n_ac * 5.0, s_ / n_ac / 9.81 ** 2))
    sc = synth.circle_group_scenarios(n_ac, 8192, dur, K, seed=3, sigma=2.0)[idx:idx + 1]
else:
    plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))
    sc = synth.circle_group_scenarios(n_ac, 8192, dur, K, seed=1)[idx:idx + 1]
dsc = ctx.dev(sc.reshape(n_ac, -1))
q = plan.init(dsc)
cost, sw, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=int(sys.argv[2]) if len(sys.argv) > 2 else 150, inner_iters=8, tol=float(os.environ.get('TOL', '1e-6')))
torch.cuda.synchronize()
print('sweeps', sw, 'last move', stats[2], 'cost', cost.sum().item())
