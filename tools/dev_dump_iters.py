#!/usr/bin/env python3
"""Development: trial counts per fit of the headline batch (default solver) -> npz, for the hand-out simulations of tools/dev_handout_sim.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
K = 50
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(0.1, K))
out = {}
for B in (4096, 32768):
    dsc = ctx.dev(synth.synth_scenarios(B, seed=20241008, obj_scale=0.1, K=K))
    q = plan.init(dsc)
    cost, iters, status, stats = plan.solve(dsc, q)
    out[f'iters_{B}'] = iters.cpu().numpy()
np.savez(sys.argv[1], **out)
print({k: (v.mean(), v.max()) for k, v in out.items()})
