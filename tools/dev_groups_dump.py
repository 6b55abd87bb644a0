#!/usr/bin/env python3
"""Development (round 6): sweeps per scenario of BASELINE configs[2] (8 aircraft x R replicas) for several seeds -> npz: is a scenario's
sweep count predictable from its rows (a hand-out prior for d2d_fit_solve_groups)?  usage: dev_groups_dump.py out.npz [R] [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip, bench
from d2dhip import synth
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
seeds = [int(s) for s in sys.argv[3:]] or [1, 101, 102]
ctx = d2dhip.Context(0)
dur, _ = bench._plan_consts()
K, n_ac = bench.K, 8
plan = d2dhip.FitPlan(ctx, bench.S_, K, dur, synth.default_wref(1.0, K))
out = {}
for sd in seeds:
    dsc = ctx.dev(synth.circle_group_scenarios(n_ac, R, dur, K, seed=sd).reshape(R * n_ac, -1))
    q = plan.init(dsc)
    c0, _, _ = plan.eval(dsc, q, want_H=False)
    plan.solve_groups(dsc, q, n_ac, max_sweeps=200, inner_iters=8, tol=1e-6)
    sw, mv = plan.group_report(R)
    out[f'sweeps_{sd}'] = sw; out[f'cost0_{sd}'] = c0.cpu().numpy().reshape(R, n_ac)
    print(sd, sw.mean(), sw.max(), flush=True)
np.savez_compressed(sys.argv[1], **out)
