#!/usr/bin/env python3
"""Distribution of Gauss-Seidel sweeps over the scenarios of BASELINE configs[2] (diagnostic)."""
import os, sys
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
for tol in (1e-10, 1e-8, 1e-6):
    q = plan.init(dsc)
    cost, sweeps, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=120, inner_iters=8, tol=tol)
    # per-scenario sweeps: the kernel leaves them in the flags (FL_ITERS) of every trajectory -> exported through a plain solve? read via cost/iters API:
    print('tol', tol, 'sweeps(max)', sweeps, 'stats', stats)
