#!/usr/bin/env python3
"""In-kernel phase stamps of the fused LM kernel on the bench batch (D2D_LM_STAMPS=1 must be set)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import bench, d2dhip
ctx = d2dhip.Context(0)
dur, wref = bench._plan_consts()
plan = d2dhip.FitPlan(ctx, 6, 50, dur, wref)
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    dsc = ctx.dev(bench.bench_scenarios(B))
    q0 = plan.init(dsc)
    for _ in range(2):
        q = q0.clone()
        cost, iters, status, stats = plan.solve(dsc, q, max_iter=150)
        plan.order_from_iters(iters)
