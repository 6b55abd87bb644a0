#!/usr/bin/env python3
"""Development: BASELINE configs[2] (8 aircraft x R replicas) against the inner iteration budget of a visit of the block
Gauss-Seidel (inner_iters): time, sweep distribution, evaluations, and whether the fixed points are the ones of inner_iters = 8.
  python tools/dev_groups_inner.py [R] [tol] [max_sweeps] [inner ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6; MS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
inners = [int(a) for a in sys.argv[4:]] or [8, 6, 4, 3, 2, 1]
K, n_ac = 50, 8
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))
dsc = ctx.dev(synth.circle_group_scenarios(n_ac, R, dur, K, seed=int(os.environ.get('SEED', '1'))).reshape(R * n_ac, -1))
q0 = plan.init(dsc)
ref = None
for inner in inners:
    best = 1e9
    for rep in range(3):
        q = q0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        cost, sw, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=MS, inner_iters=inner, tol=tol)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    sw_, mv_ = plan.group_report(R)
    qn, cn = q.cpu().numpy(), cost.cpu().numpy().reshape(R, n_ac).sum(1)
    line = (f'inner {inner}: {best * 1e3:7.2f} ms  sweeps mean {sw_.mean():.1f} p50 {np.median(sw_):.0f} p99 {np.percentile(sw_, 99):.0f} max {sw_.max()} '
            f'beyond40 {int((sw_ > 40).sum())} unsettled {int((mv_ > tol).sum())}  evals {stats[3]:.3e}  sum cost {stats[0]:.8f}')
    if ref is None: ref = (qn, cn)
    else:
        dq = (np.abs(qn - ref[0]).max(1) / (1 + np.abs(ref[0]).max(1))).reshape(R, n_ac).max(1)
        dc = np.abs(cn - ref[1]) / np.abs(ref[1])
        line += f'  | vs first: scenarios with q diff > 1e-4: {int((dq > 1e-4).sum())}, cost diff > 1e-6: {int((dc > 1e-6).sum())}, lower cost {int((cn < ref[1] * (1 - 1e-6)).sum())}'
    print(line, flush=True)
