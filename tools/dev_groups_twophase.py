#!/usr/bin/env python3
"""Development: BASELINE configs[2] in two launches -- every scenario up to C sweeps, then the unsettled ones, the largest last move
first (longest-processing-time hand-out from a prediction made inside the solve) -- against the single launch in index order
and with the previous solve's order.  Emulated with the public calls (the second call restarts its sweep counter, so the line
search engages later than it would: timing study only).   python tools/dev_groups_twophase.py [C ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
R, K, n_ac, tol = 8192, 50, 8, 1e-6
caps = [int(a) for a in sys.argv[1:]] or [8, 10, 12, 14, 16]
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))
sc = synth.circle_group_scenarios(n_ac, R, dur, K, seed=1)
dsc = ctx.dev(sc.reshape(R * n_ac, -1))
q0 = plan.init(dsc)
def run(fn, reps=3):
    best = 1e9
    for rep in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        if rep: best = min(best, time.perf_counter() - t0)
    return best, out
def single():
    q = q0.clone(); plan.solve_groups(dsc, q, n_ac, max_sweeps=200, inner_iters=8, tol=tol); return q
t, _ = run(single); print(f'single launch, index order: {t * 1e3:.2f} ms', flush=True)
plan.group_order_from_last(R)
t, _ = run(single); print(f'single launch, previous solve\'s order: {t * 1e3:.2f} ms', flush=True)
plan.group_order_from_last(R, False)
for C in caps:
    def two():
        q = q0.clone()
        plan.solve_groups(dsc, q, n_ac, max_sweeps=C, inner_iters=8, tol=tol)
        sw, mv = plan.group_report(R)                      # (host round trip: the real thing would order on the device)
        rest = np.nonzero(mv > tol)[0]
        rest = rest[np.argsort(-mv[rest], kind='stable')]
        idx = torch.as_tensor((rest[:, None] * n_ac + np.arange(n_ac)[None, :]).reshape(-1), device=q.device)
        q2 = q[idx].contiguous(); s2 = dsc[idx].contiguous()
        plan.solve_groups(s2, q2, n_ac, max_sweeps=200, inner_iters=8, tol=tol)
        return len(rest)
    t, nrest = run(two)
    def first_only():
        q = q0.clone(); plan.solve_groups(dsc, q, n_ac, max_sweeps=C, inner_iters=8, tol=tol)
    t1, _ = run(first_only)
    print(f'two launches, first capped at {C} sweeps: {t * 1e3:.2f} ms (first launch alone {t1 * 1e3:.2f} ms, {nrest} scenarios go on)', flush=True)
