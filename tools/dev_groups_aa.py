#!/usr/bin/env python3
"""Development: BASELINE configs[2] (8 aircraft x R replicas) with and without the Anderson acceleration of the sweep map
(D2D_GROUPS_AA = 0 / 1 / 2, read once per process): time, sweeps (D2D_GROUPS_DIAG=1 prints their distribution), fixed points.
  python tools/dev_groups_aa.py out.npz [R] [tol] [max_sweeps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
out = sys.argv[1]; R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6; MS = int(sys.argv[4]) if len(sys.argv) > 4 else 150
K, n_ac = 50, 8
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
SEED = int(os.environ.get('SEED', '1'))
if os.environ.get('TESTSET'):        # the configuration of tests/test_gpu_fullsize.py::test_8x8192_groups_properties
    s_ = 1.0 / K
    plan = d2dhip.FitPlan(ctx, 6, K, dur, (0.02 ** 2, s_ / n_ac * 5.0, s_ / n_ac / 9.81 ** 2))
    SEED = 3
else:
    plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))
dsc = ctx.dev(synth.circle_group_scenarios(n_ac, R, dur, K, seed=SEED).reshape(R * n_ac, -1))
q0 = plan.init(dsc)
best = 1e9
for rep in range(4):
    q = q0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    cost, sw, stats = plan.solve_groups(dsc, q, n_ac, max_sweeps=MS, inner_iters=8, tol=tol)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rep: best = min(best, dt)
print(f'AA={os.environ.get("D2D_GROUPS_AA", "default")} R={R} tol={tol:g}: {best * 1e3:.2f} ms ({R / best / 1e3:.1f} k scenarios/s), max sweeps {sw}, last move max {stats[2]:.3e}, evals {stats[3]:.0f}, sum cost {stats[0]:.10f}', flush=True)
sw_, mv_ = plan.group_report(R)
print('   unsettled at 1e-6:', int((mv_ > 1e-6).sum()), 'worst', np.argsort(-mv_)[:5], mv_[np.argsort(-mv_)[:5]], 'their sweeps', sw_[np.argsort(-mv_)[:5]])
np.savez(out, q=q.cpu().numpy(), cost=cost.cpu().numpy())
