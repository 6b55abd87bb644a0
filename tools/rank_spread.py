"""How much do the per-rank batches of the weak-scaling bench differ? (one GPU, ranks' seeds one after the other)"""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0); K = 50
plan = d2dhip.FitPlan(ctx, 6, K, synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(0.1, K))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
MAX_ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for rank in range(8):
    sc = ctx.dev(synth.synth_scenarios(B, seed=20241008, rank=rank, obj_scale=0.1, K=K)); q0 = plan.init(sc)
    plan.clear_order()
    _, it0, _, _ = plan.solve(sc, q0.clone(), check_every=200, max_iter=MAX_ITER)
    plan.order_from_iters(it0)                       # as bench.py: longest-first hand-out from the warm-up solve
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
    for _ in range(n):
        cost, iters, status, stats = plan.solve(sc, q0.clone(), check_every=200, max_iter=MAX_ITER)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    it = iters.cpu().numpy(); st = status.cpu().numpy()
    print(json.dumps({'rank': rank, 'ms': round(1e3 * dt, 3), 'mean_iters': round(float(it.mean()), 2), 'p99': int(np.percentile(it, 99)),
                      'max_iters': int(it.max()), 'status': np.bincount(st, minlength=5).tolist()}))
