#!/usr/bin/env python3
"""(GPU, development) How long do the longest fits of the headline batch take when each has a SIMD to itself?  That time is the
floor of the 4096-fit launch whatever the hand-out order.
  python tools/dev_longest_alone.py [n_longest]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import torch      # noqa: E402
import d2dhip     # noqa: E402
import bench      # noqa: E402

n_long = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = d2dhip.Context(0)
dur, wref = bench._plan_consts()
plan = d2dhip.FitPlan(ctx, 6, 50, dur, wref)
B = 4096
sc = bench.bench_scenarios(B)
dsc = ctx.dev(sc)


def timed(d, n=7, **kw):
    q0 = plan.init(d)
    ts = []
    for _ in range(n):
        q = q0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cost, iters, status, stats = plan.solve(d, q, max_iter=150, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), iters.cpu().numpy()


t_all, it = timed(dsc)
print(f'4096 fits, predicted hand-out: {t_all*1e3:.3f} ms; trials median {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}')
plan.order_from_iters(ctx.dev(it.astype(np.int32)))
t_true, _ = timed(dsc)
plan.clear_order()
print(f'4096 fits, true order        : {t_true*1e3:.3f} ms')
order = np.argsort(-it, kind='stable')
for n in (1, 4, n_long, 64, 256, 1024):
    idx = order[:n]
    d = ctx.dev(np.ascontiguousarray(sc[idx]))
    t, it2 = timed(d)
    assert np.array_equal(it2, it[idx])
    print(f'the {n:4d} longest fits alone (trials {it[idx].min()}..{it[idx].max()}): {t*1e3:.3f} ms')
idx = order[-1024:]
t, it2 = timed(ctx.dev(np.ascontiguousarray(sc[idx])))
print(f'the 1024 SHORTEST fits alone (trials {it[idx].min()}..{it[idx].max()}): {t*1e3:.3f} ms')
