#!/usr/bin/env python3
"""Development: the knot-coordinate kernel (fit_knot.hip) against the q-coordinate kernel and the scipy golden on the bench batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import torch
import d2dhip
import bench
from oracle import fit as F

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_scipy_bench1024.npz'))
ctx = d2dhip.Context(0)
dur, wref = bench._plan_consts()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = bench.bench_scenarios(B)
dsc = ctx.dev(sc)
res = {}
for name, kern in (('knot', 'knot'), ('q', 'fused')):
    plan = d2dhip.FitPlan(ctx, 6, 50, dur, wref, kernel=kern)
    q0 = plan.init(dsc)
    q = q0.clone()
    cost, iters, status, stats = plan.solve(dsc, q, max_iter=150)
    best = 1e30
    for _ in range(5):
        q = q0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cost, iters, status, stats = plan.solve(dsc, q, max_iter=150)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    c = cost.cpu().numpy(); it = iters.cpu().numpy(); st = status.cpu().numpy()
    z = plan.coeffs(dsc, q).cpu().numpy().reshape(B, -1)
    n = min(B, 1024)
    relc = np.abs(c[:n] - g['k50_cost'][:n]) / g['k50_cost'][:n]
    relz = np.abs(z[:n] - g['k50_z'][:n]).max(1) / np.abs(g['k50_z'][:n]).max(1)
    c1, g1, _ = plan.eval(dsc, q, want_H=False)
    print(f'{name}: {best*1e3:.3f} ms ({B/best/1e6:.3f} M fits/s)  status {np.bincount(st, minlength=5)}  iters mean {it.mean():.2f} max {it.max()}  evals/fit {stats[3]/B:.2f}'
          f'  same as scipy {((relc<=1e-6)&(relz<=1e-6)).mean():.4f} cost ok {(relc<=1e-6).mean():.4f}  |cost - eval| {np.abs(c1.cpu().numpy()-c).max():.2e}  max|g| {g1.abs().max().item():.2e} stats gmax {stats[1]:.2e}', flush=True)
    res[name] = (c, it, q.cpu().numpy())
    plan.close()
a, b = res['knot'], res['q']
print('knot vs q: same cost', (np.abs(a[0]-b[0]) <= 1e-6*b[0]).mean(), 'iters equal', (a[1]==b[1]).mean(), 'within 3', (np.abs(a[1]-b[1])<=3).mean(),
      'q rel diff max', np.nanmax(np.abs(a[2]-b[2]).max(1)/np.abs(b[2]).max(1)))
