#!/usr/bin/env python3
"""Development (round 6): the 4096-fit headline solve under different hand-out priors regressed on tools/data/handout_calib.npz --
the mean trial count per cell (additive, backfitted) against marginal quantiles / tail probabilities per cell (risk-aware keys: what
ends a launch is a long fit that starts late, not the average misordering).  usage: dev_handout_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip, bench
from d2dhip import synth, handout
ctx = d2dhip.Context(0)
dur, wref = bench._plan_consts()
plan = d2dhip.FitPlan(ctx, 6, 50, dur, wref)
cal = np.load(os.path.join(ROOT, 'tools', 'data', 'handout_calib.npz')); Bc = int(cal['B'])
sc = np.concatenate([synth.synth_scenarios(Bc, seed=20241008, rank=int(r), obj_scale=0.1, K=50) for r in cal['ranks']])
n = np.concatenate([cal[f'iters_{r}'].astype(np.float64) for r in cal['ranks']])
tables = {'builtin': None, 'mean': handout.fit_prior(sc, dur, n)}
for q in (0.75, 0.9, 0.95):
    tables[f'q{int(q * 100)}'] = handout.fit_prior(sc, dur, n, quantile=q)
for thr in (50, 60, 70):
    tables[f'p{thr}'] = handout.fit_prior(sc, dur, n, tail=thr)
res = {}
for rank in (0, 1, 2, 3):
    dsc = ctx.dev(bench.bench_scenarios(4096, rank))
    q0 = plan.init(dsc)
    for name in ['index'] + list(tables):
        if name != 'index':
            plan.set_handout_prior(tables[name])
        kw = dict(handout=d2dhip.HANDOUT_INDEX) if name == 'index' else {}
        ts = []
        for rep in range(6):
            q = q0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
            plan.solve(dsc, q, max_iter=150, **kw)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res.setdefault(name, []).append(np.median(ts[1:]) * 1e3)
for name, v in res.items():
    print(f'{name:8s} median ms per rank {np.round(v, 3)}  mean {np.mean(v):.3f}', flush=True)
