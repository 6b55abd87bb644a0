"""Experiment: how well does the state after a few LM iterations predict the total iteration count?"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/drone-sim-python_amd')
import numpy as np, torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0); K = 50
plan = d2dhip.FitPlan(ctx, 6, K, synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(0.1, K))
B = 4096
sc = ctx.dev(synth.synth_scenarios(B))
q = plan.init(sc); cost, iters, status, stats = plan.solve(sc, q.clone(), max_iter=200, check_every=200)
it = iters.cpu().numpy()
feats = {}
c0, g0, _ = plan.eval(sc, q)
feats['c0'] = c0.cpu().numpy(); feats['g0'] = g0.abs().max(1).values.cpu().numpy()
for nb in (5, 10, 20, 30):
    qq = q.clone(); plan.begin(B); plan.iterate(sc, qq, nb, max_iter=200); plan.finish(sc, qq)
    c, g, _ = plan.eval(sc, qq)
    feats[f'c{nb}'] = c.cpu().numpy(); feats[f'g{nb}'] = g.abs().max(1).values.cpu().numpy()
    feats[f'dc{nb}'] = feats['c0'] - feats[f'c{nb}']
long = it >= np.percentile(it, 97)
print('iters pct 50/90/97/99/100', [np.percentile(it, p) for p in (50, 90, 97, 99, 100)], 'n long', long.sum())
from scipy.stats import spearmanr
for k, v in feats.items():
    r = spearmanr(v, it).correlation
    order = np.argsort(-v)
    hit25 = long[order[: B // 4]].sum() / long.sum(); hit50 = long[order[: B // 2]].sum() / long.sum()
    print(f'{k:6s} spearman {r:+.3f}  top-3%-longest found in predicted top 25%: {hit25:.2f}  top 50%: {hit50:.2f}')
