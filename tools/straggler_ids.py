import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/drone-sim-python_amd')
import numpy as np, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0); K = 50
plan = d2dhip.FitPlan(ctx, 6, K, synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(0.1, K))
sc = ctx.dev(synth.synth_scenarios(4096))
q = plan.init(sc); cost, iters, status, stats = plan.solve(sc, q, max_iter=200, check_every=200)
it = iters.cpu().numpy(); st = status.cpu().numpy(); c = cost.cpu().numpy()
idx = np.argsort(-it)[:40]
print('idx', idx.tolist()); print('iters', it[idx].tolist()); print('status', st[idx].tolist()); print('cost', np.round(c[idx], 4).tolist())
