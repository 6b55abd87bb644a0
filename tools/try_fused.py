"""Cautious first run of the fused LM kernel: tiny batch, progress lines flushed to a file."""
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/drone-sim-python_amd')
import numpy as np, torch
import d2dhip
from d2dhip import synth
log = open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'try_fused.log'), 'w')
def P(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
ctx = d2dhip.Context(0)
K, S = 50, 6
dur = synth.planner_timing(0, 4.9, 10)[2]
plan = d2dhip.FitPlan(ctx, S, K, dur, synth.default_wref(0.1, K))
for B, budget in ((1, 1), (8, 2), (64, 8), (4096, 8)):
    sc = ctx.dev(synth.synth_scenarios(B)); q = plan.init(sc)
    plan.begin(B)
    P('B', B, 'budget', budget, 'launching'); t = time.time()
    r = plan.iterate(sc, q, budget, max_iter=200)
    P('  running after', r, 'dt', round(time.time() - t, 3))
    for i in range(30):
        r = plan.iterate(sc, q, 8, max_iter=200)
        if r == 0: break
    cost, iters, status, stats = plan.finish(sc, q)
    P('  done: mean iters', float(iters.double().mean()), 'status counts', np.bincount(status.cpu().numpy(), minlength=5).tolist(), 'stats', stats.tolist())
P('OK')
