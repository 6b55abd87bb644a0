#!/usr/bin/env python3
"""Trial counts of the default solver on synthetic batches of OTHER seeds than the bench's (rank offsets 100 ..): the data
tools/fit_handout_prior.py regresses the hand-out prior on (csrc/fit_handout_prior.h).  GPU box.
usage: dump_handout_calib.py out.npz [n_batches] [B] [K] [first_rank]   (K > 64: the long-horizon bench family, chords 100-150 m per 12 s)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, d2dhip
from d2dhip import synth
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
r0 = int(sys.argv[5]) if len(sys.argv) > 5 else 100
t1 = (K - 1) / 10.0
dur = synth.planner_timing(0, t1, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(0.1, K))
kw = {} if K <= 64 else dict(dist_range=(100. * t1 / 12.0, 150. * t1 / 12.0))
out = {'ranks': np.arange(r0, r0 + nb), 'B': B, 'K': K}
for r in range(r0, r0 + nb):
    dsc = ctx.dev(synth.synth_scenarios(B, seed=20241008, rank=r, obj_scale=0.1, K=K, **kw))
    q = plan.init(dsc)
    cost, iters, status, stats = plan.solve(dsc, q, max_iter=300)
    out[f'iters_{r}'] = iters.cpu().numpy().astype(np.int16)
    print(r, float(iters.float().mean()), int(iters.max()), flush=True)
np.savez_compressed(sys.argv[1], **out)
