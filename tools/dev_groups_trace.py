#!/usr/bin/env python3
"""Development: convergence curves of the block Gauss-Seidel of BASELINE configs[2] (8 aircraft x R replicas): one sweep per call
(max_sweeps = 1), the largest relative move of every scenario after each sweep -> npz.  python tools/dev_groups_trace.py out.npz [R] [sweeps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
out = sys.argv[1]; R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192; NS = int(sys.argv[3]) if len(sys.argv) > 3 else 150
K, n_ac = 50, 8
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))
dsc = ctx.dev(synth.circle_group_scenarios(n_ac, R, dur, K, seed=1).reshape(R * n_ac, -1))
q = plan.init(dsc)
moved = np.zeros((NS, R))
for s in range(NS):
    qp = q.clone()
    plan.solve_groups(dsc, q, n_ac, max_sweeps=1, inner_iters=8, tol=0.0)
    d = (q - qp).abs().amax(1) / (1.0 + qp.abs().amax(1))
    moved[s] = d.view(R, n_ac).amax(1).cpu().numpy()
first = np.array([np.argmax(moved[:, r] <= 1e-6) if (moved[:, r] <= 1e-6).any() else NS for r in range(R)])
print('sweeps to 1e-6: mean %.1f p50 %d p90 %d p99 %d max %d, never: %d' % (first.mean(), np.percentile(first, 50), np.percentile(first, 90), np.percentile(first, 99), first.max(), (first >= NS).sum()))
slow = np.argsort(-first)[:16]
np.savez(out, moved=moved[:, slow], slow=slow, first=first, q=q.view(R, n_ac, -1)[slow].cpu().numpy())
for r in slow[:6]:
    print(r, first[r], ' '.join('%.1e' % v for v in moved[::6, r][:25]))
