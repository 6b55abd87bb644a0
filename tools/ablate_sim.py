#!/usr/bin/env python3
"""Timing experiments on the simulation kernels (GPU box): history on/off, drones per launch.
  python tools/ablate_sim.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np                               # noqa: E402
import torch                                     # noqa: E402
import d2dhip                                    # noqa: E402

ctx = d2dhip.Context(0)
n_ac = 4
rng = np.random.default_rng(0)


def run(N, steps, record, reps=2):
    n_form = N // n_ac
    centres = np.tile(np.array([[0, -20], [25, -20], [25, -100], [0, -100.0]]), (n_form, 1)) + np.repeat(rng.uniform(-5, 5, (n_form, 2)), n_ac, 0)
    X0 = np.tile([20, 30, -np.pi / 2, 0, 10.0], (N, 1)) + np.concatenate([rng.uniform(-3, 3, (N, 2)), np.zeros((N, 3))], 1)
    dX0, dC, dR = ctx.dev(np.ascontiguousarray(X0.T)), ctx.dev(np.ascontiguousarray(centres.T)), ctx.dev(np.full(N, 60.0))
    out = ctx.gvf_run(dX0, dC, dR, n_ac, steps + 1, 0.05, 15.0, record=record); ctx.sync()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream); ctx.gvf_run(dX0, dC, dR, n_ac, steps + 1, 0.05, 15.0, record=record, out=out); e1.record(ctx.stream)
        ctx.sync()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    del out; torch.cuda.empty_cache()
    return best


for N, steps, rec in ((65536, 2000, ('X', 'U')), (65536, 2000, ()), (65536 * 4, 2000, ()), (65536 * 4, 2000, ('X', 'U')), (4096, 2000, ()),
                      (65536, 2000, ('X',)), (65536 * 16, 500, ())):
    t = run(N, steps, rec)
    print(f'N={N} steps={steps} record={rec}: {t*1e3:.2f} ms, {N*steps/t/1e9:.2f} G drone-steps/s, {t/steps*1e6:.2f} us/step', flush=True)
