"""Throughput of the fused LM solve against the batch size (one GPU): shows where the 4096-fit tail stops mattering."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import numpy as np, torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0); K = 50
plan = d2dhip.FitPlan(ctx, 6, K, synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(0.1, K))
for B in (2048, 4096, 8192, 16384, 32768, 65536):
    sc = ctx.dev(synth.synth_scenarios(B)); q0 = plan.init(sc)
    plan.solve(sc, q0.clone(), check_every=200)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 3
    for _ in range(n):
        cost, iters, status, stats = plan.solve(sc, q0.clone(), check_every=200)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(json.dumps({'batch': B, 'ms': 1e3 * dt, 'fits_per_s': B / dt, 'jtj_tflops': 470400 * stats[3] / dt / 1e12,
                      'frac_fp32_mfma': 470400 * stats[3] / dt / 1e12 / 157.3, 'mean_iters': float(iters.double().mean())}))
