#!/usr/bin/env python3
"""BASELINE configs[2]: multi_opt_planner's 8-drone circular-formation scenario x 8192 random-init replicas on one
MI355X -- block Gauss-Seidel over the aircraft with collision rows (d2d_fit_solve_groups).  Prints one JSON line.

  python tools/bench_groups.py [--replicas 8192] [--n-ac 8] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--replicas', type=int, default=8192)
    ap.add_argument('--n-ac', type=int, default=8)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--inner-iters', type=int, default=8)
    ap.add_argument('--tol', type=float, default=1e-10)
    a = ap.parse_args()
    import torch
    import d2dhip
    from d2dhip import synth
    ctx = d2dhip.Context(0)
    K, S = 50, 6
    dur = synth.planner_timing(0, 4.9, 10)[2]
    plan = d2dhip.FitPlan(ctx, S, K, dur, synth.default_wref(1.0, K))
    sc = synth.circle_group_scenarios(a.n_ac, a.replicas, dur, K, seed=1)
    dsc = ctx.dev(sc.reshape(a.replicas * a.n_ac, -1))
    q0 = plan.init(dsc)

    def run():
        q = q0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cost, sweeps, stats = plan.solve_groups(dsc, q, a.n_ac, max_sweeps=120, inner_iters=a.inner_iters, tol=a.tol)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, cost, sweeps, stats
    run()
    res = [run() for _ in range(a.reps)]
    dt = min(r[0] for r in res)
    _, cost, sweeps, stats = res[-1]
    n_traj = a.replicas * a.n_ac
    print(json.dumps({'metric': 'coupled multi-drone plans/sec (8-drone circle, collision rows, block Gauss-Seidel)',
                      'value': a.replicas / dt, 'unit': 'scenarios/s', 'trajectories_per_s': n_traj / dt, 'seconds': dt,
                      'replicas': a.replicas, 'n_ac': a.n_ac, 'sweeps': int(sweeps), 'inner_iters': a.inner_iters, 'tol': a.tol,
                      'last_sweep_max_rel_move': float(stats[2]), 'evaluations': float(stats[3]),
                      'sum_cost': float(stats[0]), 'all_finite': bool(np.isfinite(cost.cpu().numpy()).all()),
                      'dtype': 'f64 residual/gradient + f32 MFMA J^T J', 'data': 'synthetic (d2dhip.synth.circle_group_scenarios, seed 1)'}))


if __name__ == '__main__':
    main()
