#!/usr/bin/env python3
"""Wall time of the on-device three-phase chain (full_sim.full_sim_phases_batch) against the number of formations.
  python tools/bench_chain.py [n_form ...]        (GPU box)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np                               # noqa: E402
import torch                                     # noqa: E402
import full_sim as fs                            # noqa: E402
import multi_opt_planner as mop                  # noqa: E402

n_ac, r, v = 4, 60, 15
c = np.array([[0, -20], [25, -20], [25, -100], [0, -100]], float)
X1_f = np.array(((0, 40, 0, 0, 12), (25, 40, 0, 0, 12), (25, -40, 0, 0, 12), (0, -40, 0, 0, 12)), float)
X2_f = np.array(((75, 40, 0, 0, 12), (100, 40, 0, 0, 12), (100, -40, 0, 0, 12), (75, -40, 0, 0, 12)), float)
T3 = 120
th = np.linspace(0, 2 * np.pi, T3)
time_3 = np.arange(T3) * 0.1
x3 = X2_f[None, :, 0] + 30 * np.sin(th)[:, None]; y3 = X2_f[None, :, 1] + 30 * (1 - np.cos(th))[:, None]
rng = np.random.default_rng(0)
for n_form in [int(a) for a in sys.argv[1:]] or [1, 64, 1024, 4096]:
    # formations differ by a small offset of their start states (sigma 1 m), so that they are not clones
    X0 = np.tile(fs.X1_START, (n_form, n_ac, 1)); X0[:, :, :2] += rng.normal(0, 1.0, (n_form, n_ac, 2))
    cB = np.tile(c, (n_form, 1, 1))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fs.full_sim_phases_batch(cB, r, v, n_ac, X1_f, mop.trap_4, X2_f, 6, ref3=(time_3, x3, y3), t_sim_end=200.,
                                       X0=X0, t_end_1=200., record2=(), record3=())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    stop = out['phase1']['stop_row'].cpu().numpy()
    n_steps = int(stop.sum()) * n_ac + n_form * n_ac * (61 + T3 * len(out['phase3']))
    print(json.dumps({'n_form': n_form, 'drones': n_form * n_ac, 'wall_s': round(dt, 4), 'phase3_passes': len(out['phase3']),
                      'phase1_stop_rows': [int(stop.min()), int(stop.max())], 'drone_steps': n_steps,
                      'drone_steps_per_s': round(n_steps / dt), 'plan_cost_sum_mean': float(out['plan']['cost'].sum().item()) / n_form}), flush=True)
