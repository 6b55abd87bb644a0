#!/usr/bin/env python3
"""Time of the plant step alone (d2d_step, one step for many drones) against a full guidance step (GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np                               # noqa: E402
import torch                                     # noqa: E402
import d2dhip                                    # noqa: E402
ctx = d2dhip.Context(0)
N = 1 << 22
rng = np.random.default_rng(0)
X = ctx.dev(np.stack([rng.uniform(-50, 50, N), rng.uniform(-50, 50, N), rng.uniform(-3, 3, N), rng.uniform(-0.6, 0.6, N), rng.uniform(9, 15, N)]))
for amp in (0.6, 1.2, 1.4):
    U = ctx.dev(np.stack([rng.uniform(-amp, amp, N), rng.uniform(9, 15, N)]))
    for _ in range(2):
        ctx.step(X, U)
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(10):
        ctx.step(X, U)
    e1.record(ctx.stream); ctx.sync()
    t = e0.elapsed_time(e1) * 1e-3 / 10
    print(f'plant step alone, |phi_c| <= {amp}: {t*1e6:.1f} us for {N} drones = {N/t/1e9:.2f} G drone-steps/s '
          f'({t/(N/65536)*1e6:.2f} us per 65536 drones)', flush=True)
