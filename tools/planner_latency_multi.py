#!/usr/bin/env python3
"""One multi-aircraft plan through multi_opt_planner.Planner.run on the fit backend (the reference's 07_multioptyplan scenarios:
50 Hz horizons of 211 .. 526 nodes, 1 .. 4 aircraft): best of 5."""
import sys, time, io, contextlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import numpy as np, torch
import d2d.multioptyplan_scenarios as ms
import multi_opt_planner as mop
for name in sys.argv[1:] or ['exp_1', 'exp_2', 'exp_3', 'exp_5', 'gvf_trial_3ac']:
    scen = getattr(ms, name)
    with contextlib.redirect_stdout(io.StringIO()):
        p = mop.Planner(scen, initialize=True, backend='fit')
        p.run()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            p.run()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    info = getattr(p, 'info', {})
    print(f'{name}: {len(scen.p0s)} aircraft x {p.num_nodes} nodes: Planner.run {1e3 * min(ts):.2f} ms (best of 5)", '
          f'sweeps {info.get("sweeps")}, cost {info.get("obj_val")}', flush=True)
