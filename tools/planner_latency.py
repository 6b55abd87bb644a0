import sys, time, io, contextlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import numpy as np, torch
import d2d.optyplan_scenarios as sc
import single_opt_planner as sop
for backend in ('fit', 'nlp'):
    for s in (sc.exp_14, sc.exp_0, sc.exp_1):
        with contextlib.redirect_stdout(io.StringIO()):
            p = sop.Planner(s, initialize=True, backend=backend)
            x0 = p.get_initial_guess('tri')
            p.run(x0)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                p.run(x0)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(backend, s.__name__, p.num_nodes, 'nodes: Planner.run %.2f ms (best of 5)' % (1e3 * min(ts)), 'status', p.info.get('status'), 'cost', p.info.get('obj_val'))
