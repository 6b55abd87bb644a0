#!/usr/bin/env python3
"""The collocation backend over the reference's single-aircraft scenario catalogue (d2d.optyplan_scenarios.scens, every case):
status, Newton steps, cost, feasibility -- a robustness survey of d2d_nlp_solve.  python tools/nlp_catalogue.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import numpy as np


def main():
    import d2d.optyplan_scenarios as sc
    import single_opt_planner as sop
    # the reference's sweeps mutate exp_0 (exp_0_1: t1, exp_0_2: wind, exp_6: p0 / p1) and every exp_0-based scenario sees it:
    # restore exp_0 before each scenario, so that every scenario is surveyed as if it were the first one run
    keep = {k: getattr(sc.exp_0, k) for k in ('t1', 'wind', 'p0', 'p1')}
    for s in sc.scens:
        for k, v in keep.items():
            setattr(sc.exp_0, k, v)
        for case in range(s.ncases):
            s.set_case(case)
            try:
                p = sop.Planner(s, initialize=True, backend='nlp')
                t0 = time.perf_counter()
                p.run(p.get_initial_guess(getattr(s, 'initial_guess', 'tri')))
                dt = time.perf_counter() - t0
                print(json.dumps({'scen': s.name, 'case': case, 'nodes': p.num_nodes, 'status': p.info['status'], 'iters': p.info['iters'],
                                  'cost': p.info['obj_val'], 'feas': p.info['feas'], 'seconds': round(dt, 3)}), flush=True)
            except Exception as e:                      # noqa: BLE001 -- a survey: report and go on
                print(json.dumps({'scen': s.name, 'case': case, 'error': f'{type(e).__name__}: {e}'[:200]}), flush=True)


if __name__ == '__main__':
    main()
