#!/usr/bin/env python3
"""CPU counterpart of tools/nlp_catalogue.py: the ORACLE's solver (oracle/nlp.py) over the single-aircraft scenario catalogue,
for solver development (test infrastructure: imports oracle/).  python tools/nlp_oracle_catalogue.py [name-substring ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import numpy as np


def problems(filt=()):
    import contextlib, io
    import d2d.optyplan_scenarios as sc
    import single_opt_planner as sop
    from oracle import nlp
    keep = {k: getattr(sc.exp_0, k) for k in ('t1', 'wind', 'p0', 'p1')}
    for s in sc.scens:
        for k, v in keep.items():
            setattr(sc.exp_0, k, v)
        for case in range(s.ncases):
            s.set_case(case)
            tag = f'{s.__name__}[{case}]'
            if filt and not any(f in tag for f in filt):
                continue
            with contextlib.redirect_stdout(io.StringIO()):
                p = sop.Planner(s, initialize=True, backend='nlp')
                rows, _ = p.prob._rows()
                x0 = p.get_initial_guess(getattr(s, 'initial_guess', 'tri'))
            yield tag, nlp.problem_from_row(rows[0], p.num_nodes, p.time_step), nlp.from_free(x0, p.num_nodes)


def main():
    from oracle import nlp
    for tag, pb, W0 in problems(sys.argv[1:]):
        t0 = time.perf_counter()
        W, info = nlp.solve(pb, W0)
        print(json.dumps({'scen': tag, 'nodes': pb.N, 'status': info['status'], 'steps': info['inner'], 'outer': info['outer'], 'cost': info['cost'],
                          'feas': info['feas'], 'rho': info['rho'], 'seconds': round(time.perf_counter() - t0, 2)}), flush=True)


if __name__ == '__main__':
    main()
