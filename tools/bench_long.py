#!/usr/bin/env python3
"""Long-horizon fits (K > 64) on fit_lm_long_kernel: fits/s against the node count, tables in the LDS (default choice) and in
global memory (LONG_TABLES=0 in the tool's environment -> d2d_fit_plan_opts.long_tables).  python tools/bench_long.py [K ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import torch, d2dhip, bench
ctx = d2dhip.Context(0)
for K in [int(x) for x in sys.argv[1:]] or [101, 121, 151, 201, 301]:
    r = bench.long_horizon_record(ctx, torch, d2dhip, B=4096, K=K, t1=(K - 1) / 10.0, long_tables=int(os.environ.get('LONG_TABLES', '-1')))
    print(json.dumps({'K': K, 'tables': os.environ.get('LONG_TABLES', 'default'), 'fits_per_s': round(r['value']), 'ms': round(r['ms_per_step'], 2),
                      'converged_frac': r['converged_frac'], 'mean_iters': round(r['mean_iters'], 1)}), flush=True)
