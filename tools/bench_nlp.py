#!/usr/bin/env python3
"""The collocation-NLP backend at batch scale: B perturbed copies of the reference's exp_14 (121 nodes, hard bounds) in one
d2d_nlp_solve launch.  python tools/bench_nlp.py [B ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import numpy as np


def main():
    import torch, d2dhip
    from d2dhip import synth
    ctx = d2dhip.Context(0)
    for B in [int(x) for x in sys.argv[1:]] or [64, 4096]:
        rows, W0, h = synth.nlp_problems(B)
        dsc = ctx.dev(rows)
        best = 1e30
        for rep in range(2):
            W = ctx.dev(np.ascontiguousarray(W0))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = ctx.nlp_solve(dsc, W, h)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy()
        print(json.dumps({'B': B, 'seconds': best, 'problems_per_s': B / best, 'converged_frac': float((st == 1).mean()),
                          'mean_newton_steps': float(it.mean()), 'max_newton_steps': int(it.max()), 'max_feas': float(out['feas'].max().item()),
                          'mean_cost': float(out['cost'].mean().item())}), flush=True)


if __name__ == '__main__':
    main()
