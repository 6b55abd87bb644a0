#!/usr/bin/env python3
"""The collocation-NLP backend at batch scale: B perturbed copies of the reference's exp_14 (121 nodes, hard bounds) in one
d2d_nlp_solve launch.  python tools/bench_nlp.py [B ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    sys.path.insert(0, _p)
import numpy as np


def problems(B, N=121, seed=0):
    import d2dhip as D
    import d2d.opty_utils as d2ou
    rng = np.random.default_rng(seed)
    rows = np.zeros((B, D.SCEN_STRIDE)); W = np.zeros((B, 5, N))
    for b in range(B):
        p0 = np.array([-49.98, -58.14, 2.22]) + rng.normal(0, [3., 3., 0.1]); p1 = np.array([75., 40., 0.]) + rng.normal(0, [3., 3., 0.1])
        r = rows[b]
        r[D.SC_X0:D.SC_X0 + 3] = p0; r[D.SC_X1:D.SC_X1 + 3] = p1
        r[D.SC_VSP], r[D.SC_KV], r[D.SC_KPHI], r[D.SC_S] = 12., 1., 0., 1. / N
        r[D.SC_PHIMAX] = np.deg2rad(40.); r[D.SC_VMIN], r[D.SC_VMAX] = 9., 15.
        r[D.SC_XMIN], r[D.SC_XMAX], r[D.SC_YMIN], r[D.SC_YMAX] = -150, 150, -150, 150
        import contextlib, io
        x, y, psi, phi, v = d2ou.triangle(p0[:2], p1[:2], 12., 12.0, N, go_left=-1.)
        W[b] = np.stack([x, y, psi, phi, v], 0)
    return rows, W


def main():
    import torch, d2dhip
    ctx = d2dhip.Context(0)
    for B in [int(x) for x in sys.argv[1:]] or [64, 4096]:
        rows, W0 = problems(B)
        dsc = ctx.dev(rows)
        best = 1e30
        for rep in range(2):
            W = ctx.dev(np.ascontiguousarray(W0))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = ctx.nlp_solve(dsc, W, 0.1)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy()
        print(json.dumps({'B': B, 'seconds': best, 'problems_per_s': B / best, 'converged_frac': float((st == 1).mean()),
                          'mean_newton_steps': float(it.mean()), 'max_newton_steps': int(it.max()), 'max_feas': float(out['feas'].max().item()),
                          'mean_cost': float(out['cost'].mean().item())}), flush=True)


if __name__ == '__main__':
    main()
