#!/usr/bin/env python3
"""Development check of the MINPACK mode of the fused LM kernel (GPU box): GPU vs oracle.solve_minpack (fp32 Hessian / Cholesky)
vs scipy least_squares('lm') on the first N bench scenarios, then timings of 4096 fits in both modes with / without slicing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import multiprocessing as mp
import bench
from oracle import fit as F
from d2dhip import synth

K, S_ = 50, 6
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dur = synth.planner_timing(0, 4.9, 10)[2]; wref = synth.default_wref(0.1, K)
basis = F.FitBasis(S_, K, dur, wref)
sc_all = synth.synth_scenarios(4096, seed=20241008, rank=0, obj_scale=0.1, K=K)


def scipy_one(i):
    return bench._cpu_fit_one((basis, sc_all[i], None))


def oracle_one(a):
    i, fin = a
    q, c, it, st, info = F.solve_minpack(basis, sc_all[i], finish=fin, hess_dtype=np.float32, chol_dtype=np.float32)
    return c, q, it, st, info['nfac']


if __name__ == '__main__':
    with mp.get_context('fork').Pool(16) as pool:
        sres = pool.map(scipy_one, range(N), chunksize=2)
        ores = {fin: pool.map(oracle_one, [(i, fin) for i in range(N)], chunksize=2) for fin in (3, 0)}
    import torch, d2dhip
    ctx = d2dhip.Context(0)
    plan = d2dhip.FitPlan(ctx, S_, K, dur, wref)
    dsc = ctx.dev(sc_all)
    q0 = plan.init(dsc)
    cs = np.array([r[0] for r in sres]); qs = np.array([r[1] for r in sres])
    for fin in (3, 0):
        q = q0.clone()
        cost, iters, status, stats = plan.solve(dsc, q, max_iter=400, mode=d2dhip.MODE_MINPACK, mp_finish=fin)
        cg = cost.cpu().numpy()[:N]; qg = q.cpu().numpy()[:N]; itg = iters.cpu().numpy()[:N]
        co = np.array([r[0] for r in ores[fin]]); qo = np.array([r[1] for r in ores[fin]]); ito = np.array([r[2] for r in ores[fin]])
        same_s = (np.abs(cg - cs) / cs <= 1e-6) & (np.abs(qg - qs).max(1) / np.abs(qs).max(1) <= 1e-6)
        same_o = (np.abs(cg - co) / co <= 1e-6) & (np.abs(qg - qo).max(1) / np.abs(qo).max(1) <= 1e-6)
        print(f'finish={fin}: GPU vs scipy same {same_s.mean():.4f}; GPU vs oracle same {same_o.mean():.4f}; iters equal {np.mean(itg == ito):.3f} '
              f'(|d|<=2: {np.mean(np.abs(itg - ito) <= 2):.3f}); mean iters gpu {itg.mean():.1f} oracle {ito.mean():.1f}; status {np.bincount(status.cpu().numpy())}; '
              f'all-batch mean iters {iters.float().mean().item():.1f} max {iters.max().item()} evals/fit {stats[3] / 4096:.1f}', flush=True)
    for name, kw in [('minpack f3 slice16', dict(mode=0, mp_finish=3, slice=16)), ('minpack f3 slice0', dict(mode=0, mp_finish=3, slice=0)),
                     ('minpack f3 slice8', dict(mode=0, mp_finish=3, slice=8)), ('minpack f3 slice32', dict(mode=0, mp_finish=3, slice=32)),
                     ('minpack pure slice16', dict(mode=0, mp_finish=0, slice=16)), ('minpack pure slice0', dict(mode=0, mp_finish=0, slice=0)),
                     ('fast slice16', dict(mode=1, slice=16)), ('fast slice0', dict(mode=1, slice=0))]:
        for B in (4096, 32768):
            d = ctx.dev(synth.synth_scenarios(B, seed=20241008, rank=0, obj_scale=0.1, K=K))
            qq0 = plan.init(d)
            best = 1e9
            for rep in range(4):
                q = qq0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
                cost, iters, status, stats = plan.solve(d, q, max_iter=400, check_every=400, **kw)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            print(f'{name:22s} B={B}: {best * 1e3:.3f} ms  {B / best / 1e6:.3f} M fits/s  mean iters {iters.float().mean().item():.1f} max {iters.max().item()} conv {(status == 1).float().mean().item():.4f} mean cost {cost.mean().item():.6f}', flush=True)
