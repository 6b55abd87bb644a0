#!/usr/bin/env python3
"""Development sweep of d2d_fit_opts.mp_finish x mp_slow (GPU box): agreement with scipy on the first N bench scenarios of two ranks, time and
longest fit of 4096 fits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import multiprocessing as mp
import bench
from oracle import fit as F
from d2dhip import synth

K, S_ = 50, 6
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dur = synth.planner_timing(0, 4.9, 10)[2]; wref = synth.default_wref(0.1, K)
basis = F.FitBasis(S_, K, dur, wref)
RANKS = (0, 1)
SC = {r: synth.synth_scenarios(4096, seed=20241008, rank=r, obj_scale=0.1, K=K) for r in RANKS}


def scipy_one(a):
    r, i = a
    return bench._cpu_fit_one((basis, SC[r][i], None))


if __name__ == '__main__':
    with mp.get_context('fork').Pool(16) as pool:
        sres = {r: pool.map(scipy_one, [(r, i) for i in range(N)], chunksize=4) for r in RANKS}
    import torch, d2dhip
    ctx = d2dhip.Context(0)
    plan = d2dhip.FitPlan(ctx, S_, K, dur, wref)
    for fin, slow in ((3, 8), (3, 6), (3, 5), (3, 4), (3, 3), (2, 8), (2, 6), (2, 4), (4, 8), (3, 12)):
        line = f'mp_finish={fin} mp_slow={slow}:'
        for r in RANKS:
            dsc = ctx.dev(SC[r]); q = plan.init(dsc)
            best = 1e9
            for rep in range(3):
                qq = q.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
                cost, iters, status, stats = plan.solve(dsc, qq, max_iter=150, check_every=200, mp_finish=fin, mp_slow=slow)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            cs = np.array([x[0] for x in sres[r]]); qs = np.array([x[1] for x in sres[r]])
            cg = cost.cpu().numpy()[:N]; qg = qq.cpu().numpy()[:N]
            same = (np.abs(cg - cs) / cs <= 1e-6) & (np.abs(qg - qs).max(1) / np.abs(qs).max(1) <= 1e-6)
            line += f' rank {r}: same {same.mean():.4f} ({(~same).sum()} differ, gpu higher {((cg - cs)[~same] > 0).sum()}) {best * 1e3:.3f} ms mean iters {iters.float().mean().item():.1f} max {iters.max().item()} evals {stats[3] / 4096:.1f};'
        print(line, flush=True)
