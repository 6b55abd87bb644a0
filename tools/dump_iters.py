import sys, os
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import numpy as np, bench, d2dhip
ctx=d2dhip.Context(0)
dur,wref=bench._plan_consts()
plan=d2dhip.FitPlan(ctx,6,50,dur,wref)
dsc=ctx.dev(bench.bench_scenarios(4096))
q=plan.init(dsc)
cost,iters,status,stats=plan.solve(dsc,q,max_iter=200)
np.savez('gpurun_out/r2_d_iters.npz',iters=iters.cpu().numpy(),status=status.cpu().numpy(),cost=cost.cpu().numpy())
print(np.sort(iters.cpu().numpy())[-40:])
