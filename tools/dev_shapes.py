import os, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/drone-sim-python_amd')
import numpy as np, torch, d2dhip
from oracle import fit as F
ctx = d2dhip.Context(0)
for (S,K) in ((4,50),(3,40),(5,64)):
    dur=(K-1)/10.0; s=0.1/K
    p = d2dhip.FitPlan(ctx, S, K, dur, (0.02**2, s*5.0, s/F.G_ACC**2), kernel=os.environ.get('KERNEL', 'auto'))
    sc = F.set_scale(F.synth_scenarios(4096, seed=3), 0.1, K)
    dsc = ctx.dev(sc); q0 = p.init(dsc)
    best=1e9
    for r in range(3):
        q=q0.clone(); torch.cuda.synchronize(); t0=time.perf_counter(); cost,it,st,_ = p.solve(dsc,q); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    print(os.environ.get('KERNEL','auto'), 'S',S,'K',K,p.kernel, '%.2f ms'%(best*1e3), '%.0f k fits/s'%(4096/best/1e3), 'mean cost %.8f'%cost.mean().item(), 'conv', (st==1).float().mean().item(), 'iters', it.float().mean().item())
    p.close()
