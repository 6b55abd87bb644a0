import sys, os, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import numpy as np, torch
import d2dhip
from d2dhip import synth
ctx=d2dhip.Context(0)
K,S=50,6
dur=synth.planner_timing(0,4.9,10)[2]
plan=d2dhip.FitPlan(ctx,S,K,dur,synth.default_wref(0.1,K))
B=int(os.environ.get('BATCH',4096))
sc=ctx.dev(synth.synth_scenarios(B))
q=plan.init(sc)
cost=ctx.empty(B); g=ctx.empty(B,48); H=torch.zeros(B,48,48,dtype=torch.float32,device=ctx.device)
import ctypes as C
def ev(withH=True):
    d2dhip._check(ctx.lib.d2d_fit_eval(ctx.h,plan.h,B,d2dhip._ptr(sc),d2dhip._ptr(q),d2dhip._ptr(cost),d2dhip._ptr(g),d2dhip._ptr(H) if withH else None))
for _ in range(3): ev()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
n=20
e0.record(ctx.stream)
for _ in range(n): ev()
e1.record(ctx.stream); torch.cuda.synchronize()
t=e0.elapsed_time(e1)/n*1e3
e0.record(ctx.stream)
for _ in range(n): ev(False)
e1.record(ctx.stream); torch.cuda.synchronize()
t2=e0.elapsed_time(e1)/n*1e3
print('ABLATE',os.environ.get('D2D_FIT_ABLATE','0'),'B',B,'eval+prep+sym us',round(t,1),'noH (prep+eval p1,p2) us',round(t2,1))
