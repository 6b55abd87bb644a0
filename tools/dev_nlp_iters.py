import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
ctx = d2dhip.Context(0)
rows, W0, h = synth.nlp_problems(4096)
out = ctx.nlp_solve(ctx.dev(rows), ctx.dev(np.ascontiguousarray(W0)), h)
np.savez(sys.argv[1], iters=out['iters'].cpu().numpy(), status=out['status'].cpu().numpy(), rows=rows, cost=out['cost'].cpu().numpy())
