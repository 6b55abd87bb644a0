#!/usr/bin/env python3
"""(GPU, development) What a hand-out order is worth for the collocation launch: 4096 perturbed exp_14 problems in index order, in
the order of their own measured Newton-step counts (longest first), and the rows + counts dumped for a predictability study.
  python tools/dev_nlp_order.py [B] [out.npz]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'drone-sim-python_amd')]
import torch      # noqa: E402
import d2dhip     # noqa: E402
from d2dhip import synth      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = d2dhip.Context(0)
rows, W0, h = synth.nlp_problems(B)
dsc = ctx.dev(rows); W0d = ctx.dev(W0)


def run(order=None, n=5):
    ts = []
    for _ in range(n):
        W = W0d.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ctx.nlp_solve(dsc, W, h, order=order)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), out, W


t_idx, o0, W_idx = run()
it = o0['iters'].cpu().numpy(); st = o0['status'].cpu().numpy()
print(f'index order: {t_idx*1e3:.2f} ms = {B/t_idx/1e3:.1f} k problems/s; steps mean {it.mean():.1f} p50 {np.median(it):.0f} p90 {np.percentile(it,90):.0f} p99 {np.percentile(it,99):.0f} max {it.max()}')
for s in np.unique(st):
    print('  status', s, (st == s).sum(), 'steps mean', it[st == s].mean())
order = torch.from_numpy(np.argsort(-it, kind='stable').astype(np.int32)).to(ctx.device)
t_true, o1, W_true = run(order)
print(f'true order : {t_true*1e3:.2f} ms = {B/t_true/1e3:.1f} k problems/s')
assert torch.equal(o1['iters'], o0['iters']) and torch.equal(o1['cost'], o0['cost']) and torch.equal(W_idx, W_true)
slots = 2048
print('avg load per slot (steps):', it.sum() / slots, ' longest:', it.max())
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], rows=rows, iters=it, status=st, W0=W0[:, :, [0, -1]])
