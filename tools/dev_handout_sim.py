#!/usr/bin/env python3
"""Development: makespan of hand-out policies of the fused LM kernel on measured trial counts (tools/dev_dump_iters.py), in trial-times.
A SIMD holds two waves; a wave whose SIMD partner is idle runs LONE_SPEEDUP x faster.
  fifo       index order, run to completion (the default)
  lpt        longest first (the order hint)
  ps(Q)      round robin: a fit yields after Q trials while others wait
  pin(Q, A)  round robin for fits younger than A trials, fits of age >= A never yield"""
import sys, heapq
import numpy as np
it = np.load(sys.argv[1])
SLOTS = 2048
LONE = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0      # time per trial when alone on the SIMD (1.0 = no effect)


def simulate(n, order, Q=None, A=None):
    """event simulation; slots pair up on SIMDs (slot s, s ^ 1); speeds re-evaluated at every event (piecewise constant)."""
    from collections import deque
    queue = deque(order)
    rem = n.astype(float).copy()          # remaining trials
    age = np.zeros(len(n))
    cur = [-1] * SLOTS                    # fit on slot
    seg = [0.0] * SLOTS                   # trials left in the current quantum
    t = 0.0
    for s in range(SLOTS):
        if queue:
            f = queue.popleft(); cur[s] = f
            seg[s] = rem[f] if Q is None else min(rem[f], Q)
    while True:
        act = [s for s in range(SLOTS) if cur[s] >= 0]
        if not act: break
        # time per trial of each active slot
        dt = np.inf; rate = {}
        for s in act:
            r = 1.0 if cur[s ^ 1] >= 0 else 1.0 / LONE
            rate[s] = r
            dt = min(dt, seg[s] / r)
        t += dt
        for s in act:
            d = dt * rate[s]
            f = cur[s]
            seg[s] -= d; rem[f] -= d; age[f] += d
        for s in act:
            if seg[s] <= 1e-9:
                f = cur[s]
                if rem[f] <= 1e-9:
                    cur[s] = -1
                else:
                    # quantum over: yield only when somebody waits and the fit is still young
                    if queue and (A is None or age[f] < A):
                        queue.append(f); cur[s] = -1
                    else:
                        seg[s] = min(rem[f], Q); continue
                if queue:
                    g = queue.popleft(); cur[s] = g
                    seg[s] = rem[g] if Q is None else min(rem[g], Q)
    return t


for key in ('iters_4096',):
    n = it[key]
    B = len(n)
    print(f'{key}: mean {n.mean():.1f} max {n.max()}  work / slots = {n.sum() / SLOTS:.1f} trial-times, lone factor {LONE}')
    print('  fifo           ', round(simulate(n, list(range(B))), 1))
    print('  lpt            ', round(simulate(n, list(np.argsort(-n, kind="stable"))), 1))
    for Q in (4, 8, 16):
        print(f'  ps({Q})          ', round(simulate(n, list(range(B)), Q=Q), 1))
    for Q, A in ((4, 16), (4, 24), (8, 24), (8, 32), (8, 40), (8, 48), (16, 32), (16, 48), (4, 32), (4, 40)):
        print(f'  pin({Q},{A})      ', round(simulate(n, list(range(B)), Q=Q, A=A), 1))
