#!/usr/bin/env python3
"""The contraction-only kernel (fit_jtj_kernel, d2d_fit_rows + d2d_fit_jtj) alone, HIP events.
  python tools/bench_jtj.py [B ...]        env: D2D_JTJ_GEOM="wpb,wgs" (launch geometry A/B)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import bench    # noqa: E402


def main():
    import torch
    import d2dhip
    ctx = d2dhip.Context(0)
    dur, wref = bench._plan_consts()
    plan = d2dhip.FitPlan(ctx, bench.S_, bench.K, dur, wref)
    for B in [int(x) for x in sys.argv[1:]] or [4096, 32768]:
        dsc = ctx.dev(bench.bench_scenarios(B))
        q = plan.init(dsc)
        plan.rows(dsc, q)
        for _ in range(3):
            plan.jtj(B, want_H=False)
        torch.cuda.synchronize()
        plan.profile(True)
        for _ in range(20):
            plan.jtj(B, want_H=False)
        pr = plan.profile_read()
        plan.profile(False)
        us = 1e3 * pr[6] / pr[7]
        tf = bench.ALG_FLOP_PER_EVAL * B / (us * 1e-6) / 1e12
        print(json.dumps({'B': B, 'us': us, 'tflops': tf, 'frac': tf / bench.FP32_PEAK_TFLOPS, 'geom': os.environ.get('D2D_JTJ_GEOM')}), flush=True)


if __name__ == '__main__':
    main()
