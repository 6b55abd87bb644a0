#!/usr/bin/env python3
"""Generates tests/golden/fit_scipy_variants.npz: the CPU arbiter (scipy.optimize.least_squares(method='lm', analytic Jacobian, tol
1e-15) on oracle/fit.py's residuals from the 'tri' start -- the same recipe as make_fit_scipy_golden.py, whose solver it imports) on
the scenario families whose rows take the rarer branches of the sample evaluation, and on the other sample counts of the knot kernel:

  wind_*, bankmax_*, box3_*   256 scenarios each of d2dhip.synth.variant_scenarios (K = 50): constant wind; CostBank max mode; an x
                               box with a third obstacle
  k64_*, k40_*, k57_*          128 scenarios each at K = 64 / 40 / 57 nodes (the knot kernel's general instantiation, a short and an
                               odd horizon), seed 5, chord range scaled with the duration

Keys per family as in fit_scipy_bench1024.npz (cost, z, nfev, grad_left, qstar, cstar, star_ok, scen_sha256).  CostBank's max mode is
only piecewise smooth (the row kept is the arg-max sample): `star_ok` may be False where the minimum sits on a kink.  Numbers only.
Run here (CPU, ~1 min on 8 cores):  python tests/golden/make_fit_scipy_variants_golden.py"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fit_scipy_golden as M     # noqa: E402  (puts the repo root and drone-sim-python_amd/ on the path)
import bench                          # noqa: E402
from oracle import fit as F           # noqa: E402

OUT = os.path.join(HERE, 'fit_scipy_variants.npz')
N_VAR, N_K = 256, 128
OTHER_K = (64, 40, 57)


def other_k_scenarios(K2, B):
    from d2dhip import synth
    dur = synth.planner_timing(0, (K2 - 1) / 10.0, 10)[2]
    return dur, synth.default_wref(0.1, K2), synth.synth_scenarios(B, seed=5, obj_scale=0.1, K=K2, dist_range=(30. * dur / 4.9, 55. * dur / 4.9))


def main():
    from d2dhip import synth
    out = {}
    with mp.get_context('fork').Pool(os.cpu_count()) as pool:
        dur, wref = bench._plan_consts()
        basis = F.FitBasis(bench.S_, bench.K, dur, wref)
        for kind in synth.VARIANTS:
            sc = synth.variant_scenarios(kind, N_VAR, K=bench.K)
            out.update({f'{kind}_{k}': v for k, v in M.solve(basis, sc, pool).items()})
            out[f'{kind}_scen_sha256'] = M.sha(sc)
        for K2 in OTHER_K:
            dur2, wref2, sc = other_k_scenarios(K2, N_K)
            out.update({f'k{K2}_{k}': v for k, v in M.solve(F.FitBasis(bench.S_, K2, dur2, wref2), sc, pool).items()})
            out[f'k{K2}_scen_sha256'] = M.sha(sc)
    np.savez_compressed(OUT, **out)
    for p in list(synth.VARIANTS) + [f'k{k}' for k in OTHER_K]:
        print(p, 'n', len(out[p + '_cost']), 'star_ok', out[p + '_star_ok'].mean(), 'grad_left max', out[p + '_grad_left'].max(), 'nfev mean', out[p + '_nfev'].mean())


if __name__ == '__main__':
    main()
