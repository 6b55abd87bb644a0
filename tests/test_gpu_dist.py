"""Multi-GPU path with the real HIP plan: two ranks (gloo backend, both on cuda:0) through bench.py's launcher, solve_sharded and the
convergence exchange -- started by tests/conftest.py before this process touched the GPU -- against single-rank solves of the same
two shards in this process.  (RCCL itself needs two GPUs: the driver's 8-GPU bench is its first run.)"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_equals_two_single_rank_solves(rehearsal):
    import d2dhip
    import bench
    assert rehearsal, 'the rehearsal child was not started (conftest.pytest_sessionstart)'
    assert rehearsal['rc'] == 0, rehearsal['err']
    line = json.loads(rehearsal['out'].strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['scaling'] == 'weak'
    assert line['converged_frac'] + line['stalled_frac'] >= 0.999
    assert line['config']['parallelism'] == 'trajectory-sharded x2'
    B = 4096
    ctx = d2dhip.Context(0)
    dur, wref = bench._plan_consts()
    plan = d2dhip.FitPlan(ctx, bench.S_, bench.K, dur, wref)
    tot_cost = 0.0
    try:
        for rank in range(2):
            dsc = ctx.dev(bench.bench_scenarios(B, rank))
            q = plan.init(dsc)
            cost, iters, status, stats = plan.solve(dsc, q, max_iter=150, check_every=200)
            d = np.load(os.path.join(ROOT, 'gpurun_out', f'rehearsal_rank{rank}.npz'))
            # the same fits, the same kernel, whichever process ran them: identical bits
            assert np.array_equal(d['cost'], cost.cpu().numpy()) and np.array_equal(d['iters'], iters.cpu().numpy())
            assert np.array_equal(d['q'], q.cpu().numpy())
            tot_cost += float(cost.sum().item())
        assert abs(line['mean_cost'] * 2 * B - tot_cost) <= 1e-9 * tot_cost          # the all-reduced statistic
        assert line['value'] > 0 and line['ms_per_step'] > 0
    finally:
        plan.close(); ctx.close()
