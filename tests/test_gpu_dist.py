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


def test_rccl_entry_point_single_rank():
    """d2d_allreduce_stats: the convergence exchange as a C-ABI entry point, RCCL loaded at run time.  One GPU per box:
    a communicator of one rank -- unique id, init, the grouped sum / max / sum exchange on the context's stream, destroy."""
    import ctypes as C
    import torch
    import d2dhip
    ctx = d2dhip.Context(0)
    lib = ctx.lib
    try:
        uid = (C.c_char * 128)()
        assert lib.d2d_comm_unique_id(uid) == 0, lib.d2d_last_error()
        assert any(b != b'\x00' for b in uid)
        comm = C.c_void_p()
        assert lib.d2d_comm_create(ctx.h, uid, 0, 1, C.byref(comm)) == 0, lib.d2d_last_error()
        # d2d_comm_info answers from the communicator itself (ncclCommUserRank / ncclCommCount), not from the arguments above
        r, w = C.c_int32(-7), C.c_int32(-7)
        assert lib.d2d_comm_info(comm, C.byref(r), C.byref(w)) == 0, lib.d2d_last_error()
        assert (r.value, w.value) == (0, 1)
        stats = torch.tensor([1.5, 2.5, 3.0], dtype=torch.float64, device=ctx.device)
        assert lib.d2d_allreduce_stats(ctx.h, comm, C.c_void_p(stats.data_ptr())) == 0, lib.d2d_last_error()
        ctx.sync()
        assert stats.cpu().tolist() == [1.5, 2.5, 3.0]
        assert lib.d2d_comm_destroy(comm) == 0
        assert lib.d2d_comm_create(ctx.h, uid, 2, 2, C.byref(comm)) != 0            # rank out of range: refused with a message
        assert b'rank' in lib.d2d_last_error()
    finally:
        ctx.close()


def test_sharded_solve_through_the_c_abi_collective_one_rank():
    """The collective the nccl path of solve_sharded runs is the library's own entry point: a one-rank `nccl` process group (one
    GPU per box), StatsReducer creates a d2d_comm from the id rank 0 made (d2d_comm_unique_id, carried by torch.distributed) and
    every exchange of the solve goes through d2d_allreduce_stats -- same bits as the single-process solve; a host pointer and a
    second context's wrong use are refused with a message."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    import d2dhip
    import bench
    from d2dhip.dist import StatsReducer, solve_sharded
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29617')
    ctx = d2dhip.Context(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        red = StatsReducer(dist, ctx.device, ctx)
        assert red.comm is not None and red.rccl_ranks == 1 and red.comm.info() == (0, 1), red.collective
        assert 'd2d_allreduce_stats' in red.collective and 'ncclCommCount' in red.collective
        assert red(1.25, 7.0, 3) == (1.25, 7.0, 3) and red.running_only(5) == 5
        dur, wref = bench._plan_consts()
        plan = d2dhip.FitPlan(ctx, bench.S_, bench.K, dur, wref)
        dsc = ctx.dev(bench.bench_scenarios(512, 0))
        q1 = plan.init(dsc); q2 = q1.clone()
        c1, i1, s1, st1, glob, checks = solve_sharded(plan, dsc, q1, red, 200, 150)
        c2, i2, s2, st2 = plan.solve(dsc, q2, max_iter=150, check_every=200)
        assert torch.equal(q1, q2) and torch.equal(c1, c2) and torch.equal(i1, i2)
        assert abs(glob[0] - float(c2.sum().item())) <= 1e-12 * abs(glob[0]) and glob[2] == int((s2 != d2dhip.ST_CONVERGED).sum().item())
        host = np.zeros(3)
        assert ctx.lib.d2d_allreduce_stats(ctx.h, red.comm.h, host.ctypes.data_as(C.c_void_p)) == -1          # D2D_EINVAL
        assert b'device memory' in ctx.lib.d2d_last_error()
        plan.close()
        red.comm.close()
    finally:
        dist.destroy_process_group()
        ctx.close()
