"""CPU, world_size 2 over gloo: trajectory sharding + the convergence all-reduce of
d2dhip.dist, driven with a stand-in plan that runs the oracle's LM on the shard (the HIP
plan has the same begin / iterate / finish surface)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from d2dhip.dist import shard_bounds, StatsReducer, solve_sharded
from oracle import fit as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for total in (1, 7, 8, 4096, 262144, 4097):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


class OraclePlan:
    """begin / iterate / finish with the oracle's LM; each trajectory advances `n_iters` solves per call."""

    def __init__(self, basis):
        self.b = basis

    def begin(self, B):
        self.state = None

    def iterate(self, scen, q, n_iters, max_iter=200, **tol):
        if self.state is None:
            self.state = [dict(done=False, it=0) for _ in range(scen.shape[0])]
        running = 0
        for i, st in enumerate(self.state):
            if st['done']:
                continue
            st['it'] += n_iters
            qi, c, it, status = F.lm_solve(self.b, scen[i].numpy(), max_iter=min(st['it'], max_iter))
            q[i] = torch.from_numpy(qi)
            st.update(cost=c, iters=it, status=status)
            if status != F.ST_MAXITER or st['it'] >= max_iter:
                st['done'] = True
            else:
                running += 1
        return running

    def finish(self, scen, q):
        cost = torch.tensor([s['cost'] for s in self.state], dtype=torch.float64); iters = torch.tensor([s['iters'] for s in self.state])
        status = torch.tensor([s['status'] for s in self.state])
        gmax = max(np.abs(F.eval_normal(self.b, scen[i].numpy(), q[i].numpy())[1]).max() for i in range(len(self.state)))
        notconv = sum(1 for s in self.state if s['status'] not in (F.ST_CONVERGED, F.ST_STALLED))
        return cost, iters, status, np.array([float(cost.sum()), gmax, notconv, 0.0])


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    K, S_ = 50, 6
    dur = F.planner_timing(0, 4.9, 10)[2]
    s = 0.1 / K
    basis = F.FitBasis(S_, K, dur, (0.02 ** 2, s * 5.0, s / F.G_ACC ** 2))
    sc_all = F.set_scale(F.synth_scenarios(total, seed=3), 0.1, K)
    lo, hi = shard_bounds(total, rank, world)
    scen = torch.from_numpy(sc_all[lo:hi]); q = torch.zeros(hi - lo, 2 * basis.nq, dtype=torch.float64)
    cost, iters, status, stats, glob, checks = solve_sharded(OraclePlan(basis), scen, q, StatsReducer(dist, 'cpu'),
                                                            check_every=16, max_iter=200)
    out[rank] = dict(lo=lo, hi=hi, cost=cost.numpy(), glob=glob, checks=checks, local_sum=stats[0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_sharded_solve():
    world, total = 2, 5
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert (r0['lo'], r0['hi'], r1['lo'], r1['hi']) == (0, 3, 3, 5)
    # both ranks leave the loop together and agree on the global statistics
    assert r0['checks'] == r1['checks']
    assert r0['glob'] == r1['glob']
    np.testing.assert_allclose(r0['glob'][0], r0['local_sum'] + r1['local_sum'], rtol=1e-12)
    assert r0['glob'][2] == 0
    # the sharded result equals the unsharded one
    K = 50
    dur = F.planner_timing(0, 4.9, 10)[2]
    s = 0.1 / K
    basis = F.FitBasis(6, K, dur, (0.02 ** 2, s * 5.0, s / F.G_ACC ** 2))
    sc_all = F.set_scale(F.synth_scenarios(total, seed=3), 0.1, K)
    ref = np.array([F.lm_solve(basis, sc_all[i])[1] for i in range(total)])
    np.testing.assert_allclose(np.concatenate([r0['cost'], r1['cost']]), ref, rtol=1e-9)


class _FakeComm:
    """stand-in for d2dhip.Comm over gloo: same surface (info / allreduce_stats / close)"""

    def __init__(self, dist, rank, world, log):
        self.dist, self.rank, self.world, self.log = dist, rank, world, log

    def info(self):
        return self.rank, self.world

    def allreduce_stats(self, t):
        self.log.append('abi')
        a, b, c = t[0:1].clone(), t[1:2].clone(), t[2:3].clone()
        self.dist.all_reduce(a, op=self.dist.ReduceOp.SUM); self.dist.all_reduce(b, op=self.dist.ReduceOp.MAX); self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM)
        t[0], t[1], t[2] = a[0], b[0], c[0]

    def close(self):
        self.log.append('closed')


class _FakeCtx:
    """stand-in context: fails where `case` says so, on the rank it says"""
    device = 'cpu'

    def __init__(self, dist, rank, case, log):
        self.dist, self.rank, self.case, self.log = dist, rank, case, log

    def comm_available(self):
        # d2d_comm_available: local, no communication
        return 'librccl.so.1: cannot open shared object file' if (self.case == 'rccl_missing_on_rank1' and self.rank == 1) else None

    def comm_unique_id(self):
        if self.case == 'uid_fails_on_rank0':
            raise OSError('librccl.so.1 cannot be loaded')
        return bytes(range(128))

    def comm_create(self, uid, rank, world):
        assert uid == bytes(range(128))
        assert self.case != 'rccl_missing_on_rank1', 'comm_create entered although a rank reported RCCL missing'
        if self.case == 'create_fails_on_rank1' and rank == 1:
            raise RuntimeError('ncclCommInitRank failed')
        if self.case == 'all_fine':
            self.dist.barrier()          # like ncclCommInitRank, the real one returns only when every rank has joined
        return _FakeComm(self.dist, rank, world, self.log)


def _agree_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = {}
    for case in ('rccl_missing_on_rank1', 'uid_fails_on_rank0', 'create_fails_on_rank1', 'all_fine'):
        log = []
        red = StatsReducer(dist, 'cpu', _FakeCtx(dist, rank, case, log), force_comm=True)
        # the exchange after the decision must be the SAME collective on both ranks: it completes and agrees
        tot = red(1.0 + rank, 10.0 * (rank + 1), 3 + rank)
        run = red.running_only(rank)
        res[case] = dict(uses_comm=red.comm is not None, collective=red.collective, rccl_ranks=red.rccl_ranks, tot=tot, run=run, log=log)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_comm_fallback_is_decided_collectively():
    """a d2d_comm that only SOME ranks could create must not leave the ranks on different collectives (hang).  Rank 0
    failing to make the id, and rank 1 failing to join, both end with every rank on torch.distributed; no failure ends on the d2d_comm."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_agree_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    # a rank that cannot load RCCL at all must keep EVERY rank out of comm_create (itself a collective): agreed by a preflight
    for case, uses in (('rccl_missing_on_rank1', False), ('uid_fails_on_rank0', False), ('create_fails_on_rank1', False), ('all_fine', True)):
        a, b = out[0][case], out[1][case]
        assert a['uses_comm'] == b['uses_comm'] == uses, (case, a['collective'], b['collective'])
        assert a['tot'] == b['tot'] == (3.0, 20.0, 7) and a['run'] == b['run'] == 1
        assert ('abi' in a['log']) == uses and ('abi' in b['log']) == uses
        if uses:
            assert a['rccl_ranks'] == b['rccl_ranks'] == 2 and 'd2d_allreduce_stats' in a['collective']
        else:
            assert 'd2d_comm not used' in a['collective'] and 'd2d_comm not used' in b['collective']
    # rank 0 had created its communicator when rank 1 failed: it was closed, not leaked or used
    assert out[0]['create_fails_on_rank1']['log'] == ['closed']


def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks itself (before touching any GPU) and
    returns their exit code; D2D_BENCH_SPAWN_TEST makes every rank report and leave before the GPU part."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['D2D_BENCH_SPAWN_TEST'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    recs = [json.loads(m) for m in re.findall(r'\{[^{}]*\}', r.stdout)]      # (two ranks share the pipe: their lines may run together)
    assert sorted(x['rank'] for x in recs) == [0, 1] and all(x['world'] == 2 and x['gpus'] == 2 for x in recs)
