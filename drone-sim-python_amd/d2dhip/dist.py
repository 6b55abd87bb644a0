"""Trajectory-sharded fits over the GPUs of one node: one process per GPU, no data-path
collective (trajectories are independent); the only exchange is the all-reduce of the
convergence statistics [sum cost, max |J^T r|, trajectories still running] every
`check_every` Levenberg-Marquardt iterations (RCCL over xGMI with the `nccl` backend; the
CPU tests drive the same code over `gloo`)."""
import numpy as np


def shard_bounds(total, rank, world):
    """Contiguous split of `total` trajectories; the first total % world ranks get one more."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class StatsReducer:
    """The convergence exchange.  `device` is where the backend wants its buffers (cuda:<local_rank> for nccl / RCCL, cpu for
    gloo).  Everything is allocated once: a pinned host staging tensor, the device send buffer and the gathered [world, 3]
    tensor; a call is one host write, one collective, one read-back -- no tensors are built per call.

    ctx (a d2dhip.Context) with the `nccl` backend: the collective that runs is the library's own C-ABI entry point,
    d2d_allreduce_stats on a d2d_comm (include/d2d.h: ONE grouped RCCL exchange of sum / max / sum on the context's stream) --
    rank 0 creates the 128-byte id (d2d_comm_unique_id) and torch.distributed only carries it to the other ranks.  If the
    communicator cannot be created (e.g. librccl cannot be loaded) the exchange stays on torch.distributed and `self.collective`
    says why.  `self.rccl_ranks` = the rank count the d2d_comm reports."""

    def __init__(self, dist=None, device='cpu', ctx=None, force_comm=False):
        self.dist = dist
        self.comm = None
        self.rccl_ranks = None
        self.collective = 'none (one rank)' if dist is None else f'torch.distributed ({dist.get_backend()})'
        # (force_comm: tests drive the agreement protocol of _make_comm over gloo with a stand-in context)
        if dist is not None and ctx is not None and (dist.get_backend() == 'nccl' or force_comm):
            self._make_comm(dist, ctx)
        if dist is not None:
            import torch
            self.torch = torch
            on_gpu = str(device).startswith('cuda')
            self.host = torch.zeros(3, dtype=torch.float64, pin_memory=on_gpu)
            self.send = torch.zeros(3, dtype=torch.float64, device=device)
            self.all = torch.zeros(dist.get_world_size(), 3, dtype=torch.float64, device=device)
            self.all_rows = list(self.all.unbind(0))      # views: all_gather fills the rows of self.all (gloo has no all_gather_into_tensor)
            self.all_host = torch.zeros(dist.get_world_size(), 3, dtype=torch.float64, pin_memory=on_gpu)
            self.run_host = torch.zeros(1, dtype=torch.float64, pin_memory=on_gpu)
            self.run_dev = torch.zeros(1, dtype=torch.float64, device=device)

    def _make_comm(self, dist, ctx):
        """Every step that can fail on ONE rank is preceded or followed by a collective agreement, so that all ranks end on the
        same collective: a local preflight (can RCCL be loaded here?) is agreed before anything collective on RCCL's side starts; rank 0 always takes part in the broadcast of the id (129 bytes: the id + an ok flag, zero when it could not
        be created), and after d2d_comm_create every rank all-reduces (MIN) its own ok flag over torch.distributed -- if any rank
        failed, every rank closes its d2d_comm and the exchange stays on torch.distributed."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        why = None
        # Preflight: d2d_comm_create is itself a collective (ncclCommInitRank returns when ALL ranks have joined), so a rank that
        # cannot get there -- librccl does not load on it -- would leave the others waiting inside it.  Every rank checks locally
        # (d2d_comm_available: dlopen + symbols, no communication) and the flags are agreed over torch.distributed FIRST.
        try:
            miss = ctx.comm_available()
        except Exception as e:           # noqa: BLE001
            miss = repr(e)
        ready = torch.tensor([0 if miss else 1], dtype=torch.int32, device=ctx.device)
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if int(ready.item()) != 1:
            self.comm = None
            self.collective = (f'torch.distributed ({dist.get_backend()}); d2d_comm not used: '
                               + (f'RCCL not available on this rank: {miss}' if miss else 'RCCL not available on another rank'))[:300]
            return
        msg = torch.zeros(129, dtype=torch.uint8, device=ctx.device)
        if rank == 0:
            try:
                buf = bytearray(ctx.comm_unique_id()) + bytearray([1])
                msg.copy_(torch.frombuffer(buf, dtype=torch.uint8))
            except Exception as e:       # noqa: BLE001  (msg stays zero: flag 0 tells the others)
                why = f'd2d_comm_unique_id: {e!r}'
        dist.broadcast(msg, src=0)
        raw = bytes(msg.cpu().numpy().tobytes())
        comm = None
        if raw[128] == 1:
            try:
                comm = ctx.comm_create(raw[:128], rank, world)
                info = comm.info()           # ncclCommUserRank / ncclCommCount, cross-checked against (rank, world) by the library
                if tuple(info) != (rank, world):
                    raise RuntimeError(f'd2d_comm reports rank {info[0]} of {info[1]}, expected {rank} of {world}')
            except Exception as e:       # noqa: BLE001
                why = f'd2d_comm_create: {e!r}'
                if comm is not None:
                    comm.close()
                comm = None
        elif why is None:
            why = 'rank 0 could not create the RCCL id'
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=ctx.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            self.comm = comm
            self.rccl_ranks = comm.info()[1]
            self.collective = 'd2d_allreduce_stats (RCCL through the C-ABI, d2d_comm of %d ranks by ncclCommCount)' % self.rccl_ranks
        else:
            if comm is not None:
                comm.close()
            self.comm = None
            self.collective = (f'torch.distributed ({dist.get_backend()}); d2d_comm not used: ' + (why or 'another rank could not create it'))[:300]

    def _abi(self, a, b, c):
        """one d2d_allreduce_stats: (sum a, max b, sum c) over the ranks"""
        self.host[0] = float(a); self.host[1] = float(b); self.host[2] = float(c)
        self.send.copy_(self.host, non_blocking=True)
        self.comm.allreduce_stats(self.send)
        self.host.copy_(self.send)                 # (synchronises: the host needs the numbers)
        return float(self.host[0]), float(self.host[1]), float(self.host[2])

    def __call__(self, cost_sum, gmax, running):
        """(sum of costs, max |J^T r|, trajectories not converged) over all ranks: ONE collective (d2d_allreduce_stats; over
        torch.distributed an all-gather of the three scalars, sum and max then taken locally)."""
        if self.dist is None:
            return float(cost_sum), float(gmax), int(running)
        if self.comm is not None:
            s0, s1, s2 = self._abi(cost_sum, gmax, running)
            return s0, s1, int(round(s2))
        self.host[0] = float(cost_sum); self.host[1] = float(gmax); self.host[2] = float(running)
        self.send.copy_(self.host, non_blocking=True)
        self.dist.all_gather(self.all_rows, self.send)
        self.all_host.copy_(self.all)              # (synchronises: the host needs the numbers)
        t = self.all_host
        return float(t[:, 0].sum()), float(t[:, 1].max()), int(round(float(t[:, 2].sum())))

    def running_only(self, running):
        """The per-check exchange inside the iteration loop: one all-reduce of one scalar."""
        if self.dist is None:
            return int(running)
        if self.comm is not None:
            return int(round(self._abi(0.0, 0.0, running)[2]))
        self.run_host[0] = float(running)
        self.run_dev.copy_(self.run_host, non_blocking=True)
        self.dist.all_reduce(self.run_dev, op=self.dist.ReduceOp.SUM)
        self.run_host.copy_(self.run_dev)          # (synchronises: the host decides whether another launch is needed)
        return int(round(float(self.run_host[0])))


def solve_sharded(plan, scen, q, reducer, check_every=8, max_iter=200, **tol):
    """LM solve of this rank's shard (scen, q on this rank's device) with a GLOBAL convergence
    decision: every rank keeps iterating (its converged trajectories are masked on the device)
    until no rank has a running trajectory left.  Returns (cost, iters, status, local_stats,
    global (cost_sum, gmax, not_converged), n_checks)."""
    B = scen.shape[0]
    plan.begin(B)
    checks = 0
    while True:
        running = plan.iterate(scen, q, check_every, max_iter=max_iter, **tol)
        checks += 1
        if reducer.running_only(running) == 0:
            break
    cost, iters, status, stats = plan.finish(scen, q)
    glob = reducer(stats[0], stats[1], stats[2])
    return cost, iters, status, stats, glob, checks
