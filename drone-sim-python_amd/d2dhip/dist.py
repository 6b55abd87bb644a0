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
    tensor; a call is one host write, one collective, one read-back -- no tensors are built per call."""

    def __init__(self, dist=None, device='cpu'):
        self.dist = dist
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

    def __call__(self, cost_sum, gmax, running):
        """(sum of costs, max |J^T r|, trajectories not converged) over all ranks: ONE collective (an all-gather of the three
        scalars; sum and max are then taken locally)."""
        if self.dist is None:
            return float(cost_sum), float(gmax), int(running)
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
