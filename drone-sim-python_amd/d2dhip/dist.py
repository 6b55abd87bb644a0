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
    """All-reduce of the three convergence scalars.  `device` is where the backend wants the
    buffer (cuda:<local_rank> for nccl, cpu for gloo)."""

    def __init__(self, dist=None, device='cpu'):
        self.dist = dist
        if dist is not None:
            import torch
            self.sum_buf = torch.zeros(2, dtype=torch.float64, device=device)
            self.max_buf = torch.zeros(1, dtype=torch.float64, device=device)

    def __call__(self, cost_sum, gmax, running):
        if self.dist is None:
            return float(cost_sum), float(gmax), int(running)
        self.sum_buf[0] = float(cost_sum); self.sum_buf[1] = float(running); self.max_buf[0] = float(gmax)
        self.dist.all_reduce(self.sum_buf, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(self.max_buf, op=self.dist.ReduceOp.MAX)
        return float(self.sum_buf[0].item()), float(self.max_buf[0].item()), int(round(self.sum_buf[1].item()))

    def running_only(self, running):
        """The per-check exchange inside the iteration loop (one all-reduce)."""
        if self.dist is None:
            return int(running)
        self.sum_buf[0] = 0.0; self.sum_buf[1] = float(running)
        self.dist.all_reduce(self.sum_buf, op=self.dist.ReduceOp.SUM)
        return int(round(self.sum_buf[1].item()))


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
