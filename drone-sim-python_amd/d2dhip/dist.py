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
            self.mine = torch.zeros(3, dtype=torch.float64, device=device)
            self.all = [torch.zeros(3, dtype=torch.float64, device=device) for _ in range(dist.get_world_size())]

    def __call__(self, cost_sum, gmax, running):
        """(sum of costs, max |J^T r|, trajectories not converged) over all ranks: ONE collective (an
        all-gather of the three scalars; sum and max are then taken locally)."""
        if self.dist is None:
            return float(cost_sum), float(gmax), int(running)
        import torch
        self.mine.copy_(torch.tensor([float(cost_sum), float(gmax), float(running)], dtype=torch.float64))
        self.dist.all_gather(self.all, self.mine)
        t = torch.stack(self.all).cpu()
        return float(t[:, 0].sum()), float(t[:, 1].max()), int(round(float(t[:, 2].sum())))

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
