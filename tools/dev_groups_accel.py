#!/usr/bin/env python3
"""Development: what accelerates the slow scenarios of BASELINE configs[2] (8 aircraft x R replicas)?  The sweeps run on the GPU
(one block Gauss-Seidel sweep per call on the sub-batch of slow scenarios), the acceleration between sweeps is prototyped on the host:
  plain   nothing
  ls      after a sweep whose move is >= r0 x the one before (from sweep s0 on): line search on the JOINT cost F along the sweep's
          direction d = G(X) - X, first length rho / (1 - rho), doubled while F falls
Reports, per variant, the sweeps every scenario needs until a plain sweep moves <= tol, and F there.
  python tools/dev_groups_accel.py [R] [tol] [cap]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np, torch, d2dhip
from d2dhip import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6; CAP = int(sys.argv[3]) if len(sys.argv) > 3 else 200
K, n_ac = 50, 8
dur = synth.planner_timing(0, 4.9, 10)[2]
ctx = d2dhip.Context(0)
plan = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))
plan1 = d2dhip.FitPlan(ctx, 6, K, dur, synth.default_wref(1.0, K))           # uncoupled: own costs and sampled positions
sc = synth.circle_group_scenarios(n_ac, R, dur, K, seed=int(os.environ.get('SEED', '1')))
dsc = ctx.dev(sc.reshape(R * n_ac, -1))
q0 = plan.init(dsc)
q = q0.clone()
plan.solve_groups(dsc, q, n_ac, max_sweeps=CAP, inner_iters=8, tol=tol)
sw_, mv_ = plan.group_report(R)
slow = np.argsort(-sw_)[:48]
rng = np.random.default_rng(0)
ctrl = rng.choice(np.setdiff1d(np.arange(R), slow), 48, replace=False)
ids = np.concatenate([slow, ctrl]) if not os.environ.get('ALL') else np.concatenate([slow, np.setdiff1d(np.arange(R), slow)])
Rs = len(ids)
print('slow scenarios', slow[:20], 'their sweeps', sw_[slow[:20]], flush=True)
scs = sc[ids]
dss = ctx.dev(scs.reshape(Rs * n_ac, -1))
q0s = q0.view(R, n_ac, -1)[torch.as_tensor(ids, device=q0.device)].reshape(Rs * n_ac, -1).contiguous()
kc = torch.as_tensor(2.0 / scs[:, :, synth.SC_RCOL], device=q0.device)                      # [Rs][n_ac]
wcol = torch.as_tensor(scs[:, :, synth.SC_SCOL] * scs[:, :, synth.SC_KCOL], device=q0.device)


def joint_cost(qq):
    own = plan1.rows(dss, qq)[0].view(Rs, n_ac).sum(1)
    Y = plan1.sample(dss, qq)[0].view(Rs, n_ac, 6, K)
    x, y = Y[:, :, 0], Y[:, :, 1]
    tot = own.clone()
    for i in range(n_ac):
        for j in range(i + 1, n_ac):
            ex = (x[:, i] - x[:, j]) * kc[:, i, None]; ey = (y[:, i] - y[:, j]) * kc[:, i, None]
            tot += wcol[:, i] * torch.exp(-(ex * ex + ey * ey)).sum(1)
    return tot


def sweep(qq):
    return plan.solve_groups(dss, qq, n_ac, max_sweeps=1, inner_iters=8, tol=0.0)[0]


def relmove(a, b):
    d = (a - b).abs().amax(1) / (1.0 + b.abs().amax(1))
    return d.view(Rs, n_ac).amax(1)


def run(variant, s0=6, r0=0.6, amax=64.0, a0max=8.0):
    X = q0s.clone()
    first = np.full(Rs, -1); Ffirst = np.zeros(Rs)
    mprev = torch.full((Rs,), 1e300, device=X.device, dtype=torch.float64)
    n_ls = np.zeros(Rs, int); n_acc = np.zeros(Rs, int); n_F = np.zeros(Rs, int)
    for s in range(1, CAP + 1):
        Xp = X.clone()
        sweep(X)
        mv = relmove(X, Xp)
        newly = (mv <= tol).cpu().numpy() & (first < 0)
        if newly.any():
            Fn = joint_cost(X).cpu().numpy()
            first[newly] = s; Ffirst[newly] = Fn[newly]
        if (first >= 0).all(): break
        if variant == 'ls' and s >= s0:
            rho = mv / mprev
            want = (rho >= r0) & (mv > tol) & torch.as_tensor(first < 0, device=X.device)
            if want.any():
                d = (X - Xp).view(Rs, n_ac, -1)
                a = torch.where(rho < 1.0, rho / (1.0 - rho).clamp_min(1e-3), torch.full_like(rho, a0max)).clamp(1.0, a0max)
                F0 = joint_cost(X); n_F += want.cpu().numpy()
                best_F = F0.clone(); best_a = torch.zeros_like(a)
                act = want.clone()
                for t in range(5):
                    Xt = (X.view(Rs, n_ac, -1) + (a * act)[:, None, None] * d).reshape(Rs * n_ac, -1).contiguous()
                    Ft = joint_cost(Xt); n_F += act.cpu().numpy()
                    better = act & (Ft < best_F)
                    best_F = torch.where(better, Ft, best_F); best_a = torch.where(better, a, best_a)
                    if t == 0:
                        # a first trial that does not lower F: one shorter try, then give up
                        shrink = act & ~better
                        a = torch.where(shrink, a * 0.25, a * 2.0)
                        tried_short = shrink
                    else:
                        act = act & better & ~tried_short if t == 1 else act & better
                        a = a * 2.0
                    act = act & (a <= amax)
                    if not act.any(): break
                X = (X.view(Rs, n_ac, -1) + best_a[:, None, None] * d).reshape(Rs * n_ac, -1).contiguous()
                n_ls += want.cpu().numpy(); n_acc += (best_a > 0).cpu().numpy()
        mprev = mv
    first[first < 0] = CAP + 1
    return first, Ffirst, n_ls, n_acc, n_F


# consistency of the host's joint cost: the sub-problem costs of a sweep's end add up to own + 2 x pairs
_q = q0s.clone(); _c = sweep(_q).view(Rs, n_ac).sum(1)
_own = plan1.rows(dss, _q)[0].view(Rs, n_ac).sum(1); _F = joint_cost(_q)
print('check: sum of sub-problem costs - (2 F - own), max rel', float(((_c - (2 * _F - _own)).abs() / _c.abs()).max()), flush=True)
res = {}
VAR = [('plain', {}), ('ls', dict(s0=6, r0=0.6)), ('ls', dict(s0=8, r0=0.8)), ('ls', dict(s0=6, r0=0.8)), ('ls', dict(s0=6, r0=0.7)), ('ls', dict(s0=8, r0=0.7)),
       ('ls', dict(s0=10, r0=0.8)), ('ls', dict(s0=6, r0=0.7, a0max=16.0)), ('ls', dict(s0=8, r0=0.8, a0max=4.0))]
for variant, kw in VAR:
    t0 = time.time()
    first, Ff, n_ls, n_acc, n_F = run(variant, **kw)
    key = variant + str(kw)
    res[key] = (first, Ff)
    print(f'== {key}: all {Rs}: sweeps mean {first.mean():.2f} p99 {np.percentile(first, 99):.0f} max {first.max()} beyond40 {(first > 40).sum()} | slow 48: mean {first[:48].mean():.1f} max {first[:48].max()}'
          f' | per scenario: line searches {n_ls.mean():.2f} accepted {n_acc.mean():.2f} F evaluations {n_F.mean():.2f} (slow 48: {n_F[:48].mean():.1f}) | sweeps + 0.4 F-evals: mean {(first + 0.4 * n_F).mean():.2f} max {(first + 0.4 * n_F).max():.1f}  ({time.time() - t0:.0f} s)', flush=True)
    if variant != 'plain':
        p = res["plain{}"]
        ok = p[1] != 0
        dF = (Ff[ok] - p[1][ok]) / np.abs(p[1][ok])
        print('   F vs plain (rel): lower by > 1e-6:', int((dF < -1e-6).sum()), ' higher by > 1e-6:', int((dF > 1e-6).sum()), ' worst', dF.max(), ' best', dF.min())
