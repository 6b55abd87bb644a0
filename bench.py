#!/usr/bin/env python3
"""Headline benchmark: trajectory-optimisations/sec (6-segment polynomial, 50 waypoints).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one complete Levenberg-Marquardt solve of the per-GPU batch (BASELINE.json
configs[1]: 4096 independent single-drone fits, S=6, K=50) with the scenarios and the
initial guesses already resident in HBM.  With N > 1 the batch shards by trajectory
(4096 per rank, weak scaling); the only collective is the all-reduce of the convergence
statistics [sum cost, max |J^T r|, trajectories still running] after every persistent launch
of the LM kernel (`check_every` iterations; default = max_iter, i.e. one launch and one
all-reduce per solve; RCCL, `nccl` backend).  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      -- the dominant kernel (fit_lm_kernel: the whole LM loop, J^T J on the fp32 MFMA), HIP events
                   around each of its launches inside the timed region: algorithmic flop = 200*48*49 per J^T J
                   evaluation x evaluations / summed HIP-event kernel time
  roofline_isolated -- the J^T J kernel of the split path (fit_eval_kernel) alone on the full resident
                   batch (every trajectory active), HIP events around the kernel only
  cpu_baseline  -- scipy.optimize.least_squares (method 'lm', analytic Jacobian) on the
                   oracle's residual function over a process pool, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np   # noqa: E402

K, S_ = 50, 6
OBJ_SCALE = 0.1
FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA peak
ROWS_CONTRACTED = 200             # v, phi and the two position rows of the 50 samples
NQ2 = 48
ALG_FLOP_PER_EVAL = ROWS_CONTRACTED * NQ2 * (NQ2 + 1)      # M*P*(P+1), SURVEY.md 8d


def _plan_consts():
    from d2dhip import synth
    return synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(OBJ_SCALE, K)


# ---------------------------------------------------------------------------------------
# CPU baseline (rank 0, N == 1): runs BEFORE anything touches the GPU so that the worker
# processes are plain forks of a HIP-free parent.
# ---------------------------------------------------------------------------------------
def _cpu_fit_one(args):
    from scipy.optimize import least_squares
    from oracle import fit as F
    basis, sc = args
    wp = F.waypoints(sc, basis.K, basis.duration)
    fun = lambda qq: F.residuals(basis, sc, qq, wp).reshape(-1)                        # noqa: E731
    jac = lambda qq: F.jacobian(basis, F.residuals(basis, sc, qq, wp, True)[1])        # noqa: E731
    res = least_squares(fun, F.initial_guess(basis, sc, wp), jac=jac, method='lm', xtol=1e-12, ftol=1e-12, gtol=1e-12)
    return 2 * res.cost


def _host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota and by
    the GPU box's per-GPU CPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('D2D_BENCH_CORES', 16))))


def cpu_baseline(n_sample=1024):
    import multiprocessing as mp
    from oracle import fit as F               # the oracle is the thing timed in this leg only
    from d2dhip import synth
    dur, wref = _plan_consts()
    basis = F.FitBasis(S_, K, dur, wref)
    sc = synth.synth_scenarios(n_sample, seed=20241008, obj_scale=OBJ_SCALE, K=K)
    cores = _host_cores()
    with mp.get_context('fork').Pool(cores) as pool:
        pool.map(_cpu_fit_one, [(basis, sc[i]) for i in range(min(cores, n_sample))])   # warm the workers
        t0 = time.perf_counter()
        costs = pool.map(_cpu_fit_one, [(basis, sc[i]) for i in range(n_sample)], chunksize=1)
        dt = time.perf_counter() - t0
    return {'value': n_sample / dt, 'unit': 'trajectory-optimisations/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_sample} of the {4096} bench trajectories, scipy.optimize.least_squares(method=lm, analytic '
                      f'Jacobian, tol 1e-12) on oracle/fit.py residuals, multiprocessing.Pool({cores}), {dt:.1f} s wall',
            'mean_cost': float(np.mean(costs))}


def cpu_baseline_sim_gvf(n_steps=300):
    """cpu_baseline leg of the simulation bench (tools/bench_sim.py, BASELINE configs[4]): the oracle's restatement of
    the reference's phase-1 loop body (DCF + GVF + scipy odeint) on one core."""
    from oracle import sim as S               # the oracle is the thing timed in this leg only
    c = np.array([[0, -20], [25, -20], [25, -100], [0, -100.0]])
    X0 = np.tile([20, 30, -np.pi / 2, 0, 10.0], (4, 1))
    t0 = time.perf_counter()
    S.formation_gvf_run(c, 60.0, 15.0, X0, n_steps, 0.05, integrator='odeint')
    dt = time.perf_counter() - t0
    return {'value': 4 * (n_steps - 1) / dt, 'unit': 'drone-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n_steps - 1} steps x 4 aircraft, oracle/sim.py loop with scipy.integrate.odeint as src/d2d/dynamic.py:26'}


def cpu_baseline_sim_track(n_steps=60):
    """cpu_baseline leg of the tracking bench: flatness + scipy CARE + odeint (oracle/sim.py track_run) on one core."""
    from oracle import sim as S
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'tracking_trace_carestandin.npz'))
    t0 = time.perf_counter()
    S.track_run(g['time'][:n_steps], g['x_ref'][:n_steps], g['y_ref'][:n_steps], g['X'][0], integrator='odeint')
    dt = time.perf_counter() - t0
    return {'value': 4 * (n_steps - 1) / dt, 'unit': 'drone-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n_steps - 1} steps x 4 aircraft, flatness + scipy CARE + odeint (oracle/sim.py track_run)'}


# ---------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=4096, help='trajectories per GPU')
    ap.add_argument('--check-every', type=int, default=200,
                    help='LM iterations per persistent launch = interval of the global convergence check (all-reduce)')
    ap.add_argument('--max-iter', type=int, default=150,
                    help='damped solves per trajectory before it is reported as not converged (99.9 %% of the fits need '
                         '<= 116; the library default is 200)')
    ap.add_argument('--so-lambda', type=float, default=None,
                    help='damping below which the evaluations carry the second-order term (default: library default; 0 = Gauss-Newton)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=1024)
    ap.add_argument('--iso-large', type=int, default=32768,
                    help='also time the isolated J^T J kernel on this many resident trajectories (roofline_isolated.large; 0 = skip)')
    ap.add_argument('--large-batch', type=int, default=0,
                    help='extra single solve at this batch size, reported as large_batch (off by default: its launches '
                         'would mix into the kernel statistics of the headline configuration)')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_sample)

    import torch
    import d2dhip
    from d2dhip import synth
    dist = None
    # one rank per GPU; D2D_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal of this path
    backend = os.environ.get('D2D_DIST_BACKEND', 'nccl')
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(dev_index)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ctx = d2dhip.Context(dev_index)
    dur, wref = _plan_consts()
    plan = d2dhip.FitPlan(ctx, S_, K, dur, wref)
    B = a.batch
    sc = synth.synth_scenarios(B, seed=20241008, rank=rank, obj_scale=OBJ_SCALE, K=K)
    dsc = ctx.dev(sc)
    q0 = plan.init(dsc)
    from d2dhip.dist import StatsReducer, solve_sharded
    reducer = StatsReducer(dist, ctx.device if backend == 'nccl' else 'cpu')

    tolkw = {} if a.so_lambda is None else {'so_lambda': a.so_lambda}

    def one_step():
        """Full LM solve of the resident shard with the global convergence check."""
        q = q0.clone()
        cost, iters, status, stats, glob, checks = solve_sharded(plan, dsc, q, reducer, a.check_every, a.max_iter, **tolkw)
        return (cost, iters, status, stats), q

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    # per-kernel HIP events over the timed region itself (d2d_fit_profile: two event records on the library's stream
    # around each hot-path launch, no synchronisation until they are read back after the closing barrier)
    plan.profile(True)
    n_evals = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        (cost, iters, status, stats), q = one_step()
        n_evals += stats[3]                # evaluation units of this solve (host copy already made by the solve)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctx.device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = status.cpu().numpy()
    conv = float(np.isin(st, (d2dhip.ST_CONVERGED, d2dhip.ST_STALLED)).mean())

    # ---- per-kernel roofline from the events of the timed region -------------
    roof = roof_iso = None
    ev_ms, ev_n, stp_ms, stp_n, lm_ms, lm_n = plan.profile_read()
    plan.profile(False)
    if rank == 0:
        if lm_n > 0:
            # the whole LM loop runs in one persistent kernel per convergence check: it IS the hot path
            ach = ALG_FLOP_PER_EVAL * n_evals / (lm_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'fit_lm_kernel<3,24> (fused LM loop: fp64 residual/J^T r, J^T J on v_mfma_f32_16x16x4_f32, fp32 Cholesky)',
                    'achieved': ach, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP32_PEAK_TFLOPS, 'traffic': None,
                    'alg_flop_per_unit': ALG_FLOP_PER_EVAL, 'units_per_launch_avg': n_evals / lm_n,
                    'avg_launch_us': 1e3 * lm_ms / lm_n, 'launches': int(lm_n), 'kernel_ms_total': lm_ms,
                    'note': 'achieved counts only the J^T J contraction (M*P*(P+1) per evaluation); the same kernel also does the '
                            'fp64 residual/gradient phases, the Cholesky solves and the trial costs'}
        else:
            ach = ALG_FLOP_PER_EVAL * n_evals / (ev_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'fit_eval_kernel (J^T J, v_mfma_f32_16x16x4_f32)', 'achieved': ach,
                    'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP32_PEAK_TFLOPS, 'traffic': None,
                    'alg_flop_per_unit': ALG_FLOP_PER_EVAL, 'units_per_launch_avg': n_evals / ev_n,
                    'avg_launch_us': 1e3 * ev_ms / ev_n, 'launches': int(ev_n),
                    'step_kernel_avg_launch_us': 1e3 * stp_ms / stp_n, 'eval_ms_total': ev_ms, 'step_ms_total': stp_ms}
        # isolated: the J^T J kernel alone (fit_eval_kernel), every trajectory active in one launch, bracketed by
        # HIP events on the library's stream (d2d_fit_profile); prep / untile launches of the public call excluded
        for _ in range(3):
            plan.eval(dsc, q0)
        torch.cuda.synchronize()
        plan.profile(True)
        nit = 20
        for _ in range(nit):
            plan.eval(dsc, q0)
        iso_ms, iso_n = plan.profile_read()[:2]
        plan.profile(False)
        iso_ms /= iso_n
        # the same kernel without its MFMA section (no J^T J requested): the difference is the time of the contraction
        for _ in range(3):
            plan.eval(dsc, q0, want_H=False)
        torch.cuda.synchronize()
        plan.profile(True)
        for _ in range(nit):
            plan.eval(dsc, q0, want_H=False)
        noh_ms, noh_n = plan.profile_read()[:2]
        plan.profile(False)
        noh_ms /= noh_n
        ach_i = ALG_FLOP_PER_EVAL * B / (iso_ms * 1e-3) / 1e12
        roof_iso = {'bound': 'mfma', 'kernel': 'fit_eval_kernel<3,24,true> (fp64 residual / J^T r phases + J^T J on v_mfma_f32_16x16x4_f32)',
                    'achieved': ach_i, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach_i / FP32_PEAK_TFLOPS, 'avg_launch_us': 1e3 * iso_ms, 'units_per_launch': B, 'launches': int(iso_n),
                    'avg_launch_us_without_jtj': 1e3 * noh_ms, 'jtj_section_us': 1e3 * (iso_ms - noh_ms),
                    'jtj_section_frac': ALG_FLOP_PER_EVAL * B / ((iso_ms - noh_ms) * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                    'note': 'the split-path J^T J kernel on the full resident batch (kernel only, HIP events; rocprofv3 summary in profiles/); '
                            'frac is over the whole kernel (fp64 row phases + contraction), jtj_section_* is the contraction alone, by difference '
                            'against the same kernel launched without it'}

        # the same two measurements with 32 768 trajectories resident (eight per wave slot of the J^T J kernel)
        if world == 1 and a.iso_large > B:
            Bi = a.iso_large
            dsci = ctx.dev(synth.synth_scenarios(Bi, seed=20241008, rank=0, obj_scale=OBJ_SCALE, K=K))
            q0i = plan.init(dsci)
            res = []
            for want_H in (True, False):
                for _ in range(2):
                    plan.eval(dsci, q0i, want_H=want_H)
                torch.cuda.synchronize()
                plan.profile(True)
                for _ in range(10):
                    plan.eval(dsci, q0i, want_H=want_H)
                ms, cnt = plan.profile_read()[:2]
                plan.profile(False)
                res.append(ms / cnt)
            roof_iso['large'] = {'units_per_launch': Bi, 'avg_launch_us': 1e3 * res[0], 'avg_launch_us_without_jtj': 1e3 * res[1],
                                 'frac': ALG_FLOP_PER_EVAL * Bi / (res[0] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                 'jtj_section_frac': ALG_FLOP_PER_EVAL * Bi / ((res[0] - res[1]) * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
            del dsci, q0i
            torch.cuda.empty_cache()

    # ---- the same solve on a larger resident batch (rank 0, N = 1 only): 4096 fits on 2048 wave slots are
    # bound by the last 1 % of the fits (110..200 iterations); this shows the throughput-bound regime
    large = None
    if rank == 0 and world == 1 and a.large_batch > B:
        Bl = a.large_batch
        dscl = ctx.dev(synth.synth_scenarios(Bl, seed=20241008, rank=0, obj_scale=OBJ_SCALE, K=K))
        q0l = plan.init(dscl)
        plan.solve(dscl, q0l.clone(), max_iter=a.max_iter, check_every=a.max_iter, **tolkw)
        torch.cuda.synchronize()
        tl = time.perf_counter()
        _c, _i, _s, stl = plan.solve(dscl, q0l.clone(), max_iter=a.max_iter, check_every=a.max_iter, **tolkw)
        torch.cuda.synchronize()
        tl = time.perf_counter() - tl
        large = {'batch': Bl, 'value': Bl / tl, 'unit': 'trajectory-optimisations/s', 'ms_per_step': 1e3 * tl,
                 'jtj_frac_of_fp32_mfma_peak': ALG_FLOP_PER_EVAL * float(stl[3]) / tl / 1e12 / FP32_PEAK_TFLOPS,
                 'note': 'one solve, wall clock around d2d_fit_solve; not the headline configuration'}
        del dscl, q0l

    if rank == 0:
        total = B * world * a.steps
        line = {
            'metric': 'trajectory-optimisations/sec (6-seg poly, 50 wpts)', 'value': total / dt,
            'unit': 'trajectory-optimisations/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64 residual/gradient + f32 MFMA J^T J', 'data': 'synthetic',
            'config': {'workload': f'batch={B} per GPU independent single-drone 6-seg poly fits, 50 waypoints (BASELINE configs[1])',
                       'segments': S_, 'samples': K, 'unknowns_reduced': NQ2, 'max_iter': a.max_iter,
                       'check_every': a.check_every, 'so_lambda': d2dhip.SO_LAMBDA if a.so_lambda is None else a.so_lambda,
                       'parallelism': f'trajectory-sharded x{world}'},
            'converged_frac': conv, 'mean_iters': float(iters.double().mean().item()),
            'evals_per_fit': float(stats[3] / B),          # in Gauss-Newton units (200 rows); second-order evaluations count 1.5
             'mean_cost': float(stats[0] / B),
            'roofline': roof, 'roofline_isolated': roof_iso, 'large_batch': large, 'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
