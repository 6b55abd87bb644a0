#!/usr/bin/env python3
"""Headline benchmark: trajectory-optimisations/sec (6-segment polynomial, 50 waypoints).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): bench.py starts its N ranks itself --
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a child
process, BEFORE anything in this process touches the GPU -- and exits with the child's code.  Launched by
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.  Rank 0 prints ONE JSON line.

One "step" = one complete Levenberg-Marquardt solve of the per-GPU batch (BASELINE.json configs[1]: 4096 independent
single-drone fits, S=6, K=50) with the scenarios and the initial guesses already resident in HBM.  With N > 1 the batch
shards by trajectory (4096 per rank, weak scaling); the only collective is the all-reduce of the convergence statistics
[sum cost, max |J^T r|, trajectories still running] after every persistent launch of the LM kernel (`check_every`
iterations; default = max_iter, i.e. one launch and one all-reduce per solve; RCCL, `nccl` backend).

Besides the contract fields the line carries
  roofline          the dominant kernel (fit_lm_kernel: the whole LM loop, J^T J on the fp32 MFMA), HIP events around each of
                    its launches inside the timed region: algorithmic flop = 200*48*49 per J^T J evaluation
                    x evaluations / summed HIP-event kernel time
  roofline_isolated the contraction-only kernel (fit_jtj_kernel: row records from HBM -> MFMA -> tiles to HBM; its whole
                    duration is the J^T J contraction) on 4096 and on 32 768 resident trajectories, HIP events
  config2           BASELINE configs[2]: 8-drone formation x 8192 replicas with collision rows (coupled groups), one persistent launch
  config3           BASELINE configs[3]: 32 768 fits per rank (256 k at N = 8), same solve, barrier + max over ranks
  parity            the SAME scenarios the cpu_baseline leg solved with scipy: fraction agreeing to 1e-6 (cost, coefficients)
  sim               BASELINE configs[4] (65 536 drones x 10 000 steps GVF loop) and the tracking loop, with roofline + cpu_baseline
  long_horizon      the same fit at the reference's own horizon (121 nodes, exp_14): 4096 fits on the chunked kernel fit_lm_long_kernel
  nlp               SURVEY 8 f-1: the collocation-NLP backend on 4096 perturbed copies of the reference's exp_14 (one wavefront per problem),
                    with the oracle's solver timed beside it and the cost agreement on the problems both solved
  cpu_baseline      scipy.optimize.least_squares (method 'lm', analytic Jacobian) on the oracle's residual function over a
                    process pool, bounded sample
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'drone-sim-python_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np   # noqa: E402

K, S_ = 50, 6
OBJ_SCALE = 0.1
SEED = 20241008
FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA peak
FP64_PEAK_TFLOPS = 78.6           # fp64 vector peak
HBM_PEAK_GBS = 8000.0
ROWS_CONTRACTED = 200             # v, phi and the two position rows of the 50 samples
NQ2 = 48
ALG_FLOP_PER_EVAL = ROWS_CONTRACTED * NQ2 * (NQ2 + 1)      # M*P*(P+1), SURVEY.md 8d
JTJ_BYTES_PER_UNIT = 4 * K * 16 + 6 * 1024                  # contraction-only kernel: records read + tiles written


def pmc_traffic(kernel):
    """HBM bytes per launch of a hot kernel from the committed rocprofv3 PMC passes of THIS command (tools/profile_bench.sh ->
    tools/condense_profile.py -> profiles/r02/06_bench_final_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes,
    corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot read hardware counters itself; None when the file is absent."""
    try:
        import glob
        files = next((fs for fs in (sorted(glob.glob(os.path.join(ROOT, 'profiles', r, '*_traffic.json'))) for r in ('r06', 'r05', 'r04', 'r03')) if fs),
                     [os.path.join(ROOT, 'profiles', 'r02', '06_bench_final_traffic.json')])
        t = json.load(open(files[-1]))[kernel][0]
        t['file'] = os.path.relpath(files[-1], ROOT)
    except (OSError, KeyError, IndexError, ValueError):
        return None, None
    return t['fetch_bytes_per_launch'] + t['write_bytes_per_launch'], t


def pmc_issue(kernel):
    """What the SIMDs did during a launch of the fused solver kernel, from the committed rocprofv3 PMC passes of THIS command
    (tools/profile_bench.sh -> tools/condense_profile.py -> profiles/rNN/*_issue.json): MFMA instructions per evaluation unit, the
    fraction of SIMD cycles the MFMA pipe was busy, the fraction of issue slots used, wave slots occupied (of 2).  None if absent."""
    import glob
    for rnd in ('r06', 'r05'):
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', rnd, '*_issue.json')), reverse=True):
            try:
                v = json.load(open(f))[kernel]
                v['file'] = os.path.relpath(f, ROOT)
                return v
            except (OSError, KeyError, ValueError):
                continue
    return None


def pmc_valu(kernel, launch_s):
    """roofline against the fp64 vector pipe for the kernels it bounds (the simulation loops): executed fp64 flop per launch from the
    committed rocprofv3 pass of THIS command (profiles/r04/*_valu.json, tools/condense_profile.py) over the launch time measured
    here.  None when the file is absent."""
    for rnd in ('r06', 'r05', 'r04', 'r03'):
        try:
            import glob
            f = sorted(glob.glob(os.path.join(ROOT, 'profiles', rnd, '*_valu.json')))[-1]
            v = json.load(open(f))[kernel]
        except (OSError, KeyError, IndexError, ValueError):
            continue
        tf = v['fp64_flop_per_launch'] / launch_s / 1e12
        return {'bound': 'valu_fp64', 'kernel': kernel, 'achieved': tf, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / FP64_PEAK_TFLOPS,
                'executed_fp64_flop_per_launch': v['fp64_flop_per_launch'], 'valu_wave_insts_per_launch': v['valu_insts'], 'source': os.path.basename(f),
                'note': 'executed flop = (2 FMA + MUL + ADD) x 64 from SQ_INSTS_VALU_*_F64 of the committed PMC pass; one wave per SIMD at 65 536 drones'}
    return None


def _plan_consts():
    from d2dhip import synth
    return synth.planner_timing(0, 4.9, 10)[2], synth.default_wref(OBJ_SCALE, K)


def bench_scenarios(B, rank=0):
    """The bench batch of one rank (SURVEY.md 8d synthetic inputs, seed 20241008 + rank)."""
    from d2dhip import synth
    return synth.synth_scenarios(B, seed=SEED, rank=rank, obj_scale=OBJ_SCALE, K=K)


# ---------------------------------------------------------------------------------------
# CPU baseline (rank 0, N == 1): runs BEFORE anything touches the GPU so that the worker
# processes are plain forks of a HIP-free parent.
# ---------------------------------------------------------------------------------------
def _cpu_fit_one(args):
    from scipy.optimize import least_squares
    from oracle import fit as F
    basis, sc, q0 = args
    wp = F.waypoints(sc, basis.K, basis.duration)
    fun = lambda qq: F.residuals(basis, sc, qq, wp).reshape(-1)                        # noqa: E731
    jac = lambda qq: F.jacobian(basis, F.residuals(basis, sc, qq, wp, True)[1])        # noqa: E731
    res = least_squares(fun, F.initial_guess(basis, sc, wp) if q0 is None else q0, jac=jac, method='lm',
                        xtol=1e-15, ftol=1e-15, gtol=1e-15)      # (1e-12 leaves |J^T r| ~ 2e-7: coefficients off by up to 6e-6)
    return 2 * res.cost, res.x


KNOT_DEFAULT = True      # the headline shape's default solver runs in knot coordinates (csrc/fit_knot.hip); --kernel fused: the q kernel
_KB = {}


def _cpu_oracle_lm_one(args):
    """The CPU statement of the kernel's default algorithm (lmder on the normal equations + second-order finish) with the kernel's
    precision split (fp32 Hessian and Cholesky, fp64 residuals / cost / J^T r): oracle/fit_knot.py solve_minpack_knot for the knot
    kernel (the same trial points as oracle/fit.py solve_minpack in exact arithmetic), solve_minpack with --kernel fused."""
    from oracle import fit as F
    basis, sc = args
    if KNOT_DEFAULT:
        from oracle import fit_knot as FK
        kb = _KB.get(id(basis))
        if kb is None:
            kb = _KB[id(basis)] = FK.KnotBasis(basis)
        q, c, it, st, info = FK.solve_minpack_knot(kb, sc, hess_dtype=np.float32, chol_dtype=np.float32)
    else:
        q, c, it, st, info = F.solve_minpack(basis, sc, hess_dtype=np.float32, chol_dtype=np.float32)
    return c, q, it


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


def cpu_baseline(batch, n_sample=1024, n_oracle=1024):
    """scipy least_squares('lm') on the FIRST n_sample scenarios of rank 0's bench batch (the same rows the GPU solves), and the
    oracle's own lm_solve (the line-by-line fp64 CPU statement of the algorithm the kernel runs) on the first n_oracle of them.
    Returns the cpu_baseline record and the arrays the parity record needs."""
    import multiprocessing as mp
    from oracle import fit as F               # the oracle is the thing timed in this leg only
    dur, wref = _plan_consts()
    basis = F.FitBasis(S_, K, dur, wref)
    sc = bench_scenarios(batch)[:n_sample]
    n_sample = len(sc)
    cores = _host_cores()
    with mp.get_context('fork').Pool(cores) as pool:
        pool.map(_cpu_fit_one, [(basis, sc[i], None) for i in range(min(cores, n_sample))])   # warm the workers
        t0 = time.perf_counter()
        res = pool.map(_cpu_fit_one, [(basis, sc[i], None) for i in range(n_sample)], chunksize=1)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        ores = pool.map(_cpu_oracle_lm_one, [(basis, sc[i]) for i in range(min(n_oracle, n_sample))], chunksize=1)
        dto = time.perf_counter() - t1
    costs = np.array([r[0] for r in res]); qs = np.array([r[1] for r in res])
    z = np.array([F.coefficients(basis, sc[i], qs[i]) for i in range(n_sample)])
    rec = {'value': n_sample / dt, 'unit': 'trajectory-optimisations/s', 'cores': cores, 'kind': 'port',
           'sample': f'the first {n_sample} of the {batch} bench trajectories of rank 0, scipy.optimize.least_squares(method=lm, analytic '
                     f'Jacobian, tol 1e-15) on oracle/fit.py residuals, multiprocessing.Pool({cores}), {dt:.1f} s wall',
           'mean_cost': float(np.mean(costs)),
           'host_cpu_count': os.cpu_count(), 'per_core': n_sample / dt / cores,
           # SURVEY 8d asks for os.cpu_count() cores; a job on the GPU box may run 16 worker processes per GPU (its CPU share), so
           # the pool is that size (D2D_BENCH_CORES overrides) and the whole-host figure is an EXTRAPOLATION, labelled as such
           'extrapolated_to_host_cpu_count': n_sample / dt / cores * (os.cpu_count() or cores),
           'oracle_lm': {'value': len(ores) / dto, 'unit': 'trajectory-optimisations/s',
                         'sample': f'{"oracle/fit_knot.py solve_minpack_knot" if KNOT_DEFAULT else "oracle/fit.py solve_minpack"} (the CPU statement of the kernel\'s default algorithm, fp32 Hessian / Cholesky like the kernel) '
                                   f'on the first {len(ores)} of them, same pool, {dto:.1f} s wall', 'mean_iters': float(np.mean([r[2] for r in ores]))}}
    keep = {'basis': basis, 'sc': sc, 'cost': costs, 'q': qs, 'z': z,
            'o_cost': np.array([r[0] for r in ores]), 'o_q': np.array([r[1] for r in ores])}
    return rec, keep


LONG_HORIZONS = ((121, 12.0), (301, 30.0))       # (nodes, seconds) of the long_horizon record


def _long_scenarios(B, K2, t2, rank=0):
    from d2dhip import synth
    return synth.synth_scenarios(B, seed=SEED, rank=rank, obj_scale=OBJ_SCALE, K=K2, dist_range=(100. * t2 / 12.0, 150. * t2 / 12.0))


def cpu_long_horizon(n=512, B=4096):
    """cpu leg of the long_horizon record: scipy least_squares('lm') on the first n scenarios of each horizon (oracle residuals),
    for the record's parity_vs_scipy.  Returns {nodes: (costs, q, fits per second)}."""
    import multiprocessing as mp
    from oracle import fit as F
    from d2dhip import synth
    out = {}
    cores = _host_cores()
    with mp.get_context('fork').Pool(cores) as pool:
        for (K2, t2) in LONG_HORIZONS:
            dur = synth.planner_timing(0, t2, 10)[2]
            basis = F.FitBasis(S_, K2, dur, synth.default_wref(OBJ_SCALE, K2))
            sc = _long_scenarios(B, K2, t2)[:n]             # (the first n rows of the batch the GPU solves)
            t0 = time.perf_counter()
            res = pool.map(_cpu_fit_one, [(basis, sc[i], None) for i in range(n)], chunksize=1)
            out[K2] = (np.array([r[0] for r in res]), np.array([r[1] for r in res]), n / (time.perf_counter() - t0), cores)
    return out


def cpu_polish(keep, q_gpu, n_polish=256):
    """scipy LM started FROM the GPU's solutions (CPU pool; the GPU context exists by now, so the pool is spawned, not forked):
    how far does the CPU arbiter move them?  Returns the largest relative moves of cost and unknowns."""
    import multiprocessing as mp
    n = min(n_polish, len(q_gpu))
    with mp.get_context('spawn').Pool(min(_host_cores(), 8)) as pool:
        res = pool.map(_cpu_fit_one, [(keep['basis'], keep['sc'][i], q_gpu[i]) for i in range(n)], chunksize=4)
    return np.array([r[0] for r in res]), np.array([r[1] for r in res])


def cpu_baseline_nlp(n=32):
    """cpu_baseline leg of the collocation record: the oracle's solver (numpy + banded LAPACK, the same algorithm as the kernel)
    on the first n problems of the batch, one problem per host core at a time (forked pool: runs before the GPU is touched)."""
    import multiprocessing as mp
    from d2dhip import synth
    rows, W0, h = synth.nlp_problems(n)
    cores = min(_host_cores(), n)
    with mp.get_context('fork').Pool(cores) as pool:
        pool.map(_nlp_oracle_one, [(rows[b], W0[b], h) for b in range(min(cores, n))], chunksize=1)      # warm the workers
        t0 = time.perf_counter()
        res = pool.map(_nlp_oracle_one, [(rows[b], W0[b], h) for b in range(n)], chunksize=1)
        dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'problems/s', 'cores': cores, 'kind': 'port', 'seconds': dt,
            'sample': f'oracle/nlp.py solve() on the first {n} problems of the batch, multiprocessing.Pool({cores})',
            'status': [int(r[0]) for r in res], 'cost': [float(r[1]) for r in res], 'newton_steps': [int(r[3]) for r in res]}


def _nlp_oracle_one(args):
    from oracle import nlp as ON
    row, W0, h = args
    pb = ON.problem_from_row(row, W0.shape[1], h)
    _, info = ON.solve(pb, W0.T.copy())
    return info['status'], info['cost'], info['feas'], info['inner']


def nlp_verify(rows, W0, h, st, cost, feas, n_each=32):
    """The GPU's verdicts against the oracle's solver (oracle/nlp.py, the CPU statement of the same algorithm) on a sample that
    holds n_each problems the GPU reported D2D_ST_STALLED (no feasible point: the perturbed end poses cannot be joined in 12 s at
    v <= 15) and n_each it reported converged: status, residual and cost of both.  (Spawned pool: the GPU context exists.)"""
    import multiprocessing as mp
    stalled = np.nonzero(st == 4)[0][:n_each]; conv = np.nonzero(st == 1)[0][:n_each]
    idx = np.concatenate([stalled, conv])
    with mp.get_context('spawn').Pool(min(_host_cores(), 16)) as pool:
        res = pool.map(_nlp_oracle_one, [(rows[i], W0[i], h) for i in idx], chunksize=1)
    o_st = np.array([r[0] for r in res]); o_cost = np.array([r[1] for r in res]); o_feas = np.array([r[2] for r in res])
    ns = len(stalled)
    rec = {'n': int(len(idx)), 'stalled_sampled': int(ns), 'converged_sampled': int(len(conv)),
           'what': 'oracle/nlp.py solve() on the sampled problems (spawned pool): same verdict, and for the stalled ones the same residual',
           'status_agree_frac': float(np.mean(o_st == st[idx])),
           'stalled_both': int(np.sum(o_st[:ns] == 4)),
           'stalled_residual_gpu_min_max': [float(feas[stalled].min()), float(feas[stalled].max())] if ns else None,
           'stalled_residual_oracle_min_max': [float(o_feas[:ns].min()), float(o_feas[:ns].max())] if ns else None,
           'stalled_max_rel_residual_diff': float(np.max(np.abs(o_feas[:ns] - feas[stalled]) / np.maximum(feas[stalled], 1e-300))) if ns else None,
           'converged_max_rel_cost_diff': float(np.max(np.abs(o_cost[ns:] - cost[conv]) / np.abs(cost[conv]))) if len(conv) else None}
    return rec


def nlp_record(ctx, torch, cpu, B=4096):
    """the collocation-NLP backend (d2d_nlp_solve, what opty.direct_collocation.Problem(...).solve runs) on B perturbed
    copies of the reference's exp_14, one launch, one wavefront per problem."""
    from d2dhip import synth
    rows, W0, h = synth.nlp_problems(B)
    dsc = ctx.dev(rows)
    best, out = 1e30, None
    for rep in range(2):
        W = ctx.dev(np.ascontiguousarray(W0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ctx.nlp_solve(dsc, W, h)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    st, it, cost, feas = out['status'].cpu().numpy(), out['iters'].cpu().numpy(), out['cost'].cpu().numpy(), out['feas'].cpu().numpy()
    rec = {'metric': 'collocation problems/sec (121 nodes x 5 node variables, hard bounds)', 'value': B / best, 'unit': 'problems/s',
           'workload': f'{B} perturbed copies of optyplan_scenarios.exp_14 (end poses moved by N(0, [3 m, 3 m, 0.1 rad])), tri initial guess',
           'seconds': best, 'dtype': 'f64', 'converged_frac': float((st == 1).mean()),
           'infeasible_frac': float((st == 4).mean()),     # D2D_ST_STALLED: perturbed end poses that no v <= 15 path joins in 12 s (the oracle agrees)
           'mean_newton_steps': float(it.mean()),
           'max_newton_steps': int(it.max()), 'hbm_traffic_per_launch': pmc_traffic('nlp_solve_kernel')[0], 'cpu_baseline': cpu}
    # d2d_nlp_opts.order: the same problems handed out longest first by the step counts of the solve above (a re-solve of a problem set
    # knows them; the launch holds two problems per resident wavefront, so its end is the worst pair).  Results must not move.
    order = torch.from_numpy(np.argsort(-it, kind='stable').astype(np.int32)).to(ctx.device)
    hint = 1e30
    for rep in range(2):
        W2 = ctx.dev(np.ascontiguousarray(W0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out2 = ctx.nlp_solve(dsc, W2, h, order=order)
        torch.cuda.synchronize(); hint = min(hint, time.perf_counter() - t0)
    assert torch.equal(out2['iters'], out['iters']) and torch.equal(out2['cost'], out['cost']) and torch.equal(W2, W), 'the hand-out order changed a result'
    rec['value_order_hint'] = B / hint
    rec['order_hint'] = "d2d_nlp_opts.order = argsort of the previous solve's Newton-step counts, descending; results bit-identical (checked)"
    # algorithmic HBM bytes per node and Newton step of the round-3 algorithm (every value a phase needs read once, every
    # value it produces written once, neighbours from the cache): merit 13 doubles read x ~1.3 calls, assembly 18 read + 40 written
    # (x ~1.1 with the retries), cyclic reduction 21 read + 6 written (the reduced records live in the LDS), recovery 40 read + 5
    # written, update 20 read + 15 written = 118 doubles read + 70 written = 114 kB + 68 kB per step at N = 121.  Round 5: the 18
    # doubles of a reduced record go from the assembly to the cyclic reduction through the LDS (N <= 121): 100 read + 52 written
    # = 97 kB + 50 kB per step
    alg = float(it.sum()) * (97e3 + 50e3) * (W0.shape[2] / 121.0)
    # what a launch MUST move: the scenario row in, the node values in and out (5 x N doubles each way) and the four result words per
    # problem -- everything else of `alg` is the algorithm's own workspace, streamed through HBM / L2 once per Newton step because it
    # does not fit beside two waves' registers.  `workspace_stream_frac` is therefore a utilisation of self-inflicted traffic, NOT a
    # roofline fraction; the roofline fraction of the irreducible bytes is `irreducible_frac` (latency-bound kernel: tiny by construction)
    irr = float(B) * (rows.shape[1] * 8 + 2 * W0.shape[1] * W0.shape[2] * 8 + 4 * 8)
    rec['roofline'] = {'bound': 'hbm', 'kernel': 'nlp_solve_kernel', 'achieved': alg / best / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                       'workspace_stream_frac': alg / best / 1e9 / HBM_PEAK_GBS, 'traffic': rec['hbm_traffic_per_launch'],
                       'workspace_bytes_per_launch': alg, 'irreducible_bytes_per_launch': irr,
                       'irreducible_frac': irr / best / 1e9 / HBM_PEAK_GBS,
                       'traffic_over_irreducible': (rec['hbm_traffic_per_launch'] / irr) if rec['hbm_traffic_per_launch'] else None,
                       'note': 'latency-bound (dependent fp64 chains of the assembly and of the seven cyclic-reduction levels, one wavefront per problem, two waves per SIMD; round 5: every per-node load of a phase requested up front -- loads inside branches were forty serial memory round trips per Newton step -- and the reduced records handed over through the LDS: 121 k -> 160 k problems/s); `achieved` prices the workspace stream of the algorithm (147 kB per Newton step), not a roofline: see irreducible_*; traffic = FETCH_SIZE + WRITE_SIZE of the committed PMC passes (FETCH_SIZE uncalibrated for 8-B-per-lane loads)'}
    if cpu is not None:
        rec['verdicts_vs_oracle'] = nlp_verify(rows, W0, h, st, cost, feas)
        n = len(cpu['cost'])
        both = (st[:n] == 1) & (np.array(cpu['status']) == 1)
        rec['parity'] = {'n': n, 'status_agree_frac': float(np.mean(st[:n] == np.array(cpu['status']))), 'both_converged': int(both.sum()),
                         'max_rel_cost_diff_vs_oracle': float(np.max(np.abs(cost[:n] - np.array(cpu['cost']))[both] / np.array(cpu['cost'])[both])) if both.any() else None,
                         'newton_steps_gpu': it[:n].tolist(), 'newton_steps_oracle': cpu['newton_steps']}
    return rec


def long_horizon_record(ctx, torch, d2dhip, B=4096, K=121, t1=12.0, more=LONG_HORIZONS[1:], cpu_long=None, large_B=32768, long_tables=-1):
    """The reference's own planner horizons (101 .. 151 nodes at 10 Hz: exp_14 = 121; its 50 Hz scenarios: 211 .. 601) on the chunked
    persistent kernel (fit_lm_long_kernel, K > 64; the segment formulation of csrc/fit_seg.h): B independent fits of K nodes, same
    solver as the headline.  `more`: further (nodes, seconds) horizons reported under 'horizons'."""
    from d2dhip import synth
    dur = synth.planner_timing(0, t1, 10)[2]
    plan = d2dhip.FitPlan(ctx, S_, K, dur, synth.default_wref(OBJ_SCALE, K), long_tables=long_tables)
    dsc = ctx.dev(_long_scenarios(B, K, t1))
    q0 = plan.init(dsc)
    q_first = q0.clone()
    cost, iters, status, stats = plan.solve(dsc, q_first, max_iter=300)

    def best_of(n):
        best = 1e30
        for _ in range(n):
            q = q0.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = plan.solve(dsc, q, max_iter=300)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best, r
    best_index, (cost, iters, status, stats) = best_of(3)               # index order (a long-horizon plan has no built-in hand-out prior)
    # the hand-out prior of this horizon, regressed on ONE solve of 8192 OTHER scenarios of the same family (seed rank 1000: none of
    # them is in the timed batch) -- FitPlan.learn_handout_prior; the timed solves then predict each fit's length from its row alone
    dcal = ctx.dev(_long_scenarios(8192, K, t1, rank=1000))
    qcal = plan.init(dcal)
    _, it_cal, _, _ = plan.solve(dcal, qcal, max_iter=300)
    plan.learn_handout_prior(dcal, it_cal)
    del dcal, qcal
    best, (cost_p, iters_p, status_p, _) = best_of(3)
    assert torch.equal(cost_p, cost) and torch.equal(iters_p, iters)    # (the hand-out only schedules)
    plan.order_from_iters(iters)
    hinted, _ = best_of(2)
    plan.clear_order()
    st = status.cpu().numpy()
    rec = {'metric': f'trajectory-optimisations/sec (6-seg poly, {K} nodes)', 'value': B / best, 'unit': 'trajectory-optimisations/s',
           'workload': f'{B} independent fits, {K} nodes over {t1:g} s, end poses {100. * t1 / 12.0:g}-{150. * t1 / 12.0:g} m apart' + (' (the horizon of optyplan_scenarios.exp_14)' if K == 121 else ''),
           'kernel': plan.kernel, 'solver': 'library default (MINPACK lmder path + second-order finish)',
           'handout': 'longest-first by the trial count predicted from each scenario row (d2d_fit_opts.handout = D2D_HANDOUT_PREDICTED) with a prior regressed on one '
                      'solve of 8192 OTHER scenarios of this horizon (d2d_fit_plan_set_handout_prior / FitPlan.learn_handout_prior; untimed calibration)',
           'ms_per_step': 1e3 * best, 'value_index_order': B / best_index, 'value_with_order_hint': B / hinted, 'converged_frac': float((st == 1).mean()),
           'mean_iters': float(iters.float().mean().item()), 'max_iters': int(iters.max().item()), 'evals_per_fit': float(stats[3]) / B}
    plan.close()
    if cpu_long and K in cpu_long:                  # the same scenarios through the CPU arbiter (scipy on the oracle's residuals), same start
        cs, qs, rate, cores = cpu_long[K]
        n = len(cs)
        cg, qg = cost.cpu().numpy()[:n], q_first.cpu().numpy()[:n]
        rel_c = np.abs(cg - cs) / np.abs(cs); rel_q = np.abs(qg - qs).max(1) / np.abs(qs).max(1)
        rec['parity_vs_scipy'] = {'n': n, 'same_minimum_frac': float(((rel_c <= 1e-6) & (rel_q <= 1e-6)).mean()),
                                  'cost_rel_le_1e-6_frac': float((rel_c <= 1e-6).mean()), 'gpu_cost_lower_frac': float((cg < cs * (1 - 1e-6)).mean()),
                                  'cpu_fits_per_s': rate, 'cores': cores,
                                  'what': 'd2d_fit_solve (default solver, long-horizon kernel) vs scipy.optimize.least_squares(lm, tol 1e-15) on oracle/fit.py '
                                          'residuals from the same start; same minimum = cost and unknowns within 1e-6 relative'}
    if more:
        rec['horizons'] = []
        for (K2, t2) in more:
            r2 = long_horizon_record(ctx, torch, d2dhip, B=B, K=K2, t1=t2, more=(), cpu_long=cpu_long, large_B=0)
            h = {k: r2[k] for k in ('metric', 'value', 'workload', 'ms_per_step', 'value_index_order', 'value_with_order_hint', 'converged_frac',
                                    'mean_iters', 'max_iters', 'parity_vs_scipy') if k in r2}
            if large_B:
                # the throughput regime: 4096 fits of 23 trials on average and 84 at most keep the machine a quarter full
                # (profiles/r03/06_*); 32 768 fill it.  dense_count: J^T J flops by the SURVEY's count for the rows contracted
                # (rows x P x (P + 1); the segment formulation EXECUTES one 16x16x4 MFMA per sample + 216 per evaluation instead)
                r3 = long_horizon_record(ctx, torch, d2dhip, B=large_B, K=K2, t1=t2, more=(), cpu_long=None, large_B=0)
                tf = r3['evals_per_fit'] * 470400.0 * large_B / (r3['ms_per_step'] * 1e-3) / 1e12
                h['large_batch'] = {'batch': large_B, 'value': r3['value'], 'ms_per_step': r3['ms_per_step'], 'value_with_order_hint': r3['value_with_order_hint'],
                                    'converged_frac': r3['converged_frac'], 'mean_iters': r3['mean_iters'], 'evals_per_fit': r3['evals_per_fit'],
                                    'jtj_dense_count_tflops': tf, 'jtj_dense_count_frac': tf / FP32_PEAK_TFLOPS}
            rec['horizons'].append(h)
    return rec


def cpu_baseline_sim_gvf(n_steps=300):
    """cpu_baseline leg of the simulation records (BASELINE configs[4]): the oracle's restatement of
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
    """cpu_baseline leg of the tracking record: flatness + scipy CARE + odeint (oracle/sim.py track_run) on one core."""
    from oracle import sim as S
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'tracking_trace_carestandin.npz'))
    t0 = time.perf_counter()
    S.track_run(g['time'][:n_steps], g['x_ref'][:n_steps], g['y_ref'][:n_steps], g['X'][0], integrator='odeint')
    dt = time.perf_counter() - t0
    return {'value': 4 * (n_steps - 1) / dt, 'unit': 'drone-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n_steps - 1} steps x 4 aircraft, flatness + scipy CARE + odeint (oracle/sim.py track_run)'}


# ---------------------------------------------------------------------------------------
def self_spawn(a, argv):
    """bench.py --gpus N without a launcher: start the N ranks as a child torch.distributed.run BEFORE any GPU call here."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def sim_records(ctx, torch, cpu_g, cpu_t, drones=65536, steps=10000, track_steps=2000):
    """BASELINE configs[4]: the full_sim_case1 guidance loops at scale, one launch each, histories resident (allocated before the
    timed launch).  HIP events on the library's stream."""
    import d2dhip   # noqa: F401
    n_ac, N = 4, drones
    n_form = N // n_ac
    rng = np.random.default_rng(0)
    centres = np.tile(np.array([[0, -20], [25, -20], [25, -100], [0, -100.0]]), (n_form, 1)) + np.repeat(rng.uniform(-5, 5, (n_form, 2)), n_ac, 0)
    X0 = np.tile([20, 30, -np.pi / 2, 0, 10.0], (N, 1)) + np.concatenate([rng.uniform(-3, 3, (N, 2)), np.zeros((N, 3))], 1)
    dX0, dC, dR = ctx.dev(np.ascontiguousarray(X0.T)), ctx.dev(np.ascontiguousarray(centres.T)), ctx.dev(np.full(N, 60.0))
    rows = steps + 1

    def timed(fn, reps=2):
        out = fn(None); ctx.sync()
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.sync()
            e0.record(ctx.stream); fn(out); e1.record(ctx.stream)
            ctx.sync()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        return out, best

    recs = {}
    out, dte = timed(lambda o: ctx.gvf_run(dX0, dC, dR, n_ac, rows, 0.05, 15.0, record=('X', 'U'), out=o))
    n_steps = N * steps
    recs['gvf'] = {'metric': 'drone-steps/sec (GVF+DCF guidance + plant step, full X,U history)', 'value': n_steps / dte,
                   'unit': 'drone-steps/s', 'drones': N, 'steps': steps, 'launch_s': dte, 'dtype': 'f64',
                   'workload': 'BASELINE configs[4]: full_sim_case1 guidance loop, 65k drones x 10k plant steps, per-step controller evaluation',
                   'roofline': {'bound': 'hbm', 'kernel': 'gvf_run_kernel', 'achieved': n_steps * 56 / dte / 1e9, 'peak': HBM_PEAK_GBS,
                                'unit': 'GB/s', 'frac': n_steps * 56 / dte / 1e9 / HBM_PEAK_GBS, 'traffic': pmc_traffic('gvf_run_kernel')[0], 'alg_bytes_per_unit': 56,
                                'note': 'integration-bound (fp64 VALU); HBM GB/s reported as BASELINE configs[4] asks; PMC WRITE_SIZE = '
                                        'algorithmic to 0.03 % (profiles/)'},
                   'roofline_valu': pmc_valu('gvf_run_kernel', dte), 'cpu_baseline': cpu_g}
    del out
    torch.cuda.empty_cache()
    # occupancy scaling of the GVF loop: the same history volume (drones x steps) with 2 x and 4 x the drones, i.e. two and four waves
    # per SIMD instead of one -- is configs[4] (65 536 drones = 1024 waves = ONE wave per SIMD) latency-bound by construction?
    scaling = [{'drones': N, 'steps': steps, 'waves_per_simd': N / 64 / 1024, 'value': recs['gvf']['value'], 'launch_s': dte}]
    for mult in (2, 4):
        N2, st2 = N * mult, steps // mult
        nf2 = N2 // n_ac
        c2 = np.tile(np.array([[0, -20], [25, -20], [25, -100], [0, -100.0]]), (nf2, 1)) + np.repeat(rng.uniform(-5, 5, (nf2, 2)), n_ac, 0)
        X2 = np.tile([20, 30, -np.pi / 2, 0, 10.0], (N2, 1)) + np.concatenate([rng.uniform(-3, 3, (N2, 2)), np.zeros((N2, 3))], 1)
        dX2, dC2, dR2 = ctx.dev(np.ascontiguousarray(X2.T)), ctx.dev(np.ascontiguousarray(c2.T)), ctx.dev(np.full(N2, 60.0))
        o2, dt2 = timed(lambda o: ctx.gvf_run(dX2, dC2, dR2, n_ac, st2 + 1, 0.05, 15.0, record=('X', 'U'), out=o))
        scaling.append({'drones': N2, 'steps': st2, 'waves_per_simd': N2 / 64 / 1024, 'value': N2 * st2 / dt2, 'launch_s': dt2})
        del o2, dX2, dC2, dR2
        torch.cuda.empty_cache()
    recs['gvf']['occupancy_scaling'] = scaling
    recs['gvf']['speedup_2_waves_per_simd'] = scaling[1]['value'] / scaling[0]['value']
    recs['gvf']['speedup_4_waves_per_simd'] = scaling[2]['value'] / scaling[0]['value']
    T = track_steps + 1
    t = np.arange(T) * 0.1
    ph = rng.uniform(0, 2 * np.pi, N)
    x_ref = 60 * np.sin(0.15 * t[:, None] + ph[None, :]); y_ref = 40 * np.sin(0.3 * t[:, None] + 2 * ph[None, :])
    X0t = np.stack([x_ref[0], y_ref[0], np.arctan2(y_ref[1] - y_ref[0], x_ref[1] - x_ref[0]), np.zeros(N), 12 * np.ones(N)])
    dxr, dyr, dX0t = ctx.dev(x_ref), ctx.dev(y_ref), ctx.dev(X0t)
    o, dte = timed(lambda oo: ctx.track_run(dxr, dyr, dX0t, 0.1, record=('X', 'U'), out=oo))
    n_steps = N * track_steps
    recs['track'] = {'metric': 'drone-steps/sec (flatness + 5x5 LQR/CARE + plant step, full X,U history)', 'value': n_steps / dte,
                     'unit': 'drone-steps/s', 'drones': N, 'steps': track_steps, 'launch_s': dte, 'dtype': 'f64',
                     'workload': 'implement_controller loop of 11_full_sim_case1.py (:272-290) for 65k independent drones',
                     'roofline': {'bound': 'hbm', 'kernel': 'gradient_kernel x4 + track_run_kernel', 'achieved': n_steps * 104 / dte / 1e9,
                                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': n_steps * 104 / dte / 1e9 / HBM_PEAK_GBS, 'traffic': pmc_traffic('track_run_kernel')[0],
                                  'alg_bytes_per_unit': 104, 'note': 'CARE-bound (fp64 VALU); HBM GB/s as configs[4] asks'},
                     'roofline_valu': pmc_valu('track_run_kernel', dte), 'cpu_baseline': cpu_t}
    del o
    torch.cuda.empty_cache()
    return recs


# ---------------------------------------------------------------------------------------
# The line the driver parses is the LAST stdout line and stays small (round 4's 20.9 kB line was not parsed): contract keys +
# `roofline` + `cpu_baseline` + the hoisted scalars.  Everything else goes to the side file and to an EARLIER, prefixed stdout line.
LINE_LIMIT = 4096
ROOF_KEYS = ('bound', 'priced_against', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'alg_flop_per_unit', 'units_per_launch_avg', 'avg_launch_us', 'launches',
             'executed_mfma_flop_per_unit', 'frac_of_executed_flop', 'mfma_busy_frac', 'issue_slot_frac', 'wave_slots_occupied', 'pmc_source')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample', 'host_cpu_count')
DETAIL_PREFIX = 'BENCH_DETAIL '


def _clip(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + '...'


def compact_line(line, roof, cpu, detail_file=None):
    """The final JSON line: `line` (scalars and the small `config` object only) + the contract's `roofline` and `cpu_baseline`
    objects reduced to the keys the contract names.  Strings are clipped (harder if needed) so that the line never exceeds LINE_LIMIT."""
    for n in (400, 160, 80, 40):
        out = {k: _clip(v, n) for k, v in line.items() if not isinstance(v, (dict, list))}
        out['config'] = {k: _clip(v, n) for k, v in line.get('config', {}).items()}
        for k, v in line.items():
            if isinstance(v, list):
                out[k] = v
        out['roofline'] = {k: _clip(roof[k], n) for k in ROOF_KEYS if k in roof} if roof else None
        out['cpu_baseline'] = {k: _clip(cpu[k], n) for k in CPU_KEYS if k in cpu} if cpu else None
        out['detail'] = detail_file
        s = json.dumps(out)
        if len(s) <= LINE_LIMIT:
            return s
    raise ValueError(f'bench line is {len(s)} bytes, limit {LINE_LIMIT}: too many hoisted keys')


def emit(line, detail, roof, cpu):
    """Side file (gpurun_out/ when it exists, so that it travels back from a GPU box) + prefixed detail line, THEN the small line."""
    d = os.path.join(ROOT, 'gpurun_out')
    path = os.path.join(d if os.path.isdir(d) and os.access(d, os.W_OK) else ROOT, 'bench_detail.json')
    try:
        with open(path, 'w') as f:
            json.dump(detail, f, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError:
        rel = None
    print(DETAIL_PREFIX + json.dumps(detail), flush=True)
    print(compact_line(line, roof, cpu, rel), flush=True)


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
    ap.add_argument('--config3-batch', type=int, default=32768, help='fits per rank of the config3 record (0 = skip)')
    ap.add_argument('--config3-steps', type=int, default=3)
    ap.add_argument('--no-sim', action='store_true', help='skip the simulation records (BASELINE configs[4])')
    ap.add_argument('--no-nlp', action='store_true', help='skip the collocation-NLP record (SURVEY 8 f-1)')
    ap.add_argument('--no-groups', action='store_true', help='skip the coupled-groups record (BASELINE configs[2])')
    ap.add_argument('--kernel', choices=['auto', 'fused'], default='auto',
                    help='d2d_fit_plan_opts.kernel of the headline plan: auto = the knot kernel for the default solver, fused = the q-coordinate kernel of rounds 1-4')
    ap.add_argument('--handout', choices=['predicted', 'index'], default='predicted',
                    help='d2d_fit_opts.handout of the headline: predicted = longest-first by the trial count the device predicts from each scenario row inside the '
                         'solve (the library default; nothing is known from earlier solves), index = index order (always reported beside it)')
    ap.add_argument('--order-hint', action='store_true',
                    help='headline with the fits handed out longest-first by the iteration counts of the previous (warm-up) solve of the same '
                         'batch -- the replanning pattern; default: index order, no foreknowledge (the hinted figure is always reported beside it)')
    ap.add_argument('--mode', choices=['minpack', 'fast'], default='minpack',
                    help='solver of the headline: minpack = the library default (MINPACK lmder path + second-order finish: the path scipy follows), '
                         'fast = rounds 1-2\'s loop (reported beside it in any case)')
    ap.add_argument('--no-extra-modes', action='store_true', help='skip the fast_mode / minpack_pure / order-hint records')
    ap.add_argument('--dump-costs', default=None, help='(tests) every rank writes the costs / iteration counts of its last headline solve to <prefix>_rank<r>.npz')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_spawn(a, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('D2D_BENCH_SPAWN_TEST'):       # tests/test_dist_cpu.py: the launcher path without a GPU
        print(json.dumps({'rank': rank, 'world': world, 'local_rank': local_rank, 'gpus': a.gpus}), flush=True)
        return
    B = a.batch
    cpu = keep = cpu_g = cpu_t = cpu_n = cpu_l = None
    if rank == 0 and not a.no_cpu_baseline:
        # (rank 0 of ANY world size: a --gpus N line carries the same cpu_baseline / parity records as the N = 1 line -- the other
        # ranks wait in init_process_group meanwhile; the simulation / collocation / long-horizon legs stay single-rank records)
        cpu, keep = cpu_baseline(B, a.cpu_sample)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if not a.no_sim:
            cpu_g, cpu_t = cpu_baseline_sim_gvf(), cpu_baseline_sim_track()
        if not a.no_nlp:
            cpu_n = cpu_baseline_nlp()
            cpu_l = cpu_long_horizon()

    import torch
    import d2dhip
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
    plan = d2dhip.FitPlan(ctx, S_, K, dur, wref, kernel=a.kernel)
    global KNOT_DEFAULT
    KNOT_DEFAULT = plan.kernel == 'knot'
    from d2dhip.dist import StatsReducer, solve_sharded
    red_dev = ctx.device if backend == 'nccl' else 'cpu'
    reducer = StatsReducer(dist, red_dev, ctx if backend == 'nccl' else None)
    tolkw = {} if a.so_lambda is None else {'so_lambda': a.so_lambda}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce(vals, op):
        if dist is None:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return [float(v) for v in t.cpu()]

    HO = {'predicted': d2dhip.HANDOUT_PREDICTED, 'index': d2dhip.HANDOUT_INDEX}
    MODES = {'minpack': {'mode': d2dhip.MODE_MINPACK, 'handout': HO[a.handout]}, 'fast': {'mode': d2dhip.MODE_FAST},
             'minpack_pure': {'mode': d2dhip.MODE_MINPACK, 'mp_finish': 0},
             'minpack_index': {'mode': d2dhip.MODE_MINPACK, 'handout': d2dhip.HANDOUT_INDEX},
             'minpack_predicted': {'mode': d2dhip.MODE_MINPACK, 'handout': d2dhip.HANDOUT_PREDICTED}}
    HANDOUT_TXT = {'predicted': 'longest-first by the trial count predicted on the device from each scenario row, inside the timed solve (d2d_fit_opts.handout = '
                                'D2D_HANDOUT_PREDICTED, the default: no foreknowledge of this batch)', 'index': 'index order (no foreknowledge)'}

    def timed_solves(Bn, steps, warmup, order, mode='minpack', max_iter=None, keep=True):
        """warmup + `steps` timed full solves of this rank's Bn resident scenarios with the global convergence check; barrier
        + synchronize on both sides, MAX over ranks.  order: hand the fits out longest-first by the previous solve's iteration counts.
        Returns (seconds, last solve's results, evaluation units of the timed solves, HIP-event profile of the timed region,
        global [sum cost, not converged, sum iters, evals of the last solve, stalled])."""
        dsc = ctx.dev(bench_scenarios(Bn, rank))
        q0 = plan.init(dsc)
        kw = dict(tolkw, **MODES[mode])
        mi = a.max_iter if max_iter is None else max_iter
        res = None
        for _ in range(max(warmup, 1 if order else 0)):
            q = q0.clone()
            res = solve_sharded(plan, dsc, q, reducer, a.check_every, mi, **kw)
            if order:
                plan.order_from_iters(res[1])     # longest fits of the previous solve are handed out first
        plan.profile(True)
        n_evals = 0.0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            q = q0.clone()
            res = solve_sharded(plan, dsc, q, reducer, a.check_every, mi, **kw)
            n_evals += res[3][3]
        barrier()
        dt = time.perf_counter() - t0
        dt = allreduce([dt], 'MAX')[0]
        prof = plan.profile_read()
        # per-rank spread: the solver kernel's own time per step (HIP events) on the fastest and on the slowest rank -- the wall
        # time of a step is equalised by the convergence exchange itself, the kernel time is not
        kms = float(prof[4]) / max(steps, 1)
        timed_solves.rank_kernel_ms = (allreduce([kms], 'MIN')[0], allreduce([kms], 'MAX')[0])
        plan.profile(False)
        cost, iters, status, stats = res[:4]
        conv = float((status == d2dhip.ST_CONVERGED).sum().item()); stalled = float((status == d2dhip.ST_STALLED).sum().item())
        glob = allreduce([stats[0], Bn - conv, float(iters.double().sum().item()), stats[3], stalled], 'SUM')
        plan.clear_order()
        return dt, ((cost, iters, status, stats, q, dsc, q0) if keep else None), n_evals, prof, glob

    def summary(Bn, steps, dt, glob, prof, n_evals):
        tot = Bn * world
        r = {'value': tot * steps / dt, 'ms_per_step': 1e3 * dt / steps, 'converged_frac': 1.0 - glob[1] / tot, 'stalled_frac': glob[4] / tot,
             'mean_iters': glob[2] / tot, 'evals_per_fit': glob[3] / tot, 'mean_cost': glob[0] / tot}
        if prof[5] > 0:
            r['jtj_frac_of_fp32_mfma_peak_rank0'] = ALG_FLOP_PER_EVAL * n_evals / (prof[4] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS
        return r

    # ---- headline: BASELINE configs[1], the library's default solver, index order (nothing known about the fits in advance) ----
    dt, (cost, iters, status, stats, q, dsc, q0), n_evals, prof, glob = timed_solves(B, a.steps, a.warmup, a.order_hint, a.mode)
    ev_ms, ev_n, stp_ms, stp_n, lm_ms, lm_n = prof[:6]
    head_rank_ms = timed_solves.rank_kernel_ms
    total = B * world
    headline = summary(B, a.steps, dt, glob, prof, n_evals)
    if a.dump_costs:
        np.savez(f'{a.dump_costs}_rank{rank}.npz', cost=cost.cpu().numpy(), iters=iters.cpu().numpy(), status=status.cpu().numpy(), q=q.cpu().numpy())
    q_head = q[:a.cpu_sample].clone() if keep is not None else None
    z_head = plan.coeffs(dsc[:a.cpu_sample], q[:a.cpu_sample]).cpu().numpy() if keep is not None else None
    c_head = cost[:a.cpu_sample].cpu().numpy() if keep is not None else None
    # the other solver / hand-out combinations on the same batch, same timing discipline (every rank takes part: barriers inside)
    variants = {}
    gpu_sol = {}
    if not a.no_extra_modes:
        other = 'index' if a.handout == 'predicted' else 'predicted'
        for name, (mode_v, order_v, mi) in {'default_with_order_hint': (a.mode, not a.order_hint, None),
                                            'default_' + other + '_handout': ('minpack_' + other, False, None),
                                            'fast_mode': ('fast', False, None), 'fast_mode_with_order_hint': ('fast', True, None),
                                            'minpack_pure': ('minpack_pure', False, 600)}.items():
            if mode_v == a.mode and name == 'fast_mode':
                continue
            if a.mode != 'minpack' and name.endswith('_handout'):
                continue
            dtv, rv, nev, prv, gv = timed_solves(B, max(3, a.steps // 2), 1, order_v, mode_v, mi)
            variants[name] = summary(B, max(3, a.steps // 2), dtv, gv, prv, nev)
            variants[name]['handout'] = ('longest-first by the previous solve\'s iteration counts' if order_v else
                                         (HANDOUT_TXT[other] if name.endswith('_handout') else (HANDOUT_TXT[a.handout] if mode_v == 'minpack' else 'index order')))
            if keep is not None:
                gpu_sol[name] = (rv[0][:a.cpu_sample].cpu().numpy(), rv[4][:a.cpu_sample].cpu().numpy(),
                                 plan.coeffs(rv[5][:a.cpu_sample], rv[4][:a.cpu_sample]).cpu().numpy())
            del rv

    roof = roof_iso = None
    if rank == 0:
        if lm_n > 0:
            # the whole LM loop runs in one persistent kernel per convergence check: it IS the hot path
            ach = ALG_FLOP_PER_EVAL * n_evals / (lm_ms * 1e-3) / 1e12
            knot = plan.kernel == 'knot' and a.mode == 'minpack'
            kname = 'fit_lm_knot_kernel' if knot else 'fit_lm_kernel'
            roof = {'bound': 'issue' if knot else 'mfma', 'priced_against': 'mfma',
                    'kernel': (kname + ' (fused solver loop in knot coordinates: J^T J block tridiagonal, ONE v_mfma_f32_16x16x4_f32 per sample; fp64 residual / J^T r, fp32 Cholesky), solver = ' + a.mode) if knot
                    else (kname + '<3,24> (fused solver loop: fp64 residual/J^T r, J^T J on v_mfma_f32_16x16x4_f32, fp32 Cholesky), solver = ' + a.mode),
                    'achieved': ach, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP32_PEAK_TFLOPS,
                    'traffic': pmc_traffic(kname)[0], 'traffic_source': pmc_traffic(kname)[1],
                    'alg_flop_per_unit': ALG_FLOP_PER_EVAL, 'units_per_launch_avg': n_evals / lm_n,
                    'avg_launch_us': 1e3 * lm_ms / lm_n, 'launches': int(lm_n), 'kernel_ms_total': lm_ms,
                    'note': 'achieved = the ALGORITHMIC count of SURVEY 8d, M*P*(P+1) = 200*48*49 flop per J^T J evaluation (variant R, symmetric, dense), per '
                            'kernel time; the same kernel also does the fp64 residual / gradient phases, the Cholesky solves and the trial costs'
                            + ('.  The knot kernel EXECUTES the block-sparse form SURVEY 8d allows (each sample touches the 16 columns of its segment): '
                               '50-66 MFMAs of 2048 flop per evaluation instead of 300 -- executed_mfma_flop_per_unit says what the matrix cores really did' if knot else '')}
            if knot:
                # what the matrix cores really executed: the MFMA instructions of the committed PMC pass of this command (J^T J blocks: one per
                # sample and evaluation, two in second-order mode; + the 16 tile updates of every Cholesky factorisation), 2048 flop each
                iss = pmc_issue(kname)
                if iss:
                    ex = iss['mfma_insts_per_unit'] * 2048.0
                    roof.update({'executed_mfma_flop_per_unit': ex, 'frac_of_executed_flop': ex * n_evals / (lm_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                 'mfma_busy_frac': iss['mfma_busy_frac'], 'issue_slot_frac': iss['issue_slot_frac'],
                                 'wave_slots_occupied': iss['wave_slots_occupied'], 'pmc_source': iss['file']})
                roof['bound_note'] = ('the fused kernel is bound by instruction ISSUE (VALU + LDS + SALU of the fp64 residual phases and the fp32 Cholesky), not by the '
                                      'matrix cores: `frac` prices the algorithmic J^T J flop against the MFMA peak as the contract asks, frac_of_executed_flop / '
                                      'mfma_busy_frac say what the MFMA pipe did, issue_slot_frac / wave_slots_occupied (of 2) what bounds the kernel')
        else:
            ach = ALG_FLOP_PER_EVAL * n_evals / (ev_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'fit_eval_kernel (J^T J, v_mfma_f32_16x16x4_f32)', 'achieved': ach,
                    'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP32_PEAK_TFLOPS, 'traffic': None,
                    'alg_flop_per_unit': ALG_FLOP_PER_EVAL, 'units_per_launch_avg': n_evals / ev_n,
                    'avg_launch_us': 1e3 * ev_ms / ev_n, 'launches': int(ev_n),
                    'step_kernel_avg_launch_us': 1e3 * stp_ms / stp_n, 'eval_ms_total': ev_ms, 'step_ms_total': stp_ms}

        # ---- the contraction alone: fit_jtj_kernel on the records a previous launch (d2d_fit_rows) left in HBM -----------
        def iso(dsc_i, q_i, nit):
            Bi = dsc_i.shape[0]
            plan.rows(dsc_i, q_i)
            for _ in range(3):
                plan.jtj(Bi, want_H=False)
            torch.cuda.synchronize()
            # nit back-to-back launches between ONE event pair on the library's stream (an event pair per launch adds its own ~3 us
            # to a 28 us kernel: the rocprofv3 kernel trace of the same launches is the cross-check in profiles/)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            for _ in range(nit):
                plan.jtj(Bi, want_H=False)
            e1.record(ctx.stream)
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / nit
            tf = ALG_FLOP_PER_EVAL * Bi / (us * 1e-6) / 1e12
            return {'units_per_launch': Bi, 'avg_launch_us': us, 'launches': int(nit), 'achieved': tf, 'frac': tf / FP32_PEAK_TFLOPS,
                    'alg_hbm_gbs': JTJ_BYTES_PER_UNIT * Bi / (us * 1e-6) / 1e9}
        i4 = iso(dsc, q0, 40)
        roof_iso = {'bound': 'mfma', 'kernel': 'fit_jtj_kernel<3,24> (contraction only: fp32 row records HBM -> LDS, J^T J on v_mfma_f32_16x16x4_f32, '
                                               'tile-major store)', 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'alg_flop_per_unit': ALG_FLOP_PER_EVAL, 'alg_bytes_per_unit': JTJ_BYTES_PER_UNIT,
                    'traffic': pmc_traffic('fit_jtj_kernel')[0], 'traffic_source': pmc_traffic('fit_jtj_kernel')[1],
                    'alg_bytes_per_launch': JTJ_BYTES_PER_UNIT * B, **i4,
                    'note': 'the whole kernel is the contraction (one HIP event pair around back-to-back launches; rocprofv3 kernel trace in profiles/); '
                            'ceiling of frac = 0.766: the three diagonal 16x16 tiles are computed whole'}
        if world == 1 and a.config3_batch > B:
            dsci = ctx.dev(bench_scenarios(a.config3_batch, 0))
            roof_iso['large'] = iso(dsci, plan.init(dsci), 20)
            del dsci
            torch.cuda.empty_cache()

    # ---- BASELINE configs[3]: 32 768 fits per rank (256 k at N = 8), every rank, same timing discipline -------------------
    config3 = None
    if a.config3_batch > 0:
        B3 = a.config3_batch
        dt3, r3, ne3, prof3, glob3 = timed_solves(B3, a.config3_steps, 1, a.order_hint, a.mode, keep=False)
        tot3 = B3 * world
        config3 = {'workload': f'{B3} fits per GPU ({tot3} in total), sharded by trajectory, convergence all-reduce (BASELINE configs[3])',
                   'unit': 'trajectory-optimisations/s', 'n_gpus': world, 'per_gpu_batch': B3, 'steps': a.config3_steps, 'solver': a.mode,
                   'handout': 'longest-first by the previous solve\'s iteration counts' if a.order_hint else HANDOUT_TXT[a.handout],
                   **summary(B3, a.config3_steps, dt3, glob3, prof3, ne3)}
        if not a.no_extra_modes:
            dt3f, _, ne3f, prof3f, glob3f = timed_solves(B3, a.config3_steps, 1, False, 'fast', keep=False)
            config3['fast_mode'] = summary(B3, a.config3_steps, dt3f, glob3f, prof3f, ne3f)
        del r3
        torch.cuda.empty_cache()

    # ---- BASELINE configs[2]: multi_opt_planner's 8-drone circular formation x 8192 replicas, collision rows between all pairs ----
    config2 = None
    if rank == 0 and world == 1 and not a.no_groups:
        from d2dhip import synth
        n_ac, Rg = 8, 8192
        plan_g = d2dhip.FitPlan(ctx, S_, K, dur, synth.default_wref(1.0, K))
        dscg = ctx.dev(synth.circle_group_scenarios(n_ac, Rg, dur, K, seed=1).reshape(Rg * n_ac, -1))
        q0g = plan_g.init(dscg)
        best, cold, resg = 1e30, 1e30, None
        GTOL, GSWEEPS = 1e-6, 200        # the north-star's tolerance: every scenario sweeps until ITS largest relative move is <= 1e-6
        rep_cold = None
        for rep in range(5):
            # reps 0-1: no scheduling hint (rep 0 also warms the launch up); reps 2-4: the scenarios that swept longest in the previous
            # solve start first (d2d_fit_plan_set_group_order: the replanning pattern, like the order hint of the headline solve)
            if rep == 2:
                plan_g.group_order_from_last(Rg)
            qg = q0g.clone()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            resg = plan_g.solve_groups(dscg, qg, n_ac, max_sweeps=GSWEEPS, inner_iters=8, tol=GTOL)
            torch.cuda.synchronize()
            if rep == 1:
                cold = time.perf_counter() - tg
                rep_cold = plan_g.group_report(Rg)
                evals_cold = float(resg[2][3])
            if rep >= 2:
                best = min(best, time.perf_counter() - tg)
        sw, mv = rep_cold
        # the same solve with the line search on the joint cost switched off (plain block Gauss-Seidel), no hint: what the slow
        # scenarios cost without it
        plan_g.group_order_from_last(Rg, False)
        qg = q0g.clone()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        plan_g.solve_groups(dscg, qg, n_ac, max_sweeps=GSWEEPS, inner_iters=8, tol=GTOL, gs_ls=0)       # d2d_fit_opts.gs_ls = 0
        torch.cuda.synchronize()
        plain_s = time.perf_counter() - tg
        sw_p, mv_p = plan_g.group_report(Rg)
        config2 = {'workload': '8-drone circular formation x 8192 replicas (65 536 coupled trajectories), CostCollision rows between all pairs '
                               '(BASELINE configs[2]); block Gauss-Seidel per scenario in one persistent launch (fit_groups_kernel)',
                   'tol': GTOL, 'max_sweeps': GSWEEPS,
                   # the solve WITHOUT any scheduling hint is the record's value; the hinted figure is beside it
                   'value': Rg / cold, 'unit': 'scenarios/s', 'trajectories_per_s': Rg * n_ac / cold, 'seconds': cold,
                   'seconds_with_order_hint': best,
                   'settled_frac': float((mv <= GTOL).mean()), 'last_sweep_max_rel_move': float(mv.max()),
                   'sweeps_mean': float(sw.mean()), 'sweeps_p50': int(np.percentile(sw, 50)), 'sweeps_p99': int(np.percentile(sw, 99)), 'sweeps_max': int(sw.max()),
                   'scenarios_beyond_40_sweeps': int((sw > 40).sum()),
                   'target_seconds': 0.06, 'meets_target': bool(cold <= 0.06),
                   'evaluations': evals_cold,           # of the same cold solve as `seconds`
                   'jtj_frac_of_fp32_mfma_peak': ALG_FLOP_PER_EVAL * evals_cold / cold / 1e12 / FP32_PEAK_TFLOPS,
                   'line_search': 'on the joint cost along slow sweeps (include/d2d.h D2D_GS_LS_*); same fixed points as the plain sweeps',
                   'plain_sweeps': {'seconds': plain_s, 'sweeps_mean': float(sw_p.mean()), 'sweeps_p99': int(np.percentile(sw_p, 99)), 'sweeps_max': int(sw_p.max()),
                                    'scenarios_beyond_40_sweeps': int((sw_p > 40).sum()), 'settled_frac': float((mv_p <= GTOL).mean())},
                   'round1_seconds': 0.80, 'round3_seconds_tol_1e-10_cap_120_unsettled': 0.107}
        plan_g.close()
        del dscg, q0g, qg
        torch.cuda.empty_cache()

    # ---- parity: the SAME scenarios the cpu_baseline leg solved ------------------------------------------------------------
    parity = None
    if keep is not None:
        n = len(keep['cost'])
        sc_dev = dsc[:n]

        def agreement(cg, zg):
            rel_c = np.abs(cg - keep['cost']) / np.maximum(np.abs(keep['cost']), 1e-300)
            rel_z = np.abs(zg - keep['z']).reshape(n, -1).max(1) / np.abs(keep['z']).reshape(n, -1).max(1)
            return rel_c, rel_z, (rel_c <= 1e-6) & (rel_z <= 1e-6)

        def four_numbers(name, cg, qg, same, mode):
            """For every scenario on which the GPU and scipy end in different points: (i) scipy's cost, (ii) the GPU's cost, (iii) the cost
            after a scipy solve started FROM the GPU's point, (iv) the cost after a GPU solve (same mode) started from scipy's point --
            (iii) = (ii) and (iv) = (i) to 1e-6 (cost and unknowns) says both are genuine stationary points: two minima, not a failure."""
            idx = np.nonzero(~same)[0]
            rec = {'n': int(len(idx)), 'gpu_lower': int((cg[idx] < keep['cost'][idx]).sum()), 'gpu_higher': int((cg[idx] > keep['cost'][idx]).sum())}
            if len(idx) == 0:
                return rec
            import multiprocessing as mp
            with mp.get_context('spawn').Pool(min(_host_cores(), 8)) as pool:
                pol = pool.map(_cpu_fit_one, [(keep['basis'], keep['sc'][i], qg[i]) for i in idx], chunksize=2)
            c3 = np.array([r[0] for r in pol]); q3 = np.array([r[1] for r in pol])
            qs = ctx.dev(np.ascontiguousarray(keep['q'][idx]))
            c4, *_ = plan.solve(ctx.dev(np.ascontiguousarray(keep['sc'][idx])), qs, max_iter=600 if mode == 'minpack_pure' else a.max_iter, **MODES[mode])
            c4 = c4.cpu().numpy(); q4 = qs.cpu().numpy()
            stay3 = (np.abs(c3 - cg[idx]) <= 1e-6 * cg[idx]) & (np.abs(q3 - qg[idx]).max(1) <= 1e-6 * np.abs(qg[idx]).max(1))
            stay4 = (np.abs(c4 - keep['cost'][idx]) <= 1e-6 * keep['cost'][idx]) & (np.abs(q4 - keep['q'][idx]).max(1) <= 1e-6 * np.abs(keep['q'][idx]).max(1))
            # scipy stops on ftol 1e-15 of a linearly converging Gauss-Newton tail: a GPU solve from its point may still descend a little
            lower4 = c4 <= keep['cost'][idx] * (1 + 1e-9)
            rec.update({'scipy_from_gpu_point_stays': int(stay3.sum()), 'gpu_from_scipy_point_stays': int(stay4.sum()),
                        'gpu_from_scipy_point_not_higher': int(lower4.sum()), 'both_certified_minima': int((stay3 & (stay4 | lower4)).sum()),
                        'rows_first_16': [{'scenario': int(i), 'scipy_cost': float(keep['cost'][i]), 'gpu_cost': float(cg[i]),
                                           'scipy_from_gpu_point': float(c3[k]), 'gpu_from_scipy_point': float(c4[k])} for k, i in enumerate(idx[:16])]})
            return rec

        cg, zg = c_head[:n], z_head[:n]
        rel_c, rel_z, same = agreement(cg, zg)
        no = len(keep['o_cost'])
        qg = q_head[:n].cpu().numpy()
        rel_co = np.abs(cg[:no] - keep['o_cost']) / np.maximum(np.abs(keep['o_cost']), 1e-300)
        rel_qo = np.abs(qg[:no] - keep['o_q']).max(1) / np.abs(keep['o_q']).max(1)
        pc, pq = cpu_polish(keep, qg, 256)
        npol = len(pc)
        qgp = qg[:npol]
        parity = {'scenarios': n, 'solver': a.mode,
                  'what': 'GPU (d2d_fit_solve, the headline\'s solver) vs scipy.optimize.least_squares(lm) from the same start on the same scenarios; '
                          'same = cost AND the 96 monomial coefficients within 1e-6 relative',
                  'same_minimum_frac': float(same.mean()), 'cost_rel_le_1e-6_frac': float((rel_c <= 1e-6).mean()),
                  'coeff_rel_le_1e-6_frac': float((rel_z <= 1e-6).mean()),
                  'others': four_numbers(a.mode, cg, qg, same, a.mode),
                  'mean_cost_gpu': float(cg.mean()), 'mean_cost_cpu': float(keep['cost'].mean()),
                  'vs_oracle_lm': {'scenarios': no, 'what': ('GPU vs oracle/fit_knot.py solve_minpack_knot' if KNOT_DEFAULT else 'GPU vs oracle/fit.py solve_minpack') + ' (the CPU statement of the same algorithm with the same precision split)'
                                                            if a.mode == 'minpack' else 'GPU (fast mode) vs oracle/fit.py solve_minpack: different algorithms',
                                   'same_minimum_frac': float(((rel_co <= 1e-6) & (rel_qo <= 1e-6)).mean())},
                  'polish': {'scenarios': npol, 'what': 'scipy LM (tol 1e-15) started from the GPU solutions: largest relative move',
                             'max_rel_cost_move': float(np.max(np.abs(pc - cg[:npol]) / np.abs(cg[:npol]))),
                             'max_rel_q_move': float(np.max(np.abs(pq - qgp).max(1) / np.abs(qgp).max(1))),
                             'frac_within_1e-6': float(((np.abs(pc - cg[:npol]) / np.abs(cg[:npol]) <= 1e-6) &
                                                        (np.abs(pq - qgp).max(1) / np.abs(qgp).max(1) <= 1e-6)).mean())}}
        for name, mode_v in (('minpack_pure', 'minpack_pure'), ('fast_mode', 'fast')):
            if name in gpu_sol:
                cgv, qgv, zgv = gpu_sol[name]
                rcv, rzv, samev = agreement(cgv[:n], zgv[:n])
                parity[name] = {'same_minimum_frac': float(samev.mean()), 'cost_rel_le_1e-6_frac': float((rcv <= 1e-6).mean()),
                                'mean_cost_gpu': float(cgv[:n].mean()), 'others': four_numbers(name, cgv[:n], qgv[:n], samev, mode_v)}

    # ---- BASELINE configs[4] ---------------------------------------------------------------------------------------------
    sim = None
    if rank == 0 and world == 1 and not a.no_sim:
        del dsc, q0, q, cost, iters, status
        torch.cuda.empty_cache()
        sim = sim_records(ctx, torch, cpu_g, cpu_t)
    nlp = longh = None
    if rank == 0 and world == 1 and not a.no_nlp:
        torch.cuda.empty_cache()
        nlp = nlp_record(ctx, torch, cpu_n)
        longh = long_horizon_record(ctx, torch, d2dhip, cpu_long=cpu_l)

    if rank == 0:
        line = {
            'metric': 'trajectory-optimisations/sec (6-seg poly, 50 wpts)', 'value': headline['value'],
            'unit': 'trajectory-optimisations/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': headline['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64 residual/gradient + f32 MFMA J^T J', 'data': 'synthetic',
            'config': {'workload': f'batch={B} per GPU independent single-drone 6-seg poly fits, 50 waypoints (BASELINE configs[1])',
                       'segments': S_, 'samples': K, 'unknowns_reduced': NQ2, 'max_iter': a.max_iter,
                       'check_every': a.check_every,
                       'solver': 'MINPACK lmder path on the normal equations + second-order finish (d2d_fit_opts.mode = D2D_LM_MODE_MINPACK, the library default)'
                                 if a.mode == 'minpack' else 'D2D_LM_MODE_FAST',
                       'kernel': plan.kernel,
                       'handout': 'longest-first by the iteration counts of the previous solve of the same batch (warmup)' if a.order_hint else HANDOUT_TXT[a.handout if a.mode == 'minpack' else 'index'],
                       'parallelism': f'trajectory-sharded x{world}'},
            'converged_frac': headline['converged_frac'], 'stalled_frac': headline['stalled_frac'], 'mean_iters': headline['mean_iters'],
            'evals_per_fit': headline['evals_per_fit'],          # Gauss-Newton units (200 rows); second-order evaluations count 1.5
            'mean_cost': headline['mean_cost'],
            # -- scalars of the nested records, hoisted so that a reader of the head of this line has the north-star's numbers --
            'parity_same_minimum_frac': parity['same_minimum_frac'] if parity else None,                   # vs scipy least_squares('lm'), same start
            'parity_vs_oracle_frac': parity['vs_oracle_lm']['same_minimum_frac'] if parity else None,       # vs oracle/fit.py solve_minpack
            'roofline_frac': roof['frac'] if roof else None,
            'roofline_isolated_frac_4096': roof_iso['frac'] if roof_iso else None,
            'roofline_isolated_frac_32768': roof_iso['large']['frac'] if roof_iso and 'large' in roof_iso else None,
            'config2_seconds': config2['seconds'] if config2 else None, 'config2_settled_frac': config2['settled_frac'] if config2 else None,
            'config3_value': config3['value'] if config3 else None,
            'nlp_value': nlp['value'] if nlp else None,
            'long_horizon_value_121': longh['value'] if longh else None,
            'sim_gvf_value': sim['gvf']['value'] if sim else None, 'sim_track_value': sim['track']['value'] if sim else None,
            'sim_gvf_value_2_waves_per_simd': sim['gvf']['occupancy_scaling'][1]['value'] if sim else None,
            'cpu_baseline_value': cpu['value'] if cpu else None, 'cpu_baseline_cores': cpu['cores'] if cpu else None,
            # -- multi-GPU: which collective ran, on how many RCCL ranks, and the per-rank spread of the solver kernel --
            'collective': reducer.collective, 'rccl_ranks': reducer.rccl_ranks,
            'rank_kernel_ms_per_step_min_max': list(head_rank_ms),
        }
        detail = dict(line, variants=variants, roofline=roof, roofline_isolated=roof_iso, config2=config2, config3=config3, parity=parity, sim=sim, nlp=nlp,
                      long_horizon=longh, cpu_baseline=cpu)
        emit(line, detail, roof, cpu)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
