#!/usr/bin/env python3
"""Development (CPU only): what lmpar's factorisations are spent on over the bench scenarios, and what cheaper orders of the same
solves would do to the count and to the agreement with scipy.  Usage: dev_lmpar.py [N] [variant ...]"""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'drone-sim-python_amd'))
import numpy as np
import multiprocessing as mp
import bench
from oracle import fit as F
from d2dhip import synth

K, S_ = 50, 6
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
VARIANTS = sys.argv[2:] or ['base', 'spec']
dur = synth.planner_timing(0, 4.9, 10)[2]; wref = synth.default_wref(0.1, K)
basis = F.FitBasis(S_, K, dur, wref)
sc_all = synth.synth_scenarios(4096, seed=20241008, rank=0, obj_scale=0.1, K=K)
P1 = 0.1
STATS = {}


def bump(k, v=1):
    STATS[k] = STATS.get(k, 0) + v


class Cache:
    """GN solve of the current point, kept across the rejected trials of a point (the kernel does the same)."""
    H = None


def gn_solve(H, g, dt):
    if Cache.H is H:
        return Cache.val
    bump('fac'); bump('fac_gn')
    p, isq = F._chol_solve(H, g, dt)
    if p is None:
        Cache.val = (None, np.inf, 0.0)
    else:
        dx = float(np.linalg.norm(p))
        Cache.val = (p, dx, isq(p / dx))
    Cache.H = H
    return Cache.val


def damped(H, g, par, dt):
    bump('fac'); bump('fac_damped')
    p, isq = F._chol_solve(H + par * np.eye(len(g)), g, dt)
    return p, isq


def make_lmpar(variant):
    def lmpar(H, g, delta, par, chol_dtype=np.float64):
        n = len(g)
        bump('calls')
        gnorm = float(np.linalg.norm(g))
        paru0 = gnorm / delta
        spec = None
        if variant == 'spec' and par > 0.0 and Cache.H is not H and par <= paru0:
            # the damped solve lmpar would do first if parl <= par: done before the Gauss-Newton solve
            p, isq = damped(H, g, par, chol_dtype)
            if p is not None:
                dxn = float(np.linalg.norm(p)); fp = dxn - delta
                t2 = isq(p / dxn)
                lower_gn = dxn * (1.0 + par * t2)       # convexity: ||p(0)|| >= ||p(par)|| - par * d||p||/dpar
                if -P1 * delta <= fp <= 0.0 and lower_gn > (1.0 + P1) * delta:
                    bump('spec_hit'); bump('iters', 1)
                    return p, par, 1
                spec = (p, isq, dxn, t2)
                bump('spec_miss_below' if fp < 0 else 'spec_miss_above')
        p, dxnorm, t2gn = gn_solve(H, g, chol_dtype)
        it = 0
        if p is not None:
            fp = dxnorm - delta
            if fp <= P1 * delta:
                bump('gn_inside')
                if spec is not None:
                    bump('spec_wasted')
                return p, 0.0, 1
            parl = (fp / delta) / t2gn if t2gn > 0.0 else 0.0
        else:
            dxnorm, fp, parl = np.inf, np.inf, 0.0
        paru = paru0
        if paru == 0.0:
            paru = F.MP_DWARF / min(delta, P1)
        par_in = par
        if variant == 'parl':
            par = 0.0 if parl > 0 else par
        par = min(max(par, parl), paru)
        if par == 0.0:
            par = gnorm / dxnorm
        if par_in > 0 and par != par_in:
            bump('par_clipped')
        while True:
            it += 1
            if par == 0.0:
                par = max(F.MP_DWARF, 0.001 * paru)
            if spec is not None and par == par_in:
                p, isq = spec[0], spec[1]
                spec = None
                bump('spec_reused')
            else:
                if spec is not None:
                    bump('spec_wasted'); spec = None
                p, isq = damped(H, g, par, chol_dtype)
            if p is None:
                parl = max(parl, par); par = max(2.0 * par, 0.001 * paru)
                if it >= 10:
                    return np.zeros(n), par, it
                continue
            dxnorm = float(np.linalg.norm(p))
            temp = fp
            fp = dxnorm - delta
            if abs(fp) <= (0.01 if variant == 'tight' else P1) * delta or (parl == 0.0 and fp <= temp and temp < 0.0) or it == 10:
                break
            temp2 = isq(p / dxnorm)
            parc = (fp / delta) / temp2
            if fp > 0.0:
                parl = max(parl, par)
            if fp < 0.0:
                paru = min(paru, par)
            par = max(parl, par + parc)
        bump('iters', it); bump(f'it{min(it, 5)}')
        return p, par, it
    return lmpar


def scipy_one(i):
    return bench._cpu_fit_one((basis, sc_all[i], None))


def oracle_one(a):
    i, variant = a
    STATS.clear(); Cache.H = None
    F.lmpar_normal = make_lmpar(variant)
    q, c, it, st, info = F.solve_minpack(basis, sc_all[i], finish=3, hess_dtype=np.float32, chol_dtype=np.float32)
    return c, q, it, st, info['mp_trials'], dict(STATS)


if __name__ == '__main__':
    cache = f'/tmp/scipy_{N}.pkl'
    with mp.get_context('fork').Pool(8) as pool:
        if os.path.exists(cache):
            sres = pickle.load(open(cache, 'rb'))
        else:
            sres = pool.map(scipy_one, range(N), chunksize=2)
            pickle.dump(sres, open(cache, 'wb'))
        cs = np.array([r[0] for r in sres]); qs = np.array([r[1] for r in sres])
        for v in VARIANTS:
            res = pool.map(oracle_one, [(i, v) for i in range(N)], chunksize=2)
            co = np.array([r[0] for r in res]); qo = np.array([r[1] for r in res]); ito = np.array([r[2] for r in res])
            mpt = np.array([r[4] for r in res])
            same = (np.abs(co - cs) / cs <= 1e-6) & (np.abs(qo - qs).max(1) / np.abs(qs).max(1) <= 1e-6)
            tot = {}
            for r in res:
                for k, x in r[5].items():
                    tot[k] = tot.get(k, 0) + x
            print(f'{v}: same as scipy {same.mean():.4f}; trials/fit {ito.mean():.2f} (lmder {mpt.mean():.2f}) max {ito.max()}; '
                  f'factorisations/fit lmder {tot.get("fac", 0) / N:.2f} (+ finish {(ito - mpt).mean():.2f}); per lmder trial {tot.get("fac", 0) / mpt.sum():.3f}')
            print('   ', {k: round(x / N, 2) for k, x in sorted(tot.items())}, flush=True)
