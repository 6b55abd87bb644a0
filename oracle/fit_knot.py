"""ORACLE (test infrastructure, CPU): the knot-space statement of the polynomial fit's default solver -- the algorithm the
round-5 kernel `fit_lm_knot_kernel` (csrc/fit_knot.h) runs.  Only tests/, __graft_entry__.smoke() and bench.py's cpu legs import it.

Same problem, same path, other coordinates.  oracle/fit.py states the fit in the whitened reduced unknowns q (z_axis = Zp e + Z q,
Z = Nf L^-T with L L^T = Nf^T Mref Nf): J^T J is dense there.  The reference's own parameterisation of a C^3 piecewise degree-7
polynomial is LOCAL -- CompositeTraj([MinSnapPoly(Y_j, Y_j+1, T)]) (src/d2d/trajectory.py:166-208): knot data (position, velocity,
acceleration, jerk at the S+1 knots), every sample touching the two knots of its segment only -- so in knot coordinates

    u[knot j][axis a][k] = T^k / k! * Y_a^(k)(t_j)              (Taylor-scaled; (j in {0, S}, k in {0, 1}) are the end conditions)

J^T J is block tridiagonal (8 x 8 blocks: one knot, both axes), one 16 x 16 block per segment.  q and u are related by the affine map
q = B (u - u0) with B^T B = Mu, the (banded) reference metric in knot coordinates, so MINPACK's lmder in q -- unit scaling, trust region
||p||_2 <= Delta, lmpar on J^T J + par I: the path scipy.optimize.least_squares(method='lm') follows -- is, step for step,

    (H_u + par Mu) s = g_u,   ||p||_2 = sqrt(s^T Mu s),   ||R^-T (p / ||p||)||^2 = w^T (H_u + par Mu)^-1 w  with  w = Mu s / ||s||_Mu,
    ||J^T f||_2 = sqrt(g_u^T Mu^-1 g_u),   ||x||_2 = sqrt((u - u0)^T Mu (u - u0)),   ||J p||^2 = s^T g_u - par s^T Mu s

in u: the same iterates in exact arithmetic (tests/test_oracle_knot.py checks it against oracle/fit.py lmder_solve), with banded
factorisations.  What is NOT expressible with banded quantities are the component-wise pieces of the q statement; they are restated
here (and in the kernel) in knot coordinates:
  * lmder's gradient test max_j |g_j| / (||J e_j|| ||f||) <= gtol (gtol = 1e-15: never the exit taken) uses the columns of J_u;
  * the second-order finish (oracle/fit.py lm_solve) damps with lam * Mu -- lam * I in q, where lm_solve damps with lam * diag|H_q| --
    and takes its max-norm stop tests in the metric's diagonal scaling: |g_u,j| / sqrt(Mu_jj), |s_j| sqrt(Mu_jj).
"""
import math

import numpy as np

from . import fit as F

NK = 8           # entries per knot: 2 axes x (pos, vel, acc, jerk)
GTOL_SCALE = 0.1  # the finish's gradient test max_j |g_u,j| / sqrt(Mu_jj) <= GTOL_SCALE * gtol: in the metric's diagonal scaling the
                  # test is looser than max |g_q| <= gtol of the q statement by up to the spread of that scaling (csrc/fit_knot.hip KN_GTOL_SCALE)


def hermite_unit():
    """8 x 8 map [u(knot s) (4); u(knot s+1) (4)] (Taylor-scaled knot data) -> the coefficients c_0 .. c_7 of the segment's polynomial
    in x = tau / T on [0, 1]: rows of the degree-7 Hermite interpolation (the closed form of PolynomialOne.__init__,
    src/d2d/trajectory.py:54-66, in scaled coordinates)."""
    A = np.zeros((8, 8))
    for k in range(4):
        A[k, k] = 1.0                                   # u_k(0) = c_k
        for p in range(k, 8):
            A[4 + k, p] = math.comb(p, k)               # u_k(1) = sum_p C(p, k) c_p
    return np.linalg.inv(A)


class KnotBasis:
    """Shared (batch-independent) block of the knot-space statement, built from a FitBasis (either constructor)."""

    def __init__(self, basis):
        S, K, T = basis.S, basis.K, basis.T
        self.basis, self.S, self.K, self.T = basis, S, K, T
        nq = basis.nq
        assert nq == 4 * S, 'knot coordinates: C^3 piecewise degree-7 polynomials'
        N = F.junction_map(S, T)
        fixed, free = F.knot_split(S)
        Nf, Nx = N[:, free], N[:, fixed]
        Linv_T = np.linalg.lstsq(Nf, basis.Z, rcond=None)[0]          # Z = Nf L^-T
        Pe = np.linalg.lstsq(Nf, basis.Zp - Nx, rcond=None)[0]        # Zp = Nx + Nf P  (the free knot data of q = 0 per unit end datum)
        dsc = np.array([T ** (i % 4) / math.factorial(i % 4) for i in range(4 * (S + 1))])
        # per axis: q = Bax (u_free - u0),  u_free = dsc_free * d_free,  d_free = L^-T q + P e
        self.Bax = np.linalg.inv(Linv_T) / dsc[free][None, :]
        self.Pu = dsc[free][:, None] * Pe                             # (nq, 4): u0 = Pu e
        self.dsc, self.fixed, self.free = dsc, fixed, free
        self.Mu_ax = self.Bax.T @ self.Bax                            # banded (half-bandwidth 7), = dsc^-1 Nf^T Mref Nf dsc^-1
        # interleaved knot-major order of the 2 nq free unknowns: (knot, axis, k)
        key = [(f // 4, a, f % 4) for a in range(2) for f in free]
        self.order = np.array(sorted(range(2 * nq), key=lambda i: key[i]))
        self.full_index = np.array([NK * key[i][0] + 4 * key[i][1] + key[i][2] for i in self.order])   # position in the 8 (S+1) vector
        B = np.zeros((2 * nq, 2 * nq))
        B[:nq, :nq] = self.Bax; B[nq:, nq:] = self.Bax
        self.B = B[:, self.order]                                     # q - q(u0) = B (u - u0), u in knot-major order
        self.Mu = self.B.T @ self.B
        self.Mu_inv = np.linalg.inv(self.Mu)
        self.msc = np.sqrt(np.diag(self.Mu))
        self.Binv = np.linalg.inv(self.B)
        hw = max(abs(i - j) for i in range(2 * nq) for j in range(2 * nq) if abs(self.Mu[i, j]) > 1e-13 * np.abs(self.Mu).max())
        assert hw <= 15, hw
        # Hermite tables: Y^(d)(t_k) = sum_m Hb[d][k][m] v_m over the 8 scaled knot values of sample k's segment
        _, seg, tau, _ = F.sample_segments(K, S, basis.duration)
        Hu = hermite_unit()
        self.seg = seg
        self.Hb = np.zeros((3, K, 8))
        for k in range(K):
            x = tau[k] / T
            for d in range(3):
                row = np.array([F.arr(d, p) * x ** (p - d) if p >= d else 0.0 for p in range(8)]) / T ** d
                self.Hb[d, k] = row @ Hu

    def u0(self, sc):
        dx, dy = F.end_data(sc)
        return np.concatenate([self.Pu @ dx, self.Pu @ dy])[self.order]

    def to_u(self, sc, q):
        return self.u0(sc) + self.Binv @ q

    def to_q(self, sc, u):
        return self.B @ (u - self.u0(sc))

    def full_knots(self, sc, u):
        """the 8 (S+1) vector [knot][axis][k] with the end conditions filled in"""
        w = np.zeros(NK * (self.S + 1))
        w[self.full_index] = u
        dx, dy = F.end_data(sc)
        for a, e in enumerate((dx, dy)):
            for (j, k, v) in ((0, 0, e[0]), (0, 1, e[1]), (self.S, 0, e[2]), (self.S, 1, e[3])):
                w[NK * j + 4 * a + k] = self.dsc[k] * v
        return w

    def flat_outputs(self, sc, u):
        """Y (3, 2, K) from the knot data through the Hermite tables -- must equal F.flat_outputs at q = to_q(u)"""
        w = self.full_knots(sc, u).reshape(self.S + 1, 2, 4)
        Y = np.zeros((3, 2, self.K))
        for k in range(self.K):
            s = self.seg[k]
            for a in range(2):
                v = np.concatenate([w[s, a], w[s + 1, a]])
                for d in range(3):
                    Y[d, a, k] = self.Hb[d, k] @ v
        return Y

    def eval_normal(self, sc, u, wp=None, second_order=False, hess_dtype=np.float64):
        """cost, g_u = J_u^T r, H_u = J_u^T J_u (+ second-order term) -- through the q statement (J_u = J_q B): the numbers a direct
        knot-space evaluation produces, to rounding (the kernel's direct evaluation is compared with this in tests/)."""
        c, g, H = F.eval_normal(self.basis, sc, self.to_q(sc, u), wp, second_order=second_order)
        Hu = self.B.T @ H @ self.B
        return c, self.B.T @ g, Hu.astype(hess_dtype).astype(np.float64)

    def cost(self, sc, u, wp=None):
        return F.cost(self.basis, sc, self.to_q(sc, u), wp)


def _chol_solve(A, b, dtype):
    import scipy.linalg as sl
    try:
        Lc = np.linalg.cholesky(A.astype(dtype))
    except np.linalg.LinAlgError:
        return None, None
    y = sl.solve_triangular(Lc, b.astype(dtype), lower=True)
    s = sl.solve_triangular(Lc.T, y, lower=False).astype(np.float64)
    return s, (lambda w: float(np.sum(sl.solve_triangular(Lc, w.astype(dtype), lower=True).astype(np.float64) ** 2)))


def lmpar_knot(kb, H, g, delta, par, chol_dtype=np.float64):
    """oracle/fit.py lmpar_normal in knot coordinates (module docstring).  Returns s (the step is -s), par, factorisations."""
    Mu = kb.Mu
    nfac = 1
    s, isq = _chol_solve(H, g, chol_dtype)
    it = 0
    if s is not None:
        dxnorm = math.sqrt(float(s @ Mu @ s))
        fp = dxnorm - delta
        if fp <= F.MP_P1 * delta:
            return s, 0.0, nfac, dxnorm
        temp2 = isq(Mu @ s / dxnorm)
        parl = (fp / delta) / temp2 if temp2 > 0.0 else 0.0
    else:
        dxnorm, fp, parl = np.inf, np.inf, 0.0
    gnorm = math.sqrt(max(float(g @ kb.Mu_inv @ g), 0.0))
    paru = gnorm / delta
    if paru == 0.0:
        paru = F.MP_DWARF / min(delta, F.MP_P1)
    par = min(max(par, parl), paru)
    if par == 0.0:
        par = gnorm / dxnorm
    while True:
        it += 1
        if par == 0.0:
            par = max(F.MP_DWARF, 0.001 * paru)
        s, isq = _chol_solve(H + par * Mu, g, chol_dtype)
        nfac += 1
        if s is None:
            parl = max(parl, par); par = max(2.0 * par, 0.001 * paru)
            if it >= 10:
                return np.zeros(len(g)), par, nfac, 0.0
            continue
        dxnorm = math.sqrt(float(s @ Mu @ s))
        temp = fp
        fp = dxnorm - delta
        if abs(fp) <= F.MP_P1 * delta or (parl == 0.0 and fp <= temp and temp < 0.0) or it == 10:
            break
        temp2 = isq(Mu @ s / dxnorm)
        parc = (fp / delta) / temp2
        if fp > 0.0:
            parl = max(parl, par)
        if fp < 0.0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return s, par, nfac, dxnorm


def lmder_knot(kb, sc, u0=None, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=None, hess_dtype=np.float64, chol_dtype=np.float64,
               finish=None, slow=None, trace=None):
    """oracle/fit.py lmder_solve in knot coordinates: same decisions, same hand-over rules (finish / slow)."""
    basis = kb.basis
    wp = F.waypoints(sc, basis.K, basis.duration)
    u = kb.to_u(sc, F.initial_guess(basis, sc, wp)) if u0 is None else np.array(u0, float)
    ub = kb.u0(sc)
    n = len(u)
    if max_nfev is None:
        max_nfev = 100 * n
    c, g, H = kb.eval_normal(sc, u, wp, hess_dtype=hess_dtype)
    nfev, nfac = 1, 0
    if not np.isfinite(c):
        return u, c, nfev, F.ST_NONFINITE, {}
    fnorm = math.sqrt(c)
    par = 0.0
    xnorm = math.sqrt(float((u - ub) @ kb.Mu @ (u - ub)))
    delta = F.MP_FACTOR * xnorm if xnorm > 0 else F.MP_FACTOR
    first, info, calm, nslow = True, 0, 0, 0
    while True:
        acn = np.sqrt(np.maximum(np.diag(H), 0.0))
        gnorm = 0.0
        if fnorm != 0.0:
            ok = acn > 0
            gnorm = float(np.max(np.abs(g[ok]) / (acn[ok] * fnorm))) if ok.any() else 0.0
        if gnorm <= gtol:
            info = 4
            break
        while True:
            s, par, nf, pnorm = lmpar_knot(kb, H, g, delta, par, chol_dtype)
            nfac += nf
            if first:
                delta = min(delta, pnorm)
                first = False
            ut = u - s
            ct = kb.cost(sc, ut, wp)
            nfev += 1
            fnorm1 = math.sqrt(ct) if np.isfinite(ct) else np.inf
            actred = -1.0
            if F.MP_P1 * fnorm1 < fnorm:
                actred = 1.0 - ct / c
            jp2 = max(float(s @ g) - par * pnorm * pnorm, 0.0)
            t1 = jp2 / c
            t2 = par * pnorm * pnorm / c
            prered = t1 + t2 / F.MP_P5
            dirder = -(t1 + t2)
            ratio = actred / prered if prered != 0.0 else 0.0
            if ratio <= F.MP_P25:
                temp = F.MP_P5 if actred >= 0.0 else F.MP_P5 * dirder / (dirder + F.MP_P5 * actred)
                if F.MP_P1 * fnorm1 >= fnorm or temp < F.MP_P1:
                    temp = F.MP_P1
                delta = temp * min(delta, pnorm / F.MP_P1)
                par = par / temp
            elif par == 0.0 or ratio >= F.MP_P75:
                delta = pnorm / F.MP_P5
                par = F.MP_P5 * par
            if trace is not None:
                trace.append((nfev, c, ct, par, delta, ratio))
            accepted = ratio >= F.MP_P0001
            if accepted:
                u = ut
                xnorm = math.sqrt(float((u - ub) @ kb.Mu @ (u - ub)))
                c, fnorm = ct, fnorm1
            if abs(actred) <= ftol and prered <= ftol and F.MP_P5 * ratio <= 1.0:
                info = 1
            if delta <= xtol * xnorm:
                info = 2 if info == 0 else 3
            if info == 0:
                if nfev >= max_nfev:
                    info = 5
                elif abs(actred) <= F.MP_EPS and prered <= F.MP_EPS and F.MP_P5 * ratio <= 1.0:
                    info = 6
                elif delta <= F.MP_EPS * xnorm:
                    info = 7
                elif gnorm <= F.MP_EPS:
                    info = 8
            if accepted:
                calm = calm + 1 if (par == 0.0 and ratio >= F.MP_P75) else 0
            nslow = nslow + 1 if abs(actred) <= F.MP_SLOW_TOL else 0
            if info != 0 or accepted:
                break
        if info != 0:
            break
        if finish is not None and (calm >= finish or (slow and nslow >= slow)):
            break
        c, g, H = kb.eval_normal(sc, u, wp, hess_dtype=hess_dtype)
    status = F.ST_CONVERGED if info in (1, 2, 3, 4, 6, 7, 8) else F.ST_MAXITER
    return u, c, nfev, status, {'info': info, 'nfac': nfac, 'handover': info == 0}


def finish_knot(kb, sc, u, max_iter=200, ftol=1e-14, gtol=1e-9, xtol=1e-11, hess_dtype=np.float64, chol_dtype=np.float64,
                lam0=F.LM_LAMBDA0, scaling='M', stats=None):
    """The second-order loop (oracle/fit.py lm_solve with so_lambda = inf: every evaluation carries the exact Hessian) in knot
    coordinates: (H_u + lam Dg) s = -g_u, Dg = Mu (`scaling` 'M': lam I in q) -- 'diag': diag|H_u|, for comparison --, Nielsen's gain
    ratio, shortened steps along a rejected direction, LM_FAIL_MULT after a failed factorisation; max-norm tests in the diagonal
    scaling of the metric.  Returns u, cost, iterations, status."""
    basis = kb.basis
    wp = F.waypoints(sc, basis.K, basis.duration)
    ub = kb.u0(sc)
    lam, nu = lam0, 2.0
    status = F.ST_MAXITER
    c, g, H = kb.eval_normal(sc, u, wp, second_order=True, hess_dtype=hess_dtype)
    it = 0
    for it in range(1, max_iter + 1):
        if np.max(np.abs(g) / kb.msc) <= GTOL_SCALE * gtol:
            status = F.ST_CONVERGED
            break
        Dg = kb.Mu if scaling == 'M' else np.diag(np.maximum(np.abs(np.diag(H)), F.LM_DIAG_FLOOR))
        s, _ = _chol_solve(H + lam * Dg, -g, chol_dtype)
        ok = s is not None
        rho, fin, accept = -1.0, False, False
        ct, pred = np.inf, 0.0
        if ok:
            ct = kb.cost(sc, u + s, wp)
            pred = float(s @ (lam * (Dg @ s) - g))
            fin = bool(np.isfinite(ct) and pred > 0)
            if fin:
                rho = (c - ct) / pred
        step, pred_s = None, pred
        if rho > 0:
            accept, step = True, s
            lam_new = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), F.LM_LAMBDA_MIN)
        elif fin:
            a = -2.0 * float(g @ s)
            b = a - pred
            den = 2.0 * (ct - c + a)
            al = a / den if den > 0.0 else F.LM_BT_MAX
            al = min(max(al, F.LM_BT_MIN), F.LM_BT_MAX)
            for _ in range(2):
                c2 = kb.cost(sc, u + al * s, wp)
                if np.isfinite(c2) and c2 < c:
                    accept, step, ct = True, al * s, c2
                    pred_s = a * al - b * al * al
                    lam_new = min(lam / al, F.LM_LAMBDA_MAX)
                    break
                al = max(F.LM_BT_SHRINK * al, F.LM_BT_FLOOR)
        if accept:
            small_x = np.max(np.abs(step) * kb.msc) <= xtol * (np.max(np.abs(u - ub) * kb.msc) + xtol)
            u = u + step
            lam, nu = lam_new, 2.0
            small_f = (c - ct) <= ftol * c and pred_s <= ftol * c
            c, g, H = kb.eval_normal(sc, u, wp, second_order=True, hess_dtype=hess_dtype)
            if small_f or small_x:
                status = F.ST_CONVERGED
                break
        else:
            if ok and fin and pred <= ftol * c:
                status = F.ST_CONVERGED
                break
            if ok:
                lam *= nu; nu *= 2.0
            else:
                if stats is not None:
                    stats['fails'] = stats.get('fails', 0) + 1
                lam *= F.LM_FAIL_MULT
            if lam > F.LM_LAMBDA_MAX:
                status = F.ST_STALLED
                break
    return u, c, it, status


def solve_minpack_knot(kb, sc, q0=None, finish=F.MP_FINISH, max_iter=200, mp_tol=1e-15, ftol=1e-14, gtol=1e-9, xtol=1e-11,
                       hess_dtype=np.float64, chol_dtype=np.float64, slow=F.MP_SLOW, scaling='M'):
    """The default solver in knot coordinates: lmder until calm (or stagnating), then the second-order finish.  Takes and returns q
    (the public unknowns).  Returns q, cost, trial points, status, info."""
    u0 = None if q0 is None else kb.to_u(sc, np.asarray(q0, float))
    u, c, nfev, st, info = lmder_knot(kb, sc, u0, ftol=mp_tol, xtol=mp_tol, gtol=mp_tol, max_nfev=max_iter + 1, hess_dtype=hess_dtype,
                                      chol_dtype=chol_dtype, finish=finish if finish > 0 else None, slow=slow if finish > 0 else None)
    it = nfev - 1
    out = {'nfac': info.get('nfac', 0), 'mp_trials': it, 'handover': info.get('handover', False)}
    if out['handover'] and it < max_iter:
        u, c, it2, st = finish_knot(kb, sc, u, max_iter=max_iter - it, ftol=ftol, gtol=gtol, xtol=xtol, hess_dtype=hess_dtype,
                                    chol_dtype=chol_dtype, scaling=scaling)
        it += it2
    return kb.to_q(sc, u), c, it, st, out
