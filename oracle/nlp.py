"""ORACLE (test infrastructure, CPU, numpy/scipy): the reference's direct-collocation NLP in its own parameterisation --
node values (x, y, psi, phi, v)(t_i), backward-Euler collocation equalities, end conditions, hard box bounds -- and the
solver the HIP kernel csrc/nlp_kernels.hip implements: augmented Lagrangian for the equalities, primal-dual log barrier for the
bounds, damped Newton steps on the block-tridiagonal Lagrangian Hessian (solve() below).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.

What is restated, with the reference lines (relative to the reference's repository root):
  * free-vector layout [x(N), y(N), psi(N), phi(N), v(N)]                        src/single_opt_planner.py:35-39
  * 6 end conditions (x, y, psi)(t0) = p0[:3], (t1) = p1[:3]                     src/single_opt_planner.py:46-49
  * box bounds on phi, v and optionally x, y                                     src/single_opt_planner.py:53-57
  * equations of motion, residual form, +wind sign quirk                         src/d2d/opty_utils.py:38-50
      xdot - v cos(psi) + wx,  ydot - v sin(psi) + wy,  psidot - g/v tan(phi)
  * collocation: backward Euler, for i = 1..N-1 eom(s_i, (s_i - s_{i-1})/h, u_i) = 0   (inside opty; established on the
    reference's committed solver outputs)
  * objective: the cost plug-ins' cost / cost_grad (src/d2d/opty_utils.py:55-165).  opty hands IPOPT cost_grad as THE gradient
    and no Hessian; CostObstacle's cost_grad is not the derivative of its cost (kind 1 omits (k/r)^2, :118-131), it is the
    derivative of (r/k)^2 * cost.  A KKT point of the reference's run is therefore a KKT point of the objective with the
    obstacle term scaled by (r/k)^2 -- that is the objective minimised here (`grad_quirk=True`), while the value REPORTED is
    the reference's cost().

The reference's own solver (opty -> cyipopt -> IPOPT) is not in the image and cannot be run: the solve is
pinned by the committed IPOPT output of exp_14 (cost 5.02972817, tests/golden/planner_goldens.npz) and by KKT / feasibility
checks, not node by node (IPOPT stopped at tol 1e-5 and the minimiser is not unique: the cost sees only v).
"""
import numpy as np
from scipy.linalg import solveh_banded

G_ACC = 9.81
OBS_K = 2.0
NV = 5                      # variables per node: x, y, psi, phi, v
# augmented Lagrangian / LM constants (include/d2d.h D2D_NLP_*)
RHO0, RHO_MAX, RHO_GROW = 10.0, 1e8, 10.0
LAM0, LAM_MIN, LAM_MAX = 1e-3, 1e-12, 1e12
FEAS_TOL, OPT_TOL = 1e-9, 1e-7
INNER_MAX, OUTER_MAX = 20, 120      # include/d2d.h D2D_NLP_INNER_MAX / D2D_NLP_OUTER_MAX
BANKMAX_BATCHES = 3                 # D2D_NLP_BANKMAX_BATCHES
MUB0, MUB_MIN = 1e-1, 1e-9   # barrier parameter: start, floor
GATE_PROGRESS = 1e-9      # relative decrease of the merit function over a batch of inner_max steps below which an unsolved inner problem is left
STALL_OUTERS = 5          # solved inner problems in a row (each with a tenfold penalty) that did not halve the violation: give up
GRAD_FLOOR = 1e-11        # x rho: rounding floor of the penalty gradient (|x|/h ~ 1e3 at fp64, times rho)


class Problem:
    """One aircraft: scenario data in plain attributes (what the device row carries)."""

    def __init__(self, N, h, p0, p1, vsp=12.0, kv=1.0, kphi=0.0, obj_scale=1.0, wind=(0.0, 0.0), phi_max=np.deg2rad(30.0),
                 v_min=9.0, v_max=14.0, x_box=None, y_box=None, obstacles=(), kobs=0.0, obs_kind=1, partner=None, kcol=0.0, rcol=1.0,
                 bank_max=False):
        self.N, self.h = int(N), float(h)
        # CostBank(use_mean=False), src/d2d/opty_utils.py:68-82: cost = obj_scale * kbank * max_i phi_i^2 and cost_grad one-hot at
        # the maximiser (2 obj_scale kbank phi_imax).  That is the gradient of no differentiable function; what is minimised is the
        # max itself with the maximiser frozen for the length of a Newton step (model term on that node alone, true max in the merit)
        self.bank_max = bool(bank_max)
        self.p0, self.p1 = np.asarray(p0, float)[:3], np.asarray(p1, float)[:3]
        self.vsp, self.kv, self.kphi = float(vsp), float(kv), float(kphi)
        self.s = obj_scale / N
        self.wind = (float(wind[0]), float(wind[1]))
        self.obstacles, self.kobs, self.obs_kind = tuple(obstacles), float(kobs), int(obs_kind)
        self.partner, self.kcol, self.rcol = partner, float(kcol), float(rcol)       # partner: (N, 2) frozen positions or None
        lo = np.full((N, NV), -np.inf); hi = np.full((N, NV), np.inf)
        lo[:, 3], hi[:, 3] = -phi_max, phi_max
        lo[:, 4], hi[:, 4] = v_min, v_max
        if x_box is not None:
            lo[:, 0], hi[:, 0] = x_box
        if y_box is not None:
            lo[:, 1], hi[:, 1] = y_box
        lo[0, :3] = hi[0, :3] = self.p0                 # end conditions: fixed variables
        lo[-1, :3] = hi[-1, :3] = self.p1
        self.lo, self.hi = lo, hi


def problem_from_row(row, N, h):
    """The Problem a d2dhip scenario row describes (what d2d_nlp_solve reads from it: csrc/nlp_kernels.hip nlp_load_scen)."""
    import d2dhip as D
    r = np.asarray(row, float)
    obs = []
    for i in range(D.MAX_OBS):
        c = D.SC_O0X + 3 * i if i < 2 else D.SC_OEXT + 3 * (i - 2)
        if r[c + 2] > 0.0:
            obs.append((r[c], r[c + 1], r[c + 2]))
    okind = int(r[D.SC_OKIND])
    kinds = {(okind >> i) & 1 for i in range(len(obs))}
    assert len(kinds) <= 1, 'mixed obstacle kinds: not expressible as one oracle Problem'
    pb = Problem(N, h, r[D.SC_X0:D.SC_X0 + 3], r[D.SC_X1:D.SC_X1 + 3], vsp=r[D.SC_VSP], kv=r[D.SC_KV], kphi=r[D.SC_KPHI],
                 obj_scale=r[D.SC_S] * N, wind=(-r[D.SC_WX], -r[D.SC_WY]), phi_max=r[D.SC_PHIMAX], v_min=r[D.SC_VMIN], v_max=r[D.SC_VMAX],
                 x_box=(r[D.SC_XMIN], r[D.SC_XMAX]) if r[D.SC_XMIN] < r[D.SC_XMAX] else None,
                 y_box=(r[D.SC_YMIN], r[D.SC_YMAX]) if r[D.SC_YMIN] < r[D.SC_YMAX] else None,
                 obstacles=obs, kobs=r[D.SC_KOBS], obs_kind=0 if (kinds and 1 in kinds) else 1, bank_max=r[D.SC_BANKMAX] != 0.0)
    if r[D.SC_KCOL] > 0.0 and r[D.SC_RCOL] > 0.0:        # CostCollision: scale SCOL * KCOL (no 1 / n_ac, src/d2d/multiopty_utils.py:132); set pb.partner
        pb.kcol, pb.rcol = r[D.SC_KCOL] * r[D.SC_SCOL] / r[D.SC_S], r[D.SC_RCOL]
    return pb


def from_free(free, N):
    """Reference free vector [x, y, psi, phi, v] blocks -> W (N, 5)."""
    return np.asarray(free, float).reshape(NV, N).T.copy()


def to_free(W):
    return np.ascontiguousarray(W.T).reshape(-1)


def constraints(pb, W):
    """(N-1, 3): backward-Euler collocation residuals in the reference's form (divided by h)."""
    x, y, psi, phi, v = W.T
    h = pb.h
    c1 = (x[1:] - x[:-1]) / h - v[1:] * np.cos(psi[1:]) + pb.wind[0]
    c2 = (y[1:] - y[:-1]) / h - v[1:] * np.sin(psi[1:]) + pb.wind[1]
    c3 = (psi[1:] - psi[:-1]) / h - G_ACC / v[1:] * np.tan(phi[1:])
    return np.stack([c1, c2, c3], 1)


LOG_CLIP = float(np.log(1e3))


def _obst_terms(pb, W, quirk):
    """Per node: list of (weight, e, dx, dy, k2, f) of the position-dependent exp terms: cost = sum w*e, cost_grad = -2 k2 w dx e,
    and f = the function of the node position whose gradient that is (f = e except on the clip of a kind-0 obstacle)."""
    out = []
    x, y = W[:, 0], W[:, 1]
    for (cx, cy, r) in pb.obstacles:
        dx, dy = x - cx, y - cy
        if pb.obs_kind == 0:           # e = clip(exp(r^2 - d^2), 0, 1e3) in cost AND in cost_grad = -2 dx e (src/d2d/opty_utils.py:108-131):
            arg = r * r - (dx * dx + dy * dy)      # on the clip that is the gradient of 1e3 (1 + arg - log 1e3), not of the constant cost
            e = np.exp(np.minimum(arg, LOG_CLIP))
            f = np.where(arg > LOG_CLIP, 1e3 * (1.0 + arg - LOG_CLIP), e)
            out.append((pb.s * pb.kobs, e, dx, dy, 1.0, f))
        else:                          # e = exp(-((dx k/r)^2 + (dy k/r)^2)); cost_grad = -2 s dx e (no (k/r)^2)
            k2 = (OBS_K / r) ** 2
            e = np.exp(-(dx * dx + dy * dy) * k2)
            out.append((pb.s * pb.kobs * ((1.0 / k2) if quirk else 1.0), e, dx, dy, k2, e))
    if pb.partner is not None and pb.kcol > 0.0:      # CostCollision against a frozen partner (src/d2d/multiopty_utils.py:120-153)
        dx, dy = x - pb.partner[:, 0], y - pb.partner[:, 1]
        k2 = (OBS_K / pb.rcol) ** 2
        e = np.exp(-(dx * dx + dy * dy) * k2)
        out.append((pb.s * pb.kcol * ((1.0 / k2) if quirk else 1.0), e, dx, dy, k2, e))
    return out


def _bank_value(pb, W):
    """the bank term of the cost: s kphi sum phi^2 (mean mode) or obj_scale kphi max phi^2 (max mode; pb.s = obj_scale / N)"""
    if pb.bank_max:
        return pb.s * pb.N * pb.kphi * float(np.max(W[:, 3] ** 2))
    return pb.s * pb.kphi * float(np.sum(W[:, 3] ** 2))


def cost(pb, W):
    """The reference's cost() value (CostInput / CostAirVel / CostBank / CostComposit), src/d2d/opty_utils.py:55-165."""
    c = pb.s * pb.kv * np.sum((W[:, 4] - pb.vsp) ** 2) + _bank_value(pb, W)
    for w, e, *_ in _obst_terms(pb, W, quirk=False):
        c += w * np.sum(e)
    return float(c)


def objective(pb, W):
    """The objective whose gradient is the reference's cost_grad (kind-1 obstacle terms scaled by (r/k)^2; kind-0 terms continued
    over their clip by the paraboloid whose gradient cost_grad returns there -- see the header and _obst_terms)."""
    c = pb.s * pb.kv * np.sum((W[:, 4] - pb.vsp) ** 2) + _bank_value(pb, W)
    for w, e, dx, dy, k2, f in _obst_terms(pb, W, quirk=True):
        c += w * np.sum(f)
    return float(c)


def cost_grad(pb, W):
    """(N, 5): the reference's cost_grad (= gradient of objective())."""
    g = np.zeros_like(W)
    g[:, 4] = 2 * pb.s * pb.kv * (W[:, 4] - pb.vsp)
    if pb.bank_max:                                   # one-hot at the maximiser (first on ties), :77-82
        im = int(np.argmax(np.abs(W[:, 3])))
        g[im, 3] = 2 * pb.s * pb.N * pb.kphi * W[im, 3]
    else:
        g[:, 3] = 2 * pb.s * pb.kphi * W[:, 3]
    for w, e, dx, dy, k2, _f in _obst_terms(pb, W, quirk=True):
        g[:, 0] += -2 * w * k2 * dx * e
        g[:, 1] += -2 * w * k2 * dy * e
    return g


# ----------------------------------------------------------------------------------
# least-squares form of the augmented Lagrangian:  F = sum r_cost^2 + rho * sum (c + mu)^2
# ----------------------------------------------------------------------------------
def _al_value(pb, W, mu, rho):
    c = constraints(pb, W)
    return objective(pb, W) + rho * float(np.sum((c + mu) ** 2))


def _normal_equations(pb, W, mu, rho, second_order=True):
    """Gradient g (N,5) of F/2... (of F, halved: g = J^T r) and the Gauss-Newton matrix J^T J as block tridiagonal
    D (N,5,5), E (N-1,5,5) with E[i] = block (i+1, i)."""
    N, h = pb.N, pb.h
    x, y, psi, phi, v = W.T
    D = np.zeros((N, NV, NV)); E = np.zeros((N - 1, NV, NV)); g = np.zeros((N, NV))
    # cost rows r = sqrt(s kv)(v - vsp), sqrt(s kphi) phi
    D[:, 4, 4] += pb.s * pb.kv; g[:, 4] += pb.s * pb.kv * (v - pb.vsp)
    if pb.bank_max:                                   # the maximiser of this iterate carries the whole term
        im = int(np.argmax(np.abs(phi)))
        sb = pb.s * pb.N * pb.kphi
        D[im, 3, 3] += sb; g[im, 3] += sb * phi[im]
    else:
        D[:, 3, 3] += pb.s * pb.kphi; g[:, 3] += pb.s * pb.kphi * phi
    # exp rows r = sqrt(w e):  dr = -k2 (dx, dy) r   ->  J^T r = -k2 w e (dx, dy),  J^T J = k2^2 w e (dx, dy)(dx, dy)^T
    for w, e, dx, dy, k2, _f in _obst_terms(pb, W, quirk=True):
        we = w * e
        g[:, 0] += -k2 * we * dx; g[:, 1] += -k2 * we * dy
        D[:, 0, 0] += k2 * k2 * we * dx * dx; D[:, 0, 1] += k2 * k2 * we * dx * dy
        D[:, 1, 0] += k2 * k2 * we * dx * dy; D[:, 1, 1] += k2 * k2 * we * dy * dy
    # constraint rows sqrt(rho) (c_i + mu_i), i = 1..N-1: Jacobian A (wrt node i) and -I/h on (x, y, psi) of node i-1
    c = constraints(pb, W) + mu
    sp, cp = np.sin(psi[1:]), np.cos(psi[1:])
    tp = np.tan(phi[1:]); vi = v[1:]
    A = np.zeros((N - 1, 3, NV))
    A[:, 0, 0] = 1 / h; A[:, 0, 2] = vi * sp; A[:, 0, 4] = -cp
    A[:, 1, 1] = 1 / h; A[:, 1, 2] = -vi * cp; A[:, 1, 4] = -sp
    A[:, 2, 2] = 1 / h; A[:, 2, 3] = -G_ACC * (1 + tp * tp) / vi; A[:, 2, 4] = G_ACC * tp / (vi * vi)
    D[1:] += rho * np.einsum('nki,nkj->nij', A, A)
    g[1:] += rho * np.einsum('nki,nk->ni', A, c)
    idx = np.arange(3)
    D[:-1, idx, idx] += rho / (h * h)
    g[:-1, :3] += -rho * c / h
    E[:, :, :3] += -rho * A.transpose(0, 2, 1) / h           # block (i, i-1) = A_i^T (-I/h) on the (x, y, psi) columns
    if second_order:
        # + sum_k rho (c_k + mu_k) Hessian(c_k): the constraint curvature of the Lagrangian (multiplier estimate 2 rho (c + mu)).
        # The cost's own curvature is tiny (2 s kv on v), so without this term the model is Gauss-Newton on the penalty only and
        # the inner iteration crawls along the curved feasible set.  c_k of node i is nonlinear in (psi, phi, v)_i only.
        m = rho * c
        sec2 = 1 + tp * tp
        D[1:, 2, 2] += m[:, 0] * vi * cp + m[:, 1] * vi * sp
        D[1:, 2, 4] += m[:, 0] * sp - m[:, 1] * cp; D[1:, 4, 2] += m[:, 0] * sp - m[:, 1] * cp
        D[1:, 3, 3] += -m[:, 2] * 2 * G_ACC * tp * sec2 / vi
        D[1:, 3, 4] += m[:, 2] * G_ACC * sec2 / (vi * vi); D[1:, 4, 3] += m[:, 2] * G_ACC * sec2 / (vi * vi)
        D[1:, 4, 4] += -m[:, 2] * 2 * G_ACC * tp / (vi ** 3)
    return g, D, E


def _solve_block_tridiag(D, E, rhs, free, lam):
    """(H + lam diag(H)) delta = rhs on the free variables (delta = 0 elsewhere) by banded Cholesky."""
    N = D.shape[0]
    n = N * NV
    ab = np.zeros((2 * NV, n))                      # lower band storage, bandwidth 2*NV - 1
    f = free.reshape(-1)
    for a in range(NV):
        for b in range(NV):
            col = np.arange(N) * NV + b
            row = np.arange(N) * NV + a
            v = D[:, a, b].copy()
            if a == b:
                v = np.where(free[:, a], v + lam * np.maximum(np.abs(v), 1e-12), 1.0)
            else:
                v = np.where(free[:, a] & free[:, b], v, 0.0)
            if a >= b:
                ab[a - b, col] = v
            # E[i] = block (i+1, i): entry (row (i+1)*NV + a, col i*NV + b)
            ve = np.where(free[1:, a] & free[:-1, b], E[:, a, b], 0.0)
            ab[NV + a - b, np.arange(N - 1) * NV + b] = ve
    r = np.where(f, rhs.reshape(-1), 0.0)
    return solveh_banded(ab, r, lower=True).reshape(N, NV)


def _projected_gradient(pb, W, g):
    """g with the components that push a variable out of its box removed."""
    pg = g.copy()
    pg[(W <= pb.lo) & (g > 0)] = 0.0
    pg[(W >= pb.hi) & (g < 0)] = 0.0
    return pg


def _barrier_sets(pb):
    """Masks: fixed variables (lo == hi: the end conditions), variables with a finite lower / upper bound."""
    fixed = pb.lo == pb.hi
    return fixed, np.isfinite(pb.lo) & ~fixed, np.isfinite(pb.hi) & ~fixed


def _merit(pb, W, mu, rho, mub, hasL, hasU):
    sl = np.where(hasL, W - pb.lo, 1.0); su = np.where(hasU, pb.hi - W, 1.0)
    if (sl <= 0).any() or (su <= 0).any():
        return np.inf
    return _al_value(pb, W, mu, rho) - mub * float(np.sum(np.log(sl)) + np.sum(np.log(su)))


BANKMAX_VALUE_TOL = 1e-7     # include/d2d.h D2D_NLP_BANKMAX_VALUE_TOL


def solve(pb, W0, verbose=False, rho0=RHO0, inner_max=INNER_MAX, outer_max=OUTER_MAX, feas_tol=FEAS_TOL, opt_tol=OPT_TOL):
    """Equalities by an augmented Lagrangian (scaled multiplier estimate mu, penalty rho), bounds by a primal-dual
    log barrier (parameter mub, duals zL / zU), both driven by ONE outer loop; the inner problem is solved by damped
    Newton steps on the block-tridiagonal system (Lagrangian Hessian + barrier diagonal), fraction-to-the-boundary
    rule and a backtracking line search on the barrier-AL merit function.
    Returns W, info (cost, feas, outer, inner, status, mult)."""
    fixed, hasL, hasU = _barrier_sets(pb)
    free = ~fixed
    W = np.asarray(W0, float).copy()
    W[fixed] = pb.lo[fixed]
    # push the start strictly inside the box
    width = np.where(hasL & hasU, pb.hi - pb.lo, np.inf)
    kap = np.minimum(1e-2 * np.maximum(1.0, np.abs(W)), 1e-2 * width)
    W = np.where(hasL, np.maximum(W, pb.lo + kap), W)
    W = np.where(hasU, np.minimum(W, pb.hi - kap), W)
    mu = np.zeros((pb.N - 1, 3)); rho = rho0
    mub = MUB0
    zL = np.where(hasL, mub / np.where(hasL, W - pb.lo, 1.0), 0.0)
    zU = np.where(hasU, mub / np.where(hasU, pb.hi - W, 1.0), 0.0)
    lam = LAM0
    feas_prev = np.inf
    total_inner = 0
    status = 2
    n_stalled = 0
    if pb.bank_max:                 # the value test of the max mode needs a window of its own length (include/d2d.h D2D_NLP_BANKMAX_BATCHES)
        inner_max, outer_max = BANKMAX_BATCHES * inner_max, (outer_max + BANKMAX_BATCHES - 1) // BANKMAX_BATCHES
    for outer in range(1, outer_max + 1):
        tol_in = max(opt_tol, min(1e-1, 10.0 * mub), GRAD_FLOOR * rho)
        phi_first = phi_last = None
        for it in range(inner_max):
            total_inner += 1
            sl = np.where(hasL, W - pb.lo, 1.0); su = np.where(hasU, pb.hi - W, 1.0)
            g, D, E = _normal_equations(pb, W, mu, rho)          # half gradient / half Hessian of the AL function
            # barrier KKT error of the inner problem: stationarity with the duals + complementarity
            stat = np.where(free, 2.0 * g - zL + zU, 0.0)
            comp = max(float(np.abs(np.where(hasL, zL * sl - mub, 0.0)).max()), float(np.abs(np.where(hasU, zU * su - mub, 0.0)).max()))
            err = max(float(np.abs(stat).max()), comp)
            if err <= tol_in:
                break
            # primal-dual Newton step: (H + Sigma) dw = -(grad - mub / sl + mub / su)
            sig = np.where(hasL, zL / sl, 0.0) + np.where(hasU, zU / su, 0.0)
            rhs = -(2.0 * g - np.where(hasL, mub / sl, 0.0) + np.where(hasU, mub / su, 0.0))
            Dh = D.copy()
            idx = np.arange(NV)
            Dh[:, idx, idx] += 0.5 * sig                        # (half convention: D holds H / 2)
            phi0 = _merit(pb, W, mu, rho, mub, hasL, hasU)
            if phi_first is None:
                phi_first = phi_last = phi0
            accepted = False
            for _ in range(30):
                try:
                    dw = _solve_block_tridiag(Dh, E, 0.5 * rhs, free, lam)
                except np.linalg.LinAlgError:
                    lam = min(lam * 8.0, LAM_MAX); continue
                dphi = -float(np.sum(rhs * dw))                  # directional derivative of the merit function (< 0: descent)
                if not dphi < 0.0:
                    lam = min(lam * 8.0, LAM_MAX); continue
                # fraction to the boundary
                tau = max(0.99, 1.0 - mub)
                with np.errstate(divide='ignore', invalid='ignore'):
                    aL = np.where(hasL & (dw < 0), -tau * sl / dw, np.inf)
                    aU = np.where(hasU & (dw > 0), tau * su / dw, np.inf)
                amax = min(1.0, float(aL.min()), float(aU.min()))
                a = amax
                ok = False
                for _ls in range(8):
                    Wt = W + a * dw
                    pt = _merit(pb, Wt, mu, rho, mub, hasL, hasU)
                    if np.isfinite(pt) and pt <= phi0 + 1e-4 * a * dphi:
                        ok = True
                        break
                    a *= 0.5
                if ok:
                    dzL = np.where(hasL, mub / sl - zL - zL / sl * dw, 0.0)
                    dzU = np.where(hasU, mub / su - zU + zU / su * dw, 0.0)
                    with np.errstate(divide='ignore', invalid='ignore'):
                        azL = np.where(hasL & (dzL < 0), -tau * zL / dzL, np.inf)
                        azU = np.where(hasU & (dzU < 0), -tau * zU / dzU, np.inf)
                    az = min(1.0, float(azL.min()), float(azU.min()))
                    W = Wt
                    phi_last = pt
                    zL = zL + az * dzL; zU = zU + az * dzU
                    # keep the duals in a neighbourhood of the central path (IPOPT's kappa_sigma safeguard)
                    slp = np.where(hasL, W - pb.lo, 1.0); sup = np.where(hasU, pb.hi - W, 1.0)
                    zL = np.where(hasL, np.clip(zL, mub / (1e10 * slp), 1e10 * mub / slp), 0.0)
                    zU = np.where(hasU, np.clip(zU, mub / (1e10 * sup), 1e10 * mub / sup), 0.0)
                    lam = max(lam / 3.0, LAM_MIN) if a == amax else lam
                    accepted = True
                    break
                lam = min(lam * 4.0, LAM_MAX)
            if not accepted:
                break
        c = constraints(pb, W)
        feas = float(np.abs(c).max())
        if verbose:
            print(f'outer {outer:2d} rho {rho:.1e} mub {mub:.1e} inner {it + 1:3d} cost {cost(pb, W):.10f} feas {feas:.2e} err {err:.2e} lam {lam:.1e}')
        if feas <= feas_tol and mub <= MUB_MIN * 1.0001 and err <= tol_in:
            status = 1
            break
        # CostBank max mode: the one-hot cost_grad has no zero where two nodes share the maximum (they do at a min-max optimum), so
        # the KKT error never meets its tolerance -- the solve ends "converged in value": feasible, barrier at its floor, and a whole
        # batch of steps that no longer lowers the merit function by more than BANKMAX_VALUE_TOL of itself
        if pb.bank_max and feas <= feas_tol and mub <= MUB_MIN * 1.0001 and phi_first is not None \
                and (phi_first - phi_last) <= BANKMAX_VALUE_TOL * (1.0 + abs(phi_last)):
            status = 1
            break
        # the inner problem is not solved yet and the batch still lowered the merit function by more than rounding: same
        # multipliers, penalty and barrier parameter, another batch of steps -- the schedule must not run ahead of the iterate
        if err > tol_in and accepted and (phi_first - phi_last) > (BANKMAX_VALUE_TOL if pb.bank_max else GATE_PROGRESS) * (1.0 + abs(phi_last)):
            continue
        # an infeasible problem (or an infeasible stationary point of the violation): the penalty grows tenfold per solved inner problem
        # and the violation no longer shrinks: give up (status 4)
        n_stalled = n_stalled + 1 if (feas > 0.5 * feas_prev and feas > 1e3 * feas_tol) else 0
        if n_stalled >= (3 if rho >= RHO_MAX else STALL_OUTERS):
            status = 4
            break
        mu = mu + c                                          # first-order multiplier update (lambda = 2 rho mu)
        if feas > 0.25 * feas_prev and rho < RHO_MAX:
            mu = mu / RHO_GROW; rho *= RHO_GROW              # (mu is the multiplier divided by 2 rho: rescale with rho)
        feas_prev = feas
        mub = max(MUB_MIN, min(0.2 * mub, mub ** 1.5))
    return W, dict(cost=cost(pb, W), feas=float(np.abs(constraints(pb, W)).max()), outer=outer, inner=total_inner, status=status,
                   rho=rho, mult=2 * rho * mu, zL=zL, zU=zU)


def _apply(D, E, s):
    """(block tridiagonal H) s."""
    out = np.einsum('nij,nj->ni', D, s)
    out[1:] += np.einsum('nij,nj->ni', E, s[:-1])
    out[:-1] += np.einsum('nji,nj->ni', E, s[1:])
    return out


def kkt_residual(pb, W, mult, zL=None, zU=None):
    """Stationarity of the Lagrangian cost_grad + A^T mult projected on the box, and feasibility -- the check applied to any
    candidate solution (ours or the reference's committed IPOPT output)."""
    N, h = pb.N, pb.h
    x, y, psi, phi, v = W.T
    g = cost_grad(pb, W)
    sp, cp = np.sin(psi[1:]), np.cos(psi[1:]); tp = np.tan(phi[1:]); vi = v[1:]
    m = mult
    g[1:, 0] += m[:, 0] / h; g[:-1, 0] -= m[:, 0] / h
    g[1:, 1] += m[:, 1] / h; g[:-1, 1] -= m[:, 1] / h
    g[1:, 2] += m[:, 2] / h + m[:, 0] * vi * sp - m[:, 1] * vi * cp; g[:-1, 2] -= m[:, 2] / h
    g[1:, 3] += -m[:, 2] * G_ACC * (1 + tp * tp) / vi
    g[1:, 4] += -m[:, 0] * cp - m[:, 1] * sp + m[:, 2] * G_ACC * tp / (vi * vi)
    if zL is not None:
        fixed = pb.lo == pb.hi
        return float(np.abs(np.where(fixed, 0.0, g - zL + zU)).max()), float(np.abs(constraints(pb, W)).max())
    return float(np.abs(_projected_gradient(pb, W, g)).max()), float(np.abs(constraints(pb, W)).max())
