"""ORACLE (test infrastructure only) -- CPU/numpy fp64 restatement of the batched
6-segment polynomial trajectory fit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (drone-sim-python_amd/) never does.

What is restated from the reference (file:line relative to /root/reference):
  * polynomial layout + Horner evaluation .... src/d2d/trajectory.py:41-82  (PolynomialOne)
  * composite segment lookup .................. src/d2d/trajectory.py:190-208 (CompositeTraj)
  * flatness map (x,y,derivs)->(psi,phi,va) ... src/d2d/guidance.py:22-47   (DiffFlatness)
  * node timing ............................... src/d2d/opty_utils.py:8-14   (planner_timing)
  * input cost s*(kv*sum(dv^2)+kphi*sum(phi^2)) src/d2d/opty_utils.py:85-97  (CostInput)
  * obstacle cost kind 1, k=2 ................. src/d2d/opty_utils.py:99-134 (CostObstacle)
  * 'tri' dog-leg waypoints ................... src/d2d/opty_utils.py:171-187 (triangle)
What is build-defined (no reference counterpart): the
re-parameterisation by polynomial coefficients, the elimination of the linear side
conditions, and the Levenberg-Marquardt normal-equation solver.  Parity of those is
pinned by (i) the golden vectors in tests/golden/fit_cost_golden.npz, which were
produced by the reference's own classes evaluated on polynomial trajectories, and
(ii) scipy.optimize.least_squares on the same residual function (the CPU arbiter).
"""
import math
import numpy as np

G_ACC = 9.81          # src/d2d/guidance.py:39
OBS_K = 2.0           # src/d2d/opty_utils.py:103
NCOEF = 8             # PolynomialOne with 4 derivatives -> 8 coefficients
NDER_C = 4            # derivatives matched at junctions (C^3)


# ----------------------------------------------------------------------------------
# timing / segment bookkeeping
# ----------------------------------------------------------------------------------
def planner_timing(t0, t1, hz):
    """src/d2d/opty_utils.py:8-14 (without the print)."""
    duration = t1 - t0
    num_nodes = int(duration * hz) + 1
    time_step = 1.0 / hz
    duration = (num_nodes - 1) * time_step
    return num_nodes, time_step, duration


def arr(k, n):
    """a(k,n) = n!/(n-k)!  -- src/d2d/trajectory.py:41-45."""
    a, i = 1, n
    while i > n - k:
        a *= i
        i -= 1
    return a


def sample_segments(K, S, duration):
    """Sample times linspace(0,duration,K); segment index and local time of each.

    Mirrors CompositeTraj.get (src/d2d/trajectory.py:202-208): cur = argmax(steps_end >
    t), except that the very last sample (t == duration) stays in the last segment
    instead of wrapping through fmod."""
    T = duration / S
    t = np.linspace(0.0, duration, K)
    ends = np.cumsum(np.full(S, T))
    seg = np.empty(K, dtype=np.int64)
    for k in range(K):
        gt = ends > t[k]
        seg[k] = int(np.argmax(gt)) if gt.any() else S - 1
    tau = t - seg * T
    return t, seg, tau, T


def basis_rows(tau, der):
    """Row of d^der/dt^der [1, t, ..., t^7] at local time tau (PolynomialOne layout:
    coefs[d,p] = arr(d,p+d)*coefs[0,p+d], src/d2d/trajectory.py:68-72)."""
    row = np.zeros(NCOEF)
    for p in range(der, NCOEF):
        row[p] = arr(der, p) * tau ** (p - der)
    return row


def sample_matrix(K, S, duration, der):
    """Phi_der (K x 8S): flat output derivative `der` at the K samples = Phi @ z_axis."""
    _, seg, tau, _ = sample_segments(K, S, duration)
    Phi = np.zeros((K, NCOEF * S))
    for k in range(K):
        Phi[k, NCOEF * seg[k]:NCOEF * (seg[k] + 1)] = basis_rows(tau[k], der)
    return Phi


def constraint_matrix(S, T):
    """Linear side conditions of one axis, C z = [0..0, pos0, vel0, pos1, vel1].

    Rows: C^3 continuity at the S-1 junctions (4 rows each), then the four end
    conditions.  (x,y,psi)(t0/t1) of src/single_opt_planner.py:46-49 become position
    and velocity = vref*(cos psi, sin psi) end conditions on the flat outputs."""
    n = NCOEF * S
    rows = []
    for j in range(S - 1):
        for d in range(NDER_C):
            r = np.zeros(n)
            r[NCOEF * j:NCOEF * (j + 1)] = basis_rows(T, d)
            r[NCOEF * (j + 1):NCOEF * (j + 2)] -= basis_rows(0.0, d)
            rows.append(r)
    for (s, tt, d) in ((0, 0.0, 0), (0, 0.0, 1), (S - 1, T, 0), (S - 1, T, 1)):
        r = np.zeros(n)
        r[NCOEF * s:NCOEF * (s + 1)] = basis_rows(tt, d)
        rows.append(r)
    return np.array(rows)


def hermite_map(T):
    """8x8 map [Y0(4); Y1(4)] -> coefs[0,0:8] of one segment, exactly the closed form of
    PolynomialOne.__init__ (src/d2d/trajectory.py:54-66): low = Y0[i]/i!, high =
    M4^-1 (Y1 - M3 low)."""
    M1i = np.diag([1.0 / arr(i, i) for i in range(4)])
    M3 = np.zeros((4, 4)); M4 = np.zeros((4, 4))
    for i in range(4):
        for j in range(i, 4):
            M3[i, j] = arr(i, j) * T ** (j - i)
        for j in range(4):
            M4[i, j] = arr(i, j + 4) * T ** (j - i + 4)
    M4i = np.linalg.inv(M4)
    H = np.zeros((8, 8))
    H[0:4, 0:4] = M1i
    H[4:8, 0:4] = -M4i @ M3 @ M1i
    H[4:8, 4:8] = M4i
    return H


def junction_map(S, T):
    """N (8S x 4(S+1)): knot data (pos,vel,acc,jerk at the S+1 knots) -> coefficients of a
    C^3 piecewise degree-7 polynomial.  This is the reference's own construction:
    CompositeTraj([MinSnapPoly(Y_j, Y_{j+1}, T)]) (src/d2d/trajectory.py:166-208)."""
    H = hermite_map(T)
    N = np.zeros((NCOEF * S, 4 * (S + 1)))
    for s in range(S):
        N[NCOEF * s:NCOEF * (s + 1), 4 * s:4 * s + 8] = H
    return N


def knot_split(S):
    """Column indices of N: fixed = (pos,vel) at the first and last knot (the end
    conditions (x,y,psi)(t0),(t1) of src/single_opt_planner.py:46-49 with speed vref);
    free = everything else, in knot order."""
    fixed = [0, 1, 4 * S, 4 * S + 1]
    free = [i for i in range(4 * (S + 1)) if i not in fixed]
    return fixed, free


class FitBasis:
    """Shared (batch-independent) block: z_axis = Zp @ d_axis + Z @ q_axis.

    d_axis = [pos0, vel0, pos1, vel1]; q_axis = 4S whitened knot coordinates.
    G[d]  (K x nq) = Phi_d @ Z     -- flat-output derivative d as a function of q
    Gp[d] (K x 4)  = Phi_d @ Zp    -- contribution of the end conditions
    Z = Nfree L^-T with L L^T = Nfree^T Mref Nfree (so Z^T Mref Z = I),
    Mref = sum_d w_d Phi_d^T Phi_d;  Zp = (I - Z Z^T Mref) Nfixed (minimum-Mref-energy
    particular solution).  Pinit = (G0^T G0)^-1 G0^T projects waypoints on the basis.
    The product builds the same block in C++ (csrc/fit_basis.cpp); GPU parity tests
    feed the product's own arrays through FitBasis.from_arrays."""

    def __init__(self, S, K, duration, wref):
        self.S, self.K, self.duration = S, K, duration
        self.T = duration / S
        Phi = [sample_matrix(K, S, duration, d) for d in range(3)]
        N = junction_map(S, self.T)
        fixed, free = knot_split(S)
        Nf, Nx = N[:, free], N[:, fixed]
        self.nq = len(free)
        M = sum(w * P.T @ P for w, P in zip(wref, Phi))
        L = np.linalg.cholesky(Nf.T @ M @ Nf)
        self.Z = np.linalg.solve(L, Nf.T).T                  # Nf @ L^-T
        self.Zp = Nx - self.Z @ (self.Z.T @ (M @ Nx))
        self.G = np.stack([P @ self.Z for P in Phi])         # (3,K,nq)
        self.Gp = np.stack([P @ self.Zp for P in Phi])       # (3,K,4)
        G0 = self.G[0]
        self.Pinit = np.linalg.solve(G0.T @ G0, G0.T)        # (nq,K)

    @classmethod
    def from_arrays(cls, S, K, duration, G, Gp, Z, Zp, Pinit):
        b = cls.__new__(cls)
        b.S, b.K, b.duration, b.T = S, K, duration, duration / S
        b.G, b.Gp, b.Z, b.Zp, b.Pinit = (np.asarray(a, dtype=np.float64) for a in (G, Gp, Z, Zp, Pinit))
        b.nq = b.Z.shape[1]
        return b


# ----------------------------------------------------------------------------------
# scenario parameters of one trajectory
# ----------------------------------------------------------------------------------
# scen row layout (float64[SCEN_STRIDE]); identical to include/d2d.h D2D_SCEN_*
SCEN_STRIDE = 80
MAX_OBS = 16
(SC_X0, SC_Y0, SC_PSI0, SC_X1, SC_Y1, SC_PSI1, SC_VREF, SC_VSP, SC_KV, SC_KPHI,
 SC_KOBS, SC_S, SC_WWP, SC_WX, SC_WY, SC_GOLEFT,
 SC_O0X, SC_O0Y, SC_O0R, SC_O1X, SC_O1Y, SC_O1R, SC_WBND, SC_PHIMAX, SC_VMIN, SC_VMAX,
 SC_KCOL, SC_RCOL, SC_SCOL, SC_PMASK, SC_OKIND, SC_BANKMAX, SC_OEXT) = range(33)
# (x, y, r) columns of obstacle i: the first two sit at SC_O0X.., obstacles 2.. at SC_OEXT + 3*(i-2)
SC_OBS = tuple((c, c + 1, c + 2) for c in [SC_O0X, SC_O1X] + [SC_OEXT + 3 * j for j in range(MAX_OBS - 2)])
SC_XMIN, SC_XMAX, SC_YMIN, SC_YMAX = range(SC_OEXT + 3 * (MAX_OBS - 2), SC_OEXT + 3 * (MAX_OBS - 2) + 4)
# x_constraint / y_constraint boxes (src/single_opt_planner.py:56-57) as soft rows w_b*dist(x,[XMIN,XMAX]), w_b*dist(y,..):
# an axis with MIN >= MAX has none
SC_O2X, SC_O2Y, SC_O2R = SC_OBS[2]
SC_O3X, SC_O3Y, SC_O3R = SC_OBS[3]
# SC_KCOL/RCOL/SCOL: collision weight, radius and scale (obj_scale/N, src/d2d/multiopty_utils.py:132);
# SC_PMASK: bit j set <=> coupled with aircraft j of the same group (CostCollision pairs);
# SC_OKIND: bit i set <=> obstacle i is CostObstacle kind 0 (src/d2d/opty_utils.py:108-111);
# SC_BANKMAX != 0 <=> CostBank(use_mean=False) (:72-73, :80-81)
OBS_CLIP = 1e3                                   # np.clip(errs, 0., 1e3), src/d2d/opty_utils.py:111
# default bounds of the soft bound rows (phi in +-40 deg, v in [9,15]: src/multi_opt_planner.py:192-193)
PHI_MAX = math.radians(40.0)
V_MIN, V_MAX = 9.0, 15.0


def end_data(sc):
    """d_axis = [pos0, vel0, pos1, vel1] for x and y (end speed = vref)."""
    v = sc[SC_VREF]
    dx = np.array([sc[SC_X0], v * math.cos(sc[SC_PSI0]), sc[SC_X1], v * math.cos(sc[SC_PSI1])])
    dy = np.array([sc[SC_Y0], v * math.sin(sc[SC_PSI0]), sc[SC_Y1], v * math.sin(sc[SC_PSI1])])
    return dx, dy


def triangle(p0, p1, va, duration, num_nodes, go_left=1.0):
    """'tri' initial guess, src/d2d/opty_utils.py:171-187 (x,y only)."""
    p0 = np.asarray(p0, float); p1 = np.asarray(p1, float)
    p0p1 = p1 - p0
    d = np.linalg.norm(p0p1)
    u = p0p1 / d
    v = np.array([-u[1], u[0]])
    D = va * duration
    p2 = p0 + p0p1 / 2
    if D > d:
        p2 = p2 + np.sign(go_left) * np.sqrt(D ** 2 - d ** 2) / 2 * v
    n1 = int(num_nodes / 2); n2 = num_nodes - n1
    pts = np.vstack((np.linspace(p0, p2, n1), np.linspace(p2, p1, n2)))
    return pts[:, 0], pts[:, 1]


def waypoints(sc, K, duration):
    return triangle((sc[SC_X0], sc[SC_Y0]), (sc[SC_X1], sc[SC_Y1]), sc[SC_VREF],
                    duration, K, sc[SC_GOLEFT])


# ----------------------------------------------------------------------------------
# residuals / Jacobian in reduced coordinates q = [q_x(24), q_y(24)]
# ----------------------------------------------------------------------------------
NROW = 8   # rows per sample: v, phi, wp_x, wp_y, obs0, obs1, hinge_phi, hinge_v; obstacles 2.. (rare) follow at NROW..,
           # then the collision rows of a coupled group


def n_extra_obs(sc):
    """Number of extra obstacle rows: index of the last present obstacle beyond the first two, + 1 - 2."""
    n = 0
    for i in range(2, MAX_OBS):
        if sc[SC_OBS[i][2]] > 0:
            n = i - 1
    return n


def obs_row(i):
    return 4 + i if i < 2 else NROW + (i - 2)


def has_box(sc):
    return bool(sc[SC_XMIN] < sc[SC_XMAX] or sc[SC_YMIN] < sc[SC_YMAX])


def flat_outputs(basis, sc, q):
    nq = basis.nq
    dx, dy = end_data(sc)
    Y = np.empty((3, 2, basis.K))
    for d in range(3):
        Y[d, 0] = basis.Gp[d] @ dx + basis.G[d] @ q[:nq]
        Y[d, 1] = basis.Gp[d] @ dy + basis.G[d] @ q[nq:]
    return Y


def flatness(Y, sc):
    """src/d2d/guidance.py:22-47: va, psi, phi from flat outputs (stationary wind)."""
    vax = Y[1, 0] - sc[SC_WX]; vay = Y[1, 1] - sc[SC_WY]
    va2 = vax ** 2 + vay ** 2
    va = np.sqrt(va2)
    psi = np.arctan2(vay, vax)
    num = Y[2, 1] * vax - Y[2, 0] * vay
    phi = np.arctan(num / va / G_ACC)
    return va, psi, phi


def residuals(basis, sc, q, wp=None, want_jac=False, others=None):
    """r (K,NROW+n_extra_obs+(2 if box)+n_others) and optionally the partials D (K,rows,6) wrt (x,y,xd,yd,xdd,ydd).

    others (n_others,2,K): sampled positions of the coupled aircraft (held fixed): one extra row
    sqrt(s_col*kcol*e) each, e as CostCollision (src/d2d/multiopty_utils.py:120-153)."""
    K = basis.K
    n_oth = 0 if others is None else len(others)
    Y = flat_outputs(basis, sc, q)
    x, y = Y[0]; a = Y[1, 0] - sc[SC_WX]; b = Y[1, 1] - sc[SC_WY]; c, d = Y[2]
    if wp is None:
        wp = waypoints(sc, K, basis.duration)
    va2 = a * a + b * b; va = np.sqrt(va2)
    n = d * a - c * b
    w = n / (va * G_ACC)
    phi = np.arctan(w)
    s = sc[SC_S]
    cv = math.sqrt(s * sc[SC_KV]); cphi = math.sqrt(s * sc[SC_KPHI]); cobs = math.sqrt(s * sc[SC_KOBS])
    n_x = n_extra_obs(sc)
    BROW = NROW + n_x                            # the two position-box rows (when the scenario has a box)
    CROW = BROW + (2 if has_box(sc) else 0)      # first collision row
    r = np.zeros((K, CROW + n_oth))
    r[:, 0] = cv * (va - sc[SC_VSP])
    wphi = np.full(K, cphi)                      # weight of the phi row of every sample
    if sc[SC_BANKMAX] != 0:                      # CostBank max mode: obj_scale*max(phi^2) = one row, at argmax
        wphi = np.zeros(K)
        wphi[int(np.argmax(np.square(phi)))] = math.sqrt(s * K * sc[SC_KPHI])
    r[:, 1] = wphi * phi
    r[:, 2] = sc[SC_WWP] * (x - wp[0])
    r[:, 3] = sc[SC_WWP] * (y - wp[1])
    obs = []
    for i, (ox, oy, orr) in enumerate(SC_OBS):
        rr = sc[orr]
        if rr > 0:
            if (int(sc[SC_OKIND]) >> i) & 1:     # kind 0: e = clip(exp(r^2 - d^2), 0, 1e3); the row sqrt(s*kobs*e) is flat on the clip
                kk = 1.0
                ddx = x - sc[ox]; ddy = y - sc[oy]
                e = np.exp(rr * rr - (ddx * ddx + ddy * ddy))
                live = (e <= OBS_CLIP).astype(float)
                h = cobs * np.sqrt(np.clip(e, 0.0, OBS_CLIP))
            else:                                # kind 1: e = exp(-|k (p - o) / r|^2)
                kk = OBS_K / rr
                ddx = (x - sc[ox]) * kk; ddy = (y - sc[oy]) * kk
                h = cobs * np.exp(-0.5 * (ddx * ddx + ddy * ddy))     # sqrt(s*kobs*e)
                live = 1.0
            r[:, obs_row(i)] = h
            obs.append((i, ddx, ddy, h * live, kk))
    wb = sc[SC_WBND]
    hphi = np.maximum(np.abs(phi) - sc[SC_PHIMAX], 0.0)
    hv = np.maximum(va - sc[SC_VMAX], 0.0) + np.minimum(va - sc[SC_VMIN], 0.0)
    r[:, 6] = wb * hphi
    r[:, 7] = wb * hv
    if has_box(sc):
        xlo, xhi = (sc[SC_XMIN], sc[SC_XMAX]) if sc[SC_XMIN] < sc[SC_XMAX] else (-np.inf, np.inf)
        ylo, yhi = (sc[SC_YMIN], sc[SC_YMAX]) if sc[SC_YMIN] < sc[SC_YMAX] else (-np.inf, np.inf)
        hx = np.maximum(x - xhi, 0.0) + np.minimum(x - xlo, 0.0)
        hy = np.maximum(y - yhi, 0.0) + np.minimum(y - ylo, 0.0)
        r[:, BROW] = wb * hx
        r[:, BROW + 1] = wb * hy
    col = []
    if n_oth:
        ccol = math.sqrt(sc[SC_SCOL] * sc[SC_KCOL]); kc = OBS_K / sc[SC_RCOL]
        for m in range(n_oth):
            ex = (x - others[m][0]) * kc; ey = (y - others[m][1]) * kc
            h = ccol * np.exp(-0.5 * (ex * ex + ey * ey))
            r[:, CROW + m] = h
            col.append((ex, ey, h, kc))
    if not want_jac:
        return r
    D = np.zeros((K, CROW + n_oth, 6))
    dva_a = a / va; dva_b = b / va
    D[:, 0, 2] = cv * dva_a; D[:, 0, 3] = cv * dva_b
    f = 1.0 / (1.0 + w * w)
    ivg = 1.0 / (va * G_ACC)
    dw_a = d * ivg - n * a / (va2 * va * G_ACC)
    dw_b = -c * ivg - n * b / (va2 * va * G_ACC)
    dw_c = -b * ivg
    dw_d = a * ivg
    dphi = np.stack([dw_a, dw_b, dw_c, dw_d], axis=1) * f[:, None]      # wrt a,b,c,d
    D[:, 1, 2:6] = wphi[:, None] * dphi
    D[:, 2, 0] = sc[SC_WWP]
    D[:, 3, 1] = sc[SC_WWP]
    for (i, ddx, ddy, h, kk) in obs:
        D[:, obs_row(i), 0] = -h * ddx * kk
        D[:, obs_row(i), 1] = -h * ddy * kk
    act = (hphi > 0) * np.sign(phi)
    D[:, 6, 2:6] = wb * act[:, None] * dphi
    actv = ((va > sc[SC_VMAX]) | (va < sc[SC_VMIN])).astype(float)
    D[:, 7, 2] = wb * actv * dva_a; D[:, 7, 3] = wb * actv * dva_b
    if has_box(sc):
        D[:, BROW, 0] = wb * (hx != 0); D[:, BROW + 1, 1] = wb * (hy != 0)
    for m, (ex, ey, h, kc) in enumerate(col):
        D[:, CROW + m, 0] = -h * ex * kc
        D[:, CROW + m, 1] = -h * ey * kc
    return r, D


def jacobian(basis, D):
    """J (K*NROW x 2nq) = D_k . Gk, columns [q_x | q_y]."""
    K, nq = basis.K, basis.nq
    G = basis.G
    nrow = D.shape[1]
    J = np.zeros((K, nrow, 2 * nq))
    for ax in range(2):
        J[:, :, ax * nq:(ax + 1) * nq] = (D[:, :, 0 + ax, None] * G[0][:, None, :]
                                          + D[:, :, 2 + ax, None] * G[1][:, None, :]
                                          + D[:, :, 4 + ax, None] * G[2][:, None, :])
    return J.reshape(K * nrow, 2 * nq)


def cost(basis, sc, q, wp=None, others=None):
    r = residuals(basis, sc, q, wp, others=others)
    return float(np.sum(r * r))


def curvature_blocks(basis, sc, q, wp=None):
    """W (K,6,6) = sum over the rows of sample k of r_i * Hessian_Y(r_i), Y = (x,y,xd,yd,xdd,ydd): the
    second-order part of the Hessian of 0.5*sum r^2 is sum_k G_k^T W_k G_k (the flat outputs are linear in q).
    Rows: speed row and its hinge (curvature of |v_a|), bank row and its hinge (curvature of atan(w)),
    obstacle rows (curvature of the Gaussian bump; none on the clip of kind 0); waypoint rows are linear."""
    K = basis.K
    if wp is None:
        wp = waypoints(sc, K, basis.duration)
    r = residuals(basis, sc, q, wp)
    Y = flat_outputs(basis, sc, q)
    x, y = Y[0]; a = Y[1, 0] - sc[SC_WX]; b = Y[1, 1] - sc[SC_WY]; c, d = Y[2]
    va2 = a * a + b * b; va = np.sqrt(va2)
    n = d * a - c * b
    ivg = 1.0 / (va * G_ACC)
    w = n * ivg
    phi = np.arctan(w)
    f = 1.0 / (1.0 + w * w)
    s = sc[SC_S]
    W = np.zeros((K, 6, 6))
    # |v_a|: Hessian (I - t t^T)/va on (xd, yd); rows 0 (weight cv) and 7 (weight wb where the bound is active)
    cv = math.sqrt(s * sc[SC_KV]); wb = sc[SC_WBND]
    actv = ((va > sc[SC_VMAX]) | (va < sc[SC_VMIN])).astype(float)
    kv = (cv * r[:, 0] + wb * actv * r[:, 7]) / va
    t = np.stack([a, b], 1) / va[:, None]
    Pn = np.eye(2)[None] - t[:, :, None] * t[:, None, :]
    W[:, 2:4, 2:4] += kv[:, None, None] * Pn
    # phi = atan(w): Hessian f*Hess(w) - 2 w f^2 grad(w) grad(w)^T on (a,b,c,d); rows 1 and 6
    gn = np.stack([d, -c, -b, a], 1)                       # grad n
    p4 = np.stack([a, b, np.zeros(K), np.zeros(K)], 1)
    gw = ivg[:, None] * gn - (w / va2)[:, None] * p4       # grad w
    Hn = np.zeros((4, 4)); Hn[0, 3] = Hn[3, 0] = 1.0; Hn[1, 2] = Hn[2, 1] = -1.0
    P2 = np.diag([1.0, 1.0, 0.0, 0.0])
    Hw = (ivg[:, None, None] * Hn[None]
          - (ivg / va2)[:, None, None] * (gn[:, :, None] * p4[:, None, :] + p4[:, :, None] * gn[:, None, :])
          + w[:, None, None] * (3.0 * p4[:, :, None] * p4[:, None, :] / (va2 * va2)[:, None, None] - P2[None] / va2[:, None, None]))
    Hphi = f[:, None, None] * Hw - (2.0 * w * f * f)[:, None, None] * gw[:, :, None] * gw[:, None, :]
    wphi = np.full(K, math.sqrt(s * sc[SC_KPHI]))
    if sc[SC_BANKMAX] != 0:
        wphi = np.zeros(K); wphi[int(np.argmax(np.square(phi)))] = math.sqrt(s * K * sc[SC_KPHI])
    hphi = np.maximum(np.abs(phi) - sc[SC_PHIMAX], 0.0)
    act = (hphi > 0) * np.sign(phi)
    kphi = wphi * r[:, 1] + wb * act * r[:, 6]
    W[:, 2:6, 2:6] += kphi[:, None, None] * Hphi
    # obstacles: h = cobs*exp(0.5*(c0 - |e|^2)), e = kk*(p - o): Hessian h*kk^2*(e e^T - I) on (x, y)
    for i, (ox, oy, orr) in enumerate(SC_OBS):
        rr = sc[orr]
        if rr > 0:
            kind0 = (int(sc[SC_OKIND]) >> i) & 1
            kk = 1.0 if kind0 else OBS_K / rr
            e = np.stack([(x - sc[ox]) * kk, (y - sc[oy]) * kk], 1)
            h = r[:, obs_row(i)]
            live = np.ones(K)
            if kind0:
                live = (np.exp(rr * rr - np.sum(e * e, 1)) <= OBS_CLIP).astype(float)
            W[:, 0:2, 0:2] += (h * h * live * kk * kk)[:, None, None] * (e[:, :, None] * e[:, None, :] - np.eye(2)[None])
    return W


def second_order_term(basis, sc, q, wp=None):
    """S = sum_i r_i Hessian_q(r_i) = sum_k G_k^T W_k G_k (2nq x 2nq)."""
    K, nq = basis.K, basis.nq
    W = curvature_blocks(basis, sc, q, wp)
    Gk = np.zeros((K, 6, 2 * nq))
    for der in range(3):
        Gk[:, 2 * der, :nq] = basis.G[der]
        Gk[:, 2 * der + 1, nq:] = basis.G[der]
    return np.einsum('kia,kij,kjb->ab', Gk, W, Gk)


def eval_normal(basis, sc, q, wp=None, others=None, second_order=False):
    """cost = sum r^2, g = J^T r, H = J^T J (all fp64); second_order adds sum_i r_i Hessian(r_i) to H."""
    r, D = residuals(basis, sc, q, wp, want_jac=True, others=others)
    J = jacobian(basis, D)
    rv = r.reshape(-1)
    H = J.T @ J
    if second_order:
        assert others is None
        H = H + second_order_term(basis, sc, q, wp)
    return float(rv @ rv), J.T @ rv, H


def initial_guess(basis, sc, wp=None):
    """q0 = least-squares projection of the 'tri' waypoints on the reduced basis."""
    if wp is None:
        wp = waypoints(sc, basis.K, basis.duration)
    dx, dy = end_data(sc)
    P = basis.Pinit
    qx = P @ (wp[0] - basis.Gp[0] @ dx)
    qy = P @ (wp[1] - basis.Gp[0] @ dy)
    return np.concatenate([qx, qy])


# ----------------------------------------------------------------------------------
# Levenberg-Marquardt on the normal equations -- the algorithm the HIP path implements
# (csrc/fit_kernels.hip: fit_eval_kernel + fit_step_kernel); constants mirror
# include/d2d.h D2D_LM_*.
# ----------------------------------------------------------------------------------
LM_LAMBDA0 = 1e-3
LM_LAMBDA_MIN, LM_LAMBDA_MAX = 1e-12, 1e12
LM_DIAG_FLOOR = 1e-30
LM_BT_MIN, LM_BT_MAX = 0.1, 0.5      # clip of the parabola's minimiser along a rejected step (include/d2d.h D2D_LM_BT_*)
LM_BT_SHRINK, LM_BT_FLOOR = 0.25, 0.02   # second attempt: a quarter of the first, not below 2 % of the step
LM_FAIL_MULT = 8.0      # damping growth after a failed factorisation (indefinite exact Hessian)
LM_SO_LAMBDA = 1e-4     # below this damping the next evaluation carries the second-order term (include/d2d.h D2D_LM_SO_LAMBDA)
ST_RUNNING, ST_CONVERGED, ST_MAXITER, ST_NONFINITE, ST_STALLED = 0, 1, 2, 3, 4


def lm_solve(basis, sc, q0=None, max_iter=200, ftol=1e-14, gtol=1e-9, xtol=1e-11,
             hess_dtype=np.float64, chol_dtype=np.float64, others=None, so_lambda=None, backtrack=True, lam0=LM_LAMBDA0, fail_floor=0.0, stats=None):
    """(H + lam*diag|H|) delta = -J^T r with Nielsen's gain-ratio damping, H = J^T J (Gauss-Newton) or, once the
    damping has fallen to so_lambda, J^T J + sum_i r_i Hessian(r_i) (the exact Hessian of 0.5*sum r^2).

    One "iteration" = one damped solve + one trial cost; H is re-evaluated only after an accepted step, in the
    mode decided by the damping BEFORE that step (the kernel computes the rows of the trial point
    speculatively, before it knows the gain ratio).  A step whose gain ratio is not positive is not thrown away:
    the cost along it is known at 0 (value and slope) and at 1, so the minimiser of the parabola through those,
    alpha = a / (2 (c1 - c0 + a)) clipped to [LM_BT_MIN, LM_BT_MAX], is tried (and LM_BT_SHRINK of it if that fails
    too); an accepted shortened step divides the damping by... multiplies it by 1/alpha (the step of a damped system
    shrinks like 1/lam), instead of re-solving with a doubled, quadrupled, ... damping.  Only when both shortened steps
    fail does the damping grow by Nielsen's nu; a factorisation that fails (indefinite exact Hessian) multiplies it by
    LM_FAIL_MULT.  so_lambda: None = LM_SO_LAMBDA for single trajectories and off for coupled groups; 0 = Gauss-Newton
    only.  hess_dtype / chol_dtype = np.float32 mimic the HIP path's fp32 MFMA Hessian and fp32 Cholesky (residuals,
    cost and J^T r stay fp64).  backtrack=False: the plain reject-and-grow rule (round 1's algorithm).
    Returns q, cost, iters, status."""
    if so_lambda is None:
        so_lambda = LM_SO_LAMBDA if others is None else 0.0
    wp = waypoints(sc, basis.K, basis.duration)
    q = initial_guess(basis, sc, wp) if q0 is None else np.array(q0, float)
    lam, nu = lam0, 2.0
    lam_fail = 0.0
    status = ST_MAXITER
    c, g, H = eval_normal(basis, sc, q, wp, others, second_order=(so_lambda > 0 and lam0 <= so_lambda))
    H = H.astype(hess_dtype).astype(np.float64)
    it = 0
    if not np.isfinite(c):
        return q, c, 0, ST_NONFINITE
    for it in range(1, max_iter + 1):
        if np.max(np.abs(g)) <= gtol:
            status = ST_CONVERGED
            break
        so_next = so_lambda > 0 and lam <= so_lambda
        dg = np.maximum(np.abs(np.diag(H)), LM_DIAG_FLOOR)
        A = (H + lam * np.diag(dg)).astype(chol_dtype)
        ok = True
        try:
            L = np.linalg.cholesky(A)
            delta = -np.linalg.solve(L.T, np.linalg.solve(L, g.astype(chol_dtype))).astype(np.float64)
        except np.linalg.LinAlgError:
            ok = False
        rho, fin, accept = -1.0, False, False
        ct, pred = np.inf, 0.0
        if ok:
            ct = cost(basis, sc, q + delta, wp, others)
            pred = float(delta @ (lam * dg * delta - g))
            fin = bool(np.isfinite(ct) and pred > 0)
            if fin:
                rho = (c - ct) / pred
        step, pred_s = None, pred
        if rho > 0:
            accept, step = True, delta
            lam_new = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), LM_LAMBDA_MIN)
        elif fin and backtrack:
            # cost along the step: c at 0 with slope -a, ct at 1  ->  parabola, its minimiser alpha
            a = -2.0 * float(g @ delta)
            b = a - pred                                     # delta^T H delta (the model's curvature along the step)
            den = 2.0 * (ct - c + a)
            al = a / den if den > 0.0 else LM_BT_MAX
            al = min(max(al, LM_BT_MIN), LM_BT_MAX)
            for attempt in range(2):
                c2 = cost(basis, sc, q + al * delta, wp, others)
                if np.isfinite(c2) and c2 < c:
                    accept, step, ct = True, al * delta, c2
                    pred_s = a * al - b * al * al
                    lam_new = min(lam / al, LM_LAMBDA_MAX)
                    break
                al = max(LM_BT_SHRINK * al, LM_BT_FLOOR)
        if accept:
            small_x = np.max(np.abs(step)) <= xtol * (np.max(np.abs(q)) + xtol)
            q = q + step
            lam, nu = lam_new, 2.0
            if fail_floor > 0.0:
                lam = max(lam, fail_floor * lam_fail)
                lam_fail *= 0.5
            small_f = (c - ct) <= ftol * c and pred_s <= ftol * c
            c, g, H = eval_normal(basis, sc, q, wp, others, second_order=so_next)
            H = H.astype(hess_dtype).astype(np.float64)
            if small_f or small_x:
                status = ST_CONVERGED
                break
        else:
            if ok and fin and pred <= ftol * c:     # rejected on the rounding floor of the cost: converged
                status = ST_CONVERGED
                break
            if ok or not backtrack:
                lam *= nu; nu *= 2.0
            else:
                lam_fail = max(lam_fail, lam)
                if stats is not None:
                    stats['fails'] = stats.get('fails', 0) + 1
                lam *= LM_FAIL_MULT
            if lam > LM_LAMBDA_MAX:
                status = ST_STALLED
                break
    return q, c, it, status


# ----------------------------------------------------------------------------------
# groups of coupled aircraft (collision rows between the aircraft of one scenario)
# ----------------------------------------------------------------------------------
def partners(sc, self_idx, n_ac):
    m = int(sc[SC_PMASK])
    return [j for j in range(n_ac) if j != self_idx and (m >> j) & 1]


def group_positions(basis, scs, qs):
    """(n_ac, 2, K) sampled x, y of every aircraft of the group."""
    return np.array([flat_outputs(basis, scs[i], qs[i])[0] for i in range(len(scs))])


def group_cost(basis, scs, qs):
    """Joint cost: own rows of every aircraft + each coupled pair ONCE."""
    n = len(scs)
    pos = group_positions(basis, scs, qs)
    tot = sum(cost(basis, scs[i], qs[i]) for i in range(n))
    for i in range(n):
        for j in partners(scs[i], i, n):
            if j > i:
                kc = OBS_K / scs[i][SC_RCOL]
                ex = (pos[i][0] - pos[j][0]) * kc; ey = (pos[i][1] - pos[j][1]) * kc
                tot += scs[i][SC_SCOL] * scs[i][SC_KCOL] * float(np.sum(np.exp(-(ex * ex + ey * ey))))
    return tot


def group_residuals(basis, scs, qflat):
    """Stacked residual of the joint problem (for the scipy arbiter): own rows + pair rows once."""
    n = len(scs); nq2 = 2 * basis.nq
    qs = qflat.reshape(n, nq2)
    pos = group_positions(basis, scs, qs)
    out = [residuals(basis, scs[i], qs[i]).reshape(-1) for i in range(n)]
    for i in range(n):
        for j in partners(scs[i], i, n):
            if j > i:
                kc = OBS_K / scs[i][SC_RCOL]
                ex = (pos[i][0] - pos[j][0]) * kc; ey = (pos[i][1] - pos[j][1]) * kc
                out.append(math.sqrt(scs[i][SC_SCOL] * scs[i][SC_KCOL]) * np.exp(-0.5 * (ex * ex + ey * ey)))
    return np.concatenate(out)


GS_LS_SWEEP0, GS_LS_RATIO, GS_LS_FIRST_MAX, GS_LS_MAX = 8, 0.8, 8.0, 64.0      # include/d2d.h D2D_GS_LS_*


def group_merit(basis, scs, qs):
    """The joint cost as fit_groups_kernel's line search adds it up: every aircraft's cost against its partners' positions
    (own rows + ITS collision rows) minus half of its collision part -- each coupled pair once; equals group_cost when the
    masks and collision parameters of a group are symmetric."""
    n = len(scs)
    pos = group_positions(basis, scs, qs)
    tot = 0.0
    for i in range(n):
        oth = [pos[j] for j in partners(scs[i], i, n)]
        c_all = cost(basis, scs[i], qs[i], others=oth)
        tot += c_all - 0.5 * (c_all - cost(basis, scs[i], qs[i]))
    return tot


def bgs_solve(basis, scs, q0s=None, sweeps=30, inner_iters=8, tol=1e-12, ls=False, ls_s0=GS_LS_SWEEP0, ls_r0=GS_LS_RATIO,
              trace=None, **lm_kw):
    """Block Gauss-Seidel over the aircraft of one group -- the algorithm d2d_fit_solve_groups runs:
    visit aircraft 0..n-1 in order; each visit restarts the LM state and runs at most `inner_iters`
    damped solves on that aircraft's unknowns with the others' sampled positions frozen.  Stops when
    a whole sweep changes no unknown by more than tol*(1+|q|).
    ls (the persistent kernel fit_groups_kernel; include/d2d.h D2D_GS_LS_*): after a sweep -- from sweep ls_s0 on -- that moved at
    least ls_r0 x the move of the sweep before it, a line search on the joint cost along the sweep's direction d = X_k - X_{k-1}:
    first length rho / (1 - rho) clipped to [1, 8] (8 when the moves grow), doubled while the joint cost falls (<= 64), one try
    at a quarter when the first does not lower it; the best point below F(X_k) is taken.  The sweeps stay a descent on the
    joint cost and the stop test stays the move of a plain sweep.  trace: list that receives (sweep, moved, step length)."""
    n = len(scs)
    qs = [initial_guess(basis, scs[i]) if q0s is None else np.array(q0s[i], float) for i in range(n)]
    moved_prev = 1e300
    for sw in range(1, sweeps + 1):
        moved = 0.0
        q_before = [q.copy() for q in qs]
        for i in range(n):
            pos = group_positions(basis, scs, qs)
            oth = [pos[j] for j in partners(scs[i], i, n)]
            qn, c, it, st = lm_solve(basis, scs[i], q0=qs[i], max_iter=inner_iters, others=oth, **lm_kw)
            moved = max(moved, float(np.max(np.abs(qn - qs[i])) / (1.0 + np.max(np.abs(qs[i])))))
            qs[i] = qn
        if moved <= tol:
            if trace is not None: trace.append((sw, moved, 0.0))
            break
        best_al = 0.0
        if ls and sw >= ls_s0 and sw < sweeps and moved >= ls_r0 * moved_prev:
            rho = moved / moved_prev
            al = rho / max(1.0 - rho, 1e-3) if rho < 1.0 else GS_LS_FIRST_MAX
            al = min(max(al, 1.0), GS_LS_FIRST_MAX)
            d = [qs[i] - q_before[i] for i in range(n)]
            best_F, shrunk = group_merit(basis, scs, qs), False
            for t in range(5):
                if al > GS_LS_MAX:
                    break
                Ft = group_merit(basis, scs, [qs[i] + al * d[i] for i in range(n)])
                better = Ft < best_F
                if better:
                    best_F, best_al = Ft, al
                if t == 0:
                    if better: al *= 2.0
                    else: al *= 0.25; shrunk = True
                else:
                    if shrunk or not better: break
                    al *= 2.0
            if best_al > 0.0:
                qs = [qs[i] + best_al * d[i] for i in range(n)]
        if trace is not None: trace.append((sw, moved, best_al))
        moved_prev = moved
    return np.array(qs), group_cost(basis, scs, qs), sw


def coefficients(basis, sc, q):
    """Map q back to the reference's monomial layout coefs[0,:] per segment:
    returns z (2, S, 8) -- axis, segment, power (src/d2d/trajectory.py:54-72)."""
    nq = basis.nq
    dx, dy = end_data(sc)
    zx = basis.Zp @ dx + basis.Z @ q[:nq]
    zy = basis.Zp @ dy + basis.Z @ q[nq:]
    return np.stack([zx, zy]).reshape(2, basis.S, NCOEF)


def horner(coefs8, t):
    """PolynomialOne.get for derivative rows 0..3 (src/d2d/trajectory.py:74-82)."""
    out = np.zeros(4)
    for d in range(4):
        row = [arr(d, p + d) * coefs8[p + d] for p in range(NCOEF - d)] + [0.0] * d
        v = row[-1]
        for j in range(NCOEF - 2, -1, -1):
            v = v * t + row[j]
        out[d] = v
    return out


# ----------------------------------------------------------------------------------
# synthetic scenario generator (SURVEY.md 8d "synthetic inputs")
# ----------------------------------------------------------------------------------
def synth_scenarios(B, seed=20241008, rank=0, n_obs=2, wbnd=1.0, wwp=0.02):
    rng = np.random.default_rng(seed + rank)
    sc = np.zeros((B, SCEN_STRIDE))
    p0 = rng.uniform(-100, 100, (B, 2)); psi0 = rng.uniform(-np.pi, np.pi, B)
    dist = rng.uniform(30, 55, B); beta = rng.uniform(-np.pi, np.pi, B)
    p1 = p0 + dist[:, None] * np.stack([np.cos(beta), np.sin(beta)], 1)
    psi1 = rng.uniform(-np.pi, np.pi, B)
    sc[:, SC_X0], sc[:, SC_Y0], sc[:, SC_PSI0] = p0[:, 0], p0[:, 1], psi0
    sc[:, SC_X1], sc[:, SC_Y1], sc[:, SC_PSI1] = p1[:, 0], p1[:, 1], psi1
    sc[:, SC_VREF] = 12.0; sc[:, SC_VSP] = 12.0
    sc[:, SC_KV] = 5.0; sc[:, SC_KPHI] = 1.0; sc[:, SC_KOBS] = 1.0
    sc[:, SC_WWP] = wwp; sc[:, SC_GOLEFT] = -1.0; sc[:, SC_WBND] = wbnd
    sc[:, SC_PHIMAX] = PHI_MAX; sc[:, SC_VMIN] = V_MIN; sc[:, SC_VMAX] = V_MAX
    along = rng.uniform(0.2, 0.8, (B, 2)); lat = rng.uniform(5, 15, (B, 2)) * rng.choice([-1.0, 1.0], (B, 2))
    rad = rng.uniform(5, 15, (B, 2))
    u = (p1 - p0) / dist[:, None]; nrm = np.stack([-u[:, 1], u[:, 0]], 1)
    for i, (ox, oy, orr) in enumerate(SC_OBS[:2]):
        c = p0 + along[:, i, None] * (p1 - p0) + lat[:, i, None] * nrm
        sc[:, ox], sc[:, oy] = c[:, 0], c[:, 1]
        sc[:, orr] = rad[:, i] if i < n_obs else 0.0
    return sc


def set_scale(sc, obj_scale, K):
    sc[..., SC_S] = obj_scale / K
    return sc


# ----------------------------------------------------------------------------------
# MINPACK lmder restated on the normal equations -- the path scipy.optimize.least_squares(method='lm')
# follows (scipy/optimize/_lsq/least_squares.py call_minpack: diag = 1/x_scale = 1 -> mode 2, factor = 100,
# maxfev = 100 n).  lmder / lmpar work on the QR factors of J; every quantity they use is a function of
# J^T J and J^T f alone (the norms ||R^-T w|| of lmpar's Newton correction are w^T (J^T J + par I)^-1 w, the
# predicted reduction is p^T J^T J p + 2 par p^T p), so the same decisions can be taken from the Cholesky
# factor of J^T J + par I.  This is the `mode = D2D_LM_MINPACK` algorithm of the HIP path.
# ----------------------------------------------------------------------------------
MP_FACTOR = 100.0
MP_P1, MP_P5, MP_P25, MP_P75, MP_P0001 = 0.1, 0.5, 0.25, 0.75, 1e-4
MP_DWARF = 2.2250738585072014e-308
MP_EPS = 2.220446049250313e-16
MP_SLOW_TOL = 1e-4     # include/d2d.h D2D_LM_MP_SLOW_TOL


def _chol_solve(A, b, dtype):
    """p with A p = b and w |-> ||L^-1 w||^2 for the Cholesky factor L of A (in `dtype`); None when A is not positive definite."""
    try:
        L = np.linalg.cholesky(A.astype(dtype))
    except np.linalg.LinAlgError:
        return None, None
    import scipy.linalg as sl
    y = sl.solve_triangular(L, b.astype(dtype), lower=True)
    p = sl.solve_triangular(L.T, y, lower=False).astype(np.float64)
    return p, (lambda w: float(np.sum(sl.solve_triangular(L, w.astype(dtype), lower=True).astype(np.float64) ** 2)))


def lmpar_normal(H, g, delta, par, chol_dtype=np.float64):
    """MINPACK lmpar on the normal equations: p (the step is -p) and par with | ||p|| - delta | <= 0.1 delta, or par = 0 and the
    Gauss-Newton direction when that one is inside the region.  Returns p, par, number of factorisations."""
    n = len(g)
    I = np.eye(n)
    nfac = 1
    p, isq = _chol_solve(H, g, chol_dtype)
    it = 0
    if p is not None:
        dxnorm = float(np.linalg.norm(p))
        fp = dxnorm - delta
        if fp <= MP_P1 * delta:
            return p, 0.0, nfac
        temp2 = isq(p / dxnorm)                  # ||R^-T (D^2 p / ||D p||)||^2
        parl = (fp / delta) / temp2 if temp2 > 0.0 else 0.0
    else:                                        # rank-deficient (not positive definite to working precision): no lower bound
        dxnorm, fp, parl = np.inf, np.inf, 0.0
    gnorm = float(np.linalg.norm(g))
    paru = gnorm / delta
    if paru == 0.0:
        paru = MP_DWARF / min(delta, MP_P1)
    par = min(max(par, parl), paru)
    if par == 0.0:
        par = gnorm / dxnorm
    while True:
        it += 1
        if par == 0.0:
            par = max(MP_DWARF, 0.001 * paru)
        p, isq = _chol_solve(H + par * I, g, chol_dtype)
        nfac += 1
        if p is None:                            # cannot happen in exact arithmetic (par > 0): raise the damping
            parl = max(parl, par); par = max(2.0 * par, 0.001 * paru)
            if it >= 10:
                return np.zeros(n), par, nfac
            continue
        dxnorm = float(np.linalg.norm(p))
        temp = fp
        fp = dxnorm - delta
        if abs(fp) <= MP_P1 * delta or (parl == 0.0 and fp <= temp and temp < 0.0) or it == 10:
            break
        temp2 = isq(p / dxnorm)
        parc = (fp / delta) / temp2
        if fp > 0.0:
            parl = max(parl, par)
        if fp < 0.0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return p, par, nfac


def lmder_solve(basis, sc, q0=None, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=None, hess_dtype=np.float64,
                chol_dtype=np.float64, finish=None, trace=None, slow=None):
    """MINPACK's lmder (the solver behind scipy least_squares(method='lm')) on the normal equations, unit scaling (mode 2).
    f = the residual vector (cost c = ||f||^2), J its Jacobian.  One "iteration" here = one trial point (one nfev).
    finish = (n_ok, ...) see lm_finish_rule: hand over to the second-order loop once the trust region has been inactive
    (par = 0, ratio >= 0.75) for n_ok consecutive steps -- None: pure lmder.  slow (only with finish): hand over as well once
    `slow` trials in a row, accepted or not, each changed the cost by no more than MP_SLOW_TOL of itself (lmder stagnates in the
    zig-zag of Gauss-Newton on a large-residual fit: the full step overshoots, the radius shrinks, a damped step gains 1e-7, ...;
    scipy follows it for hundreds of evaluations and it is never calm) -- the hand-over is taken after an accepted step, like
    the calm one.
    Returns q, cost, nfev, status (ST_*), info dict."""
    wp = waypoints(sc, basis.K, basis.duration)
    x = initial_guess(basis, sc, wp) if q0 is None else np.array(q0, float)
    n = len(x)
    if max_nfev is None:
        max_nfev = 100 * n
    c, g, H = eval_normal(basis, sc, x, wp)
    H = H.astype(hess_dtype).astype(np.float64)
    nfev, nfac = 1, 0
    if not np.isfinite(c):
        return x, c, nfev, ST_NONFINITE, {}
    fnorm = math.sqrt(c)
    par = 0.0
    xnorm = float(np.linalg.norm(x))
    delta = MP_FACTOR * xnorm if xnorm > 0 else MP_FACTOR
    first = True
    info = 0
    calm = 0
    nslow = 0
    while True:
        acn = np.sqrt(np.maximum(np.diag(H), 0.0))
        gnorm = 0.0
        if fnorm != 0.0:
            ok = acn > 0
            gnorm = float(np.max(np.abs(g[ok]) / (acn[ok] * fnorm))) if ok.any() else 0.0
        if gnorm <= gtol:
            info = 4
            break
        while True:                               # inner loop: until a step is accepted
            p, par, nf = lmpar_normal(H, g, delta, par, chol_dtype)
            nfac += nf
            step = -p
            pnorm = float(np.linalg.norm(step))
            if first:
                delta = min(delta, pnorm)
                first = False
            xt = x + step
            ct = cost(basis, sc, xt, wp)
            nfev += 1
            fnorm1 = math.sqrt(ct) if np.isfinite(ct) else np.inf
            actred = -1.0
            if MP_P1 * fnorm1 < fnorm:
                actred = 1.0 - ct / c             # 1 - (fnorm1 / fnorm)^2
            # ||J p||^2 = p^T H p = p^T g - par p^T p  (H p = g - par p)
            jp2 = max(float(p @ g) - par * pnorm * pnorm, 0.0)
            t1 = jp2 / c                          # (||J p|| / fnorm)^2
            t2 = par * pnorm * pnorm / c          # (sqrt(par) ||p|| / fnorm)^2
            prered = t1 + t2 / MP_P5
            dirder = -(t1 + t2)
            ratio = actred / prered if prered != 0.0 else 0.0
            if ratio <= MP_P25:
                temp = MP_P5 if actred >= 0.0 else MP_P5 * dirder / (dirder + MP_P5 * actred)
                if MP_P1 * fnorm1 >= fnorm or temp < MP_P1:
                    temp = MP_P1
                delta = temp * min(delta, pnorm / MP_P1)
                par = par / temp
            elif par == 0.0 or ratio >= MP_P75:
                delta = pnorm / MP_P5
                par = MP_P5 * par
            if trace is not None:
                trace.append((nfev, c, ct, par, delta, ratio))
            accepted = ratio >= MP_P0001
            if accepted:
                x = xt
                xnorm = float(np.linalg.norm(x))
                c, fnorm = ct, fnorm1
            if abs(actred) <= ftol and prered <= ftol and MP_P5 * ratio <= 1.0:
                info = 1
            if delta <= xtol * xnorm:
                info = 2 if info == 0 else 3
            if info == 0:
                if nfev >= max_nfev:
                    info = 5
                elif abs(actred) <= MP_EPS and prered <= MP_EPS and MP_P5 * ratio <= 1.0:
                    info = 6
                elif delta <= MP_EPS * xnorm:
                    info = 7
                elif gnorm <= MP_EPS:
                    info = 8
            if accepted:
                calm = calm + 1 if (par == 0.0 and ratio >= MP_P75) else 0
            nslow = nslow + 1 if abs(actred) <= MP_SLOW_TOL else 0
            if info != 0 or accepted:
                break
        if info != 0:
            break
        if finish is not None and (calm >= finish or (slow and nslow >= slow)):
            break
        c, g, H = eval_normal(basis, sc, x, wp)
        H = H.astype(hess_dtype).astype(np.float64)
    status = ST_CONVERGED if info in (1, 2, 3, 4, 6, 7, 8) else ST_MAXITER
    return x, c, nfev, status, {'info': info, 'nfac': nfac, 'handover': info == 0}


MP_FINISH = 3          # include/d2d.h D2D_LM_MP_FINISH
MP_SLOW = 8            # include/d2d.h D2D_LM_MP_SLOW


def solve_minpack(basis, sc, q0=None, finish=MP_FINISH, max_iter=200, mp_tol=1e-15, ftol=1e-14, gtol=1e-9, xtol=1e-11,
                  hess_dtype=np.float64, chol_dtype=np.float64, slow=MP_SLOW):
    """The default solver of the HIP path (d2d_fit_opts.mode = D2D_LM_MODE_MINPACK): lmder until the trust region has been
    inactive for `finish` accepted steps in a row (or has stagnated for `slow` trials in a row, lmder_solve), then the second-order loop of lm_solve (every evaluation with the exact
    Hessian, damping restarted at LM_LAMBDA0) to the end; finish = 0: lmder alone.  max_iter bounds the trial points of both
    phases together.  Returns q, cost, iterations (trial points), status, info (factorisations of the lmder phase, its trials)."""
    q, c, nfev, st, info = lmder_solve(basis, sc, q0, ftol=mp_tol, xtol=mp_tol, gtol=mp_tol, max_nfev=max_iter + 1,
                                       hess_dtype=hess_dtype, chol_dtype=chol_dtype, finish=finish if finish > 0 else None,
                                       slow=slow if finish > 0 else None)
    it = nfev - 1
    out = {'nfac': info['nfac'], 'mp_trials': it, 'handover': info['handover']}
    if info['handover'] and it < max_iter:
        q, c, it2, st = lm_solve(basis, sc, q0=q, max_iter=max_iter - it, ftol=ftol, gtol=gtol, xtol=xtol, hess_dtype=hess_dtype,
                                 chol_dtype=chol_dtype, so_lambda=1e300)
        it += it2
    return q, c, it, st, out
