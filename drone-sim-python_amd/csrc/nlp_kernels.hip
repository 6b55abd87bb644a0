// Direct-collocation NLP of the reference's planners in its own parameterisation -- node values (x, y, psi, phi, v)(t_i),
// backward-Euler collocation equalities, end conditions, HARD box bounds -- solved on the device: the backend that stands
// where the reference builds `opty.direct_collocation.Problem(...)` and calls `.solve(x0)` (IPOPT)
// (src/single_opt_planner.py:62-71,124; src/multi_opt_planner.py:69-78,86).  Restates oracle/nlp.py line by line:
//   equalities  -> augmented Lagrangian (scaled multiplier estimate mu, penalty rho)
//   bounds      -> primal-dual log barrier (parameter mub, duals zL / zU), fraction-to-the-boundary rule
//   inner step  -> damped Newton on the block-tridiagonal system  H/2 + Sigma/2 + lam |diag|  (Lagrangian Hessian incl. the
//                  constraint curvature, 5x5 blocks, block Cholesky), backtracking line search on the barrier-AL merit function
// One problem per LANE: the recursion over the N nodes is sequential, the batch supplies the parallelism; every per-node
// quantity lives plane-major in HBM ([node][component][problem]: a wavefront touches 64 consecutive doubles).
#include <cmath>

#include "common.h"
#include "fit_device.h"      // scenario row columns (D2D_SC_*), FIT_G, FIT_OBS_K

#define NLP_NV 5
#define NLP_FAC 45           // per node: L (15, lower triangle by rows), Lo (25, block (i, i-1) of the factor), y (5)

struct NlpDims {
  int B, N;
  double h;
};

// per-problem scenario constants in registers
struct NlpScen {
  double p0[3], p1[3];
  double skv, skphi, vsp, wx, wy;            // s*kv, s*kphi (s = obj_scale / N [/ n_ac]); wind as it enters the eom (+w)
  double lo[NLP_NV], hi[NLP_NV];             // box (+-1e300: open)
  double wobs, wcol, kc2;                    // s*kobs, scol*kcol, (k / rcol)^2
  int n_obs, okind;
};

__device__ __forceinline__ NlpScen nlp_load_scen(const double *__restrict__ sc, const d2d_nlp_opts &o) {
  NlpScen s;
  s.p0[0] = sc[D2D_SC_X0]; s.p0[1] = sc[D2D_SC_Y0]; s.p0[2] = sc[D2D_SC_PSI0];
  s.p1[0] = sc[D2D_SC_X1]; s.p1[1] = sc[D2D_SC_Y1]; s.p1[2] = sc[D2D_SC_PSI1];
  const double ss = sc[D2D_SC_S];
  s.skv = ss * sc[D2D_SC_KV]; s.skphi = ss * sc[D2D_SC_KPHI]; s.vsp = sc[D2D_SC_VSP];
  s.wx = -sc[D2D_SC_WX]; s.wy = -sc[D2D_SC_WY];      // the row stores -w (planner convention, single_opt_planner.scen_row)
  s.lo[0] = -1e300; s.hi[0] = 1e300; s.lo[1] = -1e300; s.hi[1] = 1e300; s.lo[2] = -1e300; s.hi[2] = 1e300;
  if (sc[D2D_SC_XMIN] < sc[D2D_SC_XMAX]) { s.lo[0] = sc[D2D_SC_XMIN]; s.hi[0] = sc[D2D_SC_XMAX]; }
  if (sc[D2D_SC_YMIN] < sc[D2D_SC_YMAX]) { s.lo[1] = sc[D2D_SC_YMIN]; s.hi[1] = sc[D2D_SC_YMAX]; }
  s.lo[3] = -sc[D2D_SC_PHIMAX]; s.hi[3] = sc[D2D_SC_PHIMAX];
  s.lo[4] = sc[D2D_SC_VMIN]; s.hi[4] = sc[D2D_SC_VMAX];
  s.wobs = ss * sc[D2D_SC_KOBS];
  const double kcol = sc[D2D_SC_KCOL], rcol = sc[D2D_SC_RCOL];
  s.wcol = (kcol > 0.0 && rcol > 0.0) ? sc[D2D_SC_SCOL] * kcol : 0.0;
  s.kc2 = rcol > 0.0 ? (FIT_OBS_K / rcol) * (FIT_OBS_K / rcol) : 0.0;
  s.okind = (int)sc[D2D_SC_OKIND];
  int n = 0;
  for (int i = 0; i < D2D_MAX_OBS; ++i) {
    const double r = sc[(i < 2 ? D2D_SC_O0R + 3 * i : D2D_SC_OEXT + 3 * (i - 2) + 2)];
    if (r > 0.0) n = i + 1;
  }
  s.n_obs = n;
  (void)o;
  return s;
}

__device__ __forceinline__ void nlp_obs(const double *__restrict__ sc, int i, double &cx, double &cy, double &r) {
  const int c = i < 2 ? D2D_SC_O0X + 3 * i : D2D_SC_OEXT + 3 * (i - 2);
  cx = sc[c]; cy = sc[c + 1]; r = sc[c + 2];
}

// position-dependent exp terms of one node (oracle/nlp.py _obst_terms): adds the objective value (the function whose
// gradient is the reference's cost_grad: kind-1 terms scaled by (r/k)^2), its half gradient and half Gauss-Newton block;
// cost_ref accumulates the reference's cost() value of the same terms.
__device__ __forceinline__ void nlp_exp_terms(const NlpScen &s, const double *__restrict__ sc, const double *__restrict__ partner,
                                              long pidx, long pstride, double x, double y, double &obj, double &cost_ref,
                                              double *gx, double *gy, double *dxx, double *dxy, double *dyy) {
  for (int i = 0; i < s.n_obs; ++i) {
    double cx, cy, r;
    nlp_obs(sc, i, cx, cy, r);
    if (!(r > 0.0)) continue;
    const double dx = x - cx, dy = y - cy;
    double k2, w, e;
    if ((s.okind >> i) & 1) {       // kind 0: e = clip(exp(r^2 - d^2), 0, 1e3); cost_grad ignores the clip (src/d2d/opty_utils.py:108-127)
      k2 = 1.0; w = s.wobs;
      e = exp(fmin(r * r - (dx * dx + dy * dy), 6.907755278982137));
      cost_ref += w * e;
    } else {                        // kind 1: e = exp(-|k (p - c) / r|^2); cost_grad omits (k/r)^2 (:118-131)
      k2 = (FIT_OBS_K / r) * (FIT_OBS_K / r);
      e = exp(-(dx * dx + dy * dy) * k2);
      cost_ref += s.wobs * e;
      w = s.wobs / k2;
    }
    const double we = w * e;
    obj += we;
    if (gx) {
      *gx += -k2 * we * dx; *gy += -k2 * we * dy;
      *dxx += k2 * k2 * we * dx * dx; *dxy += k2 * k2 * we * dx * dy; *dyy += k2 * k2 * we * dy * dy;
    }
  }
  if (partner != nullptr && s.wcol > 0.0) {     // CostCollision against the frozen partner (src/d2d/multiopty_utils.py:120-153)
    const double dx = x - partner[pidx], dy = y - partner[pidx + pstride];
    const double e = exp(-(dx * dx + dy * dy) * s.kc2);
    cost_ref += s.wcol * e;
    const double we = s.wcol / s.kc2 * e;
    obj += we;
    if (gx) {
      *gx += -s.kc2 * we * dx; *gy += -s.kc2 * we * dy;
      *dxx += s.kc2 * s.kc2 * we * dx * dx; *dxy += s.kc2 * s.kc2 * we * dx * dy; *dyy += s.kc2 * s.kc2 * we * dy * dy;
    }
  }
}

struct NlpBuf {
  double *W, *zL, *zU, *mu, *fac, *dw, *rhs;     // [N][5], [N][5], [N][5], [N][3], [N][45], [N][5], [N][5]  (x B, plane-major)
};

#define NLP_AT(arr, C, i, c) (arr)[((long)(i) * (C) + (c)) * B + b]

// collocation residual of node i (>= 1) from the states of node i-1 and the node itself (oracle/nlp.py constraints)
__device__ __forceinline__ void nlp_constraint(const NlpScen &s, double h, const double wp[3], const double w[NLP_NV], double c[3]) {
  double sp, cp;
  sincos(w[2], &sp, &cp);
  c[0] = (w[0] - wp[0]) / h - w[4] * cp + s.wx;
  c[1] = (w[1] - wp[1]) / h - w[4] * sp + s.wy;
  c[2] = (w[2] - wp[2]) / h - FIT_G / w[4] * tan(w[3]);
}

__device__ __forceinline__ bool nlp_fixed(int i, int N, int c) { return c < 3 && (i == 0 || i == N - 1); }

// Merit function of the inner problem at W + a*dw: objective + rho sum (c + mu)^2 - mub sum log(slacks); +inf outside the box.
// Also returns (a == 0 only) nothing else.  One sweep over the nodes.
__device__ double nlp_merit(const NlpDims &d, const NlpScen &s, const double *__restrict__ sc, const double *__restrict__ partner,
                            const NlpBuf &u, int b, double a, double rho, double mub, double *cost_ref_out, double *feas_out) {
  const int B = d.B, N = d.N;
  double val = 0.0, bar = 0.0, cref = 0.0, feas = 0.0;
  double wp[3] = {0, 0, 0};
  bool inside = true;
  for (int i = 0; i < N; ++i) {
    double w[NLP_NV];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) w[c] = NLP_AT(u.W, NLP_NV, i, c) + (a != 0.0 ? a * NLP_AT(u.dw, NLP_NV, i, c) : 0.0);
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      if (nlp_fixed(i, N, c)) continue;
      if (s.lo[c] > -1e299) { const double sl = w[c] - s.lo[c]; inside = inside && sl > 0.0; bar += log(sl > 0.0 ? sl : 1.0); }
      if (s.hi[c] < 1e299) { const double su = s.hi[c] - w[c]; inside = inside && su > 0.0; bar += log(su > 0.0 ? su : 1.0); }
    }
    const double dv = w[4] - s.vsp;
    double obj = s.skv * dv * dv + s.skphi * w[3] * w[3];
    cref += obj;
    nlp_exp_terms(s, sc, partner, (long)i * 2 * B + b, B, w[0], w[1], obj, cref, nullptr, nullptr, nullptr, nullptr, nullptr);
    val += obj;
    if (i >= 1) {
      double c3[3];
      nlp_constraint(s, d.h, wp, w, c3);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double cm = c3[k] + NLP_AT(u.mu, 3, i, k);
        val += rho * cm * cm;
        feas = fmax(feas, fabs(c3[k]));
      }
    }
    wp[0] = w[0]; wp[1] = w[1]; wp[2] = w[2];
  }
  if (cost_ref_out) *cost_ref_out = cref;
  if (feas_out) *feas_out = feas;
  if (!inside || !(fabs(val) <= 1.79e308)) return INFINITY;
  return val - mub * bar;
}

// Sweep 1: assemble node by node (half gradient g, half Hessian blocks D, E with the constraint curvature), add the barrier
// diagonal and the damping, block Cholesky + forward substitution; stores L, Lo, y and rhs.  Returns false if a pivot is not
// positive.  err_out: barrier KKT error of the inner problem (stationarity with the duals, complementarity).
__device__ bool nlp_factor(const NlpDims &d, const NlpScen &s, const double *__restrict__ sc, const double *__restrict__ partner,
                           const NlpBuf &u, int b, double rho, double mub, double lam, double *err_out) {
  const int B = d.B, N = d.N;
  const double h = d.h, ih = 1.0 / h;
  double Lp[15], yp[NLP_NV];                    // factor of the previous node's diagonal block, its forward-substituted rhs
  double wp[3] = {0, 0, 0}, wc[NLP_NV], wn[NLP_NV];
  double cc[3] = {0, 0, 0}, cn[3] = {0, 0, 0};  // (c + mu) of the constraint that ends at this node / at the next one
  double Ac[3][NLP_NV];                         // Jacobian of the current constraint wrt this node
  double err = 0.0;
  bool ok = true;
#pragma unroll
  for (int c = 0; c < NLP_NV; ++c) wc[c] = NLP_AT(u.W, NLP_NV, 0, c);
  for (int i = 0; i < N; ++i) {
    const bool has_next = i + 1 < N;
    if (has_next) {
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) wn[c] = NLP_AT(u.W, NLP_NV, i + 1, c);
      nlp_constraint(s, h, wc, wn, cn);
#pragma unroll
      for (int k = 0; k < 3; ++k) cn[k] += NLP_AT(u.mu, 3, i + 1, k);
    }
    double g[NLP_NV] = {0, 0, 0, 0, 0};
    double D[NLP_NV][NLP_NV];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a)
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) D[a][c] = 0.0;
    // cost rows
    D[4][4] += s.skv; g[4] += s.skv * (wc[4] - s.vsp);
    D[3][3] += s.skphi; g[3] += s.skphi * wc[3];
    {
      double obj = 0.0, cref = 0.0;
      nlp_exp_terms(s, sc, partner, (long)i * 2 * B + b, B, wc[0], wc[1], obj, cref, &g[0], &g[1], &D[0][0], &D[0][1], &D[1][1]);
      D[1][0] = D[0][1];
    }
    double E[NLP_NV][3];                        // block (i, i-1): only the (x, y, psi) columns of node i-1 are non-zero
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) { E[a][0] = 0.0; E[a][1] = 0.0; E[a][2] = 0.0; }
    if (i >= 1) {
      double sp, cp;
      sincos(wc[2], &sp, &cp);
      const double tp = tan(wc[3]), vi = wc[4], sec2 = 1.0 + tp * tp;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < NLP_NV; ++c) Ac[k][c] = 0.0;
      Ac[0][0] = ih; Ac[0][2] = vi * sp; Ac[0][4] = -cp;
      Ac[1][1] = ih; Ac[1][2] = -vi * cp; Ac[1][4] = -sp;
      Ac[2][2] = ih; Ac[2][3] = -FIT_G * sec2 / vi; Ac[2][4] = FIT_G * tp / (vi * vi);
#pragma unroll
      for (int a = 0; a < NLP_NV; ++a) {
#pragma unroll
        for (int c = 0; c < NLP_NV; ++c) D[a][c] += rho * (Ac[0][a] * Ac[0][c] + Ac[1][a] * Ac[1][c] + Ac[2][a] * Ac[2][c]);
        g[a] += rho * (Ac[0][a] * cc[0] + Ac[1][a] * cc[1] + Ac[2][a] * cc[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) E[a][k] = -rho * Ac[k][a] * ih;
      }
      // + rho (c + mu) Hessian(c): the constraint curvature of the Lagrangian
      const double m0 = rho * cc[0], m1 = rho * cc[1], m2 = rho * cc[2];
      D[2][2] += m0 * vi * cp + m1 * vi * sp;
      const double d24 = m0 * sp - m1 * cp;
      D[2][4] += d24; D[4][2] += d24;
      D[3][3] += -m2 * 2.0 * FIT_G * tp * sec2 / vi;
      const double d34 = m2 * FIT_G * sec2 / (vi * vi);
      D[3][4] += d34; D[4][3] += d34;
      D[4][4] += -m2 * 2.0 * FIT_G * tp / (vi * vi * vi);
    }
    if (has_next) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { D[k][k] += rho * ih * ih; g[k] += -rho * cn[k] * ih; }
    }
    // barrier terms, stationarity / complementarity error, right-hand side
    double rhs[NLP_NV];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      const bool fx = nlp_fixed(i, N, c);
      double sig = 0.0, r = -2.0 * g[c], st = 2.0 * g[c];
      if (!fx && s.lo[c] > -1e299) {
        const double sl = wc[c] - s.lo[c], z = NLP_AT(u.zL, NLP_NV, i, c);
        sig += z / sl; r += mub / sl; st -= z;
        err = fmax(err, fabs(z * sl - mub));
      }
      if (!fx && s.hi[c] < 1e299) {
        const double su = s.hi[c] - wc[c], z = NLP_AT(u.zU, NLP_NV, i, c);
        sig += z / su; r -= mub / su; st += z;
        err = fmax(err, fabs(z * su - mub));
      }
      if (!fx) err = fmax(err, fabs(st));
      rhs[c] = fx ? 0.0 : r;
      NLP_AT(u.rhs, NLP_NV, i, c) = rhs[c];
      D[c][c] += 0.5 * sig;
      D[c][c] += lam * fmax(fabs(D[c][c]), 1e-12);
      if (fx) {
#pragma unroll
        for (int a = 0; a < NLP_NV; ++a) { D[c][a] = 0.0; D[a][c] = 0.0; }
        D[c][c] = 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) E[c][k] = 0.0;
      }
    }
    if (i >= 1 && (i - 1 == 0)) {               // columns of E that belong to fixed variables of node i-1
#pragma unroll
      for (int a = 0; a < NLP_NV; ++a) { E[a][0] = 0.0; E[a][1] = 0.0; E[a][2] = 0.0; }
    }
    // Lo = E Lp^-T (row a of Lo solves Lp x = E[a,:]^T), S = D - Lo Lo^T, L = chol(S), y = L^-1 (rhs/2 - Lo yp)
    double Lo[NLP_NV][NLP_NV];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a)
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) Lo[a][c] = 0.0;
    double t[NLP_NV];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) t[c] = 0.5 * rhs[c];
    if (i >= 1) {
#pragma unroll
      for (int a = 0; a < NLP_NV; ++a) {
#pragma unroll
        for (int c = 0; c < NLP_NV; ++c) {
          double v = c < 3 ? E[a][c] : 0.0;
#pragma unroll
          for (int k = 0; k < c; ++k) v -= Lp[c * (c + 1) / 2 + k] * Lo[a][k];
          Lo[a][c] = v / Lp[c * (c + 1) / 2 + c];
        }
      }
#pragma unroll
      for (int a = 0; a < NLP_NV; ++a) {
#pragma unroll
        for (int c = 0; c < NLP_NV; ++c) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < NLP_NV; ++k) v += Lo[a][k] * Lo[c][k];
          D[a][c] -= v;
        }
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NLP_NV; ++k) v += Lo[a][k] * yp[k];
        t[a] -= v;
      }
    }
    double L[15];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) {
#pragma unroll
      for (int c = 0; c <= a; ++c) {
        double v = D[a][c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= L[a * (a + 1) / 2 + k] * L[c * (c + 1) / 2 + k];
        if (a == c) {
          if (!(v > 0.0)) { ok = false; v = 1.0; }
          L[a * (a + 1) / 2 + a] = sqrt(v);
        } else {
          L[a * (a + 1) / 2 + c] = v / L[c * (c + 1) / 2 + c];
        }
      }
    }
    double y[NLP_NV];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) {
      double v = t[a];
#pragma unroll
      for (int k = 0; k < a; ++k) v -= L[a * (a + 1) / 2 + k] * y[k];
      y[a] = v / L[a * (a + 1) / 2 + a];
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) NLP_AT(u.fac, NLP_FAC, i, k) = L[k];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a)
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) NLP_AT(u.fac, NLP_FAC, i, 15 + a * NLP_NV + c) = Lo[a][c];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) NLP_AT(u.fac, NLP_FAC, i, 40 + a) = y[a];
    // slide the window
#pragma unroll
    for (int k = 0; k < 15; ++k) Lp[k] = L[k];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) yp[a] = y[a];
    wp[0] = wc[0]; wp[1] = wc[1]; wp[2] = wc[2];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) wc[c] = wn[c];
#pragma unroll
    for (int k = 0; k < 3; ++k) cc[k] = cn[k];
  }
  (void)wp;
  *err_out = err;
  return ok;
}

// Sweep 2 (backward): dw_i = L_i^-T (y_i - Lo_{i+1}^T dw_{i+1}); directional derivative of the merit function, largest primal
// step (fraction to the boundary), and the dual steps' largest fraction.
__device__ void nlp_backsolve(const NlpDims &d, const NlpScen &s, const NlpBuf &u, int b, double mub, double tau, double *dphi_out,
                              double *amax_out, double *az_out) {
  const int B = d.B, N = d.N;
  double dn[NLP_NV] = {0, 0, 0, 0, 0}, Lon[NLP_NV][NLP_NV];
  double dphi = 0.0, amax = 1.0, az = 1.0;
  for (int i = N - 1; i >= 0; --i) {
    double L[15], t[NLP_NV], dw[NLP_NV];
#pragma unroll
    for (int k = 0; k < 15; ++k) L[k] = NLP_AT(u.fac, NLP_FAC, i, k);
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) t[a] = NLP_AT(u.fac, NLP_FAC, i, 40 + a);
    if (i + 1 < N) {
#pragma unroll
      for (int a = 0; a < NLP_NV; ++a) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NLP_NV; ++k) v += Lon[k][a] * dn[k];
        t[a] -= v;
      }
    }
#pragma unroll
    for (int a = NLP_NV - 1; a >= 0; --a) {
      double v = t[a];
#pragma unroll
      for (int k = a + 1; k < NLP_NV; ++k) v -= L[k * (k + 1) / 2 + a] * dw[k];
      dw[a] = v / L[a * (a + 1) / 2 + a];
    }
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      const bool fx = nlp_fixed(i, N, c);
      if (fx) dw[c] = 0.0;
      NLP_AT(u.dw, NLP_NV, i, c) = dw[c];
      dphi -= NLP_AT(u.rhs, NLP_NV, i, c) * dw[c];
      if (fx) continue;
      const double w = NLP_AT(u.W, NLP_NV, i, c);
      if (s.lo[c] > -1e299) {
        const double sl = w - s.lo[c], z = NLP_AT(u.zL, NLP_NV, i, c);
        if (dw[c] < 0.0) amax = fmin(amax, -tau * sl / dw[c]);
        const double dz = mub / sl - z - z / sl * dw[c];
        if (dz < 0.0) az = fmin(az, -tau * z / dz);
      }
      if (s.hi[c] < 1e299) {
        const double su = s.hi[c] - w, z = NLP_AT(u.zU, NLP_NV, i, c);
        if (dw[c] > 0.0) amax = fmin(amax, tau * su / dw[c]);
        const double dz = mub / su - z + z / su * dw[c];
        if (dz < 0.0) az = fmin(az, -tau * z / dz);
      }
    }
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) {
      dn[a] = dw[a];
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) Lon[a][c] = NLP_AT(u.fac, NLP_FAC, i, 15 + a * NLP_NV + c);
    }
  }
  *dphi_out = dphi; *amax_out = amax; *az_out = az;
}

// Sweep 4: take the step -- W += a dw, duals += az dz (dz from the step's dw), duals kept near the central path.
__device__ void nlp_apply(const NlpDims &d, const NlpScen &s, const NlpBuf &u, int b, double a, double az, double mub) {
  const int B = d.B, N = d.N;
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      if (nlp_fixed(i, N, c)) continue;
      const double w = NLP_AT(u.W, NLP_NV, i, c), dw = NLP_AT(u.dw, NLP_NV, i, c);
      const double wn = w + a * dw;
      NLP_AT(u.W, NLP_NV, i, c) = wn;
      if (s.lo[c] > -1e299) {
        const double sl = w - s.lo[c], z = NLP_AT(u.zL, NLP_NV, i, c), sn = wn - s.lo[c];
        double zn = z + az * (mub / sl - z - z / sl * dw);
        zn = fmin(fmax(zn, mub / (1e10 * sn)), 1e10 * mub / sn);
        NLP_AT(u.zL, NLP_NV, i, c) = zn;
      }
      if (s.hi[c] < 1e299) {
        const double su = s.hi[c] - w, z = NLP_AT(u.zU, NLP_NV, i, c), sn = s.hi[c] - wn;
        double zn = z + az * (mub / su - z + z / su * dw);
        zn = fmin(fmax(zn, mub / (1e10 * sn)), 1e10 * mub / sn);
        NLP_AT(u.zU, NLP_NV, i, c) = zn;
      }
    }
  }
}

__global__ void __launch_bounds__(64)
nlp_solve_kernel(NlpDims d, d2d_nlp_opts o, const double *__restrict__ scen, const double *__restrict__ partner, NlpBuf u,
                 double *__restrict__ cost_out, double *__restrict__ feas_out, int32_t *__restrict__ iters_out,
                 int32_t *__restrict__ status_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int B = d.B, N = d.N;
  const double *sc = scen + (size_t)b * D2D_SCEN_STRIDE;
  const NlpScen s = nlp_load_scen(sc, o);
  // ---- start: end conditions in place, everything else pushed strictly inside the box; duals on the central path
  double mub = o.mub0;
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      double w = NLP_AT(u.W, NLP_NV, i, c);
      double zl = 0.0, zu = 0.0;
      if (nlp_fixed(i, N, c)) {
        w = (i == 0) ? s.p0[c] : s.p1[c];
      } else {
        const bool hl = s.lo[c] > -1e299, hu = s.hi[c] < 1e299;
        const double width = (hl && hu) ? s.hi[c] - s.lo[c] : 1e300;
        const double kap = fmin(1e-2 * fmax(1.0, fabs(w)), 1e-2 * width);
        if (hl) w = fmax(w, s.lo[c] + kap);
        if (hu) w = fmin(w, s.hi[c] - kap);
        if (hl) zl = mub / (w - s.lo[c]);
        if (hu) zu = mub / (s.hi[c] - w);
      }
      NLP_AT(u.W, NLP_NV, i, c) = w;
      NLP_AT(u.zL, NLP_NV, i, c) = zl; NLP_AT(u.zU, NLP_NV, i, c) = zu;
      NLP_AT(u.dw, NLP_NV, i, c) = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) NLP_AT(u.mu, 3, i, k) = 0.0;
  }
  double rho = o.rho0, lam = D2D_LM_LAMBDA0, feas_prev = INFINITY;
  int total_inner = 0, status = D2D_ST_MAXITER;
  double err = 0.0, cost_ref = 0.0, feas = 0.0;
  int n_stalled = 0;                        // outer iterations in a row at the largest penalty without feasibility progress
  for (int outer = 1; outer <= o.outer_max; ++outer) {
    const double tol_in = fmax(fmax(o.opt_tol, fmin(1e-1, 10.0 * mub)), D2D_NLP_GRAD_FLOOR * rho);
    // merit value of the current point for this (mub, rho, mu): one sweep here, afterwards the accepted trial's value
    double phi0 = nlp_merit(d, s, sc, partner, u, b, 0.0, rho, mub, nullptr, nullptr);
    for (int it = 0; it < o.inner_max; ++it) {
      ++total_inner;
      bool accepted = false, converged = false;
      for (int tr = 0; tr < 30; ++tr) {
        const bool pd = nlp_factor(d, s, sc, partner, u, b, rho, mub, lam, &err);
        if (err <= tol_in) { converged = true; break; }      // (the error does not depend on the damping)
        if (!pd) { lam = fmin(lam * 8.0, D2D_LM_LAMBDA_MAX); continue; }
        const double tau = fmax(0.99, 1.0 - mub);
        double dphi, amax, az;
        nlp_backsolve(d, s, u, b, mub, tau, &dphi, &amax, &az);
        if (!(dphi < 0.0)) { lam = fmin(lam * 8.0, D2D_LM_LAMBDA_MAX); continue; }
        double a = amax, pt = 0.0;
        bool ok = false;
        for (int ls = 0; ls < 8; ++ls) {
          pt = nlp_merit(d, s, sc, partner, u, b, a, rho, mub, nullptr, nullptr);
          if (pt <= phi0 + 1e-4 * a * dphi) { ok = true; break; }
          a *= 0.5;
        }
        if (ok) {
          phi0 = pt;
          nlp_apply(d, s, u, b, a, az, mub);
          if (a == amax) lam = fmax(lam / 3.0, D2D_LM_LAMBDA_MIN);
          accepted = true;
          break;
        }
        lam = fmin(lam * 4.0, D2D_LM_LAMBDA_MAX);
      }
      if (converged || !accepted) break;
    }
    (void)nlp_merit(d, s, sc, partner, u, b, 0.0, rho, mub, &cost_ref, &feas);
    if (feas <= o.feas_tol && mub <= o.mub_min * 1.0001 && err <= tol_in) { status = D2D_ST_CONVERGED; break; }
    // an infeasible problem (e.g. end points too far apart for v_max) sits at the largest penalty with its constraint violation
    // no longer shrinking: give up instead of holding the whole batch for outer_max x inner_max steps
    n_stalled = (rho >= D2D_NLP_RHO_MAX && feas > 0.5 * feas_prev && feas > 1e3 * o.feas_tol) ? n_stalled + 1 : 0;
    if (n_stalled >= 3) { status = D2D_ST_STALLED; break; }
    // first-order multiplier update (lambda = 2 rho mu); the penalty grows when feasibility stalls
    const bool grow = feas > 0.25 * feas_prev && rho < D2D_NLP_RHO_MAX;
    {
      double wp[3] = {NLP_AT(u.W, NLP_NV, 0, 0), NLP_AT(u.W, NLP_NV, 0, 1), NLP_AT(u.W, NLP_NV, 0, 2)};
      for (int i = 1; i < N; ++i) {
        double w[NLP_NV], c3[3];
#pragma unroll
        for (int c = 0; c < NLP_NV; ++c) w[c] = NLP_AT(u.W, NLP_NV, i, c);
        nlp_constraint(s, d.h, wp, w, c3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double m = NLP_AT(u.mu, 3, i, k) + c3[k];
          NLP_AT(u.mu, 3, i, k) = grow ? m / D2D_NLP_RHO_GROW : m;
        }
        wp[0] = w[0]; wp[1] = w[1]; wp[2] = w[2];
      }
    }
    if (grow) rho *= D2D_NLP_RHO_GROW;
    feas_prev = feas;
    mub = fmax(o.mub_min, fmin(0.2 * mub, mub * sqrt(mub)));
  }
  (void)nlp_merit(d, s, sc, partner, u, b, 0.0, rho, mub, &cost_ref, &feas);
  cost_out[b] = cost_ref;
  feas_out[b] = feas;
  if (iters_out) iters_out[b] = total_inner;
  if (status_out) status_out[b] = status;
}

extern "C" {

int d2d_nlp_workspace_doubles(int N) { return N * (3 * NLP_NV + 3 + NLP_FAC + NLP_NV); }

int d2d_nlp_solve(d2d_ctx *ctx, int B, int N, double h, const double *scen, const d2d_nlp_opts *opts, double *W,
                  const double *partner, double *work, double *mult, double *cost, double *feas, int32_t *iters, int32_t *status) {
  D2D_REQUIRE(ctx && scen && W && work && cost && feas, "d2d_nlp_solve: null argument");
  D2D_REQUIRE(B >= 1 && N >= 3 && h > 0, "d2d_nlp_solve: B >= 1, N >= 3, h > 0 required (B=%d N=%d h=%g)", B, N, h);
  d2d_nlp_opts o = {D2D_NLP_RHO0, D2D_NLP_MUB0, D2D_NLP_MUB_MIN, 1e-9, 1e-7, 60, 40};
  if (opts) o = *opts;
  D2D_REQUIRE(o.inner_max >= 1 && o.outer_max >= 1 && o.rho0 > 0 && o.mub0 > 0 && o.mub_min > 0, "d2d_nlp_solve: bad options");
  NlpBuf u;
  const size_t nb = (size_t)N * B;
  u.W = W;
  u.zL = work; u.zU = u.zL + NLP_NV * nb; u.dw = u.zU + NLP_NV * nb; u.rhs = u.dw + NLP_NV * nb;
  u.fac = u.rhs + NLP_NV * nb;
  u.mu = mult ? mult : u.fac + NLP_FAC * nb;
  const NlpDims d{B, N, h};
  hipLaunchKernelGGL(nlp_solve_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, d, o, scen, partner, u, cost, feas, iters, status);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

}  // extern "C"
