// Direct-collocation NLP of the reference's planners in its own parameterisation -- node values (x, y, psi, phi, v)(t_i),
// backward-Euler collocation equalities, end conditions, HARD box bounds -- solved on the device: the backend that stands
// where the reference builds `opty.direct_collocation.Problem(...)` and calls `.solve(x0)` (IPOPT)
// (src/single_opt_planner.py:62-71,124; src/multi_opt_planner.py:69-78,86).  The algorithm is oracle/nlp.py's, step for step:
//   equalities  -> augmented Lagrangian (scaled multiplier estimate mu, penalty rho)
//   bounds      -> primal-dual log barrier (parameter mub, duals zL / zU), fraction-to-the-boundary rule
//   inner step  -> damped Newton on the block-tridiagonal system  H/2 + Sigma/2 + lam |diag|  (Lagrangian Hessian incl. the
//                  constraint curvature), backtracking line search on the barrier-AL merit function
// (the oracle solves that system with banded LAPACK; here: per-node elimination of phi, v + 3x3 block Cholesky, see below).
// One problem per WAVEFRONT.  Everything that is independent from node to node -- merit function, assembly of the Hessian
// blocks, ratio tests, the update -- runs with lane = node (chunks of 64, DPP reductions); only the two block recursions
// (Cholesky + forward substitution, back substitution) are serial in the nodes and run wave-uniform with the next node's
// inputs requested a node ahead.  Per-problem data lives in planes of N doubles ([component][node]: coalesced for lane = node,
// one broadcast request for the serial recursions); control flow is uniform, so no problem waits for another one's line search.
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "fit_device.h"      // scenario row columns (D2D_SC_*), FIT_G, FIT_OBS_K

#define NLP_NV 5

// per-problem scenario constants in registers
struct NlpScen {
  double p0[3], p1[3];
  double skv, skphi, vsp, wx, wy;            // s*kv, s*kphi (s = obj_scale / N [/ n_ac]); wind as it enters the eom (+w)
  double sbank;                              // CostBank(use_mean=False): obj_scale * kbank of the term sbank * max_i phi_i^2 (skphi = 0 then); 0 = mean mode
  double lo[NLP_NV], hi[NLP_NV];             // box (+-1e300: open)
  double wobs, wcol, kc2;                    // s*kobs, scol*kcol, (k / rcol)^2
  int n_obs, okind;
};

__device__ __forceinline__ NlpScen nlp_load_scen(const double *__restrict__ sc, const d2d_nlp_opts &o, const double *__restrict__ bnd) {
  NlpScen s;
  s.p0[0] = sc[D2D_SC_X0]; s.p0[1] = sc[D2D_SC_Y0]; s.p0[2] = sc[D2D_SC_PSI0];
  s.p1[0] = sc[D2D_SC_X1]; s.p1[1] = sc[D2D_SC_Y1]; s.p1[2] = sc[D2D_SC_PSI1];
  const double ss = sc[D2D_SC_S];
  s.skv = ss * sc[D2D_SC_KV]; s.skphi = ss * sc[D2D_SC_KPHI]; s.vsp = sc[D2D_SC_VSP];
  s.sbank = 0.0;                             // (nlp_solve_one sets it for D2D_SC_BANKMAX rows: it needs the node count)
  s.wx = -sc[D2D_SC_WX]; s.wy = -sc[D2D_SC_WY];      // the row stores -w (planner convention, single_opt_planner.scen_row)
  s.lo[0] = -1e300; s.hi[0] = 1e300; s.lo[1] = -1e300; s.hi[1] = 1e300; s.lo[2] = -1e300; s.hi[2] = 1e300;
  if (sc[D2D_SC_XMIN] < sc[D2D_SC_XMAX]) { s.lo[0] = sc[D2D_SC_XMIN]; s.hi[0] = sc[D2D_SC_XMAX]; }
  if (sc[D2D_SC_YMIN] < sc[D2D_SC_YMAX]) { s.lo[1] = sc[D2D_SC_YMIN]; s.hi[1] = sc[D2D_SC_YMAX]; }
  s.lo[3] = -sc[D2D_SC_PHIMAX]; s.hi[3] = sc[D2D_SC_PHIMAX];
  s.lo[4] = sc[D2D_SC_VMIN]; s.hi[4] = sc[D2D_SC_VMAX];
  if (bnd) {                                           // d2d_nlp_opts.bounds: an asymmetric phi interval, a box on psi
    if (bnd[0] < bnd[1]) { s.lo[3] = bnd[0]; s.hi[3] = bnd[1]; }
    if (bnd[2] < bnd[3]) { s.lo[2] = bnd[2]; s.hi[2] = bnd[3]; }
  }
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
    double k2, w, e, f;
    if ((s.okind >> i) & 1) {       // kind 0: e = clip(exp(r^2 - d^2), 0, 1e3) in cost AND in cost_grad = -2 dx e (src/d2d/opty_utils.py:108-131):
      k2 = 1.0; w = s.wobs;         // on the clip that is the gradient of 1e3 (1 + arg - log 1e3), not of the constant cost -- the
      const double arg = r * r - (dx * dx + dy * dy);      // objective minimised continues the term that way (oracle/nlp.py _obst_terms)
      e = exp(fmin(arg, 6.907755278982137));
      f = arg > 6.907755278982137 ? 1e3 * (1.0 + arg - 6.907755278982137) : e;
      cost_ref += w * e;
    } else {                        // kind 1: e = exp(-|k (p - c) / r|^2); cost_grad omits (k/r)^2 (:118-131)
      k2 = (FIT_OBS_K / r) * (FIT_OBS_K / r);
      e = exp(-(dx * dx + dy * dy) * k2);
      cost_ref += s.wobs * e;
      w = s.wobs / k2;
      f = e;
    }
    const double we = w * e;
    obj += w * f;
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

// ---- per-problem workspace (WS_TOTAL * N doubles) ---------------------------------------------------------------------------
// Data that the node-parallel phases touch lives in planes of N doubles ([component][node]: lane = node reads 64 consecutive
// doubles).  What the two serial recursions read and write is NODE-major ([node][component]): one base address per node and
// immediate offsets (plane-major, every component would need its own 64-bit address -- measured: that address arithmetic
// alone cost more than the block arithmetic).
enum {
  WS_ZL = 0, WS_ZU = 5,                  // bound duals (5 + 5 planes)
  WS_MU = 10,                            // scaled multiplier estimates (3 planes) unless the caller wants them (mult)
  WS_DW = 13,                            // step (5 planes)
  WS_RHS = 18,                           // right-hand side (5 planes)
  WS_EL = 23,                            // local elimination of (phi, v) (17 planes): 1/l00, l10, 1/l11; Qt [j][k] (6); Rt [j][k] (6); tt (2)
  WS_UP = 40,                            // what node i hands to node i-1 (9 planes): Rt^T Rt lower triangle (6), Rt^T tt (3)
  WS_SIN = 49,                           // reduced node [N][18]: D' lower triangle (6), E' [a][k] (9), t' (3)
  WS_SF = 67,                            // factor of the reduced system [N][18]: L (6, RECIPROCAL diagonal), Lo [a][k] (9), y (3)
  WS_DS = 85,                            // reduced step [N][3]
  WS_TOTAL = 88,
  EL_LP = 0, EL_Q = 3, EL_R = 9, EL_T = 15,
  UP_RR = 0, UP_RT = 6,
  SIN_D = 0, SIN_E = 6, SIN_T = 15, SIN_N = 18,
  SF_L = 0, SF_LO = 6, SF_Y = 15, SF_N = 18
};

struct NlpProb {
  int N;
  double h;
  double *W;                 // [5][N] node values
  double *ws;                // WS_TOTAL * N doubles (see above)
  double *mu;                // [3][N]
  const double *partner;     // [2][N] or null
  double *lds;               // this wavefront's NLP_LDS_DOUBLES doubles of LDS (nlp_assemble: neighbour hand-over, record transposition)
};
// LDS of one wavefront: up [65][9] (what a node hands to the node before it; row 64 = the first node of the chunk processed before),
// then rec [64][19] (the reduced-node records of a chunk, odd stride: conflict-free both ways)
#define NLP_LDS_UP 0
#define NLP_LDS_REC (65 * 9)
#define NLP_REC_STRIDE 19
#define NLP_BCR_STRIDE 19                 // odd: a lane per node reads its record without 8-way bank conflicts
#define NLP_BCR_LDS_NODES 121
#define NLP_LDS_DOUBLES 2304              // >= 65 * 9 + 64 * NLP_REC_STRIDE (assembly) and >= NLP_BCR_LDS_NODES * NLP_BCR_STRIDE (cyclic reduction)
static_assert(64 * NLP_BCR_STRIDE >= 65 * 9 && NLP_REC_STRIDE == NLP_BCR_STRIDE, "nlp_assemble rec_lds: records of the chunks above the first lie behind the hand-over rows");
#define NLP_W(c, i) pb.W[(c) * pb.N + (i)]
#define NLP_P(plane, i) pb.ws[(plane) * pb.N + (i)]
#define NLP_MU(k, i) pb.mu[(k) * pb.N + (i)]
#define NLP_DW(c, i) NLP_P(WS_DW + (c), i)
#define NLP_RHS(c, i) NLP_P(WS_RHS + (c), i)
#define NLP_EL(k, i) NLP_P(WS_EL + (k), i)
#define NLP_UP(k, i) NLP_P(WS_UP + (k), i)
#define NLP_SIN(k, i) pb.ws[(size_t)WS_SIN * pb.N + (i) * SIN_N + (k)]
#define NLP_SF(k, i) pb.ws[(size_t)WS_SF * pb.N + (i) * SF_N + (k)]
#define NLP_DS(k, i) pb.ws[(size_t)WS_DS * pb.N + (i) * 3 + (k)]

// stores of one phase -> loads of the next phase by OTHER lanes of the same wavefront (one wavefront owns a problem)
__device__ __forceinline__ void nlp_phase_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }

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
// Node-parallel (lane = node, chunks of 64) + wave reductions.  All results are wave-uniform.
__device__ double nlp_merit(const NlpProb &pb, const NlpScen &s, const double *__restrict__ sc, int lane, double a, double rho,
                            double mub, double *cost_ref_out, double *feas_out) {
  const int N = pb.N;
  double val = 0.0, bar = 0.0, cref = 0.0, feas = 0.0, phi2max = 0.0;
  int outside = 0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    if (i < N) {
      // (every value the node needs is requested up front, from clamped indices where a neighbour does not exist: a load inside a
      // branch is a memory round trip of its own -- the compiler does not speculate loads -- and a Newton step had forty of those)
      const int im = i >= 1 ? i - 1 : 0;
      double w[NLP_NV], wp[3], dwl[NLP_NV], dwp[3], muv[3];
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) { w[c] = NLP_W(c, i); dwl[c] = NLP_DW(c, i); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { wp[c] = NLP_W(c, im); dwp[c] = NLP_DW(c, im); muv[c] = NLP_MU(c, i); }
      if (a != 0.0) {
#pragma unroll
        for (int c = 0; c < NLP_NV; ++c) w[c] += a * dwl[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) wp[c] += a * dwp[c];
      }
      if (i < 1) { wp[0] = 0.0; wp[1] = 0.0; wp[2] = 0.0; }
      double prod = 1.0;
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) {
        if (nlp_fixed(i, N, c)) continue;
        if (s.lo[c] > -1e299) { const double sl = w[c] - s.lo[c]; if (!(sl > 0.0)) outside = 1; prod *= (sl > 0.0 ? sl : 1.0); }
        if (s.hi[c] < 1e299) { const double su = s.hi[c] - w[c]; if (!(su > 0.0)) outside = 1; prod *= (su > 0.0 ? su : 1.0); }
      }
      bar += log(prod);               // one log per node: at most ten slacks in [1e-12, 1e3], their product stays in range
      const double dv = w[4] - s.vsp;
      double obj = s.skv * dv * dv + s.skphi * w[3] * w[3];
      phi2max = fmax(phi2max, w[3] * w[3]);
      cref += obj;
      nlp_exp_terms(s, sc, pb.partner, i, N, w[0], w[1], obj, cref, nullptr, nullptr, nullptr, nullptr, nullptr);
      val += obj;
      if (i >= 1) {
        double c3[3];
        nlp_constraint(s, pb.h, wp, w, c3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double cm = c3[k] + muv[k];
          val += rho * cm * cm;
          feas = fmax(feas, fabs(c3[k]));
        }
      }
    }
  }
  val = wave_sum(val); bar = wave_sum(bar);
  cref = wave_sum(cref);
  if (s.sbank > 0.0) {                       // CostBank max mode (src/d2d/opty_utils.py:68-82): obj_scale * kbank * max_i phi_i^2
    const double bk = s.sbank * wave_max(phi2max);
    val += bk; cref += bk;
  }
  if (cost_ref_out) *cost_ref_out = cref;
  if (feas_out) *feas_out = wave_max(feas);
  const bool any_out = __builtin_amdgcn_ballot_w64(outside != 0) != 0ull;
  if (any_out || !(fabs(val) <= 1.79e308)) return INFINITY;
  return val - mub * bar;
}

// 1/sqrt(v), v > 0 (normal range): hardware estimate + two Newton steps (the pivots of the serial recursion are a dependent
// chain -- the library's sqrt followed by a division is four times as long)
__device__ __forceinline__ double nlp_rsqrt(double v) {
  double y = __builtin_amdgcn_rsq(v);
  const double hv = 0.5 * v;
  y = fma(y, fma(-hv * y, y, 0.5), y);
  y = fma(y, fma(-hv * y, y, 0.5), y);
  return y;
}

// Assembly + local elimination (node-parallel, one pass, the 5x5 blocks never leave the registers): half gradient g, half Hessian
// blocks D (diagonal) and E (i, i-1) of the barrier-AL Lagrangian incl. the constraint curvature, barrier diagonal, damping,
// right-hand side; then the elimination of (phi, v) described below.  Returns the barrier KKT error of the inner problem
// (wave-uniform; it does not depend on the damping); *pd_out = false if a 2x2 pivot is not positive.
// node with the largest |phi| at the current iterate (first on ties), wave-uniform
__device__ int nlp_bank_argmax(const NlpProb &pb, int lane) {
  const int N = pb.N;
  double best = -1.0;
  int istar = 0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    const double a = i < N ? fabs(NLP_W(3, i)) : -1.0;
    const double m = wave_max(a);
    if (m > best) { best = m; istar = i0 + (int)__builtin_ctzll(__ballot(a == m)); }
  }
  return __builtin_amdgcn_readfirstlane(istar);
}

// imax (CostBank max mode, s.sbank > 0): the node whose bank angle is the largest at the current iterate -- the reference's
// cost_grad is one-hot there (2 obj_scale kbank phi_imax), the step's model carries the term sbank * phi_imax^2 on that node
// alone (the maximiser frozen for the step; the merit function of the line search is the true max)
__device__ double nlp_assemble(const NlpProb &pb, const NlpScen &s, const double *__restrict__ sc, int lane, double rho, double mub,
                               double lam, bool *pd_out, int imax, bool rec_lds) {
  const int N = pb.N;
  const double h = pb.h, ih = 1.0 / h;
  double err = 0.0;
  int bad = 0;
  // The chunks of 64 nodes are visited from the LAST to the first: node i needs what node i+1 hands to it (the hand-over of the
  // elimination, below), and for the last lane of a chunk that is the first node of the chunk visited before.
  double *lds_up = pb.lds + NLP_LDS_UP, *lds_rec = pb.lds + NLP_LDS_REC;
  for (int i0 = ((N - 1) / 64) * 64; i0 >= 0; i0 -= 64) {
    const int i = i0 + lane;
    const bool live = i < N;
    double sinv[SIN_N], upv[9];
    if (live) {
    const bool has_next = i + 1 < N;
    double rhsv[NLP_NV];
    double wp[3] = {0, 0, 0}, wc[NLP_NV], wn[NLP_NV] = {0, 0, 0, 0, 1};
    double cc[3] = {0, 0, 0}, cn[3] = {0, 0, 0};     // (c + mu) of the constraint that ends at this node / at the next one
    // (all loads of the node up front, neighbours from clamped indices: see nlp_merit)
    const int im = i >= 1 ? i - 1 : 0, ip = has_next ? i + 1 : i;
    double zlv[NLP_NV], zuv[NLP_NV], mu_c[3], mu_n[3];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) { wc[c] = NLP_W(c, i); wn[c] = NLP_W(c, ip); zlv[c] = NLP_P(WS_ZL + c, i); zuv[c] = NLP_P(WS_ZU + c, i); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { wp[c] = NLP_W(c, im); mu_c[c] = NLP_MU(c, i); mu_n[c] = NLP_MU(c, ip); }
    if (i < 1) { wp[0] = 0.0; wp[1] = 0.0; wp[2] = 0.0; }
    if (has_next) {
      nlp_constraint(s, h, wc, wn, cn);
#pragma unroll
      for (int k = 0; k < 3; ++k) cn[k] += mu_n[k];
    } else {
      wn[0] = 0.0; wn[1] = 0.0; wn[2] = 0.0; wn[3] = 0.0; wn[4] = 1.0;
    }
    double g[NLP_NV] = {0, 0, 0, 0, 0};
    double D[NLP_NV][NLP_NV];
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a)
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) D[a][c] = 0.0;
    D[4][4] += s.skv; g[4] += s.skv * (wc[4] - s.vsp);
    D[3][3] += s.skphi; g[3] += s.skphi * wc[3];
    if (i == imax) { D[3][3] += s.sbank; g[3] += s.sbank * wc[3]; }
    {
      double obj = 0.0, cref = 0.0;
      nlp_exp_terms(s, sc, pb.partner, i, N, wc[0], wc[1], obj, cref, &g[0], &g[1], &D[0][0], &D[0][1], &D[1][1]);
      D[1][0] = D[0][1];
    }
    double E[NLP_NV][3];                        // block (i, i-1): only the (x, y, psi) columns of node i-1 are non-zero
#pragma unroll
    for (int a = 0; a < NLP_NV; ++a) { E[a][0] = 0.0; E[a][1] = 0.0; E[a][2] = 0.0; }
    if (i >= 1) {
      double sp, cp;
      sincos(wc[2], &sp, &cp);
      const double tp = tan(wc[3]), vi = wc[4], sec2 = 1.0 + tp * tp;
      cc[0] = (wc[0] - wp[0]) * ih - vi * cp + s.wx + mu_c[0];
      cc[1] = (wc[1] - wp[1]) * ih - vi * sp + s.wy + mu_c[1];
      cc[2] = (wc[2] - wp[2]) * ih - FIT_G / vi * tp + mu_c[2];
      double Ac[3][NLP_NV];                     // Jacobian of the constraint wrt this node
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
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      const bool fx = nlp_fixed(i, N, c);
      double sig = 0.0, r = -2.0 * g[c], st = 2.0 * g[c];
      if (!fx && s.lo[c] > -1e299) {
        const double sl = wc[c] - s.lo[c], z = zlv[c], isl = 1.0 / sl;
        sig += z * isl; r += mub * isl; st -= z;
        err = fmax(err, fabs(z * sl - mub));
      }
      if (!fx && s.hi[c] < 1e299) {
        const double su = s.hi[c] - wc[c], z = zuv[c], isu = 1.0 / su;
        sig += z * isu; r -= mub * isu; st += z;
        err = fmax(err, fabs(z * su - mub));
      }
      if (!fx) err = fmax(err, fabs(st));
      rhsv[c] = fx ? 0.0 : r;
      NLP_RHS(c, i) = rhsv[c];
      D[c][c] += 0.5 * sig;
      if (!fx) D[c][c] += lam * fmax(fabs(D[c][c]), 1e-12);
      if (fx) {
#pragma unroll
        for (int a = 0; a < NLP_NV; ++a) { D[c][a] = 0.0; D[a][c] = 0.0; }
        D[c][c] = 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) E[c][k] = 0.0;
      }
    }
    if (i == 1) {                               // columns of E that belong to the fixed variables of node 0
#pragma unroll
      for (int a = 0; a < NLP_NV; ++a) { E[a][0] = 0.0; E[a][1] = 0.0; E[a][2] = 0.0; }
    }
    // ---- elimination of (phi, v) of this node: P = LP LP^T, Qt = LP^-1 M[pv, s_i], Rt = LP^-1 M[pv, s_{i-1}], tt = LP^-1 rhs_pv / 2
    double P00 = D[3][3], v11;
    if (!(P00 > 0.0)) { bad = 1; P00 = 1.0; }
    const double i00 = nlp_rsqrt(P00), l10 = D[4][3] * i00;
    v11 = D[4][4] - l10 * l10;
    if (!(v11 > 0.0)) { bad = 1; v11 = 1.0; }
    const double i11 = nlp_rsqrt(v11);
    NLP_EL(EL_LP + 0, i) = i00; NLP_EL(EL_LP + 1, i) = l10; NLP_EL(EL_LP + 2, i) = i11;
    double Qt[2][3], Rt[2][3], tt[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Qt[0][k] = D[3][k] * i00; Qt[1][k] = (D[4][k] - l10 * Qt[0][k]) * i11;
      Rt[0][k] = E[3][k] * i00; Rt[1][k] = (E[4][k] - l10 * Rt[0][k]) * i11;
    }
    tt[0] = 0.5 * rhsv[3] * i00;
    tt[1] = (0.5 * rhsv[4] - l10 * tt[0]) * i11;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { NLP_EL(EL_Q + j * 3 + k, i) = Qt[j][k]; NLP_EL(EL_R + j * 3 + k, i) = Rt[j][k]; }
      NLP_EL(EL_T + j, i) = tt[j];
    }
    // what node i-1 gets from this node's elimination, and this node's own part of the reduced node
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int c = 0; c <= a; ++c) {
        upv[UP_RR + a * (a + 1) / 2 + c] = Rt[0][a] * Rt[0][c] + Rt[1][a] * Rt[1][c];
        sinv[SIN_D + a * (a + 1) / 2 + c] = D[a][c] - (Qt[0][a] * Qt[0][c] + Qt[1][a] * Qt[1][c]);
      }
      upv[UP_RT + a] = Rt[0][a] * tt[0] + Rt[1][a] * tt[1];
#pragma unroll
      for (int k = 0; k < 3; ++k) sinv[SIN_E + a * 3 + k] = E[a][k] - (Qt[0][a] * Rt[0][k] + Qt[1][a] * Rt[1][k]);
      sinv[SIN_T + a] = 0.5 * rhsv[a] - (Qt[0][a] * tt[0] + Qt[1][a] * tt[1]);
    }
    }
    // ---- hand-over between neighbours, D'_i -= (Rt^T Rt)_{i+1}, t'_i -= (Rt^T tt)_{i+1}, through the LDS (the records used to make
    // a round trip through HBM for it), then the node-major records of the chunk transposed through the LDS so that they leave as
    // full 512-byte rows: 8-byte stores at a stride of 144 bytes reached the memory as partial sectors (WRITE_SIZE 1.6 x algorithmic)
    if (live) {
#pragma unroll
      for (int k = 0; k < 9; ++k) lds_up[lane * 9 + k] = upv[k];
    }
    nlp_phase_sync();
    if (live && i + 1 < N) {
      const double *nb = lds_up + (lane + 1) * 9;           // lane 63: row 64 = node i0 + 64, left there by the chunk visited before
#pragma unroll
      for (int k = 0; k < 6; ++k) sinv[SIN_D + k] -= nb[UP_RR + k];
#pragma unroll
      for (int a = 0; a < 3; ++a) sinv[SIN_T + a] -= nb[UP_RT + a];
    }
    if (rec_lds) {
      // The cyclic reduction that follows works in this wave's LDS block on records [node][NLP_BCR_STRIDE] (N <= NLP_BCR_LDS_NODES):
      // the records go there directly -- no store to the workspace, no load back (three memory round trips of a Newton step).
      // Chunks above the first lie behind the hand-over rows (64 * 19 >= 65 * 9); the first chunk's own records overwrite the
      // hand-over rows once every neighbour has read them.
      nlp_phase_sync();
      if (lane == 0 && i0 > 0) {                            // this chunk's first node for the chunk below
#pragma unroll
        for (int k = 0; k < 9; ++k) lds_up[64 * 9 + k] = upv[k];
      }
      if (live) {
#pragma unroll
        for (int k = 0; k < SIN_N; ++k) pb.lds[(size_t)i * NLP_BCR_STRIDE + k] = sinv[k];
      }
      nlp_phase_sync();
      continue;
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < SIN_N; ++k) lds_rec[lane * NLP_REC_STRIDE + k] = sinv[k];
    }
    nlp_phase_sync();
    if (lane == 0) {                                        // this chunk's first node for the chunk below
#pragma unroll
      for (int k = 0; k < 9; ++k) lds_up[64 * 9 + k] = upv[k];
    }
    {
      const int cnt = (N - i0 < 64 ? N - i0 : 64) * SIN_N;
      double *dst = pb.ws + (size_t)WS_SIN * N + (size_t)i0 * SIN_N;
      for (int e = lane; e < cnt; e += 64) {
        const int node = e / SIN_N;
        dst[e] = lds_rec[node * NLP_REC_STRIDE + (e - node * SIN_N)];
      }
    }
    nlp_phase_sync();
  }
  *pd_out = __builtin_amdgcn_ballot_w64(bad != 0) == 0ull;
  return wave_max(err);
}

// The Newton system  M dw = rhs / 2  (M: block tridiagonal, 5x5 blocks D_i + damping, sub-diagonal blocks E_i whose only
// non-zero columns are x, y, psi of node i-1).  phi_i and v_i are coupled to (x, y, psi) of nodes i and i-1 only and to no
// other node's phi, v: they are eliminated NODE BY NODE IN PARALLEL (2x2 Cholesky + Schur complement), which leaves a block
// tridiagonal system with 3x3 blocks in (x, y, psi) for the serial recursion -- 80 instead of 280 fp64 operations per node on
// the serial chain.
// The elimination itself is the tail of nlp_assemble (the blocks are in registers there); what remains is the hand-over between
// neighbours (lane = node):  D'_i -= (Rt^T Rt)_{i+1},  t'_i -= (Rt^T tt)_{i+1}.

// The two serial recursions, TWISTED: the block Cholesky runs from both ends of the horizon towards the middle node m = N/2 at
// the same time -- lanes 0..31 eliminate nodes 0, 1, .. m-1, lanes 32..63 nodes N-1, N-2, .. m+1, one instruction stream (inside a
// half every lane computes and stores the same values) -- and the back substitution runs from the middle outwards the same way:
// the dependent chain is N/2 nodes long instead of N.
//   step of a half:  Lo = C Lp^-T,  S = D' - Lo Lo^T,  L = chol(S),  y = L^-1 (t' - Lo yp)
//   C = block (this node, node eliminated before it): E'_i for the lower half, E'_{i+1}^T for the upper half
//   middle:          S_m = D'_m - Lo Lo^T (from m-1) - Uo Uo^T (from m+1),  y_m = L_m^-1 (t'_m - Lo y_{m-1} - Uo y_{m+1})
//   back:            d_m = L_m^-T y_m;  d_i = L_i^-T (y_i - Lon^T dn)  with Lon, dn of the node next to i on the middle's side
// Own functions, not inlined (inside the solver's loop nest the compiler spilled a dozen scalar registers around every node);
// global address space stated explicitly (through a generic pointer the loads would be FLAT ones, whose waits also wait for the
// stores); a ring of four node buffers with the loop unrolled by four (the node three steps ahead is requested before a node is
// processed: one node's arithmetic is shorter than an L2 round trip, and no buffer is ever copied); every lane of a half stores the
// same values to the same address (a store under `if (lane == 0)` sits behind a branch, and the wait for the next loads then
// also waits for the stores: vmcnt counts in order and the compiler cannot count through a branch); a full wait before each
// loop so that the loop-head wait is the back edge's.
struct NlpNodeIn {
  double v[SIN_N];
};

__device__ __forceinline__ double nlp_other_half(double v) {              // the value the lane 32 places away holds
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(((threadIdx.x ^ 32) & 63) << 2, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(((threadIdx.x ^ 32) & 63) << 2, __double2loint(v)));
}

// one elimination step from the reduced node `in` (D' 6, C 9 row-major, t' 3) and the factor (Lp, yp) of the node before it
__device__ __forceinline__ void nlp_block_lo(const double (&C)[3][3], const double (&Lp)[6], double (&Lo)[3][3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = C[a][c];
#pragma unroll
      for (int k = 0; k < c; ++k) v -= Lp[c * (c + 1) / 2 + k] * Lo[a][k];
      Lo[a][c] = v * Lp[c * (c + 1) / 2 + c];             // (the diagonal slots hold 1 / L_cc; first node of a half: Lp = 0 -> Lo = 0)
    }
  }
}
__device__ __forceinline__ bool nlp_block_chol(const double (&S)[6], const double (&t)[3], double (&L)[6], double (&y)[3]) {
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int c = 0; c <= a; ++c) {
      double v = S[a * (a + 1) / 2 + c];
#pragma unroll
      for (int k = 0; k < c; ++k) v -= L[a * (a + 1) / 2 + k] * L[c * (c + 1) / 2 + k];
      if (a == c) {
        if (!(v > 0.0)) { ok = false; v = 1.0; }
        L[a * (a + 1) / 2 + a] = nlp_rsqrt(v);             // the factor keeps the RECIPROCAL diagonal: every later use divides by it
      } else {
        L[a * (a + 1) / 2 + c] = v * L[c * (c + 1) / 2 + c];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double v = t[a];
#pragma unroll
    for (int k = 0; k < a; ++k) v -= L[a * (a + 1) / 2 + k] * y[k];
    y[a] = v * L[a * (a + 1) / 2 + a];
  }
  return ok;
}


typedef __attribute__((address_space(1))) double gdouble;
// ---- the reduced system by BLOCK CYCLIC REDUCTION (round 3) ----------------------------------------------------------------
// The block-tridiagonal system of the reduced nodes (3x3 blocks D'_i, E'_i = block (i, i-1), right-hand sides t'_i) is symmetric
// positive definite when the step is usable, so any elimination order is a Cholesky factorisation of it.  Odd-even order: at the
// level of stride s the nodes j = s (2m + 1) are eliminated ALL AT ONCE, one lane per node, against their neighbours a = j - s and
// b = j + s, which survive with
//     D_a -= E_j^T P_j,  t_a -= E_j^T r_j;      D_b -= E_b Q_j,  t_b -= E_b r_j,  E_b <- -E_b P_j  (now block (b, a))
//     P_j = D_j^-1 E_j,  Q_j = D_j^-1 E_b^T,  r_j = D_j^-1 t_j
// and after ceil(log2 N) levels node 0 is alone; the unknowns come back level by level, x_j = r_j - P_j x_a - Q_j x_b.  The
// twisted recursion it replaces is a dependent chain of N/2 block pivots (88 k + 27 k of the 200 k cycles of a Newton step at
// N = 121); this is 7 levels of independent 3x3 work.  A node has one left and one right producer per level: the producers apply
// their update to the left neighbour first, then (after a wave-level sync) to the right one, so no two lanes touch a record at once.
// Up to NLP_BCR_LDS_NODES nodes the records live in the wave's LDS (stride 19), P_j and Q_j take the place of the eliminated
// node's record and x_j that of P_j; longer horizons run the same code on the records in global memory.
__device__ __forceinline__ void nlp_solve3(const double (&L)[6], const double b0, const double b1, const double b2, double &x0, double &x1, double &x2) {
  // L L^T x = b with the factor of nlp_block_chol (reciprocal diagonal)
  const double y0 = b0 * L[0];
  const double y1 = (b1 - L[1] * y0) * L[2];
  const double y2 = (b2 - L[3] * y0 - L[4] * y1) * L[5];
  x2 = y2 * L[5];
  x1 = (y1 - L[4] * x2) * L[2];
  x0 = (y0 - L[1] * x1 - L[3] * x2) * L[0];
}

// RT: pointer type of the records -- LDS (address space 3) or global (1): stated, so that the accesses are ds_ / global_ instructions
// (through generic pointers they were flat_ ones: 51 k cycles per Newton step instead of the 30 k of this version)
typedef __attribute__((address_space(3))) double ldouble;
template <typename RT, bool in_lds>
__device__ __attribute__((noinline)) bool nlp_bcr_t(double *sin_generic, double *sf_generic, double *ds_generic, double *lds_generic, int N,
                                                    bool preloaded) {
  const int lane = threadIdx.x & 63;
  gdouble *sin_g = (gdouble *)sin_generic;
  gdouble *ds_g = (gdouble *)ds_generic;
  RT *lds = (RT *)lds_generic;
  RT *rec = in_lds ? (RT *)lds_generic : (RT *)sin_generic;      // records [node][rs]: D' (6), E' (9), t' (3)
  constexpr int rs = in_lds ? NLP_BCR_STRIDE : SIN_N;
  RT *pq = in_lds ? (RT *)lds_generic : (RT *)sf_generic;        // P (9), Q (9) of an eliminated node [node][ps]
  constexpr int ps = in_lds ? NLP_BCR_STRIDE : SF_N;
  if (in_lds && !preloaded) {
    // (batches of 12 rows of 64: every load of a batch is in flight before its first LDS store -- as a plain loop the compiler waited
    // for each load before storing it, 34 memory round trips in a row at 121 nodes, the longest serial stretch of a Newton step)
    const int total = N * SIN_N;
    for (int e0 = 0; e0 < total; e0 += 12 * 64) {
      double v[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) { const int e = e0 + q * 64 + lane; v[q] = e < total ? (double)sin_g[e] : 0.0; }
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int e = e0 + q * 64 + lane;
        const int node = e / SIN_N;
        if (e < total) lds[node * NLP_BCR_STRIDE + (e - node * SIN_N)] = v[q];
      }
    }
    nlp_phase_sync();
  }
  int bad = 0;
  int s = 1;
  for (; s < N; s <<= 1) {
    const int n_el = (N - 1 - s) / (2 * s) + 1;       // eliminated nodes of this level: j = s (2m + 1) < N
    for (int m0 = 0; m0 < n_el; m0 += 64) {
      const int m = m0 + lane;
      const bool act = m < n_el;
      const int j = act ? s * (2 * m + 1) : s, a = j - s, b = j + s;
      const bool has_b = act && b < N;
      double D[6], E[3][3], t[3], Eb[3][3], L[6], y[3], P[3][3], Q[3][3], r[3];
      const RT *pj = rec + (size_t)j * rs, *pb = rec + (size_t)(has_b ? b : j) * rs;
#pragma unroll
      for (int q = 0; q < 6; ++q) D[q] = pj[SIN_D + q];
#pragma unroll
      for (int q = 0; q < 9; ++q) { E[q / 3][q % 3] = pj[SIN_E + q]; Eb[q / 3][q % 3] = has_b ? pb[SIN_E + q] : 0.0; }
#pragma unroll
      for (int q = 0; q < 3; ++q) t[q] = pj[SIN_T + q];
      const bool ok = nlp_block_chol(D, t, L, y);
      if (act && !ok) bad = 1;
      nlp_solve3(L, t[0], t[1], t[2], r[0], r[1], r[2]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        nlp_solve3(L, E[0][k], E[1][k], E[2][k], P[0][k], P[1][k], P[2][k]);           // column k of E_j
        nlp_solve3(L, Eb[k][0], Eb[k][1], Eb[k][2], Q[0][k], Q[1][k], Q[2][k]);        // column k of E_b^T = row k of E_b
      }
      // the left neighbour: D_a -= E_j^T P_j, t_a -= E_j^T r_j
      if (act) {
        RT *pa = rec + (size_t)a * rs;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
          for (int v = 0; v <= u; ++v)
            pa[SIN_D + u * (u + 1) / 2 + v] -= E[0][u] * P[0][v] + E[1][u] * P[1][v] + E[2][u] * P[2][v];
          pa[SIN_T + u] -= E[0][u] * r[0] + E[1][u] * r[1] + E[2][u] * r[2];
        }
      }
      nlp_phase_sync();
      // the right neighbour: D_b -= E_b Q_j, t_b -= E_b r_j, E_b <- -E_b P_j
      if (has_b) {
        RT *pbw = rec + (size_t)b * rs;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
          for (int v = 0; v <= u; ++v)
            pbw[SIN_D + u * (u + 1) / 2 + v] -= Eb[u][0] * Q[0][v] + Eb[u][1] * Q[1][v] + Eb[u][2] * Q[2][v];
          pbw[SIN_T + u] -= Eb[u][0] * r[0] + Eb[u][1] * r[1] + Eb[u][2] * r[2];
#pragma unroll
          for (int v = 0; v < 3; ++v) pbw[SIN_E + u * 3 + v] = -(Eb[u][0] * P[0][v] + Eb[u][1] * P[1][v] + Eb[u][2] * P[2][v]);
        }
      }
      // what the way back needs of node j: P_j, Q_j (in place of its record / in the factor's workspace), r_j (in the step's slot)
      if (act) {
        RT *o = pq + (size_t)j * ps;
#pragma unroll
        for (int q = 0; q < 9; ++q) { o[q] = P[q / 3][q % 3]; o[9 + q] = Q[q / 3][q % 3]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) ds_g[(size_t)j * 3 + q] = r[q];
      }
      nlp_phase_sync();
    }
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0ull) return false;
  // node 0 alone
  {
    double D[6], t[3], L[6], y[3], x[3];
#pragma unroll
    for (int q = 0; q < 6; ++q) D[q] = rec[SIN_D + q];
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = rec[SIN_T + q];
    if (!nlp_block_chol(D, t, L, y)) return false;
    nlp_solve3(L, t[0], t[1], t[2], x[0], x[1], x[2]);
    nlp_phase_sync();
    // (every lane stores the same values: no branch between the loads above and these stores)
#pragma unroll
    for (int q = 0; q < 3; ++q) { ds_g[q] = x[q]; if (in_lds) lds[q] = x[q]; }
    nlp_phase_sync();
  }
  // the way back: the unknowns of a level from those of the coarser ones (in the LDS path x_j takes the place of P_j)
  const RT *xs = in_lds ? (const RT *)lds_generic : (const RT *)ds_generic;
  constexpr int xst = in_lds ? NLP_BCR_STRIDE : 3;
  // LDS path (one chunk per level: N <= 121): r_j sits in global memory -- the only global read of a level, a round trip of its own
  // on the serial chain of seven levels.  The lane that will need r_j of the NEXT level is known (node s/2 (2 lane + 1)): its three
  // values are requested one level ahead and arrive behind the current level's arithmetic.
  double rn[3] = {0.0, 0.0, 0.0};
  auto request_r = [&](int sl) {
    if (!in_lds || sl < 1) return;
    const int n_l = (N - 1 - sl) / (2 * sl) + 1;
    const int jl = lane < n_l ? sl * (2 * lane + 1) : sl;
#pragma unroll
    for (int u = 0; u < 3; ++u) rn[u] = ds_g[(size_t)jl * 3 + u];
  };
  request_r(s >> 1);
  for (s >>= 1; s >= 1; s >>= 1) {
    const int n_el = (N - 1 - s) / (2 * s) + 1;
    for (int m0 = 0; m0 < n_el; m0 += 64) {
      const int m = m0 + lane;
      const bool act = m < n_el;
      const int j = act ? s * (2 * m + 1) : s, a = j - s, b = j + s;
      const bool has_b = act && b < N;
      double x[3], rj[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) rj[u] = in_lds ? rn[u] : (act ? (double)ds_g[(size_t)j * 3 + u] : 0.0);
      request_r(s >> 1);
      if (act) {
        const RT *o = pq + (size_t)j * ps;
        const RT *xa = xs + (size_t)a * xst, *xb = xs + (size_t)(has_b ? b : a) * xst;
        const double a0 = xa[0], a1 = xa[1], a2 = xa[2];
        const double b0 = has_b ? xb[0] : 0.0, b1 = has_b ? xb[1] : 0.0, b2 = has_b ? xb[2] : 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u)
          x[u] = rj[u] - (o[u * 3] * a0 + o[u * 3 + 1] * a1 + o[u * 3 + 2] * a2)
                 - (o[9 + u * 3] * b0 + o[9 + u * 3 + 1] * b1 + o[9 + u * 3 + 2] * b2);
      }
      nlp_phase_sync();                       // (every P_j of the chunk has been read before an x_j overwrites one)
      if (act) {
#pragma unroll
        for (int u = 0; u < 3; ++u) { ds_g[(size_t)j * 3 + u] = x[u]; if (in_lds) lds[(size_t)j * NLP_BCR_STRIDE + u] = x[u]; }
      }
      nlp_phase_sync();
    }
  }
  return true;
}
// preloaded: the records are in the LDS block already (nlp_assemble with rec_lds; only for N <= NLP_BCR_LDS_NODES)
__device__ __forceinline__ bool nlp_bcr(double *sin_g, double *sf_g, double *ds_g, double *lds, int N, bool preloaded) {
  if (N <= NLP_BCR_LDS_NODES) return nlp_bcr_t<ldouble, true>(sin_g, sf_g, ds_g, lds, N, preloaded);
  return nlp_bcr_t<gdouble, false>(sin_g, sf_g, ds_g, lds, N, false);
}

// mid: 9 doubles of scratch for the second coupling block of the middle node (its coupling to node m+1)
__device__ __attribute__((noinline)) bool nlp_factor(double *sin_generic, double *sf_generic, double *mid_generic, int N) {
  const gdouble *sin = (const gdouble *)sin_generic;
  gdouble *sf = (gdouble *)sf_generic;
  gdouble *mid = (gdouble *)mid_generic;
  const bool up = (threadIdx.x & 32) != 0;                   // upper half: nodes N-1 down to m+1
  const int m = N >> 1, nlo = m, cnt = up ? N - 1 - m : m;   // (cnt >= 1 for N >= 3; the upper half has nlo or nlo - 1 nodes)
  // inputs of step k of this lane's half: D', t' of its node i; C from E' of node i (lower) / E'^T of node i+1 (upper)
  auto load = [&](int k, NlpNodeIn &n) {
    const int kk = k < cnt ? k : cnt - 1;
    const int i = up ? N - 1 - kk : kk;
    const int ie = up ? (i + 1 < N ? i + 1 : i) : i;
    const gdouble *pd = sin + (long)i * SIN_N, *pe = sin + (long)ie * SIN_N;
#pragma unroll
    for (int q = 0; q < 6; ++q) n.v[SIN_D + q] = pd[SIN_D + q];
#pragma unroll
    for (int q = 0; q < 9; ++q) n.v[SIN_E + q] = pe[SIN_E + q];
#pragma unroll
    for (int q = 0; q < 3; ++q) n.v[SIN_T + q] = pd[SIN_T + q];
  };
  auto coupling = [&](const NlpNodeIn &n, bool zero, double (&C)[3][3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v = up ? n.v[SIN_E + c * 3 + a] : n.v[SIN_E + a * 3 + c];
        C[a][c] = zero ? 0.0 : v;
      }
  };
  double Lp[6], yp[3];                         // factor of the node eliminated before (reciprocal diagonal), its y
#pragma unroll
  for (int k = 0; k < 6; ++k) Lp[k] = 0.0;
#pragma unroll
  for (int c = 0; c < 3; ++c) yp[c] = 0.0;
  bool ok = true;
  NlpNodeIn buf[4];
  load(0, buf[0]); load(1, buf[1]); load(2, buf[2]);
  __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
  for (int k0 = 0; k0 < nlo; k0 += 4) {
#pragma unroll
   for (int u = 0; u < 4; ++u) {
    const int k = k0 + u;
    if (k >= nlo) break;
    load(k + 3, buf[(u + 3) & 3]);
    const NlpNodeIn &cur = buf[u];
    if (k < cnt) {                                           // (the upper half may be one node shorter)
      const int i = up ? N - 1 - k : k;
      double C[3][3], Lo[3][3], S[6], t[3], L[6], y[3];
      coupling(cur, k == 0, C);                              // (first node of a half: nothing before it)
      nlp_block_lo(C, Lp, Lo);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c <= a; ++c)
          S[a * (a + 1) / 2 + c] = cur.v[SIN_D + a * (a + 1) / 2 + c] - (Lo[a][0] * Lo[c][0] + Lo[a][1] * Lo[c][1] + Lo[a][2] * Lo[c][2]);
        t[a] = cur.v[SIN_T + a] - (Lo[a][0] * yp[0] + Lo[a][1] * yp[1] + Lo[a][2] * yp[2]);
      }
      ok = nlp_block_chol(S, t, L, y) && ok;
      gdouble *o = sf + (long)i * SF_N;
#pragma unroll
      for (int q = 0; q < 6; ++q) o[SF_L + q] = L[q];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[SF_LO + a * 3 + c] = Lo[a][c];
#pragma unroll
      for (int a = 0; a < 3; ++a) o[SF_Y + a] = y[a];
#pragma unroll
      for (int q = 0; q < 6; ++q) Lp[q] = L[q];
#pragma unroll
      for (int a = 0; a < 3; ++a) yp[a] = y[a];
    }
   }
  }
  // ---- the middle node: both halves hand in their Schur complement
  {
    NlpNodeIn md;
    const int ie = up ? m + 1 : m;
    const gdouble *pd = sin + (long)m * SIN_N, *pe = sin + (long)ie * SIN_N;
#pragma unroll
    for (int q = 0; q < 6; ++q) md.v[SIN_D + q] = pd[SIN_D + q];
#pragma unroll
    for (int q = 0; q < 9; ++q) md.v[SIN_E + q] = pe[SIN_E + q];
#pragma unroll
    for (int q = 0; q < 3; ++q) md.v[SIN_T + q] = pd[SIN_T + q];
    double C[3][3], Lo[3][3], S[6], t[3], L[6], y[3];
    coupling(md, false, C);
    nlp_block_lo(C, Lp, Lo);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int c = 0; c <= a; ++c) {
        const double p = Lo[a][0] * Lo[c][0] + Lo[a][1] * Lo[c][1] + Lo[a][2] * Lo[c][2];
        S[a * (a + 1) / 2 + c] = md.v[SIN_D + a * (a + 1) / 2 + c] - (p + nlp_other_half(p));
      }
      const double q = Lo[a][0] * yp[0] + Lo[a][1] * yp[1] + Lo[a][2] * yp[2];
      t[a] = md.v[SIN_T + a] - (q + nlp_other_half(q));
    }
    ok = nlp_block_chol(S, t, L, y) && ok;
    gdouble *o = sf + (long)m * SF_N;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[SF_L + q] = L[q];
#pragma unroll
    for (int a = 0; a < 3; ++a) o[SF_Y + a] = y[a];
    gdouble *oc = up ? mid : o + SF_LO;                      // coupling to m-1 with the node, coupling to m+1 in the scratch
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) oc[a * 3 + c] = Lo[a][c];
  }
  return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}

// Serial recursion 2, from the middle outwards in both halves: ds_m = L_m^-T y_m;  ds_i = L_i^-T (y_i - Lon^T dn).
__device__ __attribute__((noinline)) void nlp_backsolve(double *sf_generic, double *mid_generic, double *ds_generic, int N) {
  const gdouble *sf = (const gdouble *)sf_generic;
  const gdouble *mid = (const gdouble *)mid_generic;
  gdouble *dsv = (gdouble *)ds_generic;
  const bool up = (threadIdx.x & 32) != 0;
  const int m = N >> 1, nlo = m, cnt = up ? N - 1 - m : m;
  auto node_of = [&](int k) { const int kk = k < cnt ? k : cnt - 1; return up ? m + 1 + kk : m - 1 - kk; };
  auto load = [&](int k, NlpNodeIn &n) {
    const gdouble *p = sf + (long)node_of(k) * SF_N;
#pragma unroll
    for (int q = 0; q < SF_N; ++q) n.v[q] = p[q];
  };
  auto subst = [&](const double *v, const double (&t)[3], double (&ds)[3]) {       // ds = L^-T t (v: the node's factor record)
#pragma unroll
    for (int a = 2; a >= 0; --a) {
      double x = t[a];
#pragma unroll
      for (int k = a + 1; k < 3; ++k) x -= v[SF_L + k * (k + 1) / 2 + a] * ds[k];
      ds[a] = x * v[SF_L + a * (a + 1) / 2 + a];
    }
  };
  NlpNodeIn buf[4];
  load(0, buf[0]); load(1, buf[1]); load(2, buf[2]);
  double dn[3], Lon[3][3];
  {
    NlpNodeIn md;
    const gdouble *p = sf + (long)m * SF_N;
#pragma unroll
    for (int q = 0; q < SF_N; ++q) md.v[q] = p[q];
    const gdouble *pc = up ? mid : p + SF_LO;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) Lon[a][c] = pc[a * 3 + c];
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    double t[3] = {md.v[SF_Y], md.v[SF_Y + 1], md.v[SF_Y + 2]};
    subst(md.v, t, dn);
#pragma unroll
    for (int c = 0; c < 3; ++c) dsv[(long)m * 3 + c] = dn[c];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int k0 = 0; k0 < nlo; k0 += 4) {
#pragma unroll
   for (int u = 0; u < 4; ++u) {
    const int k = k0 + u;
    if (k >= nlo) break;
    load(k + 3, buf[(u + 3) & 3]);
    const NlpNodeIn &cur = buf[u];
    if (k < cnt) {
      const int i = up ? m + 1 + k : m - 1 - k;
      double t[3], ds[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) t[a] = cur.v[SF_Y + a] - (Lon[0][a] * dn[0] + Lon[1][a] * dn[1] + Lon[2][a] * dn[2]);
      subst(cur.v, t, ds);
      const bool fx = i == 0 || i == N - 1;                 // end conditions: fixed variables
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (fx) ds[c] = 0.0;
        dn[c] = ds[c];
        dsv[(long)i * 3 + c] = ds[c];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Lon[a][c] = cur.v[SF_LO + a * 3 + c];
    }
   }
  }
}

// Recovery of the eliminated (phi, v) steps + step statistics (node-parallel): d(phi, v)_i = LP^-T (tt - Qt ds_i - Rt ds_{i-1});
// directional derivative of the merit function, largest primal step (fraction to the boundary) and the dual steps' largest
// fraction.  Wave-uniform results.
__device__ void nlp_recover_stats(const NlpProb &pb, const NlpScen &s, int lane, double mub, double tau, double *dphi_out,
                                  double *amax_out, double *az_out) {
  const int N = pb.N;
  double dphi = 0.0, amax = 1.0, az = 1.0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    if (i >= N) continue;
    // (all loads of the node up front: see nlp_merit)
    const int im = i >= 1 ? i - 1 : 0;
    double dwv[NLP_NV], dsp[3], wv[NLP_NV], rhv[NLP_NV], zlv[NLP_NV], zuv[NLP_NV], elq[6], elr[6], ellp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { dwv[c] = NLP_DS(c, i); dsp[c] = NLP_DS(c, im); }
    if (i < 1) { dsp[0] = 0.0; dsp[1] = 0.0; dsp[2] = 0.0; }
    double z0 = NLP_EL(EL_T + 0, i), z1 = NLP_EL(EL_T + 1, i);
#pragma unroll
    for (int k = 0; k < 6; ++k) { elq[k] = NLP_EL(EL_Q + k, i); elr[k] = NLP_EL(EL_R + k, i); }
#pragma unroll
    for (int k = 0; k < 3; ++k) ellp[k] = NLP_EL(EL_LP + k, i);
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) { wv[c] = NLP_W(c, i); rhv[c] = NLP_RHS(c, i); zlv[c] = NLP_P(WS_ZL + c, i); zuv[c] = NLP_P(WS_ZU + c, i); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      z0 -= elq[k] * dwv[k] + elr[k] * dsp[k];
      z1 -= elq[3 + k] * dwv[k] + elr[3 + k] * dsp[k];
    }
    dwv[4] = z1 * ellp[2];
    dwv[3] = (z0 - ellp[1] * dwv[4]) * ellp[0];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      const double dw = dwv[c];
      NLP_DW(c, i) = dw;
      dphi -= rhv[c] * dw;
      if (nlp_fixed(i, N, c)) continue;
      const double w = wv[c];
      if (s.lo[c] > -1e299) {
        const double sl = w - s.lo[c], z = zlv[c];
        if (-dw * amax > tau * sl) amax = -tau * sl / dw;          // (divide only when the bound is the binding one so far)
        const double dz = (mub - z * dw) / sl - z;
        if (-dz * az > tau * z) az = -tau * z / dz;
      }
      if (s.hi[c] < 1e299) {
        const double su = s.hi[c] - w, z = zuv[c];
        if (dw * amax > tau * su) amax = tau * su / dw;
        const double dz = (mub + z * dw) / su - z;
        if (-dz * az > tau * z) az = -tau * z / dz;
      }
    }
  }
  *dphi_out = wave_sum(dphi); *amax_out = wave_min(amax); *az_out = wave_min(az);
}

// Take the step (node-parallel): W += a dw, duals += az dz (dz from the step's dw), duals kept near the central path.
__device__ void nlp_apply(const NlpProb &pb, const NlpScen &s, int lane, double a, double az, double mub) {
  const int N = pb.N;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    if (i >= N) continue;
    // (all loads of the node up front: see nlp_merit)
    double wv[NLP_NV], dwl[NLP_NV], zlv[NLP_NV], zuv[NLP_NV];
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) { wv[c] = NLP_W(c, i); dwl[c] = NLP_DW(c, i); zlv[c] = NLP_P(WS_ZL + c, i); zuv[c] = NLP_P(WS_ZU + c, i); }
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      if (nlp_fixed(i, N, c)) continue;
      const double w = wv[c], dw = dwl[c];
      const double wn = w + a * dw;
      NLP_W(c, i) = wn;
      if (s.lo[c] > -1e299) {
        const double sl = w - s.lo[c], z = zlv[c], mn = mub / (wn - s.lo[c]);
        double zn = z + az * ((mub - z * dw) / sl - z);
        zn = fmin(fmax(zn, 1e-10 * mn), 1e10 * mn);
        NLP_P(WS_ZL + c, i) = zn;
      }
      if (s.hi[c] < 1e299) {
        const double su = s.hi[c] - w, z = zuv[c], mn = mub / (s.hi[c] - wn);
        double zn = z + az * ((mub + z * dw) / su - z);
        zn = fmin(fmax(zn, 1e-10 * mn), 1e10 * mn);
        NLP_P(WS_ZU + c, i) = zn;
      }
    }
  }
}

struct NlpOut { double cost, feas; int iters, status; };

// The solve of ONE problem by one wavefront (lane = threadIdx.x & 63): sc its scenario row, Wb [5][N] in/out, wsb its workspace,
// multb [3][N] or null, partner [2][N] frozen positions of the CostCollision partner or null.  Wave-uniform control flow.
__device__ __forceinline__ void nlp_solve_one(int N, double h, const d2d_nlp_opts &o, const double *__restrict__ sc, const double *partner,
                                              double *Wb, double *wsb, double *multb, int lane, NlpOut &out, unsigned long long *stamps, double *ldsw,
                                              const double *__restrict__ bnd) {
  // diagnostics (D2D_NLP_STAMPS): cycles per phase -- merit, assembly, factorisation, back substitution, ratio tests, update
  unsigned long long st_t = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NLP_STAMP(k) if (st_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; }
  const bool st_on = stamps != nullptr;
  if (st_on) st_t = __builtin_amdgcn_s_memtime();
  NlpScen s = nlp_load_scen(sc, o, bnd);
  if (sc[D2D_SC_BANKMAX] != 0.0) { s.sbank = s.skphi * (double)N; s.skphi = 0.0; }     // obj_scale * kbank (the row's S is obj_scale / N)
  NlpProb pb;
  pb.N = N; pb.h = h;
  pb.W = Wb;
  pb.ws = wsb;
  pb.mu = multb ? multb : pb.ws + (size_t)WS_MU * N;
  pb.partner = partner;
  pb.lds = ldsw;
  // (a row whose bounds are unset or inverted would give a zero-width box, infinite duals and NaN pivots for
  // outer_max x 30 assemblies -- refuse it at once)
  {
    bool bad = !(s.hi[3] > s.lo[3]) || !(s.hi[4] > s.lo[4]) || !(s.lo[4] > 0.0) || !(s.hi[0] > s.lo[0]) || !(s.hi[1] > s.lo[1]) || !(s.hi[2] > s.lo[2]);
    for (int c = 0; c < 3; ++c) bad = bad || !(fabs(s.p0[c]) <= 1.79e308) || !(fabs(s.p1[c]) <= 1.79e308);
    if (bad) {
      out.cost = out.feas = __builtin_nan(""); out.iters = 0; out.status = D2D_ST_NONFINITE;
      return;
    }
  }
  // ---- start: end conditions in place, everything else pushed strictly inside the box; duals on the central path
  double mub = o.mub0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    if (i >= N) continue;
#pragma unroll
    for (int c = 0; c < NLP_NV; ++c) {
      double w = NLP_W(c, i);
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
      NLP_W(c, i) = w;
      NLP_P(WS_ZL + c, i) = zl; NLP_P(WS_ZU + c, i) = zu;
      NLP_DW(c, i) = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) NLP_MU(k, i) = 0.0;
  }
  nlp_phase_sync();
  const bool rec_lds = !o.serial && N <= NLP_BCR_LDS_NODES;       // the reduced records go from the assembly to the cyclic reduction through the LDS
  double rho = o.rho0, lam = D2D_LM_LAMBDA0, feas_prev = INFINITY;
  int total_inner = 0, status = D2D_ST_MAXITER;
  double err = 0.0, cost_ref = 0.0, feas = 0.0;
  int n_stalled = 0;                        // outer iterations in a row at the largest penalty without feasibility progress
  // CostBank max mode ends "converged in value": a whole batch of steps that no longer lowers the merit function.  That test needs
  // a window of its own length, whatever the batch of the schedule is: D2D_NLP_BANKMAX_BATCHES batches in one (same step budget).
  const int inner_n = s.sbank > 0.0 ? D2D_NLP_BANKMAX_BATCHES * o.inner_max : o.inner_max;
  const int outer_n = s.sbank > 0.0 ? (o.outer_max + D2D_NLP_BANKMAX_BATCHES - 1) / D2D_NLP_BANKMAX_BATCHES : o.outer_max;
  for (int outer = 1; outer <= outer_n; ++outer) {
    const double tol_in = fmax(fmax(o.opt_tol, fmin(1e-1, 10.0 * mub)), D2D_NLP_GRAD_FLOOR * rho);
    // merit value of the current point for this (mub, rho, mu): one pass here, afterwards the accepted trial's value
    NLP_STAMP(7)
    double phi0 = nlp_merit(pb, s, sc, lane, 0.0, rho, mub, nullptr, nullptr);
    const double phi_first = phi0;
    NLP_STAMP(0)
    bool accepted = false;
    for (int it = 0; it < inner_n; ++it) {
      ++total_inner;
      bool converged = false;
      accepted = false;
      const int imax = s.sbank > 0.0 ? nlp_bank_argmax(pb, lane) : -1;
      for (int tr = 0; tr < 30; ++tr) {
        bool pd;
        err = nlp_assemble(pb, s, sc, lane, rho, mub, lam, &pd, imax, rec_lds);  // (a retry with another damping assembles again: rare)
        nlp_phase_sync();
        NLP_STAMP(1)
        if (tr == 0 && err <= tol_in) { converged = true; break; }
        if (pd) {
          if (o.serial) pd = nlp_factor(pb.ws + (size_t)WS_SIN * N, pb.ws + (size_t)WS_SF * N, pb.ws + (size_t)WS_UP * N, N);   // (UP is free by now)
          else pd = nlp_bcr(pb.ws + (size_t)WS_SIN * N, pb.ws + (size_t)WS_SF * N, pb.ws + (size_t)WS_DS * N, pb.lds, N, rec_lds);
        }
        NLP_STAMP(2)
        if (!pd) { lam = fmin(lam * 8.0, D2D_LM_LAMBDA_MAX); continue; }
        nlp_phase_sync();
        if (o.serial) nlp_backsolve(pb.ws + (size_t)WS_SF * N, pb.ws + (size_t)WS_UP * N, pb.ws + (size_t)WS_DS * N, N);
        nlp_phase_sync();
        NLP_STAMP(3)
        const double tau = fmax(0.99, 1.0 - mub);
        double dphi, amax, az;
        nlp_recover_stats(pb, s, lane, mub, tau, &dphi, &amax, &az);
        nlp_phase_sync();
        NLP_STAMP(4)
        if (!(dphi < 0.0)) { lam = fmin(lam * 8.0, D2D_LM_LAMBDA_MAX); continue; }
        double a = amax, pt = 0.0;
        bool ok = false;
        for (int ls = 0; ls < 8; ++ls) {
          pt = nlp_merit(pb, s, sc, lane, a, rho, mub, nullptr, nullptr);
          if (pt <= phi0 + 1e-4 * a * dphi) { ok = true; break; }
          a *= 0.5;
        }
        NLP_STAMP(0)
        if (ok) {
          phi0 = pt;
          nlp_apply(pb, s, lane, a, az, mub);
          nlp_phase_sync();
          NLP_STAMP(5)
          if (a == amax) lam = fmax(lam / 3.0, D2D_LM_LAMBDA_MIN);
          accepted = true;
          break;
        }
        lam = fmin(lam * 4.0, D2D_LM_LAMBDA_MAX);
      }
      if (converged || !accepted) break;
    }
    (void)nlp_merit(pb, s, sc, lane, 0.0, rho, mub, &cost_ref, &feas);
    if (!(fabs(phi0) <= 1.79e308) || !(fabs(err) <= 1.79e308)) { status = D2D_ST_NONFINITE; break; }
    if (feas <= o.feas_tol && mub <= o.mub_min * 1.0001 && err <= tol_in) { status = D2D_ST_CONVERGED; break; }
    // CostBank max mode: the one-hot cost_grad has no zero where two nodes share the maximum (they do at a min-max optimum): the
    // solve ends converged IN VALUE -- feasible, barrier at its floor, a whole batch of steps that lowered the merit by no more
    // than D2D_NLP_BANKMAX_VALUE_TOL of itself (oracle/nlp.py solve)
    if (s.sbank > 0.0 && feas <= o.feas_tol && mub <= o.mub_min * 1.0001 && (phi_first - phi0) <= D2D_NLP_BANKMAX_VALUE_TOL * (1.0 + fabs(phi0))) {
      status = D2D_ST_CONVERGED; break;
    }
    // the inner problem is not solved yet and the batch still lowered the merit function by more than rounding: same multipliers,
    // penalty and barrier parameter, another batch of steps -- the schedule must not run ahead of the iterate
    if (err > tol_in && accepted && (phi_first - phi0) > (s.sbank > 0.0 ? D2D_NLP_BANKMAX_VALUE_TOL : D2D_NLP_GATE_PROGRESS) * (1.0 + fabs(phi0))) continue;
    // an infeasible problem (e.g. end points too far apart for v_max) or an infeasible stationary point of the violation: the
    // penalty grows tenfold per solved inner problem and the violation no longer halves -- give up instead of running
    // outer_max x inner_max steps
    n_stalled = (feas > 0.5 * feas_prev && feas > 1e3 * o.feas_tol) ? n_stalled + 1 : 0;
    if (n_stalled >= (rho >= D2D_NLP_RHO_MAX ? 3 : D2D_NLP_STALL_OUTERS)) { status = D2D_ST_STALLED; break; }
    // first-order multiplier update (lambda = 2 rho mu); the penalty grows when feasibility stalls
    const bool grow = feas > 0.25 * feas_prev && rho < D2D_NLP_RHO_MAX;
    for (int i0 = 0; i0 < N; i0 += 64) {
      const int i = i0 + lane;
      if (i < 1 || i >= N) continue;
      double wp[3], w[NLP_NV], c3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) wp[c] = NLP_W(c, i - 1);
#pragma unroll
      for (int c = 0; c < NLP_NV; ++c) w[c] = NLP_W(c, i);
      nlp_constraint(s, h, wp, w, c3);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double m = NLP_MU(k, i) + c3[k];
        NLP_MU(k, i) = grow ? m / D2D_NLP_RHO_GROW : m;
      }
    }
    nlp_phase_sync();
    if (grow) rho *= D2D_NLP_RHO_GROW;
    feas_prev = feas;
    mub = fmax(o.mub_min, fmin(0.2 * mub, mub * sqrt(mub)));
  }
  (void)nlp_merit(pb, s, sc, lane, 0.0, rho, mub, &cost_ref, &feas);
  out.cost = cost_ref; out.feas = feas; out.iters = total_inner; out.status = status;
  if (lane == 0 && st_on) {
    NLP_STAMP(7)
    for (int k = 0; k < 8; ++k) stamps[k] = st_acc[k];
    stamps[8] = (unsigned long long)total_inner;
  }
#undef NLP_STAMP
}

#ifndef NLP_WAVES_PER_SIMD
#define NLP_WAVES_PER_SIMD 2       // (A/B: -DNLP_WAVES_PER_SIMD=1 gives the assembly 512 registers and no scratch)
#endif
// One wavefront per problem.  Node-parallel phases (merit, assembly, step statistics, update) run with lane = node; the two
// block recursions are serial in the nodes and run wave-uniform.  Control flow is uniform: no lane waits for another problem.
// Two waves per SIMD: the assembly spills for it (640-708 B / lane, once per Newton step; the serial recursions are separate functions
// and do not).  Round 2 measured +12 % problems/s for it; since round 4's schedule change, and again after round 5's, one
// wave per SIMD (-DNLP_WAVES_PER_SIMD=1: no spill in the kernel) runs at the SAME speed at 512, 4096 and 65 536 problems -- the
// spills are off the critical path, and so is the second wave.
// Persistent: the grid is one wavefront per wave slot of the chip (or per problem, if there are fewer), problems are handed out through
// a device counter, and the WORKSPACE BELONGS TO THE SLOT (work + blockIdx.x * WS_TOTAL * N), not to the problem: nothing in it
// outlives a solve, and with one workspace per problem every solve streamed its 100+ kB through cold lines while the ones it
// followed were written back -- slots keep the chip's working set at (resident waves) x (workspace) whatever the batch.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NLP_WAVES_PER_SIMD, NLP_WAVES_PER_SIMD)))
nlp_solve_kernel(int B, int N, double h, d2d_nlp_opts o, const double *__restrict__ scen, const double *partner, double *W,
                 double *work, double *mult, double *__restrict__ cost_out, double *__restrict__ feas_out,
                 int32_t *__restrict__ iters_out, int32_t *__restrict__ status_out, unsigned long long *stamps, int32_t *queue) {
  const int lane = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) double nlp_lds[];
  double *wsb = work + (size_t)blockIdx.x * WS_TOTAL * N;
  const int32_t *__restrict__ order = o.order;
  for (int t = blockIdx.x; t < B;) {
    const int b = order ? __builtin_amdgcn_readfirstlane(order[t]) : t;
    if ((unsigned)b >= (unsigned)B) {        // (an entry that is no problem index is skipped, not dereferenced)
      int tn = 0;
      if (lane == 0) tn = (int)gridDim.x + atomicAdd(queue, 1);
      t = __builtin_amdgcn_readfirstlane(tn);
      continue;
    }
    NlpOut out;
    nlp_solve_one(N, h, o, scen + (size_t)b * D2D_SCEN_STRIDE, partner ? partner + (size_t)b * 2 * N : nullptr, W + (size_t)b * NLP_NV * N,
                  wsb, mult ? mult + (size_t)b * 3 * N : nullptr, lane, out, b == 0 ? stamps : nullptr, nlp_lds,
                  o.bounds ? o.bounds + (size_t)b * 4 : nullptr);
    if (lane == 0) {
      cost_out[b] = out.cost;
      feas_out[b] = out.feas;
      if (iters_out) iters_out[b] = out.iters;
      if (status_out) status_out[b] = out.status;
    }
    nlp_phase_sync();                        // (the next problem's first stores to the workspace follow this one's last loads)
    int tn = 0;
    if (lane == 0) tn = (int)gridDim.x + atomicAdd(queue, 1);
    t = __builtin_amdgcn_readfirstlane(tn);
  }
}

// The reference's multi-aircraft Problem (src/multi_opt_planner.py:69-78,86) in ONE launch: a workgroup takes a scenario, wavefront
// a its aircraft a.  The aircraft are coupled through the objective only -- CostCollision on the pair (0, 1), src/d2d/
// multiopty_utils.py:120-153; every constraint is per aircraft -- so a fixed point of block Gauss-Seidel over the aircraft (each
// block = the full collocation solve of one aircraft against its partner's frozen node positions) is a KKT point of the joint
// problem.  Sweep 0 solves every aircraft uncoupled, concurrently; then aircraft 0 and 1 take turns (workgroup barriers between
// the turns; the partner's positions are read from its W in global memory, which its wave does not touch meanwhile) until neither
// moved by more than tol in a sweep, or max_sweeps.  No host round trips.  prev [R][2][N] scratch.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NLP_WAVES_PER_SIMD, NLP_WAVES_PER_SIMD)))
nlp_groups_kernel(int R, int n_ac, int N, double h, d2d_nlp_opts o, int max_sweeps, double tol, const double *__restrict__ scen, double *W,
                  double *work, double *mult, double *prev, double *__restrict__ cost_out, double *__restrict__ feas_out,
                  int32_t *__restrict__ iters_out, int32_t *__restrict__ status_out, int32_t *__restrict__ sweeps_out,
                  double *__restrict__ moved_out) {
  __shared__ double moved_s[2];
  extern __shared__ __attribute__((aligned(16))) double nlp_lds[];
  const int r = blockIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = r * n_ac + wave;
  const double *sc = scen + (size_t)b * D2D_SCEN_STRIDE;
  double *Wb = W + (size_t)b * NLP_NV * N;
  double *wsb = work + (size_t)b * WS_TOTAL * N;
  double *mb = mult ? mult + (size_t)b * 3 * N : nullptr;
  const bool coupled = n_ac >= 2 && scen[(size_t)(r * n_ac) * D2D_SCEN_STRIDE + D2D_SC_KCOL] > 0.0;       // wave-uniform for the whole group
  NlpOut out;
  int iters_total = 0;
  double *ldsw = nlp_lds + (size_t)wave * NLP_LDS_DOUBLES;
  const double *bnd = o.bounds ? o.bounds + (size_t)b * 4 : nullptr;
  nlp_solve_one(N, h, o, sc, nullptr, Wb, wsb, mb, lane, out, nullptr, ldsw, bnd);
  iters_total += out.iters;
  if (threadIdx.x < 2) moved_s[threadIdx.x] = 0.0;
  __threadfence_block();
  __syncthreads();
  int sweep = 0;
  double moved = 0.0;
  if (coupled) {
    double *pv = prev + (size_t)r * 2 * N;
    for (sweep = 1; sweep <= max_sweeps; ++sweep) {
      for (int turn = 0; turn < 2; ++turn) {
        if (wave == turn) {
          const double *pw = W + (size_t)(r * n_ac + (1 - turn)) * NLP_NV * N;      // the partner's x and y planes
          for (int i = lane; i < 2 * N; i += 64) pv[i] = Wb[i];
          nlp_solve_one(N, h, o, sc, pw, Wb, wsb, mb, lane, out, nullptr, ldsw, bnd);
          iters_total += out.iters;
          double m = 0.0;
          for (int i = lane; i < 2 * N; i += 64) m = fmax(m, fabs(Wb[i] - pv[i]));
          m = wave_max(m);
          if (lane == 0) moved_s[turn] = m;
        }
        __threadfence_block();
        __syncthreads();
      }
      moved = fmax(moved_s[0], moved_s[1]);
      __syncthreads();
      if (moved <= tol) break;
    }
    if (sweep > max_sweeps) sweep = max_sweeps;
    // not settled after the last sweep: the pair is reported as such (the inner solves each converged, the alternation did not)
    if (moved > tol && wave < 2 && out.status == D2D_ST_CONVERGED) out.status = D2D_ST_MAXITER;
  }
  if (lane == 0) {
    cost_out[b] = out.cost;
    feas_out[b] = out.feas;
    if (iters_out) iters_out[b] = iters_total;
    if (status_out) status_out[b] = out.status;
    if (wave == 0) {
      if (sweeps_out) sweeps_out[r] = sweep;
      if (moved_out) moved_out[r] = moved;
    }
  }
}

extern "C" {

int d2d_nlp_workspace_doubles(int N) { return N * WS_TOTAL; }

int d2d_nlp_solve(d2d_ctx *ctx, int B, int N, double h, const double *scen, const d2d_nlp_opts *opts, double *W,
                  const double *partner, double *work, double *mult, double *cost, double *feas, int32_t *iters, int32_t *status) {
  D2D_REQUIRE(ctx && scen && W && work && cost && feas, "d2d_nlp_solve: null argument");
  D2D_REQUIRE(B >= 1 && N >= 3 && h > 0, "d2d_nlp_solve: B >= 1, N >= 3, h > 0 required (B=%d N=%d h=%g)", B, N, h);
  d2d_nlp_opts o = {D2D_NLP_RHO0, D2D_NLP_MUB0, D2D_NLP_MUB_MIN, 1e-9, 1e-7, 60, 40, 0, 0, nullptr, nullptr};
  if (opts) o = *opts;
  D2D_REQUIRE(o.inner_max >= 1 && o.outer_max >= 1 && o.rho0 > 0 && o.mub0 > 0 && o.mub_min > 0, "d2d_nlp_solve: bad options");
  unsigned long long *stamps = nullptr;
  if (getenv("D2D_NLP_STAMPS")) D2D_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&stamps), 16 * sizeof(unsigned long long)));
  // one wavefront per wave slot (d2d_nlp_opts.slots > 0: the caller's count)
  static int n_cu = 0;
  if (n_cu == 0) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256; }
  int slots = n_cu * 4 * NLP_WAVES_PER_SIMD;
  if (o.slots > 0) slots = o.slots;
  const int grid = B < slots ? B : (slots < 1 ? 1 : slots);
  int32_t *queue = ctx->counter_dev + 2;
  D2D_CHECK_HIP(hipMemsetAsync(queue, 0, sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(nlp_solve_kernel, dim3(grid), dim3(64), NLP_LDS_DOUBLES * sizeof(double), ctx->stream, B, N, h, o, scen, partner, W, work, mult, cost, feas, iters,
                     status, stamps, queue);
  D2D_LAUNCH_CHECK();
  if (stamps) {                                            // diagnostics: synchronous
    unsigned long long hs[9];
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    D2D_CHECK_HIP(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
    D2D_CHECK_HIP(hipFree(stamps));
    static const char *nm[8] = {"merit", "assemble", "factor", "backsolve", "recover+ratio tests", "update", "eliminate", "other"};
    fprintf(stderr, "[nlp] problem 0: %llu Newton steps; shader clocks per phase:", hs[8]);
    for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %llu", nm[k], hs[k]);
    fprintf(stderr, "\n");
  }
  return D2D_OK;
}

int d2d_nlp_solve_groups(d2d_ctx *ctx, int R, int n_ac, int N, double h, const double *scen, const d2d_nlp_opts *opts, int max_sweeps,
                         double tol, double *W, double *work, double *mult, double *cost, double *feas, int32_t *iters, int32_t *status,
                         int32_t *sweeps, double *moved) {
  D2D_REQUIRE(ctx && scen && W && work && cost && feas, "d2d_nlp_solve_groups: null argument");
  D2D_REQUIRE(R >= 1 && n_ac >= 1 && n_ac <= 8 && N >= 3 && h > 0, "d2d_nlp_solve_groups: R >= 1, 1 <= n_ac <= 8, N >= 3, h > 0 required (R=%d n_ac=%d N=%d h=%g)", R, n_ac, N, h);
  D2D_REQUIRE(max_sweeps >= 1 && tol >= 0, "d2d_nlp_solve_groups: max_sweeps >= 1 and tol >= 0 required");
  d2d_nlp_opts o = {D2D_NLP_RHO0, D2D_NLP_MUB0, D2D_NLP_MUB_MIN, 1e-9, 1e-7, 60, 40, 0, 0, nullptr, nullptr};
  if (opts) o = *opts;
  D2D_REQUIRE(o.inner_max >= 1 && o.outer_max >= 1 && o.rho0 > 0 && o.mub0 > 0 && o.mub_min > 0, "d2d_nlp_solve_groups: bad options");
  // scratch for the positions before a turn: the tail of aircraft 0's workspace is not free, so it lives behind the workspaces
  double *prev = work + (size_t)R * n_ac * WS_TOTAL * N;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(nlp_groups_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NLP_LDS_DOUBLES * (int)sizeof(double));
  hipLaunchKernelGGL(nlp_groups_kernel, dim3(R), dim3(64 * n_ac), (size_t)n_ac * NLP_LDS_DOUBLES * sizeof(double), ctx->stream, R, n_ac, N, h, o, max_sweeps, tol, scen, W, work, mult, prev,
                     cost, feas, iters, status, sweeps, moved);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

}  // extern "C"
