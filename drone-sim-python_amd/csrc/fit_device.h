// Device functions of the polynomial trajectory fit: one wavefront per trajectory,
// lane = sample for the flat-output / residual phase, lane = unknown for J^T r and the
// triangular solves, v_mfma_f32_16x16x4_f32 for J^T J.
// Restates oracle/fit.py; reference lines are relative to the reference repository root.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/d2d.h"

#define FIT_G 9.81        // src/d2d/guidance.py:39
#define FIT_OBS_K 2.0     // src/d2d/opty_utils.py:103
#define FIT_OBS_ARGMAX 3.4538776394910684   // 0.5 * ln(1e3): the clip of kind-0 obstacles, :111

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Compiler-level ordering of LDS traffic between the lanes of ONE wavefront (the LDS
// itself executes a wave's DS instructions in order).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Value known to be equal in every lane -> scalar registers, so that branches on it are scalar
// branches (uniform control flow) instead of EXEC-masked divergent ones.
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Wave reductions on the DPP cross-lane path (no LDS round trips): xor-1 and xor-2 inside quads,
// row_half_mirror, row_mirror -> every lane holds the total of its row of 16; the four row totals
// are combined through scalar registers.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#define DPP_QUAD_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<DPP_QUAD_XOR1>(v);
  v += dpp_move<DPP_QUAD_XOR2>(v);
  v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_move<DPP_ROW_MIRROR>(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_move<DPP_QUAD_XOR1>(v));
  v = fmax(v, dpp_move<DPP_QUAD_XOR2>(v));
  v = fmax(v, dpp_move<DPP_ROW_HALF_MIRROR>(v));
  v = fmax(v, dpp_move<DPP_ROW_MIRROR>(v));
  return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}

// Geometry of the shared LDS image of the basis block.
struct FitGeom {
  int K, nq, gstr;   // gstr = nq + 1: odd row stride -> conflict-free for lane=sample AND lane=unknown
};

// Scenario row in registers (wave-uniform values).
struct Scen {
  double x0, y0, psi0, x1, y1, psi1, vref, vsp, kv, kphi, kobs, s, wwp, wx, wy, goleft;
  double o0x, o0y, o0r, o1x, o1y, o1r, wbnd, phimax, vmin, vmax;
  double dx[4], dy[4];          // end data [pos0, vel0, pos1, vel1] per axis
  double p2x, p2y;              // apex of the 'tri' dog-leg
};

__device__ __forceinline__ Scen load_scen(const double *__restrict__ sc, double duration) {
  Scen s;
  s.x0 = sc[D2D_SC_X0]; s.y0 = sc[D2D_SC_Y0]; s.psi0 = sc[D2D_SC_PSI0];
  s.x1 = sc[D2D_SC_X1]; s.y1 = sc[D2D_SC_Y1]; s.psi1 = sc[D2D_SC_PSI1];
  s.vref = sc[D2D_SC_VREF]; s.vsp = sc[D2D_SC_VSP]; s.kv = sc[D2D_SC_KV]; s.kphi = sc[D2D_SC_KPHI];
  s.kobs = sc[D2D_SC_KOBS]; s.s = sc[D2D_SC_S]; s.wwp = sc[D2D_SC_WWP]; s.wx = sc[D2D_SC_WX];
  s.wy = sc[D2D_SC_WY]; s.goleft = sc[D2D_SC_GOLEFT];
  s.o0x = sc[D2D_SC_O0X]; s.o0y = sc[D2D_SC_O0Y]; s.o0r = sc[D2D_SC_O0R];
  s.o1x = sc[D2D_SC_O1X]; s.o1y = sc[D2D_SC_O1Y]; s.o1r = sc[D2D_SC_O1R];
  s.wbnd = sc[D2D_SC_WBND]; s.phimax = sc[D2D_SC_PHIMAX]; s.vmin = sc[D2D_SC_VMIN]; s.vmax = sc[D2D_SC_VMAX];
  double s0, c0, s1, c1;
  sincos(s.psi0, &s0, &c0);
  sincos(s.psi1, &s1, &c1);
  s.dx[0] = s.x0; s.dx[1] = s.vref * c0; s.dx[2] = s.x1; s.dx[3] = s.vref * c1;
  s.dy[0] = s.y0; s.dy[1] = s.vref * s0; s.dy[2] = s.y1; s.dy[3] = s.vref * s1;
  // triangle(), src/d2d/opty_utils.py:171-187
  const double ex = s.x1 - s.x0, ey = s.y1 - s.y0;
  const double d = sqrt(ex * ex + ey * ey);
  const double ux = ex / d, uy = ey / d;
  const double D = s.vref * duration;
  s.p2x = s.x0 + ex / 2; s.p2y = s.y0 + ey / 2;
  if (D > d) {
    const double sg = (s.goleft > 0.0) ? 1.0 : ((s.goleft < 0.0) ? -1.0 : 0.0);
    const double h = sg * sqrt(D * D - d * d) / 2;
    s.p2x += h * (-uy); s.p2y += h * ux;
  }
  return s;
}

// numpy.linspace(a, b, n)[i]
__device__ __forceinline__ double linspace_at(double a, double b, int n, int i) {
  if (n == 1) return a;
  if (i == n - 1) return b;
  const double step = (b - a) / (n - 1);
  return i * step + a;
}

__device__ __forceinline__ void waypoint_at(const Scen &s, int K, int k, double &wx, double &wy) {
  const int n1 = K / 2, n2 = K - n1;
  if (k < n1) { wx = linspace_at(s.x0, s.p2x, n1, k); wy = linspace_at(s.y0, s.p2y, n1, k); }
  else { wx = linspace_at(s.p2x, s.x1, n2, k - n1); wy = linspace_at(s.p2y, s.y1, n2, k - n1); }
}

// Flat outputs at sample k:  Y = [x, y, xd, yd, xdd, ydd]  (oracle/fit.py flat_outputs)
template <typename ScenT>
__device__ __forceinline__ void flat_outputs(const FitGeom &g, const double *__restrict__ G64,
                                             const double *__restrict__ Gp64,
                                             const double *__restrict__ q, const ScenT &s, int k,
                                             double Y[6]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double *gp = Gp64 + ((size_t)d * g.K + k) * 4;
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { ax = fma(gp[c], s.dx[c], ax); ay = fma(gp[c], s.dy[c], ay); }
    Y[2 * d] = ax; Y[2 * d + 1] = ay;
  }
  const double *g0 = G64 + (size_t)k * g.gstr;
  const double *g1 = g0 + (size_t)g.K * g.gstr;
  const double *g2 = g1 + (size_t)g.K * g.gstr;
  for (int j = 0; j < g.nq; ++j) {
    const double qx = q[j], qy = q[g.nq + j];
    const double a0 = g0[j], a1 = g1[j], a2 = g2[j];
    Y[0] = fma(a0, qx, Y[0]); Y[1] = fma(a0, qy, Y[1]);
    Y[2] = fma(a1, qx, Y[2]); Y[3] = fma(a1, qy, Y[3]);
    Y[4] = fma(a2, qx, Y[4]); Y[5] = fma(a2, qy, Y[5]);
  }
}

// ---------------------------------------------------------------------------------------
// Hot-path scenario row: everything that is uniform per trajectory is derived once by
// fit_prep_kernel (double[FIT_PREP_STRIDE] per trajectory) so that the eval / step kernels
// fetch it with scalar loads instead of recomputing sincos / sqrt / divisions in all 64 lanes.
#define FIT_PREP_STRIDE 112
enum { PR_DX = 0, PR_DY = 4, PR_P2X = 8, PR_P2Y, PR_SAX, PR_SAY, PR_SBX, PR_SBY, PR_CV, PR_CPHI, PR_COBS,
       PR_K0, PR_K1, PR_CV2, PR_CPHI2, PR_WB2, PR_WWP, PR_WBND, PR_VSP, PR_WX, PR_WY, PR_PHIMAX, PR_VMIN,
       PR_VMAX, PR_O0X, PR_O0Y, PR_O1X, PR_O1Y, PR_X0, PR_Y0, PR_X1, PR_Y1, PR_CCOL, PR_KC, PR_PMASK,
       PR_C0, PR_C1, PR_CPHIMAX,
       PR_EXT,                           // obstacles 2.. : {x, y, k, c} each (k = 0: absent), D2D_MAX_OBS - 2 of them
       PR_BOX = PR_EXT + 4 * (D2D_MAX_OBS - 2),      // xmin, xmax, ymin, ymax of the soft position box (+-1e300: open)
       PR_NEXT = PR_BOX + 4 };   // index of the last present extra obstacle + 1, + FIT_XBOX if there is a position box
                                 // (0: a trajectory with at most two obstacles and no box -- the fast path)
#define FIT_XBOX 64
#define FIT_XIN 128         // (sample_terms: position-row sums handed in by the caller)
static_assert(PR_NEXT < FIT_PREP_STRIDE, "prep row overflow");

struct ScenP {
  double cv, cphi, cobs, k0, k1, cv2, cphi2, wb2, wwp, wbnd, vsp, wx, wy, phimax, vmin, vmax;
  double o0x, o0y, o1x, o1y;
  double ccol, kc;      // collision rows: sqrt(s_col*kcol), k / rcol
  int pmask;            // bit j: coupled with aircraft j of the group
  double c0, c1;        // obstacle rows: exponent offset (r^2 for kind 0, 0 for kind 1)
  double cphimax;       // CostBank max mode: weight sqrt(obj_scale*kphi) of the one selected phi row, 0 = mean mode
  const double *ext;    // the prep row's PR_EXT block (obstacles 2..), read where it is used: it costs the common
                        // trajectory (<= 2 obstacles, no box) one load of the code per sample and no registers
};

// Group coupling context of one trajectory (collision rows against the other aircraft of its
// scenario, whose sampled positions pos [B][2][K] are frozen during this block's solve).
struct GroupCtx {
  const double *pos;    // NULL: uncoupled
  int n_ac, self, gbase, nds;   // group size, own index, first trajectory of the group, padded row slots
};

__device__ __forceinline__ void prep_row(const double *__restrict__ sc, double duration, int K,
                                         double *__restrict__ o) {
  const Scen s = load_scen(sc, duration);
  for (int c = 0; c < 4; ++c) { o[PR_DX + c] = s.dx[c]; o[PR_DY + c] = s.dy[c]; }
  const int n1 = K / 2, n2 = K - n1;
  o[PR_P2X] = s.p2x; o[PR_P2Y] = s.p2y;
  o[PR_SAX] = n1 > 1 ? (s.p2x - s.x0) / (n1 - 1) : 0.0; o[PR_SAY] = n1 > 1 ? (s.p2y - s.y0) / (n1 - 1) : 0.0;
  o[PR_SBX] = n2 > 1 ? (s.x1 - s.p2x) / (n2 - 1) : 0.0; o[PR_SBY] = n2 > 1 ? (s.y1 - s.p2y) / (n2 - 1) : 0.0;
  o[PR_CV2] = s.s * s.kv; o[PR_CPHI2] = s.s * s.kphi;
  o[PR_CV] = sqrt(s.s * s.kv); o[PR_CPHI] = sqrt(s.s * s.kphi); o[PR_COBS] = sqrt(s.s * s.kobs);
  // obstacle row h = cobs * exp(min(0.5 * (c - |k (p - o)|^2), FIT_OBS_ARGMAX)): kind 1 (CostObstacle, e =
  // exp(-|2 (p-o)/r|^2)): k = 2/r, c = 0; kind 0 (e = clip(exp(r^2 - |p-o|^2), 0, 1e3)): k = 1, c = r^2
  const int okind = (int)sc[D2D_SC_OKIND];
  o[PR_K0] = s.o0r > 0.0 ? ((okind & 1) ? 1.0 : FIT_OBS_K / s.o0r) : 0.0;
  o[PR_K1] = s.o1r > 0.0 ? ((okind & 2) ? 1.0 : FIT_OBS_K / s.o1r) : 0.0;
  o[PR_C0] = (okind & 1) ? s.o0r * s.o0r : 0.0; o[PR_C1] = (okind & 2) ? s.o1r * s.o1r : 0.0;
  o[PR_CPHIMAX] = sc[D2D_SC_BANKMAX] != 0.0 ? sqrt(s.s * K * s.kphi) : 0.0;
  int next = 0;
  for (int i = 2; i < D2D_MAX_OBS; ++i) {
    const double *ob = sc + D2D_SC_OEXT + 3 * (i - 2);
    double *e = o + PR_EXT + 4 * (i - 2);
    const bool kind0 = (okind >> i) & 1;
    e[0] = ob[0]; e[1] = ob[1];
    e[2] = ob[2] > 0.0 ? (kind0 ? 1.0 : FIT_OBS_K / ob[2]) : 0.0;
    e[3] = kind0 ? ob[2] * ob[2] : 0.0;
    if (ob[2] > 0.0) next = i - 1;
  }
  const bool bx = sc[D2D_SC_XMIN] < sc[D2D_SC_XMAX], by = sc[D2D_SC_YMIN] < sc[D2D_SC_YMAX];
  o[PR_BOX + 0] = bx ? sc[D2D_SC_XMIN] : -1e300; o[PR_BOX + 1] = bx ? sc[D2D_SC_XMAX] : 1e300;
  o[PR_BOX + 2] = by ? sc[D2D_SC_YMIN] : -1e300; o[PR_BOX + 3] = by ? sc[D2D_SC_YMAX] : 1e300;
  o[PR_NEXT] = (double)(next + ((bx || by) ? FIT_XBOX : 0));
  o[PR_WB2] = s.wbnd * s.wbnd; o[PR_WWP] = s.wwp; o[PR_WBND] = s.wbnd; o[PR_VSP] = s.vsp;
  o[PR_WX] = s.wx; o[PR_WY] = s.wy; o[PR_PHIMAX] = s.phimax; o[PR_VMIN] = s.vmin; o[PR_VMAX] = s.vmax;
  o[PR_O0X] = s.o0x; o[PR_O0Y] = s.o0y; o[PR_O1X] = s.o1x; o[PR_O1Y] = s.o1y;
  o[PR_X0] = s.x0; o[PR_Y0] = s.y0; o[PR_X1] = s.x1; o[PR_Y1] = s.y1;
  const double kcol = sc[D2D_SC_KCOL], rcol = sc[D2D_SC_RCOL], scol = sc[D2D_SC_SCOL];
  o[PR_CCOL] = (kcol > 0.0 && scol > 0.0) ? sqrt(scol * kcol) : 0.0;
  o[PR_KC] = rcol > 0.0 ? FIT_OBS_K / rcol : 0.0;
  o[PR_PMASK] = sc[D2D_SC_PMASK];
}

__device__ __forceinline__ ScenP load_scenp(const double *__restrict__ p) {
  ScenP s;
  s.cv = p[PR_CV]; s.cphi = p[PR_CPHI]; s.cobs = p[PR_COBS]; s.k0 = p[PR_K0]; s.k1 = p[PR_K1];
  s.cv2 = p[PR_CV2]; s.cphi2 = p[PR_CPHI2]; s.wb2 = p[PR_WB2]; s.wwp = p[PR_WWP]; s.wbnd = p[PR_WBND];
  s.vsp = p[PR_VSP]; s.wx = p[PR_WX]; s.wy = p[PR_WY]; s.phimax = p[PR_PHIMAX]; s.vmin = p[PR_VMIN]; s.vmax = p[PR_VMAX];
  s.o0x = p[PR_O0X]; s.o0y = p[PR_O0Y]; s.o1x = p[PR_O1X]; s.o1y = p[PR_O1Y];
  s.ccol = p[PR_CCOL]; s.kc = p[PR_KC]; s.pmask = (int)p[PR_PMASK];
  s.c0 = p[PR_C0]; s.c1 = p[PR_C1]; s.cphimax = p[PR_CPHIMAX];
  s.ext = p + PR_EXT;
  return s;
}

// Per-sample constants of one trajectory (fit_prepk_kernel), pk [B][FIT_PK][K]:
//   rows 0..5 = the end-condition part of the flat outputs, Gp_d[k] . d_axis (x,y,xd,yd,xdd,ydd)
//   rows 6..7 = the 'tri' waypoint of sample k (numpy.linspace semantics, src/d2d/opty_utils.py:171-187)
#define FIT_PK 8
__device__ __forceinline__ void prepk_entry(const double *__restrict__ pr, const double *__restrict__ Gp64, int K, int k,
                                            double o[FIT_PK]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double *gp = Gp64 + ((size_t)d * K + k) * 4;
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { ax = fma(gp[c], pr[PR_DX + c], ax); ay = fma(gp[c], pr[PR_DY + c], ay); }
    o[2 * d] = ax; o[2 * d + 1] = ay;
  }
  const int n1 = K / 2, n2 = K - n1;
  if (k < n1) {
    o[6] = (k == n1 - 1 && n1 > 1) ? pr[PR_P2X] : k * pr[PR_SAX] + pr[PR_X0];
    o[7] = (k == n1 - 1 && n1 > 1) ? pr[PR_P2Y] : k * pr[PR_SAY] + pr[PR_Y0];
  } else {
    const int i = k - n1;
    o[6] = (i == n2 - 1 && n2 > 1) ? pr[PR_X1] : i * pr[PR_SBX] + pr[PR_P2X];
    o[7] = (i == n2 - 1 && n2 > 1) ? pr[PR_Y1] : i * pr[PR_SBY] + pr[PR_P2Y];
  }
}

// LDS copy of the unknowns of one trajectory: the two axes interleaved, q_lds[2j] = x-axis unknown j,
// q_lds[2j+1] = y-axis unknown j (one 16-byte broadcast read per j).  q_slot: where unknown `lane` goes.
__device__ __forceinline__ int q_slot(int lane, int nq) { return lane < nq ? 2 * lane : 2 * (lane - nq) + 1; }

// Flat outputs of sample k from the per-sample base pk[0..5] plus G_d[k] . q  (oracle/fit.py flat_outputs).
// NQ > 0: compile-time nq -- unrolled in groups of four unknowns, every LDS read with an immediate offset
// and the 16 reads of a group requested before its 24 FMAs.
template <int NQ>
__device__ __forceinline__ void flat_outputs_pk(const FitGeom &g, const double *__restrict__ G64,
                                                const double *__restrict__ q, const double pk[FIT_PK], int k, double Y[6]) {
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
  const int nq = NQ ? NQ : g.nq, gstr = NQ ? NQ + 1 : g.gstr;
#pragma unroll
  for (int c = 0; c < 6; ++c) Y[c] = pk[c];
  const double *g0 = G64 + (size_t)k * gstr;
  const double *g1 = g0 + (size_t)g.K * gstr;
  const double *g2 = g1 + (size_t)g.K * gstr;
  int j = 0;
  if (NQ) {
    // two unknowns per stage, software-pipelined by hand: the 10 reads of stage s+1 are requested before
    // the 12 FMAs of stage s run; the scheduling barriers keep the compiler from requesting everything
    // at once (it would hold ~180 VGPRs of operands)
    constexpr int NS = NQ / 2;
    f64x2a qq[2][2];
    double a0[2][2], a1[2][2], a2[2][2];
#define FO_LOAD(S, B)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                          \
    qq[B][i] = *reinterpret_cast<const f64x2a *>(q + 2 * (2 * (S) + i));                   \
    a0[B][i] = g0[2 * (S) + i]; a1[B][i] = g1[2 * (S) + i]; a2[B][i] = g2[2 * (S) + i];    \
  }
    FO_LOAD(0, 0)
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int cur = st & 1;
      if (st + 1 < NS) { FO_LOAD(st + 1, cur ^ 1) }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        Y[0] = fma(a0[cur][i], qq[cur][i].x, Y[0]); Y[1] = fma(a0[cur][i], qq[cur][i].y, Y[1]);
        Y[2] = fma(a1[cur][i], qq[cur][i].x, Y[2]); Y[3] = fma(a1[cur][i], qq[cur][i].y, Y[3]);
        Y[4] = fma(a2[cur][i], qq[cur][i].x, Y[4]); Y[5] = fma(a2[cur][i], qq[cur][i].y, Y[5]);
      }
      // all six chains advance in every stage (without this the optimiser sinks four of them behind the
      // loop and holds their 96 operands in registers)
      asm volatile("" : "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]), "+v"(Y[4]), "+v"(Y[5]));
      __builtin_amdgcn_sched_barrier(0);
    }
#undef FO_LOAD
    j = 2 * NS;
  }
  for (; j < nq; ++j) {
    const double qx = q[2 * j], qy = q[2 * j + 1];
    const double a0 = g0[j], a1 = g1[j], a2 = g2[j];
    Y[0] = fma(a0, qx, Y[0]); Y[1] = fma(a0, qy, Y[1]);
    Y[2] = fma(a1, qx, Y[2]); Y[3] = fma(a1, qy, Y[3]);
    Y[4] = fma(a2, qx, Y[4]); Y[5] = fma(a2, qy, Y[5]);
  }
}

// Residual rows of one sample (oracle/fit.py residuals).  Returns sum r^2.
// With WANT_JAC: u[6] = D^T r (for J^T r, fp64) and the fp32 coefficients of the four rows the
// MFMA contracts -- v, phi (bound rows merged into their weights), and two position rows (obs0, obs1, or
// the contraction of all position rows when there are more: see below).  Row rho's entry
// for unknown j of axis a is  cA[rho][a]*TA_rho[k][j] + cB[rho][a]*TB_rho[k][j]  with
// (TA,TB) = (G1,-) for v, (G1,G2) for phi, (G0,-) for the obstacles:
//   coef[rho] = {cA_x, cB_x, cA_y, cB_y}
// |w| = |tan(phi)| of one sample: what CostBank's max mode ranks the samples by (phi = atan(w) is monotone)
__device__ __forceinline__ double sample_absw(const ScenP &s, const double Y[6]) {
  const double a = Y[2] - s.wx, b = Y[3] - s.wy;
  return fabs(Y[5] * a - Y[4] * b) / (sqrt(a * a + b * b) * FIT_G);
}

// bank_sel: CostBank max mode only -- true for the one sample whose phi row is kept (weight s.cphimax).
// so != nullptr (second-order mode): instead of the four Gauss-Newton row coefficient sets, coef[m] receives row m
// of the 4x4 block M_vel = sum_rows (grad r grad r^T + r Hess r) over (a,b,c,d) = (xd-wx, yd-wy, xdd, ydd), laid
// out as (M[m][a], M[m][c], M[m][b], M[m][d]) (x-axis G1/G2 pair, y-axis G1/G2 pair), and so[m] row m of the 2x2
// position block M_pos of the obstacle rows (oracle/fit.py curvature_blocks + the Gauss-Newton part).
template <bool WANT_JAC>
// xin != nullptr: further position rows that the caller has already summed over -- {sum h^2, sum o_x h, sum o_y h, sum o_x^2,
// sum o_x o_y, sum o_y^2} with o = d row / d (x, y): the collision rows of a coupled group (partner_sums).  They join the
// two contracted position rows like obstacles 2.. do.
__device__ __forceinline__ double sample_terms(const ScenP &s, const double Y[6], double wpx,
                                               double wpy, double u[6], f32x4 coef[4], bool bank_sel = false,
                                               float2 *so = nullptr, const double *xin = nullptr, bool xin_on = true) {
  const double cphi = s.cphimax > 0.0 ? (bank_sel ? s.cphimax : 0.0) : s.cphi;
  const double cphi2 = s.cphimax > 0.0 ? cphi * cphi : s.cphi2;
  const double x = Y[0], y = Y[1];
  const double a = Y[2] - s.wx, b = Y[3] - s.wy, c = Y[4], d = Y[5];
  const double va2 = a * a + b * b, va = sqrt(va2);
  const double n = d * a - c * b;
  const double ivg = 1.0 / (va * FIT_G);
  const double w = n * ivg;
  const double phi = atan(w);                       // src/d2d/guidance.py:40
  const double r0 = s.cv * (va - s.vsp);            // CostInput, src/d2d/opty_utils.py:85-97
  const double r1 = cphi * phi;
  const double r2 = s.wwp * (x - wpx), r3 = s.wwp * (y - wpy);
  double h0 = 0.0, h1 = 0.0, e0x = 0.0, e0y = 0.0, e1x = 0.0, e1y = 0.0;
  bool clip0 = false, clip1 = false;                // kind 0 only: the row sits on its clip (zero slope)
  if (s.k0 > 0.0) {                                 // CostObstacle, :99-134 (both kinds, see prep_row)
    e0x = (x - s.o0x) * s.k0; e0y = (y - s.o0y) * s.k0;
    const double arg = 0.5 * (s.c0 - (e0x * e0x + e0y * e0y));
    clip0 = arg > FIT_OBS_ARGMAX;
    h0 = s.cobs * exp(clip0 ? FIT_OBS_ARGMAX : arg);
  }
  if (s.k1 > 0.0) {
    e1x = (x - s.o1x) * s.k1; e1y = (y - s.o1y) * s.k1;
    const double arg = 0.5 * (s.c1 - (e1x * e1x + e1y * e1y));
    clip1 = arg > FIT_OBS_ARGMAX;
    h1 = s.cobs * exp(clip1 ? FIT_OBS_ARGMAX : arg);
  }
  // obstacles 2.. and the position box (rare): their rows only enter through sums -- cost, D^T r and the 2x2
  // position block
  double xh2 = 0.0, xux = 0.0, xuy = 0.0, xpxx = 0.0, xpxy = 0.0, xpyy = 0.0;
  // (xin_on: a runtime switch beside the pointer -- a caller that SELECTS between an array and nullptr forces the array into
  // scratch memory and its reads through a dynamic address)
  const bool have_xin = xin != nullptr && xin_on;
  if (have_xin) { xh2 = xin[0]; xux = xin[1]; xuy = xin[2]; xpxx = xin[3]; xpxy = xin[4]; xpyy = xin[5]; }
  const int xcode = (int)s.ext[PR_NEXT - PR_EXT] | (have_xin ? FIT_XIN : 0);
  const int next = xcode & (FIT_XBOX - 1);
  if (xcode & FIT_XBOX) {                             // rows w_b*dist(x, [xmin, xmax]), w_b*dist(y, [ymin, ymax])
    const double *bx = s.ext + (PR_BOX - PR_EXT);
    const double hx = fmax(x - bx[1], 0.0) + fmin(x - bx[0], 0.0);
    const double hy = fmax(y - bx[3], 0.0) + fmin(y - bx[2], 0.0);
    xh2 = s.wb2 * (hx * hx + hy * hy);
    if (WANT_JAC) {                                   // piecewise linear: no curvature, the same block in both modes
      xux = s.wb2 * hx; xuy = s.wb2 * hy;
      xpxx = hx != 0.0 ? s.wb2 : 0.0; xpyy = hy != 0.0 ? s.wb2 : 0.0;
    }
  }
  for (int i = 0; i < next; ++i) {
    const double *ob = s.ext + 4 * i;
    const double ki = ob[2];
    if (ki > 0.0) {
      const double ex = (x - ob[0]) * ki, ey = (y - ob[1]) * ki;
      const double arg = 0.5 * (ob[3] - (ex * ex + ey * ey));
      const bool clip = arg > FIT_OBS_ARGMAX;
      const double h = s.cobs * exp(clip ? FIT_OBS_ARGMAX : arg);
      xh2 = fma(h, h, xh2);
      if (WANT_JAC && !clip) {
        const double ox = -h * ex * ki, oy = -h * ey * ki;
        xux = fma(ox, h, xux); xuy = fma(oy, h, xuy);
        if (so == nullptr) {                          // Gauss-Newton: sum of o o^T
          xpxx = fma(ox, ox, xpxx); xpxy = fma(ox, oy, xpxy); xpyy = fma(oy, oy, xpyy);
        } else {                                      // second order: h^2 k^2 (2 e e^T - I)
          const double gi = h * h * ki * ki;
          xpxx = fma(gi, 2.0 * ex * ex - 1.0, xpxx); xpxy = fma(2.0 * gi * ex, ey, xpxy);
          xpyy = fma(gi, 2.0 * ey * ey - 1.0, xpyy);
        }
      }
    }
  }
  const double hphi = fmax(fabs(phi) - s.phimax, 0.0);
  const double hv = fmax(va - s.vmax, 0.0) + fmin(va - s.vmin, 0.0);
  const double r6 = s.wbnd * hphi, r7 = s.wbnd * hv;
  const double cost = r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3 + h0 * h0 + h1 * h1 + r6 * r6 + r7 * r7 + xh2;
  if (WANT_JAC) {
    const double iva = 1.0 / va;
    const double dva_a = a * iva, dva_b = b * iva;
    const double f = 1.0 / (1.0 + w * w);
    const double nv3 = n * ivg * iva * iva;
    const double dp_a = (d * ivg - nv3 * a) * f, dp_b = (-c * ivg - nv3 * b) * f;
    const double dp_c = -b * ivg * f, dp_d = a * ivg * f;
    const double actp = (hphi > 0.0) ? ((phi > 0.0) ? 1.0 : -1.0) : 0.0;
    const double actv = (va > s.vmax || va < s.vmin) ? 1.0 : 0.0;
    const double o0x = clip0 ? 0.0 : -h0 * e0x * s.k0, o0y = clip0 ? 0.0 : -h0 * e0y * s.k0;
    const double o1x = clip1 ? 0.0 : -h1 * e1x * s.k1, o1y = clip1 ? 0.0 : -h1 * e1y * s.k1;
    // u = D^T r over all eight rows
    const double tv = s.cv * r0 + s.wbnd * actv * r7;               // multiplies d va
    const double tp = cphi * r1 + s.wbnd * actp * r6;               // multiplies d phi
    u[0] = s.wwp * r2 + o0x * h0 + o1x * h1 + xux;
    u[1] = s.wwp * r3 + o0y * h0 + o1y * h1 + xuy;
    u[2] = tv * dva_a + tp * dp_a;
    u[3] = tv * dva_b + tp * dp_b;
    u[4] = tp * dp_c;
    u[5] = tp * dp_d;
    if (so == nullptr) {
      // merged row weights for J^T J: (cv^2 + wb^2 actv) dva dva^T, (cphi^2 + wb^2 |actp|) dphi dphi^T
      const double mv = sqrt(s.cv2 + s.wb2 * actv);
      const double mp = sqrt(cphi2 + s.wb2 * actp * actp);
      coef[0] = f32x4{(float)(mv * dva_a), 0.f, (float)(mv * dva_b), 0.f};
      coef[1] = f32x4{(float)(mp * dp_a), (float)(mp * dp_c), (float)(mp * dp_b), (float)(mp * dp_d)};
      if (xcode == 0) {
        coef[2] = f32x4{(float)o0x, 0.f, (float)o0y, 0.f};
        coef[3] = f32x4{(float)o1x, 0.f, (float)o1y, 0.f};
      } else {
        // more than two position rows (obstacles, box): their J^T J part at this sample is G0^T P G0 with the 2x2
        // P = sum o o^T, so two rows carry it whatever their number: the rows of the Cholesky factor of P,
        // (l11, l21) and (0, l22)
        const double pxx = o0x * o0x + o1x * o1x + xpxx, pxy = o0x * o0y + o1x * o1y + xpxy;
        const double pyy = o0y * o0y + o1y * o1y + xpyy;
        const double l11 = sqrt(pxx), l21 = pxx > 0.0 ? pxy / l11 : 0.0;
        const double l22 = sqrt(fmax(pyy - l21 * l21, 0.0));
        coef[2] = f32x4{(float)l11, 0.f, (float)l21, 0.f};
        coef[3] = f32x4{0.f, 0.f, (float)l22, 0.f};
      }
    } else {
      // ---- velocity block: mv2 t t^T + kv (I - t t^T)  +  (mp2 - 2 w tp) dphi dphi^T  +  tp f Hess(w)
      const double mv2 = s.cv2 + s.wb2 * actv, mp2 = cphi2 + s.wb2 * actp * actp;
      const double kv = (s.cv * r0 + s.wbnd * actv * r7) * iva;
      const double e1 = mv2 - kv;                               // coefficient of t t^T on top of kv I
      const double e2 = mp2 - 2.0 * w * tp;
      const double i2 = iva * iva, q1 = ivg * i2, tf = tp * f, wi2 = w * i2;
      const double hw_aa = -2.0 * q1 * d * a + wi2 * (3.0 * a * a * i2 - 1.0);
      const double hw_ab = -q1 * (d * b - a * c) + 3.0 * wi2 * i2 * a * b;
      const double hw_bb = 2.0 * q1 * c * b + wi2 * (3.0 * b * b * i2 - 1.0);
      const double hw_ac = q1 * a * b, hw_ad = ivg - q1 * a * a, hw_bc = -ivg + q1 * b * b, hw_bd = -q1 * a * b;
      const double m_aa = kv + e1 * dva_a * dva_a + e2 * dp_a * dp_a + tf * hw_aa;
      const double m_ab = e1 * dva_a * dva_b + e2 * dp_a * dp_b + tf * hw_ab;
      const double m_bb = kv + e1 * dva_b * dva_b + e2 * dp_b * dp_b + tf * hw_bb;
      const double m_ac = e2 * dp_a * dp_c + tf * hw_ac, m_ad = e2 * dp_a * dp_d + tf * hw_ad;
      const double m_bc = e2 * dp_b * dp_c + tf * hw_bc, m_bd = e2 * dp_b * dp_d + tf * hw_bd;
      const double m_cc = e2 * dp_c * dp_c, m_cd = e2 * dp_c * dp_d, m_dd = e2 * dp_d * dp_d;
      coef[0] = f32x4{(float)m_aa, (float)m_ac, (float)m_ab, (float)m_ad};     // row a: (a, c | b, d)
      coef[1] = f32x4{(float)m_ab, (float)m_bc, (float)m_bb, (float)m_bd};     // row b
      coef[2] = f32x4{(float)m_ac, (float)m_cc, (float)m_bc, (float)m_cd};     // row c
      coef[3] = f32x4{(float)m_ad, (float)m_cd, (float)m_bd, (float)m_dd};     // row d
      // ---- position block: sum over the unclipped obstacles of h^2 k^2 (2 e e^T - I)
      const double g0 = clip0 ? 0.0 : h0 * h0 * s.k0 * s.k0, g1 = clip1 ? 0.0 : h1 * h1 * s.k1 * s.k1;
      const double p_xx = g0 * (2.0 * e0x * e0x - 1.0) + g1 * (2.0 * e1x * e1x - 1.0) + xpxx;
      const double p_xy = 2.0 * (g0 * e0x * e0y + g1 * e1x * e1y) + xpxy;
      const double p_yy = g0 * (2.0 * e0y * e0y - 1.0) + g1 * (2.0 * e1y * e1y - 1.0) + xpyy;
      so[0] = float2{(float)p_xx, (float)p_xy};
      so[1] = float2{(float)p_xy, (float)p_yy};
    }
  }
  return cost;
}

// The collision rows of sample k against the coupled aircraft of the group as SUMS (cost, D^T r and the 2x2 Gauss-Newton
// position block): out = {sum h^2, sum o_x h, sum o_y h, sum o_x^2, sum o_x o_y, sum o_y^2} -- sample_terms' xin.
__device__ __forceinline__ void partner_sums(const ScenP &s, const GroupCtx &gc, int K, int k, double x, double y, double out[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) out[i] = 0.0;
  if (gc.pos && s.ccol > 0.0) {
    for (int m = 0; m < gc.n_ac; ++m) {
      if (m == gc.self || !((s.pmask >> m) & 1)) continue;
      const double *pm = gc.pos + (size_t)(gc.gbase + m) * 2 * K;
      const double ex = (x - pm[k]) * s.kc, ey = (y - pm[K + k]) * s.kc;
      const double h = s.ccol * exp(-0.5 * (ex * ex + ey * ey));
      const double ox = -h * ex * s.kc, oy = -h * ey * s.kc;
      out[0] = fma(h, h, out[0]); out[1] = fma(ox, h, out[1]); out[2] = fma(oy, h, out[2]);
      out[3] = fma(ox, ox, out[3]); out[4] = fma(ox, oy, out[4]); out[5] = fma(oy, oy, out[5]);
    }
  }
}

// Collision rows of sample k against the coupled aircraft of the group (CostCollision,
// src/d2d/multiopty_utils.py:120-153, as residual rows sqrt(s_col*kcol*e)).  Adds to u[0..1]; writes
// the fp32 row coefficients (d row/dx, d row/dy) into cfd_k[0..nds) (zero for unused slots).
template <bool WANT_JAC>
__device__ __forceinline__ double partner_terms(const ScenP &s, const GroupCtx &gc, int K, int k, double x, double y,
                                                double u[6], float2 *cfd_k) {
  double cost = 0.0;
  int slot = 0;
  if (gc.pos && s.ccol > 0.0) {
    for (int m = 0; m < gc.n_ac; ++m) {
      if (m == gc.self || !((s.pmask >> m) & 1)) continue;
      const double *pm = gc.pos + (size_t)(gc.gbase + m) * 2 * K;
      const double ex = (x - pm[k]) * s.kc, ey = (y - pm[K + k]) * s.kc;
      const double h = s.ccol * exp(-0.5 * (ex * ex + ey * ey));
      cost += h * h;
      if (WANT_JAC) {
        const double ox = -h * ex * s.kc, oy = -h * ey * s.kc;
        u[0] += ox * h; u[1] += oy * h;
        if (slot < gc.nds) cfd_k[slot] = float2{(float)ox, (float)oy};
      }
      ++slot;
    }
  }
  if (WANT_JAC)
    for (; slot < gc.nds; ++slot) cfd_k[slot] = float2{0.f, 0.f};
  return cost;
}

// CostBank max mode: index of the sample with the largest |phi| (the first one on ties, as numpy.argmax,
// src/d2d/opty_utils.py:80); -1 in mean mode.  Wave-cooperative, the result is wave-uniform.
template <int NQ>
__device__ __forceinline__ int bank_argmax(const FitGeom &g, const double *G64, const double *__restrict__ pkb,
                                           const double *q, const ScenP &s, int lane) {
  if (!(s.cphimax > 0.0)) return -1;
  double best = -1.0;
  int kstar = 0;
  for (int k0 = 0; k0 < g.K; k0 += 64) {
    const int k = k0 + lane;
    double aw = -1.0;
    if (k < g.K) {
      double pk[FIT_PK], Y[6];
#pragma unroll
      for (int c = 0; c < FIT_PK; ++c) pk[c] = pkb[(size_t)c * g.K + k];
      flat_outputs_pk<NQ>(g, G64, q, pk, k, Y);
      aw = sample_absw(s, Y);
    }
    const double m = wave_max(aw);
    if (m > best) {
      best = m;
      kstar = k0 + (int)__builtin_ctzll(__ballot(aw == m));
    }
  }
  return __builtin_amdgcn_readfirstlane(kstar);
}

// Cost at q (wave-cooperative): sum over samples of sum r^2.  pkb = this trajectory's [FIT_PK][K] block.
__device__ __forceinline__ double wave_cost(const FitGeom &g, const double *G64, const double *__restrict__ pkb,
                                            const double *q, const ScenP &s, int lane,
                                            const GroupCtx &gc = GroupCtx{nullptr, 1, 0, 0, 0}) {
  double acc = 0.0;
  const int kbank = bank_argmax<0>(g, G64, pkb, q, s, lane);
  for (int k0 = 0; k0 < g.K; k0 += 64) {
    const int k = k0 + lane;
    if (k < g.K) {
      double pk[FIT_PK], Y[6];
#pragma unroll
      for (int c = 0; c < FIT_PK; ++c) pk[c] = pkb[(size_t)c * g.K + k];
      flat_outputs_pk<0>(g, G64, q, pk, k, Y);
      acc += sample_terms<false>(s, Y, pk[6], pk[7], nullptr, nullptr, k == kbank);
      acc += partner_terms<false>(s, gc, g.K, k, Y[0], Y[1], nullptr, nullptr);
    }
  }
  return wave_sum(acc);
}
