// Shared basis block of the polynomial fit (host build + device copies).
#pragma once
#include <vector>

#include "common.h"

// per-trajectory solver state shared by the persistent kernels (fit_kernels.hip, fit_knot.hip):
// flags[b][4] = status, iters, need_eval, evaluations (half-units: 2 per Gauss-Newton, 3 per second-order one)
enum { FL_STATUS = 0, FL_ITERS = 1, FL_NEED = 2, FL_NEVAL = 3 };
// lm[b][LM_STRIDE]: 0 lambda, 1 nu, 2 gmax, 3 second-order mode of the pending evaluation; MINPACK mode: 4 par, 5 trust-region radius,
// 6 phase | first << 1 | calm << 2 | slow << 16, 7 factorisations so far
#define LM_STRIDE 8

// Tables of the knot-space statement (fit_basis.cpp fit_basis_knots; kernel: fit_knot.hip)
#define KN_HB_STRIDE 34          // doubles per sample of the fp64 Hermite table [K][8][4] (+2: conflict-free 16-byte rows)
struct KnotTables {
  int NE = 0, NV = 0;            // full entries 8 (S+1); entries of one axis 4 (S+1)
  int k0[D2D_FIT_MAX_S + 2] = {0};   // first sample of every segment, k0[S] = K
  std::vector<double> Hb64, Bq, BiT, Binv, Minv, Mrow, Pu, msc;
  std::vector<float> Hb32, Wseg, Md32, Mrow32, Mi32;
  // device copies (one allocation each)
  double *d_Hb64 = nullptr, *d_Bq = nullptr, *d_BiT = nullptr, *d_Binv = nullptr, *d_Minv = nullptr, *d_Pu = nullptr, *d_msc = nullptr;
  float *d_Hb32 = nullptr, *d_Wseg = nullptr, *d_Md32 = nullptr, *d_Mrow32 = nullptr, *d_Mi32 = nullptr;
  double *d_u = nullptr;         // [cap_B][64] the knot vector of every fit between launches (bit-exact resume)
  int wpb = 0;                   // wavefronts per workgroup of fit_lm_knot_kernel (0: the plan does not use it)
};

struct d2d_fit_plan {
  int device;
  int S, K, nq;        // segments, samples, reduced unknowns per axis (4*S)
  double duration, T;
  double wref[3];
  // host copies (fp64)
  std::vector<double> G, Gp, Z, Zp, Pinit, G0tG0;
  std::vector<double> Lw, Pe;       // whitening factor L (nq x nq, lower) and the free knot data of q = 0 per unit end datum (nq x 4)
  KnotTables kn;
  // segment formulation of the long-horizon kernel (fit_seg.h; fit_basis.cpp fit_basis_segments)
  std::vector<int> seg;             // [K]  segment of every sample
  std::vector<double> tau;          // [K]  its local time
  std::vector<double> Zl, Zlp, sx;  // [8S][nq] / [8S][4] maps to the Legendre coefficients of the segments; [K] x_k = 2 tau_k / T - 1
  int seg_S = 0, seg_nchunk = 0, seg_l0[7] = {0}, seg_k0[6] = {0}, seg_Ks[6] = {0};
  // device copies
  double *d_G = nullptr;     // [3][K][GSTR]   GSTR = nq+1 (odd stride: conflict-free LDS image)
  double *d_GT = nullptr;    // [3][nq][K]    transposed copy for the long-horizon kernel (lane = sample reads are contiguous)
  double *d_Gp = nullptr;    // [3][K][4]
  float *d_G32 = nullptr;    // [3][K][nq] fp32 planes for the MFMA operand generation
  float *d_W32 = nullptr;    // [nq][nq]  G0^T G0 (waypoint rows' constant J^T J block)
  double *d_Z = nullptr;     // [8S][nq]
  double *d_Zp = nullptr;    // [8S][4]
  double *d_Pinit = nullptr; // [nq][K]
  double *d_Zl64 = nullptr;  // [8S][nq+1]  (odd stride: conflict-free rows and columns in the LDS)
  float *d_Zl32 = nullptr;   // [8S][nq]
  double *d_Zlp = nullptr;   // [8S][4]
  double *d_sx = nullptr;    // [K]
  // solver scratch, grown on demand (d2d_fit_solve)
  int cap_B = 0;
  double *d_g = nullptr;     // [B][2nq]
  float *d_H = nullptr;      // [B][2nq][2nq]
  double *d_cost = nullptr;  // [B]
  double *d_lm = nullptr;    // [B][8] per-trajectory solver state (fit_kernels.hip LM_STRIDE)
  int32_t *d_flags = nullptr;  // [B][4] status, iters, need_eval, nevals
  double *d_prep = nullptr;    // [B][FIT_PREP_STRIDE] derived scenario rows (fit_prep_kernel)
  double *d_pk = nullptr;      // [B][FIT_PK][K] per-sample constants (fit_prepk_kernel)
  double *d_pos = nullptr;     // [B][2][K] sampled positions (coupled groups)
  double *d_qprev = nullptr;   // [B][2nq] unknowns at the start of a Gauss-Seidel sweep
  float *d_rows = nullptr;     // [B][K+1][4] f32x4 row records of the last d2d_fit_rows (input of d2d_fit_jtj)
  int rows_B = 0;              // trajectories whose records d_rows holds
  int32_t *d_order = nullptr;  // [B] hand-out order of the persistent LM kernel (longest fits of the previous solve first)
  int order_B = 0;             // batch size the order was built for (0: none)
  // predicted hand-out (d2d_fit_opts.handout = D2D_HANDOUT_PREDICTED): the prior table, the keys and the order of the current solve
  float *d_hprior = nullptr;   // [2][D2D_HANDOUT_NB][D2D_HANDOUT_ND] (plan lifetime; built-in table or d2d_fit_plan_set_handout_prior)
  bool has_prior = false;      // the table says something about THIS plan's fits: the built-in one for the shape it was regressed on (S = 6, K <= 64), a caller's for any
  int32_t *d_hkey = nullptr;   // [B] sort keys (scratch)
  int32_t *d_order_pred = nullptr;   // [B] (scratch; d_order stays the caller's explicit hint)
  const int32_t *last_order = nullptr;   // what the last solve launch handed out by (nullptr: index order) ...
  int last_order_B = 0;              // ... over this many trajectories (d2d_fit_plan_get_order)
  int32_t *d_ring = nullptr;   // [ring_cap] ring of yielded fits of the persistent LM kernel (-1 = empty slot)
  int ring_cap = 0;            // power of two >= cap_B
  int gorder_R = 0, gsweeps_R = 0;   // coupled groups: d_order[0, R) holds a scenario order / d_order[R, 2R) the sweeps of the last solve
  int n_group = 1, nds = 0;    // aircraft per coupled group and padded collision-row slots per sample
  const double *prep_valid_for = nullptr;   // scen pointer d_prep was derived from
  // launch geometry chosen at plan creation from the LDS footprint
  bool g32_lds = true;
  int wpb_eval = 8, wpb_step = 8, wpb_lm = 0;
  bool use_lm = false;      // whole LM loop in one persistent launch (fit_lm_kernel)
  bool use_long = false;    // ... in fit_lm_long_kernel: K > 64, samples in chunks of 64, basis through L2
  bool split_ok = true;     // the split-path kernels' LDS image holds this K (d2d_fit_eval, coupled groups)
  int n_cu = 256;
  int kernel_req = D2D_FIT_KERNEL_AUTO, long_tables = -1;   // d2d_fit_plan_opts
  int it_done = 0, active_B = 0;   // LM loop state between d2d_fit_begin / iterate / finish
  const double *solve_q = nullptr; // the q buffer the solve in progress owns (first d2d_fit_iterate after d2d_fit_begin)
  int solve_kernel = -1;           // ... and the persistent kernel that holds its state: 0 fit_lm_kernel, 1 fit_lm_knot_kernel, 2 long (-1: none yet)
  int last_slice = 0, last_running = -1;   // ... slice of the last d2d_fit_iterate (0: no time-sliced hand-out), its count of RUNNING fits (-1: not counted)
  d2d_fit_opts last_opts = {};     // ... and its options (d2d_fit_finish sweeps up what a sliced launch left in the ring)
  // optional per-launch timing (d2d_fit_profile)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;   // start/stop pairs
  std::vector<int> prof_kind;        // 0 = eval (J^T J) launch, 1 = step launch, 2 = fused LM launch, 3 = contraction-only (fit_jtj) launch
};

int fit_basis_build(d2d_fit_plan *pl);   // fills the host vectors
int fit_basis_segments(d2d_fit_plan *pl);   // ... of the segment formulation (after fit_basis_build)
int fit_basis_knots(d2d_fit_plan *pl);      // ... of the knot-space statement (after fit_basis_segments)
// fit_knot.hip: the persistent LM kernel in knot coordinates (S = 6, K <= 64, default solver)
int fit_knot_plan_init(d2d_fit_plan *pl);
void fit_knot_plan_free(d2d_fit_plan *pl);
int fit_knot_ensure(d2d_fit_plan *pl, int cap_B);
int fit_knot_launch(d2d_ctx *ctx, d2d_fit_plan *pl, int B, double *q, const d2d_fit_opts &o, int iter_cap, const int32_t *order, int prio_at);
