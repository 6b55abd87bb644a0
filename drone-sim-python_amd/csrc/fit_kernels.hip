// Polynomial trajectory fit: kernels and C-ABI entry points (include/d2d.h).
//
//   fit_lm_kernel   : the whole Levenberg-Marquardt loop of a trajectory in one persistent launch
//                     (hot path of d2d_fit_solve for 6-segment plans)
//   fit_eval_kernel : flat outputs + residuals (fp64), J^T r (fp64), J^T J (fp32 MFMA)
//   fit_step_kernel : damped Cholesky solve (fp32), trial cost (fp64), Nielsen gain-ratio update
//                     (one LM iteration per launch pair: d2d_fit_eval, coupled groups, other plans)
// One wavefront per trajectory; the basis block shared by the whole batch is staged in LDS
// once per workgroup; the phases themselves live in fit_phases.h.  Restates oracle/fit.py
// (lm_solve, eval_normal, bgs_solve).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fit_device.h"
#include "fit_phases.h"
#include "fit_seg.h"
#include "fit_plan.h"
#include "fit_handout_prior.h"

// wavefronts (= trajectories) per workgroup, fewer when K is large.  The launch bounds set the VGPR cap:
// 16 waves -> 128 VGPRs (fit_eval_kernel needs ~85), 12 waves -> 168 (fit_step_kernel: ~135)
#define FIT_EVAL_WPB_MAX 16
#define FIT_WPB_MAX 12
#define FIT_THREADS (64 * FIT_WPB_MAX)
#define FIT_LDS_BYTES (160 * 1024)
// fit_eval_kernel and fit_step_kernel carry 256 B of LDS of the compiler's own beside the dynamic block (kernel-resource-usage: "LDS
// Size 256"); a layout that fills the 160 KiB to the last 224 bytes -- K = 64: 163 616 B dynamic -- was refused by the queue
// (HSA_STATUS_ERROR_INVALID_ALLOCATION, found by the round-6 tests that evaluate a K = 64 plan): their layouts leave that room
#define FIT_LDS_STATIC 256

// (flags[b][4] and lm[b][LM_STRIDE]: fit_plan.h)

struct FitLds {
  // byte offsets into dynamic LDS
  int G64, G32, wave0, wave_stride;
  int q, u, coef, cfd;     // offsets inside a wave's private block
  int total;
};

// sub-batch addressing and group coupling of one launch: trajectory t = off + i*stride, i < B
struct GroupArgs {
  const double *pos;      // [Btotal][2][K] sampled positions, NULL when uncoupled
  int n_ac, off, stride, nds;
};

static inline int align16(int v) { return (v + 15) & ~15; }

// g32_lds: stage the fp32 operand tables in LDS (else they are read through L1/L2);
// wpb: wavefronts per workgroup.  pick_eval_layout chooses the largest that fits 160 KiB.
static FitLds eval_lds_layout(int K, int nq, bool g32_lds, int wpb, int nds = 0) {
  FitLds L;
  const int gstr = nq + 1;
  int o = 64;                                   // the first 64 bytes: one flag word per wave (fit_lm_kernel: exclusive SIMDs for the longest fits)
  L.G64 = o; o = align16(o + 3 * K * gstr * 8);
  L.G32 = o; o = align16(o + (g32_lds ? (3 * K + 1) * nq * 4 : 0));   // + one padded sample row
  L.wave0 = o;
  int w = 0;
  L.q = w; w = align16(w + 2 * nq * 8);
  L.u = w; w = align16(w + K * 6 * 8);
  L.coef = w; w = align16(w + (K + 1) * 4 * 16);   // + one padded sample
  L.cfd = w; w = align16(w + (K + 1) * nds * 8);
  L.wave_stride = w;
  L.total = o + wpb * w;
  return L;
}

static bool pick_eval_layout(int K, int nq, bool *g32_lds, int *wpb, int nds = 0) {
  for (int pass = 0; pass < 2; ++pass) {
    const bool in_lds = pass == 0;
    for (int w = FIT_EVAL_WPB_MAX; w >= (in_lds ? 4 : 1); --w)
      if (eval_lds_layout(K, nq, in_lds, w, nds).total <= FIT_LDS_BYTES - FIT_LDS_STATIC) { *g32_lds = in_lds; *wpb = w; return true; }
  }
  return false;
}

// Cooperative copy global -> LDS (whole workgroup), 8-byte granules.
__device__ __forceinline__ void stage(void *dst, const void *src, int bytes) {
  double *d = reinterpret_cast<double *>(dst);
  const double *s = reinterpret_cast<const double *>(src);
  for (int i = threadIdx.x; i < bytes / 8; i += blockDim.x) d[i] = s[i];
}

// ------------------------------------------------------------------------------------
// derived scenario rows (fit_device.h prep_row), one thread per trajectory
__global__ void __launch_bounds__(256)
fit_prep_kernel(int B, int K, double duration, const double *__restrict__ scen, double *__restrict__ prep) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  prep_row(scen + (size_t)b * D2D_SCEN_STRIDE, duration, K, prep + (size_t)b * FIT_PREP_STRIDE);
}

// per-sample constants of every trajectory: pk [B][FIT_PK][K] (fit_device.h prepk_entry)
__global__ void __launch_bounds__(256)
fit_prepk_kernel(int B, int K, const double *__restrict__ prep, const double *__restrict__ Gp64, double *__restrict__ pk) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * K) return;
  const int b = i / K, k = i - (long)b * K;
  double o[FIT_PK];
  prepk_entry(prep + (size_t)b * FIT_PREP_STRIDE, Gp64, K, k, o);
#pragma unroll
  for (int c = 0; c < FIT_PK; ++c) pk[((size_t)b * FIT_PK + c) * K + k] = o[c];
}

// ------------------------------------------------------------------------------------
// K1 + K2: cost, J^T r, J^T J.   H is written as the upper block triangle of 16x16 tiles
// of a [n][n] row-major matrix (n = 2nq), in the accumulator layout; untile_kernel expands it for the public API.
template <int NB, int NQ, bool G32_LDS>   // NB = ceil(2nq/16) column blocks of the MFMA tiling; NQ = nq or 0 (runtime)
__global__ void __launch_bounds__(64 * FIT_EVAL_WPB_MAX)
fit_eval_kernel(int B, FitGeom g, FitLds L, int dbg, GroupArgs ga, const double *__restrict__ gG64,
                const double *__restrict__ pk, const float *__restrict__ gG32,
                const float *__restrict__ gW32, const double *__restrict__ prep,
                const double *__restrict__ q_in, int32_t *__restrict__ flags,
                double *__restrict__ cost_out, double *__restrict__ g_out, float *__restrict__ H_out,
                f32x4 *__restrict__ rows_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  // persistent over the batch: wave w of workgroup g takes sub-batch entries g*wpb + w, + gridDim*wpb, ...
  // (after their first trajectory the waves of a SIMD drift apart, so one wave's MFMA phase overlaps
  // the fp64 VALU phases of the others)
  const int stride_bl = gridDim.x * wpb;
  bool any = false;
  for (int bl = blockIdx.x * wpb + wave; bl < B; bl += stride_bl) {
    const int b = ga.off + bl * ga.stride;
    any = any || !flags || (flags[4 * b + FL_STATUS] == D2D_ST_RUNNING && flags[4 * b + FL_NEED] != 0);
  }
  // nothing to do for this workgroup: leave before paying for the LDS image
  if (!__syncthreads_or(any ? 1 : 0)) return;

  const int n = 2 * g.nq;
  double *G64 = reinterpret_cast<double *>(lds + L.G64);
  if (!(dbg & 16)) {
    stage(G64, gG64, 3 * g.K * g.gstr * 8);
    if (G32_LDS) stage(lds + L.G32, gG32, (3 * g.K + 1) * g.nq * 4);
  }
  unsigned char *wl = lds + L.wave0 + wave * L.wave_stride;
  double *qs = reinterpret_cast<double *>(wl + L.q);
  double *us = reinterpret_cast<double *>(wl + L.u);
  f32x4 *cf = reinterpret_cast<f32x4 *>(wl + L.coef);
  float2 *cfd = reinterpret_cast<float2 *>(wl + L.cfd);
  __syncthreads();
  // De-phase the waves that share a SIMD (HW_ID.WAVE_ID = slot on the SIMD): waves that start together
  // stay in lockstep -- all in the fp64 VALU phases, then all queueing on the matrix pipe.  A start
  // offset of about a quarter trajectory per slot lets one wave's MFMA phase run under the others'
  // VALU phases for the rest of the launch.  Only worth it when a wave has several trajectories.
  if (B > 2 * stride_bl) {
    const int slot = __builtin_amdgcn_s_getreg(6148) & 3;       // hwreg(HW_REG_HW_ID, 0, 4)
    for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(100);
  }

  for (int bl = blockIdx.x * wpb + wave; bl < B; bl += stride_bl) {
    const int b = ga.off + bl * ga.stride;
    if (flags && !(flags[4 * b + FL_STATUS] == D2D_ST_RUNNING && flags[4 * b + FL_NEED] != 0)) continue;
    if (lane < n) qs[q_slot(lane, g.nq)] = q_in[(size_t)b * n + lane];
    const ScenP s = load_scenp(prep + (size_t)b * FIT_PREP_STRIDE);
    wave_lds_sync();
    double g_lane;
    const GroupCtx gc{ga.pos, ga.n_ac, ga.n_ac > 0 ? b % ga.n_ac : 0, ga.n_ac > 0 ? (b / ga.n_ac) * ga.n_ac : 0, ga.pos ? ga.nds : 0};
    const double cost = eval_cost_grad<NQ>(g, G64, pk + (size_t)b * FIT_PK * g.K, qs, us, cf, s, lane, dbg, g_lane, gc, cfd);
    if (lane < n && g_out) g_out[(size_t)b * n + lane] = g_lane;
    if (rows_out) {      // the fp32 row records of this evaluation, for a later fit_jtj_kernel launch (d2d_fit_rows / d2d_fit_jtj)
      f32x4 *dst = rows_out + (size_t)b * 4 * (g.K + 1);
      for (int i = lane; i < 4 * g.K; i += 64) dst[i] = cf[i];
    }
    if (lane == 0) {
      if (cost_out) cost_out[b] = cost;
      if (flags) {
        flags[4 * b + FL_NEED] = 0;
        flags[4 * b + FL_NEVAL] += 2;      // half-units: see fit_stats_kernel
        if (!(fabs(cost) <= 1.79e308)) flags[4 * b + FL_STATUS] = D2D_ST_NONFINITE;
      }
    }
    if (H_out) {
      f32x4 acc[NB * (NB + 1) / 2];
      jtj_mfma<NB, NQ, G32_LDS>(g, lds, L.G32, gG32, L.wave0 + wave * L.wave_stride + L.coef, lane, (dbg & 4) ? 1 : g.K, acc,
                                L.wave0 + wave * L.wave_stride + L.cfd, gc.nds);
      // epilogue: tile-major store [tile][reg][lane], 256 B per store instruction.  The waypoint rows'
      // constant block wwp^2 G0^T G0 is added by the consumer (fit_step_kernel / untile_kernel).
      float *Hb = H_out + (size_t)b * (NB * (NB + 1) / 2) * 256;
#pragma unroll
      for (int t = 0; t < NB * (NB + 1) / 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Hb[(t * 4 + r) * 64 + lane] = acc[t][r];
    }
    wave_lds_sync();     // the next trajectory reuses this wave's LDS block
  }
}

// tile-major J^T J (fit_eval_kernel's layout: [tile (I<=J)][reg][lane], C/D map row = 16I + 4(lane>>4) + reg,
// col = 16J + (lane&15)) -> full symmetric row-major [n][n] for the public d2d_fit_eval
__global__ void __launch_bounds__(256)
untile_kernel(int B, int n, int NB, const float *__restrict__ Ht, const float *__restrict__ Wt,
              const double *__restrict__ prep, float *__restrict__ H) {
  const int b = blockIdx.x;
  const float ww = (float)(prep[(size_t)b * FIT_PREP_STRIDE + PR_WWP] * prep[(size_t)b * FIT_PREP_STRIDE + PR_WWP]);
  const float *src = Ht + (size_t)b * (NB * (NB + 1) / 2) * 256;
  float *dst = H + (size_t)b * n * n;
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) {
    int r = i / n, c = i - r * n;
    if ((r >> 4) > (c >> 4)) { const int t = r; r = c; c = t; }
    const int I = r >> 4, J = c >> 4;
    const int tile = I * NB - I * (I - 1) / 2 + (J - I);
    const int rr = r & 15, reg = rr & 3, ln = (rr >> 2) * 16 + (c & 15);
    dst[i] = fmaf(ww, Wt[(tile * 4 + reg) * 64 + ln], src[(tile * 4 + reg) * 64 + ln]);
  }
}

// ------------------------------------------------------------------------------------
// The contraction alone: J^T J of every trajectory from the fp32 row records a previous launch left in HBM
// (fit_eval_kernel with rows_out; d2d_fit_rows / d2d_fit_jtj).  The whole kernel is the MFMA pass of
// jtj_mfma -- records HBM -> the wave's LDS block, operands from the LDS copy of the basis planes,
// 6 x 50 v_mfma_f32_16x16x4_f32 per trajectory, tile-major store -- so its duration IS the contraction's
// (the figure bench.py reports as roofline_isolated).  Algorithmic HBM bytes per trajectory: 4K records x 16 B
// read + NB(NB+1)/2 tiles x 1 KiB written (3.2 kB + 6 kB at K = 50, nq = 24).
struct JtjLds {
  int G32, wave0, wave_stride, total;
};
static JtjLds jtj_lds_layout(int K, int nq, int wpb) {
  JtjLds L;
  L.G32 = 0;
  L.wave0 = align16((3 * K + 1) * nq * 4);
  L.wave_stride = align16((K + 1) * 4 * 16);
  L.total = L.wave0 + wpb * L.wave_stride;
  return L;
}

// clk != NULL (diagnostic launch, D2D_JTJ_CLOCK=1): wave 0 of every workgroup adds its shader-clock and 100 MHz real-time ticks
// around the trajectory loop to clk[0..1] (in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, MICROARCH guide).
template <int NB, int NQ>
__global__ void __launch_bounds__(64 * FIT_EVAL_WPB_MAX)
fit_jtj_kernel(int B, FitGeom g, JtjLds L, const float *__restrict__ gG32, const f32x4 *__restrict__ rows,
               float *__restrict__ H_out, unsigned long long *__restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  stage(lds + L.G32, gG32, (3 * g.K + 1) * g.nq * 4);
  const int cf_off = L.wave0 + wave * L.wave_stride;
  f32x4 *cf = reinterpret_cast<f32x4 *>(lds + cf_off);
  __syncthreads();
  const int stride_bl = gridDim.x * wpb;
  const int nrec = 4 * g.K;
  // the records of the wave's next trajectory travel in registers while the MFMAs of the current one run
  constexpr int NPF = 4;                          // 4 x 64 records = K <= 64 samples; longer horizons load the rest late
  f32x4 pf[NPF];
  unsigned long long t0 = 0, r0 = 0;
  if (clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  int b = blockIdx.x * wpb + wave;
  if (b < B) {
#pragma unroll
    for (int m = 0; m < NPF; ++m)
      if (lane + 64 * m < nrec) pf[m] = rows[(size_t)b * 4 * (g.K + 1) + lane + 64 * m];
  }
  for (; b < B; b += stride_bl) {
#pragma unroll
    for (int m = 0; m < NPF; ++m)
      if (lane + 64 * m < nrec) cf[lane + 64 * m] = pf[m];
    for (int i = lane + 64 * NPF; i < nrec; i += 64) cf[i] = rows[(size_t)b * 4 * (g.K + 1) + i];
    const int bn = b + stride_bl;
    if (bn < B) {
#pragma unroll
      for (int m = 0; m < NPF; ++m)
        if (lane + 64 * m < nrec) pf[m] = rows[(size_t)bn * 4 * (g.K + 1) + lane + 64 * m];
    }
    wave_lds_sync();
    f32x4 acc[NB * (NB + 1) / 2];
    jtj_mfma<NB, NQ, true>(g, lds, L.G32, gG32, cf_off, lane, g.K, acc);
    float *Hb = H_out + (size_t)b * (NB * (NB + 1) / 2) * 256;
#pragma unroll
    for (int t = 0; t < NB * (NB + 1) / 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) Hb[(t * 4 + r) * 64 + lane] = acc[t][r];
    wave_lds_sync();     // the next trajectory's records overwrite this wave's LDS block
  }
  if (clk && wave == 0 && lane == 0) {
    atomicAdd(&clk[0], __builtin_amdgcn_s_memtime() - t0);
    atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
  }
}

// ------------------------------------------------------------------------------------
// K3: one damped solve + trial + accept/reject.  N = padded system size (16*NB).
struct StepLds {
  int G64, wave0, wave_stride, Lm, vec, qt, total;
};
static StepLds step_lds_layout(int K, int nq, int N, int wpb) {
  StepLds L;
  const int gstr = nq + 1;
  int o = 64;                                   // the first 64 bytes: one flag word per wave (fit_lm_kernel: exclusive SIMDs for the longest fits)
  L.G64 = o; o = align16(o + 3 * K * gstr * 8);
  L.wave0 = o;
  int w = 0;
  {
    L.Lm = w; w = align16(w + CHOL_IMAGE_BYTES(N));     // image of J^T J, then of its Cholesky factor
  }
  L.vec = w; w = align16(w + N * 4);
  L.qt = w; w = align16(w + N * 8);
  L.wave_stride = w;
  L.total = o + wpb * w;
  return L;
}

static bool pick_step_layout(int K, int nq, int N, int *wpb) {
  for (int w = FIT_WPB_MAX; w >= 1; --w)
    if (step_lds_layout(K, nq, N, w).total <= FIT_LDS_BYTES - FIT_LDS_STATIC) { *wpb = w; return true; }
  return false;
}

template <int N>
__global__ void __launch_bounds__(FIT_THREADS)
fit_step_kernel(int B, FitGeom g, StepLds L, d2d_fit_opts opts, GroupArgs ga,
                const double *__restrict__ gG64, const double *__restrict__ pk, const float *__restrict__ gWt,
                const double *__restrict__ prep, double *__restrict__ q_io,
                const double *__restrict__ g_in, const float *__restrict__ H_in,
                double *__restrict__ cost_io, double *__restrict__ lm, int32_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int bl = blockIdx.x * (blockDim.x >> 6) + wave;
  const int b = ga.off + bl * ga.stride;
  const bool active = bl < B && flags[4 * b + FL_STATUS] == D2D_ST_RUNNING;
  if (!__syncthreads_or(active ? 1 : 0)) return;
  double *G64 = reinterpret_cast<double *>(lds + L.G64);
  stage(G64, gG64, 3 * g.K * g.gstr * 8);
  __syncthreads();
  if (!active) return;
  unsigned char *wl = lds + L.wave0 + wave * L.wave_stride;
  float *Lm = reinterpret_cast<float *>(wl + L.Lm);       // [N][N+4]
  double *qt = reinterpret_cast<double *>(wl + L.qt);     // trial point
  const int n = 2 * g.nq;

  const double lam = lm[LM_STRIDE * b + 0], nu = lm[LM_STRIDE * b + 1];
  const double c = cost_io[b];
  const int iters = flags[4 * b + FL_ITERS];
  const bool act = lane < n;
  const double gi = act ? g_in[(size_t)b * n + lane] : 0.0;
  const double qi = act ? q_io[(size_t)b * n + lane] : 0.0;
  const double gmax = wave_max(fabs(gi));
  if (lane == 0) lm[LM_STRIDE * b + 2] = gmax;
  if (gmax <= opts.gtol) {                                 // wave-uniform
    if (lane == 0) flags[4 * b + FL_STATUS] = D2D_ST_CONVERGED;
    return;
  }
  // H arrives tile-major (fit_eval_kernel: coalesced loads in the accumulator layout): add the waypoint
  // block, write the row-major LDS image, read row `lane`; the image then becomes the Cholesky factor.
  constexpr int NBs = N / 16, NT = NBs * (NBs + 1) / 2;
  f32x2 hrow[N / 2];
  {
    const float *Hb = H_in + (size_t)b * NT * 256;
    const double wwp = prep[(size_t)b * FIT_PREP_STRIDE + PR_WWP];
    const float ww = (float)(wwp * wwp);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(ww, gWt[(t * 4 + r) * 64 + lane], Hb[(t * 4 + r) * 64 + lane]);
    tiles_to_image<N>(acc, Lm, lane);
    image_put_rhs<N>(Lm, lane, gi);
  }
  wave_lds_sync();
  image_row<N>(Lm, lane, hrow);
  const float hdiag = image_diag<N>(Lm, lane);
  wave_lds_sync();
  float dgi, dl;
  const bool ok = damped_solve<N>(hrow, lam, act, lane, Lm, dgi, dl, nullptr, false, 0, 0.0, nullptr, nullptr, hdiag, true);
  const double delta = (double)dl;
  // ---- trial points (the full step, then up to two shortened ones), predicted and actual reduction ------
  const ScenP s = load_scenp(prep + (size_t)b * FIT_PREP_STRIDE);
  const GroupCtx gc{ga.pos, ga.n_ac, ga.n_ac > 0 ? b % ga.n_ac : 0, ga.n_ac > 0 ? (b / ga.n_ac) * ga.n_ac : 0, ga.pos ? ga.nds : 0};
  const double pred = wave_sum(delta * (lam * (double)dgi * delta - gi));
  const double dmax = wave_max(fabs(delta)), qmax = wave_max(fabs(qi));
  double ct = 0.0, pred_s = pred, alpha = 1.0;
  bool fin = false, accept = false;
  if (ok) {
    if (act) qt[q_slot(lane, g.nq)] = qi + delta;
    wave_lds_sync();
    ct = wave_cost(g, G64, pk + (size_t)b * FIT_PK * g.K, qt, s, lane, gc);
    fin = (fabs(ct) <= 1.79e308) && (pred > 0.0);
    accept = fin && (c - ct) / pred > 0.0;
    if (!accept && fin) {
      const double a = -2.0 * wave_sum(gi * delta), bq = a - pred;
      double al = bt_first_alpha(a, c, ct);
      for (int att = 0; att < 2; ++att) {
        wave_lds_sync();
        if (act) qt[q_slot(lane, g.nq)] = qi + al * delta;
        wave_lds_sync();
        const double c2 = wave_cost(g, G64, pk + (size_t)b * FIT_PK * g.K, qt, s, lane, gc);
        if ((fabs(c2) <= 1.79e308) && c2 < c) { accept = true; alpha = al; ct = c2; pred_s = a * al - bq * al * al; break; }
        al = fmax(D2D_LM_BT_SHRINK * al, D2D_LM_BT_FLOOR);
      }
    }
  }
  const StepOutcome so = lm_update(ok, fin, accept, alpha, c, ct, pred, pred_s, dmax, qmax, lam, nu, opts);
  int status = so.status;
  const double lam_n = so.lam, nu_n = so.nu;
  if (so.accept) {
    if (act) q_io[(size_t)b * n + lane] = qi + alpha * delta;
    if (lane == 0) {
      cost_io[b] = ct;
      flags[4 * b + FL_NEED] = 1;     // gradient / Hessian at the new point (also when converged)
    }
  }
  if (lane == 0) {
    lm[LM_STRIDE * b + 0] = lam_n;
    lm[LM_STRIDE * b + 1] = nu_n;
    flags[4 * b + FL_ITERS] = iters + 1;
    if (status == D2D_ST_RUNNING && iters + 1 >= opts.max_iter) status = D2D_ST_MAXITER;
    flags[4 * b + FL_STATUS] = status;
  }
}

// ------------------------------------------------------------------------------------
// The whole Levenberg-Marquardt loop in ONE persistent launch: a wavefront takes a trajectory,
// runs eval -> damped solve -> trial -> accept/reject until the trajectory stops (or its
// iteration budget for this launch is spent), stores the state and takes its next trajectory.  J^T J never leaves the CU (MFMA accumulators -> LDS tiles -> the
// Cholesky's registers), waves drift apart so that one wave's MFMA phase overlaps another's
// fp64 VALU phases, and trajectories that converge early free their wave for the tail.
#define FIT_LM_WPB_MAX 8
#define GROUPS_LS_STRIDE 128    // doubles per aircraft of the line search of fit_groups_kernel (the unknowns before the sweep [48], the trial point [48] behind them; lives in d_H)
struct FusedLds {
  int G64, G32, Wt, wave0, wave_stride;
  int qs, sp, big, cf, cfp; // inside a wave's block; `big` holds us + cf + cfp, then the image of J^T J / its factor
  int total;
};
static FusedLds fused_lds_layout(int K, int nq, int N, int wpb, int nds = 0) {
  FusedLds L;
  const int gstr = nq + 1;
  int o = 64;                                   // the first 64 bytes: one flag word per wave (fit_lm_kernel: exclusive SIMDs for the longest fits)
  L.G64 = o; o = align16(o + 3 * K * gstr * 8);
  L.G32 = o; o = align16(o + (3 * K + 1) * nq * 4);
  L.Wt = o; o = align16(o + (N / 16) * (N / 16 + 1) / 2 * 256 * 4);      // waypoint rows' constant block, tile-major
  L.wave0 = o;
  int w = 0;
  L.qs = w; w = align16(w + N * 8);
  L.sp = w; w = align16(w + FIT_PREP_STRIDE * 8);      // the scenario row (fit_device.h PR_*)
  L.big = w;
  const int us_bytes = align16(K * 6 * 8), cf_bytes = (K + 1) * 4 * 16;
  L.cf = w + us_bytes;
  L.cfp = L.cf + cf_bytes;                      // second-order mode: position-block records [K+1][2] float2
  // (coupled groups, fit_groups_kernel: the collision-row records [K+1][nds] float2 take the place of the second-order records)
  int big = us_bytes + cf_bytes + align16((K + 1) * (nds > 2 ? nds : 2) * 8);
  if (CHOL_IMAGE_BYTES(N) > big) big = CHOL_IMAGE_BYTES(N);
  w = align16(w + big);
  L.wave_stride = w;
  L.total = o + wpb * w;
  return L;
}
static bool pick_fused_layout(int K, int nq, int N, int *wpb, int nds = 0) {
  if (K > 64) return false;      // the fused kernel keeps one sample per lane
  for (int w = FIT_LM_WPB_MAX; w >= 4; --w)
    if (fused_lds_layout(K, nq, N, w, nds).total <= FIT_LDS_BYTES) { *wpb = w; return true; }
  return false;
}

// Loads of per-fit state that another wavefront of the same launch may have written (a fit that yielded its wavefront is
// resumed by whoever pops it from the ring, possibly on another XCD): device-scope atomic loads, after an acquire fence.
__device__ __forceinline__ double ld_dev(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_dev(const int32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_dev(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE = D2D_LM_MODE_MINPACK: phase 0 is MINPACK's lmder (fit_phases.h mp_*; Gauss-Newton rows, trust region), phase 1 the
// second-order loop that finishes it; D2D_LM_MODE_FAST: the second-order loop alone, from the start, with the so_lambda rule.
// One pass of the inner loop = at most one factorisation and at most one trial point, whatever the phase: ONE inlined copy of
// damped_solve and ONE of the trial's phase 1 serve lmpar's Gauss-Newton and damped solves, lmder's trial, the finish's
// damped solve and its full and shortened trials (the loop nest is ~35 kB of code as it is; the instruction cache holds 64).
template <int NB, int NQ, bool STAMPS, int MODE>
__global__ void __launch_bounds__(64 * FIT_LM_WPB_MAX)
fit_lm_kernel(int B, FitGeom g, FusedLds L, d2d_fit_opts opts, int iter_cap,
              const double *__restrict__ gG64, const double *__restrict__ pk,
              const float *__restrict__ gG32, const float *__restrict__ gWt,
              const double *__restrict__ prep, double *q_io, double *cost_io,
              double *g_io, double *lm, int32_t *flags,
              int32_t *queue, int32_t *ring, int ring_mask, unsigned long long *__restrict__ stamps,
              const int32_t *__restrict__ order, int prio_at) {
  // iter_cap: a fit stops (state saved, still D2D_ST_RUNNING) once it has used this many iterations in total (the host's
  // convergence poll, d2d_fit_iterate).  order != NULL: hand-out position i takes trajectory order[i] (d2d_fit_plan_set_order).
  // prio_at: a fit that has used this many iterations raises its wave's priority (s_setprio): the stragglers that decide when
  // the launch ends get the SIMD's issue slots ahead of the co-resident wave instead of sharing them.
  // stamps != NULL (D2D_LM_STAMPS=1, diagnostics only): per-phase wave-cycle totals, see launch_lm
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0, st_solve[5] = {0, 0, 0, 0, 0};
#define LM_STAMP(i)                                                     \
  if (STAMPS) {                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
    st_acc[i] += t_ - st_last;                                          \
    st_last = t_;                                                       \
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int N = 16 * NB, NT = NB * (NB + 1) / 2;
  double *G64 = reinterpret_cast<double *>(lds + L.G64);
  stage(G64, gG64, 3 * g.K * g.gstr * 8);
  stage(lds + L.G32, gG32, (3 * g.K + 1) * g.nq * 4);
  stage(lds + L.Wt, gWt, NT * 256 * 4);
  __syncthreads();
  const float *Wt = reinterpret_cast<const float *>(lds + L.Wt);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned char *wl = lds + L.wave0 + wave * L.wave_stride;
  double *qs = reinterpret_cast<double *>(wl + L.qs);
  double *sp = reinterpret_cast<double *>(wl + L.sp);
  double *us = reinterpret_cast<double *>(wl + L.big);
  f32x4 *cf = reinterpret_cast<f32x4 *>(wl + L.cf);
  float2 *cfp = reinterpret_cast<float2 *>(wl + L.cfp);
  float *big = reinterpret_cast<float *>(wl + L.big);          // image of J^T J, then of its Cholesky factor
  const int n = 2 * g.nq;
  const bool act = lane < n;

  // Work distribution.  The first trajectory of a wave is static and workgroup-major (wave w of workgroup g takes hand-out
  // position g + gridDim*w: a batch smaller than the grid's wave count spreads over all CUs, and over the four SIMDs of a CU);
  // the positions beyond come from a device-wide counter (queue[0]).  A fit that has run opts.slice iterations while others
  // are waiting -- positions not yet handed out, or fits in the ring -- saves its state, goes to the back of the ring
  // (queue[2] = head, queue[3] = tail, queue[4] = entries ready to be taken, ring[] = trajectory indices, -1 = empty slot) and its wave takes the next waiting
  // fit: every fit of the batch advances at the same rate (processor sharing), so the launch ends when the work is done or
  // when its longest fit is, whichever is later, without knowing the lengths in advance.  A wave leaves when nothing waits;
  // a wave only pushes when something waits and pops right after, so no entry is left behind.  queue[1] counts the waves that
  // have left: the last one zeroes the counters for the next launch.
  const int stride = gridDim.x * (blockDim.x >> 6);
  auto take = [&](bool first) -> int {          // trajectory index, or -1: nothing waits
    if (first) {
      const int bi = blockIdx.x + gridDim.x * wave;
      if (bi >= B) return -1;
      return order ? __builtin_amdgcn_readfirstlane(order[bi]) : bi;
    }
    int t = -1;
    if (lane == 0) {
      if (stride + __hip_atomic_load(queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B) {
        const int p = stride + atomicAdd(queue, 1);
        if (p < B) t = order ? order[p] : p;
      }
      if (t < 0) {
        // (fetch-adds only: a compare-and-swap on the head under 2048 waves that reach their slice boundary together is quadratic)
        // A failed taker's transient -1 on the count can hide an entry from a taker that comes between its two adds, so a wave only
        // gives up when the ring is EMPTY by the tickets: no push begun (queue[3]) that a take has not begun for (queue[2]).
        // The count is only decremented when a plain load shows it positive: waves that WAIT here (a push has begun, nothing is
        // ready yet) must not keep the count negative with their own transient decrements -- with hundreds of idle waves cycling
        // through (-1, +1) the one entry that then arrives is never seen positive by anybody, and the launch never ends
        // (tests/test_gpu_fullsize.py::test_time_sliced_handout_is_bit_identical hung on exactly that, round 4).
        // (the wait is bounded -- ~0.5 s -- so that the grid drains whatever happens: a fit left in the ring keeps D2D_ST_RUNNING,
        // d2d_fit_solve / d2d_fit_iterate count those and launch again)
        for (int spin = 0;; ++spin) {
          if (__hip_atomic_load(queue + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
            const int a = atomicAdd(queue + 4, -1);
            if (a > 0) {
              const int h = atomicAdd(queue + 2, 1);            // an entry is ours: completed pushes >= successful takes
              int32_t *slot = ring + (h & ring_mask);
              int v;
              while ((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(1);
              __hip_atomic_store(slot, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              t = v | 0x40000000;                               // (bit 30: resumed from the ring)
              break;
            }
            atomicAdd(queue + 4, 1);
          }
          const int pend = __hip_atomic_load(queue + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                           __hip_atomic_load(queue + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pend <= 0) break;
          if (spin > (1 << 18)) { atomicAdd(queue + 5, 1); break; }      // gave up with a push pending: counted, d2d_fit_finish reports it
          __builtin_amdgcn_s_sleep(16);
        }
      }
    }
    return __builtin_amdgcn_readfirstlane(t);
  };
  auto others_waiting = [&]() -> bool {
    int w = 0;
    if (lane == 0) {
      w = (stride + __hip_atomic_load(queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B) ||
          __hip_atomic_load(queue + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
    }
    return __builtin_amdgcn_readfirstlane(w) != 0;
  };
  if (STAMPS) st_last = __builtin_amdgcn_s_memtime();
  bool first_take = true;
  int reserved = -1;                              // the fit a yielding wave secured BEFORE it gave its own to the ring
  for (;;) {
    int b = reserved >= 0 ? reserved : take(first_take);
    reserved = -1;
    first_take = false;
    if (b < 0) break;
    const bool resumed = (b & 0x40000000) != 0;
    b &= 0x3fffffff;
    if (!resumed && flags[4 * b + FL_STATUS] != D2D_ST_RUNNING) continue;      // (a resumed fit's state is read with device-scope loads below)
    const double *prow = prep + (size_t)b * FIT_PREP_STRIDE;
    const double *pkb = pk + (size_t)b * FIT_PK * g.K;
    double *lmb = lm + (size_t)b * LM_STRIDE;
    double qi = 0.0;
    {
      int lane_ld = lane;      // (laundered like the output addresses below: not worth a VGPR pair -- or a spill -- across the whole launch)
      LAUNDER(lane_ld);
      if (act) qi = ld_dev(q_io + (size_t)b * n + lane_ld);
    }
    // Wave-uniform state lives in scalar registers, of which there are ~100: values that are never live together share one.
    //   s0, s1       phase 1: V_lam, V_nu (damping, Nielsen's growth factor); phase 0: V_par, the trust-region radius
    //   u0 .. u7     scratch of one iteration: phase 0 (lmpar + lmder) V_parl, V_paru, V_fp, V_pn, V_gnrm, V_gnorm, V_par;
    //                phase 1 V_pred, V_dmax, V_qmax, V_ct, V_pred_s, V_bt_a, V_bt_b, V_alpha
    double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0, u4 = 0.0, u5 = 0.0, u6 = 0.0, u7 = 1.0;
#define V_lam s0
#define V_nu s1
#define V_mp_par s0
#define V_mp_delta s1
#define V_parl u0
#define V_paru u1
#define V_fp u2
#define V_pn u3
#define V_gnrm u4
#define V_gnorm u5
#define V_par u6
#define V_pred u0
#define V_dmax u1
#define V_qmax u2
#define V_ct u3
#define V_pred_s u4
#define V_bt_a u5
#define V_bt_b u6
#define V_alpha u7
    int iters = uniform_i(ld_dev(flags + 4 * b + FL_ITERS));
    int nev = 0, local = 0, status = D2D_ST_RUNNING;
    bool so_rows = uniform_i(ld_dev(lmb + 3) != 0.0 ? 1 : 0) != 0;      // mode of the rows in us / cf (and of hrow after the MFMA pass)
    int phase = 1;
    MpState mp;
    mp.pgn_lds = nullptr;
    mp.dx_gn = 0.0; mp.t2_gn = 0.0; mp.p_gn = 0.f; mp.gn_valid = 0; mp.gn_ok = 0; mp.first = 0; mp.calm = 0; mp.slow = 0; mp.nfac = 0;
    if (MODE == D2D_LM_MODE_MINPACK) {
      const int pw = uniform_i((int)ld_dev(lmb + 6));                   // phase | first << 1 | calm << 2 | slow << 16
      phase = pw & 1; mp.first = (pw >> 1) & 1; mp.calm = (pw >> 2) & 0x3fff; mp.slow = pw >> 16;
    }
    if (phase == 0) { V_mp_par = uniform_d(ld_dev(lmb + 4)); V_mp_delta = uniform_d(ld_dev(lmb + 5)); }
    else { V_lam = uniform_d(ld_dev(lmb + 0)); V_nu = uniform_d(ld_dev(lmb + 1)); }
    double c = 0.0, gi = 0.0;
    float hdiag = 0.f;
    f32x2 hrow[N / 2];
#pragma unroll
    for (int m = 0; m < N / 2; ++m) hrow[m] = f32x2{0.f, 0.f};
    if (act) qs[q_slot(lane, g.nq)] = qi;
    for (int i = lane; i < FIT_PREP_STRIDE; i += 64) sp[i] = prow[i];
    double pkr[FIT_PK];
#pragma unroll
    for (int cc = 0; cc < FIT_PK; ++cc) pkr[cc] = lane < g.K ? pkb[(size_t)cc * g.K + lane] : 0.0;
    wave_lds_sync();
    if (MODE == D2D_LM_MODE_MINPACK && phase == 0 && mp.first && V_mp_delta <= 0.0) {   // lmder's first radius: factor * ||x||
      const double xn = sqrt(uniform_d(wave_sum(qi * qi)));
      V_mp_delta = xn > 0.0 ? 100.0 * xn : 100.0;
    }
    LM_STAMP(0)
    bool yield = false;
    // Invariant at the top of a pass with sub == 0: us / cf hold phase 1 (rows in mode so_rows) at the point qs (= qi) whose
    // cost is c; `fresh` says they still need phase 2 + MFMA (J^T r, J^T J) before the next factorisation.
    for (bool reenter = true; reenter;) {
      reenter = false;
      c = uniform_d(eval_phase1_reg<NQ>(g, G64, pkr, pkb, sp, qs, us, cf, cfp, so_rows, lane));
      LM_STAMP(1)
      bool fresh = true;
      if (!(fabs(c) <= 1.79e308)) { status = D2D_ST_NONFINITE; fresh = false; }
      // sub-state of an iteration that needs further passes: 1 = lmpar's damped solves (phase 0), 2 = shortened trials (phase 1)
      int sub = 0, lp_it = 0, att = 0;
      V_alpha = 1.0;
      float dl = 0.f, dgi = 0.f;
      bool fin = false, accept = false, so_next = false;
      while (status == D2D_ST_RUNNING || fresh) {
        bool do_solve = false, is_gn = false, do_trial = false;
        double solve_lam = 0.0;
        int isq_mode = 0;
        if (sub == 0) {
          if (fresh) {
            gi = eval_phase2<NQ>(g, G64, us, lane, 0);
            LM_STAMP(2)
            fresh = false;
            if (status != D2D_ST_RUNNING) break;             // accepted + converged: J^T r refreshed, done
            f32x4 acc[NT];
            if (so_rows) jtj_mfma_so<NB, NQ>(g, lds, L.G32, L.wave0 + wave * L.wave_stride + L.cf, L.wave0 + wave * L.wave_stride + L.cfp, lane, acc);
            else jtj_mfma<NB, NQ, true>(g, lds, L.G32, gG32, L.wave0 + wave * L.wave_stride + L.cf, lane, g.K, acc);
            nev += so_rows ? 3 : 2;                          // contracted rows in units of 100 (the roofline unit is 200 rows)
            wave_lds_sync();                                 // every lane is done with us / cf before they are overwritten
            LM_STAMP(3)
            const float ww = (float)(sp[PR_WWP] * sp[PR_WWP]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(ww, Wt[(t * 4 + r) * 64 + lane], acc[t][r]);
            tiles_to_image<N>(acc, big, lane);
            image_put_rhs<N>(big, lane, gi);
            wave_lds_sync();
            image_row<N>(big, lane, hrow);
            hdiag = image_diag<N>(big, lane);
            wave_lds_sync();
            LM_STAMP(4)
          }
          if (iters >= iter_cap || iters >= opts.max_iter) break;
          if (opts.slice > 0 && queue != nullptr && local >= opts.slice && ((local - opts.slice) & 3) == 0 && others_waiting()) {
            // reserve before giving: the wave pushes its own fit only once it HOLDS the next one (a position of the queue or an
            // entry of the ring); if the waiting work went to others meanwhile it simply goes on with its fit
            reserved = take(false);
            if (reserved >= 0) { yield = true; break; }
          }
          if (iters >= prio_at) __builtin_amdgcn_s_setprio(2);
          LM_STAMP(6)
          if (MODE == D2D_LM_MODE_MINPACK && phase == 0) {
            // lmder's test on the scaled gradient, then lmpar
            const double fnorm = sqrt(c);
            double gl = 0.0;
            if (act && hdiag > 0.f && fnorm > 0.0) gl = fabs(gi) / (sqrt((double)hdiag) * fnorm);
            V_gnorm = uniform_d(wave_max(gl));
            if (V_gnorm <= opts.mp_gtol) { status = D2D_ST_CONVERGED; break; }
            V_gnrm = sqrt(uniform_d(wave_sum(gi * gi)));
            V_par = V_mp_par; V_parl = 0.0; V_paru = 0.0; V_fp = 0.0; lp_it = 0; V_alpha = 1.0;
            if (!mp.gn_valid) { do_solve = true; is_gn = true; solve_lam = 0.0; isq_mode = 1; }
            else sub = 3;                                     // (cached Gauss-Newton step: straight to its post-processing)
          } else {
            const double gmax = uniform_d(wave_max(fabs(gi)));
            if (gmax <= opts.gtol) { status = D2D_ST_CONVERGED; break; }
            do_solve = true; solve_lam = V_lam;
          }
        } else if (sub == 1) {                                // lmpar: the next damped solve
          if (V_par == 0.0) V_par = fmax(MP_DWARF, 0.001 * V_paru);
          do_solve = true; solve_lam = V_par; isq_mode = lp_it + 1 < 10 ? 2 : 0;
        } else if (sub == 2) {
          do_trial = true;                                    // a shortened step along the rejected direction
        }
        // ---- at most one factorisation per pass ----
        bool ok = true;
        if (do_solve) {
          double dxn = 0.0, t2 = 0.0;
          float dls;
          const bool unit = MODE == D2D_LM_MODE_MINPACK && phase == 0;
          ok = uniform_i(damped_solve<N, MODE == D2D_LM_MODE_MINPACK, (NQ > 0 && 2 * NQ == N)>(hrow, solve_lam, act, lane, big, dgi, dls, STAMPS ? st_solve : nullptr,
                                                                    unit, isq_mode, V_mp_delta, &dxn, &t2, hdiag, true) ? 1 : 0) != 0;
          LM_STAMP(5)
          if (MODE == D2D_LM_MODE_MINPACK && phase == 0) {
            ++mp.nfac;
            if (is_gn) {
              mp.gn_ok = ok ? 1 : 0; mp.p_gn = dls; mp.dx_gn = dxn; mp.t2_gn = t2; mp.gn_valid = 1;
              sub = 3;
            } else {
              ++lp_it;
              if (!ok) {                                      // cannot happen in exact arithmetic (V_par > 0): raise the damping
                V_parl = fmax(V_parl, V_par); V_par = fmax(2.0 * V_par, 0.001 * V_paru);
                if (lp_it >= 10) { dl = 0.f; V_pn = 0.0; do_trial = true; }
              } else {
                const double temp = V_fp;
                V_fp = dxn - V_mp_delta;
                if (fabs(V_fp) <= 0.1 * V_mp_delta || (V_parl == 0.0 && V_fp <= temp && temp < 0.0) || lp_it == 10) { dl = dls; V_pn = dxn; do_trial = true; }
                else {
                  const double parc = (V_fp / V_mp_delta) / t2;
                  if (V_fp > 0.0) V_parl = fmax(V_parl, V_par);
                  if (V_fp < 0.0) V_paru = fmin(V_paru, V_par);
                  V_par = fmax(V_parl, V_par + parc);
                }
              }
            }
          } else {
            // second-order / FAST loop: the full step first (att 0), then up to two shortened ones along it
            dl = dls;
            const double delta = (double)dl;
            so_next = MODE == D2D_LM_MODE_MINPACK ? true : (opts.so_lambda > 0.0 && V_lam <= opts.so_lambda);
            V_pred = uniform_d(wave_sum(delta * (V_lam * (double)dgi * delta - gi)));
            V_dmax = uniform_d(wave_max(fabs(delta))); V_qmax = uniform_d(wave_max(fabs(qi)));
            V_ct = 0.0; V_pred_s = V_pred; V_alpha = 1.0; V_bt_a = 0.0; V_bt_b = 0.0; fin = false; accept = false; att = 0;
            if (ok) do_trial = true;
          }
        }
        if (MODE == D2D_LM_MODE_MINPACK && sub == 3) {        // lmpar after the Gauss-Newton step (fresh or cached)
          if (mp.gn_ok && mp.dx_gn - V_mp_delta <= 0.1 * V_mp_delta) { dl = mp.p_gn; V_pn = mp.dx_gn; V_par = 0.0; do_trial = true; }
          else {
            V_fp = mp.gn_ok ? mp.dx_gn - V_mp_delta : 1.79e308;
            V_parl = (mp.gn_ok && mp.t2_gn > 0.0) ? (V_fp / V_mp_delta) / mp.t2_gn : 0.0;
            V_paru = V_gnrm / V_mp_delta;
            if (V_paru == 0.0) V_paru = MP_DWARF / fmin(V_mp_delta, 0.1);
            V_par = fmin(fmax(V_par, V_parl), V_paru);
            if (V_par == 0.0) V_par = mp.gn_ok ? V_gnrm / mp.dx_gn : 0.0;
            sub = 1;
          }
        }
        if (!do_trial) {
          if (MODE == D2D_LM_MODE_MINPACK && phase == 0) continue;      // lmpar goes on
          if (sub == 0 && !ok) {                                        // failed factorisation: no trial, the damping grows
            const StepOutcome so = lm_update(false, false, false, 1.0, c, 0.0, V_pred, V_pred, V_dmax, V_qmax, V_lam, V_nu, opts);
            ++iters; ++local;
            V_lam = so.lam; V_nu = so.nu; status = so.status;
            LM_STAMP(6)
          }
          continue;
        }
        // ---- at most one trial point per pass: a full phase 1 (if the step is accepted its rows are the next evaluation) ----
        if (MODE == D2D_LM_MODE_MINPACK && phase == 0 && mp.first) { V_mp_delta = fmin(V_mp_delta, V_pn); mp.first = 0; }
        const bool so_trial = (MODE == D2D_LM_MODE_MINPACK && phase == 0) ? false : so_next;
        if (act) qs[q_slot(lane, g.nq)] = qi + V_alpha * (double)dl;
        wave_lds_sync();
        const double ca = uniform_d(eval_phase1_reg<NQ>(g, G64, pkr, pkb, sp, qs, us, cf, cfp, so_trial, lane));
        LM_STAMP(1)
        if (MODE == D2D_LM_MODE_MINPACK && phase == 0) {
          // lmder: actual against predicted reduction, radius and damping updates, convergence tests
          const double fnorm = sqrt(c);
          const bool ctfin = fabs(ca) <= 1.79e308;
          const double fnorm1 = ctfin ? sqrt(ca) : 1.79e308;
          double actred = -1.0;
          if (0.1 * fnorm1 < fnorm) actred = 1.0 - ca / c;
          const double pg = -uniform_d(wave_sum((double)dl * gi));         // p^T J^T f with p = -dl
          const double jp2 = fmax(pg - V_par * V_pn * V_pn, 0.0);                // ||J p||^2 = p^T g - V_par p^T p
          const double t1 = jp2 / c, t2v = V_par * V_pn * V_pn / c;
          const double prered = t1 + t2v / 0.5, dirder = -(t1 + t2v);
          const double ratio = prered != 0.0 ? actred / prered : 0.0;
          if (ratio <= 0.25) {
            double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
            if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
            V_mp_delta = temp * fmin(V_mp_delta, V_pn / 0.1);
            V_par = V_par / temp;
          } else if (V_par == 0.0 || ratio >= 0.75) {
            V_mp_delta = V_pn / 0.5;
            V_par = 0.5 * V_par;
          }
          V_mp_par = V_par;
          const bool taken = ratio >= 1e-4;
          if (taken) {
            qi += (double)dl; c = ca;
            mp.gn_valid = 0;
            mp.calm = (V_par == 0.0 && ratio >= 0.75) ? mp.calm + 1 : 0;
            fresh = true;
          }
          mp.slow = fabs(actred) <= D2D_LM_MP_SLOW_TOL ? mp.slow + 1 : 0;
          ++iters; ++local;
          sub = 0;
          const double xnorm = sqrt(uniform_d(wave_sum(qi * qi)));
          int info = 0;
          if (fabs(actred) <= opts.mp_ftol && prered <= opts.mp_ftol && 0.5 * ratio <= 1.0) info = 1;
          if (V_mp_delta <= opts.mp_xtol * xnorm) info = 2;
          if (info == 0) {
            if (fabs(actred) <= MP_EPSMCH && prered <= MP_EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            else if (V_mp_delta <= MP_EPSMCH * xnorm) info = 7;
            else if (V_gnorm <= MP_EPSMCH) info = 8;
          }
          if (info != 0) status = D2D_ST_CONVERGED;
          if (taken && status == D2D_ST_RUNNING && opts.mp_finish > 0 && (mp.calm >= opts.mp_finish || (opts.mp_slow > 0 && mp.slow >= opts.mp_slow))) {
            // the trust region has been inactive for mp_finish steps, or lmder has stagnated for mp_slow trials (include/d2d.h):
            // second-order finish from here (rows of this point in that mode)
            phase = 1; V_lam = D2D_LM_LAMBDA0; V_nu = 2.0; so_rows = true;
            reenter = true;
            break;
          }
          LM_STAMP(6)
          continue;
        }
        // second-order / FAST loop
        bool decided = false;
        if (att == 0) {
          V_ct = ca;
          fin = (fabs(V_ct) <= 1.79e308) && (V_pred > 0.0);
          if (fin && (c - V_ct) / V_pred > 0.0) { accept = true; decided = true; }
          else if (!fin) decided = true;
          else {
            V_bt_a = uniform_d(-2.0 * wave_sum(gi * (double)dl)); V_bt_b = V_bt_a - V_pred;
            V_alpha = bt_first_alpha(V_bt_a, c, V_ct);
          }
        } else {
          if ((fabs(ca) <= 1.79e308) && ca < c) { accept = true; V_ct = ca; V_pred_s = V_bt_a * V_alpha - V_bt_b * V_alpha * V_alpha; decided = true; }
          else V_alpha = fmax(D2D_LM_BT_SHRINK * V_alpha, D2D_LM_BT_FLOOR);
        }
        ++att;
        if (!decided && att < 3) { sub = 2; continue; }
        sub = 0;
        const StepOutcome so = lm_update(true, fin, accept, accept ? V_alpha : 1.0, c, V_ct, V_pred, V_pred_s, V_dmax, V_qmax, V_lam, V_nu, opts);
        ++iters; ++local;
        V_lam = so.lam; V_nu = so.nu; status = so.status;
        if (so.accept) {
          qi += V_alpha * (double)dl; c = V_ct;
          fresh = true;                                    // also when converged: refresh J^T r
          so_rows = so_next;
        }
        V_alpha = 1.0;
        LM_STAMP(6)
      }
    }
    if (status == D2D_ST_RUNNING && iters >= opts.max_iter) status = D2D_ST_MAXITER;
    const double gmax = uniform_d(wave_max(fabs(gi)));
    // The state a resuming wave reads (q, the lm words, the iteration / evaluation counters) is stored with device-scope
    // atomic stores: they go through to memory on their own (the eight XCDs have separate L2s), so handing a fit over needs no
    // L2 write-back / invalidate -- a device-scope fence per yield flushed the L2 under 2048 waves and cost 30x the solve.
    {
      int lane_io = lane;      // (laundered: the per-lane output addresses are not worth two VGPR pairs held across the whole LM loop)
      LAUNDER(lane_io);
      if (act) {
        st_dev(q_io + (size_t)b * n + lane_io, qi);
        if (!yield) g_io[(size_t)b * n + lane_io] = gi;
      }
    }
    if (lane == 0) {
      if (!yield) cost_io[b] = c;
      st_dev(lmb + 2, gmax); st_dev(lmb + 3, so_rows ? 1.0 : 0.0);
      if (phase == 0) { st_dev(lmb + 4, V_mp_par); st_dev(lmb + 5, V_mp_delta); } else { st_dev(lmb + 0, V_lam); st_dev(lmb + 1, V_nu); }
      if (MODE == D2D_LM_MODE_MINPACK) { st_dev(lmb + 6, (double)(phase | (mp.first << 1) | ((mp.calm & 0x3fff) << 2) | (mp.slow << 16))); st_dev(lmb + 7, ld_dev(lmb + 7) + (double)mp.nfac); }
      st_dev(flags + 4 * b + FL_STATUS, status); st_dev(flags + 4 * b + FL_ITERS, iters); st_dev(flags + 4 * b + FL_NEED, 1);
      st_dev(flags + 4 * b + FL_NEVAL, ld_dev(flags + 4 * b + FL_NEVAL) + nev);
    }
    __builtin_amdgcn_s_setprio(0);
    if (yield) {                                 // to the back of the ring: the stores above have completed before the index is published
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) {
        const int tl = atomicAdd(queue + 3, 1);
        int32_t *slot = ring + (tl & ring_mask);
        while (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(slot, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        atomicAdd(queue + 4, 1);                              // the entry is in its slot: one more to take
      }
    }
    LM_STAMP(0)
  }
  if (queue != nullptr && lane == 0) {
    if (atomicAdd(queue + 1, 1) == stride - 1) { queue[0] = 0; queue[1] = 0; queue[2] = 0; queue[3] = 0; queue[4] = 0; }     // every wave has stopped pulling
  }
  if (STAMPS && lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&stamps[i], st_acc[i]);
    for (int i = 0; i < 5; ++i) atomicAdd(&stamps[8 + i], st_solve[i]);
  }
#undef LM_STAMP
#undef V_lam
#undef V_nu
#undef V_mp_par
#undef V_mp_delta
#undef V_parl
#undef V_paru
#undef V_fp
#undef V_pn
#undef V_gnrm
#undef V_gnorm
#undef V_par
#undef V_pred
#undef V_dmax
#undef V_qmax
#undef V_ct
#undef V_pred_s
#undef V_bt_a
#undef V_bt_b
#undef V_alpha
}

// ------------------------------------------------------------------------------------
// Coupled groups (BASELINE configs[2]) in ONE persistent launch: a wavefront takes a whole scenario (group of n_ac aircraft,
// consecutive trajectories) and runs the block Gauss-Seidel of oracle/fit.py bgs_solve on it by itself -- sweep after sweep,
// aircraft after aircraft, each visit a restarted LM solve of at most inner_iters iterations with the partners' sampled
// positions frozen -- until ITS OWN scenario stops moving (largest relative move of a sweep <= tol) or max_sweeps.  No
// cross-wave synchronisation, no host round trips; the positions table pos [B][2][K] lives in HBM (L2) and is read and
// written by the owning wave only.  The launch-pair driver ran every scenario as long as the slowest one and paid two
// launches per iteration (7 700 launch pairs at 8 x 8192); a scenario that settles after 30 sweeps now costs 30.
template <int NB, int NQ>
__global__ void __launch_bounds__(64 * FIT_LM_WPB_MAX)
fit_groups_kernel(int R, int n_ac, int nds, FitGeom g, FusedLds L, d2d_fit_opts opts, int max_sweeps, int inner_iters, double tol,
                  const double *__restrict__ gG64, const double *__restrict__ pk, const float *__restrict__ gG32,
                  const float *__restrict__ gWt, const double *__restrict__ prep, double *q_io, double *pos,
                  double *__restrict__ cost_out, double *__restrict__ g_out, int32_t *__restrict__ flags,
                  int32_t *__restrict__ sweeps_out, double *__restrict__ moved_out, int32_t *__restrict__ queue,
                  const int32_t *__restrict__ order, double *ls_hist, int ls_s0, double ls_r0, int prio_at) {
  // ls_s0 > 0: line search on the joint cost along the direction of a slow sweep, from sweep ls_s0 on (see below); ls_hist
  // [B][GROUPS_LS_STRIDE] doubles: per aircraft the unknowns before the sweep [48] and the trial point [48]
  // order != NULL: hand-out position i takes scenario order[i] (d2d_fit_plan_set_group_order: the scenarios that swept longest
  // in a previous solve start first, so the tail of the launch is not one late straggler)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int N = 16 * NB, NT = NB * (NB + 1) / 2;
  double *G64 = reinterpret_cast<double *>(lds + L.G64);
  stage(G64, gG64, 3 * g.K * g.gstr * 8);
  stage(lds + L.G32, gG32, (3 * g.K + 1) * g.nq * 4);
  stage(lds + L.Wt, gWt, NT * 256 * 4);
  __syncthreads();
  const float *Wt = reinterpret_cast<const float *>(lds + L.Wt);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int woff = L.wave0 + wave * L.wave_stride;
  unsigned char *wl = lds + woff;
  double *qs = reinterpret_cast<double *>(wl + L.qs);
  double *sp = reinterpret_cast<double *>(wl + L.sp);
  double *us = reinterpret_cast<double *>(wl + L.big);
  f32x4 *cf = reinterpret_cast<f32x4 *>(wl + L.cf);
  float *big = reinterpret_cast<float *>(wl + L.big);
  const int n = 2 * g.nq;
  const bool act = lane < n;
  const int stride = gridDim.x * (blockDim.x >> 6);
  auto next_index = [&](int r) -> int {
    int t = 0;
    if (lane == 0) t = stride + atomicAdd(queue, 1);
    return __builtin_amdgcn_readfirstlane(t);
  };
  for (int ri = blockIdx.x + gridDim.x * wave; (unsigned)ri < (unsigned)R; ri = next_index(ri)) {
    const int r = order ? __builtin_amdgcn_readfirstlane(order[ri]) : ri;
    const int gbase = r * n_ac;
    // sampled positions of every aircraft of the group (at the start, and after an extrapolated step)
    // (src / sstride: the unknowns of trajectory b are src[b * sstride + 0 .. n) -- q_io, or the trial points of the line search)
    auto publish_positions = [&](const double *src, size_t sstride) {
      for (int a = 0; a < n_ac; ++a) {
        const int b = gbase + a;
        const double *pkb = pk + (size_t)b * FIT_PK * g.K;
        const double *qb = src + (size_t)b * sstride;
        double x = 0.0, y = 0.0;
        if (lane < g.K) {
          x = pkb[lane]; y = pkb[g.K + lane];
          const double *g0 = G64 + (size_t)lane * g.gstr;
          for (int j = 0; j < g.nq; ++j) { x = fma(g0[j], qb[j], x); y = fma(g0[j], qb[g.nq + j], y); }
          pos[(size_t)b * 2 * g.K + lane] = x; pos[(size_t)b * 2 * g.K + g.K + lane] = y;
        }
      }
    };
    publish_positions(q_io, n);
    __threadfence_block();
    int sweep = 0, nev_total = 0;
    double moved = 0.0;
    bool settled = false;
    // Line search on the JOINT cost along a slow sweep (oracle/fit.py bgs_solve, ls_s0 / ls_r0).  Block Gauss-Seidel is a descent
    // method on F(X) = sum of the aircraft's own rows + every coupled pair once; the slow scenarios of BASELINE configs[2] converge
    // LINEARLY at 0.9 .. 0.93 per sweep, and the slowest DRIFTS away from a saddle of F with moves that grow by 0.5 % per sweep for
    // > 100 sweeps (tools/dev_groups_trace.py).  After a sweep (from sweep ls_s0 on) that moved at least ls_r0 x the move of the
    // sweep before, F is evaluated along the sweep's own direction d = X_k - X_{k-1}: first length rho / (1 - rho) (the fixed point
    // of a single linear mode; 8 when the moves grow), doubled while F falls (up to 64), one shorter try (x 1/4) when the first
    // does not lower F; the best point with F below F(X_k) is taken, otherwise nothing changes.  The stop test stays the move of
    // a PLAIN sweep, and the iteration stays a descent on F -- unlike the Anderson extrapolation of the sweep map (a root finder of
    // G(X) - X, measured in round 4: attracted by the repelling fixed points that the plain sweeps leave), it cannot be drawn to
    // a saddle.  F through the cost-only phase 1 of every aircraft against the published positions: c_a counts a's collision rows
    // against all partners, so F = sum_a (c_a - coll_a / 2).
    auto joint_merit = [&](const double *src, size_t sstride) -> double {
      double F = 0.0;
      for (int a = 0; a < n_ac; ++a) {
        const int b = gbase + a;
        const double *prow = prep + (size_t)b * FIT_PREP_STRIDE;
        const double *pkb = pk + (size_t)b * FIT_PK * g.K;
        const GroupCtx gc{pos, n_ac, a, gbase, nds};
        if (act) qs[q_slot(lane, g.nq)] = src[(size_t)b * sstride + lane];
        for (int i = lane; i < FIT_PREP_STRIDE; i += 64) sp[i] = prow[i];
        double pkr[FIT_PK];
#pragma unroll
        for (int cc = 0; cc < FIT_PK; ++cc) pkr[cc] = lane < g.K ? pkb[(size_t)cc * g.K + lane] : 0.0;
        wave_lds_sync();
        double coll = 0.0;
        const double c = uniform_d(eval_cost_grp<NQ>(g, G64, pkr, pkb, sp, qs, gc, lane, coll));
        F += c - 0.5 * uniform_d(coll);
        wave_lds_sync();
      }
      return F;
    };
    double moved_prev = 1e300;
    for (sweep = 1; sweep <= max_sweeps; ++sweep) {
      moved = 0.0;
      if (sweep == prio_at) __builtin_amdgcn_s_setprio(2);      // a scenario that is still sweeping decides when the launch ends
      for (int a = 0; a < n_ac; ++a) {
        const int b = gbase + a;
        const double *prow = prep + (size_t)b * FIT_PREP_STRIDE;
        const double *pkb = pk + (size_t)b * FIT_PK * g.K;
        const GroupCtx gc{pos, n_ac, a, gbase, nds};
        double qi = act ? q_io[(size_t)b * n + lane] : 0.0;
        double lam = D2D_LM_LAMBDA0, nu = 2.0;
        int iters = 0, status = D2D_ST_RUNNING;
        double c = 0.0, gi = 0.0;
        f32x2 hrow[N / 2];
        float hdiag = 0.f;
#pragma unroll
        for (int m = 0; m < N / 2; ++m) hrow[m] = f32x2{0.f, 0.f};
        if (act) qs[q_slot(lane, g.nq)] = qi;
        for (int i = lane; i < FIT_PREP_STRIDE; i += 64) sp[i] = prow[i];
        double pkr[FIT_PK];
#pragma unroll
        for (int cc = 0; cc < FIT_PK; ++cc) pkr[cc] = lane < g.K ? pkb[(size_t)cc * g.K + lane] : 0.0;
        wave_lds_sync();
        c = uniform_d(eval_phase1_grp<NQ>(g, G64, pkr, pkb, sp, qs, us, cf, gc, lane));
        bool fresh = true;
        if (!(fabs(c) <= 1.79e308)) { status = D2D_ST_NONFINITE; fresh = false; }
        while (status == D2D_ST_RUNNING || fresh) {
          if (fresh) {
            gi = eval_phase2<NQ>(g, G64, us, lane, 0);
            fresh = false;
            if (status != D2D_ST_RUNNING) break;
            // (a visit that finds its aircraft stationary -- most visits of the late sweeps -- needs no Hessian)
            if (iters >= inner_iters) break;
            if (uniform_d(wave_max(fabs(gi))) <= opts.gtol) { status = D2D_ST_CONVERGED; break; }
            f32x4 acc[NT];
            jtj_mfma<NB, NQ, true>(g, lds, L.G32, gG32, woff + L.cf, lane, g.K, acc);
            nev_total += 2;
            wave_lds_sync();
            const float ww = (float)(sp[PR_WWP] * sp[PR_WWP]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) acc[t][rr] = fmaf(ww, Wt[(t * 4 + rr) * 64 + lane], acc[t][rr]);
            tiles_to_image<N>(acc, big, lane);
            image_put_rhs<N>(big, lane, gi);
            wave_lds_sync();
            image_row<N>(big, lane, hrow);
            hdiag = image_diag<N>(big, lane);
            wave_lds_sync();
          }
          if (iters >= inner_iters) break;
          const double gmax = uniform_d(wave_max(fabs(gi)));
          if (gmax <= opts.gtol) { status = D2D_ST_CONVERGED; break; }
          float dgi, dl;
          const int ok = uniform_i(damped_solve<N, false, (NQ > 0 && 2 * NQ == N)>(hrow, lam, act, lane, big, dgi, dl, nullptr, false, 0, 0.0, nullptr, nullptr, hdiag, true) ? 1 : 0);
          const double delta = (double)dl;
          const double pred = uniform_d(wave_sum(delta * (lam * (double)dgi * delta - gi)));
          const double dmax = uniform_d(wave_max(fabs(delta))), qmax = uniform_d(wave_max(fabs(qi)));
          double ct = 0.0, pred_s = pred, alpha = 1.0, bt_a = 0.0, bt_b = 0.0;
          bool fin = false, accept = false;
          if (ok) {
            for (int att = 0; att < 3; ++att) {
              if (act) qs[q_slot(lane, g.nq)] = qi + alpha * delta;
              wave_lds_sync();
              const double ca = uniform_d(eval_phase1_grp<NQ>(g, G64, pkr, pkb, sp, qs, us, cf, gc, lane));
              if (att == 0) {
                ct = ca;
                fin = (fabs(ct) <= 1.79e308) && (pred > 0.0);
                if (fin && (c - ct) / pred > 0.0) { accept = true; break; }
                if (!fin) break;
                bt_a = uniform_d(-2.0 * wave_sum(gi * delta)); bt_b = bt_a - pred;
                alpha = bt_first_alpha(bt_a, c, ct);
              } else {
                if ((fabs(ca) <= 1.79e308) && ca < c) { accept = true; ct = ca; pred_s = bt_a * alpha - bt_b * alpha * alpha; break; }
                alpha = fmax(D2D_LM_BT_SHRINK * alpha, D2D_LM_BT_FLOOR);
              }
            }
          }
          const StepOutcome so = lm_update(ok != 0, fin, accept, accept ? alpha : 1.0, c, ct, pred, pred_s, dmax, qmax, lam, nu, opts);
          ++iters;
          lam = so.lam; nu = so.nu; status = so.status;
          if (so.accept) { qi += alpha * delta; c = ct; fresh = true; }
        }
        // the visit is over: store q, publish this aircraft's positions for the partners, account the move
        {
          int lane_io = lane;      // (laundered address; the start value is read back instead of being held across the LM loop)
          LAUNDER(lane_io);
          const double q_start = act ? q_io[(size_t)b * n + lane_io] : 0.0;
          const double dq = uniform_d(wave_max(fabs(qi - q_start))), qa = uniform_d(wave_max(fabs(q_start)));
          moved = fmax(moved, dq / (1.0 + qa));
          if (act) { q_io[(size_t)b * n + lane_io] = qi; qs[q_slot(lane, g.nq)] = qi; }
          if (ls_s0 > 0 && act) ls_hist[(size_t)b * GROUPS_LS_STRIDE + lane_io] = q_start;
        }
        wave_lds_sync();
        if (lane < g.K) {
          double Y[6];
          flat_outputs_pk<NQ>(g, G64, qs, pkr, lane, Y);
          pos[(size_t)b * 2 * g.K + lane] = Y[0]; pos[(size_t)b * 2 * g.K + g.K + lane] = Y[1];
        }
        __threadfence_block();
      }
      if (moved <= tol) { settled = true; break; }
      if (ls_s0 > 0 && sweep >= ls_s0 && sweep < max_sweeps && moved >= ls_r0 * moved_prev) {
        const double rho = moved / moved_prev;
        double al = rho < 1.0 ? rho / fmax(1.0 - rho, 1e-3) : D2D_GS_LS_FIRST_MAX;
        al = fmin(fmax(al, 1.0), D2D_GS_LS_FIRST_MAX);
        __threadfence_block();
        double bestF = joint_merit(q_io, n), best_al = 0.0;      // (the published positions are those of q_io)
        bool shrunk = false;
        double *trial = ls_hist + N;                             // trajectory b: trial[b * GROUPS_LS_STRIDE + 0 .. n)
        for (int t = 0; t < 5 && al <= D2D_GS_LS_MAX; ++t) {
          if (act) {
            for (int a = 0; a < n_ac; ++a) {
              const size_t b = (size_t)(gbase + a);
              const double qk = q_io[b * n + lane];
              trial[b * GROUPS_LS_STRIDE + lane] = fma(al, qk - ls_hist[b * GROUPS_LS_STRIDE + lane], qk);
            }
          }
          __threadfence_block();
          publish_positions(trial, GROUPS_LS_STRIDE);
          __threadfence_block();
          const double Ft = joint_merit(trial, GROUPS_LS_STRIDE);
          const bool better = Ft < bestF;                        // (false for NaN)
          if (better) { bestF = Ft; best_al = al; }
          if (t == 0) {
            if (better) al *= 2.0; else { al *= 0.25; shrunk = true; }
          } else {
            if (shrunk || !better) break;
            al *= 2.0;
          }
        }
        if (best_al > 0.0 && act) {
          for (int a = 0; a < n_ac; ++a) {
            const size_t b = (size_t)(gbase + a);
            const double qk = q_io[b * n + lane];
            q_io[b * n + lane] = fma(best_al, qk - ls_hist[b * GROUPS_LS_STRIDE + lane], qk);
          }
        }
        __threadfence_block();
        publish_positions(q_io, n);
        __threadfence_block();
      }
      moved_prev = moved;
    }
    if (sweep > max_sweeps) sweep = max_sweeps;
    __builtin_amdgcn_s_setprio(0);
    // per-aircraft sub-problem cost and gradient with everybody's final positions
    for (int a = 0; a < n_ac; ++a) {
      const int b = gbase + a;
      const double *prow = prep + (size_t)b * FIT_PREP_STRIDE;
      const double *pkb = pk + (size_t)b * FIT_PK * g.K;
      const GroupCtx gc{pos, n_ac, a, gbase, nds};
      const double qi = act ? q_io[(size_t)b * n + lane] : 0.0;
      if (act) qs[q_slot(lane, g.nq)] = qi;
      for (int i = lane; i < FIT_PREP_STRIDE; i += 64) sp[i] = prow[i];
      double pkr[FIT_PK];
#pragma unroll
      for (int cc = 0; cc < FIT_PK; ++cc) pkr[cc] = lane < g.K ? pkb[(size_t)cc * g.K + lane] : 0.0;
      wave_lds_sync();
      const double c = uniform_d(eval_phase1_grp<NQ>(g, G64, pkr, pkb, sp, qs, us, cf, gc, lane));
      const double gi = eval_phase2<NQ>(g, G64, us, lane, 0);
      wave_lds_sync();
      if (act) g_out[(size_t)b * n + lane] = gi;
      if (lane == 0) {
        cost_out[b] = c;
        flags[4 * b + FL_STATUS] = settled ? D2D_ST_CONVERGED : D2D_ST_MAXITER;
        flags[4 * b + FL_ITERS] = sweep;
        flags[4 * b + FL_NEVAL] = (a == 0) ? nev_total : 0;
      }
    }
    if (lane == 0) { sweeps_out[r] = sweep; moved_out[r] = moved; }
  }
  if (lane == 0) {
    if (atomicAdd(queue + 1, 1) == stride - 1) { queue[0] = 0; queue[1] = 0; }
  }
}

// ------------------------------------------------------------------------------------
// Long horizons (K > 64: the reference's own single-aircraft scenarios have 101 .. 151 nodes, its 50 Hz caches 351 .. 1501):
// the same LM loop with the samples processed in chunks of 64 (lane = sample inside a chunk) and the basis tables read
// from global memory / L2 (fit_phases.h long_*), so that the LDS footprint does not depend on K.  One evaluation =
// for every chunk: rows (phase 1) -> J^T r (phase 2) -> MFMA pass accumulating into the same six tiles.  Trial points
// are cost-only passes (no row records are kept across chunks), an accepted step is followed by a full evaluation.
struct LongLds {
  int Wt, G64, G32, wave0, wave_stride, qs, sp, big, cf, cfp, total;      // G64 / G32: only with the tables in the LDS
};
// tables: the fp64 basis block [3][K][nq+1] and the fp32 planes [(3K+1)][nq] live in the LDS too (K, nq > 0), otherwise in global memory
static LongLds long_lds_layout(int N, int wpb, int K = 0, int nq = 0, bool g32 = true) {
  LongLds L;
  int o = 0;
  L.Wt = o; o = align16(o + (N / 16) * (N / 16 + 1) / 2 * 256 * 4);
  L.G64 = L.G32 = 0;
  if (K > 0) {
    L.G64 = o; o = align16(o + 3 * K * (nq + 1) * 8);
    if (g32) { L.G32 = o; o = align16(o + (3 * K + 1) * nq * 4); }
  }
  L.wave0 = o;
  int w = 0;
  L.qs = w; w = align16(w + N * 8);
  L.sp = w; w = align16(w + FIT_PREP_STRIDE * 8);
  L.big = w;
  const int us_bytes = 64 * 6 * 8, cf_bytes = 65 * 4 * 16, cfp_bytes = align16(65 * 2 * 8);
  L.cf = w + us_bytes;
  L.cfp = L.cf + cf_bytes;
  int big = us_bytes + cf_bytes + cfp_bytes;
  if (CHOL_IMAGE_BYTES(N) > big) big = CHOL_IMAGE_BYTES(N);
  w = align16(w + big);
  L.wave_stride = w;
  L.total = o + wpb * w;
  return L;
}
// Waves per workgroup of the long-horizon kernel with its tables in the LDS (0: they do not fit beside at least FIT_LONG_TL_MIN_WAVES
// per-wave blocks -- the kernel then reads them from global memory / L2 with FIT_LM_WPB_MAX waves)
#define FIT_LONG_TL_MIN_WAVES 3
static int long_tables_waves(int K, int nq, int N, bool g32) {
  for (int w = FIT_LM_WPB_MAX; w >= FIT_LONG_TL_MIN_WAVES; --w)
    if (long_lds_layout(N, w, K, nq, g32).total <= FIT_LDS_BYTES) return w;
  return 0;
}

// LDS of the segment formulation (fit_seg.h): plan constants + per-wave blocks, none of which depends on K
static SegLds seg_lds_layout(int N, int nq, int S, int wpb) {
  SegLds L;
  int o = 0;
  L.Wt = o; o = align16(o + (N / 16) * (N / 16 + 1) / 2 * 256 * 4);
  L.Zl64 = o; o = align16(o + 8 * S * (nq + 1) * 8);
  L.Zl32 = o; o = align16(o + 8 * S * nq * 4);
  L.wave0 = o;
  int w = 0;
  L.qs = w; w = align16(w + N * 8);
  L.sp = w; w = align16(w + FIT_PREP_STRIDE * 8);
  L.zc = w; w = align16(w + 8 * D2D_FIT_MAX_S * 2 * 8);
  L.park = w; w = align16(w + 5 * 64 * 8 + 2 * 64 * 4);   // per-lane constants of the running fit (fit_lm_long_kernel), [5][64], + [2][64] floats of lmder's state
  L.big = w;
  L.cf = w;
  L.cfp = L.cf + SEG_ROWS * 4 * 16;
  L.psi = align16(L.cfp + SEG_ROWS * 2 * 8);
  int big = L.psi + 3 * SEG_ROWS * 8 * 4 - L.big;
  if (CHOL_IMAGE_BYTES(N) > big) big = CHOL_IMAGE_BYTES(N);             // image of J^T J / its Cholesky factor
  if (64 * SEG_RED_STRIDE * 8 > big) big = 64 * SEG_RED_STRIDE * 8;         // moment reduction scratch
  w = align16(w + big);
  L.wave_stride = w;
  L.total = o + wpb * w;
  return L;
}
struct SegArgs {
  SegMap m;
  SegLds L;
  const double *Zl64, *Zlp, *sx;
  const float *Zl32;
  double c1;              // 2 / T
  unsigned long long *stamps;   // D2D_LM_STAMPS=1 (diagnostics): wave-cycle totals per phase, NULL otherwise
};

// TL: the basis tables are staged into the LDS once per workgroup (they fit beside >= 3 per-wave blocks: K <= 121 at nq = 24) and
// the phases read them there like the K <= 64 kernel does -- from global memory every phase of every chunk is a chain of L2
// round trips (measured at K = 121: 287 us per LM iteration and wave with 8 waves per CU against 40 us with 3).
// SEG (round 3, the default): the segment formulation of fit_seg.h -- no K-sized table at all, one MFMA per sample.
// MODE: as fit_lm_kernel (D2D_LM_MODE_MINPACK: lmder's trials through mp_trial, then the second-order loop)
template <int NB, int NQ, bool TL, bool TL32, int MODE, bool SEG = false>
__global__ void __launch_bounds__(64 * FIT_LM_WPB_MAX)
fit_lm_long_kernel(int B, FitGeom g, LongLds L, d2d_fit_opts opts, int iter_budget,
                   const double *__restrict__ GTg, const double *__restrict__ G64gl, const double *__restrict__ pk,
                   const float *__restrict__ gG32, const float *__restrict__ gWt, const double *__restrict__ prep,
                   double *__restrict__ q_io, double *__restrict__ cost_io, double *__restrict__ g_io,
                   double *__restrict__ lm, int32_t *__restrict__ flags, int32_t *__restrict__ queue,
                   const int32_t *__restrict__ order, GroupArgs ga, SegArgs sa) {
  // ga (coupled groups): trajectory of hand-out position i = ga.off + i * ga.stride (one aircraft index of every scenario per
  // launch), collision rows against the positions table ga.pos; uncoupled: {nullptr, 0, 0, 1, 0}
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int N = 16 * NB, NT = NB * (NB + 1) / 2;
  if (SEG) {
    stage(lds + sa.L.Wt, gWt, NT * 256 * 4);
    stage(lds + sa.L.Zl64, sa.Zl64, 8 * sa.m.S * (g.nq + 1) * 8);
    stage(lds + sa.L.Zl32, sa.Zl32, 8 * sa.m.S * g.nq * 4);
  } else {
    stage(lds + L.Wt, gWt, NT * 256 * 4);
    if (TL) {
      stage(lds + L.G64, G64gl, 3 * g.K * g.gstr * 8);
      if (TL32) stage(lds + L.G32, gG32, (3 * g.K + 1) * g.nq * 4);
    }
  }
  __syncthreads();
  const float *Wt = reinterpret_cast<const float *>(lds + (SEG ? sa.L.Wt : L.Wt));
  // phase 1 table / phase 2 table: the LDS copy of the row-major block, or the transposed / row-major global tables
  const double *GT = TL ? reinterpret_cast<const double *>(lds + L.G64) : GTg;
  const double *G64g = TL ? reinterpret_cast<const double *>(lds + L.G64) : G64gl;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int woff = SEG ? sa.L.wave0 + wave * sa.L.wave_stride : L.wave0 + wave * L.wave_stride;
  unsigned char *wl = lds + woff;
  double *qs = reinterpret_cast<double *>(wl + (SEG ? sa.L.qs : L.qs));
  double *sp = reinterpret_cast<double *>(wl + (SEG ? sa.L.sp : L.sp));
  double *us = reinterpret_cast<double *>(wl + L.big);
  f32x4 *cf = reinterpret_cast<f32x4 *>(wl + (SEG ? sa.L.cf : L.cf));
  float2 *cfp = reinterpret_cast<float2 *>(wl + (SEG ? sa.L.cfp : L.cfp));
  float *big = reinterpret_cast<float *>(wl + (SEG ? sa.L.big : L.big));          // image of J^T J, then of its Cholesky factor (aliases us / cf / cfp)
  // segment formulation: Legendre coefficients of the trial point, operand planes, plan constants, this lane's segment
  double *zc = reinterpret_cast<double *>(wl + sa.L.zc);
  float *psi = reinterpret_cast<float *>(wl + sa.L.psi);
  const double *Zl64 = reinterpret_cast<const double *>(lds + sa.L.Zl64);
  const float *Zl32 = reinterpret_cast<const float *>(lds + sa.L.Zl32);
  const LaneSeg ls = lane_segment(sa.m, lane);
  // diagnostics (SEG, sa.stamps != NULL): 0 coefficients, 1 rows of trial points, 2 rows of evaluations, 3 MFMA passes, 4 J^T r, 5 projection,
  // 6 image, 7 everything else (solves, bookkeeping)
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
  const bool st_on = SEG && sa.stamps != nullptr;
#define SEG_STAMP(i) if (st_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; }
  if (st_on) st_last = __builtin_amdgcn_s_memtime();
  const int nq = NQ ? NQ : g.nq, n = 2 * nq;
  const bool act = lane < n;
  const int stride = gridDim.x * (blockDim.x >> 6);
  auto next_index = [&](int b) -> int {
    if (queue == nullptr) return b + stride;
    int t = 0;
    if (lane == 0) t = stride + atomicAdd(queue, 1);
    return __builtin_amdgcn_readfirstlane(t);
  };
  for (int bi = blockIdx.x + gridDim.x * wave; (unsigned)bi < (unsigned)B; bi = next_index(bi)) {
    const int b = ga.off + (order ? __builtin_amdgcn_readfirstlane(order[bi]) : bi) * ga.stride;
    if (flags[4 * b + FL_STATUS] != D2D_ST_RUNNING) continue;
    const GroupCtx gc{ga.pos, ga.n_ac, ga.n_ac > 0 ? b % ga.n_ac : 0, ga.n_ac > 0 ? (b / ga.n_ac) * ga.n_ac : 0, ga.pos ? ga.nds : 0};
    const double *pkb = pk + (size_t)b * FIT_PK * g.K;
    int lp = lane;                     // (the per-fit loads and stores below address with a re-made lane index: the per-lane pointers of the
    LAUNDER(lp);                       // loop's prologue are otherwise formed once per launch and held -- in scratch -- across every fit)
    double qi = lp < n ? q_io[(size_t)b * n + lp] : 0.0;
    double lam = uniform_d(lm[LM_STRIDE * b + 0]), nu = uniform_d(lm[LM_STRIDE * b + 1]);      // (wave-uniform: scalar registers)
    int iters = flags[4 * b + FL_ITERS];
    int nev = 0, local = 0, status = D2D_ST_RUNNING;
    bool so_rows = uniform_i(lm[LM_STRIDE * b + 3] != 0.0 ? 1 : 0) != 0;
    int phase = 1;
    MpState mp;
    mp.pgn_lds = SEG ? reinterpret_cast<float *>(wl + sa.L.park + 5 * 64 * 8) : nullptr;      // (the cached Gauss-Newton step: one register fewer across the trials)
    mp.par = 0.0; mp.delta = 0.0; mp.dx_gn = 0.0; mp.t2_gn = 0.0; mp.p_gn = 0.f; mp.gn_valid = 0; mp.gn_ok = 0; mp.first = 0; mp.calm = 0; mp.slow = 0; mp.nfac = 0;
    if (MODE == D2D_LM_MODE_MINPACK) {
      const int pw = uniform_i((int)lm[LM_STRIDE * b + 6]);            // phase | first << 1 | calm << 2 | slow << 16
      phase = pw & 1; mp.first = (pw >> 1) & 1; mp.calm = (pw >> 2) & 0x3fff; mp.slow = pw >> 16;
      mp.par = uniform_d(lm[LM_STRIDE * b + 4]); mp.delta = uniform_d(lm[LM_STRIDE * b + 5]);
      if (phase == 0 && mp.first && mp.delta <= 0.0) {                   // lmder's first radius: factor * ||x||
        const double xn = sqrt(uniform_d(wave_sum(qi * qi)));
        mp.delta = xn > 0.0 ? 100.0 * xn : 100.0;
      }
    }
    double c = 0.0, gi = 0.0;
    float hdiag = 0.f;
    f32x2 hrow[N / 2];
#pragma unroll
    for (int m = 0; m < N / 2; ++m) hrow[m] = f32x2{0.f, 0.f};
    for (int i = lp; i < FIT_PREP_STRIDE; i += 64) sp[i] = prep[(size_t)b * FIT_PREP_STRIDE + i];
    wave_lds_sync();

    // segment formulation: the end-condition part of this lane's Legendre coefficient pair (lane = (segment, degree)) and chunk 0's
    // inputs, constant for the whole fit.  They are parked in the wave's LDS block ([5][64], one 8-byte read each where a pass
    // needs them): held in registers across the solves and the MFMA passes they were ten VGPRs the kernel does not have and
    // went to scratch memory with a dozen other values.
    double *park = reinterpret_cast<double *>(wl + sa.L.park);
    if (SEG) {
      double zpx = 0.0, zpy = 0.0;
      if (lp < 8 * sa.m.S) {
#pragma unroll
        for (int mm_ = 0; mm_ < 4; ++mm_) {
          const double zv = sa.Zlp[lp * 4 + mm_];
          zpx = fma(zv, sp[PR_DX + mm_], zpx); zpy = fma(zv, sp[PR_DY + mm_], zpy);
        }
      }
      const SegIn i0 = segment_inputs(lane_segment(sa.m, lp), g.K, sa.sx, pkb, 0);
      park[lp] = zpx; park[64 + lp] = zpy; park[128 + lp] = i0.x; park[192 + lp] = i0.wpx; park[256 + lp] = i0.wpy;
    }
    double mom_none[16];

    // (segment formulation) zc holds the Legendre coefficients and kbank_t the bank-max sample of the LAST trial point: the evaluation
    // of an accepted step -- the same point -- starts from them instead of projecting q through Zl again
    bool zc_trial = false;
    int kbank_t = -1;
    auto parked_inputs = [&]() -> SegIn { int l_ = lane; LAUNDER(l_); return SegIn{park[128 + l_], park[192 + l_], park[256 + l_]}; };   // chunk 0's inputs
    // cost at qi + alpha * delta (cost-only pass over the chunks)
    auto cost_at = [&](double alpha, double delta) -> double {
      if (act) qs[q_slot(lane, nq)] = qi + alpha * delta;
      wave_lds_sync();
      double ca = 0.0;
      if (SEG) {
        SEG_STAMP(7)
        segment_coefs<NQ>(nq, sa.m.S, Zl64, qs, park, zc, lane);
        SEG_STAMP(0)
        const int kbank = segment_bank_argmax(sa.m, ls, sa.sx, sa.c1, zc, load_scenp(sp));
        SegIn nin = parked_inputs();
        for (int c = 0; c < sa.m.nchunk; ++c) {
          const SegIn in = nin;
          if (c + 1 < sa.m.nchunk) nin = segment_inputs(ls, g.K, sa.sx, pkb, c + 1);
          ca += segment_phase1<false>(ls, g.K, in, sa.c1, sp, zc, cf, cfp, psi, mom_none, false, kbank, c, lane, gc);
        }
        zc_trial = true; kbank_t = kbank;
        SEG_STAMP(1)
      } else {
        const int kbank = long_bank_argmax<NQ, TL>(g, GT, pkb, qs, load_scenp(sp), lane);
        for (int k0 = 0; k0 < g.K; k0 += 64) ca += long_phase1<NQ, false, TL>(g, GT, pkb, sp, qs, us, cf, cfp, false, kbank, k0, lane, gc);
      }
      return uniform_d(ca);
    };
    // full evaluation at qi: c, gi, and (want_H) hrow = this lane's row of J^T J (+ the waypoint block) with -g as row N
    auto eval_full = [&](bool so, bool want_H) {
      if (act) qs[q_slot(lane, nq)] = qi;
      wave_lds_sync();
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      double ca = 0.0, ga = 0.0;
      if (SEG) {
        SEG_STAMP(7)
        int kbank = kbank_t;
        if (!zc_trial) {                                 // (not the point of the last trial: the first evaluation of a fit, a resumed one)
          segment_coefs<NQ>(nq, sa.m.S, Zl64, qs, park, zc, lane);
          kbank = segment_bank_argmax(sa.m, ls, sa.sx, sa.c1, zc, load_scenp(sp));
        }
        zc_trial = false;                                // (segment_gradient below uses zc as scratch)
        SEG_STAMP(0)
        const FitGeom g8{SEG_ROWS, 8, 9};                 // the chunk's operand planes: SEG_ROWS rows of eight Legendre values per derivative order
        f32x4 bs[D2D_FIT_MAX_S][1];
        double mom[16];
#pragma unroll
        for (int s = 0; s < D2D_FIT_MAX_S; ++s) bs[s][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) mom[i] = 0.0;
        SegIn nin = parked_inputs();
        for (int c = 0; c < sa.m.nchunk; ++c) {
          const SegIn in = nin;
          if (c + 1 < sa.m.nchunk) nin = segment_inputs(ls, g.K, sa.sx, pkb, c + 1);
          ca += segment_phase1<true>(ls, g.K, in, sa.c1, sp, zc, cf, cfp, psi, mom, so, kbank, c, lane, gc);
          SEG_STAMP(2)
          if (want_H) {
            // (unrolled over the segments: every offset of a pass is an immediate.  A runtime loop with one copy of the pass and
            // of the projection -- 45 instead of 270 MFMA instructions of code -- was 8 % slower.)
#pragma unroll
            for (int s = 0; s < D2D_FIT_MAX_S; ++s) {
              if (s < sa.m.S) {
                const int Ls = sa.m.l0[s + 1] - sa.m.l0[s];
                const int left = sa.m.Ks[s] - c * Ls;
                const int kn = left < Ls ? left : Ls;
                if (kn > 0) {
                  const int poff = sa.L.psi + sa.m.l0[s] * 32, coff = sa.L.cf + sa.m.l0[s] * 64, cpoff = sa.L.cfp + sa.m.l0[s] * 16;
                  if (so) jtj_mfma_so<1, 8, true>(g8, lds, woff + poff, woff + coff, woff + cpoff, lane, bs[s], nullptr, kn, true, true);
                  else jtj_mfma<1, 8, true>(g8, lds, woff + poff, nullptr, woff + coff, lane, kn, bs[s], 0, 0, true);
                }
              }
            }
          }
          wave_lds_sync();                              // every lane is done with this chunk's records
          SEG_STAMP(3)
        }
        ga = segment_gradient<NQ>(sa.m, nq, Zl64, mom, reinterpret_cast<double *>(big), zc, lane);
        SEG_STAMP(4)
        if (want_H) {
#pragma unroll
          for (int s = 0; s < D2D_FIT_MAX_S; ++s)
            if (s < sa.m.S) segment_project<NB, NQ>(nq, Zl32, s, bs[s][0], lane, acc);
        }
        SEG_STAMP(5)
      } else {
        const int kbank = long_bank_argmax<NQ, TL>(g, GT, pkb, qs, load_scenp(sp), lane);
        for (int k0 = 0; k0 < g.K; k0 += 64) {
          const int kn = g.K - k0 < 64 ? g.K - k0 : 64;
          ca += long_phase1<NQ, true, TL>(g, GT, pkb, sp, qs, us, cf, cfp, so, kbank, k0, lane, gc);
          ga += long_phase2<NQ>(g, G64g, us, k0, kn, lane);
          if (want_H) {
            const int toff = L.G32 + 4 * k0 * nq;                            // this chunk's first row of the fp32 planes (LDS copy)
            if (so) jtj_mfma_so<NB, NQ, TL32>(g, lds, toff, woff + L.cf, woff + L.cfp, lane, acc, gG32 + (size_t)k0 * nq, kn, true);
            else jtj_mfma<NB, NQ, TL32>(g, lds, toff, gG32 + (size_t)k0 * nq, woff + L.cf, lane, kn, acc, 0, 0, true);
          }
          wave_lds_sync();                              // every lane is done with this chunk's records
        }
      }
      c = uniform_d(ca); gi = ga;
      if (want_H) {
        const float ww = (float)(sp[PR_WWP] * sp[PR_WWP]);
        int lw = lane;
        LAUNDER(lw);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(ww, Wt[(t * 4 + r) * 64 + lw], acc[t][r]);
        tiles_to_image<N>(acc, big, lane);
        image_put_rhs<N>(big, lane, gi);
        wave_lds_sync();
        image_row<N>(big, lane, hrow);
        hdiag = image_diag<N>(big, lane);
        wave_lds_sync();
        nev += (so ? 3 : 2) * ((g.K + 49) / 50);      // contracted rows in units of 100 (FL_NEVAL; one unit = 200 rows)
        SEG_STAMP(6)
      } else {
        // (an evaluation ALWAYS defines the row: the old one is then dead on entry and need not survive the passes above in
        // scratch -- 48 registers the compiler otherwise spills around every evaluation)
#pragma unroll
        for (int m = 0; m < N / 2; ++m) hrow[m] = f32x2{0.f, 0.f};
        hdiag = 0.f;
      }
    };

    bool need_eval = true;
    for (;;) {
      if (need_eval) {
        eval_full(so_rows, status == D2D_ST_RUNNING);
        need_eval = false;
        if (!(fabs(c) <= 1.79e308)) status = D2D_ST_NONFINITE;
      }
      if (status != D2D_ST_RUNNING) break;
      if (local >= iter_budget || iters >= opts.max_iter) break;
      if (MODE == D2D_LM_MODE_MINPACK && phase == 0) {
        bool taken = false;
        status = mp_trial(mp, opts, c, qi, gi, hdiag, act, iters,
                          [&](double lamv, int isq_mode, double trd, float &dls, double &dxn, double &t2) -> bool {
                            float dgi_;
                            return uniform_i(damped_solve<N, true, (NQ > 0 && 2 * NQ == N)>(hrow, lamv, act, lane, big, dgi_, dls, nullptr, true, isq_mode, trd, &dxn, &t2, hdiag, true) ? 1 : 0) != 0;
                          },
                          [&](float dls) -> double { return cost_at(1.0, (double)dls); }, taken);
        ++iters; ++local;
        if (taken) {
          need_eval = true;                                // (also when converged: cost and J^T r at the final point)
          if (status == D2D_ST_RUNNING && opts.mp_finish > 0 && (mp.calm >= opts.mp_finish || (opts.mp_slow > 0 && mp.slow >= opts.mp_slow))) { phase = 1; lam = D2D_LM_LAMBDA0; nu = 2.0; so_rows = true; }
        }
        continue;
      }
      const double gmax = uniform_d(wave_max(fabs(gi)));
      if (gmax <= opts.gtol) { status = D2D_ST_CONVERGED; break; }
      float dgi, dl;
      const int ok = uniform_i(damped_solve<N, false, (NQ > 0 && 2 * NQ == N)>(hrow, lam, act, lane, big, dgi, dl, nullptr, false, 0, 0.0, nullptr, nullptr, hdiag, true) ? 1 : 0);
      const double delta = (double)dl;
      const bool so_next = MODE == D2D_LM_MODE_MINPACK ? true : (opts.so_lambda > 0.0 && lam <= opts.so_lambda);
      const double pred = uniform_d(wave_sum(delta * (lam * (double)dgi * delta - gi)));
      const double dmax = uniform_d(wave_max(fabs(delta))), qmax = uniform_d(wave_max(fabs(qi)));
      double ct = 0.0, pred_s = pred, alpha = 1.0, bt_a = 0.0, bt_b = 0.0;
      bool fin = false, accept = false;
      if (ok) {
        for (int att = 0; att < 3; ++att) {
          const double ca = cost_at(alpha, delta);
          if (att == 0) {
            ct = ca;
            fin = (fabs(ct) <= 1.79e308) && (pred > 0.0);
            if (fin && (c - ct) / pred > 0.0) { accept = true; break; }
            if (!fin) break;
            bt_a = uniform_d(-2.0 * wave_sum(gi * delta)); bt_b = bt_a - pred;
            alpha = bt_first_alpha(bt_a, c, ct);
          } else {
            if ((fabs(ca) <= 1.79e308) && ca < c) { accept = true; ct = ca; pred_s = bt_a * alpha - bt_b * alpha * alpha; break; }
            alpha = fmax(D2D_LM_BT_SHRINK * alpha, D2D_LM_BT_FLOOR);
          }
        }
      }
      const StepOutcome so = lm_update(ok != 0, fin, accept, accept ? alpha : 1.0, c, ct, pred, pred_s, dmax, qmax, lam, nu, opts);
      ++iters; ++local;
      lam = uniform_d(so.lam); nu = uniform_d(so.nu); status = so.status;
      if (so.accept) {
        qi += alpha * delta;
        need_eval = true;                                // (also when converged: cost and J^T r at the final point)
        so_rows = so_next;
      }
    }
    if (status == D2D_ST_RUNNING && iters >= opts.max_iter) status = D2D_ST_MAXITER;
    const double gmax = uniform_d(wave_max(fabs(gi)));
    {
      int lane_io = lane;
      LAUNDER(lane_io);
      if (act) { q_io[(size_t)b * n + lane_io] = qi; g_io[(size_t)b * n + lane_io] = gi; }
    }
    if (lane == 0) {
      cost_io[b] = c;
      lm[LM_STRIDE * b + 0] = lam; lm[LM_STRIDE * b + 1] = nu; lm[LM_STRIDE * b + 2] = gmax; lm[LM_STRIDE * b + 3] = so_rows ? 1.0 : 0.0;
      if (MODE == D2D_LM_MODE_MINPACK) {
        lm[LM_STRIDE * b + 4] = mp.par; lm[LM_STRIDE * b + 5] = mp.delta;
        lm[LM_STRIDE * b + 6] = (double)(phase | (mp.first << 1) | ((mp.calm & 0x3fff) << 2) | (mp.slow << 16)); lm[LM_STRIDE * b + 7] += (double)mp.nfac;
      }
      flags[4 * b + FL_STATUS] = status; flags[4 * b + FL_ITERS] = iters; flags[4 * b + FL_NEED] = 1;
      flags[4 * b + FL_NEVAL] += nev;
    }
  }
  if (st_on) {
    SEG_STAMP(7)
    if (lane == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(&sa.stamps[i], st_acc[i]);
  }
#undef SEG_STAMP
  if (queue != nullptr && lane == 0) {
    if (atomicAdd(queue + 1, 1) == stride - 1) { queue[0] = 0; queue[1] = 0; }
  }
}

// ------------------------------------------------------------------------------------
// cost, J^T r, J^T J at given points for horizons the LDS image of fit_eval_kernel cannot hold (the public d2d_fit_eval beyond
// ~229 nodes): one evaluation in the segment formulation (fit_seg.h), Gauss-Newton rows; H in fit_eval_kernel's tile layout,
// without the waypoint block (untile_kernel adds it).
template <int NB, int NQ>
__global__ void __launch_bounds__(64 * FIT_LM_WPB_MAX)
fit_eval_seg_kernel(int B, FitGeom g, SegArgs sa, const double *__restrict__ pk, const double *__restrict__ prep,
                    const double *__restrict__ q_in, double *__restrict__ cost_out, double *__restrict__ g_out, float *__restrict__ H_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NT = NB * (NB + 1) / 2;
  stage(lds + sa.L.Zl64, sa.Zl64, 8 * sa.m.S * (g.nq + 1) * 8);
  stage(lds + sa.L.Zl32, sa.Zl32, 8 * sa.m.S * g.nq * 4);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int woff = sa.L.wave0 + wave * sa.L.wave_stride;
  unsigned char *wl = lds + woff;
  double *qs = reinterpret_cast<double *>(wl + sa.L.qs), *sp = reinterpret_cast<double *>(wl + sa.L.sp), *zc = reinterpret_cast<double *>(wl + sa.L.zc);
  f32x4 *cf = reinterpret_cast<f32x4 *>(wl + sa.L.cf);
  float2 *cfp = reinterpret_cast<float2 *>(wl + sa.L.cfp);
  float *psi = reinterpret_cast<float *>(wl + sa.L.psi);
  double *red = reinterpret_cast<double *>(wl + sa.L.big);
  const double *Zl64 = reinterpret_cast<const double *>(lds + sa.L.Zl64);
  const float *Zl32 = reinterpret_cast<const float *>(lds + sa.L.Zl32);
  const LaneSeg ls = lane_segment(sa.m, lane);
  const int nq = NQ ? NQ : g.nq, n = 2 * nq;
  const GroupCtx gc{nullptr, 1, 0, 0, 0};
  for (int b = blockIdx.x * (blockDim.x >> 6) + wave; b < B; b += gridDim.x * (blockDim.x >> 6)) {
    const double *pkb = pk + (size_t)b * FIT_PK * g.K;
    if (lane < n) qs[q_slot(lane, nq)] = q_in[(size_t)b * n + lane];
    for (int i = lane; i < FIT_PREP_STRIDE; i += 64) sp[i] = prep[(size_t)b * FIT_PREP_STRIDE + i];
    wave_lds_sync();
    double zpx = 0.0, zpy = 0.0;
    if (lane < 8 * sa.m.S) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double zv = sa.Zlp[lane * 4 + m];
        zpx = fma(zv, sp[PR_DX + m], zpx); zpy = fma(zv, sp[PR_DY + m], zpy);
      }
    }
    double *park = reinterpret_cast<double *>(wl + sa.L.park);
    park[lane] = zpx; park[64 + lane] = zpy;
    segment_coefs<NQ>(nq, sa.m.S, Zl64, qs, park, zc, lane);
    const int kbank = segment_bank_argmax(sa.m, ls, sa.sx, sa.c1, zc, load_scenp(sp));
    const FitGeom g8{SEG_ROWS, 8, 9};
    f32x4 bs[D2D_FIT_MAX_S][1];
    double mom[16], ca = 0.0;
#pragma unroll
    for (int s = 0; s < D2D_FIT_MAX_S; ++s) bs[s][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) mom[i] = 0.0;
    for (int c = 0; c < sa.m.nchunk; ++c) {
      const SegIn in = segment_inputs(ls, g.K, sa.sx, pkb, c);
      ca += segment_phase1<true>(ls, g.K, in, sa.c1, sp, zc, cf, cfp, psi, mom, false, kbank, c, lane, gc);
      if (H_out) {
#pragma unroll
        for (int s = 0; s < D2D_FIT_MAX_S; ++s) {
          if (s < sa.m.S) {
            const int Ls = sa.m.l0[s + 1] - sa.m.l0[s];
            const int left = sa.m.Ks[s] - c * Ls;
            const int kn = left < Ls ? left : Ls;
            if (kn > 0) jtj_mfma<1, 8, true>(g8, lds, woff + sa.L.psi + sa.m.l0[s] * 32, nullptr, woff + sa.L.cf + sa.m.l0[s] * 64, lane, kn, bs[s], 0, 0, true);
          }
        }
      }
      wave_lds_sync();
    }
    const double gl = segment_gradient<NQ>(sa.m, nq, Zl64, mom, red, zc, lane);
    if (lane < n && g_out) g_out[(size_t)b * n + lane] = gl;
    if (lane == 0 && cost_out) cost_out[b] = uniform_d(ca);
    if (H_out) {
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < D2D_FIT_MAX_S; ++s)
        if (s < sa.m.S) segment_project<NB, NQ>(nq, Zl32, s, bs[s][0], lane, acc);
      float *Hb = H_out + (size_t)b * NT * 256;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Hb[(t * 4 + r) * 64 + lane] = acc[t][r];
    }
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
fit_state_init_kernel(int B, int off, int stride, double *__restrict__ lm, int32_t *__restrict__ flags,
                      int32_t *__restrict__ queue) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && queue) { queue[0] = 0; queue[1] = 0; queue[2] = 0; queue[3] = 0; queue[4] = 0; queue[5] = 0; }     // work queue + ring counters of the persistent LM kernel (fit_lm_kernel); [5] = bounded waits that gave up, over the whole solve
  if (i >= B) return;
  const int b = off + i * stride;
  lm[LM_STRIDE * b + 0] = D2D_LM_LAMBDA0; lm[LM_STRIDE * b + 1] = 2.0; lm[LM_STRIDE * b + 2] = 0.0; lm[LM_STRIDE * b + 3] = 0.0;
  // MINPACK mode: par = 0, radius not yet set (factor * ||x|| on the first visit), phase 0 with the `first` flag, no factorisations
  lm[LM_STRIDE * b + 4] = 0.0; lm[LM_STRIDE * b + 5] = -1.0; lm[LM_STRIDE * b + 6] = 2.0; lm[LM_STRIDE * b + 7] = 0.0;
  flags[4 * b + FL_STATUS] = D2D_ST_RUNNING; flags[4 * b + FL_ITERS] = 0; flags[4 * b + FL_NEED] = 1;
  if (stride == 1) flags[4 * b + FL_NEVAL] = 0;       // group sweeps keep counting across visits
}


// Hand-out order of the persistent LM kernel: order[0..B) = trajectory indices sorted by the iteration counts of a previous
// solve, longest first (counting sort in one workgroup; ties in index order within a 64-wide stripe, otherwise as the
// atomics fall: the fits are independent, the order only schedules them).
#define ORDER_BINS 2048
__global__ void __launch_bounds__(1024)
fit_order_kernel(int B, const int32_t *__restrict__ iters, int32_t *__restrict__ order) {
  __shared__ int hist[ORDER_BINS];
  for (int i = threadIdx.x; i < ORDER_BINS; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int k = iters[b];
    k = k < 0 ? 0 : (k >= ORDER_BINS ? ORDER_BINS - 1 : k);
    atomicAdd(&hist[ORDER_BINS - 1 - k], 1);          // bin 0 = the longest fits
  }
  __syncthreads();
  {
    // exclusive prefix sum of the bins: thread t owns bins 2t, 2t + 1; Hillis-Steele scan of the 1024 pair sums in the LDS (ten
    // steps; the serial loop of one thread this replaces was 20 us of a 2.4 ms solve since the predicted hand-out sorts EVERY solve)
    __shared__ int part[2][1024];
    const int t = threadIdx.x;
    const int h0 = hist[2 * t], h1 = hist[2 * t + 1];
    int cur = 0;
    part[0][t] = h0 + h1;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int v = part[cur][t] + (t >= d ? part[cur][t - d] : 0);
      part[cur ^ 1][t] = v;
      cur ^= 1;
      __syncthreads();
    }
    const int excl = part[cur][t] - (h0 + h1);
    hist[2 * t] = excl; hist[2 * t + 1] = excl + h0;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int k = iters[b];
    k = k < 0 ? 0 : (k >= ORDER_BINS ? ORDER_BINS - 1 : k);
    order[atomicAdd(&hist[ORDER_BINS - 1 - k], 1)] = b;
  }
}

// Predicted hand-out (include/d2d.h d2d_fit_plan_set_handout_prior): key[b] = the expected trial count of fit b from its scenario row
// alone -- how far the end headings are from the legs of the 'tri' dog-leg the fit starts from, and the chord length -- quantised
// for fit_order_kernel (1/8 trial per bin, offset 40: prior entries are a few tens at most).  One thread per fit.
__global__ void __launch_bounds__(256)
fit_handout_key_kernel(int B, double duration, const double *__restrict__ scen, const float *__restrict__ tab, int32_t *__restrict__ key) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double *r = scen + (size_t)b * D2D_SCEN_STRIDE;
  const double dx = r[D2D_SC_X1] - r[D2D_SC_X0], dy = r[D2D_SC_Y1] - r[D2D_SC_Y0];
  const double d = sqrt(dx * dx + dy * dy), D = r[D2D_SC_VREF] * duration;
  const double h = 0.5 * sqrt(fmax(D * D - d * d, 0.0));
  const double gl = r[D2D_SC_GOLEFT] > 0.0 ? 1.0 : (r[D2D_SC_GOLEFT] < 0.0 ? -1.0 : 0.0);
  const double beta = atan2(dy, dx), a = atan2(gl * h, 0.5 * d);
  const double TWO_PI = 6.283185307179586, PI = 3.141592653589793;
  auto wrap = [&](double v) { v = fmod(v + PI, TWO_PI); if (v < 0.0) v += TWO_PI; return v; };      // [0, 2 pi): angle + pi
  auto abin = [&](double v) { int i = (int)(wrap(v) * (D2D_HANDOUT_NB / TWO_PI)); return i < 0 ? 0 : (i >= D2D_HANDOUT_NB ? D2D_HANDOUT_NB - 1 : i); };
  const int b0 = abin(r[D2D_SC_PSI0] - (beta + a)), b1 = abin(r[D2D_SC_PSI1] - (beta - a));
  const double x = D > 0.0 ? d / D : 0.0;
  int bx = (int)floor((x - D2D_HANDOUT_X_LO) * (D2D_HANDOUT_ND / (D2D_HANDOUT_X_HI - D2D_HANDOUT_X_LO)));
  bx = bx < 0 ? 0 : (bx >= D2D_HANDOUT_ND ? D2D_HANDOUT_ND - 1 : bx);
  float k = tab[b0 * D2D_HANDOUT_ND + bx] + tab[(D2D_HANDOUT_NB + b1) * D2D_HANDOUT_ND + bx];
  if (!(k == k)) k = 0.f;                                  // (a NaN end pose: the fit reports D2D_ST_NONFINITE itself)
  int ki = (int)(8.f * (k + 40.f));
  key[b] = ki < 0 ? 0 : (ki >= ORDER_BINS ? ORDER_BINS - 1 : ki);
}

// sampled x, y of every trajectory: pos [B][2][K]
__global__ void __launch_bounds__(256)
fit_pos_kernel(int B, FitGeom g, const double *__restrict__ G64, const double *__restrict__ Gp64,
               const double *__restrict__ prep, const double *__restrict__ q, double *__restrict__ pos) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * g.K) return;
  const int b = i / g.K, k = i - (long)b * g.K;
  const double *pr = prep + (size_t)b * FIT_PREP_STRIDE;
  const double *qa = q + (size_t)b * 2 * g.nq;
  const double *gp = Gp64 + (size_t)k * 4, *g0 = G64 + (size_t)k * g.gstr;
  double x = 0.0, y = 0.0;
  for (int c = 0; c < 4; ++c) { x = fma(gp[c], pr[PR_DX + c], x); y = fma(gp[c], pr[PR_DY + c], y); }
  for (int j = 0; j < g.nq; ++j) { x = fma(g0[j], qa[j], x); y = fma(g0[j], qa[g.nq + j], y); }
  pos[((size_t)b * 2) * g.K + k] = x;
  pos[((size_t)b * 2 + 1) * g.K + k] = y;
}

// moved[0] = max over trajectories of max|q - q_prev| / (1 + max|q_prev|)   (bit pattern of a non-negative double)
__global__ void __launch_bounds__(256)
fit_moved_kernel(int B, int n, const double *__restrict__ q, const double *__restrict__ qp, double *__restrict__ moved) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  double m = 0.0;
  if (b < B) {
    double d = 0.0, a = 0.0;
    for (int j = 0; j < n; ++j) { d = fmax(d, fabs(q[(size_t)b * n + j] - qp[(size_t)b * n + j])); a = fmax(a, fabs(qp[(size_t)b * n + j])); }
    m = d / (1.0 + a);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long *>(moved), (unsigned long long)__double_as_longlong(m));
}

// counts trajectories still running -> counter[0]
__global__ void __launch_bounds__(256)
fit_count_kernel(int B, const int32_t *__restrict__ flags, int32_t *__restrict__ counter) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int run = (b < B && flags[4 * b + FL_STATUS] == D2D_ST_RUNNING) ? 1 : 0;
  const unsigned long long m = __ballot(run);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (int)__popcll(m));
}

// counts trajectories still running with fewer than `cap` iterations of their own (time-sliced hand-out: left in the ring) -> counter[0]
__global__ void __launch_bounds__(256)
fit_count_below_kernel(int B, const int32_t *__restrict__ flags, int cap, int32_t *__restrict__ counter) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int run = (b < B && flags[4 * b + FL_STATUS] == D2D_ST_RUNNING && flags[4 * b + FL_ITERS] < cap) ? 1 : 0;
  const unsigned long long m = __ballot(run);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (int)__popcll(m));
}

// stats[0] += cost, stats[1] = max gmax, stats[2] += not converged, stats[3] += evaluations
__global__ void __launch_bounds__(256)
fit_stats_kernel(int B, int n, const double *__restrict__ cost, const double *__restrict__ g,
                 const int32_t *__restrict__ flags, double *__restrict__ stats) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  double c = 0.0, gm = 0.0, nr = 0.0, ne = 0.0;
  if (b < B) {
    c = cost[b];
    for (int j = 0; j < n; ++j) gm = fmax(gm, fabs(g[(size_t)b * n + j]));
    const int st = flags[4 * b + FL_STATUS];
    nr = (st == D2D_ST_CONVERGED || st == D2D_ST_STALLED) ? 0.0 : 1.0;
    ne = 0.5 * flags[4 * b + FL_NEVAL];            // evaluations in units of a Gauss-Newton one (200 rows; second-order: 1.5)
  }
  c = wave_sum(c); gm = wave_max(gm); nr = wave_sum(nr); ne = wave_sum(ne);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&stats[0], c);
    atomicAdd(&stats[2], nr);
    atomicAdd(&stats[3], ne);
    // max of non-negative doubles == max of their bit patterns
    atomicMax(reinterpret_cast<unsigned long long *>(&stats[1]), (unsigned long long)__double_as_longlong(gm));
  }
}

__global__ void __launch_bounds__(256)
fit_export_kernel(int B, const int32_t *__restrict__ flags, int32_t *__restrict__ iters,
                  int32_t *__restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (iters) iters[b] = flags[4 * b + FL_ITERS];
  if (status) status[b] = flags[4 * b + FL_STATUS];
}

// q0 = Pinit (wp - Gp0 d): one wavefront per trajectory, lane = unknown
__global__ void __launch_bounds__(256)
fit_init_kernel(int B, int K, int nq, double duration, const double *__restrict__ Gp,
                const double *__restrict__ Pinit, const double *__restrict__ scen,
                double *__restrict__ q) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const Scen s = load_scen(scen + (size_t)b * D2D_SCEN_STRIDE, duration);
  const int n = 2 * nq;
  for (int l = lane; l < n; l += 64) {
    const int ax = l >= nq, j = l - ax * nq;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      double wx, wy;
      waypoint_at(s, K, k, wx, wy);
      const double *gp = Gp + (size_t)k * 4;
      double base = 0.0;
      for (int c = 0; c < 4; ++c) base += gp[c] * (ax ? s.dy[c] : s.dx[c]);
      acc += Pinit[(size_t)j * K + k] * ((ax ? wy : wx) - base);
    }
    q[(size_t)b * n + l] = acc;
  }
}

// q0 = Pinit (xy - Gp0 d) for caller-supplied node positions xy [B][2][K]
__global__ void __launch_bounds__(256)
fit_project_kernel(int B, int K, int nq, double duration, const double *__restrict__ Gp,
                   const double *__restrict__ Pinit, const double *__restrict__ scen,
                   const double *__restrict__ xy, double *__restrict__ q) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const Scen s = load_scen(scen + (size_t)b * D2D_SCEN_STRIDE, duration);
  const int n = 2 * nq;
  for (int l = lane; l < n; l += 64) {
    const int ax = l >= nq, j = l - ax * nq;
    const double *src = xy + ((size_t)b * 2 + ax) * K;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      const double *gp = Gp + (size_t)k * 4;
      double base = 0.0;
      for (int c = 0; c < 4; ++c) base += gp[c] * (ax ? s.dy[c] : s.dx[c]);
      acc += Pinit[(size_t)j * K + k] * (src[k] - base);
    }
    q[(size_t)b * n + l] = acc;
  }
}

// z[b][axis][row] = Zp[row] . d_axis + Z[row] . q_axis
__global__ void __launch_bounds__(256)
fit_coeffs_kernel(int B, int S, int nq, double duration, const double *__restrict__ Z,
                  const double *__restrict__ Zp, const double *__restrict__ scen,
                  const double *__restrict__ q, double *__restrict__ z) {
  const int nz = 8 * S;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * 2 * nz) return;
  const int b = i / (2 * nz), r = i - (long)b * 2 * nz, ax = r / nz, row = r - ax * nz;
  const Scen s = load_scen(scen + (size_t)b * D2D_SCEN_STRIDE, duration);
  double acc = 0.0;
  for (int c = 0; c < 4; ++c) acc += Zp[(size_t)row * 4 + c] * (ax ? s.dy[c] : s.dx[c]);
  const double *qa = q + (size_t)b * 2 * nq + ax * nq;
  for (int j = 0; j < nq; ++j) acc += Z[(size_t)row * nq + j] * qa[j];
  z[i] = acc;
}

// flat outputs and flatness states at the K nodes
__global__ void __launch_bounds__(256)
fit_sample_kernel(int B, FitGeom g, double duration, const double *__restrict__ G64,
                  const double *__restrict__ Gp64, const double *__restrict__ scen,
                  const double *__restrict__ q, double *__restrict__ Yo, double *__restrict__ Xo) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * g.K) return;
  const int b = i / g.K, k = i - (long)b * g.K;
  const Scen s = load_scen(scen + (size_t)b * D2D_SCEN_STRIDE, duration);
  double Y[6];
  flat_outputs(g, G64, Gp64, q + (size_t)b * 2 * g.nq, s, k, Y);
  if (Yo)
    for (int c = 0; c < 6; ++c) Yo[((size_t)b * 6 + c) * g.K + k] = Y[c];
  if (Xo) {
    // DiffFlatness.state_and_input_from_output, src/d2d/guidance.py:22-47
    const double a = Y[2] - s.wx, bb = Y[3] - s.wy;
    const double va = sqrt(a * a + bb * bb);
    double *o = Xo + (size_t)b * 5 * g.K + k;
    o[0] = Y[0]; o[g.K] = Y[1]; o[2 * g.K] = atan2(bb, a);
    o[3 * g.K] = atan((Y[5] * a - Y[4] * bb) / va / FIT_G); o[4 * g.K] = va;
  }
}

// ------------------------------------------------------------------------------------
template <typename T>
static int upload(T **dst, const std::vector<T> &src) {
  D2D_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(dst), src.size() * sizeof(T)));
  D2D_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return D2D_OK;
}

static FitGeom geom_of(const d2d_fit_plan *pl) { return FitGeom{pl->K, pl->nq, pl->nq + 1}; }

static void free_scratch(d2d_fit_plan *pl) {
  void **ptrs[] = {(void **)&pl->d_g, (void **)&pl->d_H, (void **)&pl->d_cost, (void **)&pl->d_lm, (void **)&pl->d_flags,
                   (void **)&pl->d_prep, (void **)&pl->d_pos, (void **)&pl->d_qprev, (void **)&pl->d_pk, (void **)&pl->d_rows,
                   (void **)&pl->d_order, (void **)&pl->d_ring, (void **)&pl->d_hkey, (void **)&pl->d_order_pred};
  for (void **p : ptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  pl->cap_B = 0; pl->ring_cap = 0;
  pl->order_B = 0; pl->gorder_R = 0; pl->gsweeps_R = 0;
  pl->last_order = nullptr; pl->last_order_B = 0;
  pl->rows_B = 0;            // (the row records of d2d_fit_rows lived in the freed scratch)
}

// Scratch of a plan grows on demand.  A failed regrow leaves the plan with NO scratch (every pointer null, cap_B = 0):
// the next call allocates afresh instead of running on freed memory.
static int ensure_scratch(d2d_fit_plan *pl, int B) {
  if (B <= pl->cap_B) return D2D_OK;
  const size_t n = 2 * pl->nq;
  free_scratch(pl);
#define SCRATCH_ALLOC(field, bytes)                                                        \
  if (hipMalloc(reinterpret_cast<void **>(&pl->field), (bytes)) != hipSuccess) {           \
    pl->field = nullptr;                                                                   \
    free_scratch(pl);                                                                      \
    d2d_set_error("fit scratch: hipMalloc of %zu bytes for B=%d failed", (size_t)(bytes), B); \
    return D2D_ENOMEM;                                                                     \
  }
  SCRATCH_ALLOC(d_prep, (size_t)B * FIT_PREP_STRIDE * sizeof(double))
  SCRATCH_ALLOC(d_pk, (size_t)B * FIT_PK * pl->K * sizeof(double))
  SCRATCH_ALLOC(d_pos, (size_t)B * 2 * pl->K * sizeof(double))
  SCRATCH_ALLOC(d_qprev, (size_t)B * n * sizeof(double))
  SCRATCH_ALLOC(d_g, (size_t)B * n * sizeof(double))
  {
    const size_t nb = (n + 15) / 16, tiles = nb * (nb + 1) / 2;
    SCRATCH_ALLOC(d_H, (size_t)B * tiles * 256 * sizeof(float))
  }
  SCRATCH_ALLOC(d_rows, (size_t)B * 4 * (pl->K + 1) * 4 * sizeof(float))
  SCRATCH_ALLOC(d_order, (size_t)B * sizeof(int32_t))
  SCRATCH_ALLOC(d_hkey, (size_t)B * sizeof(int32_t))
  SCRATCH_ALLOC(d_order_pred, (size_t)B * sizeof(int32_t))
  {
    int cap = 64;
    while (cap < B) cap <<= 1;
    SCRATCH_ALLOC(d_ring, (size_t)cap * sizeof(int32_t))      // ring of yielded fits (fit_lm_kernel), -1 = empty slot
    pl->ring_cap = cap;
  }
  SCRATCH_ALLOC(d_cost, (size_t)B * sizeof(double))
  SCRATCH_ALLOC(d_lm, (size_t)B * LM_STRIDE * sizeof(double))
  SCRATCH_ALLOC(d_flags, (size_t)B * 4 * sizeof(int32_t))
#undef SCRATCH_ALLOC
  if (int rc = fit_knot_ensure(pl, B)) { free_scratch(pl); return rc; }
  pl->cap_B = B;
  return D2D_OK;
}

static int launch_prep(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *scen) {
  // (d_prep is about to describe other scenarios: the row records of an earlier d2d_fit_rows no longer pair with it)
  const_cast<d2d_fit_plan *>(pl)->rows_B = 0;
  // (one thread per trajectory writes its 112-double row: 64-thread blocks spread a 4096-fit batch over 64 CUs instead of 16 -- the
  // kernel is a chain of strided row accesses per thread, 18 us of every 2.4 ms solve with 256-thread blocks)
  hipLaunchKernelGGL(fit_prep_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, B, pl->K, pl->duration, scen, pl->d_prep);
  hipLaunchKernelGGL(fit_prepk_kernel, dim3(((long)B * pl->K + 255) / 256), dim3(256), 0, ctx->stream, B, pl->K, pl->d_prep, pl->d_Gp, pl->d_pk);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

// d_prep must hold the rows of `scen` (launch_prep) before either kernel runs
static GroupArgs no_groups() { return GroupArgs{nullptr, 0, 0, 1, 0}; }

static int launch_eval(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *q,
                       int32_t *flags, double *cost, double *g, float *H, GroupArgs ga = no_groups(),
                       f32x4 *rows = nullptr) {
  const FitGeom gm = geom_of(pl);
  const FitLds L = eval_lds_layout(pl->K, pl->nq, pl->g32_lds, pl->wpb_eval, pl->nds);
  const int NB = (2 * pl->nq + 15) / 16;
  int nblk = (B + pl->wpb_eval - 1) / pl->wpb_eval;
  if (nblk > pl->n_cu) nblk = pl->n_cu;        // persistent beyond one workgroup per CU (LDS-limited residency)
  const dim3 grid(nblk), block(64 * pl->wpb_eval);
  static const int dbg = getenv("D2D_FIT_ABLATE") ? atoi(getenv("D2D_FIT_ABLATE")) : 0;   // timing experiments only
#define LAUNCH_EVAL(NBV, INLDS)                                                                    \
  if (pl->nq == 24 && NBV == 3)                                                                    \
    hipLaunchKernelGGL((fit_eval_kernel<3, 24, INLDS>), grid, block, L.total, ctx->stream, B, gm, L, dbg, ga, \
                       pl->d_G, pl->d_pk, pl->d_G32, pl->d_W32, pl->d_prep, q, flags, cost, g, H, rows);  \
  else                                                                                             \
  hipLaunchKernelGGL((fit_eval_kernel<NBV, 0, INLDS>), grid, block, L.total, ctx->stream, B, gm, L, dbg, ga, \
                     pl->d_G, pl->d_pk, pl->d_G32, pl->d_W32, pl->d_prep, q, flags, cost, g, H, rows)
  if (pl->g32_lds) {
    if (NB == 1) LAUNCH_EVAL(1, true);
    else if (NB == 2) LAUNCH_EVAL(2, true);
    else LAUNCH_EVAL(3, true);
  } else {
    if (NB == 1) LAUNCH_EVAL(1, false);
    else if (NB == 2) LAUNCH_EVAL(2, false);
    else LAUNCH_EVAL(3, false);
  }
#undef LAUNCH_EVAL
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

static int launch_step(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, double *q, const d2d_fit_opts &o,
                       GroupArgs ga = no_groups()) {
  const FitGeom gm = geom_of(pl);
  const int NB = (2 * pl->nq + 15) / 16;
  const StepLds L = step_lds_layout(pl->K, pl->nq, 16 * NB, pl->wpb_step);
  const dim3 grid((B + pl->wpb_step - 1) / pl->wpb_step), block(64 * pl->wpb_step);
#define LAUNCH_STEP(NV)                                                                            \
  hipLaunchKernelGGL(fit_step_kernel<NV>, grid, block, L.total, ctx->stream, B, gm, L, o, ga, \
                     pl->d_G, pl->d_pk, pl->d_W32, pl->d_prep, q, pl->d_g, pl->d_H, pl->d_cost, pl->d_lm, pl->d_flags)
  if (NB == 1) LAUNCH_STEP(16);
  else if (NB == 2) LAUNCH_STEP(32);
  else LAUNCH_STEP(48);
#undef LAUNCH_STEP
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

// the time slice of the fused kernel's hand-out
static int lm_slice(const d2d_fit_opts &o) { return o.slice > 0 ? o.slice : 0; }

// which persistent kernel a launch of this plan with these options runs: 0 fit_lm_kernel, 1 fit_lm_knot_kernel, 2 fit_lm_long_kernel
static int solve_kernel_of(const d2d_fit_plan *pl, const d2d_fit_opts &o) {
  if (!pl->use_lm) return 2;
  return (pl->kn.wpb > 0 && o.mode == D2D_LM_MODE_MINPACK && lm_slice(o) <= 0) ? 1 : 0;
}

// the hand-out order of a launch over B trajectories: the caller's explicit hint, else the predicted order of this solve, else index order
static const int32_t *handout_order(const d2d_fit_plan *pl, int B) {
  if (pl->order_B == B) return pl->d_order;
  if (pl->last_order == pl->d_order_pred && pl->last_order_B == B) return pl->d_order_pred;
  return nullptr;
}

// D2D_HANDOUT_PREDICTED: keys from the scenario rows + counting sort -> d_order_pred (two small launches at the start of a solve)
static int launch_handout(d2d_ctx *ctx, d2d_fit_plan *pl, int B, const double *scen) {
  hipLaunchKernelGGL(fit_handout_key_kernel, dim3((B + 255) / 256), dim3(256), 0, ctx->stream, B, pl->duration, scen, pl->d_hprior, pl->d_hkey);
  hipLaunchKernelGGL(fit_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, B, pl->d_hkey, pl->d_order_pred);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

static int launch_lm(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, double *q, const d2d_fit_opts &o, int iter_cap) {
  const FitGeom gm = geom_of(pl);
  const FusedLds L = fused_lds_layout(pl->K, pl->nq, 48, pl->wpb_lm);
  static const bool want_stamps = getenv("D2D_LM_STAMPS") != nullptr;
  static const bool want_times = getenv("D2D_LM_TIMES") != nullptr;      // diagnostics: wall time of every launch
  std::chrono::steady_clock::time_point t0;
  if (want_times) { D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream)); t0 = std::chrono::steady_clock::now(); }
  unsigned long long *stamps = want_stamps ? reinterpret_cast<unsigned long long *>(ctx->stats_dev + 8) : nullptr;
  if (want_stamps) D2D_CHECK_HIP(hipMemsetAsync(stamps, 0, 13 * sizeof(unsigned long long), ctx->stream));
  const int blocks = B < pl->n_cu ? B : pl->n_cu;     // persistent: one workgroup per CU
  int32_t *queue = ctx->counter_dev + 8;
  const int prio_only = o.prio_at;
  d2d_fit_opts oo = o;
  oo.slice = lm_slice(o);
  if (oo.slice > 0) {                    // time-sliced hand-out: every launch starts from an empty ring and zeroed tickets
    D2D_CHECK_HIP(hipMemsetAsync(pl->d_ring, 0xff, (size_t)pl->ring_cap * sizeof(int32_t), ctx->stream));
    D2D_CHECK_HIP(hipMemsetAsync(queue + 2, 0, 3 * sizeof(int32_t), ctx->stream));
  }
  const int32_t *order = handout_order(pl, B);
  // the default solver of the headline shape runs in knot coordinates (fit_knot.hip); the time-sliced hand-out and the FAST loop
  // stay on fit_lm_kernel
  if (solve_kernel_of(pl, oo) == 1) {
    if (int rc = fit_knot_launch(ctx, const_cast<d2d_fit_plan *>(pl), B, q, oo, iter_cap, order, prio_only)) return rc;
    if (want_times) {
      D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      fprintf(stderr, "[fit_lm_knot launch] B=%d iter_cap=%d: %.1f us\n", B, iter_cap, us);
    }
    return D2D_OK;
  }
#define LAUNCH_LM(STAMPSV, MODEV)                                                                                                  \
  hipLaunchKernelGGL((fit_lm_kernel<3, 24, STAMPSV, MODEV>), dim3(blocks), dim3(64 * pl->wpb_lm), L.total, ctx->stream, B, gm, L, oo, iter_cap, \
                     pl->d_G, pl->d_pk, pl->d_G32, pl->d_W32, pl->d_prep, q, pl->d_cost, pl->d_g, pl->d_lm, pl->d_flags, queue,   \
                     pl->d_ring, pl->ring_cap - 1, stamps, order, prio_only)
  if (oo.mode == D2D_LM_MODE_FAST) { if (want_stamps) LAUNCH_LM(true, D2D_LM_MODE_FAST); else LAUNCH_LM(false, D2D_LM_MODE_FAST); }
  else { if (want_stamps) LAUNCH_LM(true, D2D_LM_MODE_MINPACK); else LAUNCH_LM(false, D2D_LM_MODE_MINPACK); }
#undef LAUNCH_LM
  D2D_LAUNCH_CHECK();
  if (want_times) {
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[fit_lm launch] B=%d iter_cap=%d blocks=%d wpb=%d: %.1f us\n", B, iter_cap, blocks, pl->wpb_lm, us);
  }
  if (want_stamps) {
    unsigned long long h[13];
    D2D_CHECK_HIP(hipMemcpyAsync(h, stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    static const char *nm[8] = {"io/loop", "phase1", "phase2", "mfma", "image", "solve", "reduce+judge", "-"};
    double tot = 0;
    for (int i = 0; i < 7; ++i) tot += (double)h[i];
    fprintf(stderr, "[fit_lm stamps] wave-cycles (s_memtime ticks), B=%d iter_cap=%d:", B, iter_cap);
    for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)h[i] / tot);
    fprintf(stderr, " total=%.3e | solve split: setup=%.1f%% steps0-15=%.1f%% 16-31=%.1f%% 32-47=%.1f%% subst=%.1f%%\n", tot,
            100.0 * h[8] / tot, 100.0 * h[9] / tot, 100.0 * h[10] / tot, 100.0 * h[11] / tot, 100.0 * h[12] / tot);
  }
  return D2D_OK;
}

static SegArgs seg_args_of(const d2d_fit_plan *pl, int NB) {
  SegArgs sa{};
  sa.m.S = pl->seg_S; sa.m.nchunk = pl->seg_nchunk;
  for (int i = 0; i <= D2D_FIT_MAX_S; ++i) sa.m.l0[i] = pl->seg_l0[i];
  for (int i = 0; i < D2D_FIT_MAX_S; ++i) { sa.m.k0[i] = pl->seg_k0[i]; sa.m.Ks[i] = pl->seg_Ks[i]; }
  sa.L = seg_lds_layout(16 * NB, pl->nq, pl->S, FIT_LM_WPB_MAX);
  sa.Zl64 = pl->d_Zl64; sa.Zl32 = pl->d_Zl32; sa.Zlp = pl->d_Zlp; sa.sx = pl->d_sx;
  sa.c1 = 2.0 / pl->T;
  return sa;
}

static int launch_lm_long(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, double *q, const d2d_fit_opts &o, int budget, GroupArgs ga = no_groups()) {
  const FitGeom gm = geom_of(pl);
  const int NB = (2 * pl->nq + 15) / 16;
  // d2d_fit_plan_opts.long_tables (A/B switch): 0 = both tables from global memory, 1 = fp64 block in the LDS and the fp32 planes from global
  // memory, 2 = both in the LDS.  Default: both if they fit beside >= 3 per-wave blocks (K <= 121 at nq = 24; measured at K = 121,
  // 4096 fits: 251 k fits/s from global memory with 8 waves per CU, 416 k with the fp64 block in the LDS and 6 waves, 542 k with
  // both tables and 3 waves; converting the MFMA operands from the fp64 block instead -- 6 waves -- gave 409 k), else the fp64 block alone
  const int force = (pl->long_tables >= 0 && pl->long_tables <= 2) ? pl->long_tables : -1;
  const int w2 = long_tables_waves(pl->K, pl->nq, 16 * NB, true), w1 = long_tables_waves(pl->K, pl->nq, 16 * NB, false);
  int mode = force >= 0 ? force : (w2 ? 2 : (w1 ? 1 : 0));
  if ((mode == 2 && !w2) || (mode == 1 && !w1) || mode > 2) mode = 0;
  const int wpb = mode == 2 ? w2 : (mode ? w1 : pl->wpb_lm);
  const LongLds L = mode ? long_lds_layout(16 * NB, wpb, pl->K, pl->nq, mode == 2) : long_lds_layout(16 * NB, wpb);
  const int blocks = B < pl->n_cu ? B : pl->n_cu;
  int32_t *queue = ctx->counter_dev + 8;
  const int32_t *order = ga.pos == nullptr ? handout_order(pl, B) : nullptr;
  // the segment formulation (fit_seg.h) is the default; d2d_fit_plan_opts.long_tables >= 0 selects the table kernels (A/B)
  const bool seg = pl->long_tables < 0;
  SegArgs sa{};
  if (seg) {
    sa = seg_args_of(pl, NB);
    static const bool want_stamps = getenv("D2D_LM_STAMPS") != nullptr;
    if (want_stamps) {
      sa.stamps = reinterpret_cast<unsigned long long *>(ctx->stats_dev + 8);
      D2D_CHECK_HIP(hipMemsetAsync(sa.stamps, 0, 8 * sizeof(unsigned long long), ctx->stream));
    }
  }
  const int wpb_l = seg ? FIT_LM_WPB_MAX : wpb;
  const int lds_l = seg ? sa.L.total : L.total;
#define LAUNCH_LONG__(NBV, NQV, TLV, TL32V, MODEV, SEGV)                                                               \
  hipLaunchKernelGGL((fit_lm_long_kernel<NBV, NQV, TLV, TL32V, MODEV, SEGV>), dim3(blocks), dim3(64 * wpb_l), lds_l, ctx->stream, B, gm, L, o, budget, \
                     pl->d_GT, pl->d_G, pl->d_pk, pl->d_G32, pl->d_W32, pl->d_prep, q, pl->d_cost, pl->d_g, pl->d_lm, pl->d_flags, queue, order, ga, sa)
#define LAUNCH_LONG_(NBV, NQV, TLV, TL32V, SEGV) do { if (o.mode == D2D_LM_MODE_FAST) LAUNCH_LONG__(NBV, NQV, TLV, TL32V, D2D_LM_MODE_FAST, SEGV); else LAUNCH_LONG__(NBV, NQV, TLV, TL32V, D2D_LM_MODE_MINPACK, SEGV); } while (0)
#define LAUNCH_LONG(NBV, NQV) do { if (seg) LAUNCH_LONG_(NBV, NQV, false, false, true); else if (mode == 2) LAUNCH_LONG_(NBV, NQV, true, true, false); else if (mode == 1) LAUNCH_LONG_(NBV, NQV, true, false, false); else LAUNCH_LONG_(NBV, NQV, false, false, false); } while (0)
  if (pl->nq == 24) LAUNCH_LONG(3, 24);
  else if (NB == 1) LAUNCH_LONG(1, 0);
  else if (NB == 2) LAUNCH_LONG(2, 0);
  else LAUNCH_LONG(3, 0);
#undef LAUNCH_LONG__
#undef LAUNCH_LONG_
#undef LAUNCH_LONG
  D2D_LAUNCH_CHECK();
  if (seg && sa.stamps) {
    unsigned long long h[8];
    D2D_CHECK_HIP(hipMemcpyAsync(h, sa.stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    static const char *nm[8] = {"coefs", "rows(trial)", "rows(eval)", "mfma", "JTr", "project", "image", "solve+rest"};
    double tot = 0;
    for (int i = 0; i < 8; ++i) tot += (double)h[i];
    fprintf(stderr, "[fit_lm_long seg stamps] K=%d B=%d:", pl->K, B);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)h[i] / tot);
    fprintf(stderr, " total=%.3e\n", tot);
  }
  return D2D_OK;
}

template <typename KernelT>
static void allow_big_lds(KernelT k) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, FIT_LDS_BYTES);
}

static d2d_fit_opts opts_or_default(const d2d_fit_opts *opts) {
  d2d_fit_opts o = {200, 8, 1e-14, 1e-9, 1e-11, D2D_LM_SO_LAMBDA, D2D_LM_MODE_MINPACK, D2D_LM_MP_FINISH, 1e-15, 1e-15, 1e-15, D2D_LM_SLICE, D2D_LM_MP_SLOW,
                    D2D_HANDOUT_PREDICTED, D2D_LM_PRIO_AT, 1, D2D_GS_LS_SWEEP0, D2D_GS_LS_RATIO, D2D_GS_PRIO_AT, 0};
  if (opts) o = *opts;
  return o;
}

extern "C" {

int d2d_fit_plan_create(d2d_ctx *ctx, int S, int K, double duration, const double *wref,
                        d2d_fit_plan **out) {
  return d2d_fit_plan_create_ex(ctx, S, K, duration, wref, nullptr, out);
}

int d2d_fit_opts_default(d2d_fit_opts *o) {
  D2D_REQUIRE(o != nullptr, "d2d_fit_opts_default: o is NULL");
  *o = opts_or_default(nullptr);
  return D2D_OK;
}

int d2d_fit_plan_create_ex(d2d_ctx *ctx, int S, int K, double duration, const double *wref, const d2d_fit_plan_opts *popts,
                           d2d_fit_plan **out) {
  D2D_REQUIRE(ctx && wref && out, "d2d_fit_plan_create: null argument");
  const int kreq = popts ? popts->kernel : D2D_FIT_KERNEL_AUTO, ltab = popts ? popts->long_tables : -1;
  D2D_REQUIRE(kreq >= D2D_FIT_KERNEL_AUTO && kreq <= D2D_FIT_KERNEL_KNOT, "d2d_fit_plan_create_ex: kernel=%d is not a D2D_FIT_KERNEL_* value", kreq);
  D2D_REQUIRE(ltab >= -1 && ltab <= 3, "d2d_fit_plan_create_ex: long_tables=%d not in -1..3", ltab);
  D2D_REQUIRE(S >= 1 && S <= D2D_FIT_MAX_S, "d2d_fit_plan_create: S=%d not in 1..%d", S, D2D_FIT_MAX_S);
  D2D_REQUIRE(K >= 8 * S / 3 + 2, "d2d_fit_plan_create: K=%d too small for S=%d", K, S);
  D2D_REQUIRE(duration > 0, "d2d_fit_plan_create: duration must be > 0");
  d2d_fit_plan *pl = new d2d_fit_plan();
  pl->device = ctx->device;
  pl->S = S; pl->K = K; pl->duration = duration;
  pl->kernel_req = kreq; pl->long_tables = ltab;
  for (int i = 0; i < 3; ++i) pl->wref[i] = wref[i];
  if (int rc = fit_basis_build(pl)) { delete pl; return rc; }
  if (int rc = fit_basis_segments(pl)) { delete pl; return rc; }
  const int nq = pl->nq, gstr = nq + 1;
  D2D_CHECK_HIP(hipSetDevice(ctx->device));
  // the split-path kernels (public d2d_fit_eval, coupled groups) stage the whole basis block in LDS: K <~ 229 at S = 6
  pl->split_ok = pick_eval_layout(K, nq, &pl->g32_lds, &pl->wpb_eval, 0) &&
                 pick_step_layout(K, nq, 16 * ((2 * nq + 15) / 16), &pl->wpb_step);
  const bool want_split = kreq == D2D_FIT_KERNEL_SPLIT;
  if (want_split && !pl->split_ok) {
    d2d_set_error("d2d_fit_plan_create_ex: D2D_FIT_KERNEL_SPLIT: K=%d, S=%d does not fit the LDS image of the launch-pair kernels", K, S);
    delete pl;
    return D2D_EINVAL;
  }
  pl->use_lm = (nq == 24) && pick_fused_layout(K, nq, 48, &pl->wpb_lm) && !want_split;
  if ((kreq == D2D_FIT_KERNEL_FUSED || kreq == D2D_FIT_KERNEL_KNOT) && !pl->use_lm) {
    d2d_set_error("d2d_fit_plan_create_ex: the fused / knot kernels serve S = 6 with K <= 64 only (S=%d, K=%d)", S, K);
    delete pl;
    return D2D_EINVAL;
  }
  // everything else -- long horizons, segment counts other than six -- runs on the chunked persistent kernel (the segment
  // formulation needs no K-sized table and deals its lanes to any number of segments); D2D_FIT_KERNEL_LONG forces it for S = 6,
  // K <= 64 (tests do), D2D_FIT_KERNEL_SPLIT selects the launch-pair path where its LDS image holds K
  pl->use_long = !pl->use_lm && !want_split;
  if (kreq == D2D_FIT_KERNEL_LONG) { pl->use_lm = false; pl->use_long = true; }
  if (pl->use_long) pl->wpb_lm = FIT_LM_WPB_MAX;
  // the headline shape (S = 6, K <= 64) runs the default solver in knot coordinates (fit_knot.hip); D2D_FIT_KERNEL_FUSED: in q.
  // A plan whose knot tables cannot be built (fit_basis_knots rejects the metric, an upload fails) keeps the fused q kernel --
  // unless the caller asked for the knot kernel by name
  if (pl->use_lm && kreq != D2D_FIT_KERNEL_FUSED) {
    const int rc = fit_knot_plan_init(pl);
    if (rc != D2D_OK) fit_knot_plan_free(pl);
    if ((rc != D2D_OK || pl->kn.wpb == 0) && kreq == D2D_FIT_KERNEL_KNOT) {
      if (rc == D2D_OK) d2d_set_error("d2d_fit_plan_create_ex: D2D_FIT_KERNEL_KNOT: the knot kernel's LDS layout does not hold K=%d", K);
      delete pl;
      return rc != D2D_OK ? rc : D2D_EINVAL;
    }
    if (rc != D2D_OK) fprintf(stderr, "[d2d] fit plan S=%d K=%d: knot tables not built (%s); the fused q kernel serves the default solver\n", S, K, d2d_last_error());
  }
  if (!pl->split_ok && !pl->use_long) {
    d2d_set_error("d2d_fit_plan_create: K=%d, S=%d does not fit the 160 KiB LDS image of the basis block", K, S);
    delete pl;
    return D2D_EINVAL;
  }
  {
    hipDeviceProp_t prop;
    D2D_CHECK_HIP(hipGetDeviceProperties(&prop, ctx->device));
    pl->n_cu = prop.multiProcessorCount;
  }
  // device images: G64 with odd row stride, G32 interleaved (G0,G1,G2,0), W32 = G0^T G0
  std::vector<double> g64((size_t)3 * K * gstr, 0.0);
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < K; ++k)
      for (int j = 0; j < nq; ++j) g64[((size_t)d * K + k) * gstr + j] = pl->G[((size_t)d * K + k) * nq + j];
  std::vector<float> g32((size_t)(3 * K + 1) * nq, 0.f), w32((size_t)nq * nq);
  for (size_t i = 0; i < (size_t)3 * K * nq; ++i) g32[i] = (float)pl->G[i];        // [d][k][j] planes + one zero row
  {
    // waypoint rows' constant J^T J block, pre-arranged in the MFMA C/D tile layout
    const int n = 2 * nq, NBh = (n + 15) / 16, NT = NBh * (NBh + 1) / 2;
    w32.assign((size_t)NT * 256, 0.f);
    int t = 0;
    for (int I = 0; I < NBh; ++I)
      for (int J = I; J < NBh; ++J, ++t)
        for (int r = 0; r < 4; ++r)
          for (int l = 0; l < 64; ++l) {
            const int row = 16 * I + 4 * (l >> 4) + r, col = 16 * J + (l & 15);
            if (row < n && col < n && (row >= nq) == (col >= nq))
              w32[((size_t)t * 4 + r) * 64 + l] = (float)pl->G0tG0[(size_t)(row % nq) * nq + (col % nq)];
          }
  }
  std::vector<double> gt((size_t)3 * nq * K);            // [d][j][k]: the chunked kernel's flat-output pass reads 64 consecutive k
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k < K; ++k)
      for (int j = 0; j < nq; ++j) gt[((size_t)d * nq + j) * K + k] = pl->G[((size_t)d * K + k) * nq + j];
  D2D_CHECK_HIP(hipSetDevice(ctx->device));
  int rc = upload(&pl->d_G, g64);
  if (!rc) rc = upload(&pl->d_GT, gt);
  if (!rc) rc = upload(&pl->d_Gp, pl->Gp);
  if (!rc) rc = upload(&pl->d_G32, g32);
  if (!rc) rc = upload(&pl->d_W32, w32);
  if (!rc) rc = upload(&pl->d_Z, pl->Z);
  if (!rc) rc = upload(&pl->d_Zp, pl->Zp);
  if (!rc) rc = upload(&pl->d_Pinit, pl->Pinit);
  {
    std::vector<double> zl64((size_t)8 * S * gstr, 0.0);
    std::vector<float> zl32((size_t)8 * S * nq);
    for (int i = 0; i < 8 * S; ++i)
      for (int j = 0; j < nq; ++j) { zl64[(size_t)i * gstr + j] = pl->Zl[(size_t)i * nq + j]; zl32[(size_t)i * nq + j] = (float)pl->Zl[(size_t)i * nq + j]; }
    if (!rc) rc = upload(&pl->d_Zl64, zl64);
    if (!rc) rc = upload(&pl->d_Zl32, zl32);
    if (!rc) rc = upload(&pl->d_Zlp, pl->Zlp);
    if (!rc) rc = upload(&pl->d_sx, pl->sx);
  }
  if (!rc) {
    std::vector<float> prior(&kHandoutPrior[0][0][0], &kHandoutPrior[0][0][0] + 2 * D2D_HANDOUT_NB * D2D_HANDOUT_ND);
    rc = upload(&pl->d_hprior, prior);
    // (regressed on the fused shape's default solver: it knows nothing about other horizons -- measured rank correlation with the
    // trial counts of the 121- and 301-node bench families: -0.13 / 0.06 -- so plans of the long-horizon kernel hand out in index order
    // until the caller installs a prior of their own, d2d_fit_plan_set_handout_prior)
    pl->has_prior = pl->use_lm;
  }
  if (rc) { d2d_fit_plan_destroy(pl); return rc; }
  // opt in to large dynamic LDS
  allow_big_lds(&fit_eval_kernel<1, 0, true>); allow_big_lds(&fit_eval_kernel<2, 0, true>); allow_big_lds(&fit_eval_kernel<3, 0, true>);
  allow_big_lds(&fit_eval_kernel<1, 0, false>); allow_big_lds(&fit_eval_kernel<2, 0, false>); allow_big_lds(&fit_eval_kernel<3, 0, false>);
  allow_big_lds(&fit_eval_kernel<3, 24, true>); allow_big_lds(&fit_eval_kernel<3, 24, false>);
  allow_big_lds(&fit_jtj_kernel<1, 0>); allow_big_lds(&fit_jtj_kernel<2, 0>); allow_big_lds(&fit_jtj_kernel<3, 0>); allow_big_lds(&fit_jtj_kernel<3, 24>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, false, false, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<3, 0, false, false, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<2, 0, false, false, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<1, 0, false, false, D2D_LM_MODE_MINPACK>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, false, false, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<3, 0, false, false, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<2, 0, false, false, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<1, 0, false, false, D2D_LM_MODE_FAST>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, true, true, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<3, 0, true, true, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<2, 0, true, true, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<1, 0, true, true, D2D_LM_MODE_MINPACK>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, true, true, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<3, 0, true, true, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<2, 0, true, true, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<1, 0, true, true, D2D_LM_MODE_FAST>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, true, false, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<3, 0, true, false, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<2, 0, true, false, D2D_LM_MODE_MINPACK>); allow_big_lds(&fit_lm_long_kernel<1, 0, true, false, D2D_LM_MODE_MINPACK>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, true, false, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<3, 0, true, false, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<2, 0, true, false, D2D_LM_MODE_FAST>); allow_big_lds(&fit_lm_long_kernel<1, 0, true, false, D2D_LM_MODE_FAST>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, false, false, D2D_LM_MODE_MINPACK, true>); allow_big_lds(&fit_lm_long_kernel<3, 0, false, false, D2D_LM_MODE_MINPACK, true>); allow_big_lds(&fit_lm_long_kernel<2, 0, false, false, D2D_LM_MODE_MINPACK, true>); allow_big_lds(&fit_lm_long_kernel<1, 0, false, false, D2D_LM_MODE_MINPACK, true>);
  allow_big_lds(&fit_lm_long_kernel<3, 24, false, false, D2D_LM_MODE_FAST, true>); allow_big_lds(&fit_lm_long_kernel<3, 0, false, false, D2D_LM_MODE_FAST, true>); allow_big_lds(&fit_lm_long_kernel<2, 0, false, false, D2D_LM_MODE_FAST, true>); allow_big_lds(&fit_lm_long_kernel<1, 0, false, false, D2D_LM_MODE_FAST, true>);
  allow_big_lds(&fit_eval_seg_kernel<3, 24>); allow_big_lds(&fit_eval_seg_kernel<3, 0>); allow_big_lds(&fit_eval_seg_kernel<2, 0>); allow_big_lds(&fit_eval_seg_kernel<1, 0>);
  allow_big_lds(&fit_groups_kernel<3, 24>);
  allow_big_lds(&fit_lm_kernel<3, 24, false, D2D_LM_MODE_MINPACK>);
  allow_big_lds(&fit_lm_kernel<3, 24, true, D2D_LM_MODE_MINPACK>);
  allow_big_lds(&fit_lm_kernel<3, 24, false, D2D_LM_MODE_FAST>);
  allow_big_lds(&fit_lm_kernel<3, 24, true, D2D_LM_MODE_FAST>);
  allow_big_lds(&fit_step_kernel<16>); allow_big_lds(&fit_step_kernel<32>); allow_big_lds(&fit_step_kernel<48>);
  (void)hipGetLastError();
  *out = pl;
  return D2D_OK;
}

int d2d_fit_plan_destroy(d2d_fit_plan *pl) {
  if (!pl) return D2D_OK;
  hipSetDevice(pl->device);
  for (hipEvent_t e : pl->prof_ev) (void)hipEventDestroy(e);
  void *ptrs[] = {pl->d_G, pl->d_GT, pl->d_Gp, pl->d_G32, pl->d_W32, pl->d_Z, pl->d_Zp, pl->d_Pinit, pl->d_Zl64, pl->d_Zl32, pl->d_Zlp, pl->d_sx, pl->d_hprior};
  for (void *p : ptrs)
    if (p) hipFree(p);
  free_scratch(pl);
  fit_knot_plan_free(pl);
  delete pl;
  return D2D_OK;
}

int d2d_fit_plan_kernel(const d2d_fit_plan *pl) {
  if (!pl) return D2D_EINVAL;
  return pl->use_lm ? (pl->kn.wpb > 0 ? D2D_FIT_KERNEL_KNOT : D2D_FIT_KERNEL_FUSED) : (pl->use_long ? D2D_FIT_KERNEL_LONG : D2D_FIT_KERNEL_SPLIT);
}

int d2d_fit_plan_get(const d2d_fit_plan *pl, double *G, double *Gp, double *Z, double *Zp, double *Pinit) {
  D2D_REQUIRE(pl != nullptr, "d2d_fit_plan_get: plan is NULL");
  if (G) std::memcpy(G, pl->G.data(), pl->G.size() * sizeof(double));
  if (Gp) std::memcpy(Gp, pl->Gp.data(), pl->Gp.size() * sizeof(double));
  if (Z) std::memcpy(Z, pl->Z.data(), pl->Z.size() * sizeof(double));
  if (Zp) std::memcpy(Zp, pl->Zp.data(), pl->Zp.size() * sizeof(double));
  if (Pinit) std::memcpy(Pinit, pl->Pinit.data(), pl->Pinit.size() * sizeof(double));
  return D2D_OK;
}

int d2d_fit_init(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *scen, double *q) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_init: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_init: B must be >= 1");
  hipLaunchKernelGGL(fit_init_kernel, dim3((B + 3) / 4), dim3(256), 0, ctx->stream, B, pl->K, pl->nq, pl->duration,
                     pl->d_Gp, pl->d_Pinit, scen, q);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_fit_project(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *scen, const double *xy, double *q) {
  D2D_REQUIRE(ctx && pl && scen && xy && q, "d2d_fit_project: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_project: B must be >= 1");
  hipLaunchKernelGGL(fit_project_kernel, dim3((B + 3) / 4), dim3(256), 0, ctx->stream, B, pl->K, pl->nq, pl->duration,
                     pl->d_Gp, pl->d_Pinit, scen, xy, q);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

static int prof_begin(d2d_ctx *ctx, d2d_fit_plan *pl, int kind);
static int prof_end(d2d_ctx *ctx, d2d_fit_plan *pl);

int d2d_fit_eval(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *scen, const double *q,
                 double *cost, double *g, float *H) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_eval: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_eval: B must be >= 1");
  D2D_REQUIRE(pl->split_ok || pl->n_group <= 1, "d2d_fit_eval: K=%d with coupled groups does not fit the LDS image of the evaluation kernel", pl->K);
  d2d_fit_plan *plm = const_cast<d2d_fit_plan *>(pl);
  if (plm->active_B != 0 && B > plm->cap_B) { d2d_set_error("d2d_fit_eval: a solve of a smaller batch is in progress on this plan"); return D2D_ESTATE; }
  if (int rc = ensure_scratch(plm, B)) return rc;
  if (int rc = launch_prep(ctx, pl, B, scen)) return rc;
  if (int rc = prof_begin(ctx, plm, 0)) return rc;       // (d2d_fit_profile: the J^T J kernel alone, every trajectory active)
  if (pl->split_ok) {
    if (int rc = launch_eval(ctx, pl, B, q, nullptr, cost, g, H ? plm->d_H : nullptr)) return rc;
  } else {
    // a horizon the LDS image of fit_eval_kernel cannot hold: the same evaluation in the segment formulation (no K-sized table)
    const int NB = (2 * pl->nq + 15) / 16;
    const SegArgs sa = seg_args_of(pl, NB);
    const FitGeom gm = geom_of(pl);
    const int wpb = FIT_LM_WPB_MAX;
    int blocks = (B + wpb - 1) / wpb;
    if (blocks > pl->n_cu) blocks = pl->n_cu;
#define LAUNCH_EVAL_SEG(NBV, NQV) hipLaunchKernelGGL((fit_eval_seg_kernel<NBV, NQV>), dim3(blocks), dim3(64 * wpb), sa.L.total, ctx->stream, B, gm, sa, \
                                                     pl->d_pk, pl->d_prep, q, cost, g, H ? plm->d_H : nullptr)
    if (pl->nq == 24) LAUNCH_EVAL_SEG(3, 24);
    else if (NB == 1) LAUNCH_EVAL_SEG(1, 0);
    else if (NB == 2) LAUNCH_EVAL_SEG(2, 0);
    else LAUNCH_EVAL_SEG(3, 0);
#undef LAUNCH_EVAL_SEG
    D2D_LAUNCH_CHECK();
  }
  if (int rc = prof_end(ctx, plm)) return rc;
  plm->prep_valid_for = nullptr;
  if (H) {
    hipLaunchKernelGGL(untile_kernel, dim3(B), dim3(256), 0, ctx->stream, B, 2 * pl->nq, (2 * pl->nq + 15) / 16, plm->d_H, pl->d_W32, pl->d_prep, H);
    D2D_LAUNCH_CHECK();
  }
  return D2D_OK;
}


// The two halves of d2d_fit_eval as separate launches (bench.py's contraction-only roofline): d2d_fit_rows evaluates the
// rows at q (cost, J^T r) and leaves the fp32 row records in the plan's scratch; d2d_fit_jtj contracts them.
int d2d_fit_rows(d2d_ctx *ctx, d2d_fit_plan *pl, int B, const double *scen, const double *q, double *cost, double *g) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_rows: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_rows: B must be >= 1");
  D2D_REQUIRE(pl->n_group <= 1, "d2d_fit_rows: not available for coupled groups");
  D2D_REQUIRE(pl->split_ok, "d2d_fit_rows: K=%d does not fit the LDS image of the evaluation kernel", pl->K);
  if (pl->active_B != 0 && B > pl->cap_B) { d2d_set_error("d2d_fit_rows: a solve of a smaller batch is in progress on this plan"); return D2D_ESTATE; }
  if (int rc = ensure_scratch(pl, B)) return rc;
  if (int rc = launch_prep(ctx, pl, B, scen)) return rc;
  pl->prep_valid_for = nullptr;
  if (int rc = launch_eval(ctx, pl, B, q, nullptr, cost, g, nullptr, no_groups(), reinterpret_cast<f32x4 *>(pl->d_rows))) return rc;
  pl->rows_B = B;
  return D2D_OK;
}

int d2d_fit_jtj(d2d_ctx *ctx, d2d_fit_plan *pl, int B, float *H) {
  D2D_REQUIRE(ctx && pl, "d2d_fit_jtj: null argument");
  if (B < 1 || B != pl->rows_B || B > pl->cap_B) {
    d2d_set_error("d2d_fit_jtj: call d2d_fit_rows(B=%d) on this plan first (records held: %d)", B, pl->rows_B);
    return D2D_ESTATE;
  }
  const int NB = (2 * pl->nq + 15) / 16;
  // launch geometry (A/B knobs for tools/bench_jtj.py: D2D_JTJ_GEOM="wpb,wgs": wavefronts per workgroup, workgroups per CU)
  static int env_wpb = 0, env_wgs = 0;
  static const bool geom_read = [] { if (const char *g = getenv("D2D_JTJ_GEOM")) sscanf(g, "%d,%d", &env_wpb, &env_wgs); return true; }();
  (void)geom_read;
  int wpb = (env_wpb >= 1 && env_wpb <= FIT_EVAL_WPB_MAX) ? env_wpb : FIT_EVAL_WPB_MAX;
  while (wpb > 1 && jtj_lds_layout(pl->K, pl->nq, wpb).total > FIT_LDS_BYTES) --wpb;
  const JtjLds L = jtj_lds_layout(pl->K, pl->nq, wpb);
  D2D_REQUIRE(L.total <= FIT_LDS_BYTES, "d2d_fit_jtj: K=%d does not fit the LDS", pl->K);
  int nblk = (B + wpb - 1) / wpb;
  const int max_blk = pl->n_cu * (env_wgs >= 1 ? env_wgs : 1);
  if (nblk > max_blk) nblk = max_blk;
  const FitGeom gm = geom_of(pl);
  const f32x4 *rows = reinterpret_cast<const f32x4 *>(pl->d_rows);
  if (int rc = prof_begin(ctx, pl, 3)) return rc;
  static const bool want_clock = getenv("D2D_JTJ_CLOCK") != nullptr;
  unsigned long long *clk = want_clock ? reinterpret_cast<unsigned long long *>(ctx->stats_dev + 24) : nullptr;
  if (want_clock) D2D_CHECK_HIP(hipMemsetAsync(clk, 0, 2 * sizeof(unsigned long long), ctx->stream));
#define LAUNCH_JTJ(NBV, NQV) \
  hipLaunchKernelGGL((fit_jtj_kernel<NBV, NQV>), dim3(nblk), dim3(64 * wpb), L.total, ctx->stream, B, gm, L, pl->d_G32, rows, pl->d_H, clk)
  if (pl->nq == 24) LAUNCH_JTJ(3, 24);
  else if (NB == 1) LAUNCH_JTJ(1, 0);
  else if (NB == 2) LAUNCH_JTJ(2, 0);
  else LAUNCH_JTJ(3, 0);
#undef LAUNCH_JTJ
  D2D_LAUNCH_CHECK();
  if (int rc = prof_end(ctx, pl)) return rc;
  if (want_clock) {
    unsigned long long h[2];
    D2D_CHECK_HIP(hipMemcpyAsync(h, clk, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[fit_jtj clock] B=%d: %.3f GHz in-kernel (s_memtime / s_memrealtime over the trajectory loop, %d workgroups)\n", B,
            h[1] ? 0.1 * (double)h[0] / (double)h[1] : 0.0, nblk);
  }
  if (H) {
    hipLaunchKernelGGL(untile_kernel, dim3(B), dim3(256), 0, ctx->stream, B, 2 * pl->nq, NB, pl->d_H, pl->d_W32, pl->d_prep, H);
    D2D_LAUNCH_CHECK();
  }
  return D2D_OK;
}

// ---- profiling: HIP event pairs around every eval / step launch of the LM loop -------
static int prof_begin(d2d_ctx *ctx, d2d_fit_plan *pl, int kind) {
  if (!pl->prof_on) return D2D_OK;
  hipEvent_t a, b;
  D2D_CHECK_HIP(hipEventCreate(&a));
  D2D_CHECK_HIP(hipEventCreate(&b));
  pl->prof_ev.push_back(a);
  pl->prof_ev.push_back(b);
  pl->prof_kind.push_back(kind);
  D2D_CHECK_HIP(hipEventRecord(a, ctx->stream));
  return D2D_OK;
}
static int prof_end(d2d_ctx *ctx, d2d_fit_plan *pl) {
  if (!pl->prof_on) return D2D_OK;
  D2D_CHECK_HIP(hipEventRecord(pl->prof_ev.back(), ctx->stream));
  return D2D_OK;
}

int d2d_fit_profile(d2d_fit_plan *pl, int enable) {
  D2D_REQUIRE(pl != nullptr, "d2d_fit_profile: plan is NULL");
  for (hipEvent_t e : pl->prof_ev) (void)hipEventDestroy(e);
  pl->prof_ev.clear();
  pl->prof_kind.clear();
  pl->prof_on = enable != 0;
  return D2D_OK;
}

int d2d_fit_profile_read(d2d_fit_plan *pl, double *out) {
  D2D_REQUIRE(pl && out, "d2d_fit_profile_read: null argument");
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  for (size_t i = 0; i < pl->prof_kind.size(); ++i) {
    D2D_CHECK_HIP(hipEventSynchronize(pl->prof_ev[2 * i + 1]));
    float ms = 0.f;
    D2D_CHECK_HIP(hipEventElapsedTime(&ms, pl->prof_ev[2 * i], pl->prof_ev[2 * i + 1]));
    out[2 * pl->prof_kind[i]] += ms;
    out[2 * pl->prof_kind[i] + 1] += 1.0;
  }
  return D2D_OK;
}


int d2d_fit_begin(d2d_ctx *ctx, d2d_fit_plan *pl, int B) {
  D2D_REQUIRE(ctx && pl, "d2d_fit_begin: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_begin: B must be >= 1");
  if (pl->active_B != 0 && B > pl->cap_B) {
    d2d_set_error("d2d_fit_begin: a begin/iterate sequence of %d trajectories is in progress on this plan (finish it first)", pl->active_B);
    return D2D_ESTATE;
  }
  if (int rc = ensure_scratch(pl, B)) return rc;
  hipLaunchKernelGGL(fit_state_init_kernel, dim3((B + 255) / 256), dim3(256), 0, ctx->stream, B, 0, 1, pl->d_lm, pl->d_flags, ctx->counter_dev + 8);
  D2D_LAUNCH_CHECK();
  D2D_CHECK_HIP(hipMemsetAsync(pl->d_ring, 0xff, (size_t)pl->ring_cap * sizeof(int32_t), ctx->stream));
  pl->it_done = 0;
  pl->active_B = B;
  pl->solve_q = nullptr; pl->solve_kernel = -1;
  pl->last_order = nullptr; pl->last_order_B = 0;
  pl->prep_valid_for = nullptr;
  pl->rows_B = 0;              // (a solve rewrites d_prep and, on the launch-pair path, d_H)
  pl->last_slice = 0; pl->last_running = -1;
  pl->gsweeps_R = 0;           // (... and d_lm: the sweeps / moves of an earlier d2d_fit_solve_groups are gone -- d2d_fit_group_report says D2D_ESTATE)
  return D2D_OK;
}

int d2d_fit_iterate(d2d_ctx *ctx, d2d_fit_plan *pl, int B, const double *scen, double *q,
                    const d2d_fit_opts *opts, int n_iters, int32_t *n_running) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_iterate: null argument");
  if (pl->active_B != B) { d2d_set_error("d2d_fit_iterate: call d2d_fit_begin(B=%d) first", B); return D2D_ESTATE; }
  const d2d_fit_opts o = opts_or_default(opts);
  D2D_REQUIRE(o.max_iter >= 1 && n_iters >= 1, "d2d_fit_iterate: max_iter and n_iters must be >= 1");
  const dim3 g1((B + 255) / 256), b1(256);
  if (pl->prep_valid_for != scen) {
    if (int rc = launch_prep(ctx, pl, B, scen)) return rc;
    pl->prep_valid_for = scen;
  }
  if ((pl->use_lm || pl->use_long) && pl->n_group <= 1) {
    // the solve owns q and stays on the kernel that holds its state (include/d2d.h: d2d_fit_iterate)
    const int sk = solve_kernel_of(pl, o);
    if (pl->solve_kernel < 0) {
      pl->solve_q = q; pl->solve_kernel = sk;
      // hand-out of this solve: the caller's explicit hint, else (default) longest-first by the predicted trial counts when the batch
      // exceeds the resident wavefronts (<= 8 per CU) -- below that every fit starts at once and the order is immaterial
      pl->last_order = pl->order_B == B ? pl->d_order : nullptr;
      if (pl->order_B != B && pl->has_prior && o.handout == D2D_HANDOUT_PREDICTED && o.mode == D2D_LM_MODE_MINPACK && B > 8 * pl->n_cu) {
        if (int rc = launch_handout(ctx, pl, B, scen)) return rc;
        pl->last_order = pl->d_order_pred;
      }
      pl->last_order_B = pl->last_order ? B : 0;
    } else {
      if (q != pl->solve_q) {
        d2d_set_error("d2d_fit_iterate: q differs from the buffer this solve started on (between d2d_fit_begin and d2d_fit_finish the solve owns q)");
        return D2D_EINVAL;
      }
      if (sk != pl->solve_kernel) {
        d2d_set_error("d2d_fit_iterate: these options (mode / slice) would move the solve in progress to another kernel; finish it and begin again");
        return D2D_ESTATE;
      }
    }
    int budget = o.max_iter - pl->it_done;
    if (budget > n_iters) budget = n_iters;
    // Time-sliced hand-out: a fit that a launch left in its ring keeps D2D_ST_RUNNING with its own iteration count below the
    // cap.  Once the plan's budget is spent such fits are finished by further launches under the full cap (every fit stops at
    // max_iter iterations of its own) -- without them the caller's `while (running > 0)` would spin on a count that cannot fall.
    pl->last_slice = pl->use_lm ? lm_slice(o) : 0;
    pl->last_opts = o;
    const bool sweep_up = budget <= 0 && pl->last_slice > 0 && pl->last_running != 0;     // (nothing left RUNNING by the last count: no launch)
    if (budget > 0 || sweep_up) {
      if (int rc = prof_begin(ctx, pl, 2)) return rc;
      if (int rc = pl->use_lm ? launch_lm(ctx, pl, B, q, o, sweep_up ? o.max_iter : pl->it_done + budget) : launch_lm_long(ctx, pl, B, q, o, budget)) return rc;
      if (int rc = prof_end(ctx, pl)) return rc;
      if (budget > 0) pl->it_done += budget;
    }
  } else
  for (int i = 0; i < n_iters && pl->it_done < o.max_iter; ++i, ++pl->it_done) {
    if (int rc = prof_begin(ctx, pl, 0)) return rc;
    if (int rc = launch_eval(ctx, pl, B, q, pl->d_flags, pl->d_cost, pl->d_g, pl->d_H)) return rc;
    if (int rc = prof_end(ctx, pl)) return rc;
    if (int rc = prof_begin(ctx, pl, 1)) return rc;
    if (int rc = launch_step(ctx, pl, B, q, o)) return rc;
    if (int rc = prof_end(ctx, pl)) return rc;
  }
  if (n_running) {
    if (pl->it_done >= o.max_iter && !(pl->use_lm && lm_slice(o) > 0)) {
      *n_running = 0;                    // the iteration budget is spent: nothing to count (d2d_fit_finish synchronises)
    } else {                             // (with the time-sliced hand-out the flags are counted even then: a fit left in the ring would show)
      D2D_CHECK_HIP(hipMemsetAsync(ctx->counter_dev, 0, sizeof(int32_t), ctx->stream));
      hipLaunchKernelGGL(fit_count_kernel, g1, b1, 0, ctx->stream, B, pl->d_flags, ctx->counter_dev);
      D2D_LAUNCH_CHECK();
      D2D_CHECK_HIP(hipMemcpyAsync(ctx->counter_host, ctx->counter_dev, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
      D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      *n_running = ctx->counter_host[0];
      pl->last_running = *n_running;
    }
  }
  return D2D_OK;
}

// Time-sliced hand-out only: fits a launch left in its ring (a bounded wait that gave up) keep D2D_ST_RUNNING below their cap.  The
// caller may never look at n_running (one d2d_fit_iterate, then d2d_fit_finish), so d2d_fit_finish sweeps them up itself: launches
// under the cap the plan has reached (it_done: a fit that is RUNNING with that many iterations of its own was stopped by the
// caller's budget, not left behind) until no fit below it is RUNNING; if that does not happen it says so instead of reporting them.
static int sweep_up_sliced(d2d_ctx *ctx, d2d_fit_plan *pl, int B, double *q) {
  const dim3 g1((B + 255) / 256), b1(256);
  for (int tries = 0;; ++tries) {
    D2D_CHECK_HIP(hipMemsetAsync(ctx->counter_dev, 0, sizeof(int32_t), ctx->stream));
    const int cap = pl->it_done < pl->last_opts.max_iter ? pl->it_done : pl->last_opts.max_iter;
    hipLaunchKernelGGL(fit_count_below_kernel, g1, b1, 0, ctx->stream, B, pl->d_flags, cap, ctx->counter_dev);
    D2D_LAUNCH_CHECK();
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->counter_host, ctx->counter_dev, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->counter_host + 1, ctx->counter_dev + 8 + 5, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const int left = ctx->counter_host[0], gave_up = ctx->counter_host[1];
    if (left == 0) return D2D_OK;
    if (tries >= 8) {
      d2d_set_error("d2d_fit_finish: %d fits are still RUNNING below the iteration cap after %d sweep-up launches of the time-sliced hand-out "
                    "(%d bounded ring waits gave up during this solve)", left, tries, gave_up);
      return D2D_ESTATE;
    }
    if (int rc = launch_lm(ctx, pl, B, q, pl->last_opts, cap)) return rc;
  }
}

int d2d_fit_finish(d2d_ctx *ctx, d2d_fit_plan *pl, int B, const double *scen, const double *q,
                   double *cost, int32_t *iters, int32_t *status, double *stats) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_finish: null argument");
  if (pl->active_B != B) { d2d_set_error("d2d_fit_finish: call d2d_fit_begin(B=%d) first", B); return D2D_ESTATE; }
  if (pl->solve_q != nullptr && q != pl->solve_q) {
    d2d_set_error("d2d_fit_finish: q differs from the buffer the d2d_fit_iterate calls of this solve worked on");
    return D2D_EINVAL;
  }
  const dim3 g1((B + 255) / 256), b1(256);
  // converged trajectories carry a pending evaluation at the accepted point: refresh cost / J^T r
  if (pl->prep_valid_for != scen) {
    if (int rc = launch_prep(ctx, pl, B, scen)) return rc;
    pl->prep_valid_for = scen;
  }
  if (pl->use_lm && pl->n_group <= 1 && pl->last_slice > 0 && pl->it_done > 0)
    if (int rc = sweep_up_sliced(ctx, pl, B, const_cast<double *>(q))) return rc;
  // (the persistent LM kernel leaves cost and J^T r of every trajectory evaluated at its final point)
  if (!((pl->use_lm || pl->use_long) && pl->n_group <= 1 && pl->it_done > 0))
    if (int rc = launch_eval(ctx, pl, B, q, nullptr, pl->d_cost, pl->d_g, nullptr)) return rc;
  if (cost) D2D_CHECK_HIP(hipMemcpyAsync(cost, pl->d_cost, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(fit_export_kernel, g1, b1, 0, ctx->stream, B, pl->d_flags, iters, status);
  D2D_LAUNCH_CHECK();
  if (stats) {
    D2D_CHECK_HIP(hipMemsetAsync(ctx->stats_dev, 0, 4 * sizeof(double), ctx->stream));
    // (a thread reads its trajectory's whole gradient row: 64-thread blocks, more CUs -- as launch_prep)
    hipLaunchKernelGGL(fit_stats_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, B, 2 * pl->nq, pl->d_cost, pl->d_g, pl->d_flags, ctx->stats_dev);
    D2D_LAUNCH_CHECK();
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->stats_host, ctx->stats_dev, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (stats)
    for (int i = 0; i < 4; ++i) stats[i] = ctx->stats_host[i];
  pl->active_B = 0;
  return D2D_OK;
}

int d2d_fit_solve(d2d_ctx *ctx, const d2d_fit_plan *plc, int B, const double *scen, double *q,
                  const d2d_fit_opts *opts, double *cost, int32_t *iters, int32_t *status, double *stats) {
  D2D_REQUIRE(ctx && plc && scen && q, "d2d_fit_solve: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_solve: B must be >= 1");
  d2d_fit_plan *pl = const_cast<d2d_fit_plan *>(plc);
  const d2d_fit_opts o = opts_or_default(opts);
  D2D_REQUIRE(o.max_iter >= 1 && o.check_every >= 1, "d2d_fit_solve: max_iter and check_every must be >= 1");
  if (int rc = d2d_fit_begin(ctx, pl, B)) return rc;
  int32_t running = B;
  // the persistent LM kernel masks finished trajectories itself and ends when its last one stops: one
  // launch for the whole solve; the split path counts the running trajectories every check_every iterations
  const int per_call = ((pl->use_lm || pl->use_long) && pl->n_group <= 1) ? o.max_iter : o.check_every;
  for (int calls = 0; running > 0; ++calls) {
    if (calls > o.max_iter + 64) {        // (cannot happen: every call either spends budget or finishes fits a sliced launch left behind)
      d2d_set_error("d2d_fit_solve: %d trajectories still running after %d launches", (int)running, calls);
      return D2D_ESTATE;
    }
    if (int rc = d2d_fit_iterate(ctx, pl, B, scen, q, &o, per_call, &running)) return rc;
  }
  return d2d_fit_finish(ctx, pl, B, scen, q, cost, iters, status, stats);
}

int d2d_fit_plan_set_order(d2d_ctx *ctx, d2d_fit_plan *pl, int B, const int32_t *iters) {
  D2D_REQUIRE(ctx && pl, "d2d_fit_plan_set_order: null argument");
  if (iters == nullptr) { pl->order_B = 0; return D2D_OK; }
  D2D_REQUIRE(B >= 1, "d2d_fit_plan_set_order: B must be >= 1");
  if (pl->active_B != 0 && B > pl->cap_B) { d2d_set_error("d2d_fit_plan_set_order: a solve of a smaller batch is in progress on this plan"); return D2D_ESTATE; }
  if (int rc = ensure_scratch(pl, B)) return rc;
  hipLaunchKernelGGL(fit_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, B, iters, pl->d_order);
  D2D_LAUNCH_CHECK();
  pl->order_B = B; pl->gorder_R = 0; pl->gsweeps_R = 0;
  return D2D_OK;
}

int d2d_fit_plan_set_handout_prior(d2d_ctx *ctx, d2d_fit_plan *pl, const float *table) {
  D2D_REQUIRE(ctx && pl, "d2d_fit_plan_set_handout_prior: null argument");
  const size_t n = (size_t)2 * D2D_HANDOUT_NB * D2D_HANDOUT_ND;
  const float *src = table ? table : &kHandoutPrior[0][0][0];
  for (size_t i = 0; i < n; ++i)
    D2D_REQUIRE(src[i] == src[i] && fabsf(src[i]) <= 1e4f, "d2d_fit_plan_set_handout_prior: table[%zu] is not a finite trial count", i);
  D2D_CHECK_HIP(hipSetDevice(pl->device));
  // (synchronous with respect to the stream: an earlier solve may still be reading the table)
  D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  D2D_CHECK_HIP(hipMemcpy(pl->d_hprior, src, n * sizeof(float), hipMemcpyHostToDevice));
  pl->has_prior = table != nullptr || pl->use_lm;
  return D2D_OK;
}

int d2d_fit_plan_get_order(d2d_ctx *ctx, d2d_fit_plan *pl, int B, int32_t *order) {
  D2D_REQUIRE(ctx && pl && order, "d2d_fit_plan_get_order: null argument");
  if (pl->last_order == nullptr || pl->last_order_B != B || B > pl->cap_B) {
    d2d_set_error("d2d_fit_plan_get_order: the last solve launch of this plan was not handed out by an order over B=%d trajectories", B);
    return D2D_ESTATE;
  }
  D2D_CHECK_HIP(hipMemcpyAsync(order, pl->last_order, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return D2D_OK;
}

int d2d_fit_plan_set_group_order(d2d_ctx *ctx, d2d_fit_plan *pl, int R, int from_last) {
  D2D_REQUIRE(ctx && pl, "d2d_fit_plan_set_group_order: null argument");
  if (!from_last) { pl->gorder_R = 0; return D2D_OK; }
  D2D_REQUIRE(R >= 1 && pl->n_group >= 2, "d2d_fit_plan_set_group_order: needs R >= 1 and a plan in group mode");
  if (pl->gsweeps_R != R) {
    d2d_set_error("d2d_fit_plan_set_group_order: the last d2d_fit_solve_groups of this plan was not over R=%d scenarios", R);
    return D2D_ESTATE;
  }
  hipLaunchKernelGGL(fit_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, R, pl->d_order + R, pl->d_order);
  D2D_LAUNCH_CHECK();
  pl->gorder_R = R;
  return D2D_OK;
}

int d2d_fit_group_report(d2d_ctx *ctx, d2d_fit_plan *pl, int R, int32_t *sweeps, double *moved) {
  D2D_REQUIRE(ctx && pl, "d2d_fit_group_report: null argument");
  if (pl->gsweeps_R != R || R < 1) {
    d2d_set_error("d2d_fit_group_report: the last d2d_fit_solve_groups of this plan was not a persistent-kernel solve over R=%d scenarios", R);
    return D2D_ESTATE;
  }
  if (sweeps) D2D_CHECK_HIP(hipMemcpyAsync(sweeps, pl->d_order + R, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (moved) D2D_CHECK_HIP(hipMemcpyAsync(moved, pl->d_lm, (size_t)R * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return D2D_OK;
}

int d2d_fit_plan_set_groups(d2d_fit_plan *pl, int n_ac) {
  D2D_REQUIRE(pl != nullptr, "d2d_fit_plan_set_groups: plan is NULL");
  D2D_REQUIRE(n_ac >= 1 && n_ac <= 8, "d2d_fit_plan_set_groups: n_ac=%d not in 1..8", n_ac);
  if (n_ac == 1) {                      // back to independent trajectories (any K)
    pl->n_group = 1; pl->nds = 0;
    bool g32;
    int we;
    if (pl->split_ok && pick_eval_layout(pl->K, pl->nq, &g32, &we, 0)) { pl->g32_lds = g32; pl->wpb_eval = we; }
    return D2D_OK;
  }
  const int nds = n_ac > 1 ? 4 * ((n_ac - 1 + 3) / 4) : 0;
  if (!pl->split_ok) {                  // long horizons: the visits of d2d_fit_solve_groups run on the chunked persistent kernel (any K)
    D2D_REQUIRE(pl->use_long, "d2d_fit_plan_set_groups: K=%d fits neither the coupled-group kernels nor the long-horizon kernel", pl->K);
    pl->n_group = n_ac; pl->nds = nds;
    return D2D_OK;
  }
  bool g32;
  int we, ws;
  const int N = 16 * ((2 * pl->nq + 15) / 16);
  if (!pick_eval_layout(pl->K, pl->nq, &g32, &we, nds) || !pick_step_layout(pl->K, pl->nq, N, &ws)) {
    d2d_set_error("d2d_fit_plan_set_groups: K=%d with %d coupled aircraft does not fit the LDS", pl->K, n_ac);
    return D2D_EINVAL;
  }
  pl->n_group = n_ac; pl->nds = nds; pl->g32_lds = g32; pl->wpb_eval = we; pl->wpb_step = ws;
  return D2D_OK;
}

int d2d_fit_solve_groups(d2d_ctx *ctx, d2d_fit_plan *pl, int R, const double *scen, double *q,
                         const d2d_fit_opts *opts, int max_sweeps, int inner_iters, double tol,
                         double *cost, int32_t *sweeps_done, double *stats) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_solve_groups: null argument");
  D2D_REQUIRE(R >= 1 && max_sweeps >= 1 && inner_iters >= 1, "d2d_fit_solve_groups: R, max_sweeps, inner_iters must be >= 1");
  D2D_REQUIRE(pl->n_group >= 2, "d2d_fit_solve_groups: call d2d_fit_plan_set_groups(n_ac >= 2) first");
  const int n_ac = pl->n_group, B = R * n_ac, n = 2 * pl->nq;
  d2d_fit_opts o = opts_or_default(opts);
  o.max_iter = inner_iters;
  if (int rc = ensure_scratch(pl, B)) return rc;
  if (int rc = launch_prep(ctx, pl, B, scen)) return rc;
  pl->prep_valid_for = nullptr;
  pl->active_B = 0;
  pl->gsweeps_R = 0;           // only the persistent-kernel path below leaves a report; the launch-pair paths overwrite d_lm
  // ---- one persistent launch: a wavefront per scenario runs the whole block Gauss-Seidel of its group (fit_groups_kernel)
  int wpb_g = 0;
  if (pl->nq == 24 && pl->kernel_req != D2D_FIT_KERNEL_SPLIT && pick_fused_layout(pl->K, pl->nq, 48, &wpb_g)) {
    const FusedLds L = fused_lds_layout(pl->K, pl->nq, 48, wpb_g);
    const FitGeom gm = geom_of(pl);
    int32_t *queue = ctx->counter_dev + 8;
    D2D_CHECK_HIP(hipMemsetAsync(queue, 0, 2 * sizeof(int32_t), ctx->stream));
    // d_order [B >= 2R]: [0, R) the scenario hand-out order (if set), [R, 2R) the sweeps of this solve; a trajectory hint is dropped
    int32_t *d_sweeps = pl->d_order + R;
    pl->order_B = 0;
    const int32_t *gorder = (pl->gorder_R == R) ? pl->d_order : nullptr;
    double *d_moved = pl->d_lm;                           // [R] scratch
    const int blocks = R < pl->n_cu ? R : pl->n_cu;
    // Line search on the joint cost along slow sweeps (fit_groups_kernel): on from sweep 8 for sweeps that move >= 0.8 x the one
    // before; d2d_fit_opts.gs_ls = 0 switches it off (plain block Gauss-Seidel), gs_ls_s0 / gs_ls_r0 move the thresholds.  Its per-trajectory scratch (the unknowns before the sweep, the trial point) lives in d_H, unused on this path.
    // Measured on configs[2] (tools/dev_groups_accel.py, 8192 scenarios at tol 1e-6): the slowest scenario 200+ -> 39 sweeps, 19
    // scenarios beyond 40 sweeps -> 0, 0.15 evaluations of F per scenario on average, the same fixed point as the plain sweeps
    // in every scenario (F within 1e-12).  (The Anderson extrapolation of the sweep map tried before -- mean 10.5 -> 8.6 sweeps
    // -- was attracted by repelling fixed points, saddles of F, and is gone: DESIGN.md 5.5b.)
    const int ls_s0 = o.gs_ls > 0 ? (o.gs_ls_s0 < 2 ? 2 : o.gs_ls_s0) : 0;
    const double ls_r0_env = o.gs_ls_r0;
    if (int rc = prof_begin(ctx, pl, 2)) return rc;
    hipLaunchKernelGGL((fit_groups_kernel<3, 24>), dim3(blocks), dim3(64 * wpb_g), L.total, ctx->stream, R, n_ac, pl->nds, gm, L, o,
                       max_sweeps, inner_iters, tol, pl->d_G, pl->d_pk, pl->d_G32, pl->d_W32, pl->d_prep, q, pl->d_pos, pl->d_cost,
                       pl->d_g, pl->d_flags, d_sweeps, d_moved, queue, gorder, reinterpret_cast<double *>(pl->d_H), ls_s0, ls_r0_env,
                       o.gs_prio_at);
    D2D_LAUNCH_CHECK();
    if (int rc = prof_end(ctx, pl)) return rc;
    if (cost) D2D_CHECK_HIP(hipMemcpyAsync(cost, pl->d_cost, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    const dim3 gBs((B + 255) / 256), b1s(256);
    D2D_CHECK_HIP(hipMemsetAsync(ctx->stats_dev, 0, 4 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(fit_stats_kernel, gBs, b1s, 0, ctx->stream, B, n, pl->d_cost, pl->d_g, pl->d_flags, ctx->stats_dev);
    D2D_LAUNCH_CHECK();
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->stats_host, ctx->stats_dev, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> hs(R);
    std::vector<double> hm(R);
    D2D_CHECK_HIP(hipMemcpyAsync(hs.data(), d_sweeps, (size_t)R * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipMemcpyAsync(hm.data(), d_moved, (size_t)R * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    int smax = 0;
    double mmax = 0.0;
    for (int r = 0; r < R; ++r) { if (hs[r] > smax) smax = hs[r]; if (hm[r] > mmax) mmax = hm[r]; }
    if (getenv("D2D_GROUPS_DIAG")) {                       // diagnostics: distribution of the sweeps over the scenarios
      std::vector<int32_t> srt(hs);
      std::sort(srt.begin(), srt.end());
      long tot = 0;
      int n_cap = 0;
      for (int v : srt) { tot += v; n_cap += v >= max_sweeps; }
      fprintf(stderr, "[groups] R=%d sweeps: mean %.1f p50 %d p90 %d p99 %d max %d, at the cap: %d scenarios\n", R, (double)tot / R,
              srt[R / 2], srt[(size_t)R * 9 / 10], srt[(size_t)R * 99 / 100], srt[R - 1], n_cap);
    }
    if (stats) {
      for (int i = 0; i < 4; ++i) stats[i] = ctx->stats_host[i];
      stats[2] = mmax;                                    // the largest last-sweep move of any scenario
    }
    if (sweeps_done) *sweeps_done = smax;
    pl->gsweeps_R = R;
    return D2D_OK;
  }
  // ---- launch-pair path (other plan shapes): every scenario sweeps until the slowest one has settled
  // (plans of the long-horizon kernel -- more than 64 nodes -- run every visit as ONE launch of it: 1.6 .. 3 x faster than the
  // launch pairs at 71 .. 201 nodes, tools/dev_groups_long.py; d2d_fit_opts.gs_pairs = 1 keeps the launch pairs where their LDS image
  // holds K, tests compare the two)
  const bool long_path = !pl->split_ok || (pl->use_long && o.gs_pairs == 0);
  d2d_fit_opts o_long = o;
  o_long.mode = D2D_LM_MODE_FAST; o_long.so_lambda = 0.0; o_long.max_iter = inner_iters; o_long.slice = 0;
  const FitGeom gm = geom_of(pl);
  const dim3 gB((B + 255) / 256), b1(256), gR((R + 255) / 256);
  hipLaunchKernelGGL(fit_state_init_kernel, gB, b1, 0, ctx->stream, B, 0, 1, pl->d_lm, pl->d_flags, (int32_t *)nullptr);
  D2D_LAUNCH_CHECK();
  int sw = 0;
  for (sw = 1; sw <= max_sweeps; ++sw) {
    D2D_CHECK_HIP(hipMemcpyAsync(pl->d_qprev, q, (size_t)B * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    for (int i = 0; i < n_ac; ++i) {
      // freeze the others where they are now
      hipLaunchKernelGGL(fit_pos_kernel, dim3(((long)B * pl->K + 255) / 256), b1, 0, ctx->stream, B, gm, pl->d_G, pl->d_Gp, pl->d_prep, q, pl->d_pos);
      hipLaunchKernelGGL(fit_state_init_kernel, gR, b1, 0, ctx->stream, R, i, n_ac, pl->d_lm, pl->d_flags, (int32_t *)nullptr);
      D2D_LAUNCH_CHECK();
      const GroupArgs ga{pl->d_pos, n_ac, i, n_ac, pl->nds};
      if (long_path) {
        // long horizons (the basis does not fit the launch-pair kernels' LDS image, e.g. 07_multioptyplan's 276- and 401-node
        // scenarios): the visit is ONE launch of the chunked persistent kernel on aircraft i of every scenario, Gauss-Newton rows
        D2D_CHECK_HIP(hipMemsetAsync(ctx->counter_dev + 8, 0, 2 * sizeof(int32_t), ctx->stream));
        if (int rc = launch_lm_long(ctx, pl, R, q, o_long, inner_iters, ga)) return rc;
        continue;
      }
      for (int it = 0; it < inner_iters; ++it) {
        if (int rc = launch_eval(ctx, pl, R, q, pl->d_flags, pl->d_cost, pl->d_g, pl->d_H, ga)) return rc;
        if (int rc = launch_step(ctx, pl, R, q, o, ga)) return rc;
      }
    }
    D2D_CHECK_HIP(hipMemsetAsync(ctx->stats_dev + 8, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(fit_moved_kernel, gB, b1, 0, ctx->stream, B, n, q, pl->d_qprev, ctx->stats_dev + 8);
    D2D_LAUNCH_CHECK();
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->stats_host + 8, ctx->stats_dev + 8, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stats_host[8] <= tol) break;
  }
  if (sw > max_sweeps) sw = max_sweeps;
  // final per-aircraft sub-problem costs / gradients with everybody's final positions
  hipLaunchKernelGGL(fit_pos_kernel, dim3(((long)B * pl->K + 255) / 256), b1, 0, ctx->stream, B, gm, pl->d_G, pl->d_Gp, pl->d_prep, q, pl->d_pos);
  D2D_LAUNCH_CHECK();
  const GroupArgs gall{pl->d_pos, n_ac, 0, 1, pl->nds};
  if (long_path) {       // an evaluation-only pass of the chunked kernel: every trajectory RUNNING with no iterations left
    hipLaunchKernelGGL(fit_state_init_kernel, gB, b1, 0, ctx->stream, B, 0, 1, pl->d_lm, pl->d_flags, ctx->counter_dev + 8);
    D2D_LAUNCH_CHECK();
    if (int rc = launch_lm_long(ctx, pl, B, q, o_long, 0, gall)) return rc;
  } else if (int rc = launch_eval(ctx, pl, B, q, nullptr, pl->d_cost, pl->d_g, nullptr, gall)) return rc;
  if (cost) D2D_CHECK_HIP(hipMemcpyAsync(cost, pl->d_cost, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (stats) {
    D2D_CHECK_HIP(hipMemsetAsync(ctx->stats_dev, 0, 4 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(fit_stats_kernel, gB, b1, 0, ctx->stream, B, n, pl->d_cost, pl->d_g, pl->d_flags, ctx->stats_dev);
    D2D_LAUNCH_CHECK();
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->stats_host, ctx->stats_dev, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (stats) {
    for (int i = 0; i < 4; ++i) stats[i] = ctx->stats_host[i];
    stats[2] = ctx->stats_host[8];       // last sweep's largest relative move instead of the LM status count
  }
  if (sweeps_done) *sweeps_done = sw;
  return D2D_OK;
}

int d2d_fit_coeffs(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *scen, const double *q, double *z) {
  D2D_REQUIRE(ctx && pl && scen && q && z, "d2d_fit_coeffs: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_coeffs: B must be >= 1");
  const long tot = (long)B * 16 * pl->S;
  hipLaunchKernelGGL(fit_coeffs_kernel, dim3((tot + 255) / 256), dim3(256), 0, ctx->stream, B, pl->S, pl->nq,
                     pl->duration, pl->d_Z, pl->d_Zp, scen, q, z);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_fit_sample(d2d_ctx *ctx, const d2d_fit_plan *pl, int B, const double *scen, const double *q, double *Y, double *Xs) {
  D2D_REQUIRE(ctx && pl && scen && q, "d2d_fit_sample: null argument");
  D2D_REQUIRE(B >= 1, "d2d_fit_sample: B must be >= 1");
  const long tot = (long)B * pl->K;
  hipLaunchKernelGGL(fit_sample_kernel, dim3((tot + 255) / 256), dim3(256), 0, ctx->stream, B, geom_of(pl), pl->duration,
                     pl->d_G, pl->d_Gp, scen, q, Y, Xs);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

}  // extern "C"
