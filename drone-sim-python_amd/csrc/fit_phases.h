// Wave-level building blocks of the fit kernels (one wavefront = one trajectory), shared by
// fit_eval_kernel / fit_step_kernel (one LM iteration per launch pair) and fit_lm_kernel (the
// whole LM loop in one persistent launch).
#pragma once
#include "fit_device.h"

// ---- phases 1+2: cost and J^T r ------------------------------------------------------------
// qs: q of this trajectory in LDS; us [K][6] fp64 and cf [K+1][4] f32x4 are the wave's scratch.
// Returns sum r^2; g_lane = (J^T r)[lane] for lane < 2nq.
// cfd [K+1][nds] float2: row coefficients of the collision rows (only with a coupled group).
// phase 1 (lane = sample): flat outputs, rows, u_k = D_k^T r_k -> us, fp32 row coefficients -> cf / cfd.
__device__ __forceinline__ double eval_phase1(const FitGeom &g, const double *G64, const double *__restrict__ pkb,
                                              const double *qs, double *us, f32x4 *cf, const ScenP &s, int lane, int dbg,
                                              const GroupCtx &gc = GroupCtx{nullptr, 1, 0, 0, 0}, float2 *cfd = nullptr) {
  double cacc = 0.0;
  for (int k0 = 0; k0 < g.K; k0 += 64) {
    const int k = k0 + lane;
    if (k < g.K) {
      double Y[6] = {1.0 + k, 2.0, 11.0, 3.0, 0.1, 0.2}, u[6] = {0, 0, 0, 0, 0, 0}, pk[FIT_PK];
      f32x4 coef[4] = {f32x4{1.f, 0.f, 1.f, 0.f}, f32x4{1.f, 0.f, 1.f, 0.f}, f32x4{1.f, 0.f, 1.f, 0.f}, f32x4{1.f, 0.f, 1.f, 0.f}};
#pragma unroll
      for (int c = 0; c < FIT_PK; ++c) pk[c] = pkb[(size_t)c * g.K + k];
      if (!(dbg & 8)) flat_outputs_pk(g, G64, qs, pk, k, Y);
      if (!(dbg & 1)) cacc += sample_terms<true>(s, Y, pk[6], pk[7], u, coef);
      if (gc.nds) cacc += partner_terms<true>(s, gc, g.K, k, Y[0], Y[1], u, cfd + (size_t)k * gc.nds);
#pragma unroll
      for (int c = 0; c < 6; ++c) us[k * 6 + c] = u[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) cf[k * 4 + r] = coef[r];
    }
  }
  const double cost = wave_sum(cacc);
  wave_lds_sync();
  return cost;
}

// phase 2 (lane = unknown): (J^T r)[lane] = sum_k G_k^T u_k, three independent fp64 accumulation chains
__device__ __forceinline__ double eval_phase2(const FitGeom &g, const double *G64, const double *us, int lane, int dbg) {
  const int n = 2 * g.nq;
  double g_lane = 0.0;
  if (lane < n && !(dbg & 2)) {
    const int ax = lane >= g.nq ? 1 : 0, jj = lane - ax * g.nq;
    const double *g0 = G64 + jj, *g1 = g0 + (size_t)g.K * g.gstr, *g2 = g1 + (size_t)g.K * g.gstr;
    const double *uk = us + ax;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 5
    for (int k = 0; k < g.K; ++k) {
      a0 = fma(uk[k * 6], g0[k * g.gstr], a0);
      a1 = fma(uk[k * 6 + 2], g1[k * g.gstr], a1);
      a2 = fma(uk[k * 6 + 4], g2[k * g.gstr], a2);
    }
    g_lane = (a0 + a1) + a2;
  }
  return g_lane;
}

__device__ __forceinline__ double eval_cost_grad(const FitGeom &g, const double *G64, const double *__restrict__ pkb,
                                                 const double *qs, double *us, f32x4 *cf, const ScenP &s,
                                                 int lane, int dbg, double &g_lane,
                                                 const GroupCtx &gc = GroupCtx{nullptr, 1, 0, 0, 0}, float2 *cfd = nullptr) {
  const double cost = eval_phase1(g, G64, pkb, qs, us, cf, s, lane, dbg, gc, cfd);
  g_lane = eval_phase2(g, G64, us, lane, dbg);
  return cost;
}

template <typename T, int ALIGN>
__device__ __forceinline__ T lds_load(const unsigned char *p) {
  T v;
  __builtin_memcpy(&v, __builtin_assume_aligned(p, ALIGN), sizeof(T));
  return v;
}

// ---- phase 3: J^T J by v_mfma_f32_16x16x4_f32 ------------------------------------------------
// One MFMA k-step = the four contracted rows (v, phi, obs0, obs1) of one sample.  Lane l supplies
// J[row rho = l>>4][col 16c + (l&15)] as A- and as B-operand alike:
//   J = cA * TA[k][j] + cB * TB[k][j],  TA = G1 (v, phi) or G0 (obstacles), TB = G2 (phi only).
// Operands of sample k+1 are fetched before the MFMAs of sample k are issued (the tables carry
// one padded row).  acc: upper triangle of the NB x NB grid of 16x16 tiles.
template <int NB, int NQ, bool T_LDS>
__device__ __forceinline__ void jtj_mfma(const FitGeom &g, const unsigned char *lds_base, int t32_off,
                                         const float *T32g, int cf_off, int lane, int Kmf,
                                         f32x4 (&acc)[NB * (NB + 1) / 2], int cfd_off = 0, int nds = 0) {
  // All LDS operands are addressed as lds_base + integer byte offset so that the compiler keeps them
  // in the LDS address space (ds_read with immediate offsets) through the unrolled loop.
  const int rho = lane >> 4, ci = lane & 15;
  const int nq = NQ ? NQ : g.nq;
  const int n = 2 * nq;
  const int plane = g.K * nq;
  bool jok[NB], ayc[NB];
  int oc[NB], oa[NB], ob[NB], jj[NB];     // byte offsets: (cA,cB) pair, TA entry, TB entry
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int col = 16 * c + ci;
    jok[c] = col < n;
    ayc[c] = col >= nq;
    jj[c] = jok[c] ? col - (ayc[c] ? nq : 0) : 0;
    oc[c] = cf_off + rho * 16 + (ayc[c] ? 8 : 0);
    oa[c] = 4 * (((rho < 2) ? plane : 0) + jj[c]);
    ob[c] = 4 * (2 * plane + jj[c]);
  }
  // (loads go through memcpy: the row coefficients are stored as f32x4 by other lanes, and a
  // type-punned load would let type-based alias analysis reorder it against those stores)
#define LDS_F(off) lds_load<float, 4>(lds_base + (off))
#define LDS_F2(off) lds_load<float2, 8>(lds_base + (off))
#define T32_AT(off) (T_LDS ? LDS_F(t32_off + (off)) : T32g[(off) >> 2])
#pragma unroll
  for (int t = 0; t < NB * (NB + 1) / 2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float2 cc[NB];
  float ta[NB], tb[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    cc[c] = LDS_F2(oc[c]);
    ta[c] = T32_AT(oa[c]); tb[c] = T32_AT(ob[c]);
  }
  // one k-step; KK is the compile-time offset from the running offsets so that every LDS read of
  // the unrolled body carries an immediate offset (no per-sample address arithmetic).  Operands of
  // sample k+1 are fetched before the MFMAs of sample k are issued.
#define JTJ_KSTEP(KK)                                                                          \
  {                                                                                            \
    float v[NB];                                                                               \
    _Pragma("unroll") for (int c = 0; c < NB; ++c) {                                           \
      const float val = fmaf(cc[c].y, tb[c], cc[c].x * ta[c]);                                 \
      v[c] = (2 * nq == 16 * NB || jok[c]) ? val : 0.f;                                        \
    }                                                                                          \
    _Pragma("unroll") for (int c = 0; c < NB; ++c) {                                           \
      cc[c] = LDS_F2(oc[c] + ((KK) + 1) * 64);                                                 \
      ta[c] = T32_AT(oa[c] + ((KK) + 1) * nq * 4); tb[c] = T32_AT(ob[c] + ((KK) + 1) * nq * 4); \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    int t = 0;                                                                                 \
    _Pragma("unroll") for (int I = 0; I < NB; ++I)                                             \
      _Pragma("unroll") for (int J = I; J < NB; ++J, ++t)                                      \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[I], v[J], acc[t], 0, 0, 0);            \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
  int k = 0;
  for (; k + 5 <= Kmf; k += 5) {
    JTJ_KSTEP(0) JTJ_KSTEP(1) JTJ_KSTEP(2) JTJ_KSTEP(3) JTJ_KSTEP(4)
#pragma unroll
    for (int c = 0; c < NB; ++c) { oc[c] += 5 * 64; oa[c] += 5 * nq * 4; ob[c] += 5 * nq * 4; }
  }
  for (; k < Kmf; ++k) {
    JTJ_KSTEP(0)
#pragma unroll
    for (int c = 0; c < NB; ++c) { oc[c] += 64; oa[c] += nq * 4; ob[c] += nq * 4; }
  }
#undef JTJ_KSTEP
  // collision rows of a coupled group: further k-steps of four rows each, all of the form c * G0[k][j]
  for (int grp = 0; grp * 4 < nds; ++grp) {
    for (int kk = 0; kk < Kmf; ++kk) {
      const float2 c2 = LDS_F2(cfd_off + (kk * nds + grp * 4 + rho) * 8);
      float v[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const float val = (ayc[c] ? c2.y : c2.x) * T32_AT(4 * (kk * nq + jj[c]));
        v[c] = (2 * nq == 16 * NB || jok[c]) ? val : 0.f;
      }
      int t = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J, ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[I], v[J], acc[t], 0, 0, 0);
    }
  }
#undef LDS_F
#undef LDS_F2
#undef T32_AT
}

// LDS image of the J^T J tiles used between the MFMA and the Cholesky: element (reg, lane=(rho,cc)) of tile t
// sits at t*TILE_T + reg*TILE_S1 + rho*TILE_S2 + cc.  The strides are chosen so that the accumulator
// stores (fixed reg, all lanes) and the row gather (fixed column, 16 rows of a tile) both spread over
// the 32 LDS banks.
#define TILE_S2 17
#define TILE_S1 70
#define TILE_T (4 * TILE_S1)
__device__ __forceinline__ int tile_slot(int reg, int lane) { return reg * TILE_S1 + (lane >> 4) * TILE_S2 + (lane & 15); }

// Row `lane` of the symmetric matrix whose upper block triangle sits in the padded tile image.
template <int N>
__device__ __forceinline__ void gather_row(const float *tiles, int lane, int n, bool act, float (&row)[N]) {
  constexpr int NBs = N / 16;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float v = 0.f;
    if (act && j < n) {
      const int J = j >> 4, I = lane >> 4;
      const bool up = I <= J;
      const int r = up ? lane : j, c = up ? j : lane;
      const int ti = r >> 4, tj = c >> 4;
      const int tile = ti * NBs - ti * (ti - 1) / 2 + (tj - ti);
      const int rr = r & 15;
      v = tiles[tile * TILE_T + (rr & 3) * TILE_S1 + (rr >> 2) * TILE_S2 + (c & 15)];
    }
    row[j] = v;
  }
}

// ---- damped normal-equation solve -----------------------------------------------------------
// hrow = row `lane` of J^T J.  Solves (J^T J + lam*diag(max(J^T J_ii, floor))) delta = -g in fp32:
// left-looking Cholesky with the row owned by each lane in registers and row j broadcast by
// v_readlane (all indices compile-time), forward substitution in registers, back substitution
// through an LDS copy Lm [N][N+1] of the factor.  Returns false if a pivot is not positive.
#define CHOL_LS (N + 4)      // row stride of the LDS factor image: multiple of 4 floats -> aligned ds_read_b128
template <int N>
__device__ __forceinline__ bool damped_solve(const float (&hrow)[N], double lam, double gi, bool act, int lane,
                                             float *Lm, float &dgi, float &delta) {
  constexpr int LS = CHOL_LS;
  float row[N];
  float d = 1.f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    row[j] = hrow[j];
    if (j == lane) d = hrow[j];
  }
  if (!act) d = 1.f;
  dgi = fmaxf(d, (float)D2D_LM_DIAG_FLOOR);
  const float add = (float)(lam * (double)dgi);
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (j == lane) row[j] = act ? (d + add) : 1.f;
  // Left-looking Cholesky: lane i owns row i in registers; finished columns are mirrored into the LDS
  // image Lm [N][LS] so that row j can be broadcast to every lane by a few ds_read_b128.
  bool ok = true;
  float myinv = 1.f;                 // 1 / L[lane][lane]
  const float *Lrow = Lm;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float s0 = row[j], s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j > 0) wave_lds_sync();
#pragma unroll
    for (int k4 = 0; k4 < (j + 3) / 4; ++k4) {
      f32x4 l;       // L[j][4k4 .. 4k4+3], same address in every lane (memcpy: no type-based aliasing assumptions)
      __builtin_memcpy(&l, __builtin_assume_aligned(Lrow + j * LS + 4 * k4, 16), 16);
      if (4 * k4 + 0 < j) s0 = fmaf(-row[4 * k4 + 0], l.x, s0);
      if (4 * k4 + 1 < j) s1 = fmaf(-row[4 * k4 + 1], l.y, s1);
      if (4 * k4 + 2 < j) s2 = fmaf(-row[4 * k4 + 2], l.z, s2);
      if (4 * k4 + 3 < j) s3 = fmaf(-row[4 * k4 + 3], l.w, s3);
    }
    const float sacc = (s0 + s1) + (s2 + s3);
    const float djj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sacc), j));
    ok = ok && (djj > 0.f);
    const float inv = rsqrtf(fmaxf(djj, 1e-30f));
    row[j] = (lane >= j) ? sacc * inv : 0.f;   // L[lane][j]; diagonal = sqrt(djj)
    if (lane == j) myinv = inv;
    if (lane < N) Lm[lane * LS + j] = row[j];
  }
  // forward substitution L y = -g (lane i keeps y_i)
  float y = (float)(-gi);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float yj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y * myinv), j));
    if (lane == j) y = yj;
    else if (lane > j) y = fmaf(-row[j], yj, y);
  }
  // back substitution L^T delta = y through the columns of the LDS image
  wave_lds_sync();
  float dl = y;
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const float di = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl * myinv), i));
    if (lane == i) dl = di;
    else if (lane < i) dl = fmaf(-Lm[i * LS + lane], di, dl);      // L[i][lane]
  }
  delta = act ? dl : 0.f;
  return ok;
}

// register / v_readlane variant of damped_solve (row j broadcast lane by lane)
template <int N>
__device__ __forceinline__ bool damped_solve_rl(const float (&hrow)[N], double lam, double gi, bool act, int lane,
                                             float *Lm, float &dgi, float &delta) {
  constexpr int LS = CHOL_LS;
  float row[N];
  float d = 1.f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    row[j] = hrow[j];
    if (j == lane) d = hrow[j];
  }
  if (!act) d = 1.f;
  dgi = fmaxf(d, (float)D2D_LM_DIAG_FLOOR);
  const float add = (float)(lam * (double)dgi);
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (j == lane) row[j] = act ? (d + add) : 1.f;
  // Left-looking Cholesky: lane i owns row i in registers; finished columns are mirrored into the LDS
  // image Lm [N][LS] so that row j can be broadcast to every lane by a few ds_read_b128.
  bool ok = true;
  float myinv = 1.f;                 // 1 / L[lane][lane]
  const float *Lrow = Lm;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float s0 = row[j], s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int k = 0; k < j; ++k) {
      const float ljk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row[k]), j));
      if ((k & 3) == 0) s0 = fmaf(-row[k], ljk, s0);
      else if ((k & 3) == 1) s1 = fmaf(-row[k], ljk, s1);
      else if ((k & 3) == 2) s2 = fmaf(-row[k], ljk, s2);
      else s3 = fmaf(-row[k], ljk, s3);
    }
    const float sacc = (s0 + s1) + (s2 + s3);
    const float djj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sacc), j));
    ok = ok && (djj > 0.f);
    const float inv = rsqrtf(fmaxf(djj, 1e-30f));
    row[j] = (lane >= j) ? sacc * inv : 0.f;   // L[lane][j]; diagonal = sqrt(djj)
    if (lane == j) myinv = inv;
    if (lane < N) Lm[lane * LS + j] = row[j];
  }
  // forward substitution L y = -g (lane i keeps y_i)
  float y = (float)(-gi);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float yj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y * myinv), j));
    if (lane == j) y = yj;
    else if (lane > j) y = fmaf(-row[j], yj, y);
  }
  // back substitution L^T delta = y through the columns of the LDS image
  wave_lds_sync();
  float dl = y;
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const float di = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl * myinv), i));
    if (lane == i) dl = di;
    else if (lane < i) dl = fmaf(-Lm[i * LS + lane], di, dl);      // L[i][lane]
  }
  delta = act ? dl : 0.f;
  return ok;
}

// Outcome of one damped step (Nielsen gain-ratio rule, oracle/fit.py lm_solve).  All values are
// wave-uniform.
struct StepOutcome {
  bool accept;
  int status;
  double lam, nu, ct;
};
__device__ __forceinline__ StepOutcome judge_step(bool ok, double c, double ct, double pred, double dmax,
                                                  double qmax, double lam, double nu, const d2d_fit_opts &o) {
  StepOutcome r;
  const bool fin = ok && (fabs(ct) <= 1.79e308) && (pred > 0.0);
  const double rho = fin ? (c - ct) / pred : -1.0;
  r.status = D2D_ST_RUNNING;
  r.ct = ct;
  r.accept = rho > 0.0;
  if (r.accept) {
    const double t = 2.0 * rho - 1.0;
    r.lam = fmax(lam * fmax(1.0 / 3.0, 1.0 - t * t * t), D2D_LM_LAMBDA_MIN);
    r.nu = 2.0;
    const bool small_x = dmax <= o.xtol * (qmax + o.xtol);
    const bool small_f = ((c - ct) <= o.ftol * c) && (pred <= o.ftol * c);
    if (small_f || small_x) r.status = D2D_ST_CONVERGED;
  } else {
    r.lam = lam * nu;
    r.nu = nu * 2.0;
    if (r.lam > D2D_LM_LAMBDA_MAX) r.status = D2D_ST_STALLED;
  }
  return r;
}
