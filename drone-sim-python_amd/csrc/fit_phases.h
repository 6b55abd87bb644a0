// Wave-level building blocks of the fit kernels (one wavefront = one trajectory), shared by
// fit_eval_kernel / fit_step_kernel (one LM iteration per launch pair) and fit_lm_kernel (the
// whole LM loop in one persistent launch).
#pragma once
#include "fit_device.h"

// Opaque re-definition of a per-lane value (no instruction is emitted).
#define LAUNDER(v) asm volatile("" : "+v"(v))

// ---- phases 1+2: cost and J^T r ------------------------------------------------------------
// qs: q of this trajectory in LDS; us [K][6] fp64 and cf [K+1][4] f32x4 are the wave's scratch.
// Returns sum r^2; g_lane = (J^T r)[lane] for lane < 2nq.
// cfd [K+1][nds] float2: row coefficients of the collision rows (only with a coupled group).
// phase 1 (lane = sample): flat outputs, rows, u_k = D_k^T r_k -> us, fp32 row coefficients -> cf / cfd.
// qs is the interleaved LDS copy of the unknowns (fit_device.h q_slot).
template <int NQ = 0>
__device__ __forceinline__ double eval_phase1(const FitGeom &g, const double *G64, const double *__restrict__ pkb,
                                              const double *qs, double *us, f32x4 *cf, const ScenP &s, int lane, int dbg,
                                              const GroupCtx &gc = GroupCtx{nullptr, 1, 0, 0, 0}, float2 *cfd = nullptr) {
  double cacc = 0.0;
  // (LAUNDER: inside an iteration loop every address below is loop-invariant; re-defining the lane id
  // keeps the compiler from hoisting dozens of them out of the loop and spilling them)
  LAUNDER(lane);
  const int kbank = bank_argmax<NQ>(g, G64, pkb, qs, s, lane);
  for (int k0 = 0; k0 < g.K; k0 += 64) {
    const int k = k0 + lane;
    if (k < g.K) {
      double Y[6] = {1.0 + k, 2.0, 11.0, 3.0, 0.1, 0.2}, u[6] = {0, 0, 0, 0, 0, 0}, pk[FIT_PK];
      f32x4 coef[4] = {f32x4{1.f, 0.f, 1.f, 0.f}, f32x4{1.f, 0.f, 1.f, 0.f}, f32x4{1.f, 0.f, 1.f, 0.f}, f32x4{1.f, 0.f, 1.f, 0.f}};
#pragma unroll
      for (int c = 0; c < FIT_PK; ++c) pk[c] = pkb[(size_t)c * g.K + k];
      if (!(dbg & 8)) flat_outputs_pk<NQ>(g, G64, qs, pk, k, Y);
      if (!(dbg & 1)) cacc += sample_terms<true>(s, Y, pk[6], pk[7], u, coef, k == kbank);
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

// phase 1 of the fused kernel (K <= 64, no group coupling): the per-sample constants pkr live in registers
// for the whole fit (lane = sample never changes) and the scenario row sp is the wave's LDS copy, read
// where it is used instead of being held in ~44 registers across the flat-output loop.
template <int NQ>
__device__ __forceinline__ double eval_phase1_reg(const FitGeom &g, const double *G64, const double (&pkr)[FIT_PK],
                                                  const double *__restrict__ pkb, const double *sp, const double *qs,
                                                  double *us, f32x4 *cf, float2 *cfp, bool so, int lane) {
  // so (wave-uniform): write the second-order block records (cf [K+1][4] rows of M_vel, cfp [K+1][2] rows of
  // M_pos, fit_device.h sample_terms) instead of the four Gauss-Newton row coefficient sets
  double cacc = 0.0;
  LAUNDER(lane);
  const int k = lane;
  int kbank = -1;
  if (sp[PR_CPHIMAX] > 0.0) kbank = bank_argmax<NQ>(g, G64, pkb, qs, load_scenp(sp), lane);   // (rare mode: extra pass)
  if (k < g.K) {
    double Y[6], u[6] = {0, 0, 0, 0, 0, 0};
    f32x4 coef[4];
    flat_outputs_pk<NQ>(g, G64, qs, pkr, k, Y);
    const ScenP s = load_scenp(sp);
    if (so) {
      float2 pos[2];
      cacc = sample_terms<true>(s, Y, pkr[6], pkr[7], u, coef, k == kbank, pos);
      cfp[k * 2] = pos[0]; cfp[k * 2 + 1] = pos[1];
    } else {
      cacc = sample_terms<true>(s, Y, pkr[6], pkr[7], u, coef, k == kbank);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) us[k * 6 + c] = u[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) cf[k * 4 + r] = coef[r];
  } else if (so && k == g.K) {       // the padded sample of an odd K: zero records
#pragma unroll
    for (int r = 0; r < 4; ++r) cf[k * 4 + r] = f32x4{0.f, 0.f, 0.f, 0.f};
    cfp[k * 2] = float2{0.f, 0.f}; cfp[k * 2 + 1] = float2{0.f, 0.f};
  }
  const double cost = wave_sum(cacc);
  wave_lds_sync();
  return cost;
}

// phase 1 of the coupled-group kernel (fit_groups_kernel; K <= 64): eval_phase1_reg plus the collision rows against the
// partners' frozen positions.  Every collision row depends on (x, y) only, so -- like obstacles 2.. -- they join the two
// contracted position rows through the 2x2 block sum o o^T (partner_sums -> sample_terms' xin): the MFMA pass of a coupled
// aircraft is the same 300 instructions as an uncoupled one (as extra k-steps, 7 partners cost 600 more).  Gauss-Newton rows only.
template <int NQ>
__device__ __forceinline__ double eval_phase1_grp(const FitGeom &g, const double *G64, const double (&pkr)[FIT_PK],
                                                  const double *__restrict__ pkb, const double *sp, const double *qs,
                                                  double *us, f32x4 *cf, const GroupCtx &gc, int lane) {
  double cacc = 0.0;
  LAUNDER(lane);
  const int k = lane;
  int kbank = -1;
  if (sp[PR_CPHIMAX] > 0.0) kbank = bank_argmax<NQ>(g, G64, pkb, qs, load_scenp(sp), lane);
  if (k < g.K) {
    double Y[6], u[6] = {0, 0, 0, 0, 0, 0}, xin[6];
    f32x4 coef[4];
    flat_outputs_pk<NQ>(g, G64, qs, pkr, k, Y);
    const ScenP s = load_scenp(sp);
    partner_sums(s, gc, g.K, k, Y[0], Y[1], xin);
    cacc = sample_terms<true>(s, Y, pkr[6], pkr[7], u, coef, k == kbank, nullptr, xin);
#pragma unroll
    for (int c = 0; c < 6; ++c) us[k * 6 + c] = u[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) cf[k * 4 + r] = coef[r];
  }
  const double cost = wave_sum(cacc);
  wave_lds_sync();
  return cost;
}

// The cost alone of a coupled aircraft at qs against the partners' published positions (the line search of fit_groups_kernel):
// returns sum r^2 over all rows of the aircraft, coll = the part of its collision rows; both are per-lane copies of wave sums.
template <int NQ>
__device__ __forceinline__ double eval_cost_grp(const FitGeom &g, const double *G64, const double (&pkr)[FIT_PK],
                                                const double *__restrict__ pkb, const double *sp, const double *qs,
                                                const GroupCtx &gc, int lane, double &coll) {
  double cacc = 0.0, ccol = 0.0;
  LAUNDER(lane);
  const int k = lane;
  int kbank = -1;
  if (sp[PR_CPHIMAX] > 0.0) kbank = bank_argmax<NQ>(g, G64, pkb, qs, load_scenp(sp), lane);
  if (k < g.K) {
    double Y[6], xin[6];
    flat_outputs_pk<NQ>(g, G64, qs, pkr, k, Y);
    const ScenP s = load_scenp(sp);
    partner_sums(s, gc, g.K, k, Y[0], Y[1], xin);
    cacc = sample_terms<false>(s, Y, pkr[6], pkr[7], nullptr, nullptr, k == kbank, nullptr, xin);
    ccol = xin[0];
  }
  coll = wave_sum(ccol);
  return wave_sum(cacc);
}

// phase 2 (lane = unknown): (J^T r)[lane] = sum_k G_k^T u_k, three independent fp64 accumulation chains.
// NQ > 0: nq is a compile-time constant, so every LDS read of a chunk of five samples carries an
// immediate offset and the whole chunk (15 basis values + 15 u's) is requested before its FMAs run.
template <int NQ>
__device__ __forceinline__ double eval_phase2(const FitGeom &g, const double *G64, const double *us, int lane, int dbg) {
  const int nq = NQ ? NQ : g.nq, gstr = NQ ? NQ + 1 : g.gstr;
  const int n = 2 * nq;
  double g_lane = 0.0;
  LAUNDER(lane);
  if (lane < n && !(dbg & 2)) {
    const int ax = lane >= nq ? 1 : 0, jj = lane - ax * nq;
    const double *g0 = G64 + jj, *g1 = g0 + (size_t)g.K * gstr, *g2 = g1 + (size_t)g.K * gstr;
    const double *uk = us + ax;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int k = 0;
    for (; k + 5 <= g.K; k += 5) {
      double gv[15], uv[15];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        gv[3 * i] = g0[i * gstr]; gv[3 * i + 1] = g1[i * gstr]; gv[3 * i + 2] = g2[i * gstr];
        uv[3 * i] = uk[i * 6]; uv[3 * i + 1] = uk[i * 6 + 2]; uv[3 * i + 2] = uk[i * 6 + 4];
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        a0 = fma(uv[3 * i], gv[3 * i], a0);
        a1 = fma(uv[3 * i + 1], gv[3 * i + 1], a1);
        a2 = fma(uv[3 * i + 2], gv[3 * i + 2], a2);
      }
      g0 += 5 * gstr; g1 += 5 * gstr; g2 += 5 * gstr; uk += 30;
    }
    for (; k < g.K; ++k) {
      a0 = fma(uk[0], g0[0], a0);
      a1 = fma(uk[2], g1[0], a1);
      a2 = fma(uk[4], g2[0], a2);
      g0 += gstr; g1 += gstr; g2 += gstr; uk += 6;
    }
    g_lane = (a0 + a1) + a2;
  }
  return g_lane;
}

template <int NQ = 0>
__device__ __forceinline__ double eval_cost_grad(const FitGeom &g, const double *G64, const double *__restrict__ pkb,
                                                 const double *qs, double *us, f32x4 *cf, const ScenP &s,
                                                 int lane, int dbg, double &g_lane,
                                                 const GroupCtx &gc = GroupCtx{nullptr, 1, 0, 0, 0}, float2 *cfd = nullptr) {
  const double cost = eval_phase1<NQ>(g, G64, pkb, qs, us, cf, s, lane, dbg, gc, cfd);
  g_lane = eval_phase2<NQ>(g, G64, us, lane, dbg);
  return cost;
}

// ---- long horizons (K > 64): the same phases over chunks of 64 samples, basis tables read from global memory --------
// fit_lm_long_kernel keeps lane = sample inside a chunk (sample k = k0 + lane).  The basis does not fit the LDS beside
// the per-wave blocks once K grows (3 K (nq+1) 8 B: 90 kB at K = 151, 300 kB at K = 501), so it stays in HBM / L2:
//   GT  [3][nq][K]   fp64, TRANSPOSED: the flat-output pass reads GT[d][j][k0 + lane] -- 64 consecutive doubles per load
//   G64 [3][K][gstr] fp64, row-major (the split path's table): the J^T r pass reads G64[d][k][lane's unknown]
//   G32 [3][K][nq]   fp32 planes for the MFMA operands (jtj_mfma<.., T_LDS = false>)
// Flat outputs of sample k from the per-sample base pk[0..5] plus G_d[k] . q; qs = interleaved LDS copy of q.
template <int NQ>
__device__ __forceinline__ void flat_outputs_gt(const FitGeom &g, const double *__restrict__ GT, const double *qs,
                                                const double pk[FIT_PK], int k, double Y[6]) {
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
  const int nq = NQ ? NQ : g.nq;
#pragma unroll
  for (int c = 0; c < 6; ++c) Y[c] = pk[c];
  const double *t0 = GT + k, *t1 = t0 + (size_t)nq * g.K, *t2 = t1 + (size_t)nq * g.K;
#pragma unroll 4
  for (int j = 0; j < nq; ++j) {
    const f64x2a qq = *reinterpret_cast<const f64x2a *>(qs + 2 * j);
    const double a0 = t0[(size_t)j * g.K], a1 = t1[(size_t)j * g.K], a2 = t2[(size_t)j * g.K];
    Y[0] = fma(a0, qq.x, Y[0]); Y[1] = fma(a0, qq.y, Y[1]);
    Y[2] = fma(a1, qq.x, Y[2]); Y[3] = fma(a1, qq.y, Y[3]);
    Y[4] = fma(a2, qq.x, Y[4]); Y[5] = fma(a2, qq.y, Y[5]);
  }
}

// CostBank max mode over a long horizon: index of the sample with the largest |phi| (first on ties); -1 in mean mode.
// TL: GT is the row-major table [3][K][gstr] in the LDS (plans whose tables fit beside the per-wave blocks), otherwise the
// transposed table [3][nq][K] in global memory.
template <int NQ, bool TL = false>
__device__ __forceinline__ int long_bank_argmax(const FitGeom &g, const double *GT, const double *__restrict__ pkb,
                                                const double *qs, const ScenP &s, int lane) {
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
      if (TL) flat_outputs_pk<NQ>(g, GT, qs, pk, k, Y); else flat_outputs_gt<NQ>(g, GT, qs, pk, k, Y);
      aw = sample_absw(s, Y);
    }
    const double m = wave_max(aw);
    if (m > best) { best = m; kstar = k0 + (int)__builtin_ctzll(__ballot(aw == m)); }
  }
  return __builtin_amdgcn_readfirstlane(kstar);
}

// Phase 1 of one chunk (samples k0 .. k0+63): rows, u = D^T r -> us [64][6], row records -> cf [65][4] (+ cfp [65][2] in
// second-order mode), chunk-local indices; a lane beyond the horizon writes zero records (the MFMA passes read the
// whole chunk up to its last sample and one padded sample).  Returns the chunk's sum r^2 (wave-uniform).
// WANT_JAC = false: the cost alone (trial points).
// gc (coupled groups, the long-horizon path of d2d_fit_solve_groups): the collision rows against the partners' frozen positions
// join the two contracted position rows through partner_sums -> sample_terms' xin, as in eval_phase1_grp (Gauss-Newton rows only).
template <int NQ, bool WANT_JAC, bool TL = false>
__device__ __forceinline__ double long_phase1(const FitGeom &g, const double *GT, const double *__restrict__ pkb,
                                              const double *sp, const double *qs, double *us, f32x4 *cf, float2 *cfp,
                                              bool so, int kbank, int k0, int lane, const GroupCtx &gc = GroupCtx{nullptr, 1, 0, 0, 0}) {
  LAUNDER(lane);
  const int k = k0 + lane;
  double cacc = 0.0;
  if (k < g.K) {
    double pk[FIT_PK], Y[6], u[6] = {0, 0, 0, 0, 0, 0};
    f32x4 coef[4];
#pragma unroll
    for (int c = 0; c < FIT_PK; ++c) pk[c] = pkb[(size_t)c * g.K + k];
    if (TL) flat_outputs_pk<NQ>(g, GT, qs, pk, k, Y); else flat_outputs_gt<NQ>(g, GT, qs, pk, k, Y);
    const ScenP s = load_scenp(sp);
    double xin[6] = {0, 0, 0, 0, 0, 0};
    const bool grp = gc.pos != nullptr;
    if (grp) partner_sums(s, gc, g.K, k, Y[0], Y[1], xin);
    if (!WANT_JAC) {
      cacc = sample_terms<false>(s, Y, pk[6], pk[7], nullptr, nullptr, k == kbank, nullptr, xin, grp);
    } else {
      if (so) {
        float2 pos[2];
        cacc = sample_terms<true>(s, Y, pk[6], pk[7], u, coef, k == kbank, pos);
        cfp[lane * 2] = pos[0]; cfp[lane * 2 + 1] = pos[1];
      } else {
        cacc = sample_terms<true>(s, Y, pk[6], pk[7], u, coef, k == kbank, nullptr, xin, grp);
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) us[lane * 6 + c] = u[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) cf[lane * 4 + r] = coef[r];
    }
  } else if (WANT_JAC) {
#pragma unroll
    for (int c = 0; c < 6; ++c) us[lane * 6 + c] = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) cf[lane * 4 + r] = f32x4{0.f, 0.f, 0.f, 0.f};
    cfp[lane * 2] = float2{0.f, 0.f}; cfp[lane * 2 + 1] = float2{0.f, 0.f};
  }
  if (WANT_JAC && lane == 0) {             // the padded sample behind the chunk (index 64)
#pragma unroll
    for (int r = 0; r < 4; ++r) cf[64 * 4 + r] = f32x4{0.f, 0.f, 0.f, 0.f};
    cfp[64 * 2] = float2{0.f, 0.f}; cfp[64 * 2 + 1] = float2{0.f, 0.f};
  }
  const double cost = wave_sum(cacc);
  wave_lds_sync();
  return cost;
}

// Phase 2 of one chunk (lane = unknown): sum over the chunk's samples of G_k^T u_k, basis rows from the row-major global table.
template <int NQ>
__device__ __forceinline__ double long_phase2(const FitGeom &g, const double *G64g, const double *us, int k0,
                                              int kn, int lane) {
  const int nq = NQ ? NQ : g.nq, gstr = nq + 1;
  double g_lane = 0.0;
  LAUNDER(lane);
  if (lane < 2 * nq) {
    const int ax = lane >= nq ? 1 : 0, jj = lane - ax * nq;
    const double *g0 = G64g + (size_t)k0 * gstr + jj, *g1 = g0 + (size_t)g.K * gstr, *g2 = g1 + (size_t)g.K * gstr;
    const double *uk = us + ax;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 4
    for (int k = 0; k < kn; ++k) {
      a0 = fma(uk[k * 6], g0[(size_t)k * gstr], a0);
      a1 = fma(uk[k * 6 + 2], g1[(size_t)k * gstr], a1);
      a2 = fma(uk[k * 6 + 4], g2[(size_t)k * gstr], a2);
    }
    g_lane = (a0 + a1) + a2;
  }
  return g_lane;
}

// (accesses whose type differs from the one the bytes were stored with go through may_alias types:
// type-based alias analysis must not reorder them against those stores)
template <typename T>
__device__ __forceinline__ T lds_get(const void *p) {
  typedef T __attribute__((may_alias)) Ta;
  return *reinterpret_cast<const Ta *>(p);
}
template <typename T>
__device__ __forceinline__ void lds_put(void *p, const T &v) {
  typedef T __attribute__((may_alias)) Ta;
  *reinterpret_cast<Ta *>(p) = v;
}

// ---- phase 3: J^T J by v_mfma_f32_16x16x4_f32 ------------------------------------------------
// One MFMA k-step = the four contracted rows (v, phi, and the two position rows: obs0, obs1 or, with more
// obstacles / a position box, the two rows of the Cholesky factor of their 2x2 block) of one sample.  Lane l supplies
// J[row rho = l>>4][col 16c + (l&15)] as A- and as B-operand alike:
//   J = cA * TA[k][j] + cB * TB[k][j],  TA = G1 (v, phi) or G0 (obstacles), TB = G2 (phi only).
// Operands of sample k+1 are fetched before the MFMAs of sample k are issued (the tables carry
// one padded row).  acc: upper triangle of the NB x NB grid of 16x16 tiles.
template <int NB, int NQ, bool T_LDS>
__device__ __forceinline__ void jtj_mfma(const FitGeom &g, const unsigned char *lds_base, int t32_off,
                                         const float *T32g, int cf_off, int lane, int Kmf,
                                         f32x4 (&acc)[NB * (NB + 1) / 2], int cfd_off = 0, int nds = 0,
                                         bool accumulate = false) {
  // accumulate (long horizons, fit_lm_long_kernel): the pass covers one chunk of samples -- the records of samples
  // k0 .. k0+Kmf-1 at cf_off, T32g already advanced by k0 rows -- and adds to acc instead of starting from zero
  // All LDS operands are addressed as lds_base + integer byte offset so that the compiler keeps them
  // in the LDS address space (ds_read with immediate offsets) through the unrolled loop.
  LAUNDER(lane);
  const int rho = lane >> 4, ci = lane & 15;
  const int nq = NQ ? NQ : g.nq;
  const int n = 2 * nq;
  // byte geometry of the fp32 basis planes: element size, row step (one sample), plane step (one derivative order)
  constexpr int TE = 4;
  const int tstep = nq * 4, plane = g.K * tstep;
  bool jok[NB], ayc[NB];
  int oc[NB], oa[NB], ob[NB], jj[NB];     // byte offsets: (cA,cB) pair, TA entry, TB entry
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int col = 16 * c + ci;
    jok[c] = col < n;
    ayc[c] = col >= nq;
    jj[c] = jok[c] ? col - (ayc[c] ? nq : 0) : 0;
    oc[c] = cf_off + rho * 16 + (ayc[c] ? 8 : 0);
    oa[c] = ((rho < 2) ? plane : 0) + TE * jj[c];
    ob[c] = 2 * plane + TE * jj[c];
  }
#define LDS_F(off) lds_get<float>(lds_base + (off))
#define LDS_F2(off) lds_get<float2>(lds_base + (off))
#define T32_AT(off) (T_LDS ? LDS_F(t32_off + (off)) : T32g[(off) >> 2])
  if (!accumulate) {
#pragma unroll
    for (int t = 0; t < NB * (NB + 1) / 2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float2 cc[NB];
  float ta[NB], tb[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    cc[c] = LDS_F2(oc[c]);
    ta[c] = T32_AT(oa[c]); tb[c] = T32_AT(ob[c]);
  }
  // one k-step; KK is the compile-time offset from the running offsets so that every LDS read of
  // the unrolled body carries an immediate offset (no per-sample address arithmetic).  The records of
  // sample k+1 are fetched while the MFMAs of sample k execute.
#define JTJ_KSTEP(KK)                                                                          \
  {                                                                                            \
    float v[NB];                                                                               \
    _Pragma("unroll") for (int c = 0; c < NB; ++c) {                                           \
      const float val = fmaf(cc[c].y, tb[c], cc[c].x * ta[c]);                                 \
      v[c] = (2 * nq == 16 * NB || jok[c]) ? val : 0.f;                                        \
    }                                                                                          \
    _Pragma("unroll") for (int c = 0; c < NB; ++c) {                                           \
      cc[c] = LDS_F2(oc[c] + ((KK) + 1) * 64);                                                 \
      ta[c] = T32_AT(oa[c] + ((KK) + 1) * tstep); tb[c] = T32_AT(ob[c] + ((KK) + 1) * tstep);     \
    }                                                                                          \
    int t = 0;                                                                                 \
    _Pragma("unroll") for (int I = 0; I < NB; ++I)                                             \
      _Pragma("unroll") for (int J = I; J < NB; ++J, ++t)                                      \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[I], v[J], acc[t], 0, 0, 0);            \
    /* issue order: the operand VALU of this sample, then its MFMAs with the next sample's LDS reads in their   \
       shadow (two per MFMA): +4 % on the pass.  VALU in that shadow costs time instead (measured: fp32 MFMA and \
       VALU share the ALUs), and the same treatment of the second-order pass below made no difference. */       \
    __builtin_amdgcn_sched_group_barrier(0x002, 2 * NB, 0);                                    \
    _Pragma("unroll") for (int u = 0; u < NB * (NB + 1) / 2; ++u) {                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                       \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
  int k = 0;
  for (; k + 5 <= Kmf; k += 5) {
    JTJ_KSTEP(0) JTJ_KSTEP(1) JTJ_KSTEP(2) JTJ_KSTEP(3) JTJ_KSTEP(4)
#pragma unroll
    for (int c = 0; c < NB; ++c) { oc[c] += 5 * 64; oa[c] += 5 * tstep; ob[c] += 5 * tstep; }
  }
  for (; k < Kmf; ++k) {
    JTJ_KSTEP(0)
#pragma unroll
    for (int c = 0; c < NB; ++c) { oc[c] += 64; oa[c] += tstep; ob[c] += tstep; }
  }
#undef JTJ_KSTEP
  // collision rows of a coupled group: further k-steps of four rows each, all of the form c * G0[k][j]
  for (int grp = 0; grp * 4 < nds; ++grp) {
    for (int kk = 0; kk < Kmf; ++kk) {
      const float2 c2 = LDS_F2(cfd_off + (kk * nds + grp * 4 + rho) * 8);
      float v[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const float val = (ayc[c] ? c2.y : c2.x) * T32_AT(kk * tstep + TE * jj[c]);
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

// ---- phase 3, second-order mode: H = sum_k G_k^T M_k G_k with the per-sample blocks of sample_terms -------
// A-operand = the plain basis rows G_k (component m of sample k: a -> G1 on the x columns, b -> G1 on the y
// columns, c -> G2 on x, d -> G2 on y; position rows: G0 on x / on y), B-operand = M_k G_k (rows with the
// coefficient records).  One velocity k-step per sample (its four rows) and one position k-step per PAIR of
// samples (two rows each): 1.5 k-steps per sample.  Only the upper block triangle is accumulated; M_k is
// symmetric, so A^T B is (to fp32 rounding).
template <int NB, int NQ, bool T_LDS = true>
__device__ __forceinline__ void jtj_mfma_so(const FitGeom &g, const unsigned char *lds_base, int t32_off, int cf_off,
                                            int cfp_off, int lane, f32x4 (&acc)[NB * (NB + 1) / 2],
                                            const float *T32g = nullptr, int Kmf = -1, bool accumulate = false,
                                            bool tail_mask = false) {
  // T_LDS = false / Kmf / accumulate: one chunk of a long horizon (see jtj_mfma); Kmf samples, records at cf_off / cfp_off,
  // T32g = the fp32 planes in global memory advanced by the chunk's first sample
  // tail_mask (segment formulation, fit_seg.h): the record behind the last sample is not a zero record but the next segment's
  // first sample -- an odd Kmf leaves the second half of its last pair out instead
  if (Kmf < 0) Kmf = g.K;
  LAUNDER(lane);
  const int rho = lane >> 4, ci = lane & 15;
  const int nq = NQ ? NQ : g.nq;
  const int plane = g.K * nq;
  bool ayc[NB];
  int jj[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int col = 16 * c + ci;
    ayc[c] = col >= nq;
    jj[c] = col < 2 * nq ? col - (ayc[c] ? nq : 0) : 0;
  }
 if (!accumulate) {
#pragma unroll
    for (int t = 0; t < NB * (NB + 1) / 2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool row_y = (rho & 1) != 0;          // velocity rows: a, c act on the x columns, b, d on the y columns
  const bool row_2 = rho >= 2;                // velocity rows c, d use G2; position rows 2, 3 belong to the next sample
#define SO_T(pl, kk, j) (T_LDS ? lds_get<float>(lds_base + t32_off + 4 * ((pl) * plane + (kk) * nq + (j))) : T32g[(pl) * plane + (kk) * nq + (j)])
#define SO_MFMAS                                                                              \
  {                                                                                           \
    int t = 0;                                                                                \
    _Pragma("unroll") for (int I = 0; I < NB; ++I)                                            \
      _Pragma("unroll") for (int J = I; J < NB; ++J, ++t)                                     \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(va_[I], vb_[J], acc[t], 0, 0, 0);       \
  }
  for (int k = 0; k < Kmf; k += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {                  // velocity k-steps of samples k and k+1
      const int ks = k + half;
      if (tail_mask && ks >= Kmf) break;
      const f32x4 rec = lds_get<f32x4>(lds_base + cf_off + (ks * 4 + rho) * 16);
      float va_[NB], vb_[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const float g1 = SO_T(1, ks, jj[c]), g2 = SO_T(2, ks, jj[c]);
        const float cx = ayc[c] ? rec.z : rec.x, cy = ayc[c] ? rec.w : rec.y;
        const bool live = 16 * c + ci < 2 * nq;
        vb_[c] = live ? fmaf(cy, g2, cx * g1) : 0.f;
        va_[c] = (live && row_y == ayc[c]) ? (row_2 ? g2 : g1) : 0.f;
      }
      SO_MFMAS
      if (half == 0) {                                      // position k-step of the pair (k, k+1)
        const int kp = k + (row_2 ? 1 : 0);
        const float2 rp = lds_get<float2>(lds_base + cfp_off + (kp * 2 + (rho & 1)) * 8);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          const float g0 = SO_T(0, kp, jj[c]);
          const bool live = 16 * c + ci < 2 * nq && !(tail_mask && kp >= Kmf);
          vb_[c] = live ? (ayc[c] ? rp.y : rp.x) * g0 : 0.f;
          va_[c] = (live && row_y == ayc[c]) ? g0 : 0.f;
        }
        SO_MFMAS
      }
    }
  }
#undef SO_T
#undef SO_MFMAS
}

// ---- J^T J: accumulator tiles -> row-owned registers ------------------------------------------
// Row-major LDS image Hs [N+3][N+4] (row N: right-hand side, then two scratch rows of the solve) of the full symmetric matrix (later overwritten in place by the
// Cholesky factor).  Lane l holds elements (16I + 4(l>>4) + r, 16J + (l&15)) of tile (I,J): stored
// as they are (one address register + immediate offsets, conflict-free) and, for off-diagonal
// tiles, mirrored with one 16-byte store (rows 16J + (l&15), columns 16I + 4(l>>4) .. +3).
// The row stride N+4 floats (an odd number of 16-byte quads) makes the row reads conflict-free.
#define CHOL_LS (N + 4)
template <int N>
__device__ __forceinline__ void tiles_to_image(const f32x4 (&acc)[(N / 16) * (N / 16 + 1) / 2], float *Hs, int lane) {
  constexpr int NBs = N / 16, LS = CHOL_LS;
  LAUNDER(lane);
  float *wb = Hs + 4 * (lane >> 4) * LS + (lane & 15);
  float *mb = Hs + (lane & 15) * LS + 4 * (lane >> 4);
  int t = 0;
#pragma unroll
  for (int I = 0; I < NBs; ++I)
#pragma unroll
    for (int J = I; J < NBs; ++J, ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wb[(16 * I + r) * LS + 16 * J] = acc[t][r];
      if (J > I) lds_put<f32x4>(mb + 16 * J * LS + 16 * I, acc[t]);
    }
}

// The right-hand side -g of the damped system becomes row N of the image: lane N then owns it like any
// other row and the factorisation itself performs the forward substitution (row N of the factor of
// [[A, b], [b^T, .]] is y^T with L y = b).
template <int N>
__device__ __forceinline__ void image_put_rhs(float *Hs, int lane, double gi) {
  LAUNDER(lane);
  if (lane < N) Hs[N * CHOL_LS + lane] = (float)(-gi);
}

// row `lane` of the image (lane N: the right-hand side; lanes > N read it too, their values are not used)
template <int N>
__device__ __forceinline__ void image_row(const float *Hs, int lane, f32x2 (&hrow)[N / 2]) {
  constexpr int LS = CHOL_LS;
  LAUNDER(lane);
  const float *src = Hs + (lane < N ? lane : N) * LS;
#pragma unroll
  for (int m = 0; m < N / 4; ++m) {
    const f32x4 v = lds_get<f32x4>(src + 4 * m);
    hrow[2 * m] = f32x2{v.x, v.y};
    hrow[2 * m + 1] = f32x2{v.z, v.w};
  }
}

// diagonal entry of row `lane` of the image (1 for the lanes that own no row of it)
template <int N>
__device__ __forceinline__ float image_diag(const float *Hs, int lane) {
  LAUNDER(lane);
  const int li = lane < N ? lane : 0;
  return Hs[li * CHOL_LS + li];
}

__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// ---- damped normal-equation solve -----------------------------------------------------------
// hrow = row `lane` of J^T J (pairs of columns); lane N holds the right-hand side b = -g (image_put_rhs).
// Solves (J^T J + lam*diag(max(J^T J_ii, floor))) delta = b in fp32 by a Cholesky factorisation in two levels:
//   * BLOCK COLUMNS of 16 (right-looking, on the matrix cores).  Once the 16 columns c0 .. c0+15 of the factor are in the LDS
//     image Lm, their contribution L[R][c0..] . L[C][c0..]^T to every later block column is accumulated by
//     v_mfma_f32_16x16x4_f32 into 16x16 tiles T[I'][J] (rows of block J, columns of block I').  The operand of a tile is ONE
//     16-byte read per lane of the image -- lane (m, g) takes L[16 J + m][c0 + 4 g .. + 3], the k-index of MFMA q is 4 g + q, the
//     same fragment serves as A and as B operand -- and the matrix core distributes it; in rounds 1-3 (and in the first version
//     of this round) every lane formed these dot products itself from rows of the factor that reached it as BROADCAST reads of
//     the image, 276 x 1 KiB per factorisation, and the launch was bound by exactly that LDS traffic (tools/dev/chol_unit.hip:
//     7.7 us per solve with them, 3.6 us without).  The rows of the right-hand side ride along as a fourth row block (row N).
//   * inside a block column, lane i owns row i (prow[], strictly lower triangular: zeros on and above the diagonal, the
//     reciprocal diagonal in a side row of the image -- no per-lane selects in the substitutions) and the 16 columns are
//     factorised left-looking in PANELS of four: the finished tiles come back through the image (the block column's own, not
//     yet written, entries) as the lane's 16 values D[lane][.], the panel-internal dot products run as packed FMA chains on
//     broadcast reads (24 per block column), the 4x4 diagonal block is factorised across lanes j0 .. j0+3 with v_readlane and
//     v_rsq_f32, and the four new entries of every row are published with ONE 16-byte store and one wave-level sync per panel.
// The damping is added to the pivot itself (pivot_j = (a_jj - sum_k L_jk^2) + damp_j, the damp row travels through the image
// as one broadcast quad per panel), so the rows need no copy with a modified diagonal.  Lane N rides along as row N of the
// augmented matrix, which makes its entries the forward substitution L y = b; the back substitution reads the columns of Lm.
// A non-positive pivot is not clamped: it turns the step into NaN, which the gain-ratio test rejects like any failed step.
// Returns false if a pivot is not positive.
// hd / have_hd: the caller already holds the diagonal entry of its row (read from the image after tiles_to_image).
// FULL: every lane < N is a row of the system (2 nq == N): the rows need no masking.
// MP (the MINPACK mode, mp_trial below): `unit` (wave-uniform) damps with lam * I instead of lam * diag; dxnorm = ||delta||_2;
// isq_mode 1 (always) / 2 (only when | ||delta|| - tr_delta | > 0.1 tr_delta): isq = || L^-1 (delta / ||delta||) ||^2 -- the
// quantity lmpar's Newton correction of the damping needs -- by a forward substitution through the rows of the factor
// (right-looking: z_j = w_j / L_jj, w_i -= L_ij z_j; the lane reads its row back from the image).
#define CHOL_ROWS (N + 4)          // rows of the image: N of the factor, the right-hand side / y, a dummy row, 1 / diagonal, damping
#define CHOL_IMAGE_BYTES(n) (((n) + 4) * ((n) + 4) * 4)

// v on the lanes above lane j (a compile-time number once the panel loops are unrolled), zero on the others: the lane mask is
// ones << (j + 1) in a scalar register pair (one s_lshl_b64), no per-lane compare.  `ones` = ~0 made opaque to the compiler:
// as a literal the 64-bit mask reaches s_mov_b64 as a 32-bit literal, which the hardware zero-extends (lanes 32 .. 63 lost).
__device__ __forceinline__ float lanes_above(float v, int j, unsigned long long ones) {
  const unsigned long long m = j >= 63 ? 0ull : (ones << (j + 1));
  float r;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
  return r;
}

template <class M> struct nd_is_banded { static constexpr bool value = false; };

// Metric (fit_knot.hip: the solve in knot coordinates, where lmder's norm is ||s||_M and the damping is lam M): an object with
//   apply(delta) -> (M delta)[lane]      dxnorm = sqrt(delta^T M delta), isq = || L^-1 (M delta / dxnorm) ||^2
//   prepare(lane); damp(j0), scale -> M[lane][j0 .. j0+3] and lam: scale * damp is added to the lane's matrix row where the panel of columns
//                      j0 .. j0+3 reads it (scale = 0 on the lanes that are not rows of the system: lane N carries the right-hand side)
// The default (int) is the Euclidean norm with the damping on the pivots.
template <int N, bool MP = false, bool FULL = false, class Metric = int>
__device__ __forceinline__ bool damped_solve(const f32x2 (&hrow)[N / 2], double lam, bool act, int lane,
                                             float *Lm, float &dgi, float &delta, unsigned long long *tt = nullptr,
                                             bool unit = false, int isq_mode = 0, double tr_delta = 0.0,
                                             double *dxnorm = nullptr, double *isq = nullptr, float hd = 0.f, bool have_hd = false,
                                             Metric metric = Metric()) {
  constexpr int LS = CHOL_LS, NBK = N / 16;
  // tt (diagnostics, fused kernel with D2D_LM_STAMPS): cycles of setup, block columns [0,N/3), [N/3,2N/3), [2N/3,N), substitution
  unsigned long long tl = 0;
#define DS_STAMP(i) if (tt) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tt[i] += t_ - tl; tl = t_; }
  if (tt) tl = __builtin_amdgcn_s_memtime();
  LAUNDER(lane);
  if constexpr (!__is_same(Metric, int)) metric.prepare(lane);       // (per-solve addressing of the metric's rows, from the re-made lane)
  const bool live = act || lane == N;           // rows of the system + the right-hand side row
  float d = hd;
  if (!have_hd) {
    d = 1.f;
#pragma unroll
    for (int m = 0; m < N / 2; ++m) {
      if (2 * m == lane) d = hrow[m].x;
      if (2 * m + 1 == lane) d = hrow[m].y;
    }
  }
  if (!act) d = 1.f;
  dgi = fmaxf(fabsf(d), (float)D2D_LM_DIAG_FLOOR);      // (|.|: the second-order Hessian may have a negative diagonal)
  // what is added to the pivot of this lane's column; a row that is not part of the system is masked to zero below and gets pivot 1
  const float dadd = act ? ((MP && unit) ? (float)lam : (float)(lam * (double)dgi)) : 1.f;
  float *wrow = Lm + (lane <= N ? lane : N + 1) * LS;  // lanes > N write a dummy row
  float *dinv = Lm + (N + 2) * LS;                     // [N] reciprocal diagonal
  float *damp = Lm + (N + 3) * LS;                     // [N] damping of the pivots
  damp[lane < N ? lane : N] = dadd;
  wave_lds_sync();
  const int m16 = lane & 15, g4 = lane >> 4;           // tile coordinates of this lane (MFMA operand and accumulator layouts)
  // the tiles of the block columns to come: T[I'][J] = rows of block J (J == NBK: the block of the right-hand side, row N),
  // columns of block I'; lane (m, g) holds rows 4 g + r, column m
  f32x4 T[NBK][NBK + 1];
#pragma unroll
  for (int ip = 0; ip < NBK; ++ip)
#pragma unroll
    for (int j = 0; j <= NBK; ++j) T[ip][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define HEL(k) (((k) & 1) ? hrow[(k) >> 1].y : hrow[(k) >> 1].x)
  unsigned long long ones = ~0ull;
  asm volatile("" : "+s"(ones));
  DS_STAMP(0)
#pragma unroll
  for (int I = 0; I < NBK; ++I) {
    if (NBK == 3 && I == 1) { DS_STAMP(1) }
    if (NBK == 3 && I == 2) { DS_STAMP(2) }
    const int c0 = 16 * I;
    f32x2 prow[8];                     // this lane's entries of the block column (pairs of columns)
    f32x4 dv[4];                       // D[lane][c0 .. c0+15]: what the block columns before contribute to this lane's dot products
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) dv[sp] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (I > 0) {
      // the finished tiles of this block column -> the image (its own entries of rows >= c0, not written yet) -> lane = row
#pragma unroll
      for (int J = I; J < NBK; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) Lm[(16 * J + 4 * g4 + r) * LS + c0 + m16] = T[I][J][r];
      Lm[(g4 == 0 ? N : N + 1) * LS + c0 + m16] = T[I][NBK][0];        // row N of the right-hand side's block (the rest: dummy row)
      wave_lds_sync();
      const float *own = Lm + (lane <= N ? lane : N) * LS + c0;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) dv[sp] = lds_get<f32x4>(own + 4 * sp);
    }
    f32x4 ring[3][4];                  // chunk kq (columns c0 + 4 kq .. + 3) of rows j0 .. j0+3
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      const int j0 = c0 + 4 * sp;
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 dq = lds_get<f32x4>(damp + j0);
      f32x4 mq = f32x4{0.f, 0.f, 0.f, 0.f};             // (requested here, with the panel's other reads: used after the dot products)
      if constexpr (!__is_same(Metric, int)) mq = metric.damp(j0);
      // the newest chunk (published by the panel before); the older ones were requested before that panel's diagonal block
      if (sp > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) ring[sp - 1][c] = lds_get<f32x4>(Lm + (j0 + c) * LS + c0 + 4 * (sp - 1));
      }
      f32x2 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = f32x2{0.f, 0.f};
#pragma unroll
      for (int kq = 0; kq < sp; ++kq) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 l = ring[kq][c];
          acc[c] = __builtin_elementwise_fma(prow[2 * kq], f32x2{l.x, l.y}, acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 l = ring[kq][c];
          acc[c] = __builtin_elementwise_fma(prow[2 * kq + 1], f32x2{l.z, l.w}, acc[c]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the older chunks of the next panel (rows j0+4 .., columns c0 .. j0-1): under the diagonal block's dependent chain
      if (sp < 3) {
#pragma unroll
        for (int kq = 0; kq < sp; ++kq)
#pragma unroll
          for (int c = 0; c < 4; ++c) ring[kq][c] = lds_get<f32x4>(Lm + (j0 + 4 + c) * LS + c0 + 4 * kq);
      }
      __builtin_amdgcn_sched_barrier(0);
      float s[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a = HEL(j0 + c);
        if constexpr (!__is_same(Metric, int)) a = fmaf(metric.scale, mq[c], a);
        if (!FULL) a = live ? a : 0.f;
        if (I > 0) a -= dv[sp][c];
        s[c] = sp > 0 ? a - (acc[c].x + acc[c].y) : a;
      }
      // 4x4 diagonal block across lanes j0 .. j0+3; l[c] = L[lane][j0+c] (valid on the lanes below the pivot)
      float l[4], inv[4];
#if defined(ABL_NODIAG)
#pragma unroll
      for (int c = 0; c < 4; ++c) { inv[c] = dq[c] + 1.f; l[c] = s[c] * inv[c]; }
#elif !defined(DS_DIAG_UNIFORM)
      // every broadcast on the dependent chain: readlane, add, rsq, mul, readlane, fma per column
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float piv = lane_value(s[c], j0 + c) + dq[c];
        inv[c] = __builtin_amdgcn_rsqf(piv);
        l[c] = s[c] * inv[c];
#pragma unroll
        for (int c2 = c + 1; c2 < 4; ++c2) s[c2] = fmaf(-l[c], lane_value(l[c], j0 + c2), s[c2]);
      }
#else
      // -DDS_DIAG_UNIFORM (measured in round 5, not the default): the ten entries of the diagonal block broadcast FIRST (independent
      // lane reads), its 4 x 4 factor formed redundantly in every lane from wave-uniform values, the lane's own four entries after
      // it -- the same operations in the same order as the chain above (bit-identical) with a dependent chain of rsq -> mul -> fma
      // per column.  A lone wave per SIMD: 3.87 -> 3.67 us per solve; two waves per SIMD, where the solve is bound by issue
      // slots and this form has 16 more vector instructions per panel: 5.40 -> 5.75 us.
      {
        const float a00 = lane_value(s[0], j0), a10 = lane_value(s[0], j0 + 1), a20 = lane_value(s[0], j0 + 2), a30 = lane_value(s[0], j0 + 3);
        const float a11 = lane_value(s[1], j0 + 1), a21 = lane_value(s[1], j0 + 2), a31 = lane_value(s[1], j0 + 3);
        const float a22 = lane_value(s[2], j0 + 2), a32 = lane_value(s[2], j0 + 3), a33 = lane_value(s[3], j0 + 3);
        inv[0] = __builtin_amdgcn_rsqf(a00 + dq[0]);
        const float L10 = a10 * inv[0], L20 = a20 * inv[0], L30 = a30 * inv[0];
        inv[1] = __builtin_amdgcn_rsqf(fmaf(-L10, L10, a11) + dq[1]);
        const float L21 = fmaf(-L20, L10, a21) * inv[1], L31 = fmaf(-L30, L10, a31) * inv[1];
        inv[2] = __builtin_amdgcn_rsqf(fmaf(-L21, L21, fmaf(-L20, L20, a22)) + dq[2]);
        const float L32 = fmaf(-L31, L21, fmaf(-L30, L20, a32)) * inv[2];
        inv[3] = __builtin_amdgcn_rsqf(fmaf(-L32, L32, fmaf(-L31, L31, fmaf(-L30, L30, a33))) + dq[3]);
        l[0] = s[0] * inv[0];
        l[1] = fmaf(-l[0], L10, s[1]) * inv[1];
        l[2] = fmaf(-l[1], L21, fmaf(-l[0], L20, s[2])) * inv[2];
        l[3] = fmaf(-l[2], L32, fmaf(-l[1], L31, fmaf(-l[0], L30, s[3]))) * inv[3];
      }
#endif
      float lm[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) lm[c] = lanes_above(l[c], j0 + c, ones);      // strictly lower
      prow[2 * sp] = f32x2{lm[0], lm[1]};
      prow[2 * sp + 1] = f32x2{lm[2], lm[3]};
      lds_put<f32x4>(wrow + j0, f32x4{lm[0], lm[1], lm[2], lm[3]});
      lds_put<f32x4>(dinv + j0, f32x4{inv[0], inv[1], inv[2], inv[3]});   // uniform values, one address
      wave_lds_sync();                                                     // columns j0 .. j0+3 are in the image
    }
    // block column I is complete: its contribution to the tiles of the block columns behind it
    if (I + 1 < NBK) {
      f32x4 op[NBK + 1];               // op[J]: L[16 J + m][c0 + 4 g .. + 3]; the rows behind N (block NBK) are not used
      // (a Metric that says `banded`: the matrix has half-bandwidth <= 15, so block column I has no entries in the rows of the
      // block columns beyond I + 1 -- their operands are zero and their tiles are left alone: 16 MFMAs per factorisation instead of 28)
      constexpr bool BANDED = nd_is_banded<Metric>::value;
#pragma unroll
      for (int J = I + 1; J <= NBK; ++J) {
        if (BANDED && J > I + 1 && J < NBK) continue;
        const int rowi = (J < NBK) ? 16 * J + m16 : (N + (m16 & 3));
        op[J] = lds_get<f32x4>(Lm + rowi * LS + c0 + 4 * g4);
      }
#pragma unroll
      for (int ip = I + 1; ip < NBK; ++ip)
#pragma unroll
        for (int J = ip; J <= NBK; ++J)
          if (!(BANDED && (ip > I + 1 || (J > I + 1 && J < NBK))))
#pragma unroll
#ifndef ABL_NOMFMA
          for (int q = 0; q < 4; ++q) T[ip][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[J][q], op[ip][q], T[ip][J], 0, 0, 0);
#else
          for (int q = 0; q < 1; ++q) T[ip][J] += op[J] * op[ip];
#endif
    }
  }
#undef HEL
  DS_STAMP(3)
  // back substitution L^T delta = y through the columns of the image; y = row N of the factor
  LAUNDER(lane);
  const int li = lane < N ? lane : 0;
  const float myinv = dinv[li];                          // 1 / L[lane][lane]
  // every pivot positive <=> every reciprocal root finite and positive (a non-positive pivot leaves NaN or inf in all that follow)
  const bool pos = __all(!(lane < N) || (myinv > 0.f && myinv < 3.0e38f));
  const float *col = Lm + li;
  float dl = col[N * LS] * myinv;                        // running (y_i - sum_{k>i} L[k][i] delta_k) / L[i][i]
#ifndef ABL_NOSUB
#ifndef ABL_SUB_LAZY
  // the lane's column of the factor, in blocks of 16 rows, each block requested one block AHEAD of the dependent chain that uses it
  // (two blocks in flight at the start; 32 registers): left to the scheduler the reads were issued two at a time inside the chain,
  // each pair behind a full LDS wait -- 23 exposed LDS latencies on a chain of 47 readlane + fma steps
  {
    constexpr int NBS = N / 16;
    float cm[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cm[(NBS - 1) & 1][r] = col[(16 * (NBS - 1) + r) * LS];
#pragma unroll
    for (int bk = NBS - 1; bk >= 0; --bk) {
      if (bk > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[(bk - 1) & 1][r] = col[(16 * (bk - 1) + r) * LS];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) cm[bk & 1][r] *= -myinv;
#pragma unroll
      for (int r = 15; r >= 0; --r)
        if (16 * bk + r >= 1) dl = fmaf(cm[bk & 1][r], lane_value(dl, 16 * bk + r), dl);      // L[i][lane] = 0 for i <= lane
    }
  }
#else
#pragma unroll
  for (int i = N - 1; i >= 1; --i) dl = fmaf(-col[i * LS] * myinv, lane_value(dl, i), dl);   // L[i][lane] = 0 for i <= lane
#endif
#endif
  delta = act ? dl : 0.f;
  if (MP) {
    float mdl = delta;                                   // (M delta)[lane]
    if constexpr (!__is_same(Metric, int)) mdl = metric.apply(delta);
    const double dn2 = uniform_d(wave_sum((double)delta * (double)mdl));
    const double dn = sqrt(dn2);
    *dxnorm = dn;
    *isq = 0.0;
    const bool want = isq_mode == 1 || (isq_mode == 2 && fabs(dn - tr_delta) > 0.1 * tr_delta);
    if (want && dn > 0.0 && pos) {
      f32x4 lrow[N / 4];                                 // this lane's row of the factor, zero on and above the diagonal
#pragma unroll
      for (int t = 0; t < N / 4; ++t) lrow[t] = lds_get<f32x4>(Lm + li * LS + 4 * t);
#ifndef ABL_ISQ_LAZY
      // (the row scaled by 1 / L[lane][lane] off the chain: the chain itself is readlane + fma per column, no multiply)
#pragma unroll
      for (int t = 0; t < N / 4; ++t) lrow[t] *= -myinv;
      float wcur = mdl * (float)(1.0 / dn) * myinv;      // z_lane so far: (w_lane - sum_{j < lane} L[lane][j] z_j) / L[lane][lane]
#pragma unroll
      for (int j = 0; j < N - 1; ++j) wcur = fmaf(lrow[j >> 2][j & 3], lane_value(wcur, j), wcur);
      const float z = act ? wcur : 0.f;
#else
      float wcur = mdl * (float)(1.0 / dn);
#pragma unroll
      for (int j = 0; j < N - 1; ++j) {
        const float zj = lane_value(wcur * myinv, j);
        wcur = fmaf(-lrow[j >> 2][j & 3], zj, wcur);
      }
      const float z = act ? wcur * myinv : 0.f;
#endif
      *isq = uniform_d(wave_sum((double)z * (double)z));
    }
  }
  DS_STAMP(4)
#undef DS_STAMP
  return pos;
}

// Outcome of one damped step (oracle/fit.py lm_solve): Nielsen's gain-ratio rule for a full step, the parabola
// rule for a shortened one.  All values are wave-uniform.
struct StepOutcome {
  bool accept;
  int status;
  double lam, nu;
};
// The trial loop of one LM iteration tries the full step first (att = 0) and, if its gain ratio is not positive,
// up to two shortened steps alpha*delta along the same direction: the cost along the step is known at 0 (c, slope -a,
// a = -2 g.delta) and at 1 (ct), the parabola through those has its minimum at a / (2 (ct - c + a)).
struct BtState {
  double a, b, alpha;     // slope, model curvature delta^T H delta (= a - pred), current step fraction
};
__device__ __forceinline__ double bt_first_alpha(double a, double c, double ct) {
  const double den = 2.0 * (ct - c + a);
  const double al = den > 0.0 ? a / den : D2D_LM_BT_MAX;
  return fmin(fmax(al, D2D_LM_BT_MIN), D2D_LM_BT_MAX);
}
// ok: the factorisation succeeded; fin: the full step's trial cost is finite and its predicted reduction positive;
// accept: a step (full or shortened, fraction alpha) was taken to a point of cost ct with model reduction pred_s;
// pred: the full step's predicted reduction; dmax = max |delta| of the FULL step.
__device__ __forceinline__ StepOutcome lm_update(bool ok, bool fin, bool accept, double alpha, double c, double ct, double pred,
                                                 double pred_s, double dmax, double qmax, double lam, double nu,
                                                 const d2d_fit_opts &o) {
  StepOutcome r;
  r.status = D2D_ST_RUNNING;
  r.accept = accept;
  if (accept) {
    if (alpha == 1.0) {
      const double rho = (c - ct) / pred_s;
      const double t = 2.0 * rho - 1.0;
      r.lam = fmax(lam * fmax(1.0 / 3.0, 1.0 - t * t * t), D2D_LM_LAMBDA_MIN);
    } else {
      r.lam = fmin(lam / alpha, D2D_LM_LAMBDA_MAX);       // the step of a damped system shrinks like 1 / lam
    }
    r.nu = 2.0;
    const bool small_x = alpha * dmax <= o.xtol * (qmax + o.xtol);
    const bool small_f = ((c - ct) <= o.ftol * c) && (pred_s <= o.ftol * c);
    if (small_f || small_x) r.status = D2D_ST_CONVERGED;
  } else {
    // a rejected step whose predicted reduction is already below ftol: the iterate sits on the rounding
    // floor of the cost (typical after the quadratic phase of the second-order mode) -- converged
    if (ok && fin && pred <= o.ftol * c) r.status = D2D_ST_CONVERGED;
    if (ok) { r.lam = lam * nu; r.nu = nu * 2.0; }
    else { r.lam = lam * D2D_LM_FAIL_MULT; r.nu = nu; }   // indefinite exact Hessian: no step was tried
    if (r.status == D2D_ST_RUNNING && r.lam > D2D_LM_LAMBDA_MAX) r.status = D2D_ST_STALLED;
  }
  return r;
}

// ---- MINPACK's lmder on the normal equations (oracle/fit.py lmder_solve, lmpar_normal) -------------------------------
// The path scipy.optimize.least_squares(method='lm') follows: a trust region ||p|| <= Delta on the Gauss-Newton model (unit
// scaling: scipy's x_scale = 1 is MINPACK's mode 2), lmpar's safeguarded Newton iteration on the damping `par` until
// | ||p(par)|| - Delta | <= 0.1 Delta, the ratio test of actual against predicted reduction, Delta and par updated by lmder's rules.
// lmder / lmpar take every decision from quantities that are functions of J^T J and J^T f alone, so they are taken here from
// the Cholesky factor of J^T J + par I (damped_solve<N, true>) instead of the QR factors of J.
struct MpState {
  double par, delta;        // damping of the last lmpar, trust-region radius
  double dx_gn, t2_gn;      // cached Gauss-Newton step of the current point: its norm, || L^-1 (p / ||p||) ||^2
  float p_gn;               // ... and the step itself (this lane's entry)
  float *pgn_lds;           // (optional, wave-uniform) [64] floats of the wave's LDS block that hold p_gn instead of a register
  __device__ __forceinline__ void put_pgn(float v) {
    if (pgn_lds) pgn_lds[__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))] = v; else p_gn = v;
  }
  __device__ __forceinline__ float get_pgn() const {
    return pgn_lds ? pgn_lds[__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))] : p_gn;
  }
  int gn_valid, gn_ok;      // the cache holds the step of the current J^T J / it could be factorised
  int first, calm, nfac;    // first trial of the fit (Delta = min(Delta, ||p||)); accepted steps in a row with par = 0 and ratio >= 0.75
  int slow;                 // trials in a row (accepted or not) that changed the cost by no more than D2D_LM_MP_SLOW_TOL of itself
};
#define MP_DWARF 2.2250738585072014e-308
#define MP_EPSMCH 2.220446049250313e-16

// One trial of lmder at the point qi (cost c = ||f||^2, gi = (J^T f)[lane], hdiag = (J^T J)[lane][lane]): lmpar, the trial point,
// the ratio test and the updates of Delta / par.  solve(lam, isq_mode, tr_delta, dl, dxn, t2) -> bool factorises J^T J + lam I
// and returns the step dl = -(J^T J + lam I)^-1 J^T f of this lane, its norm and (see damped_solve) t2; trial(dl) -> the cost at
// qi + dl.  Every decision is wave-uniform.  On acceptance qi and c are updated.  Returns the new status; accepted reports
// whether the trial point was taken; on D2D_ST_RUNNING the caller goes on (re-evaluating J^T f, J^T J after an accepted step).
template <class Solve, class Trial>
__device__ __forceinline__ int mp_trial(MpState &s, const d2d_fit_opts &o, double &c, double &qi, double gi, float hdiag, bool act,
                                        int nfev, Solve solve, Trial trial, bool &accepted) {
  accepted = false;
  const double fnorm = sqrt(c);
  double gl = 0.0;
  if (act && hdiag > 0.f && fnorm > 0.0) gl = fabs(gi) / (sqrt((double)hdiag) * fnorm);
  const double gnorm = uniform_d(wave_max(gl));
  if (gnorm <= o.mp_gtol) return D2D_ST_CONVERGED;
  const double gnrm = sqrt(uniform_d(wave_sum(gi * gi)));
  // ---- lmpar ----
  double par = s.par, parl = 0.0, paru = 0.0, fp = 0.0, pn = 0.0;
  float dl = 0.f;
  int it = 0;
  bool done = false;
  auto gn_post = [&]() -> bool {
    if (s.gn_ok && s.dx_gn - s.delta <= 0.1 * s.delta) { dl = s.get_pgn(); pn = s.dx_gn; par = 0.0; return true; }
    fp = s.gn_ok ? s.dx_gn - s.delta : 1.79e308;
    parl = (s.gn_ok && s.t2_gn > 0.0) ? (fp / s.delta) / s.t2_gn : 0.0;
    paru = gnrm / s.delta;
    if (paru == 0.0) paru = MP_DWARF / fmin(s.delta, 0.1);
    par = fmin(fmax(par, parl), paru);
    if (par == 0.0) par = s.gn_ok ? gnrm / s.dx_gn : 0.0;
    return false;
  };
  if (s.gn_valid) done = gn_post();
  while (!done) {
    const bool gn = !s.gn_valid;
    if (!gn && par == 0.0) par = fmax(MP_DWARF, 0.001 * paru);
    float dls;
    double dxn, t2;
    const bool ok = solve(gn ? 0.0 : par, gn ? 1 : (it + 1 < 10 ? 2 : 0), s.delta, dls, dxn, t2);
    ++s.nfac;
    if (gn) {
      s.gn_ok = ok ? 1 : 0; s.put_pgn(dls); s.dx_gn = uniform_d(dxn); s.t2_gn = uniform_d(t2); s.gn_valid = 1;      // (state that lives across trials: scalar registers)
      done = gn_post();
      continue;
    }
    ++it;
    if (!ok) {                                   // cannot happen in exact arithmetic (par > 0): raise the damping
      parl = fmax(parl, par); par = fmax(2.0 * par, 0.001 * paru);
      if (it >= 10) { dl = 0.f; pn = 0.0; done = true; }
      continue;
    }
    const double temp = fp;
    fp = dxn - s.delta;
    if (fabs(fp) <= 0.1 * s.delta || (parl == 0.0 && fp <= temp && temp < 0.0) || it == 10) { dl = dls; pn = dxn; done = true; continue; }
    const double parc = (fp / s.delta) / t2;
    if (fp > 0.0) parl = fmax(parl, par);
    if (fp < 0.0) paru = fmin(paru, par);
    par = fmax(parl, par + parc);
  }
  // ---- the trial point and lmder's updates ----
  if (s.first) { s.delta = uniform_d(fmin(s.delta, pn)); s.first = 0; }
  // (with the LDS slots: the step sits out the trial evaluation -- the kernel's register peak -- in the wave's block)
  if (s.pgn_lds) s.pgn_lds[64 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))] = dl;
  const double ct = trial(dl);
  if (s.pgn_lds) dl = s.pgn_lds[64 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))];
  const bool ctfin = fabs(ct) <= 1.79e308;
  const double fnorm1 = ctfin ? sqrt(ct) : 1.79e308;
  double actred = -1.0;
  if (0.1 * fnorm1 < fnorm) actred = 1.0 - ct / c;
  const double pg = -uniform_d(wave_sum((double)dl * gi));           // p^T J^T f with p = -dl
  const double jp2 = fmax(pg - par * pn * pn, 0.0);                  // ||J p||^2 = p^T g - par p^T p
  const double t1 = jp2 / c, t2v = par * pn * pn / c;
  const double prered = t1 + t2v / 0.5, dirder = -(t1 + t2v);
  const double ratio = prered != 0.0 ? actred / prered : 0.0;
  if (ratio <= 0.25) {
    double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
    s.delta = uniform_d(temp * fmin(s.delta, pn / 0.1));
    par = par / temp;
  } else if (par == 0.0 || ratio >= 0.75) {
    s.delta = uniform_d(pn / 0.5);
    par = 0.5 * par;
  }
  s.par = uniform_d(par);
  accepted = ratio >= 1e-4;
  if (accepted) {
    qi += (double)dl; c = ct;
    s.gn_valid = 0;
    s.calm = (par == 0.0 && ratio >= 0.75) ? s.calm + 1 : 0;
  }
  s.slow = fabs(actred) <= D2D_LM_MP_SLOW_TOL ? s.slow + 1 : 0;
  const double xnorm = sqrt(uniform_d(wave_sum(qi * qi)));
  int info = 0;
  if (fabs(actred) <= o.mp_ftol && prered <= o.mp_ftol && 0.5 * ratio <= 1.0) info = 1;
  if (s.delta <= o.mp_xtol * xnorm) info = 2;
  if (info == 0) {
    if (fabs(actred) <= MP_EPSMCH && prered <= MP_EPSMCH && 0.5 * ratio <= 1.0) info = 6;
    else if (s.delta <= MP_EPSMCH * xnorm) info = 7;
    else if (gnorm <= MP_EPSMCH) info = 8;
  }
  return info != 0 ? D2D_ST_CONVERGED : D2D_ST_RUNNING;
}
