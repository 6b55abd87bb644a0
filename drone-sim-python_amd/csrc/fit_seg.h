// Long horizons without K-sized tables: the segment formulation of the fit's phases (fit_lm_long_kernel<.., SEG = true>).
//
// The basis of the fit is piecewise: sample k lies in one segment s(k) and its basis rows are
//     G_d[k] = Psi_d(x_k) . Zl_s ,      x_k = 2 tau_k / T - 1 in [-1, 1],
// Psi_d = d-th time derivative of the eight Legendre polynomials P_0 .. P_7 on the segment (three-term recurrences, a few
// dozen flops per sample) and Zl_s [8][nq] the map from the reduced unknowns of one axis to the Legendre coefficients of segment s
// (plan constants, 9 kB for every K: fit_basis.cpp fit_basis_segments).  Monomials instead of Legendre polynomials would make the
// same statement, but Z in that basis has entries of 1e3 .. 1e4 and the fp32 products below lose every digit to cancellation;
// in the Legendre basis |Zl| <= 4 and the fp32 Hessian is as accurate as the direct contraction.
// What follows from it, per evaluation of one trajectory:
//   * flat outputs: the 16 Legendre coefficients of each segment once per trial point (lane = (segment, degree), 2 nq FMAs per
//     lane), then 48 FMAs per sample instead of 6 nq = 144, and no table read;
//   * J^T r = sum_s Zl_s^T m_s with the segment moments m_s = sum_{k in s} Psi(x_k)^T u_k: every lane serves samples of ONE
//     segment in every chunk (segment_map), so the moments accumulate in registers over the chunks and are reduced across the
//     lanes of a segment once per evaluation (through the LDS), instead of K x 6 FMAs + K x 6 table reads per unknown;
//   * J^T J = sum_s Zt_s^T B_s Zt_s with the 16x16 segment blocks B_s = sum_{k in s} rows_k^T rows_k over the 2 x 8 Legendre
//     columns: ONE v_mfma_f32_16x16x4_f32 per sample (jtj_mfma<1, 8>: the same pass as the K <= 64 kernel on a basis of eight
//     functions per axis) instead of six, and 36 MFMAs per segment for the projection (segment_project; the operands of both
//     products are the accumulator registers themselves and one set of Zt fragments: no layout change, no LDS round trip).
// LDS per workgroup: 22 kB of plan constants + 13.5 kB per wave, whatever K is: eight waves per CU at every horizon.
#pragma once
#include "fit_phases.h"

// Lanes of a wave -> segments (plan constant): lanes l0[s] .. l0[s+1]-1 serve segment s; its samples k0[s] .. k0[s]+Ks[s]-1 are
// visited l0[s+1]-l0[s] at a time, chunk c taking the next ones.
struct SegMap {
  int S, nchunk;
  int l0[D2D_FIT_MAX_S + 1], k0[D2D_FIT_MAX_S], Ks[D2D_FIT_MAX_S];
};

struct SegLds {   // byte offsets
  int Wt, Zl64, Zl32, wave0, wave_stride, qs, sp, zc, park, big, cf, cfp, psi, total;
};
#define SEG_ROWS 65          // records per chunk: one per lane + one padded row (the MFMA passes fetch one record ahead)
#define SEG_RED_STRIDE 17    // doubles per lane of the moment reduction scratch (odd: conflict-free column reads)

// Legendre polynomials P_0..P_7 at x with first and second derivatives with respect to TIME (x = 2 tau / T - 1: c1 = 2 / T)
__device__ __forceinline__ void legendre_rows(double x, double c1, double P[8], double dP[8], double ddP[8]) {
  double d1[8], d2[8];
  P[0] = 1.0; P[1] = x; d1[0] = 0.0; d1[1] = 1.0; d2[0] = 0.0; d2[1] = 0.0;
#pragma unroll
  for (int n = 1; n < 7; ++n) {
    const double a = (2.0 * n + 1.0) / (n + 1.0), b = (double)n / (n + 1.0);
    P[n + 1] = a * x * P[n] - b * P[n - 1];
    d1[n + 1] = fma(2.0 * n + 1.0, P[n], d1[n - 1]);
    d2[n + 1] = fma(2.0 * n + 1.0, d1[n], d2[n - 1]);
  }
  const double c2 = c1 * c1;
#pragma unroll
  for (int i = 0; i < 8; ++i) { dP[i] = c1 * d1[i]; ddP[i] = c2 * d2[i]; }
}

// Legendre coefficients of the trial point: lane = (segment, degree) computes zc[lane] = (x-axis, y-axis) coefficient
// = the end-condition part zp (constant per fit, this lane's registers) + Zl64[lane] . q.  qs: interleaved LDS copy of q.
template <int NQ>
__device__ __forceinline__ void segment_coefs(int nq_rt, int S, const double *Zl64, const double *qs, const double *park,
                                              double *zc, int lane) {
  // park: [0][lane] / [1][lane] = the end-condition parts zpx, zpy of this lane's coefficient pair (the wave's LDS block)
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
  const int nq = NQ ? NQ : nq_rt;
  LAUNDER(lane);
  if (lane < 8 * S) {
    const double *zr = Zl64 + (size_t)lane * (nq + 1);
    double ax = park[lane], ay = park[64 + lane];
#pragma unroll 8
    for (int j = 0; j < nq; ++j) {
      const f64x2a qq = *reinterpret_cast<const f64x2a *>(qs + 2 * j);
      const double a = zr[j];
      ax = fma(a, qq.x, ax); ay = fma(a, qq.y, ay);
    }
    f64x2a o; o.x = ax; o.y = ay;
    *reinterpret_cast<f64x2a *>(zc + 2 * lane) = o;
  }
  wave_lds_sync();
}

// What a lane needs of the map (fixed for the whole launch): its segment, its place in the segment's lane group, the group's
// size and the segment's samples.  (Selects over the map's entries: the map lives in scalar registers, a per-lane index into it
// would send it through scratch memory.)
struct LaneSeg {
  int sg, r, L, Ks, k0;
};
__device__ __forceinline__ LaneSeg lane_segment(const SegMap &m, int lane) {
  LaneSeg t{0, lane, m.l0[1], m.Ks[0], m.k0[0]};
#pragma unroll
  for (int s = 1; s < D2D_FIT_MAX_S; ++s)
    if (s < m.S && lane >= m.l0[s]) { t.sg = s; t.r = lane - m.l0[s]; t.L = m.l0[s + 1] - m.l0[s]; t.Ks = m.Ks[s]; t.k0 = m.k0[s]; }
  if (lane >= m.l0[m.S < D2D_FIT_MAX_S ? m.S : D2D_FIT_MAX_S]) t.Ks = 0;        // lanes beyond the last group: idle
  return t;
}
// This lane's sample in chunk c (-1: none)
__device__ __forceinline__ int segment_sample(const LaneSeg &t, int c) {
  const int i = c * t.L + t.r;
  return i < t.Ks ? t.k0 + i : -1;
}

// Flat outputs of the lane's sample from the Legendre coefficients of its segment
__device__ __forceinline__ void segment_flat(const double *zc, int sg, const double P[8], const double dP[8], const double ddP[8],
                                             double Y[6]) {
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
#pragma unroll
  for (int c = 0; c < 6; ++c) Y[c] = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f64x2a z = *reinterpret_cast<const f64x2a *>(zc + 2 * (8 * sg + i));
    Y[0] = fma(P[i], z.x, Y[0]); Y[1] = fma(P[i], z.y, Y[1]);
    Y[2] = fma(dP[i], z.x, Y[2]); Y[3] = fma(dP[i], z.y, Y[3]);
    Y[4] = fma(ddP[i], z.x, Y[4]); Y[5] = fma(ddP[i], z.y, Y[5]);
  }
}

// CostBank max mode (rare): index of the sample with the largest |phi|, first on ties; -1 in mean mode
__device__ __forceinline__ int segment_bank_argmax(const SegMap &m, const LaneSeg &t, const double *__restrict__ sx, double c1,
                                                   const double *zc, const ScenP &s) {
  if (!(s.cphimax > 0.0)) return -1;
  double best = -1.0;
  int kstar = 0x7fffffff;
  for (int c = 0; c < m.nchunk; ++c) {
    const int k = segment_sample(t, c);
    double aw = -1.0;
    if (k >= 0) {
      double P[8], dP[8], ddP[8], Y[6];
      legendre_rows(sx[k], c1, P, dP, ddP);
      segment_flat(zc, t.sg, P, dP, ddP, Y);
      aw = sample_absw(s, Y);
    }
    const double mx = wave_max(aw);
    // (the samples of a chunk are not in index order across the lanes: the smallest index among the maxima)
    int kc = (aw == mx && k >= 0) ? k : 0x7fffffff;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) kc = min(kc, __shfl_xor(kc, o));
    if (mx > best || (mx == best && kc < kstar)) { best = mx; kstar = kc; }
  }
  return __builtin_amdgcn_readfirstlane(kstar);
}

// Phase 1 of one chunk: rows of the lane's sample.  WANT_JAC: the row records -> cf [SEG_ROWS][4] (+ cfp in second-order mode),
// the fp32 operand planes psi [3][SEG_ROWS][8] (plane d = d-th derivative of the eight Legendre rows), and the lane's share of
// the segment moments added to mom[16] (mom[2 i + axis] += Psi_i . u over the three derivative orders).  Returns the chunk's
// sum r^2 (wave-uniform).
// The per-sample inputs of a chunk (abscissa, 'tri' waypoint) are requested one chunk ahead (SegIn): three loads per lane whose
// latency would otherwise open every chunk.
struct SegIn {
  double x, wpx, wpy;
};
__device__ __forceinline__ SegIn segment_inputs(const LaneSeg &t, int K, const double *__restrict__ sx,
                                                const double *__restrict__ pkb, int c) {
  const int k = segment_sample(t, c);
  SegIn in{0.0, 0.0, 0.0};
  if (k >= 0) { in.x = sx[k]; in.wpx = pkb[(size_t)6 * K + k]; in.wpy = pkb[(size_t)7 * K + k]; }
  return in;
}
template <bool WANT_JAC>
__device__ __forceinline__ double segment_phase1(const LaneSeg &t, int K, const SegIn &in, double c1,
                                                 const double *sp, const double *zc,
                                                 f32x4 *cf, float2 *cfp, float *psi, double (&mom)[16], bool so, int kbank, int c,
                                                 int lane, const GroupCtx &gc) {
  LAUNDER(lane);
  const int sg = t.sg;
  const int k = segment_sample(t, c);
  double cacc = 0.0;
  if (k >= 0) {
    double P[8], dP[8], ddP[8], Y[6], u[6] = {0, 0, 0, 0, 0, 0};
    f32x4 coef[4];
    legendre_rows(in.x, c1, P, dP, ddP);
    const double wpx = in.wpx, wpy = in.wpy;
    segment_flat(zc, sg, P, dP, ddP, Y);
    const ScenP s = load_scenp(sp);
    double xin[6] = {0, 0, 0, 0, 0, 0};
    const bool grp = gc.pos != nullptr;
    if (grp) partner_sums(s, gc, K, k, Y[0], Y[1], xin);
    if (!WANT_JAC) {
      cacc = sample_terms<false>(s, Y, wpx, wpy, nullptr, nullptr, k == kbank, nullptr, xin, grp);
    } else {
      if (so) {
        float2 pos[2];
        cacc = sample_terms<true>(s, Y, wpx, wpy, u, coef, k == kbank, pos);
        cfp[lane * 2] = pos[0]; cfp[lane * 2 + 1] = pos[1];
      } else {
        cacc = sample_terms<true>(s, Y, wpx, wpy, u, coef, k == kbank, nullptr, xin, grp);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) cf[lane * 4 + q] = coef[q];
#pragma unroll
      for (int i = 0; i < 8; i += 4) {
        lds_put<f32x4>(psi + lane * 8 + i, f32x4{(float)P[i], (float)P[i + 1], (float)P[i + 2], (float)P[i + 3]});
        lds_put<f32x4>(psi + (SEG_ROWS + lane) * 8 + i, f32x4{(float)dP[i], (float)dP[i + 1], (float)dP[i + 2], (float)dP[i + 3]});
        lds_put<f32x4>(psi + (2 * SEG_ROWS + lane) * 8 + i, f32x4{(float)ddP[i], (float)ddP[i + 1], (float)ddP[i + 2], (float)ddP[i + 3]});
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        mom[2 * i] = fma(P[i], u[0], fma(dP[i], u[2], fma(ddP[i], u[4], mom[2 * i])));
        mom[2 * i + 1] = fma(P[i], u[1], fma(dP[i], u[3], fma(ddP[i], u[5], mom[2 * i + 1])));
      }
    }
  }
  const double cost = wave_sum(cacc);
  wave_lds_sync();
  return cost;
}

// J^T r from the lanes' moment shares: the shares cross the LDS (red [64][SEG_RED_STRIDE], the chunk records are dead by now),
// lane (s, i) adds up the lanes of segment s, the 16 S segment moments go to mm [8 S][2] and lane = unknown projects them
// through Zl64.  Returns (J^T r)[lane] (lanes 0 .. nq-1: x axis, nq .. 2nq-1: y axis).
template <int NQ>
__device__ __forceinline__ double segment_gradient(const SegMap &m, int nq_rt, const double *Zl64, const double (&mom)[16],
                                                   double *red, double *mm, int lane) {
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
  const int nq = NQ ? NQ : nq_rt;
  LAUNDER(lane);
#pragma unroll
  for (int i = 0; i < 16; ++i) red[lane * SEG_RED_STRIDE + i] = mom[i];
  wave_lds_sync();
  if (lane < 8 * m.S) {
    const int sg = lane >> 3, i = lane & 7;
    int la = 0, lb = m.l0[1];
#pragma unroll
    for (int s = 1; s < D2D_FIT_MAX_S; ++s)
      if (s == sg) { la = m.l0[s]; lb = m.l0[s + 1]; }
    double ax = 0.0, ay = 0.0;
    for (int l = la; l < lb; ++l) {
      ax += red[l * SEG_RED_STRIDE + 2 * i];
      ay += red[l * SEG_RED_STRIDE + 2 * i + 1];
    }
    f64x2a o; o.x = ax; o.y = ay;
    *reinterpret_cast<f64x2a *>(mm + 2 * lane) = o;
  }
  wave_lds_sync();
  double g = 0.0;
  if (lane < 2 * nq) {
    const int ax = lane >= nq ? 1 : 0, j = lane - ax * nq;
    const double *zcol = Zl64 + j;
    double g0 = 0.0, g1 = 0.0;
    for (int row = 0; row < 8 * m.S; row += 2) {
      g0 = fma(zcol[(size_t)row * (nq + 1)], mm[2 * row + ax], g0);
      g1 = fma(zcol[(size_t)(row + 1) * (nq + 1)], mm[2 * row + 2 + ax], g1);
    }
    g = g0 + g1;
  }
  wave_lds_sync();
  return g;
}

// H += Zt_s^T B_s Zt_s for one segment: bs = the 16x16 block over (axis, degree) in the accumulator layout (lane l, register r:
// element (4 (l >> 4) + r, l & 15)), Zt_s = [[Zl_s, 0], [0, Zl_s]] (16 x 2 nq).  Both products contract over the 16 rows of Zt_s in
// the order kappa = 4 (l >> 4) + r' at k-step r': then the A operand of T = B_s Zt_s is the lane's own register r' of bs (B_s is
// symmetric), the B operand of Zt_s^T T is its own register r' of T, and the Zt_s fragments zf[c][r'] = Zt_s[kappa][16 c + (l & 15)]
// serve both products.
template <int NB, int NQ>
__device__ __forceinline__ void segment_project(int nq_rt, const float *Zl32, int sg, const f32x4 &bs, int lane,
                                                f32x4 (&acc)[NB * (NB + 1) / 2]) {
  const int nq = NQ ? NQ : nq_rt;
  LAUNDER(lane);
  float zf[NB][4];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int col = 16 * c + (lane & 15);
    const int axc = col >= nq ? 1 : 0, j = col - axc * nq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * (lane >> 4) + r;            // (axis, degree) = (row >> 3, row & 7)
      const bool live = col < 2 * nq && (row >> 3) == axc;
      zf[c][r] = live ? Zl32[(size_t)(8 * sg + (row & 7)) * nq + j] : 0.f;
    }
  }
  f32x4 T[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    T[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) T[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[r], zf[c][r], T[c], 0, 0, 0);
  }
  int t = 0;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = I; J < NB; ++J, ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[I][r], T[J][r], acc[t], 0, 0, 0);
}
