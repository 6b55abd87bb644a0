// Banded SPD solve of the knot-coordinate fit (fit_knot.hip): A = H_u + lam Mu is block tridiagonal in 8 x 8 blocks (seven knots,
// 56 rows with the eight end conditions as identity rows).  One wavefront, lane e = row e = 8 b + r (knot b, row r); lanes 56 .. 62
// carry the right-hand side as extra rows of the augmented matrix, so the forward substitution is part of the factorisation.
//
// NESTED DISSECTION of the knot chain: level 1 eliminates knots {0, 2, 4, 6} at once, level 2 knots {1, 5}, level 3 knot 3 --
// 24 pivot steps instead of 56, each step the same instructions for every lane:
//   * a lane holds up to two 8-entry panels: P (its entries in the columns of pivot knot pivP) and Q (pivot knot pivQ).  Level 1: an
//     even-knot row has P = its own diagonal block row; an odd-knot row has P / Q = its couplings to the even knots left / right of
//     it; the right-hand side lanes have P = the piece of -g of one pivot knot.
//   * step t: every lane publishes P[t] (one 4-byte LDS write into a double-buffered column), reads the raw column t of its
//     pivot knots back (8 floats each), and does  l = P[t] rs,  P[c] -= (l rs) raw[c]  (c > t)  with rs = rsqrt(raw[t]) -- the
//     pivot rows, the rows below them and the coupled rows of the neighbouring knots all in the same instructions.
//   * between the levels the Schur complement  S -= N N^T  of the eliminated knots' neighbours (N = their l-vectors, written
//     row-major into an operand image) runs on v_mfma_f32_16x16x4_f32: 20 MFMAs after level 1 (24 neighbour rows + the
//     right-hand side row, 32 pivot columns), 4 after level 2.  The tiles come back through an LDS image as row corrections.
// Substitutions: inside a knot (8 consecutive lanes) the 8-step triangular solves broadcast with two DPP moves per step (quad_perm +
// a bank-masked row shift by 4); across knots the coupling terms are 8- or 16-term dot products against LDS images.
// A non-positive pivot is not clamped (NaN / inf follow); `pos` reports it.
#pragma once
#include "fit_device.h"

#define ND_ROWS 56
#define ND_LOP_LS 36             // floats per row of the level-1 operand image (32 pivot columns + pad: 16-byte rows, odd quad count)
#define ND_T_LS 28               // ... of the tile image (25 columns used)
#define ND_L2_LS 20              // ... of the level-2 operand / tile images (16 + pad)
// float offsets inside a wave's solver block
#define ND_COLBUF 0              // [2][64] double-buffered pivot column
#define ND_YBUF 128              // [64] forward-substituted right-hand side, lane = row layout
#define ND_SBUF 192              // [8 pad][56][8 pad] solution / vectors for coupling terms
#define ND_ZBUF 264              // [64]
#define ND_G3 328                // [8] right-hand side piece of knot 3 between the levels
#define ND_LIMG 336              // [56][8] strictly lower rows of the pivot blocks' factors
#define ND_LOP2 784              // [16][ND_L2_LS]
#define ND_LOP 1104              // [26][ND_LOP_LS]   rows 0..23 neighbour rows of level 1 (knots 1, 3, 5), 24 the right-hand side, 25 zero
#define ND_TBUF 2040             // [26][ND_T_LS]     row 25: zero
#define ND_FLOATS (2040 + 26 * ND_T_LS)
#define ND_BYTES (ND_FLOATS * 4)

template <int CTRL, int BANK>
__device__ __forceinline__ float nd_dpp(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, BANK, false));
}
// value of lane 8 (lane >> 3) + T of every 8-lane group, in all lanes of the group
template <int T>
__device__ __forceinline__ float nd_bcast8(float v) {
  constexpr int q = T & 3, QP = q | (q << 2) | (q << 4) | (q << 6);
  const float x = nd_dpp<QP, 0xf>(v, v);                        // every quad: its own lane q
  if (T < 4) return nd_dpp<0x114, 0xA>(x, x);                   // row_shr:4 into the upper quad of every 8-group
  return nd_dpp<0x104, 0x5>(x, x);                              // row_shl:4 into the lower quad
}

__device__ __forceinline__ void nd_read8(const float *p, float (&v)[8]) {
  const f32x4 a = lds_get<f32x4>(p), b = lds_get<f32x4>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void nd_write8(float *p, const float (&v)[8]) {
  lds_put<f32x4>(p, f32x4{v[0], v[1], v[2], v[3]});
  lds_put<f32x4>(p + 4, f32x4{v[4], v[5], v[6], v[7]});
}

// one pivot step of a level (see the head of the file); buf: the column buffer of this step (double-buffered by the caller)
template <int T>
__device__ __forceinline__ void nd_step(float (&P)[8], float (&Q)[8], float *buf, int pivP, int pivQ, int lane, bool pivlane,
                                        float &minp, float &dinv) {
  buf[lane] = P[T];
  wave_lds_sync();
  float rP[8], rQ[8];
  nd_read8(buf + 8 * pivP, rP);
  nd_read8(buf + 8 * pivQ, rQ);
  const float pP = rP[T], pQ = rQ[T];
  if (pivlane) minp = fminf(minp, pP);
  const float rsP = __builtin_amdgcn_rsqf(pP), rsQ = __builtin_amdgcn_rsqf(pQ);
  const float lP = P[T] * rsP, lQ = Q[T] * rsQ;
  const float fP = lP * rsP, fQ = lQ * rsQ;
#pragma unroll
  for (int c = T + 1; c < 8; ++c) { P[c] = fmaf(-fP, rP[c], P[c]); Q[c] = fmaf(-fQ, rQ[c], Q[c]); }
  P[T] = lP; Q[T] = lQ;
  if (pivlane && (lane & 7) == T) dinv = rsP;
}

template <int T>
__device__ __forceinline__ void nd_steps(float (&P)[8], float (&Q)[8], float *colbuf, int pivP, int pivQ, int lane, bool pivlane,
                                         float &minp, float &dinv) {
  nd_step<T>(P, Q, colbuf + 64 * (T & 1), pivP, pivQ, lane, pivlane, minp, dinv);
  if constexpr (T < 7) nd_steps<T + 1>(P, Q, colbuf, pivP, pivQ, lane, pivlane, minp, dinv);
}

// the pivot lanes of a level: strictly lower part of their factor row -> Limg; the right-hand side lanes: their piece of y -> ybuf
__device__ __forceinline__ void nd_store_level(const float (&P)[8], float *nd, int lane, bool pivlane, bool rhslane, int rhsknot) {
  if (pivlane) {
    const int r = lane & 7;
    float L[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) L[c] = c < r ? P[c] : 0.f;
    nd_write8(nd + ND_LIMG + 8 * lane, L);
  }
  if (rhslane) nd_write8(nd + ND_YBUF + 8 * rhsknot, P);
}

// 8-step forward solve  L z = w  inside every knot of `inlevel` lanes (Lr: strictly lower row of the lane's pivot block, dinv = 1 / diagonal)
template <int T>
__device__ __forceinline__ void nd_fwd_steps(const float (&Lr)[8], float dinv, float &w) {
  const float zb = nd_bcast8<T>(w * dinv);
  w = fmaf(-Lr[T], zb, w);
  if constexpr (T < 7) nd_fwd_steps<T + 1>(Lr, dinv, w);
}
// 8-step backward solve  L^T s = z  (Lc[t] = L[t][r]: column r of the pivot block below the diagonal)
template <int T>
__device__ __forceinline__ void nd_bwd_steps(const float (&Lc)[8], float dinv, float &z) {
  const float sb = nd_bcast8<T>(z * dinv);
  z = fmaf(-Lc[T], sb, z);
  if constexpr (T > 0) nd_bwd_steps<T - 1>(Lc, dinv, z);
}

struct NdLane {                  // per-lane constants (computed once)
  int b, r;                      // knot and row inside the knot (lanes >= 56: b = 7)
  bool row;                      // lane < 56
  bool even, odd, k15, k3;       // knot classes
};
__device__ __forceinline__ NdLane nd_lane(int lane) {
  NdLane l;
  l.b = lane >> 3; l.r = lane & 7; l.row = lane < ND_ROWS;
  l.even = l.row && !(l.b & 1); l.odd = l.row && (l.b & 1); l.k15 = l.row && (l.b == 1 || l.b == 5); l.k3 = l.row && l.b == 3;
  return l;
}

// Factorisation of A (rows: left[8] | own[8] | right[8] = the lane's entries in the columns of knots b-1, b, b+1; identity rows for
// the end conditions) with the right-hand side rhs (lane = row) riding along.  Leaves the factor in the wave's LDS block `nd` and
// dinv (this lane's reciprocal pivot root) / returns pos (every pivot positive).  The rows of A are not modified.
__device__ __forceinline__ bool nd_factor(const float (&left)[8], const float (&own_in)[8], const float (&right)[8], float rhs,
                                          float *nd, int lane, float &dinv) {
  LAUNDER(lane);
  const NdLane L = nd_lane(lane);
  float *colbuf = nd + ND_COLBUF;
  float minp = 1.0f;
  dinv = 1.f;
  float own[8], P[8], Q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) own[c] = own_in[c];
  // the right-hand side in lane = row layout, for the lanes that carry its pieces
  nd[ND_YBUF + lane] = L.row ? rhs : 0.f;
  wave_lds_sync();
  // ---- level 1: pivots = knots 0, 2, 4, 6
  {
    const bool rhsl = lane >= 56 && lane < 60;
    const int pk = rhsl ? 2 * (lane - 56) : L.b;                  // pivot knot of a right-hand side lane / own knot
    const int pivP = L.odd ? L.b - 1 : (L.even || rhsl ? pk : 0), pivQ = L.odd ? L.b + 1 : pivP;
    float g8[8];
    nd_read8(nd + ND_YBUF + 8 * (rhsl ? pk : 0), g8);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      P[c] = L.odd ? left[c] : (L.even ? own[c] : (rhsl ? g8[c] : 0.f));
      Q[c] = L.odd ? right[c] : 0.f;
    }
    wave_lds_sync();                                              // (ybuf is rewritten below)
    nd_steps<0>(P, Q, colbuf, pivP, pivQ, lane, L.even, minp, dinv);
    nd_store_level(P, nd, lane, L.even, rhsl, pk);
    // operand image: rows of knots 1, 3, 5 (their l-vectors at the column blocks of their two pivot knots, zero elsewhere), row 24
    // = the right-hand side pieces y_0 | y_2 | y_4 | y_6
    if (L.odd) {
      float *row = nd + ND_LOP + (8 * (L.b >> 1) + L.r) * ND_LOP_LS;
#pragma unroll
      for (int i = 0; i < 8; ++i) lds_put<f32x4>(row + 4 * i, f32x4{0.f, 0.f, 0.f, 0.f});
      nd_write8(row + 8 * (L.b >> 1), P);
      nd_write8(row + 8 * (L.b >> 1) + 8, Q);
    }
    if (rhsl) nd_write8(nd + ND_LOP + 24 * ND_LOP_LS + 8 * (lane - 56), P);
    wave_lds_sync();
    // T = N N^T on the matrix cores: tiles (0,0) rows/cols of knots 1, 3; (0,1) knots 1, 3 x (knot 5, rhs); (1,1) (knot 5, rhs)^2
    const int m = lane & 15, g = lane >> 4;
    const float *o0 = nd + ND_LOP + m * ND_LOP_LS + 4 * g;
    const float *o1 = nd + ND_LOP + (m <= 8 ? 16 + m : 25) * ND_LOP_LS + 4 * g;      // rows beyond the right-hand side: the zero row
    const f32x4 a00 = lds_get<f32x4>(o0), a01 = lds_get<f32x4>(o0 + 16), a10 = lds_get<f32x4>(o1), a11 = lds_get<f32x4>(o1 + 16);
    f32x4 T00 = f32x4{0.f, 0.f, 0.f, 0.f}, T01 = T00, T11 = T00;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      T00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[q], a00[q], T00, 0, 0, 0);
      T01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[q], a10[q], T01, 0, 0, 0);
      T00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[q], a01[q], T00, 0, 0, 0);
      T01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[q], a11[q], T01, 0, 0, 0);
      T11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11[q], a11[q], T11, 0, 0, 0);
    }
    // lane (m, g) holds T[I][J][4 g + rr][m]: tiles -> row-major image (rows 0..24, columns 0..24)
    float *tb = nd + ND_TBUF;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      tb[(4 * g + rr) * ND_T_LS + m] = T00[rr];
      if (m < 12) tb[(4 * g + rr) * ND_T_LS + 16 + m] = T01[rr];
      if (m < 12 && 4 * g + rr <= 8) tb[(16 + 4 * g + rr) * ND_T_LS + 16 + m] = T11[rr];
    }
    if (m == 8) lds_put<f32x4>(tb + 24 * ND_T_LS + 4 * g, T01);              // row 24 (the right-hand side) x columns of knots 1, 3
    wave_lds_sync();
    // corrections: own block of the odd rows; the fill blocks of knot 3 against knots 1 and 5; the right-hand side pieces
    const bool r60 = lane == 60, r61 = lane == 61, r62 = lane == 62;
    const int rix = 8 * (L.b >> 1) + L.r;
    const float *zero = tb + 25 * ND_T_LS;
    const float *tO = L.odd ? tb + rix * ND_T_LS + 8 * (L.b >> 1) : zero;
    const float *tA = L.k3 ? tb + rix * ND_T_LS : (r60 ? tb + 24 * ND_T_LS : (r61 ? tb + 24 * ND_T_LS + 16 : (r62 ? tb + 24 * ND_T_LS + 8 : zero)));
    const float *tB = L.k3 ? tb + rix * ND_T_LS + 16 : zero;
    const float *gp = nd + ND_YBUF + (r60 ? 8 : (r61 ? 40 : (r62 ? 24 : 0)));
    float cO[8], cA[8], cB[8], gv[8];
    nd_read8(tO, cO); nd_read8(tA, cA); nd_read8(tB, cB); nd_read8(gp, gv);
    const bool has_g = r60 || r61 || r62;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      own[c] -= cO[c];
      const float x = (has_g ? gv[c] : 0.f) - cA[c];
      P[c] = L.k15 ? own[c] : (r62 ? 0.f : x);
      Q[c] = -cB[c];
      gv[c] = x;
    }
    if (r62) nd_write8(nd + ND_G3, gv);
    wave_lds_sync();
  }
  // ---- level 2: pivots = knots 1, 5; neighbour rows = knot 3 (P: fill block against knot 1, Q: against knot 5)
  {
    const bool r60 = lane == 60, r61 = lane == 61;
    const int pivP = L.k15 ? L.b : (r61 ? 5 : 1), pivQ = L.k3 ? 5 : pivP;
    nd_steps<0>(P, Q, colbuf, pivP, pivQ, lane, L.k15, minp, dinv);
    nd_store_level(P, nd, lane, L.k15, r60 || r61, r61 ? 5 : 1);
    float *l2 = nd + ND_LOP2;
    if (L.k3) { nd_write8(l2 + L.r * ND_L2_LS, P); nd_write8(l2 + L.r * ND_L2_LS + 8, Q); }
    if (r60) nd_write8(l2 + 8 * ND_L2_LS, P);
    if (r61) nd_write8(l2 + 8 * ND_L2_LS + 8, P);
    wave_lds_sync();
    const int m = lane & 15, g = lane >> 4;
    const f32x4 a = lds_get<f32x4>(l2 + (m <= 8 ? m : 9) * ND_L2_LS + 4 * g);        // row 9: zero
    f32x4 T = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) T = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], a[q], T, 0, 0, 0);
    float *tb = nd + ND_TBUF;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      if (m < 12 && 4 * g + rr <= 8) tb[(4 * g + rr) * ND_L2_LS + m] = T[rr];
    wave_lds_sync();
    const bool r62 = lane == 62;
    const float *zero = nd + ND_TBUF + 25 * ND_T_LS;
    float cO[8], gv[8];
    nd_read8(L.k3 ? tb + L.r * ND_L2_LS : (r62 ? tb + 8 * ND_L2_LS : zero), cO);
    nd_read8(nd + ND_G3, gv);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      own[c] -= cO[c];
      P[c] = L.k3 ? own[c] : (r62 ? gv[c] - cO[c] : 0.f);
      Q[c] = 0.f;
    }
    wave_lds_sync();
  }
  // ---- level 3: pivot = knot 3
  {
    nd_steps<0>(P, Q, colbuf, 3, 3, lane, L.k3, minp, dinv);
    nd_store_level(P, nd, lane, L.k3, lane == 62, 3);
    wave_lds_sync();
  }
  return __all(minp > 0.f) && __all(!(L.row) || (dinv > 0.f && dinv < 3.0e38f));
}

// Back substitution L^T s = y with the factor and y left by nd_factor: returns s[lane] (0 on the lanes that are not rows)
__device__ __forceinline__ float nd_back(float *nd, int lane, float dinv) {
  LAUNDER(lane);
  const NdLane L = nd_lane(lane);
  float Lc[8];                                         // column r of the lane's pivot block, below the diagonal
#pragma unroll
  for (int t = 0; t < 8; ++t) Lc[t] = nd[ND_LIMG + 8 * (8 * (L.row ? L.b : 0) + t) + L.r];
  float z = L.row ? nd[ND_YBUF + lane] : 0.f;
  float s = 0.f;
  float *sb = nd + ND_SBUF + 8;                        // (8 floats of zero padding on either side)
  // level 3
  { float zt = z; nd_bwd_steps<7>(Lc, dinv, zt); if (L.k3) s = zt * dinv; }
  if (L.k3) sb[lane] = s;
  wave_lds_sync();
  // level 2: knots 1, 5 against knot 3
  {
    const float *col = nd + ND_LOP2 + (L.b == 5 ? 8 : 0) + L.r;
    float s3[8];
    nd_read8(sb + 24, s3);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = fmaf(col[t * ND_L2_LS], s3[t], acc);
    if (L.k15) z -= acc;
    float zt = z; nd_bwd_steps<7>(Lc, dinv, zt); if (L.k15) s = zt * dinv;
  }
  if (L.k15) sb[lane] = s;
  wave_lds_sync();
  // level 1: even knots against their odd neighbours (operand image rows 8 ((b-1) >> 1) + t and 8 (b >> 1) + t, column block b >> 1)
  {
    const int be = L.even ? L.b : 0;
    const float *colL = nd + ND_LOP + (be > 0 ? 8 * ((be - 2) >> 1) : 25) * ND_LOP_LS + 8 * (be >> 1) + L.r;
    const float *colR = nd + ND_LOP + (be < 6 ? 8 * (be >> 1) : 25) * ND_LOP_LS + 8 * (be >> 1) + L.r;
    const int strL = be > 0 ? ND_LOP_LS : 0, strR = be < 6 ? ND_LOP_LS : 0;
    float sL[8], sR[8];
    nd_read8(sb + 8 * (be - 1), sL);                   // (knot -1: the zero padding)
    nd_read8(sb + 8 * (be + 1), sR);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc = fmaf(colL[t * strL], sL[t], acc); acc = fmaf(colR[t * strR], sR[t], acc); }
    if (L.even) z -= acc;
    float zt = z; nd_bwd_steps<7>(Lc, dinv, zt); if (L.even) s = zt * dinv;
  }
  return L.row ? s : 0.f;
}

// || L^-1 w ||^2 for a vector w given lane = row (the quantity lmpar's Newton correction needs), with the factor left by nd_factor
__device__ __forceinline__ double nd_isq(float *nd, int lane, float dinv, float w) {
  LAUNDER(lane);
  const NdLane L = nd_lane(lane);
  float Lr[8];
  nd_read8(nd + ND_LIMG + 8 * (L.row ? lane : 0), Lr);
  float *zb = nd + ND_ZBUF;
  float z = 0.f;
  w = L.row ? w : 0.f;
  // level 1
  { float wt = w; nd_fwd_steps<0>(Lr, dinv, wt); if (L.even) z = wt * dinv; }
  if (L.even) zb[lane] = z;
  wave_lds_sync();
  {
    const float *row = nd + ND_LOP + (8 * ((L.odd ? L.b : 1) >> 1) + L.r) * ND_LOP_LS + 8 * ((L.odd ? L.b : 1) >> 1);
    float nP[8], nQ[8], zP[8], zQ[8];
    nd_read8(row, nP); nd_read8(row + 8, nQ);
    nd_read8(zb + 8 * ((L.odd ? L.b : 1) - 1), zP); nd_read8(zb + 8 * ((L.odd ? L.b : 1) + 1), zQ);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc = fmaf(nP[t], zP[t], acc); acc = fmaf(nQ[t], zQ[t], acc); }
    if (L.odd) w -= acc;
  }
  // level 2
  { float wt = w; nd_fwd_steps<0>(Lr, dinv, wt); if (L.k15) z = wt * dinv; }
  wave_lds_sync();
  if (L.k15) zb[lane] = z;
  wave_lds_sync();
  {
    float nP[8], nQ[8], zP[8], zQ[8];
    nd_read8(nd + ND_LOP2 + L.r * ND_L2_LS, nP); nd_read8(nd + ND_LOP2 + L.r * ND_L2_LS + 8, nQ);
    nd_read8(zb + 8, zP); nd_read8(zb + 40, zQ);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc = fmaf(nP[t], zP[t], acc); acc = fmaf(nQ[t], zQ[t], acc); }
    if (L.k3) w -= acc;
  }
  // level 3
  { float wt = w; nd_fwd_steps<0>(Lr, dinv, wt); if (L.k3) z = wt * dinv; }
  return uniform_d(wave_sum(L.row ? (double)z * (double)z : 0.0));
}

// once per kernel: the rows of the images that must read as zero
__device__ __forceinline__ void nd_init(float *nd, int lane) {
  for (int i = lane; i < ND_FLOATS; i += 64) nd[i] = 0.f;
  wave_lds_sync();
}
