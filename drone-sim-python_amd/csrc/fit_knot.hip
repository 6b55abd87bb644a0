// fit_lm_knot_kernel: the whole default solve (MINPACK's lmder + second-order finish) of a 6-segment, K <= 64 fit in KNOT
// coordinates -- oracle/fit_knot.py is the CPU statement.  Same problem and same path as fit_lm_kernel (fit_kernels.hip; q
// coordinates, oracle/fit.py), other coordinates: the unknowns are the Taylor-scaled knot data of the reference's own local
// parameterisation of a C^3 piecewise degree-7 polynomial, CompositeTraj([MinSnapPoly(Y_j, Y_j+1, T)]) (src/d2d/trajectory.py:166-208),
//     u[8 j + 4 a + k] = T^k / k! * Y_a^(k)(t_j),    j = 0..6 knots, a = axis, k = 0..3;  (j in {0, 6}, k < 2) = the end conditions,
// so that a sample touches the 16 entries of the two knots of its segment only:
//   * flat outputs: 48 fp64 FMAs per sample through a Hermite table (144 against the dense basis),
//   * J^T r: a lane (= entry) sums over the <= 18 samples of its knot's two segments (50 in q),
//   * J^T J: ONE v_mfma_f32_16x16x4_f32 per sample into the 16 x 16 block of its segment (six in q): 50 MFMAs per evaluation
//     instead of 300 -- J^T J is block tridiagonal in 8 x 8 blocks (one knot, both axes).
// q = B (u - u0) with B^T B = Mu, the reference metric in knot coordinates (banded), so lmder in q (unit scaling: the path of
// scipy.optimize.least_squares(method='lm')) becomes: (H_u + par Mu) s = -g_u, ||p|| = sqrt(s^T Mu s), lmpar's Newton correction from
// || L^-1 (Mu s / ||s||_Mu) ||^2, ||J^T f|| = sqrt(g_u^T Mu^-1 g_u), ||x|| = ||u - u0||_Mu -- step for step the same iterates
// (oracle/fit_knot.py, tests/test_oracle_knot.py).  The second-order finish damps with lam Mu and takes its max-norm tests in the
// diagonal scaling of Mu.  The factorisation is the dense fp32 Cholesky of fit_phases.h (damped_solve with the metric hook) on the
// 48 free entries.
#include <cstdlib>
#include <vector>

#include "fit_phases.h"
#include "fit_plan.h"

#define KN_WPB_MAX 8
#define KN_LDS_BYTES (160 * 1024)
#define KN_N 48                 // free entries (the dense system)
#define KN_NE 56                // entries with the end conditions
#define KN_WV_KNOT 10           // doubles per knot of a wave's knot vector (80 B: the six segments' reads fall into different banks)
#define KN_US 10                // doubles per sample of the u records [axis][4] (80 B)
#define KN_GTOL_SCALE 0.1        // the finish's gradient test in the metric's diagonal scaling: a tenth of gtol (oracle/fit_knot.py GTOL_SCALE)
#define KN_SEG_MAX 11            // samples of one segment at K <= 64, S = 6
#define KN_IMG_LS (KN_N + 4)    // row stride of the dense image (fit_phases.h CHOL_LS)

namespace {

// Opaque re-definition of a wave-uniform value (no instruction): the per-sample LDS offsets below are invariant across the
// iterations of a fit, and hoisted out of the iteration loop they no longer fit the scalar registers (the first build fetched each
// of them back with v_readlane from a spill VGPR)
#define LAUNDER_S(v) do { (v) = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); } while (0)

struct KnotGeom {
  int K, S, smax;               // samples, segments, samples of the longest segment
  int k0[D2D_FIT_MAX_S + 2];    // first sample of every segment; k0[S] = K
};

struct KnotLds {
  int Hb64, Hb32, Wseg, Md32, Mi32, Mr32, Msc, wave0, wave_stride;
  int wv, sp, sfull, gfull, park, big, cf, cfp;   // inside a wave's block; `big`: the u records, then cf, cfp -- overlaid by the image
  int total;
};

inline int align16(int v) { return (v + 15) & ~15; }

KnotLds knot_lds_layout(int K, int wpb) {
  KnotLds L;
  int o = 0;
  L.Hb64 = o; o = align16(o + K * KN_HB_STRIDE * 8);
  L.Hb32 = o; o = align16(o + K * 32 * 4);
  L.Wseg = o; o = align16(o + D2D_FIT_MAX_S * 4 * 64 * 4);
  L.Md32 = o; o = align16(o + KN_N * KN_N * 4);
  L.Mi32 = o; o = align16(o + KN_NE * 28 * 4);
  L.Mr32 = o; o = align16(o + KN_NE * 12 * 4);
  L.Msc = o; o = align16(o + 64 * 8);
  L.wave0 = o;
  int w = 0;
  L.wv = w; w = align16(w + 7 * KN_WV_KNOT * 8);
  L.sp = w; w = align16(w + FIT_PREP_STRIDE * 8);
  L.sfull = w; w = align16(w + 9 * 8 * 4);            // one padding knot on either side of the 7 x 8 floats
  L.gfull = w; w = align16(w + KN_NE * 8);
  L.park = w; w = align16(w + 3 * 64 * 8);           // per-fit lane constants: the two waypoint coordinates of the lane's sample, u0 of its entry
  L.big = w;
  const int us_bytes = align16(K * KN_US * 8), cf_bytes = (K + 1) * 4 * 16, cfp_bytes = align16((K + 1) * 2 * 8);
  L.cf = w + us_bytes;
  L.cfp = L.cf + cf_bytes;
  int big = us_bytes + cf_bytes + cfp_bytes;
  if (CHOL_IMAGE_BYTES(KN_N) > big) big = CHOL_IMAGE_BYTES(KN_N);
  w = align16(w + big);
  L.wave_stride = w;
  L.total = o + wpb * w;
  return L;
}

struct KnotDev {                 // device tables of the plan
  const double *Hb64, *Bq, *BiT, *Binv, *Minv, *Pu, *msc;
  const float *Hb32, *Wseg, *Md32, *Mrow32, *Mi32;
};

__device__ __forceinline__ void stage(void *dst, const void *src, int bytes) {
  double *d = reinterpret_cast<double *>(dst);
  const double *s = reinterpret_cast<const double *>(src);
  for (int i = threadIdx.x; i < bytes / 8; i += blockDim.x) d[i] = s[i];
}

// dense index (0 .. 47) of the free entries in ascending order <-> full entry (0 .. 55)
__device__ __forceinline__ int kn_entry_of(int i) { return i < 4 ? (i < 2 ? 2 + i : 4 + i) : (i < 44 ? i + 4 : (i < 46 ? i + 6 : i + 8)); }
// ... with lanes 48 .. 55 on the eight end conditions (entries 0, 1, 4, 5, 48, 49, 52, 53)
__device__ __forceinline__ int kn_entry_all(int l) {
  return l < 48 ? kn_entry_of(l) : (l < 56 ? (l < 52 ? 0 : 48) + ((l & 2) ? 4 : 0) + (l & 1) : 0);
}

// ---- phase 1 (lane = sample): flat outputs from the 16 knot values of the sample's segment, rows, records -------------------------
// wv: the wave's knot vector [7][KN_WV_KNOT] (knot j: x (4), y (4)); seg: this lane's segment.  Records: us [K][KN_US] = u_k as
// [axis][d] (d = 0..2: position, velocity, acceleration part of D_k^T r_k), cf / cfp as fit_phases.h eval_phase1_reg.
__device__ __forceinline__ double knot_phase1(int K, const double *Hb64, const double *sp, const double *wv, int seg, const double *park,
                                              double *us, f32x4 *cf, float2 *cfp, bool so, int lane) {
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
  double cacc = 0.0;
  LAUNDER(lane);
  const int k = lane;
  double Y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (k < K) {
    const double *hb = Hb64 + (size_t)k * KN_HB_STRIDE;
    const double *v0 = wv + seg * KN_WV_KNOT;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const double *vk = v0 + (m >> 2) * KN_WV_KNOT + (m & 3);
      const double vx = vk[0], vy = vk[4];
      const f64x2a h01 = *reinterpret_cast<const f64x2a *>(hb + 4 * m);
      const double h2 = hb[4 * m + 2];
      Y[0] = fma(h01.x, vx, Y[0]); Y[1] = fma(h01.x, vy, Y[1]);
      Y[2] = fma(h01.y, vx, Y[2]); Y[3] = fma(h01.y, vy, Y[3]);
      Y[4] = fma(h2, vx, Y[4]); Y[5] = fma(h2, vy, Y[5]);
    }
  }
  int kbank = -1;
  if (sp[PR_CPHIMAX] > 0.0) {                            // CostBank max mode (rare): the one sample whose phi row is kept
    const ScenP s = load_scenp(sp);
    const double aw = k < K ? sample_absw(s, Y) : -1.0;
    const double mx = wave_max(aw);
    kbank = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(__ballot(aw == mx)));
  }
  if (k < K) {
    double u[6] = {0, 0, 0, 0, 0, 0};
    f32x4 coef[4];
    const ScenP s = load_scenp(sp);
    const double wpx = park[k], wpy = park[64 + k];       // the 'tri' waypoint of this sample (parked per fit)
    if (so) {
      float2 pos[2];
      cacc = sample_terms<true>(s, Y, wpx, wpy, u, coef, k == kbank, pos);
      cfp[k * 2] = pos[0]; cfp[k * 2 + 1] = pos[1];
    } else {
      cacc = sample_terms<true>(s, Y, wpx, wpy, u, coef, k == kbank);
    }
    double *uk = us + k * KN_US;
    *reinterpret_cast<f64x2a *>(uk) = f64x2a{u[0], u[2]}; uk[2] = u[4];
    *reinterpret_cast<f64x2a *>(uk + 4) = f64x2a{u[1], u[3]}; uk[6] = u[5];
#pragma unroll
    for (int r = 0; r < 4; ++r) cf[k * 4 + r] = coef[r];
  }
  const double cost = wave_sum(cacc);
  wave_lds_sync();
  return cost;
}

// ---- phase 2 (lane = entry): (J_u^T r)[e] over the samples of the entry's two segments ------------------------------------------
// kb / km / ke: first sample of segment j-1 (or j at the first knot), of segment j, and the end of segment j (j-1 at the last knot);
// a sample before km sees the entry as the END knot of its segment (Hermite function 4 + kd), one after as the START knot (kd).
// Two counted loops over at most KN_SEG_MAX samples each; a lane is masked out of the iterations beyond its own range.
__device__ __forceinline__ double knot_phase2(const KnotGeom &kg, const double *Hb64, const double *us, int kb, int km, int ke,
                                              int a, int kd, bool live, int lane) {
  typedef double __attribute__((ext_vector_type(2), may_alias)) f64x2a;
  LAUNDER(lane);
  LAUNDER(kb);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  int smax = kg.smax;
  LAUNDER_S(smax);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k0 = half ? km : kb, k1 = half ? ke : km;
    const double *hb = Hb64 + (size_t)k0 * KN_HB_STRIDE + 4 * ((half ? 0 : 4) + kd);
    const double *uk = us + k0 * KN_US + 4 * a;
    const int n = live ? k1 - k0 : 0;
    for (int t = 0; t < smax; t += 3) {                   // three samples per iteration: their twelve reads, then their nine FMAs
      f64x2a h01[3], u01[3];
      double h2[3], u2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (t + i < n) {
          h01[i] = *reinterpret_cast<const f64x2a *>(hb + i * KN_HB_STRIDE); u01[i] = *reinterpret_cast<const f64x2a *>(uk + i * KN_US);
          h2[i] = hb[i * KN_HB_STRIDE + 2]; u2[i] = uk[i * KN_US + 2];
        } else {
          h01[i] = f64x2a{0.0, 0.0}; u01[i] = f64x2a{0.0, 0.0}; h2[i] = 0.0; u2[i] = 0.0;
        }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        a0 = fma(h01[i].x, u01[i].x, a0);
        a1 = fma(h01[i].y, u01[i].y, a1);
        a2 = fma(h2[i], u2[i], a2);
      }
      hb += 3 * KN_HB_STRIDE; uk += 3 * KN_US;
    }
  }
  return (a0 + a1) + a2;
}

// ---- phase 3: the six segment blocks B_s = sum_{k in s} rows_k^T rows_k (16 x 16: knot s (x4, y4), knot s+1 (x4, y4)) --------------
// one MFMA k-step = the four contracted rows of one sample (v, phi, two position rows: fit_device.h sample_terms); lane (c, r) builds
// J[row r][col c] = cA * TA + cB * TB from one 8-byte read of the fp32 Hermite table (d1, d2 for the v / phi rows, d0, d1 for the
// position rows whose cB is zero) and one of the row record.  The end conditions' columns are zero in the table.  The waypoint rows'
// constant block (w^2 sum Hb0^T Hb0, per segment) is the accumulators' start value.
template <int SEGMAX>
__device__ __forceinline__ void knot_mfma(const KnotGeom &kg, const unsigned char *lds, int hb32_off, int cf_off, const float *Wseg,
                                          float ww, int lane, f32x4 (&acc)[D2D_FIT_MAX_S]) {
  LAUNDER(lane);
  const int c = lane & 15, r = lane >> 4;
  const int m = 4 * (c >> 3) + (c & 3), a = (c >> 2) & 1;
  const int tab = hb32_off + (4 * m + (r < 2 ? 1 : 0)) * 4;
  const int cfo = cf_off + r * 16 + a * 8;
  // three segments at a time: their operands first -- 2 x 8-byte reads per sample, all in flight together (clamped past the end
  // of a segment) -- then their k-steps round-robin over the three accumulators
#pragma unroll
  for (int s0 = 0; s0 < D2D_FIT_MAX_S; s0 += 3) {
    float v[3][SEGMAX];
    int n[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int s = s0 + j;
      int kb = kg.k0[s];
      n[j] = kg.k0[s + 1] - kg.k0[s];
      LAUNDER_S(kb); LAUNDER_S(n[j]);
#pragma unroll
      for (int i = 0; i < SEGMAX; ++i) {
        const int kk = kb + (i < n[j] ? i : (n[j] > 0 ? n[j] - 1 : 0));
        const float2 t = lds_get<float2>(lds + tab + kk * 128);
        const float2 cc = lds_get<float2>(lds + cfo + kk * 64);
        v[j][i] = fmaf(cc.y, t.y, cc.x * t.x);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[s][rr] = ww * Wseg[(s * 4 + rr) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < SEGMAX; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (i < n[j]) acc[s0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], v[j][i], acc[s0 + j], 0, 0, 0);
  }
}

// second-order mode: B_s = sum_k G_k^T M_k G_k (fit_phases.h jtj_mfma_so): per sample one velocity k-step (A = the plain rows a, b,
// c, d = G1 on x, G1 on y, G2 on x, G2 on y; B = the M_vel-weighted rows from the block records) and one position k-step (rows
// x, y; rows 2, 3 zero)
__device__ __forceinline__ void knot_mfma_so(const KnotGeom &kg, const unsigned char *lds, int hb32_off, int cf_off, int cfp_off,
                                             const float *Wseg, float ww, int lane, f32x4 (&acc)[D2D_FIT_MAX_S]) {
  LAUNDER(lane);
  const int c = lane & 15, rho = lane >> 4;
  const int m = 4 * (c >> 3) + (c & 3);
  const bool ay = ((c >> 2) & 1) != 0, row_y = (rho & 1) != 0, row_2 = rho >= 2;
  const int tab = hb32_off + 4 * m * 4;
#pragma unroll
  for (int s = 0; s < D2D_FIT_MAX_S; ++s) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[s][rr] = ww * Wseg[(s * 4 + rr) * 64 + lane];
    int k1 = kg.k0[s + 1];
    int k = kg.k0[s];
    LAUNDER_S(k1); LAUNDER_S(k);
    if (k < k1) {
      f32x4 hb = lds_get<f32x4>(lds + tab + k * 128);                      // d0, d1, d2, -
      f32x4 rec = lds_get<f32x4>(lds + cf_off + (k * 4 + rho) * 16);
      float2 rp = lds_get<float2>(lds + cfp_off + (k * 2 + (rho & 1)) * 8);
      for (; k < k1; ++k) {
        const int kn = k + 1 < k1 ? k + 1 : k;                             // the next sample's records: requested before this one's k-steps
        const f32x4 hbn = lds_get<f32x4>(lds + tab + kn * 128);
        const f32x4 recn = lds_get<f32x4>(lds + cf_off + (kn * 4 + rho) * 16);
        const float2 rpn = lds_get<float2>(lds + cfp_off + (kn * 2 + (rho & 1)) * 8);
        const float h0 = hb.x, h1 = hb.y, h2 = hb.z;
        const float cx = ay ? rec.z : rec.x, cy = ay ? rec.w : rec.y;
        const float vb = fmaf(cy, h2, cx * h1);
        const float va = (row_y == ay) ? (row_2 ? h2 : h1) : 0.f;
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, vb, acc[s], 0, 0, 0);
        const float vbp = row_2 ? 0.f : (ay ? rp.y : rp.x) * h0;
        const float vap = (!row_2 && row_y == ay) ? h0 : 0.f;
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(vap, vbp, acc[s], 0, 0, 0);
        hb = hbn; rec = recn; rp = rpn;
      }
    }
  }
}

// ---- the segment blocks -> the dense image of J_u^T J_u over the 48 free entries (fit_phases.h tiles_to_image's layout) ---------------
// lane (c, g4) holds B_s[4 g4 + rr][c]: full-entry row 8 s + 4 g4 + rr, column 8 s + c.  Consecutive segments overlap on the 8 x 8
// block of their common knot: the upper-knot quadrant of B_{s-1} (lanes c >= 8, g4 >= 2) is fetched into the lower-knot quadrant of
// B_s (lanes c < 8, g4 < 2: lane + 40) with ds_bpermute and added in registers, so every image entry has ONE writer: plain stores
// after the image is zeroed (LDS atomics on 64 lanes cost several times a store).  ic0 / ic5, rv0 / rv5: dense column of this
// lane and validity of its rows in the first / last segment, where the end conditions' entries are left out (-1 / false).
__device__ __forceinline__ void knot_blocks_to_image(f32x4 (&acc)[D2D_FIT_MAX_S], float *Hs, int lane) {
  LAUNDER(lane);
  constexpr int LS = KN_IMG_LS;
  for (int i = lane; i < KN_N * LS / 4; i += 64) lds_put<f32x4>(Hs + 4 * i, f32x4{0.f, 0.f, 0.f, 0.f});
  const int c = lane & 15, g4 = lane >> 4;
  const bool lowq = c < 8 && g4 < 2, highq = c >= 8 && g4 >= 2;
  const int src = ((lane + 40) & 63) << 2;
  auto fetch = [&](float v) -> float {
    const float up = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v)));
    return lowq ? up : 0.f;
  };
#pragma unroll
  for (int s = D2D_FIT_MAX_S - 1; s >= 1; --s) {
    // (component by component: with a loop over the vector subscript the compiler fetched component 0 for all four)
    const float u0 = fetch(acc[s - 1].x), u1 = fetch(acc[s - 1].y), u2 = fetch(acc[s - 1].z), u3 = fetch(acc[s - 1].w);
    acc[s].x += u0; acc[s].y += u1; acc[s].z += u2; acc[s].w += u3;
  }
  wave_lds_sync();
  // dense index = full entry - 4 for the knots 1 .. 5; knot 0: entries 2, 3, 6, 7 -> 0 .. 3; knot 6: entries 50, 51, 54, 55 -> 44 .. 47
  const int cq = c & 7, cd = 2 * (cq >> 2) + (cq & 3) - 2;            // dense offset of an end knot's column, valid if (cq & 3) >= 2
  const bool cv = (cq & 3) >= 2;
#pragma unroll
  for (int s = 0; s < D2D_FIT_MAX_S; ++s) {
    // column
    int iC = 8 * s + c - 4;
    bool okc = true;
    if (s == 0) { iC = c < 8 ? cd : c - 4; okc = c >= 8 || cv; }
    if (s == D2D_FIT_MAX_S - 1) { iC = c < 8 ? 8 * s + c - 4 : 44 + cd; okc = c < 8 || cv; }
    // the upper-knot quadrant of every segment but the last went into the next segment's accumulators
    const bool mine = okc && !(highq && s < D2D_FIT_MAX_S - 1);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      int iR = 8 * s + 4 * g4 + rr - 4;
      bool okr = true;
      if (s == 0 && rr < 2) okr = g4 >= 2;                               // rows 0, 1, 4, 5 of knot 0: end conditions
      if (s == 0 && rr >= 2) iR = g4 < 2 ? 2 * g4 + rr - 2 : 4 * g4 + rr - 4;
      if (s == D2D_FIT_MAX_S - 1 && rr < 2) okr = g4 < 2;                // rows 48, 49, 52, 53
      if (s == D2D_FIT_MAX_S - 1 && rr >= 2) iR = g4 < 2 ? 8 * s + 4 * g4 + rr - 4 : 44 + 2 * (g4 - 2) + rr - 2;
      if (mine && okr) Hs[iR * LS + iC] = acc[s][rr];
    }
  }
  wave_lds_sync();
}

struct KnotMetric {
  // everything addressed as byte offsets from the workgroup's LDS base (wave-uniform bases, lane-dependent parts recomputed from
  // the lane where they are used): held as per-lane pointers across the solver loop they were a dozen VGPRs that went to scratch
  unsigned char *lds;
  int sfull_off, md32_off, mr32_off;   // the wave's scatter buffer; the shared tables of the dense metric rows and of the banded ones
  int lane;
  float lam;                    // lam on the rows of the system, 0 elsewhere (lane N carries the right-hand side)
  float scale;                  // = lam
  __device__ __forceinline__ float apply(float v) const {
#ifdef KN_ABL_APPLY
    return v;
#endif
    int l = lane;
    LAUNDER(l);
    const bool act = l < KN_N;
    const int e = act ? kn_entry_of(l) : 0;
    float *sf = reinterpret_cast<float *>(lds + sfull_off);
    wave_lds_sync();
    if (act) sf[8 + e] = v;
    wave_lds_sync();
    const float *bq = sf + 8 * (e >> 3) + 4 * ((e >> 2) & 1);        // knot j-1 (8 floats of padding in front)
    const float *mrow = reinterpret_cast<const float *>(lds + mr32_off) + e * 12;
    const f32x4 s0 = lds_get<f32x4>(bq), s1 = lds_get<f32x4>(bq + 8), s2 = lds_get<f32x4>(bq + 16);
    const f32x4 m0 = lds_get<f32x4>(mrow), m1 = lds_get<f32x4>(mrow + 4), m2 = lds_get<f32x4>(mrow + 8);
    float r = m0.x * s0.x;
    r = fmaf(m0.y, s0.y, r); r = fmaf(m0.z, s0.z, r); r = fmaf(m0.w, s0.w, r);
    r = fmaf(m1.x, s1.x, r); r = fmaf(m1.y, s1.y, r); r = fmaf(m1.z, s1.z, r); r = fmaf(m1.w, s1.w, r);
    r = fmaf(m2.x, s2.x, r); r = fmaf(m2.y, s2.y, r); r = fmaf(m2.z, s2.z, r); r = fmaf(m2.w, s2.w, r);
    return act ? r : 0.f;
  }
  int row_off;                  // byte offset of this lane's row of the dense metric: formed by prepare() inside the solve, dead outside it
  __device__ __forceinline__ void prepare(int l) { row_off = md32_off + (l < KN_N ? l : 0) * (KN_N * 4); }
  __device__ __forceinline__ f32x4 damp(int j0) const {
#ifdef KN_ABL_DAMP
    return f32x4{lam, lam, lam, lam};
#endif
    const float *mdrow = reinterpret_cast<const float *>(lds + row_off);
    return lam != 0.f ? lds_get<f32x4>(mdrow + j0) : f32x4{0.f, 0.f, 0.f, 0.f};     // (unscaled: damped_solve multiplies by `scale`)
  }
};

}  // namespace
// (the factorisation may leave out the tiles a half-bandwidth <= 15 keeps zero: fit_phases.h damped_solve)
template <> struct nd_is_banded<KnotMetric> { static constexpr bool value = true; };
namespace {

// SEG9: no segment holds more than nine samples (K <= 54 at S = 6: the bench's K = 50) -- the MFMA pass keeps 27 operands, not 33
template <bool STAMPS, bool SEG9>
__global__ void __launch_bounds__(64 * KN_WPB_MAX)
fit_lm_knot_kernel(int B, KnotGeom kg, KnotLds L, d2d_fit_opts opts, int iter_cap, KnotDev T, const double *__restrict__ pk,
                   const double *__restrict__ prep, double *q_io, double *cost_io, double *g_io, double *lm, int32_t *flags,
                   int32_t *queue, const int32_t *__restrict__ order, int prio_at, double *u_io, unsigned long long *__restrict__ stamps,
                   float *dbg) {
  // (everything a caller sees -- q, cost, J^T r in q, the lm words, flags, iter_cap / order / prio_at -- means what it means in
  // fit_lm_kernel; u_io [B][64] keeps the knot vector of a fit between launches so that a budgeted solve resumes bit-exactly)
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#define KN_STAMP(i)                                                     \
  if (STAMPS) {                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
    st_acc[i] += t_ - st_last;                                          \
    st_last = t_;                                                       \
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int N = KN_N;
  stage(lds + L.Hb64, T.Hb64, kg.K * KN_HB_STRIDE * 8);
  stage(lds + L.Hb32, T.Hb32, kg.K * 32 * 4);
  stage(lds + L.Wseg, T.Wseg, D2D_FIT_MAX_S * 4 * 64 * 4);
  stage(lds + L.Md32, T.Md32, N * N * 4);
  stage(lds + L.Mi32, T.Mi32, KN_NE * 28 * 4);
  stage(lds + L.Mr32, T.Mrow32, KN_NE * 12 * 4);
  stage(lds + L.Msc, T.msc, KN_NE * 8);
  __syncthreads();
  const double *Hb64 = reinterpret_cast<const double *>(lds + L.Hb64);
  const float *Wseg = reinterpret_cast<const float *>(lds + L.Wseg);
  const float *Mi32 = reinterpret_cast<const float *>(lds + L.Mi32);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned char *wl = lds + L.wave0 + wave * L.wave_stride;
  double *wv = reinterpret_cast<double *>(wl + L.wv);
  double *sp = reinterpret_cast<double *>(wl + L.sp);
  float *sfull = reinterpret_cast<float *>(wl + L.sfull);
  double *gfull = reinterpret_cast<double *>(wl + L.gfull);
  double *park = reinterpret_cast<double *>(wl + L.park);
  double *us = reinterpret_cast<double *>(wl + L.big);
  f32x4 *cf = reinterpret_cast<f32x4 *>(wl + L.cf);
  float2 *cfp = reinterpret_cast<float2 *>(wl + L.cfp);
  float *big = reinterpret_cast<float *>(wl + L.big);
  const bool act = lane < N;
  // this lane's entry: dense index `lane` -> full entry e = 8 j + 4 a + kd; lanes 48 .. 55 own the eight end conditions
  const int e = kn_entry_all(lane);
  const int ej = e >> 3, ea = (e >> 2) & 1, ekd = e & 3;
  const int wv_slot = ej * KN_WV_KNOT + 4 * ea + ekd;
  // samples of the entry's two segments (phase 2)
  // (recomputed where they are used -- a dozen selects -- rather than held in registers across the whole launch)
  auto p2_ranges = [&](int &kb, int &km, int &ke) {
    int lj = lane;
    LAUNDER(lj);
    const int e2 = lj < N ? kn_entry_of(lj) : 0, j2 = e2 >> 3;
    kb = 0; km = 0; ke = 0;
#pragma unroll
    for (int j = 0; j <= D2D_FIT_MAX_S; ++j)
      if (j == j2) { kb = kg.k0[j > 0 ? j - 1 : 0]; km = kg.k0[j]; ke = kg.k0[j < kg.S ? j + 1 : kg.S]; }
  };
  // the sample's segment (phase 1)
  auto seg_of_lane = [&]() -> int {
    int lj = lane;
    LAUNDER(lj);
    int sg = 0;
#pragma unroll
    for (int s = 1; s < D2D_FIT_MAX_S; ++s) sg += (lj >= kg.k0[s] && s < kg.S) ? 1 : 0;
    return sg;
  };
  auto msc_of_lane = [&]() -> double {      // sqrt(Mu_ee): the finish's max-norm scaling (shared table)
    int l = lane;
    LAUNDER(l);
    return reinterpret_cast<const double *>(lds + L.Msc)[l < N ? kn_entry_of(l) : 0];
  };
  for (int i = lane; i < 9 * 8; i += 64) sfull[i] = 0.f;
  // (M v)[lane] for a vector given lane-wise on the free entries: scattered to the [knot][axis][4] layout, the three same-axis
  // quads around the own knot read back (the metric is banded: one knot to either side); damp: lam * (row of M) for the solve
  KnotMetric metric;
  metric.lds = lds; metric.sfull_off = L.wave0 + wave * L.wave_stride + L.sfull; metric.md32_off = L.Md32; metric.mr32_off = L.Mr32;
  metric.lane = lane; metric.lam = 0.f; metric.scale = 0.f; metric.row_off = 0;

  const int stride = gridDim.x * (blockDim.x >> 6);
  auto take = [&](bool first) -> int {
    if (first) {
      const int bi = blockIdx.x + gridDim.x * wave;
      if (bi >= B) return -1;
      return order ? __builtin_amdgcn_readfirstlane(order[bi]) : bi;
    }
    int t = -1;
    int32_t *qp = queue;
    asm volatile("" : "+s"(qp));        // (re-made from the scalar argument here: its vector copy is not held across the fits)
    if (lane == 0 && stride + __hip_atomic_load(qp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B) {
      const int p = stride + atomicAdd(qp, 1);
      if (p < B) t = order ? order[p] : p;
    }
    return __builtin_amdgcn_readfirstlane(t);
  };
  if (STAMPS) st_last = __builtin_amdgcn_s_memtime();
  bool first_take = true;
  for (;;) {
    int b = take(first_take);
    first_take = false;
    if (b < 0) break;
    if (flags[4 * b + FL_STATUS] != D2D_ST_RUNNING) continue;
    const double *prow = prep + (size_t)b * FIT_PREP_STRIDE;
    double *lmb = lm + (size_t)b * LM_STRIDE;       // (formed again from the scalar b in the epilogue: not held across the solver loop)
    int iters = uniform_i(flags[4 * b + FL_ITERS]);
    {
      int lane_ld = lane;
      LAUNDER(lane_ld);
      for (int i = lane_ld; i < FIT_PREP_STRIDE; i += 64) sp[i] = prow[i];
      double wpx = 0.0, wpy = 0.0;
      if (lane_ld < kg.K) { wpx = pk[((size_t)b * FIT_PK + 6) * kg.K + lane_ld]; wpy = pk[((size_t)b * FIT_PK + 7) * kg.K + lane_ld]; }
      park[lane_ld] = wpx; park[64 + lane_ld] = wpy;
    }
    wave_lds_sync();
    // u0 (the knot data of q = 0; for the end conditions: the scaled datum itself) and the start point
    double ui = 0.0;
    auto u0_of_lane = [&]() -> double { int l = lane; LAUNDER(l); return park[128 + l]; };
    {
      double u0i = 0.0;
      int lane_ld = lane;
      LAUNDER(lane_ld);
      const int e_ld = kn_entry_all(lane_ld);        // (from the laundered lane: the per-lane table pointers are not kept across fits)
      if (lane_ld < KN_NE) {
        const double *pu = T.Pu + (size_t)e_ld * 4;
        const double *ed = sp + (((e_ld >> 2) & 1) ? PR_DY : PR_DX);
        u0i = fma(pu[0], ed[0], fma(pu[1], ed[1], fma(pu[2], ed[2], pu[3] * ed[3])));
      }
      ui = u0i;
      park[128 + lane_ld] = u0i;
      if (iters > 0) {
        if (lane_ld < N) ui = u_io[(size_t)b * 64 + lane_ld];
      } else {
        double *qb = reinterpret_cast<double *>(big);
        if (lane_ld < N) qb[lane_ld] = q_io[(size_t)b * N + lane_ld];
        wave_lds_sync();
        if (lane_ld < N) {
          const double *bi = T.Binv + (size_t)e_ld * (N / 2);
          const double *qa = qb + ((e_ld >> 2) & 1) * (N / 2);
          double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
          for (int j = 0; j < N / 2; j += 2) { acc0 = fma(bi[j], qa[j], acc0); acc1 = fma(bi[j + 1], qa[j + 1], acc1); }
          ui = u0i + (acc0 + acc1);
        }
        wave_lds_sync();
      }
      if (lane_ld < KN_NE) wv[wv_slot] = ui;
    }
    // wave-uniform solver state in scalar registers (fit_lm_kernel's scheme)
    double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0, u4 = 0.0, u5 = 0.0, u6 = 0.0, u7 = 1.0;
#define V_lam s0
#define V_nu s1
#define V_mp_par s0
#define V_mp_delta s1
#define V_parl u0
#define V_paru u1
#define V_fp u2
#define V_pn u3
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
    int nev = 0, status = D2D_ST_RUNNING;
    bool so_rows = uniform_i(lmb[3] != 0.0 ? 1 : 0) != 0;
    int phase = 1;
    MpState mp;
    mp.pgn_lds = nullptr;
    mp.dx_gn = 0.0; mp.t2_gn = 0.0; mp.p_gn = 0.f; mp.gn_valid = 0; mp.gn_ok = 0; mp.first = 0; mp.calm = 0; mp.slow = 0; mp.nfac = 0;
    {
      const int pw = uniform_i((int)lmb[6]);
      phase = pw & 1; mp.first = (pw >> 1) & 1; mp.calm = (pw >> 2) & 0x3fff; mp.slow = pw >> 16;
    }
    if (phase == 0) { V_mp_par = uniform_d(lmb[4]); V_mp_delta = uniform_d(lmb[5]); }
    else { V_lam = uniform_d(lmb[0]); V_nu = uniform_d(lmb[1]); }
    double c = 0.0, gi = 0.0, gnrm = -1.0, xub = 0.0;
    float hdiag = 0.f;
    f32x2 hrow[N / 2];
#pragma unroll
    for (int m = 0; m < N / 2; ++m) hrow[m] = f32x2{0.f, 0.f};
    wave_lds_sync();
    if (phase == 0) {                                               // ||x|| = ||u - u0||_M: lmder's first radius is factor * ||x||
      const float dv = act ? (float)(ui - u0_of_lane()) : 0.f;
      const float mv = metric.apply(dv);
      const double xn = sqrt(fmax(uniform_d(wave_sum((double)dv * (double)mv)), 0.0));
      xub = xn * (1.0 + 1e-6);
      if (mp.first && V_mp_delta <= 0.0) V_mp_delta = xn > 0.0 ? 100.0 * xn : 100.0;
    }
    KN_STAMP(0)
    for (bool reenter = true; reenter;) {
      reenter = false;
      c = uniform_d(knot_phase1(kg.K, Hb64, sp, wv, seg_of_lane(), park, us, cf, cfp, so_rows, lane));
      KN_STAMP(1)
      bool fresh = true;
      if (!(fabs(c) <= 1.79e308)) { status = D2D_ST_NONFINITE; fresh = false; }
      int sub = 0, lp_it = 0, att = 0;
      V_alpha = 1.0;
      float dl = 0.f;
      bool fin = false, accept = false;
      while (status == D2D_ST_RUNNING || fresh) {
        bool do_solve = false, is_gn = false, do_trial = false;
        double solve_lam = 0.0;
        int isq_mode = 0;
        if (sub == 0) {
          if (fresh) {
            int p2_kb, p2_km, p2_ke;
            p2_ranges(p2_kb, p2_km, p2_ke);
            gi = knot_phase2(kg, Hb64, us, p2_kb, p2_km, p2_ke, ea, ekd, act, lane);
            KN_STAMP(2)
            fresh = false;
            if (status != D2D_ST_RUNNING) break;
            f32x4 acc[D2D_FIT_MAX_S];
            const float ww = (float)(sp[PR_WWP] * sp[PR_WWP]);
            const int wbase = L.wave0 + wave * L.wave_stride;
            if (so_rows) knot_mfma_so(kg, lds, L.Hb32, wbase + L.cf, wbase + L.cfp, Wseg, ww, lane, acc);
            else if (SEG9) knot_mfma<9>(kg, lds, L.Hb32, wbase + L.cf, Wseg, ww, lane, acc);
            else knot_mfma<KN_SEG_MAX>(kg, lds, L.Hb32, wbase + L.cf, Wseg, ww, lane, acc);
            nev += so_rows ? 3 : 2;
            wave_lds_sync();
            KN_STAMP(3)
            knot_blocks_to_image(acc, big, lane);
            image_put_rhs<N>(big, lane, gi);
            wave_lds_sync();
            image_row<N>(big, lane, hrow);
            hdiag = image_diag<N>(big, lane);
            wave_lds_sync();
            if (STAMPS && dbg != nullptr && iters == 0 && act) {            // development: the first evaluation's H_u row and g_u of every fit
              float *d = dbg + (size_t)b * (N * N + 4 * N) + lane * N;
#pragma unroll
              for (int m = 0; m < N / 2; ++m) { d[2 * m] = hrow[m].x; d[2 * m + 1] = hrow[m].y; }
              dbg[(size_t)b * (N * N + 4 * N) + N * N + lane] = (float)gi;
              dbg[(size_t)b * (N * N + 4 * N) + N * N + N + lane] = (float)ui;
            }
            gnrm = -1.0;                                       // (||J^T f||_2 of this point: computed when lmpar asks for it)
            KN_STAMP(4)
          }
          if (iters >= iter_cap || iters >= opts.max_iter) break;
          if (iters >= prio_at) __builtin_amdgcn_s_setprio(2);
          if (phase == 0) {
            const double fnorm = sqrt(c);
            double gl = 0.0;
            if (act && hdiag > 0.f && fnorm > 0.0) gl = fabs(gi) / (sqrt((double)hdiag) * fnorm);
            V_gnorm = uniform_d(wave_max(gl));
            if (V_gnorm <= opts.mp_gtol) { status = D2D_ST_CONVERGED; break; }
            V_par = V_mp_par; V_parl = 0.0; V_paru = 0.0; V_fp = 0.0; lp_it = 0; V_alpha = 1.0;
            if (!mp.gn_valid) { do_solve = true; is_gn = true; solve_lam = 0.0; isq_mode = 1; }
            else sub = 3;
          } else {
            const double gmax = uniform_d(wave_max(act ? fabs(gi) / msc_of_lane() : 0.0));
            if (gmax <= KN_GTOL_SCALE * opts.gtol) { status = D2D_ST_CONVERGED; break; }
            do_solve = true; solve_lam = V_lam;
          }
        } else if (sub == 1) {
          if (V_par == 0.0) V_par = fmax(MP_DWARF, 0.001 * V_paru);
          do_solve = true; solve_lam = V_par; isq_mode = lp_it + 1 < 10 ? 2 : 0;
        } else if (sub == 2) {
          do_trial = true;
        }
        bool ok = true;
        if (do_solve) {
          double dxn = 0.0, t2 = 0.0;
          float dls, dgi;
          // the damped matrix H_u + lam Mu: the metric's row is added where the factorisation reads the lane's row (damped_solve)
          metric.lam = lane < N ? (float)solve_lam : 0.f; metric.scale = metric.lam;
          ok = uniform_i(damped_solve<N, true, true>(hrow, 0.0, act, lane, big, dgi, dls, nullptr, true, isq_mode, V_mp_delta, &dxn, &t2,
                                                     hdiag, true, metric) ? 1 : 0) != 0;
          KN_STAMP(5)
          if (STAMPS && dbg != nullptr && iters == 0 && mp.nfac == 0 && phase == 0 && act) {
            dbg[(size_t)b * (N * N + 4 * N) + N * N + 2 * N + lane] = dls;
            float *x = dbg + (size_t)b * (N * N + 4 * N) + N * N + 3 * N;
            if (lane == 0) { x[0] = (float)dxn; x[1] = (float)t2; x[2] = (float)gnrm; x[3] = ok ? 1.f : 0.f; x[4] = (float)V_mp_delta; x[5] = (float)solve_lam; }
          }
          if (phase == 0) {
            ++mp.nfac;
            if (is_gn) {
              mp.gn_ok = ok ? 1 : 0; mp.p_gn = dls; mp.dx_gn = dxn; mp.t2_gn = t2; mp.gn_valid = 1;
              sub = 3;
            } else {
              ++lp_it;
              if (!ok) {
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
            // second-order finish (oracle/fit_knot.py finish_knot): (H + lam Mu) s = -g; predicted reduction s^T (lam Mu s - g)
            dl = dls;
            const double delta = (double)dl;
            V_pred = V_lam * dxn * dxn - uniform_d(wave_sum(delta * gi));
            const double msci = msc_of_lane();
            V_dmax = uniform_d(wave_max(act ? fabs(delta) * msci : 0.0)); V_qmax = uniform_d(wave_max(act ? fabs(ui - u0_of_lane()) * msci : 0.0));
            V_ct = 0.0; V_pred_s = V_pred; V_alpha = 1.0; V_bt_a = 0.0; V_bt_b = 0.0; fin = false; accept = false; att = 0;
            if (ok) do_trial = true;
          }
        }
        if (sub == 3) {                                        // lmpar after the Gauss-Newton step (fresh or cached)
          if (mp.gn_ok && mp.dx_gn - V_mp_delta <= 0.1 * V_mp_delta) { dl = mp.p_gn; V_pn = mp.dx_gn; V_par = 0.0; do_trial = true; }
          else {
            if (gnrm < 0.0) {
              // ||J^T f||_2 in q = sqrt(g_u^T Mu^-1 g_u) (lmpar's upper bound and its first damping): g scattered to the
              // [knot][axis][4] layout in fp32, the lane's row of Mu^-1 (fp32, shared table) against the seven same-axis quads
              wave_lds_sync();
              if (act) sfull[8 + e] = (float)gi;
              wave_lds_sync();
              float mg = 0.f;
              const float *mi = Mi32 + e * 28, *gv = sfull + 8 + 4 * ea;
#pragma unroll
              for (int j = 0; j < 7; ++j) {
                const f32x4 a4 = lds_get<f32x4>(mi + 4 * j), g4 = lds_get<f32x4>(gv + 8 * j);
                mg = fmaf(a4.x, g4.x, mg); mg = fmaf(a4.y, g4.y, mg); mg = fmaf(a4.z, g4.z, mg); mg = fmaf(a4.w, g4.w, mg);
              }
              gnrm = sqrt(fmax(uniform_d(wave_sum(act ? (double)mg * gi : 0.0)), 0.0));
            }
            V_fp = mp.gn_ok ? mp.dx_gn - V_mp_delta : 1.79e308;
            V_parl = (mp.gn_ok && mp.t2_gn > 0.0) ? (V_fp / V_mp_delta) / mp.t2_gn : 0.0;
            V_paru = gnrm / V_mp_delta;
            if (V_paru == 0.0) V_paru = MP_DWARF / fmin(V_mp_delta, 0.1);
            V_par = fmin(fmax(V_par, V_parl), V_paru);
            if (V_par == 0.0) V_par = mp.gn_ok ? gnrm / mp.dx_gn : 0.0;
            sub = 1;
          }
        }
        if (!do_trial) {
          if (phase == 0) continue;
          if (sub == 0 && !ok) {
            const StepOutcome so = lm_update(false, false, false, 1.0, c, 0.0, V_pred, V_pred, V_dmax, V_qmax, V_lam, V_nu, opts);
            ++iters;
            V_lam = so.lam; V_nu = so.nu; status = so.status;
            KN_STAMP(6)
          }
          continue;
        }
        // ---- one trial point per pass: a full phase 1 (rows in the mode of the evaluation that follows an acceptance) ----
        if (phase == 0 && mp.first) { V_mp_delta = fmin(V_mp_delta, V_pn); mp.first = 0; }
        const bool so_trial = phase != 0;
        if (act) wv[wv_slot] = ui + V_alpha * (double)dl;
        wave_lds_sync();
        const double ca = uniform_d(knot_phase1(kg.K, Hb64, sp, wv, seg_of_lane(), park, us, cf, cfp, so_trial, lane));
        KN_STAMP(1)
        if (phase == 0) {
          const double fnorm = sqrt(c);
          const bool ctfin = fabs(ca) <= 1.79e308;
          const double fnorm1 = ctfin ? sqrt(ca) : 1.79e308;
          double actred = -1.0;
          if (0.1 * fnorm1 < fnorm) actred = 1.0 - ca / c;
          const double pg = -uniform_d(wave_sum((double)dl * gi));
          const double jp2 = fmax(pg - V_par * V_pn * V_pn, 0.0);
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
            ui += (double)dl; c = ca;
            mp.gn_valid = 0;
            mp.calm = (V_par == 0.0 && ratio >= 0.75) ? mp.calm + 1 : 0;
            fresh = true;
          }
          mp.slow = fabs(actred) <= D2D_LM_MP_SLOW_TOL ? mp.slow + 1 : 0;
          ++iters;
          sub = 0;
          // ||x|| = ||u - u0||_M enters only through  radius <= xtol ||x||  (xtol, eps ~ 1e-15): xub >= ||x|| (triangle inequality,
          // grown by every accepted step) tells when those tests cannot fire, and the norm itself is computed only otherwise
          if (taken) xub += V_pn;
          double xnorm = xub;
          if (V_mp_delta <= fmax(opts.mp_xtol, MP_EPSMCH) * xub) {
            const float dv = act ? (float)(ui - u0_of_lane()) : 0.f;
            const float mv = metric.apply(dv);
            xnorm = sqrt(fmax(uniform_d(wave_sum((double)dv * (double)mv)), 0.0));
            xub = xnorm * (1.0 + 1e-6);
          }
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
            phase = 1; V_lam = D2D_LM_LAMBDA0; V_nu = 2.0; so_rows = true;
            if (act) wv[wv_slot] = ui;
            wave_lds_sync();
            reenter = true;
            break;
          }
          KN_STAMP(6)
          continue;
        }
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
        ++iters;
        V_lam = so.lam; V_nu = so.nu; status = so.status;
        if (so.accept) {
          ui += V_alpha * (double)dl; c = V_ct;
          fresh = true;
          so_rows = true;
        }
        V_alpha = 1.0;
        KN_STAMP(6)
      }
    }
    if (status == D2D_ST_RUNNING && iters >= opts.max_iter) status = D2D_ST_MAXITER;
    // back to the public unknowns: q = Bq (u - u0), J^T r in q = Binv^T g_u (lane = q index: axis a = lane / 24)
    {
      int lane_io = lane;
      LAUNDER(lane_io);
      wave_lds_sync();
      if (lane_io < KN_NE) gfull[e] = act ? ui - u0_of_lane() : 0.0;
      wave_lds_sync();
      double qv = 0.0, gq = 0.0;
      const int qa = lane_io >= N / 2 ? 1 : 0;
      if (lane_io < N) {
        const double *bq = T.Bq + (size_t)lane_io * 28;
        const double *dv = gfull + 4 * qa;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          a0 = fma(bq[4 * j], dv[8 * j], a0); a1 = fma(bq[4 * j + 1], dv[8 * j + 1], a1);
          a0 = fma(bq[4 * j + 2], dv[8 * j + 2], a0); a1 = fma(bq[4 * j + 3], dv[8 * j + 3], a1);
        }
        qv = a0 + a1;
      }
      wave_lds_sync();
      if (lane_io < KN_NE) gfull[e] = act ? gi : 0.0;
      wave_lds_sync();
      if (lane_io < N) {
        const double *bt = T.BiT + (size_t)lane_io * 28;
        const double *gv = gfull + 4 * qa;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          a0 = fma(bt[4 * j], gv[8 * j], a0); a1 = fma(bt[4 * j + 1], gv[8 * j + 1], a1);
          a0 = fma(bt[4 * j + 2], gv[8 * j + 2], a0); a1 = fma(bt[4 * j + 3], gv[8 * j + 3], a1);
        }
        gq = a0 + a1;
        q_io[(size_t)b * N + lane_io] = qv;
        g_io[(size_t)b * N + lane_io] = gq;
        if (status == D2D_ST_RUNNING) u_io[(size_t)b * 64 + lane_io] = ui;      // (only a fit that will be resumed needs its knot vector kept)
      }
      const double gmax = uniform_d(wave_max(fabs(gq)));
      if (lane == 0) {
        int bs = b;
        LAUNDER_S(bs);
        double *lmb = lm + (size_t)bs * LM_STRIDE;
        int *flags_b = flags + 4 * (size_t)bs;
        cost_io[bs] = c;
        lmb[2] = gmax; lmb[3] = so_rows ? 1.0 : 0.0;
        if (phase == 0) { lmb[4] = V_mp_par; lmb[5] = V_mp_delta; } else { lmb[0] = V_lam; lmb[1] = V_nu; }
        lmb[6] = (double)(phase | (mp.first << 1) | ((mp.calm & 0x3fff) << 2) | (mp.slow << 16));
        lmb[7] = lmb[7] + (double)mp.nfac;
        flags_b[FL_STATUS] = status; flags_b[FL_ITERS] = iters; flags_b[FL_NEED] = 1;
        flags_b[FL_NEVAL] = flags_b[FL_NEVAL] + nev;
      }
    }
    __builtin_amdgcn_s_setprio(0);
    KN_STAMP(0)
  }
  if (queue != nullptr && lane == 0) {
    if (atomicAdd(queue + 1, 1) == stride - 1) { queue[0] = 0; queue[1] = 0; queue[2] = 0; queue[3] = 0; queue[4] = 0; }
  }
  if (STAMPS && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&stamps[i], st_acc[i]);
#undef KN_STAMP
#undef V_lam
#undef V_nu
#undef V_mp_par
#undef V_mp_delta
#undef V_parl
#undef V_paru
#undef V_fp
#undef V_pn
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

template <typename T>
int upload(T **dst, const std::vector<T> &src) {
  D2D_CHECK_HIP(hipMalloc(dst, src.size() * sizeof(T)));
  D2D_CHECK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return D2D_OK;
}

}  // namespace

// The headline shape -- S = 6 (seven knots, 48 free entries), K <= 64, default solver -- runs on the knot kernel;
// d2d_fit_plan_opts.kernel = D2D_FIT_KERNEL_FUSED at plan creation keeps the q-coordinate kernel (fit_lm_kernel) for A/B runs and
// for the tests that compare the two (d2d_fit_plan_create_ex does not call this function then).
int fit_knot_plan_init(d2d_fit_plan *pl) {
  pl->kn.wpb = 0;
  if (pl->S != 6 || pl->nq != 24 || pl->K > 64) return D2D_OK;
  if (int rc = fit_basis_knots(pl)) return rc;
  for (int s2 = 0; s2 < pl->S; ++s2)
    if (pl->kn.k0[s2 + 1] - pl->kn.k0[s2] > KN_SEG_MAX) return D2D_OK;      // (cannot happen at K <= 64, S = 6)
  int wpb = 0;
  for (int w = KN_WPB_MAX; w >= 4; --w)
    if (knot_lds_layout(pl->K, w).total <= KN_LDS_BYTES) { wpb = w; break; }
  if (wpb == 0) return D2D_OK;
  auto &kn = pl->kn;
  int rc = upload(&kn.d_Hb64, kn.Hb64);
  if (!rc) rc = upload(&kn.d_Hb32, kn.Hb32);
  if (!rc) rc = upload(&kn.d_Wseg, kn.Wseg);
  if (!rc) rc = upload(&kn.d_Md32, kn.Md32);
  if (!rc) rc = upload(&kn.d_Mrow32, kn.Mrow32);
  if (!rc) rc = upload(&kn.d_Mi32, kn.Mi32);
  if (!rc) rc = upload(&kn.d_Bq, kn.Bq);
  if (!rc) rc = upload(&kn.d_BiT, kn.BiT);
  if (!rc) rc = upload(&kn.d_Binv, kn.Binv);
  if (!rc) rc = upload(&kn.d_Minv, kn.Minv);
  if (!rc) rc = upload(&kn.d_Pu, kn.Pu);
  if (!rc) rc = upload(&kn.d_msc, kn.msc);
  if (rc) return rc;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_lm_knot_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, KN_LDS_BYTES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_lm_knot_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, KN_LDS_BYTES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_lm_knot_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, KN_LDS_BYTES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fit_lm_knot_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, KN_LDS_BYTES);
  kn.wpb = wpb;
  return D2D_OK;
}

void fit_knot_plan_free(d2d_fit_plan *pl) {
  auto &kn = pl->kn;
  void *p[] = {kn.d_Hb64, kn.d_Hb32, kn.d_Wseg, kn.d_Md32, kn.d_Mrow32, kn.d_Mi32, kn.d_Bq, kn.d_BiT, kn.d_Binv, kn.d_Minv, kn.d_Pu, kn.d_msc, kn.d_u};
  for (void *q : p)
    if (q) hipFree(q);
  kn.d_Hb64 = kn.d_Bq = kn.d_BiT = kn.d_Binv = kn.d_Minv = kn.d_Pu = kn.d_msc = kn.d_u = nullptr;
  kn.d_Hb32 = kn.d_Wseg = kn.d_Md32 = kn.d_Mrow32 = kn.d_Mi32 = nullptr;
  kn.wpb = 0;
}

int fit_knot_ensure(d2d_fit_plan *pl, int cap_B) {
  auto &kn = pl->kn;
  if (kn.wpb == 0) return D2D_OK;
  if (kn.d_u) { hipFree(kn.d_u); kn.d_u = nullptr; }
  D2D_CHECK_HIP(hipMalloc(&kn.d_u, (size_t)cap_B * 64 * sizeof(double)));
  // (a fit is resumed from its row only after a launch of THIS kernel left it RUNNING -- d2d_fit_iterate keeps a solve on one
  // kernel -- but no read of this buffer shall ever see uninitialised memory)
  D2D_CHECK_HIP(hipMemset(kn.d_u, 0, (size_t)cap_B * 64 * sizeof(double)));
  return D2D_OK;
}

int fit_knot_launch(d2d_ctx *ctx, d2d_fit_plan *pl, int B, double *q, const d2d_fit_opts &o, int iter_cap, const int32_t *order, int prio_at) {
  auto &kn = pl->kn;
  KnotGeom kg;
  kg.K = pl->K; kg.S = pl->S; kg.smax = 0;
  for (int s = 0; s <= D2D_FIT_MAX_S + 1; ++s) kg.k0[s] = kn.k0[s < D2D_FIT_MAX_S + 2 ? s : D2D_FIT_MAX_S + 1];
  for (int j = 0; j < pl->S; ++j)
    if (kn.k0[j + 1] - kn.k0[j] > kg.smax) kg.smax = kn.k0[j + 1] - kn.k0[j];
  const KnotLds L = knot_lds_layout(pl->K, kn.wpb);
  KnotDev T{kn.d_Hb64, kn.d_Bq, kn.d_BiT, kn.d_Binv, kn.d_Minv, kn.d_Pu, kn.d_msc, kn.d_Hb32, kn.d_Wseg, kn.d_Md32, kn.d_Mrow32, kn.d_Mi32};
  static const bool want_stamps = getenv("D2D_LM_STAMPS") != nullptr || getenv("D2D_KNOT_DEBUG") != nullptr;     // (the dump lives in the stamped build)
  unsigned long long *stamps = want_stamps ? reinterpret_cast<unsigned long long *>(ctx->stats_dev + 8) : nullptr;
  if (want_stamps) D2D_CHECK_HIP(hipMemsetAsync(stamps, 0, 8 * sizeof(unsigned long long), ctx->stream));
  const int blocks = B < pl->n_cu ? B : pl->n_cu;
  int32_t *queue = ctx->counter_dev + 8;
  float *dbg = nullptr;                  // development: D2D_KNOT_DEBUG=<file> dumps the first evaluation (H_u rows, g_u, u) of every fit
  if (getenv("D2D_KNOT_DEBUG")) D2D_CHECK_HIP(hipMalloc(&dbg, (size_t)B * (KN_N * KN_N + 4 * KN_N) * sizeof(float)));
#define KN_LAUNCH(ST, S9)                                                                                                                  \
  hipLaunchKernelGGL((fit_lm_knot_kernel<ST, S9>), dim3(blocks), dim3(64 * kn.wpb), L.total, ctx->stream, B, kg, L, o, iter_cap, T, pl->d_pk, \
                     pl->d_prep, q, pl->d_cost, pl->d_g, pl->d_lm, pl->d_flags, queue, order, prio_at, kn.d_u, stamps, dbg)
  const bool seg9 = kg.smax <= 9;
  if (want_stamps) { if (seg9) KN_LAUNCH(true, true); else KN_LAUNCH(true, false); }
  else { if (seg9) KN_LAUNCH(false, true); else KN_LAUNCH(false, false); }
#undef KN_LAUNCH
  D2D_LAUNCH_CHECK();
  if (dbg) {
    std::vector<float> h((size_t)B * (KN_N * KN_N + 4 * KN_N));
    D2D_CHECK_HIP(hipMemcpyAsync(h.data(), dbg, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (FILE *f = fopen(getenv("D2D_KNOT_DEBUG"), "wb")) { fwrite(h.data(), sizeof(float), h.size(), f); fclose(f); }
    hipFree(dbg);
  }
  if (want_stamps) {
    unsigned long long h[8];
    D2D_CHECK_HIP(hipMemcpyAsync(h, stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    static const char *nm[7] = {"io/loop", "phase1", "phase2", "mfma", "image+gnrm", "solve", "judge"};
    double tot = 0;
    for (int i = 0; i < 7; ++i) tot += (double)h[i];
    fprintf(stderr, "[fit_lm_knot stamps] wave-cycles (s_memtime ticks), B=%d iter_cap=%d:", B, iter_cap);
    for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)h[i] / tot);
    fprintf(stderr, " total=%.3e\n", tot);
  }
  return D2D_OK;
}
