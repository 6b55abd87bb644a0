// Sanitizer driver for the CPU-side host code of libd2dhip.so (`make asan`): the basis construction of
// csrc/fit_basis.cpp (dense fp64 algebra on std::vector storage, the part of the library that indexes host memory by
// hand) built with -fsanitize=address,undefined and run over the plan shapes the planners use, including the failure
// paths.  The GPU side cannot be sanitised on this pool; kernels are checked by the parity tests.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "fit_plan.h"

static std::string g_err;
void d2d_set_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

static int check_plan(int S, int K, double duration) {
  d2d_fit_plan pl;
  pl.device = 0; pl.S = S; pl.K = K; pl.duration = duration;
  const double s = 0.1 / K;
  pl.wref[0] = 4e-4; pl.wref[1] = 5.0 * s; pl.wref[2] = s / (9.81 * 9.81);
  const int rc = fit_basis_build(&pl);
  if (rc != 0) { std::printf("S=%d K=%d: rc=%d (%s)\n", S, K, rc, g_err.c_str()); return rc; }
  const int nq = pl.nq;
  // Pinit is the left inverse of G0 on the reduced unknowns: Pinit G0 = I
  double worst = 0.0;
  for (int i = 0; i < nq; ++i)
    for (int j = 0; j < nq; ++j) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += pl.Pinit[(size_t)i * K + k] * pl.G[(size_t)k * nq + j];
      worst = std::fmax(worst, std::fabs(acc - (i == j ? 1.0 : 0.0)));
    }
  std::printf("S=%d K=%d nq=%d: |Pinit G0 - I| = %.2e\n", S, K, nq, worst);
  // the segment formulation (fit_basis_segments, csrc/fit_seg.h): G_d[k] = Psi_d(x_k) . Zl_s(k) with Psi_d the d-th TIME derivative
  // of the Legendre polynomials P_0 .. P_7 on the segment -- the device's recurrences restated here -- and the lane map covers
  // every sample exactly once
  if (int rc2 = fit_basis_segments(&pl)) { std::printf("  segments: rc=%d (%s)\n", rc2, g_err.c_str()); return rc2; }
  double wseg = 0.0, gmax = 0.0;
  const double c1 = 2.0 / pl.T;
  for (int k = 0; k < K; ++k) {
    const double x = pl.sx[k];
    double P[8], d1[8], d2[8];
    P[0] = 1; P[1] = x; d1[0] = 0; d1[1] = 1; d2[0] = d2[1] = 0;
    for (int n = 1; n < 7; ++n) {
      P[n + 1] = ((2.0 * n + 1.0) * x * P[n] - n * P[n - 1]) / (n + 1.0);
      d1[n + 1] = d1[n - 1] + (2.0 * n + 1.0) * P[n];
      d2[n + 1] = d2[n - 1] + (2.0 * n + 1.0) * d1[n];
    }
    for (int d = 0; d < 3; ++d)
      for (int j = 0; j < nq; ++j) {
        double acc = 0.0;
        for (int i = 0; i < 8; ++i) {
          const double psi = d == 0 ? P[i] : (d == 1 ? c1 * d1[i] : c1 * c1 * d2[i]);
          acc += psi * pl.Zl[(size_t)(8 * pl.seg[k] + i) * nq + j];
        }
        const double gv = pl.G[((size_t)d * K + k) * nq + j];
        wseg = std::fmax(wseg, std::fabs(acc - gv));
        gmax = std::fmax(gmax, std::fabs(gv));
      }
  }
  int lanes = 0, covered = 0, zlbig = 0;
  for (int s = 0; s < S; ++s) {
    const int L = pl.seg_l0[s + 1] - pl.seg_l0[s];
    lanes += L;
    covered += pl.seg_Ks[s];
    if (L < 1 || (pl.seg_Ks[s] + L - 1) / L > pl.seg_nchunk) zlbig = 1;
  }
  double zlmax = 0.0;
  for (double v : pl.Zl) zlmax = std::fmax(zlmax, std::fabs(v));
  std::printf("  segments: |Psi Zl - G| = %.2e of %.2e, |Zl| <= %.2f, %d lanes, %d chunks\n", wseg, gmax, zlmax, lanes, pl.seg_nchunk);
  if (!(wseg < 1e-9 * gmax) || lanes > 64 || covered != K || zlbig || !(zlmax < 50.0)) return 1;
  return worst < 1e-6 ? 0 : 1;
}

int main() {
  int bad = 0;
  const int shapes[][2] = {{6, 50}, {6, 61}, {6, 121}, {6, 151}, {6, 501}, {3, 20}, {1, 8}, {4, 31}};
  for (auto &sh : shapes) bad += check_plan(sh[0], sh[1], 0.1 * (sh[1] - 1)) != 0;
  // failure path: too few samples for the unknowns -> a clean error code, no out-of-bounds access
  d2d_fit_plan pl;
  pl.device = 0; pl.S = 6; pl.K = 18; pl.duration = 1.0; pl.wref[0] = 4e-4; pl.wref[1] = 1e-2; pl.wref[2] = 1e-5;
  const int rc = fit_basis_build(&pl);
  std::printf("S=6 K=18 (rank-deficient): rc=%d (%s)\n", rc, g_err.c_str());
  bad += rc == 0;
  std::printf(bad ? "asan_host: FAILED\n" : "asan_host: ok\n");
  return bad;
}
