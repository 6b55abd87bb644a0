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
