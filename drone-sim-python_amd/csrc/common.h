// Shared host-side plumbing of libd2dhip.so: context, error reporting, launch checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/d2d.h"

struct d2d_ctx {
  int device;
  hipStream_t stream;
  // scratch owned by the context
  double *Bmat_dev = nullptr;   // incidence matrix + z_des of the last gvf run
  size_t Bmat_cap = 0;
  int32_t *counter_dev = nullptr;  // 16 small device counters: [0] fit convergence poll, [2] hand-out counter of nlp_solve_kernel, [8 .. 15] queue words of the fit kernels
  int32_t *counter_host = nullptr; // pinned mirror
  double *stats_dev = nullptr;
  double *stats_host = nullptr;
};

void d2d_set_error(const char *fmt, ...);

#define D2D_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      d2d_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return D2D_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

#define D2D_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      d2d_set_error(__VA_ARGS__);     \
      return D2D_EINVAL;              \
    }                                 \
  } while (0)

#define D2D_LAUNCH_CHECK() D2D_CHECK_HIP(hipGetLastError())
