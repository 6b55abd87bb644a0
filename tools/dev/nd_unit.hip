// Unit test + micro-benchmark of the banded nested-dissection solve (csrc/fit_nd.h): random block-tridiagonal SPD systems of the
// knot-coordinate fit's shape (56 rows, 8 x 8 blocks, the eight end-condition rows as identity) against a dense fp64 Cholesky on
// the host: the solution, || L^-1 w ||^2 for w = the solution scaled, and the detection of an indefinite matrix.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I drone-sim-python_amd/csrc tools/dev/nd_unit.hip -o drone-sim-python_amd/lib/nd_unit
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fit_phases.h"
#include "fit_nd.h"

#ifndef WPB
#define WPB 8
#endif
__global__ void __launch_bounds__(64 * WPB) nd_unit_kernel(int nprob, int reps, const float *__restrict__ A, const float *__restrict__ rhs,
                                                          float *__restrict__ s_out, double *__restrict__ misc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *nd = reinterpret_cast<float *>(lds) + wave * ND_FLOATS;
  nd_init(nd, lane);
  for (int p = blockIdx.x * WPB + wave; p < nprob; p += gridDim.x * WPB) {
    float left[8], own[8], right[8];
    const int row = lane < ND_ROWS ? lane : 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      left[c] = A[((size_t)p * ND_ROWS + row) * 24 + c];
      own[c] = A[((size_t)p * ND_ROWS + row) * 24 + 8 + c];
      right[c] = A[((size_t)p * ND_ROWS + row) * 24 + 16 + c];
    }
    const float b = lane < ND_ROWS ? rhs[(size_t)p * ND_ROWS + lane] : 0.f;
    float dinv = 1.f, s = 0.f;
    bool pos = true;
    double isq = 0.0;
    for (int r = 0; r < reps; ++r) {
      pos = nd_factor(left, own, right, b, nd, lane, dinv);
      s = nd_back(nd, lane, dinv);
      isq = nd_isq(nd, lane, dinv, s);
      wave_lds_sync();
    }
    if (lane < ND_ROWS) s_out[(size_t)p * ND_ROWS + lane] = s;
    if (lane == 0) { misc[2 * p] = isq; misc[2 * p + 1] = pos ? 1.0 : 0.0; }
  }
}

static bool fixed_row(int e) { return (e < 8 || e >= 48) && (e & 3) < 2; }

int main(int argc, char **argv) {
  const int nprob = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 1;
  const int n = ND_ROWS;
  std::vector<float> A((size_t)nprob * n * 24, 0.f), b((size_t)nprob * n);
  std::vector<double> D((size_t)nprob * n * n, 0.0);
  srand(1);
  auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
  for (int p = 0; p < nprob; ++p) {
    double *M = &D[(size_t)p * n * n];
    // sum over segments of random 16-column rows (like J^T J of local samples) + a small multiple of the identity
    for (int s = 0; s < 6; ++s)
      for (int k = 0; k < 24; ++k) {
        double v[16];
        for (int c = 0; c < 16; ++c) v[c] = rnd();
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) M[(8 * s + i) * n + 8 * s + j] += v[i] * v[j];
      }
    for (int i = 0; i < n; ++i) M[i * n + i] += 0.05;
    if (p == nprob - 1 && nprob > 1) M[20 * n + 20] -= 40.0;           // the last problem is indefinite
    for (int i = 0; i < n; ++i)
      if (fixed_row(i)) {
        for (int j = 0; j < n; ++j) { M[i * n + j] = 0.0; M[j * n + i] = 0.0; }
        M[i * n + i] = 1.0;
      }
    for (int i = 0; i < n; ++i) {
      b[(size_t)p * n + i] = fixed_row(i) ? 0.f : (float)rnd();
      const int kb = i >> 3;
      for (int sl = 0; sl < 24; ++sl) {
        const int j = 8 * (kb - 1) + sl;
        A[((size_t)p * n + i) * 24 + sl] = (j >= 0 && j < n) ? (float)M[i * n + j] : 0.f;
      }
    }
  }
  float *dA, *db, *ds;
  double *dm;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&ds, b.size() * 4); hipMalloc(&dm, (size_t)nprob * 16);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  const int blocks = (nprob + WPB - 1) / WPB < 256 ? (nprob + WPB - 1) / WPB : 256;
  const size_t lds = (size_t)WPB * ND_BYTES;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&nd_unit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(nd_unit_kernel, dim3(blocks), dim3(64 * WPB), lds, 0, nprob, 1, dA, db, ds, dm);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(nd_unit_kernel, dim3(blocks), dim3(64 * WPB), lds, 0, nprob, reps, dA, db, ds, dm);
  hipEventRecord(e1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> s((size_t)nprob * n);
  std::vector<double> misc((size_t)nprob * 2);
  hipMemcpy(s.data(), ds, s.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(misc.data(), dm, misc.size() * 8, hipMemcpyDeviceToHost);
  double worst_s = 0, worst_i = 0;
  int bad_pos = 0;
  for (int p = 0; p < nprob; ++p) {
    // dense Cholesky of the fp32-rounded matrix (what the kernel was given)
    std::vector<double> Lc((size_t)n * n, 0.0);
    const double *M = &D[(size_t)p * n * n];
    bool ok = true;
    for (int j = 0; j < n && ok; ++j) {
      double d = (double)(float)M[j * n + j];
      for (int k = 0; k < j; ++k) d -= Lc[j * n + k] * Lc[j * n + k];
      if (!(d > 0)) { ok = false; break; }
      d = std::sqrt(d); Lc[j * n + j] = d;
      for (int i = j + 1; i < n; ++i) {
        double v = (double)(float)M[i * n + j];
        for (int k = 0; k < j; ++k) v -= Lc[i * n + k] * Lc[j * n + k];
        Lc[i * n + j] = v / d;
      }
    }
    if ((misc[2 * p + 1] != 0.0) != ok) { ++bad_pos; printf("problem %d: pos %g, host says %d\n", p, misc[2 * p + 1], (int)ok); }
    if (!ok) continue;
    std::vector<double> y(n), x(n), z(n);
    for (int i = 0; i < n; ++i) { double v = b[(size_t)p * n + i]; for (int k = 0; k < i; ++k) v -= Lc[i * n + k] * y[k]; y[i] = v / Lc[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < n; ++k) v -= Lc[k * n + i] * x[k]; x[i] = v / Lc[i * n + i]; }
    double xm = 0, em = 0;
    for (int i = 0; i < n; ++i) { xm = std::fmax(xm, std::fabs(x[i])); em = std::fmax(em, std::fabs(x[i] - s[(size_t)p * n + i])); }
    worst_s = std::fmax(worst_s, em / xm);
    // || L^-1 w ||^2 with w = the kernel's own solution (what the kernel was asked for)
    double q = 0;
    for (int i = 0; i < n; ++i) { double v = s[(size_t)p * n + i]; for (int k = 0; k < i; ++k) v -= Lc[i * n + k] * z[k]; z[i] = v / Lc[i * n + i]; q += z[i] * z[i]; }
    worst_i = std::fmax(worst_i, std::fabs(q - misc[2 * p]) / q);
    if (p < 2) printf("problem %d: max rel err of s %.3g, isq %.9g (host %.9g)\n", p, em / xm, misc[2 * p], q);
  }
  printf("%d problems: worst rel err of the solution %.3g, of isq %.3g, wrong definiteness verdicts %d; %d x (factor + back + isq) per problem: %.3f ms -> %.2f us per solve and wave (%d waves per CU)\n",
         nprob, worst_s, worst_i, bad_pos, reps, ms, 1e3 * ms * blocks * WPB / ((double)nprob * reps) , WPB);
  return (worst_s < 1e-3 && worst_i < 1e-3 && bad_pos == 0) ? 0 : 2;
}
