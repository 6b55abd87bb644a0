// Development unit test of fit_phases.h damped_solve on one wavefront: random SPD systems against a host solve in double.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I drone-sim-python_amd/csrc -I include tools/dev/chol_unit.hip -o gpurun_out/chol_unit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "fit_phases.h"
#ifdef OLD_API      // the round-3 solve (tools/dev/build_unit.sh /tmp/base/... -DOLD_API): no hd / FULL parameters, image of N+3 rows
#define CHOL_IMAGE_BYTES(n) (((n) + 4) * ((n) + 4) * 4)
template <int N> __device__ __forceinline__ float image_diag(const float *Hs, int lane) { return Hs[(lane < N ? lane : 0) * (N + 5)]; }
#define SOLVE(N, MP, FULL, ...) damped_solve<N, MP>(__VA_ARGS__)
#define HDARGS
#else
#define SOLVE(N, MP, FULL, ...) damped_solve<N, MP, FULL>(__VA_ARGS__, hd, true)
#endif

template <int N, bool MP, bool FULL>
__global__ void __launch_bounds__(64) k_solve(const float *H, const float *b, double lam, int unit, int n_act, float *out, double *aux) {
  __shared__ __attribute__((aligned(16))) float img[CHOL_IMAGE_BYTES(N) / 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < N * N; i += 64) img[(i / N) * CHOL_LS + (i % N)] = H[i];
  if (lane < N) img[N * CHOL_LS + lane] = b[lane];
  __syncthreads();
  f32x2 hrow[N / 2];
  image_row<N>(img, lane, hrow);
  const float hd = image_diag<N>(img, lane);
  wave_lds_sync();
  float dgi, delta;
  double dxn = 0.0, isq = 0.0;
  const bool act = lane < n_act;
  const bool ok = SOLVE(N, MP, FULL, hrow, lam, act, lane, img, dgi, delta, nullptr, unit != 0, MP ? 1 : 0, 0.0, &dxn, &isq);
  out[lane] = delta;
  if (lane == 0) { aux[0] = dxn; aux[1] = isq; aux[2] = ok ? 1.0 : 0.0; }
}

template <int N, bool MP, bool FULL>
static int run(int n_act, double lam, int unit, unsigned seed) {
  std::mt19937 rng(seed);
  std::normal_distribution<double> nd;
  std::vector<double> A(N * N, 0.0), J(3 * N * N);
  for (auto &v : J) v = nd(rng);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0;
      for (int k = 0; k < 3 * N; ++k) s += J[k * N + i] * J[k * N + j];
      A[i * N + j] = (i < n_act && j < n_act) ? s / N : 0.0;
    }
  std::vector<float> Hf(N * N), bf(N);
  std::vector<double> b(N);
  for (int i = 0; i < N * N; ++i) Hf[i] = (float)A[i];
  for (int i = 0; i < N; ++i) { b[i] = i < n_act ? nd(rng) : 0.0; bf[i] = (float)b[i]; }
  // host: (A + damp) x = b
  std::vector<double> M(n_act * n_act), x(n_act);
  for (int i = 0; i < n_act; ++i)
    for (int j = 0; j < n_act; ++j) M[i * n_act + j] = (double)Hf[i * N + j] + (i == j ? (unit ? lam : lam * std::fmax(std::fabs((double)Hf[i * N + i]), 1e-30)) : 0.0);
  std::vector<double> L(n_act * n_act, 0.0), y(n_act);
  for (int j = 0; j < n_act; ++j) {
    double d = M[j * n_act + j];
    for (int k = 0; k < j; ++k) d -= L[j * n_act + k] * L[j * n_act + k];
    L[j * n_act + j] = std::sqrt(d);
    for (int i = j + 1; i < n_act; ++i) {
      double s = M[i * n_act + j];
      for (int k = 0; k < j; ++k) s -= L[i * n_act + k] * L[j * n_act + k];
      L[i * n_act + j] = s / L[j * n_act + j];
    }
  }
  for (int i = 0; i < n_act; ++i) { double s = (double)bf[i]; for (int k = 0; k < i; ++k) s -= L[i * n_act + k] * y[k]; y[i] = s / L[i * n_act + i]; }
  for (int i = n_act - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n_act; ++k) s -= L[k * n_act + i] * x[k]; x[i] = s / L[i * n_act + i]; }
  double xn = 0; for (double v : x) xn += v * v; xn = std::sqrt(xn);
  std::vector<double> z(n_act);
  for (int i = 0; i < n_act; ++i) { double s = x[i] / xn; for (int k = 0; k < i; ++k) s -= L[i * n_act + k] * z[k]; z[i] = s / L[i * n_act + i]; }
  double isq = 0; for (double v : z) isq += v * v;
  float *dH, *db, *dout; double *daux;
  hipMalloc(&dH, N * N * 4); hipMalloc(&db, N * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&daux, 3 * 8);
  hipMemcpy(dH, Hf.data(), N * N * 4, hipMemcpyHostToDevice); hipMemcpy(db, bf.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_solve<N, MP, FULL>), dim3(1), dim3(64), 0, 0, dH, db, lam, unit, n_act, dout, daux);
  float out[64]; double aux[3];
  hipMemcpy(out, dout, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(aux, daux, 3 * 8, hipMemcpyDeviceToHost);
  double err = 0, big = 0;
  for (int i = 0; i < n_act; ++i) { err = std::fmax(err, std::fabs(out[i] - x[i])); big = std::fmax(big, std::fabs(x[i])); }
  for (int i = n_act; i < 64; ++i) err = std::fmax(err, std::fabs((double)out[i]));
  printf("N=%d MP=%d FULL=%d n_act=%d lam=%g unit=%d: max err %.3e (|x|max %.3e) ok=%g dxn %.6e (host %.6e) isq %.6e (host %.6e)\n", N, (int)MP, (int)FULL, n_act, lam, unit,
         err, big, aux[2], aux[0], xn, aux[1], isq);
  hipFree(dH); hipFree(db); hipFree(dout); hipFree(daux);
  return err <= 2e-4 * big ? 0 : 1;
}

// Throughput of the solve under the fused kernel's occupancy: 256 workgroups x 8 wavefronts, every wave its own image in the LDS
// (11 kB per wave + a 49.5 kB block that is only allocated), REPS solves in a row.  Prints microseconds per solve and wave.
template <int N, bool MP, bool FULL>
#ifndef BENCH_THREADS
#define BENCH_THREADS 512
#endif
__global__ void __launch_bounds__(BENCH_THREADS) k_bench(const float *H, const float *b, double lam, int reps, float *out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *img = lds + (BENCH_THREADS > 512 ? 2048 : 12672) + wave * (CHOL_IMAGE_BYTES(N) / 4 + 288);
  for (int i = lane; i < N * N; i += 64) img[(i / N) * CHOL_LS + (i % N)] = H[i];
  if (lane < N) img[N * CHOL_LS + lane] = b[lane];
  wave_lds_sync();
  f32x2 hrow[N / 2];
  image_row<N>(img, lane, hrow);
  const float hd = image_diag<N>(img, lane);
  wave_lds_sync();
  float acc = 0.f;
#ifdef BENCH_STAMPS     // -DBENCH_STAMPS: cycles of setup, block columns 0 / 1 / 2, substitutions (damped_solve's tt) of wave 0 of block 0
  unsigned long long tt[5] = {0, 0, 0, 0, 0};
  unsigned long long *ttp = tt;
#else
  unsigned long long *ttp = nullptr;
#endif
  for (int r = 0; r < reps; ++r) {
    float dgi, delta;
    double dxn = 0.0, isq = 0.0;
    SOLVE(N, MP, FULL, hrow, lam * (1 + r), lane < N, lane, img, dgi, delta, ttp, true, MP ? 2 : 0, 1e30, &dxn, &isq);
    acc += delta;
#ifdef USE_ISQ
    acc += (float)isq;
#endif
  }
  out[blockIdx.x * BENCH_THREADS + threadIdx.x] = acc;
#ifdef BENCH_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("  stamps (cycles per solve): setup %llu, block columns %llu %llu %llu, substitutions %llu\n", tt[0] / reps, tt[1] / reps, tt[2] / reps, tt[3] / reps, tt[4] / reps);
#endif
}

template <int N, bool MP, bool FULL>
static void bench(int waves) {
  std::vector<float> Hf(N * N, 0.f), bf(N, 1.f);
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) Hf[i * N + j] = (i == j ? 2.f : 0.f) + 1.f / (1 + i + j);
  float *dH, *db, *dout;
  hipMalloc(&dH, N * N * 4); hipMalloc(&db, N * 4); hipMalloc(&dout, 256 * 1024 * 4);
  hipMemcpy(dH, Hf.data(), N * N * 4, hipMemcpyHostToDevice); hipMemcpy(db, bf.data(), N * 4, hipMemcpyHostToDevice);
  const int lds = 12672 * 4 + waves * (CHOL_IMAGE_BYTES(N) + 288 * 4) > 160 * 1024 ? 160 * 1024 : 12672 * 4 + waves * (CHOL_IMAGE_BYTES(N) + 288 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<N, MP, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 400;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_bench<N, MP, FULL>), dim3(256), dim3(64 * waves), lds, 0, dH, db, 1e-3, reps, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it == 2) printf("bench N=%d MP=%d waves/CU=%d: %.3f us per solve and wave (%.1f solves/us chip-wide)\n", N, (int)MP, waves, 1e3 * ms / reps, 256.0 * waves * reps / (1e3 * ms));
  }
  hipFree(dH); hipFree(db); hipFree(dout);
}

int main() {
  int bad = 0;
  bad += run<48, true, true>(48, 1e-3, 1, 1);
  bad += run<48, true, true>(48, 0.0, 1, 2);
  bad += run<48, true, true>(48, 0.5, 0, 3);
  bad += run<48, false, true>(48, 1e-3, 0, 4);
  bad += run<48, false, false>(40, 1e-3, 0, 5);
  bad += run<32, false, false>(32, 1e-2, 0, 6);
  bad += run<16, false, false>(12, 1e-2, 0, 7);
  printf(bad ? "FAILED %d\n" : "all ok\n", bad);
#if BENCH_THREADS > 512
  bench<48, true, true>(12);
  bench<48, true, true>(8);
#else
  bench<48, true, true>(8);
  bench<48, true, true>(4);
  bench<48, false, true>(8);
#endif
  return bad;
}
