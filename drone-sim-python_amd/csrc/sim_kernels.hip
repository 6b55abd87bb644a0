// Plant / guidance / tracking kernels and their C-ABI entry points (include/d2d.h).
// One drone per lane, state in registers across the whole time loop, fp64.
#include <cmath>
#include <vector>

#include "common.h"
#include "sim_device.h"

static GlMesh make_mesh(double dt, double tau_phi, double tau_v) {
  // oracle/sim.py gl_mesh: geometric panels resolve the phi boundary layer (tau_phi << dt)
  static const double edges[D2D_GL_PANELS + 1] = {0.0, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0 / 2, 1.0};
  static const double c[4] = {0.06943184420297371, 0.33000947820757187, 0.6699905217924281,
                              0.9305681557970262};
  GlMesh m;
  for (int p = 0; p < D2D_GL_PANELS; ++p) {
    const double w = dt * (edges[p + 1] - edges[p]);
    m.row[p][0] = w;
    for (int i = 0; i < 4; ++i) {
      m.row[p][1 + i] = std::exp(-c[i] * w / tau_phi);
      m.row[p][5 + i] = std::exp(-c[i] * w / tau_v);
    }
    m.row[p][9] = std::exp(-w / tau_phi);
    m.row[p][10] = std::exp(-w / tau_v);
  }
  // the one-panel mesh of a step that starts close to its bank command (sim_device.h plant_step)
  static_assert(D2D_GL_FAST_STAGES == 6, "the 6-stage Gauss nodes below");
  static const double c6[6] = {0.033765242898423975, 0.16939530676686776, 0.3806904069584015, 0.6193095930415985, 0.8306046932331322,
                               0.966234757101576};
  m.fast[0] = dt;
  for (int i = 0; i < 6; ++i) {
    m.fast[1 + i] = std::exp(-c6[i] * dt / tau_phi);
    m.fast[7 + i] = std::exp(-c6[i] * dt / tau_v);
  }
  m.fast[13] = std::exp(-dt / tau_phi);
  m.fast[14] = std::exp(-dt / tau_v);
  m.fast_dphi = (dt <= D2D_GL_FAST_RATIO * tau_phi) ? D2D_GL_FAST_DPHI : -1.0;
  return m;
}

// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) step_kernel(int n, const double *__restrict__ X,
                                                   const double *__restrict__ U, double wx,
                                                   double wy, GlMesh mesh,
                                                   double *__restrict__ Xout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  State5 s = {X[i], X[n + i], X[2 * n + i], X[3 * n + i], X[4 * n + i]};
  s = plant_step(s, U[i], U[n + i], wx, wy, mesh);
  Xout[i] = s.x; Xout[n + i] = s.y; Xout[2 * n + i] = s.psi; Xout[3 * n + i] = s.phi;
  Xout[4 * n + i] = s.v;
}

// ------------------------------------------------------------------------------------
// Circular formation: block = fpb formations of n_ac consecutive threads.
// NAC = 0: any formation size -- the phase angles / errors / stop flags of a formation travel through the LDS
//          (theta[256] | e[256] | ok[256 ints]) with two block barriers per step;
// NAC = 1, 2, 4: the formation is (part of) a DPP quad of ONE wavefront (64-thread blocks): its members' values are fetched with
//          v_mov_b32 quad_perm -- no LDS round trip, no barrier in the step (a 10 000-step loop is one dependent chain per drone and,
//          at 65 536 drones, alone on its SIMD: every exposed latency is paid in full).  Same sums in the same order: bit-identical.
// LDS (both): B[n_ac*(n_ac-1)] | zdes[n_ac-1] behind the three exchange arrays.
template <int CTRL>
__device__ __forceinline__ double quad_get(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// member K of the lane's own formation (NAC aircraft, quad-aligned): quad_perm [K,K,K,K] for four, [K,K,2+K,2+K] for two
template <int NAC, int K>
__device__ __forceinline__ double member_get(double v) {
  if (NAC == 4) return quad_get<K * 0x55>(v);
  if (NAC == 2) return quad_get<(K & 1) * 0x05 + (2 + (K & 1)) * 0x50>(v);
  return v;
}
template <int NAC, int K>
__device__ __forceinline__ int member_get_i(int v) {
  if (NAC == 4) return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, true);
  if (NAC == 2) return __builtin_amdgcn_mov_dpp(v, (K & 1) * 0x05 + (2 + (K & 1)) * 0x50, 0xf, 0xf, true);
  return v;
}

// VCONST (the launch puts one wave on a SIMD and that wave has 512 VGPRs): the tableau of the one-panel step (36 + 6 doubles in constant
// memory) and its mesh row (15, kernel arguments) are copied into vector registers before the time loop.  Wave-uniform fp64 constants
// live in scalar register pairs, of which there are ~50; what does not fit is fetched again in EVERY step (s_load: 19 per step, each
// behind its own wait -- measured with SQ_WAIT_ANY: 30 % of the wave's cycles, a lone wave hides nothing) or re-made from literals
// (s_mov_b32: a scalar instruction costs a lone wave the same issue slot as an fp64 FMA; 263 per step against 802 vector ones).
template <int NAC, bool VCONST>
__device__ __forceinline__ void
gvf_run_body(const d2d_gvf_params &p, const GlMesh &mesh, int fpb, const double *__restrict__ X0,
             const double *__restrict__ centres, const double *__restrict__ radius,
             const double *__restrict__ Bz, const double *__restrict__ X0f,
             double *__restrict__ X_hist, double *__restrict__ U_hist,
             double *__restrict__ Rr_hist, double *__restrict__ eth_hist,
             double *__restrict__ X_final, int32_t *__restrict__ stop_row, int32_t *__restrict__ conv_row) {
  extern __shared__ double lds[];
  constexpr bool QUAD = NAC != 0;
  double *sh_theta = lds;
  double *sh_e = lds + 256;
  int *sh_ok = reinterpret_cast<int *>(lds + 512);
  double *sh_B = lds + 512 + 128;
  const int n_ac = QUAD ? NAC : p.n_ac, nm = n_ac - 1;
  double *sh_z = sh_B + n_ac * nm;
  for (int i = threadIdx.x; i < n_ac * nm + nm; i += blockDim.x) sh_B[i] = Bz[i];

  const int t = threadIdx.x;
  const int fl = t / n_ac, a = t - fl * n_ac;
  const int f = blockIdx.x * fpb + fl;
  const bool live = (fl < fpb) && (f < p.n_form);
  const long N = (long)p.n_form * n_ac;
  const long d = (long)f * n_ac + a;
  const int base = live ? fl * n_ac : 0;

  State5 s = {0, 0, 0, 0, 10};
  double cx = 0, cy = 0, R = 1, fx = 0, fy = 0, fpsi = 0;
  if (live) {
    s = {X0[d], X0[N + d], X0[2 * N + d], X0[3 * N + d], X0[4 * N + d]};
    cx = centres[d]; cy = centres[N + d]; R = radius[d];
    if (X0f) { fx = X0f[d]; fy = X0f[N + d]; fpsi = X0f[2 * N + d]; }
    if (X_hist) {
      X_hist[d] = s.x; X_hist[N + d] = s.y; X_hist[2 * N + d] = s.psi; X_hist[3 * N + d] = s.phi;
      X_hist[4 * N + d] = s.v;
    }
  }
  __syncthreads();
  double fast6[3 + 2 * D2D_GL_FAST_STAGES];
#pragma unroll
  for (int r = 0; r < 3 + 2 * D2D_GL_FAST_STAGES; ++r) fast6[r] = mesh.fast[r];
  double ga6[D2D_GL_FAST_STAGES][D2D_GL_FAST_STAGES], gb6[D2D_GL_FAST_STAGES];
#pragma unroll
  for (int r = 0; r < D2D_GL_FAST_STAGES; ++r) {
    gb6[r] = GL6_B[r];
#pragma unroll
    for (int c = 0; c < D2D_GL_FAST_STAGES; ++c) ga6[r][c] = GL6_A[r][c];
  }
  if (VCONST) {
#pragma unroll
    for (int r = 0; r < 3 + 2 * D2D_GL_FAST_STAGES; ++r) asm volatile("" : "+v"(fast6[r]));
#pragma unroll
    for (int r = 0; r < D2D_GL_FAST_STAGES; ++r) {
      asm volatile("" : "+v"(gb6[r]));
      if (r < D2D_GL_FAST_STAGES / 2) {                   // (gl_panel reads the upper half of the symmetric tableau only)
#pragma unroll
        for (int c = 0; c < D2D_GL_FAST_STAGES; ++c) asm volatile("" : "+v"(ga6[r][c]));
      }
    }
  }
  // the lane's entries of the incidence matrix and its desired phase (QUAD: read once; VCONST: kept in vector registers -- re-read
  // from the LDS every step they were two exposed round trips of the step's dependent chain)
  constexpr int GVF_FA = 4;
  double bc_q[GVF_FA] = {0, 0, 0, 0}, br_q[GVF_FA - 1] = {0, 0, 0}, zd_q = 0.0;
  if (QUAD) {
    const int ac = a < nm ? a : 0;
#pragma unroll
    for (int k = 0; k < GVF_FA; ++k) bc_q[k] = (k < NAC && NAC > 1) ? sh_B[k * nm + ac] : 0.0;
#pragma unroll
    for (int m = 0; m < GVF_FA - 1; ++m) br_q[m] = m < NAC - 1 ? sh_B[a * nm + m] : 0.0;
    zd_q = NAC > 1 ? sh_z[ac] : 0.0;
    if (VCONST) {
#pragma unroll
      for (int k = 0; k < GVF_FA; ++k) asm volatile("" : "+v"(bc_q[k]));
#pragma unroll
      for (int m = 0; m < GVF_FA - 1; ++m) asm volatile("" : "+v"(br_q[m]));
      asm volatile("" : "+v"(zd_q));
    }
  }
  const int rs = p.rec_stride;
  // history rows: i % rs and i / rs of this step and of the one before, kept by counting (a division by a run-time number is ~25
  // scalar instructions, four of them per step)
  int ph_prev = 0, row_prev = 0, ph = rs == 1 ? 0 : 1, row = rs == 1 ? 1 : 0;
  int my_stop = p.n_rows;      // formation-uniform
  bool prev_all_ok = false;    // result of the stop test at the end of the previous step
  int n_true = 0, first_true = -1;   // phase-error rule (use_stop == 2): steps on which it held so far, loop index i-1 of the first one
  double sin_psi = 0.0, cos_psi = 1.0;
  // Formations of up to GVF_FA = 4 aircraft (what the reference flies): a step requests all its partners' angles / errors and the
  // matrix entries that go with them AT ONCE.  As loops over the shared arrays every term was an LDS round trip of its own in front
  // of its fma (the trip count is a run-time number): seven in a row per step at four aircraft, a tenth of a step that is one
  // dependent chain from end to end.  The sums keep their order (terms beyond n_ac add 0 * x): bit-identical.  (Held in registers
  // across the steps instead, the matrix entries put the kernel at one wave per SIMD.)
  const bool small_form = n_ac <= GVF_FA;
  for (int i = 1; i < p.n_rows; ++i) {
    // src/11_full_sim_case1.py:140 -- `if np.all(stop)==1 and t>0: break` at the top of step i
    if (p.use_stop) {
      // use_stop == 2: src/12_full_sim_case2.py:156-164 / 12_full_sim_case3.py:163-178 -- `break` at the END of the step on
      // which the rule fired, rows [:i+1] are kept, which is the same row count as a break at the top of the next step
      if (prev_all_ok && (p.use_stop == 2 || (i - 1) > 0) && my_stop == p.n_rows) my_stop = i;
      // every formation of this block (QUAD: of this wavefront, which is the block) has stopped: nothing left to integrate
      const int gone = (!live || my_stop != p.n_rows) ? 1 : 0;
      if (QUAD ? __all(gone) : __syncthreads_and(gone)) break;
    }
    const bool run = live && (my_stop == p.n_rows);
    // DCFController.get, src/d2d/guidance.py:103-126
    const double theta = atan2(s.y - cy, s.x - cx);
    double e = 0.0, Ur = 0.0;
    if (QUAD) {
      // partners' angles and errors by DPP; the matrix entries from the LDS (read-only after the barrier above: requested before
      // the atan2 result is needed)
      const double (&bc)[GVF_FA] = bc_q;
      const double (&br)[GVF_FA - 1] = br_q;
      const double zd = zd_q;
      double th[GVF_FA];
      th[0] = member_get<NAC, 0>(theta); th[1] = member_get<NAC, 1>(theta); th[2] = member_get<NAC, 2>(theta); th[3] = member_get<NAC, 3>(theta);
      double z = 0.0;
#pragma unroll
      for (int k = 0; k < GVF_FA; ++k)
        if (k < NAC) z += bc[k] * th[k];
      if (a < nm) {
        e = z - zd;
        if (e > D2D_PI) e -= D2D_TWO_PI;
        if (e <= -D2D_PI) e += D2D_TWO_PI;
      }
      double ev[GVF_FA - 1];
      ev[0] = member_get<NAC, 0>(e); ev[1] = member_get<NAC, 1>(e); ev[2] = member_get<NAC, 2>(e);
#pragma unroll
      for (int m = 0; m < GVF_FA - 1; ++m)
        if (m < NAC - 1) Ur += br[m] * ev[m];
    } else {
      sh_theta[t] = theta;
      __syncthreads();
      if (small_form) {
        double th[GVF_FA], bc[GVF_FA];
        const int ac = a < nm ? a : 0;
#pragma unroll
        for (int k = 0; k < GVF_FA; ++k) { const int kk = k < n_ac ? k : 0; th[k] = sh_theta[base + kk]; bc[k] = sh_B[kk * nm + ac]; }
        const double zd = sh_z[ac];
        double z = 0.0;
#pragma unroll
        for (int k = 0; k < GVF_FA; ++k) z += (k < n_ac ? bc[k] : 0.0) * th[k];
        if (a < nm) {
          e = z - zd;
          if (e > D2D_PI) e -= D2D_TWO_PI;
          if (e <= -D2D_PI) e += D2D_TWO_PI;
        }
      } else if (a < nm) {
        double z = 0.0;
        for (int k = 0; k < n_ac; ++k) z += sh_B[k * nm + a] * sh_theta[base + k];
        e = z - sh_z[a];
        if (e > D2D_PI) e -= D2D_TWO_PI;
        if (e <= -D2D_PI) e += D2D_TWO_PI;
      }
      sh_e[t] = e;
      __syncthreads();
      if (small_form) {
        double ev[GVF_FA - 1], br[GVF_FA - 1];
#pragma unroll
        for (int m = 0; m < GVF_FA - 1; ++m) { const int mm = m < nm ? m : 0; ev[m] = sh_e[base + mm]; br[m] = nm > 0 ? sh_B[a * nm + mm] : 0.0; }
#pragma unroll
        for (int m = 0; m < GVF_FA - 1; ++m) Ur += (m < nm ? br[m] : 0.0) * ev[m];
      } else {
        for (int m = 0; m < nm; ++m) Ur += sh_B[a * nm + m] * sh_e[base + m];
      }
    }
    Ur *= -p.kr;
    const double Rr = Ur + R;
    // (sin, cos) of the heading, shared by the guidance law and the plant: the plant step hands the pair of its new heading on
    // (rotations by the turn of a panel: one rounding each), a sincos refreshes it every 16th step
    if ((i & 15) == 1) sincos(s.psi, &sin_psi, &cos_psi);
    double tan_c;
    const double phi_c = gvf_bank_cmd(s, sin_psi, cos_psi, cx, cy, Rr, p.ke, p.kd, nullptr, nullptr, &tan_c);
    double sn_n = sin_psi, cs_n = cos_psi;
    State5 sn = plant_step<true>(s, phi_c, p.v_c, p.wx, p.wy, mesh, sn_n, cs_n, tan_c, &ga6, &gb6, fast6);
    if (run) {
      if (U_hist && ph_prev == 0) {
        const long r = (long)row_prev * 2 * N;
        U_hist[r + d] = phi_c; U_hist[r + N + d] = p.v_c;
      }
      if (ph == 0) {
        if (X_hist) {
          const long r = (long)row * 5 * N;
          X_hist[r + d] = sn.x; X_hist[r + N + d] = sn.y; X_hist[r + 2 * N + d] = sn.psi;
          X_hist[r + 3 * N + d] = sn.phi; X_hist[r + 4 * N + d] = sn.v;
        }
        if (Rr_hist) Rr_hist[(long)row * N + d] = Rr;
        if (eth_hist && a < nm) eth_hist[(long)row * p.n_form * nm + (long)f * nm + a] = e * (180.0 / D2D_PI);
      }
      s = sn;
      sin_psi = sn_n; cos_psi = cs_n;
    }
    ph_prev = ph; row_prev = row;
    if (++ph == rs) { ph = 0; ++row; }
    if (p.use_stop) {
      // :170-175 -- |X[0:3]-X0f[0:3]| <= tol for every aircraft of the formation
      bool ok = fabs(sn.x - fx) <= p.stop_tol[0] && fabs(sn.y - fy) <= p.stop_tol[1] &&
                fabs(sn.psi - fpsi) <= p.stop_tol[2];
      // phase-error rule: every (signed) inter-vehicle phase error of this step, in degrees, <= stop_tol[0]
      if (p.use_stop == 2) ok = (a < nm) ? (e * (180.0 / D2D_PI) <= p.stop_tol[0]) : true;
      bool all_ok = true;
      if (QUAD) {
        const int oki = ok ? 1 : 0;
        int oks[GVF_FA];
        oks[0] = member_get_i<NAC, 0>(oki); oks[1] = member_get_i<NAC, 1>(oki); oks[2] = member_get_i<NAC, 2>(oki); oks[3] = member_get_i<NAC, 3>(oki);
#pragma unroll
        for (int k = 0; k < GVF_FA; ++k)
          if (k < NAC) all_ok = all_ok && (oks[k] != 0);
      } else {
        sh_ok[t] = ok ? 1 : 0;
        __syncthreads();
        if (small_form) {
          int oks[GVF_FA];
#pragma unroll
          for (int k = 0; k < GVF_FA; ++k) oks[k] = sh_ok[base + (k < n_ac ? k : 0)];
#pragma unroll
          for (int k = 0; k < GVF_FA; ++k) all_ok = all_ok && (oks[k] != 0);
        } else {
          for (int k = 0; k < n_ac; ++k) all_ok = all_ok && (sh_ok[base + k] != 0);
        }
      }
      if (p.use_stop == 2) {
        // ... and the loop goes on for stop_hold more steps on which the rule holds (the time the planner needs, case 3)
        if (all_ok && my_stop == p.n_rows) {
          if (first_true < 0) first_true = i - 1;
          if (n_true >= p.stop_hold) prev_all_ok = true; else { ++n_true; prev_all_ok = false; }
        } else {
          prev_all_ok = false;
        }
      } else {
        prev_all_ok = all_ok;
      }
      // (the next iteration's first barrier orders these reads before sh_ok is rewritten)
    }
  }
  if (live) {
    X_final[d] = s.x; X_final[N + d] = s.y; X_final[2 * N + d] = s.psi; X_final[3 * N + d] = s.phi;
    X_final[4 * N + d] = s.v;
    if (a == 0 && stop_row) stop_row[f] = (p.use_stop == 2 && prev_all_ok && my_stop == p.n_rows) ? p.n_rows : my_stop;
    if (a == 0 && conv_row) conv_row[f] = first_true;
  }
}

#define GVF_ARGS d2d_gvf_params p, GlMesh mesh, int fpb, const double *__restrict__ X0, const double *__restrict__ centres,          \
                 const double *__restrict__ radius, const double *__restrict__ Bz, const double *__restrict__ X0f,                  \
                 double *__restrict__ X_hist, double *__restrict__ U_hist, double *__restrict__ Rr_hist, double *__restrict__ eth_hist, \
                 double *__restrict__ X_final, int32_t *__restrict__ stop_row, int32_t *__restrict__ conv_row
#define GVF_PASS p, mesh, fpb, X0, centres, radius, Bz, X0f, X_hist, U_hist, Rr_hist, eth_hist, X_final, stop_row, conv_row
// any formation size (LDS exchange); two waves per SIMD for 131 072 drones and more
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) gvf_run_kernel(GVF_ARGS) { gvf_run_body<0, false>(GVF_PASS); }
// quad-aligned formations (1, 2, 4 aircraft), one wavefront per block; ..._wide: the launch puts at most one wave on a SIMD
// (<= 1024 blocks: BASELINE configs[4], 65 536 drones), which then has the whole register file -- no scratch
template <int NAC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) gvf_run_quad_kernel(GVF_ARGS) { gvf_run_body<NAC, false>(GVF_PASS); }
template <int NAC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) gvf_run_quad_wide_kernel(GVF_ARGS) { gvf_run_body<NAC, true>(GVF_PASS); }

// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
ctrl_gain_kernel(d2d_track_params p, const double *__restrict__ X, const double *__restrict__ Yref,
                 double *__restrict__ Xr, double *__restrict__ dX, double *__restrict__ U,
                 double *__restrict__ Kg) {
  const int n = p.n;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;   // keep the whole wave converged in care_sda's __all()
  State5 s = {X[i], X[n + i], X[2 * n + i], X[3 * n + i], X[4 * n + i]};
  double Y[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) Y[c] = Yref[(long)c * n + i];
  const GainOut o = compute_gain(s, Y, p);
  if (!live) return;
  if (Xr) { Xr[i] = o.Xr.x; Xr[n + i] = o.Xr.y; Xr[2 * n + i] = o.Xr.psi; Xr[3 * n + i] = o.Xr.phi; Xr[4 * n + i] = o.Xr.v; }
  if (dX) {
#pragma unroll
    for (int c = 0; c < 5; ++c) dX[(long)c * n + i] = o.dX[c];
  }
  if (U) { U[i] = o.U[0]; U[n + i] = o.U[1]; }
  if (Kg) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 5; ++c) Kg[(long)(r * 5 + c) * n + i] = o.K[r][c];
  }
}

// DFFFController.get for n independent (state, reference sample) pairs
__global__ void __launch_bounds__(64)
dfff_kernel(d2d_track_params p, const double *__restrict__ X, const double *__restrict__ Yref,
            double *__restrict__ Xr, double *__restrict__ U, double *__restrict__ Kg) {
  const int n = p.n;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;   // keep the whole wave converged in care_sda's __all()
  const State5 s = {X[i], X[n + i], X[2 * n + i], X[3 * n + i], X[4 * n + i]};
  double Y[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) Y[c] = Yref[(long)c * n + i];
  const DfffOut o = dfff_gain(s, Y, p);
  if (!live) return;
  if (Xr) {
#pragma unroll
    for (int c = 0; c < 5; ++c) Xr[(long)c * n + i] = o.Xr[c];
  }
  if (U) { U[i] = o.U[0]; U[n + i] = o.U[1]; }
  if (Kg) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Kg[(long)(r * 3 + c) * n + i] = o.K[r][c];
  }
}

// run_simulation of src/05_test_simulation.py:21-34 with the legacy DFFFController (src/d2d/guidance.py:52-91):
//   U[i-1] = ctl.get(X[i-1], t[i-1]);  X[i] = disc_dyn(X[i-1], U[i-1]) + perts[i];  U[T-1] = ctl.get(X[T-1], t[T-1])
// Yref [n_rows][6][n] = the trajectory's flat outputs at the sample times (x, y, xd, yd, xdd, ydd).
__global__ void __launch_bounds__(64)
dfff_run_kernel(d2d_track_params p, GlMesh mesh, const double *__restrict__ Yref, const double *__restrict__ perts,
                const double *__restrict__ X0, double *__restrict__ X_hist, double *__restrict__ U_hist,
                double *__restrict__ Xr_hist, double *__restrict__ X_final) {
  const long n = p.n;
  long d = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const bool live = d < n;
  if (!live) d = n - 1;                               // (keeps the wave converged in care_sda's __all())
  State5 s = {X0[d], X0[n + d], X0[2 * n + d], X0[3 * n + d], X0[4 * n + d]};
  if (live && X_hist) {
    X_hist[d] = s.x; X_hist[n + d] = s.y; X_hist[2 * n + d] = s.psi; X_hist[3 * n + d] = s.phi; X_hist[4 * n + d] = s.v;
  }
  for (int i = 1; i <= p.n_rows; ++i) {
    const long q = i - 1;                             // the row the controller acts on
    double Y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) Y[c] = Yref[(q * 6 + c) * n + d];
    const DfffOut o = dfff_gain(s, Y, p);
    if (live) {
      if (U_hist) { U_hist[q * 2 * n + d] = o.U[0]; U_hist[q * 2 * n + n + d] = o.U[1]; }
      if (Xr_hist) {
#pragma unroll
        for (int c = 0; c < 5; ++c) Xr_hist[(q * 5 + c) * n + d] = o.Xr[c];
      }
    }
    if (i == p.n_rows) break;                         // the last row only gets its command (:33)
    s = plant_step(s, o.U[0], o.U[1], p.wx, p.wy, mesh);
    if (perts) {
      const double *pr = perts + (long)i * 5 * n + d;
      s.x += pr[0]; s.y += pr[n]; s.psi += pr[2 * n]; s.phi += pr[3 * n]; s.v += pr[4 * n];
    }
    if (live && X_hist) {
      double *o5 = X_hist + (long)i * 5 * n;
      o5[d] = s.x; o5[n + d] = s.y; o5[2 * n + d] = s.psi; o5[3 * n + d] = s.phi; o5[4 * n + d] = s.v;
    }
  }
  if (live && X_final) {
    X_final[d] = s.x; X_final[n + d] = s.y; X_final[2 * n + d] = s.psi; X_final[3 * n + d] = s.phi; X_final[4 * n + d] = s.v;
  }
}

// numpy.gradient(f, edge_order=2) with unit spacing at row i of a [n_rows][n] plane
__device__ __forceinline__ double grad2(const double *__restrict__ f, int i, int T, long n, long d) {
  if (i == 0) return -1.5 * f[d] + 2.0 * f[n + d] - 0.5 * f[2 * n + d];
  if (i == T - 1) return 1.5 * f[(long)(T - 1) * n + d] - 2.0 * f[(long)(T - 2) * n + d] + 0.5 * f[(long)(T - 3) * n + d];
  return 0.5 * (f[(long)(i + 1) * n + d] - f[(long)(i - 1) * n + d]);
}

// ComputeDerivatives, src/11_full_sim_case1.py:197-204: first pass; the second pass reuses it.
__global__ void __launch_bounds__(256)
gradient_kernel(int T, int n, double inv_dt, const double *__restrict__ f, double *__restrict__ out) {
  const long d = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (d >= n) return;
  out[(long)i * n + d] = grad2(f, i, T, n, d) * inv_dt;
}

// Tracking loop, src/11_full_sim_case1.py:272-290.
__global__ void __launch_bounds__(64)
track_run_kernel(d2d_track_params p, GlMesh mesh, const double *__restrict__ x_ref,
                 const double *__restrict__ y_ref, const double *__restrict__ xd,
                 const double *__restrict__ yd, const double *__restrict__ xdd,
                 const double *__restrict__ ydd, const double *__restrict__ X0,
                 double *__restrict__ X_hist, double *__restrict__ U_hist,
                 double *__restrict__ Xr_hist, double *__restrict__ dX_hist,
                 double *__restrict__ Yd_hist, double *__restrict__ Ydd_hist,
                 double *__restrict__ X_final) {
  const long n = p.n;
  long d = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const bool live = d < n;
  if (!live) d = n - 1;
  State5 s = {X0[d], X0[n + d], X0[2 * n + d], X0[3 * n + d], X0[4 * n + d]};
  if (live && X_hist) {
    X_hist[d] = s.x; X_hist[n + d] = s.y; X_hist[2 * n + d] = s.psi; X_hist[3 * n + d] = s.phi; X_hist[4 * n + d] = s.v;
  }
  for (int i = 1; i < p.n_rows; ++i) {
    const long r = (long)i * n + d;
    double Y[8] = {x_ref[r], y_ref[r], xd[r], yd[r], xdd[r], ydd[r], 0.0, 0.0};   // Yddd = [0,0] (:279)
    const GainOut o = compute_gain(s, Y, p);
    s = plant_step(s, o.U[0], o.U[1], p.wx, p.wy, mesh);
    if (live) {
      const long q = (long)(i - 1);
      if (U_hist) { U_hist[q * 2 * n + d] = o.U[0]; U_hist[q * 2 * n + n + d] = o.U[1]; }
      if (dX_hist) {
#pragma unroll
        for (int c = 0; c < 5; ++c) dX_hist[q * 5 * n + c * n + d] = o.dX[c];
      }
      if (Xr_hist) {
        double *o5 = Xr_hist + q * 5 * n;
        o5[d] = o.Xr.x; o5[n + d] = o.Xr.y; o5[2 * n + d] = o.Xr.psi; o5[3 * n + d] = o.Xr.phi; o5[4 * n + d] = o.Xr.v;
      }
      if (Yd_hist) { Yd_hist[q * 2 * n + d] = Y[2]; Yd_hist[q * 2 * n + n + d] = Y[3]; }
      if (Ydd_hist) { Ydd_hist[q * 2 * n + d] = Y[4]; Ydd_hist[q * 2 * n + n + d] = Y[5]; }
      if (X_hist) {
        double *o5 = X_hist + (long)i * 5 * n;
        o5[d] = s.x; o5[n + d] = s.y; o5[2 * n + d] = s.psi; o5[3 * n + d] = s.phi; o5[4 * n + d] = s.v;
      }
    }
  }
  if (live && X_final) {
    X_final[d] = s.x; X_final[n + d] = s.y; X_final[2 * n + d] = s.psi; X_final[3 * n + d] = s.phi; X_final[4 * n + d] = s.v;
  }
}

// ------------------------------------------------------------------------------------
// Single-evaluation entry points behind the reference's per-call helper methods.
__global__ void __launch_bounds__(256)
dcf_eval_kernel(int n_form, int n_ac, double kr, const double *__restrict__ Bz,
                const double *__restrict__ centres, const double *__restrict__ pos,
                double *__restrict__ U_r, double *__restrict__ eth_deg) {
  // one thread per formation (n_ac <= 64): tiny, latency-bound by design
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_form) return;
  const long N = (long)n_form * n_ac;
  const int nm = n_ac - 1;
  double th[64], e[64];
  for (int a = 0; a < n_ac; ++a) {
    const long d = (long)f * n_ac + a;
    th[a] = atan2(pos[N + d] - centres[N + d], pos[d] - centres[d]);
  }
  for (int m = 0; m < nm; ++m) {
    double z = 0.0;
    for (int k = 0; k < n_ac; ++k) z += Bz[k * nm + m] * th[k];
    double ee = z - Bz[n_ac * nm + m];
    if (ee > D2D_PI) ee -= D2D_TWO_PI;
    if (ee <= -D2D_PI) ee += D2D_TWO_PI;
    e[m] = ee;
    if (eth_deg) eth_deg[(long)f * nm + m] = ee * (180.0 / D2D_PI);
  }
  for (int a = 0; a < n_ac; ++a) {
    double u = 0.0;
    for (int m = 0; m < nm; ++m) u += Bz[a * nm + m] * e[m];
    U_r[(long)f * n_ac + a] = -kr * u;
  }
}

__global__ void __launch_bounds__(256)
gvf_eval_kernel(int n, double ke, double kd, const double *__restrict__ X, const double *__restrict__ e,
                const double *__restrict__ nv, const double *__restrict__ H, double *__restrict__ U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const State5 s = {X[i], X[n + i], X[2 * n + i], X[3 * n + i], X[4 * n + i]};
  const double Hm[4] = {H[i], H[n + i], H[2 * n + i], H[3 * n + i]};
  double U1, U2;
  double sin_psi, cos_psi;
  sincos(s.psi, &sin_psi, &cos_psi);
  const double Ut = gvf_control(s, sin_psi, cos_psi, e[i], nv[i], nv[n + i], Hm, ke, kd, &U1, &U2);
  U[i] = Ut; U[n + i] = U1; U[2 * n + i] = U2;
}

__global__ void __launch_bounds__(256)
flatness_kernel(int variant, int n, double wx, double wy, double tau_phi, double tau_v,
                const double *__restrict__ Yref, double *__restrict__ X, double *__restrict__ U,
                double *__restrict__ Xdot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double Y[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) Y[c] = Yref[(long)c * n + i];
  double Xo[5], Uo[2], Xd[5] = {0, 0, 0, 0, 0};
  if (variant == 0) {
    flat_state_input(Y, wx, wy, tau_phi, tau_v, Xo, Uo, Xd);
  } else {
    State5 xr;
    compute_flatness(Y, wx, wy, tau_phi, tau_v, xr, Uo[0], Uo[1]);
    Xo[0] = xr.x; Xo[1] = xr.y; Xo[2] = xr.psi; Xo[3] = xr.phi; Xo[4] = xr.v;
  }
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    X[(long)c * n + i] = Xo[c];
    if (Xdot) Xdot[(long)c * n + i] = Xd[c];
  }
  U[i] = Uo[0]; U[n + i] = Uo[1];
}

__global__ void __launch_bounds__(256)
cont_jac_kernel(int n, double tau_phi, double tau_v, const double *__restrict__ Xr,
                double *__restrict__ A, double *__restrict__ B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Mat<5> Am = cont_jac_A(Xr[2 * n + i], Xr[3 * n + i], Xr[4 * n + i], tau_phi, tau_v);
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) A[(long)(r * 5 + c) * n + i] = Am.a[r][c];
#pragma unroll
  for (int k = 0; k < 10; ++k) B[(long)k * n + i] = 0.0;
  B[(long)6 * n + i] = 1.0 / tau_phi;    // B[3][0]
  B[(long)9 * n + i] = 1.0 / tau_v;      // B[4][1]
}

struct LqrWeights { double Q[25]; double Rinv[4]; };

__global__ void __launch_bounds__(64)
lqr_kernel(int n, LqrWeights w, const double *__restrict__ A, const double *__restrict__ B,
           double *__restrict__ K, double *__restrict__ P) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  Mat<5> Am, G, Q;
  double Bm[5][2];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
#pragma unroll
    for (int c = 0; c < 5; ++c) { Am.a[r][c] = A[(long)(r * 5 + c) * n + i]; Q.a[r][c] = w.Q[r * 5 + c]; }
    Bm[r][0] = B[(long)(r * 2) * n + i]; Bm[r][1] = B[(long)(r * 2 + 1) * n + i];
  }
  double BR[5][2];                      // B R^-1
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    BR[r][0] = Bm[r][0] * w.Rinv[0] + Bm[r][1] * w.Rinv[2];
    BR[r][1] = Bm[r][0] * w.Rinv[1] + Bm[r][1] * w.Rinv[3];
  }
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) G.a[r][c] = BR[r][0] * Bm[c][0] + BR[r][1] * Bm[c][1];
  const Mat<5> Pm = care_sda<5>(Am, G, Q);
  if (!live) return;
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c)
      if (P) P[(long)(r * 5 + c) * n + i] = Pm.a[r][c];
  // K = R^-1 B^T P
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 5; ++r) acc += (w.Rinv[u * 2 + 0] * Bm[r][0] + w.Rinv[u * 2 + 1] * Bm[r][1]) * Pm.a[r][c];
      K[(long)(u * 5 + c) * n + i] = acc;
    }
}

// ------------------------------------------------------------------------------------
// Reference trajectories of the legacy simulations (src/d2d/trajectory.py, src/d2d/trajectory_factory.py) sampled on the
// device: every trajectory is a CompositeTraj of up to D2D_TRAJ_MAX_SEG segments (a plain trajectory = one segment), each a
// line, a circle arc, a slalom or a degree-7 polynomial pair, described by D2D_TRAJ_SEG_STRIDE doubles (include/d2d.h).
// One thread per (sample time, trajectory); output Yref [T][6][n] = x, y, xd, yd, xdd, ydd -- the input of d2d_sim_dfff_run.
__global__ void __launch_bounds__(256)
traj_sample_kernel(int n, int T, double t_start, double dt, const double *__restrict__ desc, double *__restrict__ Y) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)n * T) return;
  const int it = i / n, d = i - (long)it * n;
  const double t = t_start + it * dt;
  const double *td = desc + (size_t)d * D2D_TRAJ_STRIDE;
  const int nseg = (int)td[0];
  const double T0 = td[1], dur = td[2];
  // CompositeTraj.get (src/d2d/trajectory.py:202-208): lapse = fmod(t - t0, duration), first segment whose end is beyond it
  double lapse = t;
  const double *sg = td + 4;
  if (nseg > 1 || td[3] != 0.0) {
    lapse = fmod(t - T0, dur);
    int k = 0;
    for (; k < nseg - 1; ++k)
      if (sg[k * D2D_TRAJ_SEG_STRIDE + 2] > lapse) break;
    sg += k * D2D_TRAJ_SEG_STRIDE;
  }
  const int type = (int)sg[0];
  const double ts = lapse - sg[1];                       // time since the segment's own t0
  const double *p = sg + 3;
  double y[6] = {0, 0, 0, 0, 0, 0};
  if (type == D2D_TRAJ_LINE || type == D2D_TRAJ_SLALOM) {
    // TrajectoryLine.get (:132-141): p1 + un v (t - t0); slalom (trajectory_factory.py:120-136): + a sin(om (t - t0 + phi)) on y
    y[0] = p[0] + p[2] * p[4] * ts; y[1] = p[1] + p[3] * p[4] * ts;
    y[2] = p[2] * p[4]; y[3] = p[3] * p[4];
    if (type == D2D_TRAJ_SLALOM) {
      const double a = p[5], om = p[6], al = om * (ts + p[7]);
      double sn, cs;
      sincos(al, &sn, &cs);
      y[1] += a * sn; y[3] += a * om * cs; y[5] += -a * om * om * sn;
    }
  } else if (type == D2D_TRAJ_CIRCLE) {
    // TrajectoryCircle.get (:153-160): c + r (cos, sin)(omega (t - t0) + alpha0)
    const double r = p[2], om = p[3], al = ts * om + p[4];
    double sn, cs;
    sincos(al, &sn, &cs);
    y[0] = p[0] + r * cs; y[1] = p[1] + r * sn;
    y[2] = -om * r * sn; y[3] = om * r * cs;
    y[4] = -om * om * r * cs; y[5] = -om * om * r * sn;
  } else if (type == D2D_TRAJ_POLY) {
    // MinSnapPoly / PolynomialOne.get (:74-82, :166-187): Horner on coefs[0,:] and its derivative rows
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
      const double *c = p + 8 * ax;
      double v0 = c[7], v1 = 7.0 * c[7], v2 = 42.0 * c[7];
#pragma unroll
      for (int j = 6; j >= 0; --j) {
        v0 = v0 * ts + c[j];
        if (j >= 1) v1 = v1 * ts + j * c[j];
        if (j >= 2) v2 = v2 * ts + j * (j - 1) * c[j];
      }
      y[ax] = v0; y[2 + ax] = v1; y[4 + ax] = v2;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) Y[((size_t)it * 6 + c) * n + d] = y[c];
}

// ------------------------------------------------------------------------------------
static int upload_Bz(d2d_ctx *ctx, int n_ac, const double *Bmat, const double *z_des);

extern "C" {

int d2d_step(d2d_ctx *ctx, int n, const double *X, const double *U, double wx, double wy,
             double tau_phi, double tau_v, double dt, double *Xout) {
  D2D_REQUIRE(ctx && X && U && Xout, "d2d_step: null argument");
  D2D_REQUIRE(n > 0 && tau_phi > 0 && tau_v > 0 && dt > 0, "d2d_step: n, tau_phi, tau_v, dt must be > 0");
  const GlMesh mesh = make_mesh(dt, tau_phi, tau_v);
  hipLaunchKernelGGL(step_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, X, U, wx, wy, mesh, Xout);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_sim_gvf_run(d2d_ctx *ctx, const d2d_gvf_params *p, const double *X0,
                    const double *centres, const double *radius, const double *Bmat,
                    const double *z_des, const double *X0f, double *X_hist, double *U_hist,
                    double *Rr_hist, double *eth_hist, double *X_final, int32_t *stop_row, int32_t *conv_row) {
  D2D_REQUIRE(ctx && p && X0 && centres && radius && X_final, "d2d_sim_gvf_run: null argument");
  D2D_REQUIRE(p->use_stop >= 0 && p->use_stop <= 2 && p->stop_hold >= 0, "d2d_sim_gvf_run: use_stop in 0..2, stop_hold >= 0");
  D2D_REQUIRE(p->n_ac >= 1 && p->n_ac <= 64, "d2d_sim_gvf_run: n_ac=%d not in 1..64", p->n_ac);
  D2D_REQUIRE(p->n_form >= 1 && p->n_rows >= 1 && p->rec_stride >= 1, "d2d_sim_gvf_run: n_form, n_rows, rec_stride must be >= 1");
  D2D_REQUIRE(p->n_ac == 1 || (Bmat && z_des), "d2d_sim_gvf_run: Bmat / z_des missing");
  D2D_REQUIRE(p->use_stop != 1 || X0f, "d2d_sim_gvf_run: use_stop = 1 needs X0f");
  D2D_REQUIRE(p->use_stop != 2 || p->n_ac >= 2, "d2d_sim_gvf_run: the phase-error rule needs at least two aircraft");
  D2D_REQUIRE(p->dt > 0 && p->tau_phi > 0 && p->tau_v > 0, "d2d_sim_gvf_run: dt, tau_phi, tau_v must be > 0");
  const int n_ac = p->n_ac, nm = n_ac - 1;
  const size_t nb = (size_t)n_ac * nm + nm;
  if (int rc = upload_Bz(ctx, n_ac, Bmat, z_des)) return rc;
  // 64-thread blocks when that loses no lanes: more workgroups for the 256 CUs
  const int threads = (64 % n_ac == 0) ? 64 : 256;
  const int fpb = threads / n_ac;
  const int blocks = (p->n_form + fpb - 1) / fpb;
  const size_t lds = (512 + 128 + nb + 8) * sizeof(double);
  const GlMesh mesh = make_mesh(p->dt, p->tau_phi, p->tau_v);
#define GVF_LAUNCH(KERNEL)                                                                                             \
  hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(threads), lds, ctx->stream, *p, mesh, fpb, X0, centres, radius, ctx->Bmat_dev, \
                     X0f, X_hist, U_hist, Rr_hist, eth_hist, X_final, stop_row, conv_row)
  if (n_ac == 1 || n_ac == 2 || n_ac == 4) {
    // formations inside a DPP quad: no LDS exchange, no barrier in the step; a launch of at most one wave per SIMD gets the
    // instantiation that may use the whole register file
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const bool wide = blocks <= 4 * n_cu;
    if (n_ac == 4) { if (wide) GVF_LAUNCH(gvf_run_quad_wide_kernel<4>); else GVF_LAUNCH(gvf_run_quad_kernel<4>); }
    else if (n_ac == 2) { if (wide) GVF_LAUNCH(gvf_run_quad_wide_kernel<2>); else GVF_LAUNCH(gvf_run_quad_kernel<2>); }
    else { if (wide) GVF_LAUNCH(gvf_run_quad_wide_kernel<1>); else GVF_LAUNCH(gvf_run_quad_kernel<1>); }
  } else {
    GVF_LAUNCH(gvf_run_kernel);
  }
#undef GVF_LAUNCH
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

static int check_track(const d2d_track_params *p, const char *who) {
  D2D_REQUIRE(p->n >= 1, "%s: n must be >= 1", who);
  D2D_REQUIRE(p->dt > 0 && p->tau_phi > 0 && p->tau_v > 0, "%s: dt, tau_phi, tau_v must be > 0", who);
  D2D_REQUIRE(p->r_diag[0] > 0 && p->r_diag[1] > 0, "%s: R must be positive", who);
  return D2D_OK;
}

int d2d_ctrl_gain(d2d_ctx *ctx, const d2d_track_params *p, const double *X, const double *Yref,
                  double *Xr, double *dX, double *U, double *Kgain) {
  D2D_REQUIRE(ctx && p && X && Yref, "d2d_ctrl_gain: null argument");
  if (int rc = check_track(p, "d2d_ctrl_gain")) return rc;
  hipLaunchKernelGGL(ctrl_gain_kernel, dim3((p->n + 63) / 64), dim3(64), 0, ctx->stream, *p, X, Yref, Xr, dX, U, Kgain);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_dfff_eval(d2d_ctx *ctx, const d2d_track_params *p, const double *X, const double *Yref,
                  double *Xr, double *U, double *Kgain) {
  D2D_REQUIRE(ctx && p && X && Yref, "d2d_dfff_eval: null argument");
  if (int rc = check_track(p, "d2d_dfff_eval")) return rc;
  hipLaunchKernelGGL(dfff_kernel, dim3((p->n + 63) / 64), dim3(64), 0, ctx->stream, *p, X, Yref, Xr, U, Kgain);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_sim_dfff_run(d2d_ctx *ctx, const d2d_track_params *p, const double *Yref, const double *perts,
                     const double *X0, double *X_hist, double *U_hist, double *Xr_hist, double *X_final) {
  D2D_REQUIRE(ctx && p && Yref && X0, "d2d_sim_dfff_run: null argument");
  if (int rc = check_track(p, "d2d_sim_dfff_run")) return rc;
  const GlMesh mesh = make_mesh(p->dt, p->tau_phi, p->tau_v);
  hipLaunchKernelGGL(dfff_run_kernel, dim3((p->n + 63) / 64), dim3(64), 0, ctx->stream, *p, mesh, Yref, perts, X0, X_hist,
                     U_hist, Xr_hist, X_final);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_sim_track_run(d2d_ctx *ctx, const d2d_track_params *p, const double *x_ref,
                      const double *y_ref, const double *X0, double *X_hist, double *U_hist,
                      double *Xr_hist, double *dX_hist, double *Yd_hist, double *Ydd_hist,
                      double *X_final) {
  D2D_REQUIRE(ctx && p && x_ref && y_ref && X0, "d2d_sim_track_run: null argument");
  if (int rc = check_track(p, "d2d_sim_track_run")) return rc;
  D2D_REQUIRE(p->n_rows >= 3, "d2d_sim_track_run: n_rows must be >= 3 (second-order edge differences)");
  const size_t plane = (size_t)p->n_rows * p->n * sizeof(double);
  double *deriv = nullptr;
  D2D_CHECK_HIP(hipMallocAsync(reinterpret_cast<void **>(&deriv), 4 * plane, ctx->stream));
  double *xd = deriv, *yd = deriv + (size_t)p->n_rows * p->n, *xdd = yd + (size_t)p->n_rows * p->n,
         *ydd = xdd + (size_t)p->n_rows * p->n;
  const dim3 g((p->n + 255) / 256, p->n_rows), b(256);
  const double inv_dt = 1.0 / p->dt;
  hipLaunchKernelGGL(gradient_kernel, g, b, 0, ctx->stream, p->n_rows, p->n, inv_dt, x_ref, xd);
  hipLaunchKernelGGL(gradient_kernel, g, b, 0, ctx->stream, p->n_rows, p->n, inv_dt, y_ref, yd);
  hipLaunchKernelGGL(gradient_kernel, g, b, 0, ctx->stream, p->n_rows, p->n, inv_dt, xd, xdd);
  hipLaunchKernelGGL(gradient_kernel, g, b, 0, ctx->stream, p->n_rows, p->n, inv_dt, yd, ydd);
  D2D_LAUNCH_CHECK();
  const GlMesh mesh = make_mesh(p->dt, p->tau_phi, p->tau_v);
  hipLaunchKernelGGL(track_run_kernel, dim3((p->n + 63) / 64), dim3(64), 0, ctx->stream, *p, mesh, x_ref, y_ref, xd,
                     yd, xdd, ydd, X0, X_hist, U_hist, Xr_hist, dX_hist, Yd_hist, Ydd_hist, X_final);
  D2D_LAUNCH_CHECK();
  D2D_CHECK_HIP(hipFreeAsync(deriv, ctx->stream));
  return D2D_OK;
}

int d2d_traj_sample(d2d_ctx *ctx, int n, int T, double t_start, double dt, const double *desc, double *Yref) {
  D2D_REQUIRE(ctx && desc && Yref, "d2d_traj_sample: null argument");
  D2D_REQUIRE(n >= 1 && T >= 1, "d2d_traj_sample: n, T must be >= 1");
  const long tot = (long)n * T;
  hipLaunchKernelGGL(traj_sample_kernel, dim3((tot + 255) / 256), dim3(256), 0, ctx->stream, n, T, t_start, dt, desc, Yref);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

static int upload_Bz(d2d_ctx *ctx, int n_ac, const double *Bmat, const double *z_des) {
  const int nm = n_ac - 1;
  const size_t nb = (size_t)n_ac * nm + nm;
  if (ctx->Bmat_cap < nb + 1) {
    if (ctx->Bmat_dev) D2D_CHECK_HIP(hipFree(ctx->Bmat_dev));
    D2D_CHECK_HIP(hipMalloc(&ctx->Bmat_dev, (nb + 1) * sizeof(double)));
    ctx->Bmat_cap = nb + 1;
  }
  if (nb) {
    std::vector<double> h(nb);
    for (size_t i = 0; i < (size_t)n_ac * nm; ++i) h[i] = Bmat[i];
    for (int i = 0; i < nm; ++i) h[(size_t)n_ac * nm + i] = z_des[i];
    D2D_CHECK_HIP(hipMemcpyAsync(ctx->Bmat_dev, h.data(), nb * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // h goes out of scope
  }
  return D2D_OK;
}

int d2d_dcf_eval(d2d_ctx *ctx, int n_form, int n_ac, const double *Bmat, const double *z_des, double kr,
                 const double *centres, const double *pos, double *U_r, double *eth_deg) {
  D2D_REQUIRE(ctx && centres && pos && U_r, "d2d_dcf_eval: null argument");
  D2D_REQUIRE(n_ac >= 1 && n_ac <= 64 && n_form >= 1, "d2d_dcf_eval: n_ac=%d (1..64), n_form=%d", n_ac, n_form);
  D2D_REQUIRE(n_ac == 1 || (Bmat && z_des), "d2d_dcf_eval: Bmat / z_des missing");
  if (int rc = upload_Bz(ctx, n_ac, Bmat, z_des)) return rc;
  hipLaunchKernelGGL(dcf_eval_kernel, dim3((n_form + 255) / 256), dim3(256), 0, ctx->stream, n_form, n_ac, kr,
                     ctx->Bmat_dev, centres, pos, U_r, eth_deg);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_gvf_eval(d2d_ctx *ctx, int n, const double *X, const double *e, const double *nvec, const double *H,
                 double ke, double kd, double *U) {
  D2D_REQUIRE(ctx && X && e && nvec && H && U, "d2d_gvf_eval: null argument");
  D2D_REQUIRE(n >= 1, "d2d_gvf_eval: n must be >= 1");
  hipLaunchKernelGGL(gvf_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ke, kd, X, e, nvec, H, U);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_flatness(d2d_ctx *ctx, int variant, int n, const double *Yref, double wx, double wy, double tau_phi,
                 double tau_v, double *X, double *U, double *Xdot) {
  D2D_REQUIRE(ctx && Yref && X && U, "d2d_flatness: null argument");
  D2D_REQUIRE(n >= 1 && (variant == 0 || variant == 1), "d2d_flatness: n=%d, variant=%d", n, variant);
  hipLaunchKernelGGL(flatness_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, variant, n, wx, wy, tau_phi,
                     tau_v, Yref, X, U, Xdot);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_cont_jac(d2d_ctx *ctx, int n, const double *Xr, double tau_phi, double tau_v, double *A, double *B) {
  D2D_REQUIRE(ctx && Xr && A && B, "d2d_cont_jac: null argument");
  D2D_REQUIRE(n >= 1 && tau_phi > 0 && tau_v > 0, "d2d_cont_jac: n, tau_phi, tau_v must be > 0");
  hipLaunchKernelGGL(cont_jac_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, tau_phi, tau_v, Xr, A, B);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

int d2d_lqr(d2d_ctx *ctx, int n, const double *A, const double *B, const double *Q, const double *R, double *K,
            double *P) {
  D2D_REQUIRE(ctx && A && B && Q && R && K, "d2d_lqr: null argument");
  D2D_REQUIRE(n >= 1, "d2d_lqr: n must be >= 1");
  const double det = R[0] * R[3] - R[1] * R[2];
  D2D_REQUIRE(det != 0.0, "d2d_lqr: R is singular");
  LqrWeights w;
  for (int i = 0; i < 25; ++i) w.Q[i] = Q[i];
  w.Rinv[0] = R[3] / det; w.Rinv[1] = -R[1] / det; w.Rinv[2] = -R[2] / det; w.Rinv[3] = R[0] / det;
  hipLaunchKernelGGL(lqr_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, w, A, B, K, P);
  D2D_LAUNCH_CHECK();
  return D2D_OK;
}

}  // extern "C"
