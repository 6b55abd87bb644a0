// Host-side (fp64) construction of the shared basis block of the polynomial fit.
//
// A C^3 piecewise degree-7 polynomial with S segments is parameterised by its knot data
// (pos, vel, acc, jerk at the S+1 knots) exactly as the reference builds trajectories:
// CompositeTraj([MinSnapPoly(Y_j, Y_{j+1}, T)]) with PolynomialOne's closed form
// (src/d2d/trajectory.py:47-72, :166-208).  (pos, vel) at the first and last knot are the
// end conditions; the other 4S knot values per axis are the free unknowns, whitened by the
// Cholesky factor of Nfree^T Mref Nfree.
#include <cmath>
#include <cstring>

#include "fit_plan.h"

namespace {

double arr(int k, int n) {   // n!/(n-k)!   src/d2d/trajectory.py:41-45
  double a = 1;
  for (int i = n; i > n - k; --i) a *= i;
  return a;
}

struct Dense {
  int r, c;
  std::vector<double> a;
  Dense(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

Dense matmul(const Dense &A, const Dense &B, bool tA = false) {
  const int m = tA ? A.c : A.r, k = tA ? A.r : A.c, n = B.c;
  Dense C(m, n);
  for (int i = 0; i < m; ++i)
    for (int l = 0; l < k; ++l) {
      const double v = tA ? A(l, i) : A(i, l);
      if (v == 0.0) continue;
      for (int j = 0; j < n; ++j) C(i, j) += v * B(l, j);
    }
  return C;
}

// in-place Cholesky (lower); returns false if not positive definite
bool cholesky(Dense &A) {
  const int n = A.r;
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k);
      A(i, j) = s / d;
    }
    for (int i = 0; i < j; ++i) A(i, j) = 0.0;
  }
  return true;
}

// general inverse by Gauss-Jordan with partial pivoting (4x4 only here)
bool invert(Dense &A) {
  const int n = A.r;
  Dense I(n, n);
  for (int i = 0; i < n; ++i) I(i, i) = 1.0;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(A(r, c)) > std::fabs(A(piv, c))) piv = r;
    if (A(piv, c) == 0.0) return false;
    for (int j = 0; j < n; ++j) {
      std::swap(A(c, j), A(piv, j));
      std::swap(I(c, j), I(piv, j));
    }
    const double ip = 1.0 / A(c, c);
    for (int j = 0; j < n; ++j) { A(c, j) *= ip; I(c, j) *= ip; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = A(r, c);
      for (int j = 0; j < n; ++j) { A(r, j) -= f * A(c, j); I(r, j) -= f * I(c, j); }
    }
  }
  A = I;
  return true;
}

}  // namespace

int fit_basis_build(d2d_fit_plan *pl) {
  const int S = pl->S, K = pl->K, nz = 8 * S, nk = 4 * (S + 1), nq = 4 * S;
  const double T = pl->duration / S;
  pl->T = T;
  pl->nq = nq;
  // --- segment lookup of the K samples (CompositeTraj.get, src/d2d/trajectory.py:202-208)
  std::vector<int> seg(K);
  std::vector<double> tau(K);
  {
    std::vector<double> ends(S);
    double acc = 0;
    for (int s = 0; s < S; ++s) { acc += T; ends[s] = acc; }
    for (int k = 0; k < K; ++k) {
      // numpy.linspace(0, duration, K): start + k*step, last sample forced to `duration`
      const double step = pl->duration / (K - 1);
      const double t = (k == K - 1) ? pl->duration : k * step;
      int s = S - 1;
      for (int j = 0; j < S; ++j)
        if (ends[j] > t) { s = j; break; }
      seg[k] = s;
      tau[k] = t - s * T;
    }
  }
  pl->seg = seg;
  pl->tau = tau;
  // --- Phi_d (K x 8S), d = 0..2
  std::vector<Dense> Phi;
  for (int d = 0; d < 3; ++d) {
    Dense P(K, nz);
    for (int k = 0; k < K; ++k)
      for (int p = d; p < 8; ++p) P(k, 8 * seg[k] + p) = arr(d, p) * std::pow(tau[k], p - d);
    Phi.push_back(P);
  }
  // --- Hermite map of one segment (PolynomialOne.__init__, src/d2d/trajectory.py:54-66)
  Dense M3(4, 4), M4(4, 4);
  for (int i = 0; i < 4; ++i) {
    for (int j = i; j < 4; ++j) M3(i, j) = arr(i, j) * std::pow(T, j - i);
    for (int j = 0; j < 4; ++j) M4(i, j) = arr(i, j + 4) * std::pow(T, j - i + 4);
  }
  if (!invert(M4)) { d2d_set_error("fit_basis_build: singular Hermite block"); return D2D_EINVAL; }
  Dense H(8, 8);
  for (int i = 0; i < 4; ++i) H(i, i) = 1.0 / arr(i, i);
  {
    Dense M1i(4, 4);
    for (int i = 0; i < 4; ++i) M1i(i, i) = 1.0 / arr(i, i);
    Dense t = matmul(matmul(M4, M3), M1i);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { H(4 + i, j) = -t(i, j); H(4 + i, 4 + j) = M4(i, j); }
  }
  // --- knot map N (8S x 4(S+1)), split into fixed / free columns
  Dense N(nz, nk);
  for (int s = 0; s < S; ++s)
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) N(8 * s + i, 4 * s + j) = H(i, j);
  const int fixed[4] = {0, 1, 4 * S, 4 * S + 1};
  std::vector<int> freec;
  for (int i = 0; i < nk; ++i)
    if (i != fixed[0] && i != fixed[1] && i != fixed[2] && i != fixed[3]) freec.push_back(i);
  Dense Nf(nz, nq), Nx(nz, 4);
  for (int i = 0; i < nz; ++i) {
    for (int j = 0; j < nq; ++j) Nf(i, j) = N(i, freec[j]);
    for (int j = 0; j < 4; ++j) Nx(i, j) = N(i, fixed[j]);
  }
  // --- whitening metric and its Cholesky factor
  Dense M(nz, nz);
  for (int d = 0; d < 3; ++d) {
    Dense PtP = matmul(Phi[d], Phi[d], true);
    for (size_t i = 0; i < M.a.size(); ++i) M.a[i] += pl->wref[d] * PtP.a[i];
  }
  Dense MNf = matmul(M, Nf);
  Dense Gram = matmul(Nf, MNf, true);
  for (int i = 0; i < nq; ++i)
    for (int j = i + 1; j < nq; ++j) { const double s = 0.5 * (Gram(i, j) + Gram(j, i)); Gram(i, j) = s; Gram(j, i) = s; }
  if (!cholesky(Gram)) { d2d_set_error("fit_basis_build: whitening metric not positive definite (wref, K, S?)"); return D2D_EINVAL; }
  pl->Lw = Gram.a;                                        // lower Cholesky factor L of Nf^T Mref Nf (fit_basis_knots)
  {
    // Pe = -(L L^T)^-1 Nf^T M Nx: the free knot data of q = 0 per unit end datum (Zp = Nx + Nf Pe)
    Dense R = matmul(Nf, matmul(M, Nx), true);            // nq x 4
    pl->Pe.assign((size_t)nq * 4, 0.0);
    std::vector<double> y(nq);
    for (int c = 0; c < 4; ++c) {
      for (int i = 0; i < nq; ++i) {
        double v = -R(i, c);
        for (int j = 0; j < i; ++j) v -= Gram(i, j) * y[j];
        y[i] = v / Gram(i, i);
      }
      for (int i = nq - 1; i >= 0; --i) {
        double v = y[i];
        for (int j = i + 1; j < nq; ++j) v -= Gram(j, i) * pl->Pe[(size_t)j * 4 + c];
        pl->Pe[(size_t)i * 4 + c] = v / Gram(i, i);
      }
    }
  }
  // Z = Nf L^-T  <=>  Z L^T = Nf : forward substitution over columns
  Dense Z(nz, nq);
  for (int i = 0; i < nz; ++i)
    for (int j = 0; j < nq; ++j) {
      double s = Nf(i, j);
      for (int k = 0; k < j; ++k) s -= Z(i, k) * Gram(j, k);
      Z(i, j) = s / Gram(j, j);
    }
  // Zp = Nx - Z (Z^T M Nx)
  Dense ZtMNx = matmul(Z, matmul(M, Nx), true);
  Dense Zp = Nx;
  {
    Dense c = matmul(Z, ZtMNx);
    for (size_t i = 0; i < Zp.a.size(); ++i) Zp.a[i] -= c.a[i];
  }
  pl->Z = Z.a;
  pl->Zp = Zp.a;
  pl->G.assign((size_t)3 * K * nq, 0.0);
  pl->Gp.assign((size_t)3 * K * 4, 0.0);
  for (int d = 0; d < 3; ++d) {
    Dense g = matmul(Phi[d], Z), gp = matmul(Phi[d], Zp);
    std::memcpy(&pl->G[(size_t)d * K * nq], g.a.data(), sizeof(double) * K * nq);
    std::memcpy(&pl->Gp[(size_t)d * K * 4], gp.a.data(), sizeof(double) * K * 4);
  }
  // Pinit = (G0^T G0)^-1 G0^T
  Dense G0(K, nq);
  std::memcpy(G0.a.data(), pl->G.data(), sizeof(double) * K * nq);
  Dense A = matmul(G0, G0, true);
  pl->G0tG0 = A.a;
  Dense L = A;
  if (!cholesky(L)) { d2d_set_error("fit_basis_build: G0^T G0 singular (K too small for S?)"); return D2D_EINVAL; }
  pl->Pinit.assign((size_t)nq * K, 0.0);
  std::vector<double> y(nq);
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < nq; ++i) {
      double s = G0(k, i);
      for (int j = 0; j < i; ++j) s -= L(i, j) * y[j];
      y[i] = s / L(i, i);
    }
    for (int i = nq - 1; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < nq; ++j) s -= L(j, i) * pl->Pinit[(size_t)j * K + k];
      pl->Pinit[(size_t)i * K + k] = s / L(i, i);
    }
  }
  return D2D_OK;
}

// The segment formulation of the long-horizon kernel (fit_seg.h): on segment s the polynomial of one axis is
//   sum_p z_p tau^p = sum_i l_i P_i(x),  x = 2 tau / T - 1,  l = Zl_s q_axis + Zlp_s d_axis,
// P_i the Legendre polynomials.  With sigma = tau / T and P_i(2 sigma - 1) = sum_p C[i][p] sigma^p (C lower triangular, by the
// three-term recurrence), sigma^p = sum_i Cinv[p][i] P_i and Zl_s[i][j] = sum_p Cinv[p][i] T^p Z[8 s + p][j].
// Lanes of a wavefront are dealt to the segments in proportion to their sample counts (SegMap).
int fit_basis_segments(d2d_fit_plan *pl) {
  const int S = pl->S, K = pl->K, nq = pl->nq;
  const double T = pl->T;
  long double C[8][8] = {{0}}, Ci[8][8] = {{0}};
  C[0][0] = 1.0L; C[1][0] = -1.0L; C[1][1] = 2.0L;
  for (int n = 1; n < 7; ++n)                       // (n+1) P_{n+1} = (2n+1) (2 sigma - 1) P_n - n P_{n-1}
    for (int p = 0; p < 8; ++p) {
      long double v = -(2.0L * n + 1.0L) * C[n][p] - (long double)n * C[n - 1][p];
      if (p > 0) v += (2.0L * n + 1.0L) * 2.0L * C[n][p - 1];
      C[n + 1][p] = v / (n + 1.0L);
    }
  // sigma^p = sum_i Ci[p][i] P_i: row p of Ci solves Ci[p][.] C = e_p (C lower triangular: back substitution from i = p)
  for (int p = 0; p < 8; ++p)
    for (int i = p; i >= 0; --i) {
      long double v = (i == p) ? 1.0L : 0.0L;
      for (int m = i + 1; m <= p; ++m) v -= Ci[p][m] * C[m][i];
      Ci[p][i] = v / C[i][i];
    }
  pl->Zl.assign((size_t)8 * S * nq, 0.0);
  pl->Zlp.assign((size_t)8 * S * 4, 0.0);
  for (int s = 0; s < S; ++s)
    for (int i = 0; i < 8; ++i) {
      for (int j = 0; j < nq; ++j) {
        long double v = 0;
        for (int p = i; p < 8; ++p) v += Ci[p][i] * powl((long double)T, p) * (long double)pl->Z[(size_t)(8 * s + p) * nq + j];
        pl->Zl[(size_t)(8 * s + i) * nq + j] = (double)v;
      }
      for (int j = 0; j < 4; ++j) {
        long double v = 0;
        for (int p = i; p < 8; ++p) v += Ci[p][i] * powl((long double)T, p) * (long double)pl->Zp[(size_t)(8 * s + p) * 4 + j];
        pl->Zlp[(size_t)(8 * s + i) * 4 + j] = (double)v;
      }
    }
  pl->sx.resize(K);
  for (int k = 0; k < K; ++k) pl->sx[k] = 2.0 * pl->tau[k] / T - 1.0;
  // samples of a segment are consecutive
  pl->seg_S = S;
  for (int s = 0; s < S; ++s) { pl->seg_Ks[s] = 0; pl->seg_k0[s] = 0; }
  for (int k = K - 1; k >= 0; --k) { pl->seg_Ks[pl->seg[k]]++; pl->seg_k0[pl->seg[k]] = k; }
  for (int k = 1; k < K; ++k)
    if (pl->seg[k] < pl->seg[k - 1]) { d2d_set_error("fit_basis_segments: samples not in segment order"); return D2D_EINVAL; }
  int L[D2D_FIT_MAX_S];
  int used = 0;
  for (int s = 0; s < S; ++s) { L[s] = pl->seg_Ks[s] > 0 ? 1 : 0; used += L[s]; }
  auto chunks = [&](int s) { return L[s] ? (pl->seg_Ks[s] + L[s] - 1) / L[s] : 0; };
  for (; used < 64; ++used) {                       // the next lane goes to the segment that needs the most chunks
    int best = -1;
    for (int s = 0; s < S; ++s)
      if (L[s] && L[s] < pl->seg_Ks[s] &&
          (best < 0 || chunks(s) > chunks(best) || (chunks(s) == chunks(best) && pl->seg_Ks[s] * L[best] > pl->seg_Ks[best] * L[s])))
        best = s;
    if (best < 0) break;
    L[best]++;
  }
  pl->seg_nchunk = 0;
  pl->seg_l0[0] = 0;
  for (int s = 0; s < S; ++s) {
    pl->seg_l0[s + 1] = pl->seg_l0[s] + L[s];
    if (chunks(s) > pl->seg_nchunk) pl->seg_nchunk = chunks(s);
  }
  for (int s = S; s < D2D_FIT_MAX_S; ++s) pl->seg_l0[s + 1] = pl->seg_l0[S];
  return D2D_OK;
}

// The knot-space statement of the fit (oracle/fit_knot.py; kernel: fit_knot.hip): unknowns = the Taylor-scaled knot data
//   u[8 j + 4 a + k] = T^k / k! * Y_a^(k)(t_j),  j = 0..S (knots), a = axis, k = 0..3    ((j in {0, S}, k < 2): the end conditions)
// in which a sample touches the 16 entries of its segment's two knots only: J^T J is block tridiagonal in 8 x 8 blocks.
// q = B (u_free - u0) per axis with B = L^T diag(1 / dsc), u0 = Pu e; the reference metric Mu = B^T B is banded.
int fit_basis_knots(d2d_fit_plan *pl) {
  const int S = pl->S, K = pl->K, nq = pl->nq, NE = 8 * (S + 1);
  const double T = pl->T;
  if (nq != 4 * S || (int)pl->Lw.size() != nq * nq) { d2d_set_error("fit_basis_knots: plan without the whitening factor"); return D2D_EINVAL; }
  auto &kn = pl->kn;
  kn.NE = NE;
  double dsc[4] = {1.0, T, T * T / 2.0, T * T * T / 6.0};
  // free knot-data columns of one axis (fit_basis_build: everything but (pos, vel) at the first and last knot), in knot order
  std::vector<int> freec;
  for (int i = 0; i < 4 * (S + 1); ++i)
    if (i != 0 && i != 1 && i != 4 * S && i != 4 * S + 1) freec.push_back(i);
  // axis-vector index (28 = 4 (S+1) entries: knot data of one axis) of free index i: freec[i]; full entry of (axis a, axis index v): 8 (v / 4) + 4 a + v % 4
  const int NV = 4 * (S + 1);
  kn.NV = NV;
  std::vector<int> vfree(NV, -1);
  for (int i = 0; i < nq; ++i) vfree[freec[i]] = i;
  // Hermite basis of one segment in the scaled coordinates on x in [0, 1]: coefficients c = Hu v
  Dense A(8, 8);
  for (int k = 0; k < 4; ++k) {
    A(k, k) = 1.0;
    for (int p = k; p < 8; ++p) {
      double c = 1.0;                                       // C(p, k)
      for (int i = 0; i < k; ++i) c = c * (p - i) / (i + 1);
      A(4 + k, p) = c;
    }
  }
  if (!invert(A)) { d2d_set_error("fit_basis_knots: singular Hermite block"); return D2D_EINVAL; }
  kn.Hb64.assign((size_t)K * KN_HB_STRIDE, 0.0);
  kn.Hb32.assign((size_t)K * 32, 0.f);
  for (int k = 0; k < K; ++k) {
    const double x = pl->tau[k] / T;
    const int s = pl->seg[k];
    for (int d = 0; d < 3; ++d)
      for (int m = 0; m < 8; ++m) {
        double v = 0.0;
        for (int p = d; p < 8; ++p) v += arr(d, p) * std::pow(x, p - d) * A(p, m);
        v /= std::pow(T, d);
        kn.Hb64[(size_t)k * KN_HB_STRIDE + 4 * m + d] = v;
        // the MFMA operands leave out the columns of the end conditions (they are not unknowns)
        const bool fixed = (s == 0 && m < 2) || (s == S - 1 && m >= 4 && m < 6);
        kn.Hb32[(size_t)k * 32 + 4 * m + d] = fixed ? 0.f : (float)v;
      }
  }
  for (int s = 0; s <= S; ++s) kn.k0[s] = s < S ? pl->seg_k0[s] : K;
  for (int s = 0; s < S; ++s)
    if (pl->seg_Ks[s] == 0) kn.k0[s] = s + 1 < S ? pl->seg_k0[s + 1] : K;     // (an empty segment: empty range)
  for (int s = S + 1; s <= D2D_FIT_MAX_S; ++s) kn.k0[s] = K;
  // waypoint rows' constant J^T J: per segment sum_k Hb0_k^T Hb0_k on the (x, x) and (y, y) columns, in the MFMA accumulator layout
  kn.Wseg.assign((size_t)S * 4 * 64, 0.f);
  for (int s = 0; s < S; ++s)
    for (int r = 0; r < 4; ++r)
      for (int l = 0; l < 64; ++l) {
        const int R = 4 * (l >> 4) + r, C = l & 15;
        const int aR = (R >> 2) & 1, aC = (C >> 2) & 1, mR = 4 * (R >> 3) + (R & 3), mC = 4 * (C >> 3) + (C & 3);
        if (aR != aC) continue;
        double v = 0.0;
        for (int k = kn.k0[s]; k < kn.k0[s + 1]; ++k) v += (double)kn.Hb32[(size_t)k * 32 + 4 * mR] * (double)kn.Hb32[(size_t)k * 32 + 4 * mC];
        kn.Wseg[((size_t)s * 4 + r) * 64 + l] = (float)v;
      }
  // per axis: B = L^T / dsc, Binv = dsc L^-T, Mu = B^T B = L L^T / (dsc dsc), Mu^-1
  Dense L(nq, nq), Bm(nq, nq), Bi(nq, nq), Mu(nq, nq);
  for (int i = 0; i < nq; ++i)
    for (int j = 0; j < nq; ++j) L(i, j) = pl->Lw[(size_t)i * nq + j];
  std::vector<double> ds(nq);
  for (int i = 0; i < nq; ++i) ds[i] = dsc[freec[i] % 4];
  for (int j = 0; j < nq; ++j)
    for (int i = 0; i < nq; ++i) Bm(j, i) = L(i, j) / ds[i];
  Bi = Bm;
  if (!invert(Bi)) { d2d_set_error("fit_basis_knots: singular map"); return D2D_EINVAL; }
  Mu = matmul(Bm, Bm, true);
  Dense Mi = Mu;
  if (!invert(Mi)) { d2d_set_error("fit_basis_knots: singular metric"); return D2D_EINVAL; }
  // rows per full entry e = 8 j + 4 a + k (both axes carry the same numbers); zero rows / columns for the end conditions
  kn.Bq.assign((size_t)2 * nq * NV, 0.0);      // [q index (axis-major: a nq + jq)][NV]: q = sum_v Bq[.][v] (u - u0)[axis vector]
  kn.BiT.assign((size_t)2 * nq * NV, 0.0);     // [q index][NV]: (J^T r in q) = sum_v BiT[.][v] g_u[axis vector]   (g_q = Binv^T g_u)
  kn.Md32.assign((size_t)2 * nq * 2 * nq, 0.f);   // dense rows of Mu in the order of the free entries (knot-major, axes interleaved)
  kn.Mrow32.assign((size_t)NE * 12, 0.f);
  kn.Binv.assign((size_t)NE * nq, 0.0);        // [e][nq]: (u - u0)_e = sum_j Binv[e][j] q[a nq + j]
  kn.Minv.assign((size_t)NE * NV, 0.0);        // [e][NV]: (Mu^-1 g)_e = sum_v Minv[e][v] g[axis vector]
  kn.Mrow.assign((size_t)NE * 12, 0.0);        // [e][3 knots (j-1, j, j+1)][4]: the same-axis entries of row e of Mu
  kn.Pu.assign((size_t)NE * 4, 0.0);           // [e][4]: u0_e = Pu[e] . (pos0, vel0, pos1, vel1) of its axis
  kn.msc.assign((size_t)NE, 1.0);
  for (int a = 0; a < 2; ++a)
    for (int jq = 0; jq < nq; ++jq)
      for (int i = 0; i < nq; ++i) {
        kn.Bq[((size_t)a * nq + jq) * NV + freec[i]] = Bm(jq, i);
        kn.BiT[((size_t)a * nq + jq) * NV + freec[i]] = Bi(i, jq);
      }
  for (int e = 0; e < NE; ++e) {
    const int j = e >> 3, k = e & 3, v = 4 * j + k, i = vfree[v];
    if (i < 0) {                                           // an end condition: u_e = dsc_k * datum
      kn.Pu[(size_t)e * 4 + (j == 0 ? k : 2 + k)] = dsc[k];
      continue;
    }
    for (int c = 0; c < 4; ++c) kn.Pu[(size_t)e * 4 + c] = ds[i] * pl->Pe[(size_t)i * 4 + c];
    for (int jq = 0; jq < nq; ++jq) kn.Binv[(size_t)e * nq + jq] = Bi(i, jq);
    for (int i2 = 0; i2 < nq; ++i2) kn.Minv[(size_t)e * NV + freec[i2]] = Mi(i, i2);
    for (int dj = -1; dj <= 1; ++dj)
      for (int k2 = 0; k2 < 4; ++k2) {
        const int v2 = 4 * (j + dj) + k2;
        if (v2 < 0 || v2 >= NV || vfree[v2] < 0) continue;
        kn.Mrow[(size_t)e * 12 + 4 * (dj + 1) + k2] = Mu(i, vfree[v2]);
      }
    for (int i2 = 0; i2 < nq; ++i2)
      if (Mu(i, i2) != 0.0 && std::abs(freec[i2] / 4 - j) > 1 && std::fabs(Mu(i, i2)) > 1e-12 * std::sqrt(Mu(i, i) * Mu(i2, i2))) {
        d2d_set_error("fit_basis_knots: the metric couples knots that are not neighbours");
        return D2D_EINVAL;
      }
    kn.msc[e] = std::sqrt(Mu(i, i));
  }
  for (size_t t = 0; t < kn.Mrow.size(); ++t) kn.Mrow32[t] = (float)kn.Mrow[t];
  kn.Mi32.assign(kn.Minv.size(), 0.f);
  for (size_t t = 0; t < kn.Minv.size(); ++t) kn.Mi32[t] = (float)kn.Minv[t];
  // dense index of the free entries: full entries in ascending order without the eight end conditions
  std::vector<int> efree;
  for (int e = 0; e < NE; ++e)
    if (vfree[4 * (e >> 3) + (e & 3)] >= 0) efree.push_back(e);
  for (int i = 0; i < 2 * nq; ++i)
    for (int i2 = 0; i2 < 2 * nq; ++i2) {
      const int e = efree[i], e2 = efree[i2];
      if (((e >> 2) & 1) != ((e2 >> 2) & 1)) continue;
      kn.Md32[(size_t)i * 2 * nq + i2] = (float)Mu(vfree[4 * (e >> 3) + (e & 3)], vfree[4 * (e2 >> 3) + (e2 & 3)]);
    }
  return D2D_OK;
}
