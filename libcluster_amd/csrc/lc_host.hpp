// Host-side arithmetic of the M-step and free energy: special functions,
// small dense linear algebra, and the distribution state machines.  No Eigen,
// no Boost (neither exists in the target image); plain row-major doubles.
//
// Restates (not copies) the reference:
//   WeightState    <- Dirichlet / StickBreak / GDirichlet  src/distributions.cpp:83-266
//   GaussWishState <- GaussWish                            src/distributions.cpp:273-399
//   eigpower / logdet / digamma / lgamma                   src/probutils.cpp:153-230
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace lch {

// --- constants: include/libcluster.h:122-127 (float literals!), distributions.h:39-43
constexpr double PRIORVAL = 1.0;
constexpr unsigned SPLITITER = 15;
constexpr double CONVERGE = 1e-5f;
constexpr double FENGYDEL = CONVERGE / 10;
constexpr double ZEROCUTOFF = 0.1f;
constexpr double BETAPRIOR = 1.0, ALPHA1PRIOR = 1.0, ALPHA2PRIOR = 1.0;
constexpr double EIGCONTHRESH = 1.0e-8f;  // src/probutils.cpp:39
constexpr int MAXITER = 100;              // src/probutils.cpp:40
constexpr double PI = 3.14159265358979323846264338327950288;

// ---------------------------------------------------------------------------
// special functions (Boost.Math in the reference)
// ---------------------------------------------------------------------------
// digamma for x > 0: upward recurrence to x >= 10, then the asymptotic series
// (error < 1e-16 relative at x = 10 with 7 Bernoulli terms).  Near the root
// x0 = 1.4616... the absolute error stays ~1e-16.
inline double digamma(double x) {
  if (x != x) return x;  // NaN in, NaN out (boost::math::digamma does the same; no recursion on NaN)
  if (!(x > 0.0)) {
    // poles at 0, -1, -2, ...: boost's default policy throws std::domain_error there (probutils.cpp:213)
    if (x == std::floor(x)) throw std::domain_error("digamma: evaluation at a pole");
    // reflection for completeness (never reached by the algorithms: all arguments are > 0)
    return digamma(1.0 - x) - PI / std::tan(PI * x);
  }
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double xi = 1.0 / x, x2 = xi * xi;
  // sum B_2n / (2n x^2n): 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12
  const double s =
      x2 * (1.0 / 12 - x2 * (1.0 / 120 - x2 * (1.0 / 252 - x2 * (1.0 / 240 - x2 * (1.0 / 132 - x2 * (691.0 / 32760 - x2 / 12))))));
  return r + std::log(x) - 0.5 * xi - s;
}

inline double lgam(double x) {
  int sign = 0;
  return ::lgamma_r(x, &sign);
}

// ---------------------------------------------------------------------------
// dense helpers (row-major n x n)
// ---------------------------------------------------------------------------
// In-place lower Cholesky A = L L^T (only the lower triangle is read/written,
// the strict upper triangle is zeroed).  Returns false if not positive definite.
// four doubles side by side (lowered to two SSE2 operations on a baseline x86-64, one with AVX2: the two helpers below
// are compiled for both and picked at load time; neither target has fused multiply-add, so both give the same bits)
typedef double lc_v4d __attribute__((vector_size(32), aligned(8)));
// (hipcc passes every translation unit through a device compilation as well, where function multiversioning does not
//  exist: the attribute is for the host pass only)
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define LC_HOST_NO_CLONES 1  // (an ifunc resolver runs before the ThreadSanitizer runtime is up: tools/sanitize_host.sh tsan)
#endif
#endif
#if (defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__) || defined(LC_HOST_NO_CLONES)
#define LC_HOST_CLONES
#else
#define LC_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#endif
// acc[jj][:] -= li[k][:] * lj[k][jj] for k = 0 .. nk - 1 in that order (li, lj: 4 doubles per k)
LC_HOST_CLONES inline void chol_block_update(const double* li, const double* lj, int nk,
                                                                              double (*acc)[4]) {
  lc_v4d a0, a1, a2, a3;
  __builtin_memcpy(&a0, acc[0], 32);
  __builtin_memcpy(&a1, acc[1], 32);
  __builtin_memcpy(&a2, acc[2], 32);
  __builtin_memcpy(&a3, acc[3], 32);
  for (int k = 0; k < nk; ++k) {
    lc_v4d b;
    __builtin_memcpy(&b, li + (size_t)k * 4, 32);
    const double* a = lj + (size_t)k * 4;
    a0 -= b * a[0];
    a1 -= b * a[1];
    a2 -= b * a[2];
    a3 -= b * a[3];
  }
  __builtin_memcpy(acc[0], &a0, 32);
  __builtin_memcpy(acc[1], &a1, 32);
  __builtin_memcpy(acc[2], &a2, 32);
  __builtin_memcpy(acc[3], &a3, 32);
}
// Is |M|_2^2 < tau for the lower-triangular M (row-major n x n)?  PROVED, not estimated: tau I - M^T M is factorised as
// L D L^T without pivoting and every pivot has to stay above 1e-9 tau (rounding is 1e-13 tau at n = 128) -- which holds iff
// the matrix is positive definite, i.e. iff every singular value of M is below sqrt(tau).  n^3 / 3 multiply-adds in
// contiguous saxpy loops.  Used by Context::recompute_bounded: the power method proposes tau from below, this decides.
// (the cloned body touches plain arrays only: hipcc's host pass leaves the members of a std::vector that a multiversioned
//  function instantiates undefined in the object)
LC_HOST_CLONES inline bool norm_certified_raw(const double* M, int n, double tau, double* G, double* col) {
  const size_t N = (size_t)n;
  for (int i = 0; i < n; ++i) G[(size_t)i * N + i] = tau;
  for (int l = 0; l < n; ++l) {  // G -= (row l of M)^T (row l of M), lower half
    const double* r = M + (size_t)l * N;
    for (int i = 0; i <= l; ++i) {
      const double a = r[i];
      double* g = G + (size_t)i * N;
      for (int j = 0; j <= i; ++j) g[j] -= a * r[j];
    }
  }
  const double floor_ = 1e-9 * tau;
  for (int k = 0; k < n; ++k) {
    const double d = G[(size_t)k * N + k];
    if (!(d > floor_)) return false;  // (NaN: false)
    for (int i = k + 1; i < n; ++i) col[i] = G[(size_t)i * N + k];
    for (int i = k + 1; i < n; ++i) {
      const double f = col[i] / d;
      double* g = G + (size_t)i * N;
      for (int j = k + 1; j <= i; ++j) g[j] -= f * col[j];
    }
  }
  return true;
}
inline bool norm_certified(const double* M, int n, double tau, std::vector<double>& work) {
  if (!(tau > 0.0) || !std::isfinite(tau)) return false;
  const size_t N = (size_t)n;
  work.assign(N * N + N, 0.0);
  return norm_certified_raw(M, n, tau, work.data(), work.data() + N * N);
}
// r[ii][j] += l[ii][0] rk[0][j] + ... + l[ii][3] rk[3][j] (added in that order) for j = 0 .. nj - 1
LC_HOST_CLONES inline void trinv_block_update(double* const* r, const double* const* rk,
                                                                               const double (*l)[4], int nj) {
  int j = 0;
  for (; j + 4 <= nj; j += 4) {
    lc_v4d v0, v1, v2, v3;
    __builtin_memcpy(&v0, rk[0] + j, 32);
    __builtin_memcpy(&v1, rk[1] + j, 32);
    __builtin_memcpy(&v2, rk[2] + j, 32);
    __builtin_memcpy(&v3, rk[3] + j, 32);
    for (int ii = 0; ii < 4; ++ii) {
      lc_v4d s;
      __builtin_memcpy(&s, r[ii] + j, 32);
      s += v0 * l[ii][0];
      s += v1 * l[ii][1];
      s += v2 * l[ii][2];
      s += v3 * l[ii][3];
      __builtin_memcpy(r[ii] + j, &s, 32);
    }
  }
  for (; j < nj; ++j) {
    const double v0 = rk[0][j], v1 = rk[1][j], v2 = rk[2][j], v3 = rk[3][j];
    for (int ii = 0; ii < 4; ++ii) {
      double s = r[ii][j];
      s += l[ii][0] * v0;
      s += l[ii][1] * v1;
      s += l[ii][2] * v2;
      s += l[ii][3] * v3;
      r[ii][j] = s;
    }
  }
}

inline bool cholesky(std::vector<double>& A, int n) {
  // Every element receives its terms  s -= L[i][k] * L[j][k]  for k = 0 .. j-1 in that order, a multiplication and a
  // subtraction each -- bit for bit the row-by-row dot-product form (a dot product is a serial chain the compiler may not
  // reorder).  What is free is WHICH elements advance together: 4 x 4 blocks take the terms of the columns in front of
  // their block side by side -- sixteen independent chains per two 32-byte loads -- out of a copy of the factor that keeps
  // four rows interleaved, Lp[panel][k][row in panel], so that both operand streams of a block are contiguous in k
  // (round 5: the one-element-row form made three memory operations per multiply-subtract and left L1 at D = 128;
  // 64 clusters at D = 128 were 8 ms of a 143 ms iteration on a two-core host, 16 clusters at D = 256 15 of 61).
  constexpr int B = 4;
  const int P = (n + B - 1) / B;
  std::vector<double> Lp((size_t)P * n * B, 0.0);  // Lp[(p n + k) B + ii] = L[B p + ii][k]
  std::vector<double> T((size_t)P * B * B);        // the current block column: T[(p B + jj) B + ii] = element (B p + ii, j0 + jj)
  for (int jb = 0; jb < P; ++jb) {
    const int j0 = jb * B, jw = n - j0 < B ? n - j0 : B;
    const double* lj = Lp.data() + (size_t)jb * n * B;
    for (int ib = jb; ib < P; ++ib) {
      const int i0 = ib * B;
      double acc[B][B];
      for (int jj = 0; jj < B; ++jj)
        for (int ii = 0; ii < B; ++ii)
          acc[jj][ii] = (i0 + ii < n && jj < jw) ? A[(size_t)(i0 + ii) * n + j0 + jj] : 0.0;
      chol_block_update(Lp.data() + (size_t)ib * n * B, lj, j0, acc);
      for (int jj = 0; jj < B; ++jj)
        for (int ii = 0; ii < B; ++ii) T[((size_t)ib * B + jj) * B + ii] = acc[jj][ii];
    }
    // the block's own columns, one after the other: their terms k = j0 .. j - 1 follow the ones above in order
    for (int jj = 0; jj < jw; ++jj) {
      const int j = j0 + jj;
      for (int ib = jb; ib < P; ++ib) {
        double* t = T.data() + ((size_t)ib * B + jj) * B;
        const double* li = Lp.data() + (size_t)ib * n * B;
        for (int kk = 0; kk < jj; ++kk) {
          const double ajk = lj[(size_t)(j0 + kk) * B + jj];
          for (int ii = 0; ii < B; ++ii) t[ii] -= li[(size_t)(j0 + kk) * B + ii] * ajk;
        }
      }
      const double sjj = T[((size_t)jb * B + jj) * B + jj];
      if (!(sjj > 0.0)) return false;
      const double ljj = std::sqrt(sjj);
      for (int ib = jb; ib < P; ++ib) {
        const double* t = T.data() + ((size_t)ib * B + jj) * B;
        double* li = Lp.data() + ((size_t)ib * n + j) * B;
        for (int ii = 0; ii < B; ++ii) {
          const int i = ib * B + ii;
          li[ii] = i > j && i < n ? t[ii] / ljj : 0.0;
        }
      }
      Lp[((size_t)jb * n + j) * B + jj] = ljj;
    }
  }
  for (int i = 0; i < n; ++i) {
    const double* li = Lp.data() + (size_t)(i / B) * n * B + (i % B);
    for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = li[(size_t)j * B];
    for (int j = i + 1; j < n; ++j) A[(size_t)i * n + j] = 0.0;
  }
  return true;
}

// Inverse of a lower-triangular matrix (row-major), result lower-triangular.  Row i accumulates
// L[i][k] * (row k of the inverse) for k = 0 .. i-1; element j receives its terms for k = j .. i-1 in increasing k,
// exactly as the element-by-element sum does.  Four rows i and four rows k at a time: a piece of the four target rows
// stays in registers while the four source rows pass over it in order (sixteen multiply-adds per five loads and a store,
// where the row-by-row form made one per three memory operations).
inline std::vector<double> tril_inverse(const std::vector<double>& L, int n) {
  std::vector<double> Li((size_t)n * n, 0.0);
  constexpr int B = 4;
  for (int i0 = 0; i0 < n; i0 += B) {
    const int iw = n - i0 < B ? n - i0 : B;
    if (iw == B) {
      double* r[B] = {Li.data() + (size_t)i0 * n, Li.data() + (size_t)(i0 + 1) * n, Li.data() + (size_t)(i0 + 2) * n,
                      Li.data() + (size_t)(i0 + 3) * n};
      int k0 = 0;
      for (; k0 + B <= i0; k0 += B) {
        double l[B][B];  // l[ii][kk] = L[i0 + ii][k0 + kk]
        for (int ii = 0; ii < B; ++ii)
          for (int kk = 0; kk < B; ++kk) l[ii][kk] = L[(size_t)(i0 + ii) * n + k0 + kk];
        const double* rk[B] = {Li.data() + (size_t)k0 * n, Li.data() + (size_t)(k0 + 1) * n, Li.data() + (size_t)(k0 + 2) * n,
                               Li.data() + (size_t)(k0 + 3) * n};
        // columns j <= k0: all four source rows reach them, in the order k0, k0 + 1, k0 + 2, k0 + 3
        trinv_block_update(r, rk, l, k0 + 1);
        // columns k0 < j <= k0 + 3: source row k0 + kk reaches column j only for j <= k0 + kk
        for (int j = k0 + 1; j < k0 + B; ++j)
          for (int ii = 0; ii < B; ++ii) {
            double s = r[ii][j];
            for (int kk = j - k0; kk < B; ++kk) s += l[ii][kk] * rk[kk][j];
            r[ii][j] = s;
          }
      }
      for (; k0 < i0; ++k0)  // (never taken: i0 is a multiple of B)
        for (int ii = 0; ii < B; ++ii) {
          const double lik = L[(size_t)(i0 + ii) * n + k0];
          const double* rkk = Li.data() + (size_t)k0 * n;
          for (int j = 0; j <= k0; ++j) r[ii][j] += lik * rkk[j];
        }
    } else {
      for (int ii = 0; ii < iw; ++ii) {
        double* ri = Li.data() + (size_t)(i0 + ii) * n;
        for (int k = 0; k < i0; ++k) {
          const double lik = L[(size_t)(i0 + ii) * n + k];
          const double* rk = Li.data() + (size_t)k * n;
          for (int j = 0; j <= k; ++j) ri[j] += lik * rk[j];
        }
      }
    }
    // the block's own rows, one after the other
    for (int ii = 0; ii < iw; ++ii) {
      const int i = i0 + ii;
      double* ri = Li.data() + (size_t)i * n;
      for (int k = i0; k < i; ++k) {
        const double lik = L[(size_t)i * n + k];
        const double* rk = Li.data() + (size_t)k * n;
        for (int j = 0; j <= k; ++j) ri[j] += lik * rk[j];
      }
      const double lii = L[(size_t)i * n + i];
      for (int j = 0; j < i; ++j) ri[j] = -ri[j] / lii;
      ri[i] = 1.0 / lii;
    }
  }
  return Li;
}

// ln det(A) -- src/probutils.cpp:189-202 (throws domain_error like the reference)
inline double logdet(const std::vector<double>& A, int n) {
  std::vector<double> L(A);
  if (!cholesky(L, n)) throw std::domain_error("Matrix A is not positive definite.");
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += std::log(L[(size_t)i * n + i]);
  return 2.0 * s;
}

// Principal eigenvector by the power method -- src/probutils.cpp:153-186
inline double eigpower(const std::vector<double>& A, int n, std::vector<double>& eigvec) {
  if (n == 1) {
    eigvec.assign(1, 1.0);
    return A[0];
  }
  std::vector<double> v(n), o(n);
  for (int i = 0; i < n; ++i) v[i] = -1.0 + 2.0 * i / (n - 1);  // LinSpaced(n,-1,1)
  auto norm = [&](const std::vector<double>& x) {
    double s = 0.0;
    for (double t : x) s += t * t;
    return std::sqrt(s);
  };
  double eigval = norm(v);
  eigvec.resize(n);
  for (int i = 0; i < n; ++i) eigvec[i] = v[i] / eigval;
  double vdist = std::numeric_limits<double>::infinity();
  for (int it = 0; vdist > EIGCONTHRESH && it < MAXITER; ++it) {
    o = eigvec;
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += A[(size_t)i * n + j] * o[j];
      v[i] = s;
    }
    eigval = norm(v);
    double d = 0.0;
    for (int i = 0; i < n; ++i) {
      eigvec[i] = v[i] / eigval;
      const double t = eigvec[i] - o[i];
      d += t * t;
    }
    vdist = std::sqrt(d);
  }
  return eigval;
}

// ---------------------------------------------------------------------------
// weight distributions
// ---------------------------------------------------------------------------
enum WeightKind { W_DIRICHLET = 0, W_STICKBREAK = 1, W_GDIRICHLET = 2 };

struct WeightState {
  int kind = W_DIRICHLET;
  double a1p = ALPHA1PRIOR, a2p = ALPHA2PRIOR, Fp = 0.0;
  std::vector<double> Nk, alpha1, alpha2, Elogv, Elognv, Elogpi;
  std::vector<std::pair<int, double>> ordvec;

  WeightState() : WeightState(W_DIRICHLET, ALPHA1PRIOR) {}
  WeightState(int kind_, double prior) : kind(kind_), a1p(prior) {
    if (!(prior > 0.0))
      throw std::invalid_argument(kind == W_DIRICHLET ? "Alpha prior must be > 0!"
                                                      : "Concentration parameter has to be > 0!");
    Nk.assign(1, 0.0);  // WeightDist(), distributions.h:94
    alpha1.assign(1, a1p);
    alpha2.assign(1, a2p);
    Elogv.assign(1, 0.0);
    Elognv.assign(1, 0.0);
    Elogpi.assign(1, 0.0);
    ordvec.assign(1, std::make_pair(0, 0.0));
    Fp = lgam(a1p) + lgam(a2p) - lgam(a1p + a2p);  // priorfcalc, distributions.cpp:116-121
  }

  // distributions.cpp:242-256 (Dirichlet), 124-168 (StickBreak), 186-196 (GDirichlet)
  void update(const double* nk, int K) {
    Nk.assign(nk, nk + K);
    alpha1.resize(K);
    Elogpi.resize(K);
    for (int k = 0; k < K; ++k) alpha1[k] = a1p + nk[k];
    if (kind == W_DIRICHLET) {
      double asum = 0.0;
      for (int k = 0; k < K; ++k) asum += alpha1[k];
      const double psisum = digamma(asum);
      for (int k = 0; k < K; ++k) Elogpi[k] = digamma(alpha1[k]) - psisum;
      return;
    }
    alpha2.resize(K);
    Elogv.resize(K);
    Elognv.resize(K);
    ordvec.resize(K);
    for (int k = 0; k < K; ++k) ordvec[k] = std::make_pair(k, nk[k]);
    // same comparator and (unstable) std::sort as the reference, :141-146
    std::sort(ordvec.begin(), ordvec.end(),
              [](const std::pair<int, double>& i, const std::pair<int, double>& j) { return i.second > j.second; });
    double N = 0.0;
    for (int k = 0; k < K; ++k) N += nk[k];
    double cumNk = 0.0, cumE = 0.0;
    for (int idx = 0; idx < K; ++idx) {
      const int k = ordvec[idx].first;
      cumNk += nk[k];
      alpha2[k] = a2p + (N - cumNk);
      const double psisum = digamma(alpha1[k] + alpha2[k]);
      Elogv[k] = digamma(alpha1[k]) - psisum;
      Elognv[k] = digamma(alpha2[k]) - psisum;
      Elogpi[k] = Elogv[k] + cumE;
      cumE += Elognv[k];
    }
    if (kind == W_GDIRICHLET) {
      const int smallk = ordvec.back().first;
      Elogpi[smallk] = Elogpi[smallk] - Elogv[smallk];
      Elogv[smallk] = 0.0;
      Elognv[smallk] = 0.0;
    }
  }

  // distributions.cpp:259-266, 171-179, 199-215
  double fenergy() const {
    const int K = (int)alpha1.size();
    if (kind == W_DIRICHLET) {
      double asum = 0.0, esum = 0.0, t = 0.0;
      for (int k = 0; k < K; ++k) {
        asum += alpha1[k];
        esum += Elogpi[k];
        t += (alpha1[k] - 1) * Elogpi[k] - lgam(alpha1[k]);
      }
      return lgam(asum) - (a1p - 1) * esum + t - lgam(K * a1p) + K * lgam(a1p);
    }
    auto term = [&](int k) {
      return lgam(alpha1[k] + alpha2[k]) - lgam(alpha1[k]) - lgam(alpha2[k]) + (alpha1[k] - a1p) * Elogv[k] +
             (alpha2[k] - a2p) * Elognv[k];
    };
    if (kind == W_STICKBREAK) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += term(k);
      return K * Fp + s;
    }
    const int Ko = (int)ordvec.size();
    double s = 0.0;
    for (int idx = 0; idx < Ko - 1; ++idx) s += term(ordvec[idx].first);
    return (Ko - 1) * Fp + s;
  }
};

// ---------------------------------------------------------------------------
// Gauss-Wishart cluster distribution
// ---------------------------------------------------------------------------
struct GaussWishState {
  int D = 0;
  double prior = 0.0, N = 0.0;
  double nu_p = 0, beta_p = BETAPRIOR, logdW_p = 0, F_p = 0;
  std::vector<double> m_p, iW_p;
  double nu = 0, beta = 0, logdW = 0;
  std::vector<double> m, iW;
  double N_s = 0;
  std::vector<double> x_s, xx_s;
  // cached inverse Cholesky factor of iW (iW^-1 = Li^T Li); one factorisation
  // per M-step serves logdW, the E-step whitener and fenergy
  mutable std::vector<double> Li;
  mutable bool factored = false;

  GaussWishState() {}
  GaussWishState(double clustwidth, int D_) : D(D_), prior(clustwidth), nu_p(D_) {
    if (!(clustwidth > 0.0)) throw std::invalid_argument("clustwidth must be > 0!");
    if (D < 1) throw std::invalid_argument("D must be >= 1!");
    m_p.assign(D, 0.0);
    iW_p.assign((size_t)D * D, 0.0);
    for (int d = 0; d < D; ++d) iW_p[(size_t)d * D + d] = nu_p * prior;  // distributions.cpp:286
    logdW_p = -logdet(iW_p, D);
    F_p = 0.0;
    for (int d = 1; d <= D; ++d) F_p += lgam((nu_p + 1 - d) / 2);  // :294
    clearobs();
  }

  void clearobs() {  // :340-353
    nu = nu_p;
    beta = beta_p;
    m = m_p;
    iW = iW_p;
    logdW = logdW_p;
    factored = false;
    N_s = 0.0;
    x_s.assign(D, 0.0);
    xx_s.assign((size_t)D * D, 0.0);
  }

  // equals one or more addobs() calls (:301-313) whose sums were formed on the device
  void addstats(double Ns, const double* xs, const double* xxs) {
    N_s += Ns;
    for (int d = 0; d < D; ++d) x_s[d] += xs[d];
    for (size_t i = 0; i < (size_t)D * D; ++i) xx_s[i] += xxs[i];
  }

  void update() {  // :316-337
    std::vector<double> xk(D, 0.0);
    if (N_s > 0)
      for (int d = 0; d < D; ++d) xk[d] = x_s[d] / N_s;
    N = N_s;
    nu = nu_p + N;
    beta = beta_p + N;
    for (int d = 0; d < D; ++d) m[d] = (beta_p * m_p[d] + x_s[d]) / beta;
    const double f = beta_p * N / beta;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        const double Sk = xx_s[(size_t)i * D + j] - xk[i] * x_s[j];
        iW[(size_t)i * D + j] = iW_p[(size_t)i * D + j] + Sk + f * (xk[i] - m_p[i]) * (xk[j] - m_p[j]);
      }
    // logdW = -logdet(iW) (:334); a non-PD iW surfaces as runtime_error like the reference (:335-336)
    factored = false;
    std::vector<double> L(iW);
    if (!cholesky(L, D)) throw std::runtime_error("Calc log(det(W)): Matrix A is not positive definite.");
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += std::log(L[(size_t)i * D + i]);
    logdW = -2.0 * s;
    Li = tril_inverse(L, D);
    factored = true;
  }

  const std::vector<double>& inv_factor() const {
    if (!factored) {
      std::vector<double> L(iW);
      if (!cholesky(L, D)) throw std::invalid_argument("Matrix A is not positive definite");  // probutils.cpp:131-132
      Li = tril_inverse(L, D);
      factored = true;
    }
    return Li;
  }

  double sumpsi() const {
    double s = 0.0;
    for (int d = 1; d <= D; ++d) s += digamma((nu + 1 - d) / 2);
    return s;
  }

  // constant part of Eloglike (:360-364): 0.5*(sumpsi + logdW - D(1/beta + ln pi))
  double eloglike_const() const { return 0.5 * (sumpsi() + logdW - D * (1.0 / beta + std::log(PI))); }

  // A = sqrt(nu) * chol(iW)^-1 (lower, row-major D x D): nu * maha(x) = ||A (x-m)||^2
  std::vector<double> whitener() const {
    std::vector<double> A(inv_factor());
    const double s = std::sqrt(nu);
    for (double& v : A) v *= s;
    return A;
  }

  double fenergy() const {  // :388-399
    const std::vector<double>& Li = inv_factor();
    // tr(iW^-1 iW_p) = sum_ij (Li^T Li)_ij iW_p_ji ; iW^-1 = Li^T Li
    double tr = 0.0;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        const double p = iW_p[(size_t)j * D + i];
        if (p == 0.0) continue;
        double w = 0.0;
        for (int k = std::max(i, j); k < D; ++k) w += Li[(size_t)k * D + i] * Li[(size_t)k * D + j];
        tr += w * p;
      }
    double maha = 0.0;  // (m - m_p) iW^-1 (m - m_p)^T = ||Li (m - m_p)||^2
    for (int i = 0; i < D; ++i) {
      double y = 0.0;
      for (int j = 0; j <= i; ++j) y += Li[(size_t)i * D + j] * (m[j] - m_p[j]);
      maha += y * y;
    }
    double sp = 0.0, sl = 0.0;
    for (int d = 1; d <= D; ++d) {
      sp += digamma((nu + 1 - d) / 2);
      sl += lgam((nu + 1 - d) / 2);
    }
    return F_p +
           (D * (beta_p / beta - 1 - nu - std::log(beta_p / beta)) + nu * (tr + beta_p * maha) +
            nu_p * (logdW_p - logdW) + N * sp) /
               2 -
           sl;
  }

  // splitobs (:373-385) on host rows (row-major n x D with row stride ld)
  void splitobs(const double* X, int64_t n, int64_t ld, std::vector<unsigned char>& out) const {
    std::vector<double> v;
    eigpower(iW, D, v);
    out.resize((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
      double s = 0.0;
      for (int d = 0; d < D; ++d) s += (X[r * ld + d] - m[d]) * v[d];
      out[(size_t)r] = s >= 0.0 ? 1 : 0;
    }
  }

  std::vector<double> getcov() const {  // distributions.h:311
    std::vector<double> c(iW);
    for (double& v : c) v /= nu;
    return c;
  }
};

// ---------------------------------------------------------------------------
// Diagonal Gaussian (Normal-Gamma) and Exponential (Gamma) cluster distributions
// src/distributions.cpp:406-517 and :524-589.  Their E-step is
//   log q~[n,k] = c_jk + sum_d ( w2_kd (x_nd - a_kd)^2 + w1_kd x_nd )
// with (a, w2, w1) = (m, -nu/(2L), 0) for NormGamma and (0, 0, -a*ib) for ExpGamma.
// ---------------------------------------------------------------------------
constexpr double NUPRIOR = 1.0, APRIOR = 1.0;  // include/distributions.h:40,43

struct NormGammaState {
  int D = 0;
  double prior = 0.0, N = 0.0;
  double nu_p = NUPRIOR, beta_p = BETAPRIOR, logL_p = 0.0;
  std::vector<double> m_p, L_p;
  double nu = 0, beta = 0, logL = 0;
  std::vector<double> m, L;
  double N_s = 0;
  std::vector<double> x_s, xx_s;

  NormGammaState() {}
  NormGammaState(double clustwidth, int D_) : D(D_), prior(clustwidth) {
    if (!(clustwidth > 0.0)) throw std::invalid_argument("clustwidth must be > 0!");
    if (D < 1) throw std::invalid_argument("D must be >= 1!");
    m_p.assign(D, 0.0);
    L_p.assign(D, nu_p * prior);  // :419
    logL_p = D * std::log(nu_p * prior);
    clearobs();
  }
  void clearobs() {  // :467-480
    nu = nu_p;
    beta = beta_p;
    m = m_p;
    L = L_p;
    logL = logL_p;
    N_s = 0.0;
    x_s.assign(D, 0.0);
    xx_s.assign(D, 0.0);
  }
  void addstats(double Ns, const double* xs, const double* xxs) {  // sums of :426-438
    N_s += Ns;
    for (int d = 0; d < D; ++d) {
      x_s[d] += xs[d];
      xx_s[d] += xxs[d];
    }
  }
  void update() {  // :441-464
    N = N_s;
    beta = beta_p + N;
    nu = nu_p + N / 2;
    logL = 0.0;
    for (int d = 0; d < D; ++d) {
      double xk = 0.0, Sk = 0.0;
      if (N_s > 0) {
        xk = x_s[d] / N_s;
        Sk = xx_s[d] - x_s[d] * x_s[d] / N_s;
      }
      m[d] = (beta_p * m_p[d] + x_s[d]) / beta;
      L[d] = L_p[d] + Sk / 2 + (beta_p * N / (2 * beta)) * (xk - m_p[d]) * (xk - m_p[d]);
      if (!(L[d] > 0.0)) throw std::invalid_argument("Calc log(L): Variance is zero or less!");
      logL += std::log(L[d]);
    }
  }
  // constant part of Eloglike (:490-491)
  double eloglike_const() const { return 0.5 * (D * (digamma(nu) - std::log(2 * PI) - 1.0 / beta) - logL); }
  double fenergy() const {  // :508-517; D/2 is an INTEGER division in the reference (unsigned D)
    double t1 = 0.0, t2 = 0.0;
    for (int d = 0; d < D; ++d) {
      t1 += (m[d] - m_p[d]) * (m[d] - m_p[d]) / L[d];
      t2 += L_p[d] / L[d];
    }
    return D * (lgam(nu_p) - lgam(nu) + N * digamma(nu) / 2 - nu) +
           (D / 2) * (std::log(beta) - std::log(beta_p) - 1 + beta_p / beta) + beta_p * nu / 2 * t1 +
           nu_p * (logL - logL_p) + nu * t2;
  }
  int split_axis() const {  // :499-501 (first maximum, like Eigen's maxCoeff)
    int ax = 0;
    for (int d = 1; d < D; ++d)
      if (L[d] > L[ax]) ax = d;
    return ax;
  }
};

struct ExpGammaState {
  int D = 0;
  double prior = 0.0, N = 0.0;
  double a_p = APRIOR, b_p = 0.0;
  double a = 0, logb = 0;
  std::vector<double> ib;
  double N_s = 0;
  std::vector<double> x_s;

  ExpGammaState() {}
  ExpGammaState(double obsmag, int D_) : D(D_), prior(obsmag), b_p(obsmag) {  // :524-530 (no argument check there)
    if (D < 1) throw std::invalid_argument("D must be >= 1!");
    clearobs();
  }
  void clearobs() {  // :555-565
    a = a_p;
    ib.assign(D, 1.0 / b_p);
    logb = D * std::log(b_p);
    N_s = 0.0;
    x_s.assign(D, 0.0);
  }
  void addstats(double Ns, const double* xs, const double*) {  // sums of :533-542
    N_s += Ns;
    for (int d = 0; d < D; ++d) x_s[d] += xs[d];
  }
  void update() {  // :545-552
    N = N_s;
    a = a_p + N;
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
      ib[d] = 1.0 / (b_p + x_s[d]);
      s += std::log(ib[d]);
    }
    logb = -s;
  }
  double eloglike_const() const { return D * digamma(a) - logb; }  // :570
  double fenergy() const {  // :584-589
    double sib = 0.0;
    for (int d = 0; d < D; ++d) sib += ib[d];
    return D * ((a - a_p) * digamma(a) - a - a_p * std::log(b_p) - lgam(a) + lgam(a_p)) + b_p * a * sib + a_p * logb;
  }
};

// One cluster of any family: what the (family-agnostic) driver in lc_engine.cpp holds.  The reference
// gets the same effect from its <W, C> templates (cluster.cpp:177, 564).
enum ClusterKind { C_GAUSSWISH = 0, C_NORMGAMMA = 1, C_EXPGAMMA = 2 };

struct ClusterAny {
  int kind = C_GAUSSWISH;
  GaussWishState gw;
  NormGammaState ng;
  ExpGammaState eg;

  ClusterAny() {}
  ClusterAny(int kind_, double prior, int D) : kind(kind_) {
    if (kind == C_GAUSSWISH) gw = GaussWishState(prior, D);
    else if (kind == C_NORMGAMMA) ng = NormGammaState(prior, D);
    else if (kind == C_EXPGAMMA) eg = ExpGammaState(prior, D);
    else throw std::invalid_argument("unknown cluster family");
  }
  int D() const { return kind == C_GAUSSWISH ? gw.D : kind == C_NORMGAMMA ? ng.D : eg.D; }
  double N() const { return kind == C_GAUSSWISH ? gw.N : kind == C_NORMGAMMA ? ng.N : eg.N; }
  double prior() const { return kind == C_GAUSSWISH ? gw.prior : kind == C_NORMGAMMA ? ng.prior : eg.prior; }
  void clearobs() {
    if (kind == C_GAUSSWISH) gw.clearobs();
    else if (kind == C_NORMGAMMA) ng.clearobs();
    else eg.clearobs();
  }
  void addstats(double Ns, const double* xs, const double* xxs) {
    if (kind == C_GAUSSWISH) gw.addstats(Ns, xs, xxs);
    else if (kind == C_NORMGAMMA) ng.addstats(Ns, xs, xxs);
    else eg.addstats(Ns, xs, xxs);
  }
  void update() {
    if (kind == C_GAUSSWISH) gw.update();
    else if (kind == C_NORMGAMMA) ng.update();
    else eg.update();
  }
  double fenergy() const { return kind == C_GAUSSWISH ? gw.fenergy() : kind == C_NORMGAMMA ? ng.fenergy() : eg.fenergy(); }
  double eloglike_const() const {
    return kind == C_GAUSSWISH ? gw.eloglike_const() : kind == C_NORMGAMMA ? ng.eloglike_const() : eg.eloglike_const();
  }
  // second-moment statistics per cluster: D*D (full), D (diagonal) or 0
  static size_t xx_size(int kind, int D) { return kind == C_GAUSSWISH ? (size_t)D * D : kind == C_NORMGAMMA ? (size_t)D : 0; }
};

}  // namespace lch
