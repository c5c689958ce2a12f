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
inline bool cholesky(std::vector<double>& A, int n) {
  // Works on U = L^T (row j of U = column j of L, contiguous): every element still receives its terms
  // s -= L[i][k] * L[j][k] for k = 0 .. j-1 in that order -- bit for bit the row-by-row dot-product form -- but the
  // inner loop runs over i (independent elements, unit stride), which the compiler vectorises; a dot product is a
  // serial chain it may not reorder.  32 clusters at D = 64 are one M-step of the headline configuration.
  std::vector<double> U((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) U[(size_t)j * n + i] = A[(size_t)i * n + j];
  for (int j = 0; j < n; ++j) {
    double* uj = U.data() + (size_t)j * n;
    for (int k = 0; k < j; ++k) {
      const double* uk = U.data() + (size_t)k * n;
      const double ukj = uk[j];
      for (int i = j; i < n; ++i) uj[i] -= uk[i] * ukj;
    }
    if (!(uj[j] > 0.0)) return false;
    const double ljj = std::sqrt(uj[j]);
    uj[j] = ljj;
    for (int i = j + 1; i < n; ++i) uj[i] /= ljj;
  }
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = U[(size_t)j * n + i];
    for (int j = i + 1; j < n; ++j) A[(size_t)i * n + j] = 0.0;
  }
  return true;
}

// Inverse of a lower-triangular matrix (row-major), result lower-triangular.  Row i accumulates
// L[i][k] * (row k of the inverse) for k = 0 .. i-1 (unit stride, vectorisable); element j receives its terms for
// k = j .. i-1 in increasing k, exactly as the element-by-element sum does.
inline std::vector<double> tril_inverse(const std::vector<double>& L, int n) {
  std::vector<double> Li((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    double* ri = Li.data() + (size_t)i * n;
    for (int k = 0; k < i; ++k) {
      const double lik = L[(size_t)i * n + k];
      const double* rk = Li.data() + (size_t)k * n;
      for (int j = 0; j <= k; ++j) ri[j] += lik * rk[j];
    }
    const double lii = L[(size_t)i * n + i];
    for (int j = 0; j < i; ++j) ri[j] = -ri[j] / lii;
    ri[i] = 1.0 / lii;
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
