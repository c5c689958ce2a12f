/* probutils.h -- the reference's public utility header (include/probutils.h:40-172, src/probutils.cpp) over the
 * C ABI, so that code written against libcluster's three public headers compiles unchanged.
 *
 * mahaldist() -- the one function here that sits on the hot path (GaussWish::Eloglike calls it,
 * distributions.cpp:367) -- runs on the GPU (lc_mahaldist: the E-step kernel with one cluster).  The other functions
 * are small O(N D^2) / O(D^3) host utilities the learners do not use per iteration; they are plain C++ here.
 * digamma is lc_digamma (the library's own, <= 5e-15 from boost's), lgamma the C library's.
 */
#ifndef PROBUTILS_H
#define PROBUTILS_H

#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>

#include "distributions.h"
#include "lc_matrix.h"
#include "libcluster_hip.h"

namespace probutils {

namespace detail {
/* lower Cholesky of a row-major copy; false if not positive definite */
inline bool chol(std::vector<double>& a, int n) {
  for (int j = 0; j < n; ++j) {
    double d = a[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    a[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
      a[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}
inline std::vector<double> rowmajor(const lcmat::MatrixXd& A) {
  std::vector<double> a((size_t)A.rows() * A.cols());
  for (std::ptrdiff_t i = 0; i < A.rows(); ++i)
    for (std::ptrdiff_t j = 0; j < A.cols(); ++j) a[(size_t)i * A.cols() + j] = A(i, j);
  return a;
}
}  // namespace detail

/* probutils.h:55, probutils.cpp:50-53 */
inline lcmat::RowVectorXd mean(const lcmat::MatrixXd& X) {
  lcmat::RowVectorXd m(X.cols());
  for (std::ptrdiff_t d = 0; d < X.cols(); ++d) {
    double s = 0.0;
    for (std::ptrdiff_t r = 0; r < X.rows(); ++r) s += X(r, d);
    m(d) = s / (double)X.rows();
  }
  return m;
}

/* probutils.h:65, probutils.cpp:56-74 */
inline lcmat::RowVectorXd mean(const std::vector<lcmat::MatrixXd>& X) {
  const std::ptrdiff_t D = X[0].cols();
  lcmat::RowVectorXd m(D);
  for (std::ptrdiff_t d = 0; d < D; ++d) m(d) = 0.0;
  std::ptrdiff_t N = 0;
  for (size_t j = 0; j < X.size(); ++j) {
    if (X[j].cols() != D) throw std::invalid_argument("X dimensions are inconsistent between groups!");
    for (std::ptrdiff_t d = 0; d < D; ++d)
      for (std::ptrdiff_t r = 0; r < X[j].rows(); ++r) m(d) += X[j](r, d);
    N += X[j].rows();
  }
  for (std::ptrdiff_t d = 0; d < D; ++d) m(d) /= (double)N;
  return m;
}

/* probutils.h:73, probutils.cpp:77-82 */
inline lcmat::RowVectorXd stdev(const lcmat::MatrixXd& X) {
  const lcmat::RowVectorXd mu = mean(X);
  lcmat::RowVectorXd s(X.cols());
  for (std::ptrdiff_t d = 0; d < X.cols(); ++d) {
    double v = 0.0;
    for (std::ptrdiff_t r = 0; r < X.rows(); ++r) v += (X(r, d) - mu(d)) * (X(r, d) - mu(d));
    s(d) = std::sqrt(v / (double)(X.rows() - 1));
  }
  return s;
}

/* probutils.h:87, probutils.cpp:85-93 */
inline lcmat::MatrixXd cov(const lcmat::MatrixXd& X) {
  if (X.rows() <= 1) throw std::invalid_argument("Insufficient no. of observations.");
  const lcmat::RowVectorXd mu = mean(X);
  const std::ptrdiff_t D = X.cols();
  lcmat::MatrixXd c(D, D);
  for (std::ptrdiff_t a = 0; a < D; ++a)
    for (std::ptrdiff_t b = 0; b <= a; ++b) {
      double s = 0.0;
      for (std::ptrdiff_t r = 0; r < X.rows(); ++r) s += (X(r, a) - mu(a)) * (X(r, b) - mu(b));
      c(a, b) = c(b, a) = s / (double)(X.rows() - 1);
    }
  return c;
}

/* probutils.h:102, probutils.cpp:96-116 */
inline lcmat::MatrixXd cov(const std::vector<lcmat::MatrixXd>& X) {
  const std::ptrdiff_t D = X[0].cols();
  const lcmat::RowVectorXd mu = mean(X);
  lcmat::MatrixXd c(D, D);
  for (std::ptrdiff_t a = 0; a < D; ++a)
    for (std::ptrdiff_t b = 0; b < D; ++b) c(a, b) = 0.0;
  std::ptrdiff_t N = 0;
  for (size_t j = 0; j < X.size(); ++j) {
    if (X[j].rows() <= 1) throw std::invalid_argument("Insufficient no. of observations.");
    N += X[j].rows();
    for (std::ptrdiff_t a = 0; a < D; ++a)
      for (std::ptrdiff_t b = 0; b <= a; ++b) {
        double s = 0.0;
        for (std::ptrdiff_t r = 0; r < X[j].rows(); ++r) s += (X[j](r, a) - mu(a)) * (X[j](r, b) - mu(b));
        c(a, b) += s;
        if (a != b) c(b, a) += s;
      }
  }
  for (std::ptrdiff_t a = 0; a < D; ++a)
    for (std::ptrdiff_t b = 0; b < D; ++b) c(a, b) /= (double)(N - 1);
  return c;
}

/* probutils.h:115-119, probutils.cpp:119-141 -- on the GPU */
inline lcmat::VectorXd mahaldist(const lcmat::MatrixXd& X, const lcmat::RowVectorXd& mu, const lcmat::MatrixXd& A) {
  if (X.cols() != mu.cols() || X.cols() != A.cols())
    throw std::invalid_argument("Arguments do not have the same dimensionality");
  if (A.rows() != A.cols()) throw std::invalid_argument("Matrix A must be square!");
  lcmat::VectorXd out(X.rows());
  if (X.rows() == 0) return out;
  distributions::detail::CtxGuard g;
  distributions::detail::upload(g.c, X);
  const std::vector<double> a = detail::rowmajor(A);
  std::vector<double> m((size_t)mu.cols());
  for (std::ptrdiff_t d = 0; d < mu.cols(); ++d) m[(size_t)d] = mu(d);
  distributions::detail::check(lc_mahaldist(g.c, m.data(), a.data(), out.data()));
  return out;
}

/* probutils.h:128, probutils.cpp:144-153 */
inline lcmat::VectorXd logsumexp(const lcmat::MatrixXd& X) {
  lcmat::VectorXd out(X.rows());
  for (std::ptrdiff_t r = 0; r < X.rows(); ++r) {
    double mx = X(r, 0);
    for (std::ptrdiff_t c = 1; c < X.cols(); ++c) mx = X(r, c) > mx ? X(r, c) : mx;
    double se = 0.0;
    for (std::ptrdiff_t c = 0; c < X.cols(); ++c) se += std::exp(X(r, c) - mx);
    out(r) = std::log(se) + mx;
  }
  return out;
}

/* probutils.h:140, probutils.cpp:156-189 (power method; thresholds probutils.cpp:39-40) */
inline double eigpower(const lcmat::MatrixXd& A, lcmat::VectorXd& eigvec) {
  if (A.rows() != A.cols()) throw std::invalid_argument("Matrix A must be square!");
  const std::ptrdiff_t n = A.rows();
  if (n == 1) {
    eigvec.setOnes(1);
    return A(0, 0);
  }
  const double thresh = 1.0e-8f;
  std::vector<double> v((size_t)n), o((size_t)n);
  double nrm = 0.0;
  for (std::ptrdiff_t i = 0; i < n; ++i) {
    v[(size_t)i] = -1.0 + 2.0 * (double)i / (double)(n - 1);
    nrm += v[(size_t)i] * v[(size_t)i];
  }
  double eigval = std::sqrt(nrm), vdist = std::numeric_limits<double>::infinity();
  eigvec.resize(n);
  for (std::ptrdiff_t i = 0; i < n; ++i) eigvec(i) = v[(size_t)i] / eigval;
  for (int it = 0; vdist > thresh && it < 100; ++it) {
    for (std::ptrdiff_t i = 0; i < n; ++i) o[(size_t)i] = eigvec(i);
    nrm = 0.0;
    for (std::ptrdiff_t i = 0; i < n; ++i) {
      double s = 0.0;
      for (std::ptrdiff_t j = 0; j < n; ++j) s += A(i, j) * o[(size_t)j];
      v[(size_t)i] = s;
      nrm += s * s;
    }
    eigval = std::sqrt(nrm);
    vdist = 0.0;
    for (std::ptrdiff_t i = 0; i < n; ++i) {
      eigvec(i) = v[(size_t)i] / eigval;
      vdist += (eigvec(i) - o[(size_t)i]) * (eigvec(i) - o[(size_t)i]);
    }
    vdist = std::sqrt(vdist);
  }
  return eigval;
}

/* probutils.h:150, probutils.cpp:192-205 */
inline double logdet(const lcmat::MatrixXd& A) {
  if (A.rows() != A.cols()) throw std::invalid_argument("Matrix A must be square!");
  std::vector<double> a = detail::rowmajor(A);
  const int n = (int)A.rows();
  if (!detail::chol(a, n)) throw std::domain_error("Matrix A is not positive definite.");
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += 2.0 * std::log(a[(size_t)i * n + i]);
  return s;
}

/* probutils.h:159, probutils.cpp:208-219 */
inline lcmat::MatrixXd mxdigamma(const lcmat::MatrixXd& X) {
  lcmat::MatrixXd r(X.rows(), X.cols());
  for (std::ptrdiff_t i = 0; i < X.rows(); ++i)
    for (std::ptrdiff_t j = 0; j < X.cols(); ++j) r(i, j) = lc_digamma(X(i, j));
  return r;
}

/* probutils.h:168, probutils.cpp:222-230 */
inline lcmat::MatrixXd mxlgamma(const lcmat::MatrixXd& X) {
  lcmat::MatrixXd r(X.rows(), X.cols());
  for (std::ptrdiff_t i = 0; i < X.rows(); ++i)
    for (std::ptrdiff_t j = 0; j < X.cols(); ++j) r(i, j) = std::lgamma(X(i, j));
  return r;
}

}  // namespace probutils
#endif /* PROBUTILS_H */
