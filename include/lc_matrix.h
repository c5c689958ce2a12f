/* lc_matrix.h -- the dense types of the libcluster C++ API.
 *
 * The reference's public headers are written in Eigen types
 * (include/libcluster.h:26,135-138; include/distributions.h:24).  When Eigen 3
 * is on the include path these aliases ARE the Eigen types, so existing
 * callers compile unchanged.  When it is not (it is not installed in the build
 * image), a minimal column-major stand-in with the handful of members the API
 * needs is used instead.  Either way the library itself only ever sees
 * (pointer, row stride, column stride) through the C ABI.
 */
#ifndef LC_MATRIX_H
#define LC_MATRIX_H

#include <cstddef>
#include <cstdint>
#include <vector>

#if !defined(LC_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define LC_HAVE_EIGEN 1
#endif
#endif

#ifdef LC_HAVE_EIGEN
#include <Eigen/Dense>
namespace lcmat {
typedef Eigen::MatrixXd MatrixXd;
typedef Eigen::VectorXd VectorXd;
typedef Eigen::RowVectorXd RowVectorXd;
typedef Eigen::ArrayXd ArrayXd;
typedef Eigen::Array<bool, Eigen::Dynamic, 1> ArrayXb;
inline void strides(const MatrixXd& m, int64_t& rs, int64_t& cs) {
  if (MatrixXd::IsRowMajor) { rs = m.cols(); cs = 1; } else { rs = 1; cs = m.rows(); }
}
}  // namespace lcmat
#else
namespace lcmat {

/* dynamic vector of T with the Eigen spellings the API uses */
template <typename T>
class Vec {
 public:
  Vec() {}
  explicit Vec(std::ptrdiff_t n) : d_((size_t)n) {}
  Vec(std::ptrdiff_t n, T v) : d_((size_t)n, v) {}
  std::ptrdiff_t size() const { return (std::ptrdiff_t)d_.size(); }
  std::ptrdiff_t rows() const { return size(); }
  std::ptrdiff_t cols() const { return size(); }
  void resize(std::ptrdiff_t n) { d_.resize((size_t)n); }
  void setZero(std::ptrdiff_t n) { d_.assign((size_t)n, T(0)); }
  void setZero() { d_.assign(d_.size(), T(0)); }
  void setOnes(std::ptrdiff_t n) { d_.assign((size_t)n, T(1)); }
  T& operator()(std::ptrdiff_t i) { return d_[(size_t)i]; }
  const T& operator()(std::ptrdiff_t i) const { return d_[(size_t)i]; }
  T& operator[](std::ptrdiff_t i) { return d_[(size_t)i]; }
  const T& operator[](std::ptrdiff_t i) const { return d_[(size_t)i]; }
  T* data() { return d_.data(); }
  const T* data() const { return d_.data(); }
  T sum() const { T s = T(0); for (const T& v : d_) s += v; return s; }
  std::ptrdiff_t count() const { std::ptrdiff_t c = 0; for (const T& v : d_) c += v ? 1 : 0; return c; }
 private:
  std::vector<T> d_;
};
typedef Vec<double> VectorXd;
typedef Vec<double> RowVectorXd;
typedef Vec<double> ArrayXd;
typedef Vec<unsigned char> ArrayXb;  /* Eigen::Array<bool,Dynamic,1> in the reference (distributions.h:50) */

/* column-major dynamic matrix (Eigen's default storage order) */
class MatrixXd {
 public:
  enum { IsRowMajor = 0 };
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(std::ptrdiff_t r, std::ptrdiff_t c) : r_(r), c_(c), d_((size_t)(r * c)) {}
  std::ptrdiff_t rows() const { return r_; }
  std::ptrdiff_t cols() const { return c_; }
  std::ptrdiff_t size() const { return r_ * c_; }
  void resize(std::ptrdiff_t r, std::ptrdiff_t c) { r_ = r; c_ = c; d_.resize((size_t)(r * c)); }
  void setZero(std::ptrdiff_t r, std::ptrdiff_t c) { r_ = r; c_ = c; d_.assign((size_t)(r * c), 0.0); }
  void setOnes(std::ptrdiff_t r, std::ptrdiff_t c) { r_ = r; c_ = c; d_.assign((size_t)(r * c), 1.0); }
  static MatrixXd Zero(std::ptrdiff_t r, std::ptrdiff_t c) { MatrixXd m; m.setZero(r, c); return m; }
  double& operator()(std::ptrdiff_t i, std::ptrdiff_t j) { return d_[(size_t)(i + j * r_)]; }
  const double& operator()(std::ptrdiff_t i, std::ptrdiff_t j) const { return d_[(size_t)(i + j * r_)]; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
 private:
  std::ptrdiff_t r_, c_;
  std::vector<double> d_;
};
inline void strides(const MatrixXd& m, int64_t& rs, int64_t& cs) { rs = 1; cs = m.rows(); }
}  // namespace lcmat
#endif /* LC_HAVE_EIGEN */
#endif /* LC_MATRIX_H */
