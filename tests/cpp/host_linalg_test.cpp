// The M-step's factorisations (libcluster_amd/csrc/lc_host.hpp: cholesky, tril_inverse) against their textbook forms with
// the SAME order of operations per element: the blocked kernels promise bit-identical results (an element's terms are
// taken in increasing k, one multiplication and one subtraction / addition each -- only WHICH elements advance together
// changed), so the comparison is memcmp.  Reference: probutils.cpp:128-133, 189-202 (Eigen's LLT / triangular solve in
// the reference; the order contract is this repository's own, the oracle checks the values).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "lc_host.hpp"

namespace plain {
// L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]) / L[j][j], k increasing
static bool cholesky(std::vector<double>& A, int n) {
  std::vector<double> L((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    for (int i = j; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[(size_t)j * n + j] = std::sqrt(s);
      } else {
        L[(size_t)i * n + j] = s;  // divided below, once the pivot of the column is known
      }
    }
    const double ljj = L[(size_t)j * n + j];
    for (int i = j + 1; i < n; ++i) L[(size_t)i * n + j] /= ljj;
  }
  A = L;
  return true;
}
// Li[i][j] = -(sum_{k=j}^{i-1} L[i][k] Li[k][j]) / L[i][i], k increasing; Li[i][i] = 1 / L[i][i]
static std::vector<double> tril_inverse(const std::vector<double>& L, int n) {
  std::vector<double> Li((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < i; ++j) {
      double s = 0.0;
      for (int k = j; k < i; ++k) s += L[(size_t)i * n + k] * Li[(size_t)k * n + j];
      Li[(size_t)i * n + j] = -s / L[(size_t)i * n + i];
    }
    Li[(size_t)i * n + i] = 1.0 / L[(size_t)i * n + i];
  }
  return Li;
}
}  // namespace plain

int main() {
  std::mt19937_64 g(20261003);
  std::normal_distribution<double> nd;
  int bad = 0;
  for (int n = 1; n <= 150; ++n) {
    std::vector<double> B((size_t)n * n), A((size_t)n * n);
    for (auto& v : B) v = nd(g);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += B[(size_t)i * n + k] * B[(size_t)j * n + k];
        A[(size_t)i * n + j] = s + (i == j ? 0.25 * n : 0.0);
      }
    std::vector<double> a1(A), a2(A);
    const bool o1 = plain::cholesky(a1, n), o2 = lch::cholesky(a2, n);
    if (o1 != o2 || std::memcmp(a1.data(), a2.data(), a1.size() * sizeof(double)) != 0) {
      std::printf("cholesky differs from the element-by-element form at n = %d\n", n);
      ++bad;
    }
    const std::vector<double> i1 = plain::tril_inverse(a1, n), i2 = lch::tril_inverse(a2, n);
    if (std::memcmp(i1.data(), i2.data(), i1.size() * sizeof(double)) != 0) {
      std::printf("tril_inverse differs from the element-by-element form at n = %d\n", n);
      ++bad;
    }
  }
  {  // not positive definite: both refuse
    std::vector<double> A = {1.0, 2.0, 2.0, 1.0};
    std::vector<double> a1(A), a2(A);
    if (plain::cholesky(a1, 2) || lch::cholesky(a2, 2)) {
      std::printf("an indefinite matrix was factored\n");
      ++bad;
    }
  }
  {  // norm_certified (Context::recompute_bounded's proof of |M|_2^2 < tau): against a norm found the slow way
    std::vector<double> work;
    for (int n : {1, 2, 3, 5, 16, 23, 64, 100, 128}) {
      for (int rep = 0; rep < 4; ++rep) {
        std::vector<double> M((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i)
          for (int j = 0; j <= i; ++j) M[(size_t)i * n + j] = (i == j ? 1.0 + 0.3 * rep : 0.0) + (rep == 3 ? 1.0 : 0.1) * nd(g);
        // |M|_2 by 5000 power iterations from eight random starts, the largest wins
        double best = 0.0;
        for (int st = 0; st < 8; ++st) {
          std::vector<double> v((size_t)n), w((size_t)n);
          for (auto& x : v) x = nd(g);
          double est = 0.0;
          for (int it = 0; it < 5000; ++it) {
            for (int i = 0; i < n; ++i) {
              double t = 0.0;
              for (int j = 0; j <= i; ++j) t += M[(size_t)i * n + j] * v[(size_t)j];
              w[(size_t)i] = t;
            }
            double nv = 0.0;
            for (int j = 0; j < n; ++j) {
              double t = 0.0;
              for (int i = j; i < n; ++i) t += M[(size_t)i * n + j] * w[(size_t)i];
              v[(size_t)j] = t;
              nv += t * t;
            }
            nv = std::sqrt(nv);
            for (auto& x : v) x /= nv;
            est = std::sqrt(nv);
          }
          best = std::fmax(best, est);
        }
        const bool above = lch::norm_certified(M.data(), n, (best * 1.001) * (best * 1.001), work);
        const bool below = lch::norm_certified(M.data(), n, (best * 0.999) * (best * 0.999), work);
        // the failure the certificate exists for: a bound proposed from the SECOND singular value must be refused
        const bool far_below = lch::norm_certified(M.data(), n, 0.5 * best * best, work);
        if (!above || below || far_below) {
          std::printf("norm_certified wrong at n = %d rep %d: |M|_2 = %.6g, above %d below %d far below %d\n", n, rep, best,
                      (int)above, (int)below, (int)far_below);
          ++bad;
        }
      }
    }
    const double nan = std::nan("");
    std::vector<double> one = {1.0};
    if (lch::norm_certified(one.data(), 1, nan, work) || lch::norm_certified(one.data(), 1, -1.0, work) ||
        lch::norm_certified(&nan, 1, 4.0, work)) {
      std::printf("norm_certified accepted a NaN / negative bound\n");
      ++bad;
    }
  }
  std::printf(bad ? "FAILED\n" : "host factorisations: bit-identical to the element-by-element forms for n = 1 .. 150\n");
  return bad ? 1 : 0;
}
