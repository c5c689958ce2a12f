/* libcluster.h -- learnVDP / learnBGMM / learnGMC / learnSGMC with the reference's
 * signatures (include/libcluster.h:177-186, 218-227, 356-366, 409-419), running the
 * variational E-step and sufficient statistics on an MI355X through the C ABI
 * (include/libcluster_hip.h).  Drop-in for that path: same namespaces, names,
 * argument meaning, constants and exception classes.  The other learners of the
 * reference (DGMM, BEMM, DGMC, EGMC, SCM, MCM) are not part of this build.
 */
#ifndef LIBCLUSTER_H
#define LIBCLUSTER_H

#include <stdexcept>
#include <thread>
#include <vector>

#include "distributions.h"
#include "lc_matrix.h"
#include "libcluster_hip.h"

namespace libcluster {

const double PRIORVAL = 1.0;               /* include/libcluster.h:122-127 */
const unsigned int TRUNC = 100;
const unsigned int SPLITITER = 15;
const double CONVERGE = 1e-5f;
const double FENGYDEL = CONVERGE / 10;
const double ZEROCUTOFF = 0.1f;

typedef std::vector<lcmat::MatrixXd> vMatrixXd;                /* :135 */
typedef std::vector<std::vector<lcmat::MatrixXd> > vvMatrixXd; /* :138 */

namespace detail {
inline unsigned default_threads() { /* the reference's omp_get_max_threads() default */
  const unsigned n = std::thread::hardware_concurrency();
  return n ? n : 1;
}
struct ModelGuard {
  lc_model* m;
  ModelGuard() : m(0) {}
  ~ModelGuard() { if (m) lc_model_free(m); }
};
template <class W, class C>
inline double run(int algo, const vMatrixXd& X, vMatrixXd& qZ, std::vector<W>& weights,
                  std::vector<C>& clusters, double wprior, double clusterprior, int maxclusters,
                  bool sparse, bool verbose, unsigned nthreads) {
  using distributions::detail::check;
  const int J = (int)X.size();
  if (J < 1) throw std::invalid_argument("need at least one group of observations");
  const int D = (int)X[0].cols();
  std::vector<const double*> ptr(J);
  std::vector<int64_t> n(J);
  int64_t rs = 0, cs = 0;
  vMatrixXd copies; /* strides must agree between groups: column-major groups of different N do not */
  bool same = true;
  for (int j = 0; j < J; ++j) {
    if ((int)X[j].cols() != D) throw std::invalid_argument("X dimensions are inconsistent between groups!");
    int64_t r, c;
    lcmat::strides(X[j], r, c);
    if (j == 0) { rs = r; cs = c; }
    else if (r != rs || c != cs) same = false;
  }
  std::vector<std::vector<double> > rowmajor;
  if (!same) { /* re-pack row-major once; the library transposes on upload anyway */
    rowmajor.resize(J);
    for (int j = 0; j < J; ++j) {
      rowmajor[j].resize((size_t)X[j].rows() * D);
      for (std::ptrdiff_t r = 0; r < X[j].rows(); ++r)
        for (int d = 0; d < D; ++d) rowmajor[j][(size_t)r * D + d] = X[j](r, d);
      ptr[j] = rowmajor[j].data();
    }
    rs = D; cs = 1;
  } else {
    for (int j = 0; j < J; ++j) ptr[j] = X[j].data();
  }
  for (int j = 0; j < J; ++j) n[j] = (int64_t)X[j].rows();
  ModelGuard g;
  double F = 0.0;
  check(lc_learn(algo, J, ptr.data(), n.data(), D, rs, cs, wprior, clusterprior, maxclusters, sparse ? 1 : 0,
                 verbose ? 1 : 0, nthreads, 0, &g.m, &F));
  int K = 0;
  check(lc_model_dims(g.m, 0, &K, 0));
  /* qZ, weights, clusters are overwritten exactly as the reference does (cluster.cpp:583-585, 192-193) */
  qZ.resize(J);
  weights.resize(J, W());
  for (int j = 0; j < J; ++j) {
    qZ[j].resize(X[j].rows(), K);
    int64_t r, c;
    lcmat::strides(qZ[j], r, c);
    if (X[j].rows() > 0) check(lc_model_get_qz(g.m, j, qZ[j].data(), r, c));
    lcmat::ArrayXd Nk(K);
    check(lc_model_weights(g.m, j, 0, Nk.data()));
    weights[j].update(Nk); /* same arithmetic as inside the learner => identical Elogweight() */
  }
  clusters.clear();
  std::vector<double> Fc(K), Fw(J);
  check(lc_model_fenergy(g.m, Fw.data(), Fc.data()));
  for (int k = 0; k < K; ++k) clusters.push_back(C::from_model_(g.m, k, clusterprior, (unsigned)D, Fc[k]));
  return F;
}
}  // namespace detail

/* include/libcluster.h:177-186, src/cluster.cpp:636-664 */
inline double learnVDP(const lcmat::MatrixXd& X, lcmat::MatrixXd& qZ, distributions::StickBreak& weights,
                       std::vector<distributions::GaussWish>& clusters, const double clusterprior = PRIORVAL,
                       const int maxclusters = -1, const bool verbose = false,
                       const unsigned int nthreads = detail::default_threads()) {
  vMatrixXd vX(1, X), vq;
  std::vector<distributions::StickBreak> vw(1, weights);
  const double F = detail::run(LC_ALGO_VDP, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
  qZ = vq[0];
  weights = vw[0];
  return F;
}

/* include/libcluster.h:218-227, src/cluster.cpp:667-695 */
inline double learnBGMM(const lcmat::MatrixXd& X, lcmat::MatrixXd& qZ, distributions::Dirichlet& weights,
                        std::vector<distributions::GaussWish>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  vMatrixXd vX(1, X), vq;
  std::vector<distributions::Dirichlet> vw(1, weights);
  const double F = detail::run(LC_ALGO_BGMM, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
  qZ = vq[0];
  weights = vw[0];
  return F;
}

/* include/libcluster.h:356-366, src/cluster.cpp:763-784 */
inline double learnGMC(const vMatrixXd& X, vMatrixXd& qZ, std::vector<distributions::GDirichlet>& weights,
                       std::vector<distributions::GaussWish>& clusters, const double clusterprior = PRIORVAL,
                       const int maxclusters = -1, const bool sparse = false, const bool verbose = false,
                       const unsigned int nthreads = detail::default_threads()) {
  return detail::run(LC_ALGO_GMC, X, qZ, weights, clusters, 1.0, clusterprior, maxclusters, sparse, verbose,
                     nthreads);
}

/* include/libcluster.h:409-419, src/cluster.cpp:787-807 */
inline double learnSGMC(const vMatrixXd& X, vMatrixXd& qZ, std::vector<distributions::Dirichlet>& weights,
                        std::vector<distributions::GaussWish>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool sparse = false, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  return detail::run(LC_ALGO_SGMC, X, qZ, weights, clusters, 1.0, clusterprior, maxclusters, sparse, verbose,
                     nthreads);
}

/* include/libcluster.h:262-271, src/cluster.cpp:697-726 */
inline double learnDGMM(const lcmat::MatrixXd& X, lcmat::MatrixXd& qZ, distributions::Dirichlet& weights,
                        std::vector<distributions::NormGamma>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  vMatrixXd vX(1, X), vq;
  std::vector<distributions::Dirichlet> vw(1, weights);
  const double F = detail::run(LC_ALGO_DGMM, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
  qZ = vq[0];
  weights = vw[0];
  return F;
}

/* include/libcluster.h:306-315, src/cluster.cpp:729-760; std::invalid_argument if X has a negative entry */
inline double learnBEMM(const lcmat::MatrixXd& X, lcmat::MatrixXd& qZ, distributions::Dirichlet& weights,
                        std::vector<distributions::ExpGamma>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  vMatrixXd vX(1, X), vq;
  std::vector<distributions::Dirichlet> vw(1, weights);
  const double F = detail::run(LC_ALGO_BEMM, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
  qZ = vq[0];
  weights = vw[0];
  return F;
}

/* include/libcluster.h:462-472, src/cluster.cpp:810-831 */
inline double learnDGMC(const vMatrixXd& X, vMatrixXd& qZ, std::vector<distributions::GDirichlet>& weights,
                        std::vector<distributions::NormGamma>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool sparse = false, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  return detail::run(LC_ALGO_DGMC, X, qZ, weights, clusters, 1.0, clusterprior, maxclusters, sparse, verbose,
                     nthreads);
}

/* include/libcluster.h:513-523, src/cluster.cpp:834-873; std::invalid_argument if X has a negative entry */
inline double learnEGMC(const vMatrixXd& X, vMatrixXd& qZ, std::vector<distributions::GDirichlet>& weights,
                        std::vector<distributions::ExpGamma>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool sparse = false, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  return detail::run(LC_ALGO_EGMC, X, qZ, weights, clusters, 1.0, clusterprior, maxclusters, sparse, verbose,
                     nthreads);
}

}  // namespace libcluster
#endif /* LIBCLUSTER_H */
