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
/* learnVDP / learnBGMM / learnDGMM / learnBEMM take ONE matrix: views of it as a one-group data set, so that neither X
 * (5 GB at N = 10M, D = 64) nor qZ is copied on the host -- the reference does copy both (src/cluster.cpp:651, 661,
 * 682, 692) */
struct OneX {
  const lcmat::MatrixXd* p;
  size_t size() const { return 1; }
  const lcmat::MatrixXd& operator[](size_t) const { return *p; }
};
struct OneQ {
  lcmat::MatrixXd* p;
  void resize(size_t) {}
  lcmat::MatrixXd& operator[](size_t) { return *p; }
};
template <class W, class C, class XV, class QV>
inline double run(int algo, const XV& X, QV& qZ, std::vector<W>& weights,
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
  /* the priors the caller's weight objects carry: vbem's weights.resize(J, W()) keeps existing elements and
   * default-constructs the rest (cluster.cpp:192) */
  std::vector<double> wpj(J, W().prior());
  for (int j = 0; j < J && j < (int)weights.size(); ++j) wpj[j] = weights[j].prior();
  check(lc_learn_w(algo, J, ptr.data(), n.data(), D, rs, cs, wprior, wpj.data(), clusterprior, maxclusters,
                   sparse ? 1 : 0, verbose ? 1 : 0, nthreads, 0, &g.m, &F));
  int K = 0;
  check(lc_model_dims(g.m, 0, &K, 0));
  /* qZ, weights, clusters are overwritten exactly as the reference does (cluster.cpp:583-585, 192-193) */
  qZ.resize(J);
  weights.resize(J, W());
  int64_t Ntot = 0;
  for (int j = 0; j < J; ++j) Ntot += (int64_t)X[j].rows();
  /* one pipelined transfer for all groups, straight into the (column-major) matrices */
  std::vector<double*> qptr(J);
  for (int j = 0; j < J; ++j) {
    qZ[j].resize(X[j].rows(), K);
    qptr[j] = qZ[j].data();
  }
  if (Ntot > 0) {
    if (!lcmat::MatrixXd::IsRowMajor) {
      check(lc_model_get_qz_all_colmajor(g.m, qptr.data()));
    } else { /* EIGEN_DEFAULT_TO_ROW_MAJOR builds: the row-major bulk transfer, then one block copy per group */
      std::vector<double> allq((size_t)Ntot * K);
      check(lc_model_get_qz_all(g.m, allq.data()));
      size_t o = 0;
      for (int j = 0; j < J; ++j) {
        const size_t nj = (size_t)X[j].rows() * K;
        for (size_t t = 0; t < nj; ++t) qptr[j][t] = allq[o + t];
        o += nj;
      }
    }
  }
  for (int j = 0; j < J; ++j) {
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
  const detail::OneX vX = {&X};
  detail::OneQ vq = {&qZ};
  std::vector<distributions::StickBreak> vw(1, weights);
  const double F = detail::run(LC_ALGO_VDP, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
  weights = vw[0];
  return F;
}

/* include/libcluster.h:218-227, src/cluster.cpp:667-695 */
inline double learnBGMM(const lcmat::MatrixXd& X, lcmat::MatrixXd& qZ, distributions::Dirichlet& weights,
                        std::vector<distributions::GaussWish>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  const detail::OneX vX = {&X};
  detail::OneQ vq = {&qZ};
  std::vector<distributions::Dirichlet> vw(1, weights);
  const double F = detail::run(LC_ALGO_BGMM, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
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
  const detail::OneX vX = {&X};
  detail::OneQ vq = {&qZ};
  std::vector<distributions::Dirichlet> vw(1, weights);
  const double F = detail::run(LC_ALGO_DGMM, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
  weights = vw[0];
  return F;
}

/* include/libcluster.h:306-315, src/cluster.cpp:729-760; std::invalid_argument if X has a negative entry */
inline double learnBEMM(const lcmat::MatrixXd& X, lcmat::MatrixXd& qZ, distributions::Dirichlet& weights,
                        std::vector<distributions::ExpGamma>& clusters, const double clusterprior = PRIORVAL,
                        const int maxclusters = -1, const bool verbose = false,
                        const unsigned int nthreads = detail::default_threads()) {
  const detail::OneX vX = {&X};
  detail::OneQ vq = {&qZ};
  std::vector<distributions::Dirichlet> vw(1, weights);
  const double F = detail::run(LC_ALGO_BEMM, vX, vq, vw, clusters, weights.prior(), clusterprior, maxclusters, false,
                               verbose, nthreads);
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

namespace detail {
struct TModelGuard {
  lc_tmodel* m;
  TModelGuard() : m(0) {}
  ~TModelGuard() { if (m) lc_tmodel_free(m); }
};
/* learnSCM / learnMCM over lc_learn_topic: W == 0 selects the SCM */
inline double run_topic(const vMatrixXd* W, const vvMatrixXd& X, vMatrixXd& qY, vvMatrixXd& qZ,
                        std::vector<distributions::GDirichlet>& weights_j,
                        std::vector<distributions::Dirichlet>& weights_t,
                        std::vector<distributions::GaussWish>* clusters_t,
                        std::vector<distributions::GaussWish>& clusters_k, double prior_t, double prior_k,
                        unsigned maxT, int maxK, bool verbose, unsigned nthreads) {
  using distributions::detail::check;
  const int J = (int)X.size();
  if (W) { /* mcluster.cpp:548-556 */
    if (W->size() != X.size()) throw std::invalid_argument("W and X need to have the same number of groups!");
    for (int j = 0; j < J; ++j)
      if ((size_t)(*W)[j].rows() != X[j].size())
        throw std::invalid_argument("W and X need to have the same number of 'docs'!");
  }
  if (J < 1) throw std::invalid_argument("need at least one group of observations");
  std::vector<int> Ij(J);
  std::vector<std::vector<double> > docs; /* row-major copies: documents rarely share strides */
  std::vector<const double*> ptr;
  std::vector<int64_t> n;
  int D = -1;
  for (int j = 0; j < J; ++j) {
    Ij[j] = (int)X[j].size();
    for (size_t i = 0; i < X[j].size(); ++i) {
      const lcmat::MatrixXd& x = X[j][i];
      if (D < 0) D = (int)x.cols();
      if ((int)x.cols() != D) throw std::invalid_argument("X dimensions are inconsistent between groups!");
      docs.push_back(std::vector<double>((size_t)x.rows() * D));
      for (std::ptrdiff_t r = 0; r < x.rows(); ++r)
        for (int d = 0; d < D; ++d) docs.back()[(size_t)r * D + d] = x(r, d);
      n.push_back((int64_t)x.rows());
    }
  }
  if (D < 0) throw std::invalid_argument("need at least one document");
  for (size_t i = 0; i < docs.size(); ++i) ptr.push_back(docs[i].data());
  std::vector<std::vector<double> > wrow;
  std::vector<const double*> wptr;
  int Dt = 0;
  if (W) {
    Dt = (int)(*W)[0].cols();
    wrow.resize(J);
    for (int j = 0; j < J; ++j) {
      const lcmat::MatrixXd& w = (*W)[j];
      if ((int)w.cols() != Dt) throw std::invalid_argument("W dimensions are inconsistent between groups!");
      wrow[j].resize((size_t)w.rows() * Dt);
      for (std::ptrdiff_t r = 0; r < w.rows(); ++r)
        for (int d = 0; d < Dt; ++d) wrow[j][(size_t)r * Dt + d] = w(r, d);
      wptr.push_back(wrow[j].data());
    }
  }
  TModelGuard g;
  double F = 0.0;
  check(lc_learn_topic(J, Ij.data(), ptr.data(), n.data(), D, D, 1, W ? wptr.data() : 0, Dt, 0, prior_t, prior_k, maxT,
                       maxK, verbose ? 1 : 0, nthreads, 0, &g.m, &F));
  int T = 0, K = 0;
  check(lc_tmodel_dims(g.m, 0, 0, &T, &K, 0, 0));
  qY.resize(J);
  qZ.resize(J);
  weights_j.assign(J, distributions::GDirichlet());
  weights_t.assign(T, W ? distributions::Dirichlet() : distributions::Dirichlet(prior_t));
  std::vector<double> buf;
  int64_t Ntot = 0, qrow = 0;
  for (size_t i = 0; i < n.size(); ++i) Ntot += n[i];
  const bool cm = !lcmat::MatrixXd::IsRowMajor;
  std::vector<double> allq(cm ? 0 : (size_t)Ntot * K);
  if (cm) { /* one pipelined transfer straight into the (column-major) matrices of all documents */
    std::vector<double*> qptr;
    for (int j = 0; j < J; ++j) {
      qZ[j].resize(Ij[j]);
      for (int i = 0; i < Ij[j]; ++i) {
        qZ[j][i].resize(X[j][i].rows(), K);
        qptr.push_back(qZ[j][i].data());
      }
    }
    if (Ntot > 0) check(lc_tmodel_get_qz_all_colmajor(g.m, qptr.data()));
  } else if (Ntot > 0) {
    check(lc_tmodel_get_qz_all(g.m, allq.data())); /* one transfer for all documents */
  }
  int doc = 0;
  for (int j = 0; j < J; ++j) {
    buf.resize((size_t)Ij[j] * T);
    if (Ij[j] > 0) check(lc_tmodel_get_qy(g.m, j, buf.data()));
    qY[j].resize(Ij[j], T);
    for (int i = 0; i < Ij[j]; ++i)
      for (int t = 0; t < T; ++t) qY[j](i, t) = buf[(size_t)i * T + t];
    qZ[j].resize(Ij[j]);
    for (int i = 0; i < Ij[j] && !cm; ++i, ++doc) {
      qZ[j][i].resize(X[j][i].rows(), K);
      for (std::ptrdiff_t r = 0; r < X[j][i].rows(); ++r)
        for (int k = 0; k < K; ++k) qZ[j][i](r, k) = allq[(size_t)(qrow + r) * K + k];
      qrow += (int64_t)X[j][i].rows();
    }
    lcmat::ArrayXd Nk(T);
    check(lc_tmodel_weights(g.m, 0, j, 0, Nk.data()));
    weights_j[j].update(Nk); /* same arithmetic as inside the learner */
  }
  for (int t = 0; t < T; ++t) {
    lcmat::ArrayXd Nk(K);
    check(lc_tmodel_weights(g.m, 1, t, 0, Nk.data()));
    weights_t[t].update(Nk);
  }
  clusters_k.clear();
  for (int k = 0; k < K; ++k)
    clusters_k.push_back(distributions::GaussWish::from_tmodel_(g.m, 0, k, prior_k, (unsigned)D));
  if (clusters_t) {
    clusters_t->clear();
    for (int t = 0; t < T; ++t)
      clusters_t->push_back(distributions::GaussWish::from_tmodel_(g.m, 1, t, prior_t, (unsigned)Dt));
  }
  return F;
}
}  // namespace detail

/* include/libcluster.h:583-596, src/scluster.cpp:578-605.  qY starts from std::rand() like the reference. */
inline double learnSCM(const vvMatrixXd& X, vMatrixXd& qY, vvMatrixXd& qZ,
                       std::vector<distributions::GDirichlet>& weights_j,
                       std::vector<distributions::Dirichlet>& weights_t,
                       std::vector<distributions::GaussWish>& clusters, const double dirprior = PRIORVAL,
                       const double gausprior = PRIORVAL, const unsigned int maxT = TRUNC, const int maxK = -1,
                       const bool verbose = false, const unsigned int nthreads = detail::default_threads()) {
  return detail::run_topic(0, X, qY, qZ, weights_j, weights_t, 0, clusters, dirprior, gausprior, maxT, maxK, verbose,
                           nthreads);
}

/* include/libcluster.h:661-676, src/mcluster.cpp:613-642 */
inline double learnMCM(const vMatrixXd& W, const vvMatrixXd& X, vMatrixXd& qY, vvMatrixXd& qZ,
                       std::vector<distributions::GDirichlet>& weights_j,
                       std::vector<distributions::Dirichlet>& weights_t,
                       std::vector<distributions::GaussWish>& clusters_t,
                       std::vector<distributions::GaussWish>& clusters_k, const double prior_t = PRIORVAL,
                       const double prior_k = PRIORVAL, const unsigned int maxT = TRUNC, const int maxK = -1,
                       const bool verbose = false, const unsigned int nthreads = detail::default_threads()) {
  return detail::run_topic(&W, X, qY, qZ, weights_j, weights_t, &clusters_t, clusters_k, prior_t, prior_k, maxT, maxK,
                           verbose, nthreads);
}

}  // namespace libcluster
#endif /* LIBCLUSTER_H */
