/* distributions.h -- the weight / cluster distribution classes of libcluster
 * (reference: include/distributions.h:60-337), restricted to the families on
 * the accelerated path: Dirichlet, StickBreak, GDirichlet and GaussWish.
 * Same class names, method names, argument meaning and exceptions.  The
 * arithmetic lives behind the C ABI (include/libcluster_hip.h): host M-step
 * pieces in lc_weights_update / lc_gw_mstep, the per-observation passes
 * (addobs, Eloglike) in the gfx950 kernels.  NormGamma / ExpGamma are not part
 * of this build ("next").
 */
#ifndef DISTRIBUTIONS_H
#define DISTRIBUTIONS_H

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "lc_matrix.h"
#include "libcluster_hip.h"

namespace distributions {

const double BETAPRIOR = 1.0;   /* distributions.h:39-43 */
const double NUPRIOR = 1.0;
const double ALPHA1PRIOR = 1.0;
const double ALPHA2PRIOR = 1.0;
const double APRIOR = 1.0;

typedef lcmat::ArrayXb ArrayXb;

namespace detail {
/* lc_status -> the exception classes the reference throws (libcluster.h:171-175) */
inline void check(int rc) {
  if (rc == LC_OK) return;
  const std::string msg = lc_last_error();
  if (rc == LC_EINVAL) throw std::invalid_argument(msg);
  if (rc == LC_EDOMAIN) throw std::domain_error(msg);
  throw std::runtime_error(msg);
}
struct CtxGuard {
  lc_ctx* c;
  CtxGuard() : c(0) { check(lc_ctx_create(0, 0, &c)); }
  ~CtxGuard() { lc_ctx_destroy(c); }
};
inline void upload(lc_ctx* c, const lcmat::MatrixXd& X) {
  int64_t rs, cs;
  lcmat::strides(X, rs, cs);
  const double* p = X.data();
  const int64_t n = (int64_t)X.rows();
  check(lc_ctx_set_data(c, 1, &p, &n, (int)X.cols(), rs, cs));
}
}  // namespace detail

/* ---- weights: reference distributions.h:60-189 --------------------------- */
class WeightDist {
 public:
  virtual void update(const lcmat::ArrayXd& Nk) = 0;
  virtual const lcmat::ArrayXd& Elogweight() const = 0;
  const lcmat::ArrayXd& getNk() const { return this->Nk; }
  virtual double fenergy() const = 0;
  virtual ~WeightDist() {}
 protected:
  WeightDist() : Nk(1, 0.0) {}
  lcmat::ArrayXd Nk;
};

namespace detail {
/* shared implementation: every update goes through lc_weights_update */
class WeightImpl : public WeightDist {
 public:
  void update(const lcmat::ArrayXd& Nk_) {
    const int K = (int)Nk_.size();
    lcmat::ArrayXd e(K);
    double f = 0.0;
    check(lc_weights_update(kind_, prior_, Nk_.data(), K, e.data(), &f));
    this->Nk = Nk_;
    E_logpi = e;
    F_ = f;
  }
  const lcmat::ArrayXd& Elogweight() const { return E_logpi; }
  double fenergy() const { return F_; }
  double prior() const { return prior_; } /* additive: the concentration/alpha given at construction */
 protected:
  WeightImpl(int kind, double prior, const char* msg) : kind_(kind), prior_(prior), E_logpi(1, 0.0), F_(0.0) {
    if (!(prior > 0)) throw std::invalid_argument(msg);
  }
  int kind_;
  double prior_;
  lcmat::ArrayXd E_logpi;
  double F_;
};
}  // namespace detail

class StickBreak : public detail::WeightImpl { /* distributions.h:103-140 */
 public:
  StickBreak() : WeightImpl(LC_W_STICKBREAK, ALPHA1PRIOR, "") {}
  StickBreak(const double concentration)
      : WeightImpl(LC_W_STICKBREAK, concentration, "Concentration parameter has to be > 0!") {}
  virtual ~StickBreak() {}
 protected:
  StickBreak(int kind) : WeightImpl(kind, ALPHA1PRIOR, "") {}
};

class GDirichlet : public StickBreak { /* distributions.h:147-157 */
 public:
  GDirichlet() : StickBreak(LC_W_GDIRICHLET) {}
  virtual ~GDirichlet() {}
};

class Dirichlet : public detail::WeightImpl { /* distributions.h:163-189 */
 public:
  Dirichlet() : WeightImpl(LC_W_DIRICHLET, ALPHA1PRIOR, "") {}
  Dirichlet(const double alpha) : WeightImpl(LC_W_DIRICHLET, alpha, "Alpha prior must be > 0!") {}
  virtual ~Dirichlet() {}
};

/* ---- clusters: reference distributions.h:200-337 ------------------------- */
class ClusterDist {
 public:
  virtual void addobs(const lcmat::VectorXd& qZk, const lcmat::MatrixXd& X) = 0;
  virtual void update() = 0;
  virtual void clearobs() = 0;
  virtual lcmat::VectorXd Eloglike(const lcmat::MatrixXd& X) const = 0;
  virtual double fenergy() const = 0;
  virtual ArrayXb splitobs(const lcmat::MatrixXd& X) const = 0;
  double getN() const { return this->N; }
  double getprior() const { return this->prior; }
  virtual ~ClusterDist() {}
 protected:
  ClusterDist(const double prior, const unsigned int D) : D(D), prior(prior), N(0) {}
  unsigned int D;
  double prior;
  double N;
};

class GaussWish : public ClusterDist {
 public:
  GaussWish(const double clustwidth, const unsigned int D) : ClusterDist(clustwidth, D), F_(0.0) {
    /* the prior is the posterior of "no observations": distributions.cpp:273-298 */
    if (!(clustwidth > 0)) throw std::invalid_argument("clustwidth must be > 0!");
    clearobs();
  }

  /* distributions.cpp:301-313, accumulated on the GPU (suff-stat kernel) */
  void addobs(const lcmat::VectorXd& qZk, const lcmat::MatrixXd& X) {
    if ((unsigned)X.cols() != D) throw std::invalid_argument("Mismatched dims. of cluster params and obs.!");
    if (qZk.rows() != X.rows()) throw std::invalid_argument("qZk and X ar not the same length!");
    detail::CtxGuard g;
    detail::upload(g.c, X);
    detail::check(lc_ctx_set_qz(g.c, 0, qZk.data(), 1, 1, (int64_t)qZk.size()));
    double n = 0.0;
    std::vector<double> xs(D), xxs((size_t)D * D);
    detail::check(lc_suffstat(g.c, 0, &n, xs.data(), xxs.data(), 0));
    N_s += n;
    for (unsigned d = 0; d < D; ++d) x_s[d] += xs[d];
    for (size_t i = 0; i < xxs.size(); ++i) xx_s[i] += xxs[i];
  }

  /* distributions.cpp:316-337 */
  void update() {
    detail::check(lc_gw_mstep(prior, (int)D, N_s, x_s.data(), xx_s.data(), &nu, &beta, m_.data(), iW_.data(), &logdW,
                              &F_, 0, 0));
    N = N_s;
  }

  /* distributions.cpp:340-353 */
  void clearobs() {
    N_s = 0.0;
    x_s.assign(D, 0.0);
    xx_s.assign((size_t)D * D, 0.0);
    m_.assign(D, 0.0);
    iW_.assign((size_t)D * D, 0.0);
    /* posterior := prior (lc_gw_mstep with empty statistics) */
    detail::check(lc_gw_mstep(prior, (int)D, 0.0, x_s.data(), xx_s.data(), &nu, &beta, m_.data(), iW_.data(), &logdW,
                              &F_, 0, 0));
  }

  /* distributions.cpp:356-370, evaluated by the E-step kernel (no weights, no normalisation) */
  lcmat::VectorXd Eloglike(const lcmat::MatrixXd& X) const {
    if ((unsigned)X.cols() != D) throw std::invalid_argument("Arguments do not have the same dimensionality");
    detail::CtxGuard g;
    detail::upload(g.c, X);
    detail::check(lc_eloglike(g.c, 1, &nu, &beta, m_.data(), iW_.data(), &logdW));
    lcmat::VectorXd out(X.rows());
    detail::check(lc_ctx_get_qz(g.c, 0, out.data(), 1, (int64_t)X.rows()));
    return out;
  }

  /* distributions.cpp:373-385 (host: power method + projection, via the learner's own code path) */
  ArrayXb splitobs(const lcmat::MatrixXd& X) const {
    std::vector<double> v(D, 0.0);
    {
      /* probutils.cpp:153-186 */
      if (D == 1) v[0] = 1.0;
      else {
        std::vector<double> o(D), t(D);
        double nrm = 0.0;
        for (unsigned i = 0; i < D; ++i) { t[i] = -1.0 + 2.0 * i / (D - 1); nrm += t[i] * t[i]; }
        nrm = std::sqrt(nrm);
        for (unsigned i = 0; i < D; ++i) v[i] = t[i] / nrm;
        const double thresh = 1.0e-8f;
        double dist = 1e300;
        for (int it = 0; dist > thresh && it < 100; ++it) {
          o = v;
          nrm = 0.0;
          for (unsigned i = 0; i < D; ++i) {
            double s = 0.0;
            for (unsigned j = 0; j < D; ++j) s += iW_[(size_t)i * D + j] * o[j];
            t[i] = s; nrm += s * s;
          }
          nrm = std::sqrt(nrm);
          dist = 0.0;
          for (unsigned i = 0; i < D; ++i) { v[i] = t[i] / nrm; dist += (v[i] - o[i]) * (v[i] - o[i]); }
          dist = std::sqrt(dist);
        }
      }
    }
    ArrayXb out(X.rows());
    for (std::ptrdiff_t r = 0; r < X.rows(); ++r) {
      double s = 0.0;
      for (unsigned d = 0; d < D; ++d) s += (X(r, d) - m_[d]) * v[d];
      out(r) = s >= 0.0;
    }
    return out;
  }

  double fenergy() const { return F_; } /* distributions.cpp:388-399, evaluated at update() */

  lcmat::RowVectorXd getmean() const { /* distributions.h:306 */
    lcmat::RowVectorXd r(D);
    for (unsigned d = 0; d < D; ++d) r(d) = m_[d];
    return r;
  }
  lcmat::MatrixXd getcov() const { /* distributions.h:311: iW / nu */
    lcmat::MatrixXd c(D, D);
    for (unsigned i = 0; i < D; ++i)
      for (unsigned j = 0; j < D; ++j) c(i, j) = iW_[(size_t)i * D + j] / nu;
    return c;
  }
  virtual ~GaussWish() {}

  /* additive (not in the reference): load a posterior computed by the learners */
  void set_posterior_(double N_, double nu_, double beta_, const double* m, const double* iW, double logdW_,
                      double F) {
    N = N_; nu = nu_; beta = beta_; logdW = logdW_; F_ = F;
    m_.assign(m, m + D);
    iW_.assign(iW, iW + (size_t)D * D);
  }
  /* additive: cluster idx (level 0: bottom, 1: top) of a learnt two-level model (learnSCM / learnMCM) */
  static GaussWish from_tmodel_(lc_tmodel* mdl, int level, int idx, double clusterprior, unsigned D) {
    std::vector<double> m(D), iW((size_t)D * D);
    double N, nu, beta, logdW, F;
    detail::check(lc_tmodel_cluster(mdl, level, idx, &N, m.data(), 0, &nu, &beta, iW.data(), &logdW, &F));
    GaussWish c(clusterprior, D);
    c.set_posterior_(N, nu, beta, m.data(), iW.data(), logdW, F);
    return c;
  }
  /* additive: cluster k of a learnt model (used by libcluster.h's learners) */
  static GaussWish from_model_(lc_model* mdl, int k, double clusterprior, unsigned D, double Fk) {
    std::vector<double> m(D), iW((size_t)D * D);
    double N, nu, beta, logdW;
    detail::check(lc_model_cluster(mdl, k, &N, m.data(), 0, &nu, &beta, iW.data(), &logdW));
    GaussWish c(clusterprior, D);
    c.set_posterior_(N, nu, beta, m.data(), iW.data(), logdW, Fk);
    return c;
  }

 private:
  double nu, beta, logdW, F_;
  std::vector<double> m_, iW_; /* row-major D x D */
  double N_s;
  std::vector<double> x_s, xx_s;
};

/* ---- diagonal Gaussian clusters: reference distributions.h:334-399, distributions.cpp:405-517 ------------- */
class NormGamma : public ClusterDist {
 public:
  NormGamma(const double clustwidth, const unsigned int D) : ClusterDist(clustwidth, D), F_(0.0), cst_(0.0) {
    if (clustwidth <= 0) throw std::invalid_argument("clustwidth must be > 0!"); /* distributions.cpp:414-415 */
    clearobs();
  }

  /* distributions.cpp:426-438, accumulated on the GPU (diagonal suff-stat kernel) */
  void addobs(const lcmat::VectorXd& qZk, const lcmat::MatrixXd& X) {
    if ((unsigned)X.cols() != D) throw std::invalid_argument("Mismatched dims. of cluster params and obs.!");
    if (qZk.rows() != X.rows()) throw std::invalid_argument("qZk and X ar not the same length!");
    detail::CtxGuard g;
    detail::upload(g.c, X);
    detail::check(lc_ctx_set_qz(g.c, 0, qZk.data(), 1, 1, (int64_t)qZk.size()));
    double n = 0.0;
    std::vector<double> xs(D), xxs(D);
    detail::check(lc_suffstat_diag(g.c, 0, &n, xs.data(), xxs.data(), 0));
    N_s += n;
    for (unsigned d = 0; d < D; ++d) { x_s[d] += xs[d]; xx_s[d] += xxs[d]; }
  }

  void update() { /* distributions.cpp:441-464 */
    detail::check(lc_ng_mstep(prior, (int)D, N_s, x_s.data(), xx_s.data(), &nu, &beta, m_.data(), L_.data(), &logL,
                              &F_, &cst_));
    N = N_s;
  }

  void clearobs() { /* distributions.cpp:467-480 */
    N_s = 0.0;
    x_s.assign(D, 0.0);
    xx_s.assign(D, 0.0);
    m_.assign(D, 0.0);
    L_.assign(D, 0.0);
    detail::check(lc_ng_mstep(prior, (int)D, 0.0, x_s.data(), xx_s.data(), &nu, &beta, m_.data(), L_.data(), &logL,
                              &F_, &cst_));
  }

  /* distributions.cpp:483-492, evaluated by the diagonal E-step kernel in raw mode */
  lcmat::VectorXd Eloglike(const lcmat::MatrixXd& X) const {
    if ((unsigned)X.cols() != D) throw std::invalid_argument("Arguments do not have the same dimensionality");
    std::vector<double> w2(D), w1(D, 0.0);
    for (unsigned d = 0; d < D; ++d) w2[d] = -0.5 * nu / L_[d];
    detail::CtxGuard g;
    detail::upload(g.c, X);
    detail::check(lc_estep_diag(g.c, 1, m_.data(), w2.data(), w1.data(), &cst_, 1, 0, 0));
    lcmat::VectorXd out(X.rows());
    detail::check(lc_ctx_get_qz(g.c, 0, out.data(), 1, (int64_t)X.rows()));
    return out;
  }

  ArrayXb splitobs(const lcmat::MatrixXd& X) const { /* distributions.cpp:495-505 */
    unsigned ax = 0;
    for (unsigned d = 1; d < D; ++d)
      if (L_[d] > L_[ax]) ax = d;
    ArrayXb out(X.rows());
    for (std::ptrdiff_t r = 0; r < X.rows(); ++r) out(r) = (X(r, ax) - m_[ax]) >= 0.0;
    return out;
  }

  double fenergy() const { return F_; } /* distributions.cpp:508-517, evaluated at update() */

  lcmat::RowVectorXd getmean() const { /* distributions.h:370 */
    lcmat::RowVectorXd r(D);
    for (unsigned d = 0; d < D; ++d) r(d) = m_[d];
    return r;
  }
  lcmat::RowVectorXd getcov() const { /* distributions.h:375: L * nu, as the reference defines it */
    lcmat::RowVectorXd r(D);
    for (unsigned d = 0; d < D; ++d) r(d) = L_[d] * nu;
    return r;
  }
  virtual ~NormGamma() {}

  static NormGamma from_model_(lc_model* mdl, int k, double clusterprior, unsigned D, double Fk) {
    NormGamma c(clusterprior, D);
    detail::check(lc_model_cluster(mdl, k, &c.N, c.m_.data(), 0, &c.nu, &c.beta, c.L_.data(), &c.logL));
    c.F_ = Fk;
    /* distributions.cpp:486-488 */
    c.cst_ = 0.5 * (D * (lc_digamma(c.nu) - std::log(2 * 3.14159265358979323846) - 1.0 / c.beta) - c.logL);
    return c;
  }

 private:
  double nu, beta, logL, F_, cst_;
  std::vector<double> m_, L_;
  double N_s;
  std::vector<double> x_s, xx_s;
};

/* ---- exponential clusters: reference distributions.h:406-456, distributions.cpp:524-589 ------------------- */
class ExpGamma : public ClusterDist {
 public:
  ExpGamma(const double obsmag, const unsigned int D) : ClusterDist(obsmag, D), F_(0.0), cst_(0.0) { clearobs(); }

  /* distributions.cpp:533-542 */
  void addobs(const lcmat::VectorXd& qZk, const lcmat::MatrixXd& X) {
    if ((unsigned)X.cols() != D) throw std::invalid_argument("Mismatched dims. of cluster params and obs.!");
    if (qZk.rows() != X.rows()) throw std::invalid_argument("qZk and X ar not the same length!");
    detail::CtxGuard g;
    detail::upload(g.c, X);
    detail::check(lc_ctx_set_qz(g.c, 0, qZk.data(), 1, 1, (int64_t)qZk.size()));
    double n = 0.0;
    std::vector<double> xs(D);
    detail::check(lc_suffstat_diag(g.c, 0, &n, xs.data(), 0, 0));
    N_s += n;
    for (unsigned d = 0; d < D; ++d) x_s[d] += xs[d];
  }

  void update() { /* distributions.cpp:545-552 */
    detail::check(lc_eg_mstep(prior, (int)D, N_s, x_s.data(), &a, ib_.data(), &logb, &F_, &cst_));
    N = N_s;
  }

  void clearobs() { /* distributions.cpp:555-565 */
    N_s = 0.0;
    x_s.assign(D, 0.0);
    ib_.assign(D, 0.0);
    detail::check(lc_eg_mstep(prior, (int)D, 0.0, x_s.data(), &a, ib_.data(), &logb, &F_, &cst_));
  }

  /* distributions.cpp:568-572 */
  lcmat::VectorXd Eloglike(const lcmat::MatrixXd& X) const {
    if ((unsigned)X.cols() != D) throw std::invalid_argument("Arguments do not have the same dimensionality");
    std::vector<double> z(D, 0.0), w1(D);
    for (unsigned d = 0; d < D; ++d) w1[d] = -a * ib_[d];
    detail::CtxGuard g;
    detail::upload(g.c, X);
    detail::check(lc_estep_diag(g.c, 1, z.data(), z.data(), w1.data(), &cst_, 1, 0, 0));
    lcmat::VectorXd out(X.rows());
    detail::check(lc_ctx_get_qz(g.c, 0, out.data(), 1, (int64_t)X.rows()));
    return out;
  }

  ArrayXb splitobs(const lcmat::MatrixXd& X) const { /* distributions.cpp:575-581 */
    std::vector<double> xv(X.rows());
    double mean = 0.0;
    for (std::ptrdiff_t r = 0; r < X.rows(); ++r) {
      double s = 0.0;
      for (unsigned d = 0; d < D; ++d) s += X(r, d) * (a * ib_[d]);
      xv[r] = s;
      mean += s;
    }
    if (X.rows() > 0) mean /= (double)X.rows();
    ArrayXb out(X.rows());
    for (std::ptrdiff_t r = 0; r < X.rows(); ++r) out(r) = xv[r] > mean;
    return out;
  }

  double fenergy() const { return F_; } /* distributions.cpp:584-589 */

  lcmat::RowVectorXd getrate() { /* distributions.h:433 */
    lcmat::RowVectorXd r(D);
    for (unsigned d = 0; d < D; ++d) r(d) = a * ib_[d];
    return r;
  }
  virtual ~ExpGamma() {}

  static ExpGamma from_model_(lc_model* mdl, int k, double clusterprior, unsigned D, double Fk) {
    ExpGamma c(clusterprior, D);
    detail::check(lc_model_cluster(mdl, k, &c.N, 0, 0, &c.a, 0, c.ib_.data(), &c.logb));
    c.F_ = Fk;
    c.cst_ = D * lc_digamma(c.a) - c.logb; /* distributions.cpp:570 */
    return c;
  }

 private:
  double a, logb, F_, cst_;
  std::vector<double> ib_;
  double N_s;
  std::vector<double> x_s;
};

}  // namespace distributions
#endif /* DISTRIBUTIONS_H */
