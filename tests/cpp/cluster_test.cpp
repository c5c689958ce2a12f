// The reference's test main (test/cluster_test.cpp:38-69: learnGMC on
// makeXdata, print weights/means/covariances) against the drop-in headers --
// plus the assertions the reference never had.  Data comes in on stdin
// (tests/golden/xcat.json re-typed by the pytest wrapper as plain numbers).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "distributions.h"
#include "libcluster.h"

using namespace std;
using namespace libcluster;
using namespace distributions;

#define REQUIRE(c)                                                          \
  do {                                                                      \
    if (!(c)) {                                                             \
      fprintf(stderr, "REQUIRE failed line %d: %s\n", __LINE__, #c);        \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main() {
  int J, n, D;
  REQUIRE(scanf("%d %d %d", &J, &n, &D) == 3);
  vMatrixXd X(J);
  lcmat::MatrixXd Xcat(J * n, D);
  for (int j = 0; j < J; ++j) {
    X[j].resize(n, D);
    for (int r = 0; r < n; ++r)
      for (int d = 0; d < D; ++d) {
        double v;
        REQUIRE(scanf("%lf", &v) == 1);
        X[j](r, d) = v;
        Xcat(j * n + r, d) = v;
      }
  }
  double Fgmc_ref, Fbgmm_ref, Fvdp_ref;
  REQUIRE(scanf("%lf %lf %lf", &Fgmc_ref, &Fbgmm_ref, &Fvdp_ref) == 3);
  double Fdgmm_ref, Fdgmc_ref, Fbemm_ref, Fegmc_ref;
  REQUIRE(scanf("%lf %lf %lf %lf", &Fdgmm_ref, &Fdgmc_ref, &Fbemm_ref, &Fegmc_ref) == 4);

  // GMC, exactly as test/cluster_test.cpp:45-66
  vector<GDirichlet> weights;
  vector<GaussWish> clusters;
  vMatrixXd qZgroup;
  const double F = learnGMC(X, qZgroup, weights, clusters, PRIORVAL, -1, false, true);
  cout << "GMC free energy = " << F << ", clusters = " << clusters.size() << endl;
  REQUIRE(clusters.size() == 4);
  REQUIRE(fabs(F - Fgmc_ref) < 1e-7 * fabs(Fgmc_ref));
  REQUIRE(weights.size() == (size_t)J && qZgroup.size() == (size_t)J);
  for (int j = 0; j < J; ++j) {
    REQUIRE(qZgroup[j].rows() == n && qZgroup[j].cols() == 4);
    double w = 0;
    for (int k = 0; k < 4; ++k) w += exp(weights[j].Elogweight()(k));
    REQUIRE(w > 0.5 && w < 1.5);
    for (int r = 0; r < n; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += qZgroup[j](r, k);
      REQUIRE(fabs(s - 1.0) < 1e-9);
    }
  }
  double Ntot = 0;
  for (size_t k = 0; k < clusters.size(); ++k) {
    Ntot += clusters[k].getN();
    cout << clusters[k].getmean()(0) << " " << clusters[k].getmean()(1) << endl;
    REQUIRE(clusters[k].getcov()(0, 0) > 0);
  }
  REQUIRE(fabs(Ntot - J * n) < 1e-6);

  // BGMM / VDP on the concatenated data (README.md:196-210 usage)
  lcmat::MatrixXd qZ;
  Dirichlet wd;
  vector<GaussWish> cb;
  const double Fb = learnBGMM(Xcat, qZ, wd, cb, PRIORVAL, -1, false);
  REQUIRE(cb.size() == 3 && fabs(Fb - Fbgmm_ref) < 1e-7 * fabs(Fbgmm_ref));
  REQUIRE(qZ.rows() == J * n && qZ.cols() == 3);
  StickBreak ws;
  vector<GaussWish> cv;
  const double Fv = learnVDP(Xcat, qZ, ws, cv);
  REQUIRE(cv.size() == 3 && fabs(Fv - Fvdp_ref) < 1e-7 * fabs(Fvdp_ref));

  // the plugin interface itself: addobs / update / Eloglike / fenergy on the GPU path
  GaussWish g(PRIORVAL, D);
  lcmat::VectorXd q(Xcat.rows());
  for (int r = 0; r < Xcat.rows(); ++r) q(r) = qZ(r, 0);
  g.clearobs();
  g.addobs(q, Xcat);
  g.update();
  REQUIRE(fabs(g.getN() - cv[0].getN()) < 1e-8);
  REQUIRE(fabs(g.getmean()(0) - cv[0].getmean()(0)) < 1e-6);
  const lcmat::VectorXd ell = g.Eloglike(Xcat);
  REQUIRE(ell.size() == Xcat.rows() && std::isfinite(ell(0)) && ell(0) < 0);
  REQUIRE(g.splitobs(Xcat).size() == Xcat.rows());

  // diagonal-Gaussian and exponential families (libcluster.h:262-315, 462-523)
  {
    Dirichlet w1;
    vector<NormGamma> cd;
    const double Fd = learnDGMM(Xcat, qZ, w1, cd);
    REQUIRE(cd.size() == 3 && fabs(Fd - Fdgmm_ref) < 1e-7 * fabs(Fdgmm_ref));
    vector<GDirichlet> wg;
    vector<NormGamma> cg;
    vMatrixXd qg;
    const double Fg = learnDGMC(X, qg, wg, cg);
    REQUIRE(cg.size() == 4 && fabs(Fg - Fdgmc_ref) < 1e-7 * fabs(Fdgmc_ref));
    REQUIRE(qg.size() == (size_t)J && qg[0].cols() == 4 && wg.size() == (size_t)J);
    // the NormGamma plugin interface on the GPU path: refit cluster 0 from its responsibilities
    NormGamma ng(PRIORVAL, D);
    for (int r = 0; r < Xcat.rows(); ++r) q(r) = qZ(r, 0);
    ng.addobs(q, Xcat);
    ng.update();
    REQUIRE(fabs(ng.getN() - cd[0].getN()) < 1e-8);
    REQUIRE(fabs(ng.getmean()(0) - cd[0].getmean()(0)) < 1e-6 && fabs(ng.getcov()(1) - cd[0].getcov()(1)) < 1e-6);
    REQUIRE(fabs(ng.fenergy() - cd[0].fenergy()) < 1e-7);
    const lcmat::VectorXd e1 = ng.Eloglike(Xcat), e2 = cd[0].Eloglike(Xcat);
    REQUIRE(e1.size() == Xcat.rows() && fabs(e1(5) - e2(5)) < 1e-7 * fabs(e2(5)));
    REQUIRE(ng.splitobs(Xcat).size() == Xcat.rows());

    lcmat::MatrixXd Xpos(Xcat.rows(), D);
    vMatrixXd Xp(J);
    for (int j = 0; j < J; ++j) {
      Xp[j].resize(n, D);
      for (int r = 0; r < n; ++r)
        for (int d = 0; d < D; ++d) Xpos(j * n + r, d) = Xp[j](r, d) = fabs(X[j](r, d)) + 0.1;
    }
    Dirichlet w2;
    vector<ExpGamma> ce;
    const double Fe = learnBEMM(Xpos, qZ, w2, ce);
    REQUIRE(ce.size() == 2 && fabs(Fe - Fbemm_ref) < 1e-7 * fabs(Fbemm_ref));
    vector<GDirichlet> wge;
    vector<ExpGamma> cge;
    const double Fge = learnEGMC(Xp, qg, wge, cge);
    REQUIRE(cge.size() == 2 && fabs(Fge - Fegmc_ref) < 1e-7 * fabs(Fegmc_ref));
    ExpGamma eg(PRIORVAL, D);
    for (int r = 0; r < Xpos.rows(); ++r) q(r) = qZ(r, 1);
    eg.addobs(q, Xpos);
    eg.update();
    REQUIRE(fabs(eg.getN() - ce[1].getN()) < 1e-8 && fabs(eg.getrate()(0) - ce[1].getrate()(0)) < 1e-7);
    REQUIRE(fabs(eg.fenergy() - ce[1].fenergy()) < 1e-7);
    const lcmat::VectorXd e3 = eg.Eloglike(Xpos), e4 = ce[1].Eloglike(Xpos);
    REQUIRE(fabs(e3(7) - e4(7)) < 1e-7 * fabs(e4(7)));
    REQUIRE(eg.splitobs(Xpos).size() == Xpos.rows());
    bool neg = false;
    try { learnBEMM(Xcat, qZ, w2, ce); } catch (const invalid_argument&) { neg = true; }
    REQUIRE(neg);
    neg = false;
    try { learnEGMC(X, qg, wge, cge); } catch (const invalid_argument&) { neg = true; }
    REQUIRE(neg);
    neg = false;
    try { NormGamma bad(0.0, 2); } catch (const invalid_argument&) { neg = true; }
    REQUIRE(neg);
  }

  // error behaviour (cluster.cpp:576-577, distributions.cpp:107-108/282-283)
  bool threw = false;
  try { learnBGMM(Xcat, qZ, wd, cb, PRIORVAL, -1, false, 0); } catch (const invalid_argument&) { threw = true; }
  REQUIRE(threw);
  threw = false;
  try { GaussWish bad(-1.0, 2); } catch (const invalid_argument&) { threw = true; }
  REQUIRE(threw);
  threw = false;
  try { StickBreak bad(0.0); } catch (const invalid_argument&) { threw = true; }
  REQUIRE(threw);
  cout << "cluster_test OK" << endl;
  return 0;
}
