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
