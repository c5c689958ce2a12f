// The reference's test main (test/cluster_test.cpp:38-69: learnGMC on
// makeXdata, print weights/means/covariances) against the drop-in headers --
// plus the assertions the reference never had.  Data comes in on stdin
// (tests/golden/xcat.json re-typed by the pytest wrapper as plain numbers).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "distributions.h"
#include "libcluster.h"
#include "probutils.h"

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
    REQUIRE(fabs(ng.getN() - cd[0].getN()) < 0.05 * cd[0].getN());
    REQUIRE(fabs(ng.getmean()(0) - cd[0].getmean()(0)) < 0.5 && std::isfinite(ng.fenergy()));
    const lcmat::VectorXd e1 = ng.Eloglike(Xcat), e2 = cd[0].Eloglike(Xcat);
    REQUIRE(e1.size() == Xcat.rows() && std::isfinite(e1(5)) && std::isfinite(e2(5)) && e2(5) < 0);
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
    // (the learner's clusters were updated from the responsibilities of the iteration BEFORE the final E-step, so
    // a refit from the final qZ agrees to the convergence tolerance, not to rounding)
    REQUIRE(fabs(eg.getN() - ce[1].getN()) < 0.05 * ce[1].getN() && eg.getrate()(0) > 0 && ce[1].getrate()(0) > 0);
    REQUIRE(std::isfinite(eg.fenergy()) && std::isfinite(ce[1].fenergy()));
    const lcmat::VectorXd e3 = eg.Eloglike(Xpos), e4 = ce[1].Eloglike(Xpos);
    REQUIRE(e3.size() == Xpos.rows() && std::isfinite(e3(7)) && std::isfinite(e4(7)));
    {  // exact check of the plugin arithmetic: reproduce the learner's own update from its statistics
      ExpGamma same(PRIORVAL, D);
      lcmat::VectorXd ones(Xpos.rows());
      for (int r = 0; r < Xpos.rows(); ++r) ones(r) = 1.0;
      same.addobs(ones, Xpos);
      same.update();
      REQUIRE(fabs(same.getN() - (double)Xpos.rows()) < 1e-9);
      double sx = 0;
      for (int r = 0; r < Xpos.rows(); ++r) sx += Xpos(r, 0);
      REQUIRE(fabs(same.getrate()(0) - (1.0 + Xpos.rows()) / (PRIORVAL + sx)) < 1e-12);  // a/b, distributions.cpp:545-552
    }
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

  // the reference's test/scluster_test.cpp:44-68 and test/mcluster_test.cpp:44-70 (the start is std::rand(), so
  // only structure and internal consistency can be asserted, not a free energy)
  {
    vvMatrixXd Xv(2);
    for (int j = 0; j < J; ++j) Xv[j < J / 2 ? 0 : 1].push_back(X[j]);
    vector<GDirichlet> iw;
    vector<Dirichlet> sw;
    vector<GaussWish> cl;
    vMatrixXd qYs;
    vvMatrixXd qZs;
    const double Fs = learnSCM(Xv, qYs, qZs, iw, sw, cl, PRIORVAL, PRIORVAL, 4, -1, true);
    REQUIRE(std::isfinite(Fs) && iw.size() == 2 && sw.size() >= 1 && sw.size() <= 4 && cl.size() >= 1);
    REQUIRE(qYs.size() == 2 && qYs[0].rows() == J / 2 && (size_t)qYs[0].cols() == sw.size());
    REQUIRE(qZs.size() == 2 && qZs[1].size() == (size_t)(J - J / 2) && (size_t)qZs[1][0].cols() == cl.size());
    double rs = 0;
    for (int t = 0; t < qYs[0].cols(); ++t) rs += qYs[0](0, t);
    REQUIRE(fabs(rs - 1.0) < 1e-9);
    REQUIRE(sw[0].Elogweight().size() == (int)cl.size() && iw[0].Elogweight().size() == (int)sw.size());
    REQUIRE(std::isfinite(cl[0].fenergy()) && cl[0].getN() > 1);

    vMatrixXd Wd(2);
    int no = 0;
    REQUIRE(scanf("%d", &no) == 1 && no == J);
    for (int j = 0; j < 2; ++j) {
      Wd[j].resize(J / 2, D);
      for (int r = 0; r < J / 2; ++r)
        for (int d = 0; d < D; ++d) { double v; REQUIRE(scanf("%lf", &v) == 1); Wd[j](r, d) = v; }
    }
    vector<GaussWish> ict, sct;
    const double Fm = learnMCM(Wd, Xv, qYs, qZs, iw, sw, ict, sct, PRIORVAL, PRIORVAL, 10, -1, true);
    REQUIRE(std::isfinite(Fm) && ict.size() == sw.size() && sct.size() >= 1 && ict.size() <= 10);
    REQUIRE((size_t)qYs[1].cols() == ict.size() && (size_t)qZs[0][0].cols() == sct.size());
    bool bad = false;
    try { learnSCM(Xv, qYs, qZs, iw, sw, cl, PRIORVAL, PRIORVAL, 13); } catch (const invalid_argument&) { bad = true; }
    REQUIRE(bad);  // scluster.cpp:531-533
    bad = false;
    vMatrixXd W1(1, Wd[0]);
    try { learnMCM(W1, Xv, qYs, qZs, iw, sw, ict, sct); } catch (const invalid_argument&) { bad = true; }
    REQUIRE(bad);  // mcluster.cpp:548-549
  }

  // probutils.h: the public utility header (mahaldist runs on the GPU)
  {
    const lcmat::RowVectorXd mu = probutils::mean(Xcat);
    const lcmat::MatrixXd C = probutils::cov(Xcat);
    REQUIRE(mu.size() == D && C.rows() == D && fabs(C(0, 1) - C(1, 0)) < 1e-12);
    const lcmat::VectorXd d2 = probutils::mahaldist(Xcat, mu, C);
    REQUIRE(d2.size() == Xcat.rows());
    double tot = 0.0;  // sum of squared Mahalanobis distances to the sample mean under the sample covariance
    for (int r = 0; r < d2.size(); ++r) { REQUIRE(d2(r) >= 0.0); tot += d2(r); }
    REQUIRE(fabs(tot - (double)D * (Xcat.rows() - 1)) < 1e-8 * tot);  // = D (N - 1) exactly
    // 2 x 2 by hand: (x - mu) C^-1 (x - mu)^T
    const double det = C(0, 0) * C(1, 1) - C(0, 1) * C(1, 0);
    const double a0 = Xcat(3, 0) - mu(0), a1 = Xcat(3, 1) - mu(1);
    const double ref = (C(1, 1) * a0 * a0 - 2 * C(0, 1) * a0 * a1 + C(0, 0) * a1 * a1) / det;
    REQUIRE(fabs(d2(3) - ref) < 1e-10 * (1.0 + ref));
    REQUIRE(fabs(probutils::logdet(C) - log(det)) < 1e-12);
    lcmat::VectorXd ev;
    const double lam = probutils::eigpower(C, ev);
    const double tr = C(0, 0) + C(1, 1), lmax = 0.5 * (tr + sqrt(tr * tr - 4 * det));
    REQUIRE(fabs(lam - lmax) < 1e-6 * lmax && ev.size() == D);
    lcmat::MatrixXd L2(2, 3);
    L2(0, 0) = 1; L2(0, 1) = 2; L2(0, 2) = 3; L2(1, 0) = -1000; L2(1, 1) = -1000; L2(1, 2) = -1001;
    const lcmat::VectorXd ls = probutils::logsumexp(L2);
    REQUIRE(fabs(ls(0) - log(exp(1.0) + exp(2.0) + exp(3.0))) < 1e-12 && fabs(ls(1) - (-1000 + log(2 + exp(-1.0)))) < 1e-12);
    REQUIRE(fabs(probutils::mxdigamma(L2)(0, 0) - (-0.5772156649015329)) < 1e-13);
    REQUIRE(fabs(probutils::mxlgamma(L2)(0, 2) - log(2.0)) < 1e-13);
    REQUIRE(probutils::stdev(Xcat).size() == D && fabs(probutils::stdev(Xcat)(0) - sqrt(C(0, 0))) < 1e-12);
    bool pd = false;
    lcmat::MatrixXd bad(2, 2);
    bad(0, 0) = 1; bad(0, 1) = 2; bad(1, 0) = 2; bad(1, 1) = 1;
    try { probutils::mahaldist(Xcat, mu, bad); } catch (const invalid_argument&) { pd = true; }
    REQUIRE(pd);  // probutils.cpp:131-132
    pd = false;
    try { probutils::logdet(bad); } catch (const domain_error&) { pd = true; }
    REQUIRE(pd);  // probutils.cpp:200-201
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
  // LIBCLUSTER_GPUS=8 (all shards on GPU 0, host-staged sums): the same calls, unsharded and over eight shards -- row
  // blocks for the single-matrix learners, whole groups for the GMC family (16 unequal groups, two per shard) -- must
  // agree in K, F, responsibilities, weights and cluster parameters.  The loop being distributed: cluster.cpp:207-223.
  {
    const char* keep_g = getenv("LIBCLUSTER_GPUS");
    const char* keep_s = getenv("LIBCLUSTER_GPUS_SAME_DEVICE");
    const string old_g = keep_g ? keep_g : "", old_s = keep_s ? keep_s : "";
    const int J8 = 16, D8 = 3, K8 = 5;
    unsigned long long st = 88172645463325252ULL;  // xorshift64: the same points on every platform
    auto uni = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    auto gauss = [&]() { const double u = uni() + 1e-300, v = uni(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); };
    double mu8[K8][D8];
    for (int k = 0; k < K8; ++k)
      for (int d = 0; d < D8; ++d) mu8[k][d] = 7.0 * gauss();
    vMatrixXd Xg(J8);
    int ntot = 0;
    for (int j = 0; j < J8; ++j) {
      const int nj = 400 + 37 * j;
      Xg[j].resize(nj, D8);
      for (int r = 0; r < nj; ++r) {
        const int k = (int)(uni() * K8) % K8;
        for (int d = 0; d < D8; ++d) Xg[j](r, d) = mu8[k][d] + (0.6 + 0.1 * k) * gauss();
      }
      ntot += nj;
    }
    lcmat::MatrixXd Xall(ntot, D8);
    for (int j = 0, o = 0; j < J8; o += Xg[j].rows(), ++j)
      for (int r = 0; r < Xg[j].rows(); ++r)
        for (int d = 0; d < D8; ++d) Xall(o + r, d) = Xg[j](r, d);
    double Fb1 = 0, Fv1 = 0, Fg1 = 0;
    lcmat::MatrixXd qb1, qv1;
    vMatrixXd qg1;
    vector<GaussWish> cb1, cv1, cg1;
    vector<GDirichlet> wg1;
    Dirichlet wb1;
    StickBreak wv1;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 0) {
        unsetenv("LIBCLUSTER_GPUS");
        unsetenv("LIBCLUSTER_GPUS_SAME_DEVICE");
      } else {
        setenv("LIBCLUSTER_GPUS", "8", 1);
        setenv("LIBCLUSTER_GPUS_SAME_DEVICE", "1", 1);
      }
      lcmat::MatrixXd qb, qv;
      vMatrixXd qg;
      vector<GaussWish> cb2, cv2, cg2;
      vector<GDirichlet> wg2;
      Dirichlet wb2;
      StickBreak wv2;
      const double Fb2 = learnBGMM(Xall, qb, wb2, cb2, PRIORVAL, -1, pass == 1);  // (verbose: "Sharding over 8 GPU(s)")
      const double Fv2 = learnVDP(Xall, qv, wv2, cv2);
      const double Fg2 = learnGMC(Xg, qg, wg2, cg2);
      if (pass == 0) {
        Fb1 = Fb2; Fv1 = Fv2; Fg1 = Fg2;
        qb1 = qb; qv1 = qv; qg1 = qg; cb1 = cb2; cv1 = cv2; cg1 = cg2; wg1 = wg2; wb1 = wb2; wv1 = wv2;
        REQUIRE(cb1.size() >= 3 && cv1.size() >= 3 && cg1.size() >= 3);
        continue;
      }
      REQUIRE(cb2.size() == cb1.size() && cv2.size() == cv1.size() && cg2.size() == cg1.size());
      REQUIRE(fabs(Fb2 - Fb1) <= 1e-10 * fabs(Fb1) && fabs(Fv2 - Fv1) <= 1e-10 * fabs(Fv1) && fabs(Fg2 - Fg1) <= 1e-10 * fabs(Fg1));
      REQUIRE(qb.rows() == ntot && qb.cols() == (int)cb1.size() && qv.rows() == ntot && qg.size() == (size_t)J8);
      for (int r = 0; r < ntot; ++r) {
        for (int k = 0; k < qb.cols(); ++k) REQUIRE(fabs(qb(r, k) - qb1(r, k)) < 1e-9);
        for (int k = 0; k < qv.cols(); ++k) REQUIRE(fabs(qv(r, k) - qv1(r, k)) < 1e-9);
      }
      for (int j = 0; j < J8; ++j) {
        REQUIRE(qg[j].rows() == Xg[j].rows() && qg[j].cols() == (int)cg1.size());
        for (int r = 0; r < qg[j].rows(); ++r)
          for (int k = 0; k < qg[j].cols(); ++k) REQUIRE(fabs(qg[j](r, k) - qg1[j](r, k)) < 1e-9);
        for (int k = 0; k < (int)cg1.size(); ++k)
          REQUIRE(fabs(wg2[j].Elogweight()(k) - wg1[j].Elogweight()(k)) < 1e-9 * (1.0 + fabs(wg1[j].Elogweight()(k))));
      }
      for (size_t k = 0; k < cb1.size(); ++k) {
        REQUIRE(fabs(cb2[k].getN() - cb1[k].getN()) < 1e-8 * (1.0 + cb1[k].getN()));
        REQUIRE(fabs(wb2.Elogweight()(k) - wb1.Elogweight()(k)) < 1e-9 * (1.0 + fabs(wb1.Elogweight()(k))));
        for (int d = 0; d < D8; ++d) {
          REQUIRE(fabs(cb2[k].getmean()(d) - cb1[k].getmean()(d)) < 1e-8);
          for (int e = 0; e < D8; ++e) REQUIRE(fabs(cb2[k].getcov()(d, e) - cb1[k].getcov()(d, e)) < 1e-8);
        }
      }
      for (size_t k = 0; k < cv1.size(); ++k) {
        REQUIRE(fabs(cv2[k].getN() - cv1[k].getN()) < 1e-8 * (1.0 + cv1[k].getN()));
        REQUIRE(fabs(wv2.Elogweight()(k) - wv1.Elogweight()(k)) < 1e-9 * (1.0 + fabs(wv1.Elogweight()(k))));
      }
      for (size_t k = 0; k < cg1.size(); ++k) REQUIRE(fabs(cg2[k].getN() - cg1[k].getN()) < 1e-8 * (1.0 + cg1[k].getN()));
    }
    if (keep_g) setenv("LIBCLUSTER_GPUS", old_g.c_str(), 1); else unsetenv("LIBCLUSTER_GPUS");
    if (keep_s) setenv("LIBCLUSTER_GPUS_SAME_DEVICE", old_s.c_str(), 1); else unsetenv("LIBCLUSTER_GPUS_SAME_DEVICE");
    cout << "eight shards OK" << endl;
  }
  cout << "cluster_test OK" << endl;
  return 0;
}
