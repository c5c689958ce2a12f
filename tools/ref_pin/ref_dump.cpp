// Pin the oracle on a machine that HAS the reference's dependencies (Eigen 3, Boost.Math, OpenMP): this program links
// the REAL dsteinberg/libcluster, runs its learners on the inputs of tests/golden/xcat.json (the reference's own test
// data, test/testdata.h) and writes what they return in the schema of tests/golden/xcat_traces.json, so that
//     python tools/ref_pin/compare.py ref_xcat.json
// can say whether oracle/lc_oracle.py -- and with it every parity claim of this repository -- agrees with the reference.
// It cannot be built in the container this repository was written in (neither Eigen nor Boost is installed there, and
// stand-ins for them would pin nothing): see tools/ref_pin/README.md for the three commands.
//
// Reference entry points used (include/libcluster.h of the reference): learnVDP :177, learnBGMM :218, learnGMC :356,
// learnSGMC :409, learnDGMM :262, learnBEMM :306, learnDGMC :462, learnEGMC :513; accessors distributions.h: getN :248, getmean
// :306 / :370, getcov :311 / :375, getrate :435, Elogweight :113 / :173.
//
// Input: a text file written by tools/ref_pin/export_inputs.py (no JSON parser needed here):
//     J
//     N_j D        (J times, each followed by N_j rows of D numbers)
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "distributions.h"
#include "libcluster.h"

using namespace Eigen;
using namespace libcluster;
using namespace distributions;

static void put_matrix(FILE* f, const MatrixXd& M) {
  std::fputc('[', f);
  for (int i = 0; i < M.rows(); ++i) {
    std::fputs(i ? ", [" : "[", f);
    for (int j = 0; j < M.cols(); ++j) std::fprintf(f, j ? ", %.17g" : "%.17g", M(i, j));
    std::fputc(']', f);
  }
  std::fputc(']', f);
}

template <class W, class C>
static void put_result(FILE* f, const char* name, double F, const std::vector<MatrixXd>& qZ, const std::vector<W>& weights,
                       const std::vector<C>& clusters, bool last) {
  const int K = (int)clusters.size();
  std::fprintf(f, "  \"%s\": {\"F\": %.17g, \"K\": %d, \"N\": [", name, F, K);
  for (int k = 0; k < K; ++k) std::fprintf(f, k ? ", %.17g" : "%.17g", clusters[k].getN());
  std::fputs("], \"means\": [", f);
  for (int k = 0; k < K; ++k) {
    if (k) std::fputs(", ", f);
    std::fputc('[', f);
    const RowVectorXd m = clusters[k].getmean();
    for (int d = 0; d < m.size(); ++d) std::fprintf(f, d ? ", %.17g" : "%.17g", m(d));
    std::fputc(']', f);
  }
  std::fputs("], \"covs\": [", f);
  for (int k = 0; k < K; ++k) {
    if (k) std::fputs(", ", f);
    put_matrix(f, clusters[k].getcov());
  }
  std::fputs("], \"Elogweight\": [", f);
  for (size_t j = 0; j < weights.size(); ++j) {
    if (j) std::fputs(", ", f);
    std::fputc('[', f);
    const ArrayXd e = weights[j].Elogweight();
    for (int k = 0; k < e.size(); ++k) std::fprintf(f, k ? ", %.17g" : "%.17g", e(k));
    std::fputc(']', f);
  }
  std::fputs("], \"qZ\": [", f);
  for (size_t j = 0; j < qZ.size(); ++j) {
    if (j) std::fputs(", ", f);
    put_matrix(f, qZ[j]);
  }
  std::fprintf(f, "]}%s\n", last ? "" : ",");
}

// NormGamma / ExpGamma learners (round 6): the record layout of tests/golden/family_traces.json -- "means" + "covs" (the
// diagonal as a vector: NormGamma::getcov, distributions.h:375) or "rates" (ExpGamma::getrate, distributions.h:435)
static void put_rowvec(FILE* f, const RowVectorXd& v) {
  std::fputc('[', f);
  for (int d = 0; d < v.size(); ++d) std::fprintf(f, d ? ", %.17g" : "%.17g", v(d));
  std::fputc(']', f);
}
template <class W>
static void put_common(FILE* f, const char* name, double F, int K, const std::vector<double>& N, const std::vector<MatrixXd>& qZ,
                       const std::vector<W>& weights) {
  std::fprintf(f, "  \"%s\": {\"F\": %.17g, \"K\": %d, \"N\": [", name, F, K);
  for (int k = 0; k < K; ++k) std::fprintf(f, k ? ", %.17g" : "%.17g", N[(size_t)k]);
  std::fputs("], \"Elogweight\": [", f);
  for (size_t j = 0; j < weights.size(); ++j) {
    if (j) std::fputs(", ", f);
    std::fputc('[', f);
    const ArrayXd e = weights[j].Elogweight();
    for (int k = 0; k < e.size(); ++k) std::fprintf(f, k ? ", %.17g" : "%.17g", e(k));
    std::fputc(']', f);
  }
  std::fputs("], \"qZ\": [", f);
  for (size_t j = 0; j < qZ.size(); ++j) {
    if (j) std::fputs(", ", f);
    put_matrix(f, qZ[j]);
  }
  std::fputc(']', f);
}
template <class W>
static void put_normgamma(FILE* f, const char* name, double F, const std::vector<MatrixXd>& qZ, const std::vector<W>& weights,
                          const std::vector<NormGamma>& cl, bool last) {
  std::vector<double> N;
  for (const auto& c : cl) N.push_back(c.getN());
  put_common(f, name, F, (int)cl.size(), N, qZ, weights);
  std::fputs(", \"means\": [", f);
  for (size_t k = 0; k < cl.size(); ++k) {
    if (k) std::fputs(", ", f);
    put_rowvec(f, cl[k].getmean());
  }
  std::fputs("], \"covs\": [", f);
  for (size_t k = 0; k < cl.size(); ++k) {
    if (k) std::fputs(", ", f);
    put_rowvec(f, cl[k].getcov());
  }
  std::fprintf(f, "]}%s\n", last ? "" : ",");
}
template <class W>
static void put_expgamma(FILE* f, const char* name, double F, const std::vector<MatrixXd>& qZ, const std::vector<W>& weights,
                         std::vector<ExpGamma>& cl, bool last) {  // (getrate is not const in the reference)
  std::vector<double> N;
  for (const auto& c : cl) N.push_back(c.getN());
  put_common(f, name, F, (int)cl.size(), N, qZ, weights);
  std::fputs(", \"rates\": [", f);
  for (size_t k = 0; k < cl.size(); ++k) {
    if (k) std::fputs(", ", f);
    put_rowvec(f, cl[k].getrate());
  }
  std::fprintf(f, "]}%s\n", last ? "" : ",");
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cerr << "usage: ref_dump xcat_inputs.txt ref_xcat.json\n";
    return 2;
  }
  std::ifstream in(argv[1]);
  int J = 0;
  in >> J;
  vMatrixXd X((size_t)J);
  int Ntot = 0, D = 0;
  for (int j = 0; j < J; ++j) {
    int n = 0;
    in >> n >> D;
    X[(size_t)j].resize(n, D);
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < D; ++d) in >> X[(size_t)j](i, d);
    Ntot += n;
  }
  if (!in) {
    std::cerr << "could not read " << argv[1] << "\n";
    return 2;
  }
  MatrixXd Xcat(Ntot, D);
  for (int j = 0, r = 0; j < J; ++j) {
    Xcat.middleRows(r, X[(size_t)j].rows()) = X[(size_t)j];
    r += (int)X[(size_t)j].rows();
  }
  FILE* f = std::fopen(argv[2], "w");
  if (!f) return 2;
  std::fputs("{\n", f);
  {  // test/cluster_test.cpp:52-57 runs exactly this call
    MatrixXd qZ;
    Dirichlet w;
    std::vector<GaussWish> cl;
    const double F = learnBGMM(Xcat, qZ, w, cl, PRIORVAL, -1, false, 1);
    put_result(f, "learnBGMM", F, std::vector<MatrixXd>{qZ}, std::vector<Dirichlet>{w}, cl, false);
  }
  {
    MatrixXd qZ;
    StickBreak w;
    std::vector<GaussWish> cl;
    const double F = learnVDP(Xcat, qZ, w, cl, PRIORVAL, -1, false, 1);
    put_result(f, "learnVDP", F, std::vector<MatrixXd>{qZ}, std::vector<StickBreak>{w}, cl, false);
  }
  {
    MatrixXd qZ;
    StickBreak w(2.5);
    std::vector<GaussWish> cl;
    const double F = learnVDP(Xcat, qZ, w, cl, PRIORVAL, -1, false, 1);
    put_result(f, "learnVDP_conc2.5", F, std::vector<MatrixXd>{qZ}, std::vector<StickBreak>{w}, cl, false);
  }
  {
    MatrixXd qZ;
    Dirichlet w;
    std::vector<GaussWish> cl;
    const double F = learnBGMM(Xcat, qZ, w, cl, PRIORVAL, 1, false, 1);
    put_result(f, "learnBGMM_max1", F, std::vector<MatrixXd>{qZ}, std::vector<Dirichlet>{w}, cl, false);
  }
  {
    vMatrixXd qZ;
    std::vector<GDirichlet> w;
    std::vector<GaussWish> cl;
    const double F = learnGMC(X, qZ, w, cl, PRIORVAL, -1, false, false, 1);
    put_result(f, "learnGMC", F, qZ, w, cl, false);
  }
  {
    vMatrixXd qZ;
    std::vector<Dirichlet> w;
    std::vector<GaussWish> cl;
    const double F = learnSGMC(X, qZ, w, cl, PRIORVAL, -1, false, false, 1);
    put_result(f, "learnSGMC", F, qZ, w, cl, false);
  }
  // ---- the separable families on the same data (tests/golden/make_golden.py: learnDGMM on Xcat, learnDGMC on the groups;
  //      the exponential learners on |x| + 0.1); compare.py reads these against tests/golden/family_traces.json
  {
    MatrixXd qZ;
    Dirichlet w;
    std::vector<NormGamma> cl;
    const double F = learnDGMM(Xcat, qZ, w, cl, PRIORVAL, -1, false, 1);  // libcluster.h:262
    put_normgamma(f, "learnDGMM", F, std::vector<MatrixXd>{qZ}, std::vector<Dirichlet>{w}, cl, false);
  }
  {
    vMatrixXd qZ;
    std::vector<GDirichlet> w;
    std::vector<NormGamma> cl;
    const double F = learnDGMC(X, qZ, w, cl, PRIORVAL, -1, false, false, 1);  // libcluster.h:462
    put_normgamma(f, "learnDGMC", F, qZ, w, cl, false);
  }
  vMatrixXd Xpos(X.size());
  for (size_t j = 0; j < X.size(); ++j) Xpos[j] = (X[j].array().abs() + 0.1).matrix();
  const MatrixXd Xposcat = (Xcat.array().abs() + 0.1).matrix();
  {
    MatrixXd qZ;
    Dirichlet w;
    std::vector<ExpGamma> cl;
    const double F = learnBEMM(Xposcat, qZ, w, cl, PRIORVAL, -1, false, 1);  // libcluster.h:306
    put_expgamma(f, "learnBEMM", F, std::vector<MatrixXd>{qZ}, std::vector<Dirichlet>{w}, cl, false);
  }
  {
    vMatrixXd qZ;
    std::vector<GDirichlet> w;
    std::vector<ExpGamma> cl;
    const double F = learnEGMC(Xpos, qZ, w, cl, PRIORVAL, -1, false, false, 1);  // libcluster.h:513
    put_expgamma(f, "learnEGMC", F, qZ, w, cl, true);
  }
  std::fputs("}\n", f);
  std::fclose(f);
  std::cout << "wrote " << argv[2] << std::endl;
  return 0;
}
