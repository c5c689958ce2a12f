// Host driver of the two-level models: the "Simultaneous Clustering Model" (learnSCM, src/scluster.cpp) and the
// "Multiple-source Clustering Model" (learnMCM, src/mcluster.cpp).  Every "document" X[j][i] is one group of the
// device context, so the bottom-level E-step vbeZ (scluster.cpp:93-124, mcluster.cpp:100-135) IS the E-step
// kernel with a per-document constant table, and the bottom-level M-step statistics and the per-document
// counts N_jik come out of the suff-stat pass.  The document-level quantities (qY, I_tot x T; W, I_tot x Dt) are
// small and stay on the host.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "lc_ctx.hpp"
#include "lc_host.hpp"

namespace lce {

struct TopicData {
  int J = 0;                   // groups
  std::vector<int> Ij;         // documents per group; the context's groups are the documents, group-major
  std::vector<int> doc_group;  // I_tot: group of every document
  int Itot = 0;
  const double* W = nullptr;   // MCM: I_tot x Dt row-major document observations (nullptr: SCM)
  int Dt = 0;
};

struct TopicModel {
  std::vector<lch::WeightState> weights_j;       // J   GDirichlet
  std::vector<lch::WeightState> weights_t;       // T   Dirichlet
  std::vector<lch::GaussWishState> clusters_t;   // T   (MCM only)
  std::vector<lch::GaussWishState> clusters;     // K
  std::vector<double> qY;                        // I_tot x T row-major
  int T = 0;
  // the last bottom-level E-step, for the split ordering's data term
  std::vector<double> lastA, lastm, lastc, cst;
};

struct TopicOptions {
  double prior_t = lch::PRIORVAL;  // SCM: Dirichlet alpha of weights_t; MCM: width of the top-level Gaussians
  double prior_k = lch::PRIORVAL;
  int maxit = -1;
  int fixed_iters = -1;
  int maxK = -1;
  bool verbose = false;
  unsigned nthreads = 1;
  std::vector<double>* trace = nullptr;
};

struct TopicRound {
  int T, K;
  std::vector<double> F;
};

// scluster.cpp:172-260 / mcluster.cpp:186-287 on the context's current qZ and model.qY (both updated).
double topic_vbem(lcc::Context& ctx, const TopicData& data, TopicModel& model, const TopicOptions& opt);

// scluster.cpp:493-570 / mcluster.cpp:525-605.  model.qY holds the initial (I_tot x maxT) assignment.
double topic_cluster(lcc::Context& ctx, const TopicData& data, TopicModel& model, const TopicOptions& opt,
                     std::vector<TopicRound>* rounds);

}  // namespace lce
