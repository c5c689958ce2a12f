// Host driver of the variational Bayes loops, re-written around the device
// context: the M-step, free energy, pruning and the split heuristic stay on
// the host; every pass over the data goes through lcc::Context (HIP kernels).
//
// Restates src/cluster.cpp: vbem (177-239), prune_clusters (505-552),
// split_gr (366-495), cluster (564-629), learnVDP/BGMM/GMC (636-695, 763-784).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "lc_ctx.hpp"
#include "lc_host.hpp"

namespace lce {

struct Model {
  int wkind = lch::W_DIRICHLET;
  std::vector<lch::WeightState> weights;      // J
  int ckind = lch::C_GAUSSWISH;
  std::vector<lch::ClusterAny> clusters;      // K
  std::vector<double> LLk;                    // data term of the last E-step, per cluster
  // parameters of the last E-step: (A [K*D*D], m [K*D]) for Gauss-Wishart clusters, (a, w2, w1) [K*D each] in
  // lastA = [a | w2 | w1] for the diagonal families; c [J*K]
  std::vector<double> lastA, lastm, lastc;
};

// Sufficient statistics of one qZ: N_k [K], sum q x [K*D], second moments [K*XX] (XX = D*D / D / 0 by family), N_jk [J*K]
struct StatsBlock {
  int K = 0;
  std::vector<double> Nk, xs, xxs, Njk;
};

struct VbemOptions {
  double clusterprior = lch::PRIORVAL;
  int maxit = -1;
  bool sparse = false;
  bool verbose = false;
  int fixed_iters = -1;            // >= 0: run exactly this many iterations, no convergence / increase test
  std::vector<double>* trace = nullptr;  // F after every iteration
  unsigned nthreads = 1;
  // the E-step of every iteration also produces the split-ordering data term LL_k (model.LLk is then that of the LAST
  // iteration: cluster() needs no extra pass for it)
  bool want_ll = false;
  // first iteration: use these statistics instead of a pass over the data (they must be those of the current qZ) /
  // hand the first iteration's statistics back.  The split search uses both to recompute only the two changed columns
  // per candidate (cluster.cpp:473 prescribes a full vbem; the statistics are a pure function of (X, qZ)).
  const StatsBlock* preset = nullptr;
  StatsBlock* capture = nullptr;
  // first iteration: the context holds cached distances of every cluster but these (Context::estep_cached)
  const int* cached_changed = nullptr;
  int cached_nchanged = 0;
  // ... and the iteration after it (the second of a candidate's two, cluster.cpp:473 with the off-by-one of :235-236)
  // works on what that E-step MOVED: its statistics are those of the first iteration plus the change over the rows
  // whose responsibilities moved by more than delta_tol (Context::delta_suffstat; the ordinary pass when most rows
  // moved), and its E-step recomputes the distances of the clusters whose posterior differs in any bit from the cached
  // one (cache_A / cache_m: the cache_K whiteners and means the cache was built from)
  bool delta_second = false;
  double delta_tol = 0.0;
  const double* cache_A = nullptr;
  const double* cache_m = nullptr;
  int cache_K = 0;
  // first iteration: BUILD the cache from its own clusters (raw distances of all of them, then the normalisation
  // sweep -- the same responsibilities as the ordinary E-step) and hand back what it was built from; the second
  // iteration then works as above.  For the first candidate of a round (delta_second must be set too).
  bool build_cache = false;
  std::vector<double>* built_A = nullptr;
  std::vector<double>* built_m = nullptr;
};

// fn(c) for c in [0, nchunks) on the persistent worker pool (inline when the work is small or the pool is busy)
void parallel_chunks(int nchunks, unsigned nthreads, double work_per_chunk, const std::function<void(int)>& fn);

// cluster.cpp:177-239 on the context's current qZ.  Returns F.
double vbem(lcc::Context& ctx, Model& model, const VbemOptions& opt);

struct ClusterOptions {
  double clusterprior = lch::PRIORVAL;
  int maxclusters = -1;
  bool sparse = false;
  bool verbose = false;
  unsigned nthreads = 1;
  std::vector<std::pair<int, std::vector<double>>>* trace = nullptr;  // (K, F per iteration) per round
};

// cluster.cpp:564-629.  ctx must hold the data (host upload or device-resident); model.weights may be
// pre-seeded (learnVDP/learnBGMM pass the caller's weight prior in element 0).  The whole loop,
// split search included, runs against device-resident X and qZ.
double cluster(lcc::Context& ctx, Model& model, const ClusterOptions& opt);

}  // namespace lce
