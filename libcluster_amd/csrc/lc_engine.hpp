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

// Sufficient statistics of one qZ: N_k [K], sum q x [K*D], second moments [K*XX] (XX = D*D / D / 0 by family), N_jk [J*K]
struct StatsBlock {
  int K = 0;
  std::vector<double> Nk, xs, xxs, Njk;
  int chain = 0;  // updates by moved rows (Context::delta_suffstat) since the last full pass over the data
  // vbem()'s capture_final: these are the statistics of the responsibilities BEFORE its last E-step, and the context
  // still holds that E-step's move -- finish_stats() adds it (only the caller knows whether it will be needed)
  bool pending = false;
};

// cluster(): the E-steps go through the context's distance cache and the statistics follow the rows an E-step moved
// (VbemOptions::inc).  `on` is cleared when that stops paying (most clusters' posteriors change between E-steps).
struct IncState {
  bool on = true;
  double tol = 0x1p-50;  // a row whose responsibilities all moved by <= tol does not count as moved (LC_SPLIT_DELTA_TOL)
  bool synced = false;   // the previous E-step went through the cache (its tags describe the model as of then)
  int bad = 0;           // E-steps in a row, each right after a synced one, that found most columns stale
  bool no_room = false;  // the device ran out of room for the cache: off for the rest of this cluster() call
};

struct Model {
  int wkind = lch::W_DIRICHLET;
  std::vector<lch::WeightState> weights;      // J
  int ckind = lch::C_GAUSSWISH;
  std::vector<lch::ClusterAny> clusters;      // K
  std::vector<double> LLk;                    // data term of the last E-step, per cluster
  // parameters of the last E-step: (A [K*D*D], m [K*D]) for Gauss-Wishart clusters, (a, w2, w1) [K*D each] in
  // lastA = [a | w2 | w1] for the diagonal families; c [J*K]
  std::vector<double> lastA, lastm, lastc;
  // cluster(): statistics of the converged responsibilities of the round (K = 0: not known) / of the accepted
  // candidate's responsibilities the next round starts from
  StatsBlock final_stats, next_stats;
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
  // Model selection on cached distances (Gauss-Wishart, dense): every E-step goes through Context::estep_cache --
  // only the clusters whose posterior changed in any bit since their distances were last computed are recomputed --
  // and leaves how far it moved the responsibilities; the next iteration's statistics are then the previous ones plus
  // the change over the rows that moved by more than inc->tol (Context::delta_suffstat; the ordinary pass when most
  // rows moved or the chain of such updates gets long).  cluster.cpp prescribes full passes; the statistics are a
  // pure, linear function of (X, qZ) and the distances a pure function of (X, cluster posterior).
  IncState* inc = nullptr;
  // statistics of the responsibilities the LAST E-step produced, when they can be had from the moved rows (K = 0
  // otherwise; left `pending`, see StatsBlock): the split search starts from them, and an accepted candidate hands
  // them to the next round
  StatsBlock* capture_final = nullptr;
};

// completes a block vbem() left pending (no other pass may have touched the context since); false and K = 0 when the
// statistics cannot be had from the moved rows
bool finish_stats(lcc::Context& ctx, StatsBlock& s);

// fn(c) for c in [0, nchunks) on the persistent worker pool (inline when the work is small or the pool is busy)
void parallel_chunks(int nchunks, unsigned nthreads, double work_per_chunk, const std::function<void(int)>& fn);

// cluster.cpp:177-239 on the context's current qZ.  Returns F.
double vbem(lcc::Context& ctx, Model& model, const VbemOptions& opt);

// cluster.cpp:505-552: drop the clusters with fewer than ZEROCUTOFF observations (model, qZ columns), then update the
// weights with the remaining columns' sums (no renormalisation, as the reference).  True when something was removed.
bool prune_clusters(lcc::Context& ctx, Model& model, bool verbose);

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
