#include "lc_engine.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <exception>
#include <iostream>
#include <limits>
#include <memory>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace lce {

using lch::ClusterAny;
using lch::WeightState;

namespace {

// A small persistent pool for the M-step's loop over clusters (the reference's "omp parallel for",
// cluster.cpp:215-217).  Spawning 32 std::threads per iteration cost ~1 ms of a 50 ms iteration; the pool's
// workers sleep on a condition variable between calls.  Items are handed out one by one from an atomic counter and the
// caller starts on them at once, so workers that wake late (waking 31 sleepers takes longer than a D = 64 M-step)
// simply find less, or nothing, left -- the call returns when the ITEMS are done, not when every worker has reported.
// One parallel_for runs at a time (callers on other host threads fall back to running their loop inline).
class Pool {
 public:
  // one pool per calling host thread: with LIBCLUSTER_GPUS every shard's thread runs its own (replicated) M-step
  static Pool& get() {
    static thread_local Pool p;
    return p;
  }
  template <typename F>
  bool run(int n, unsigned nt, F& fn) {
    std::unique_lock<std::mutex> busy(busy_, std::try_to_lock);
    if (!busy.owns_lock()) return false;
    grow(nt - 1);
    auto job = std::make_shared<Job>();
    job->n = n;
    job->call = [&fn](int k) { fn(k); };
    {
      std::lock_guard<std::mutex> g(m_);
      cur_ = job;
      nworkers_ = nt - 1;
      ++gen_;
    }
    cv_.notify_all();
    work(*job);
    {
      std::unique_lock<std::mutex> g(job->dm);
      job->dcv.wait(g, [&] { return job->done.load(std::memory_order_acquire) >= n; });
    }
    // (a worker that wakes from now on sees next >= n and never touches `call`, whose captures die with this frame)
    if (job->err) std::rethrow_exception(job->err);
    return true;
  }

 private:
  struct Job {
    int n = 0;
    std::function<void(int)> call;
    std::atomic<int> next{0}, done{0};
    std::mutex dm;
    std::condition_variable dcv;
    std::exception_ptr err;  // the first one (guarded by dm)
  };
  static void work(Job& j) {
    for (;;) {
      const int k = j.next.fetch_add(1, std::memory_order_relaxed);
      if (k >= j.n) return;
      try {
        j.call(k);
      } catch (...) {
        std::lock_guard<std::mutex> g(j.dm);
        if (!j.err) j.err = std::current_exception();
      }
      if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 >= j.n) {
        std::lock_guard<std::mutex> g(j.dm);
        j.dcv.notify_all();
      }
    }
  }
  Pool() = default;
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void grow(unsigned want) {
    while (th_.size() < want) {
      const unsigned id = (unsigned)th_.size();
      uint64_t seen;
      {
        std::lock_guard<std::mutex> g(m_);
        seen = gen_;
      }
      th_.emplace_back([this, id, seen]() mutable {
        for (;;) {
          std::shared_ptr<Job> job;
          {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (id >= nworkers_) continue;  // not needed for this job
            job = cur_;
          }
          work(*job);
        }
      });
    }
  }
  std::mutex busy_, m_;
  std::condition_variable cv_;
  std::vector<std::thread> th_;
  std::shared_ptr<Job> cur_;
  unsigned nworkers_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// environment switch: set and not "0" / empty
bool env_on(const char* name) {
  const char* e = std::getenv(name);
  return e && *e && !(e[0] == '0' && e[1] == 0);
}
// the same for the switches of the test suite (the split search's literal schedule): libcluster_hip_testhooks.so only
bool test_on(const char* name) {
  const char* e = lck::test_switch(name);
  return e && *e && !(e[0] == '0' && e[1] == 0);
}

// Run fn(k) for k in [0,n) on up to nthreads host threads.  Exceptions are collected and the first one
// re-thrown on the calling thread.
template <typename F>
void parallel_for(int n, unsigned nthreads, double work_per_item, F fn) {
  unsigned nt = std::min<unsigned>(nthreads, (unsigned)std::max(n, 1));
  if (nt > 64) nt = 64;
  if (nt <= 1 || work_per_item * n < 2e6 || !Pool::get().run(n, nt, fn)) {
    for (int k = 0; k < n; ++k) fn(k);
  }
}

bool anyempty(const std::vector<ClusterAny>& c) {  // src/comutils.h:114-123
  for (const auto& x : c)
    if (x.N() <= 1) return true;
  return false;
}

// E-step of the model's family with the parameters stored in the model (vbem's last iteration)
void run_estep(lcc::Context& ctx, Model& model, int K, double* Fz, double* LLk) {
  const int D = ctx.D();
  if (model.ckind == lch::C_GAUSSWISH) {
    ctx.estep(K, model.lastA.data(), model.lastm.data(), model.lastc.data(), Fz, LLk);
  } else {
    const double* a = model.lastA.data();
    ctx.estep_diag(K, a, a + (size_t)K * D, a + (size_t)2 * K * D, model.lastc.data(), Fz, LLk);
  }
}

struct GreedOrder {  // src/comutils.h:44-49
  int k;
  int tally;
  double Fk;
};
bool greedcomp(const GreedOrder& i, const GreedOrder& j) {  // src/comutils.h:60-68
  if (i.tally == j.tally) return i.Fk > j.Fk;
  return i.tally < j.tally;
}

}  // namespace

void parallel_chunks(int nchunks, unsigned nthreads, double work_per_chunk, const std::function<void(int)>& fn) {
  parallel_for(nchunks, nthreads, work_per_chunk, [&](int c) { fn(c); });
}

// ---------------------------------------------------------------------------
// cluster.cpp:177-239
// ---------------------------------------------------------------------------
// statistics updates from moved rows in a row before a full pass refreshes them
constexpr int CHAIN_CAP = 64;

double vbem(lcc::Context& ctx, Model& model, const VbemOptions& opt) {
  const int J = ctx.J(), K = ctx.K(), D = ctx.D();
  if (K < 1) throw std::invalid_argument("qZ must have at least one column");

  // weights.resize(J, W()); clusters.resize(K, C(clusterprior, D))  (:192-193)
  const int ck = model.ckind;
  const bool full = ck == lch::C_GAUSSWISH;
  if ((int)model.weights.size() > J) model.weights.resize(J);
  while ((int)model.weights.size() < J) model.weights.emplace_back(model.wkind, lch::ALPHA1PRIOR);
  if ((int)model.clusters.size() > K) model.clusters.resize(K);
  while ((int)model.clusters.size() < K) model.clusters.emplace_back(ck, opt.clusterprior, D);
  for (const auto& c : model.clusters)
    if (c.D() != D || c.kind != ck) throw std::invalid_argument("Mismatched dims. of cluster params and obs.!");

  const size_t XX = ClusterAny::xx_size(ck, D);
  std::vector<double> Nk(K), xs((size_t)K * D), xxs((size_t)K * std::max<size_t>(XX, 1)), Njk((size_t)J * K);
  // the packed E-step parameters of the last iteration stay in the model (the split search
  // re-runs that E-step once to get its data term, see data_loglik)
  std::vector<double>&A = model.lastA, &m = model.lastm, &c = model.lastc;
  A.assign(full ? (size_t)K * D * D : (size_t)3 * K * D, 0.0);
  m.assign(full ? (size_t)K * D : 0, 0.0);
  c.assign((size_t)J * K, 0.0);
  std::vector<double> cst(K);
  std::vector<unsigned char> mask;
  model.LLk.assign(K, 0.0);
  std::vector<double> nNk, nxs, nxxs, nNjk;  // statistics of the responsibilities the last (fused) E-step produced
  bool have_next = false;
  // model selection on cached distances (VbemOptions::inc): the last E-step left how far it moved the responsibilities;
  // `chain` = statistics updates from moved rows since the last full pass
  bool have_delta = false;
  int chain = 0;
  // LC_SPLIT_DELTA_FORCE=1 (tests): stay on the cache and on moved-row statistics however much moves
  static const bool inc_force = test_on("LC_SPLIT_DELTA_FORCE");

  double F = std::numeric_limits<double>::max(), Fold;
  int i = 0, done = 0;
  bool again;
  // LC_TRACE_PHASES=1: wall time per phase of every iteration on stderr (tuning aid)
  static const bool trace_phases = env_on("LC_TRACE_PHASES");
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  // statistics += their change over the rows the last E-step moved; false (nothing touched) when most rows moved
  auto add_delta = [&]() {
    std::vector<double> dN(K), dx((size_t)K * D), dxx((size_t)K * XX), dNj((size_t)J * K);
    const bool okd = ctx.delta_suffstat(K, inc_force ? 2.0 : 0.5, dN.data(), dx.data(), dxx.data(), dNj.data());
    if (trace_phases)
      std::cerr << "[vbem] statistics from " << ctx.delta_rows() << " moved rows" << (okd ? "" : " (too many: full pass)")
                << std::endl;
    if (!okd) return false;
    for (int k = 0; k < K; ++k) Nk[k] += dN[k];
    for (size_t t = 0; t < dx.size(); ++t) xs[t] += dx[t];
    for (size_t t = 0; t < dxx.size(); ++t) xxs[t] += dxx[t];
    for (size_t t = 0; t < dNj.size(); ++t) Njk[t] += dNj[t];
    ++chain;
    return true;
  };
  do {
    Fold = F;
    const auto t0 = now();
    for (auto& cl : model.clusters) cl.clearobs();  // :203-204

    // updateSS (:53-82) for all groups + weights update (:207-212)
    const unsigned char* maskp = nullptr;
    if (opt.sparse) {
      ctx.colsums(Njk.data());
      mask.resize((size_t)J * K);
      for (size_t t = 0; t < mask.size(); ++t) mask[t] = Njk[t] >= lch::ZEROCUTOFF ? 1 : 0;
      maskp = mask.data();
    }
    if (have_next) {
      // the fused pass of the previous iteration already produced the statistics of its responsibilities
      Nk.swap(nNk);
      xs.swap(nxs);
      xxs.swap(nxxs);
      Njk.swap(nNjk);
      have_next = false;
    } else if (done == 0 && opt.preset && opt.preset->K == K && !opt.sparse) {
      Nk = opt.preset->Nk;
      xs = opt.preset->xs;
      xxs = opt.preset->xxs;
      xxs.resize((size_t)K * std::max<size_t>(XX, 1));
      Njk = opt.preset->Njk;
      chain = opt.preset->chain;
    } else if (have_delta && chain < CHAIN_CAP && add_delta()) {
      // Nk, xs, xxs, Njk held the statistics of the responsibilities the last E-step overwrote
    } else if (full) {
      ctx.suffstat(maskp, Nk.data(), xs.data(), xxs.data(), Njk.data());
      chain = 0;
    } else ctx.suffstat_diag(maskp, Nk.data(), xs.data(), XX ? xxs.data() : nullptr, Njk.data());
    have_delta = false;
    if (done == 0 && opt.capture) {
      opt.capture->K = K;
      opt.capture->Nk = Nk;
      opt.capture->xs = xs;
      opt.capture->xxs = xxs;
      opt.capture->Njk = Njk;
      opt.capture->chain = chain;
    }
    for (int j = 0; j < J; ++j) model.weights[j].update(Njk.data() + (size_t)j * K, K);
    const auto t1 = now();

    // VBM for clusters (:215-217) + the per-cluster constants of the E-step
    parallel_for(K, opt.nthreads, full ? 2.0 * D * D * D : 8.0 * D, [&](int k) {
      ClusterAny& cl = model.clusters[k];
      cl.addstats(Nk[k], xs.data() + (size_t)k * D, xxs.data() + (size_t)k * XX);
      cl.update();
      if (full) {
        const std::vector<double> Ak = cl.gw.whitener();
        std::copy(Ak.begin(), Ak.end(), A.begin() + (size_t)k * D * D);
        std::copy(cl.gw.m.begin(), cl.gw.m.end(), m.begin() + (size_t)k * D);
      } else {
        double* pa = A.data() + (size_t)k * D;              // a
        double* pw2 = A.data() + (size_t)(K + k) * D;       // w2
        double* pw1 = A.data() + (size_t)(2 * K + k) * D;   // w1
        for (int d = 0; d < D; ++d) {
          if (ck == lch::C_NORMGAMMA) {  // -nu/2 * (x-m)^2 / L   (distributions.cpp:486-491)
            pa[d] = cl.ng.m[d];
            pw2[d] = -0.5 * cl.ng.nu / cl.ng.L[d];
            pw1[d] = 0.0;
          } else {  // -a * x * ib   (distributions.cpp:570-571)
            pa[d] = 0.0;
            pw2[d] = 0.0;
            pw1[d] = -cl.eg.a * cl.eg.ib[d];
          }
        }
      }
      cst[k] = cl.eloglike_const();
    });

    // VBE (:220-223): c_jk = E_logZ_j(k) + const_k ; sparse-inactive clusters get -inf (:109-112, 134-135)
    for (int j = 0; j < J; ++j) {
      const WeightState& w = model.weights[j];
      for (int k = 0; k < K; ++k) {
        double v = w.Elogpi[k] + cst[k];
        if (opt.sparse && !(w.Nk[k] >= lch::ZEROCUTOFF)) v = -std::numeric_limits<double>::infinity();
        c[(size_t)j * K + k] = v;
      }
    }
    double Fz = 0.0;
    const auto t2 = now();
    // fenergy (:145-165), the terms that depend on the posteriors only: computed on this thread while the E-step runs on
    // the device (the context calls the hook between its last launch and its wait; the M-step's pool is idle then)
    double Fw = 0.0, Fc = 0.0, fe_ms = 0.0;
    std::vector<double> fck(K);
    auto posterior_terms = [&]() {
      const auto ta = now();
      for (const auto& w : model.weights) Fw += w.fenergy();
      parallel_for(K, opt.nthreads, full ? 1.0 * D * D : 8.0 * D, [&](int k) { fck[k] = model.clusters[k].fenergy(); });
      for (int k = 0; k < K; ++k) Fc += fck[k];
      fe_ms = ms(ta, now());
    };
    ctx.set_overlap(posterior_terms);
    struct ClearHook {  // (the hook refers to this frame: it must not outlive it, whatever the E-step throws)
      lcc::Context& c;
      ~ClearHook() { c.set_overlap(nullptr); }
    } clear_hook{ctx};
    // small observations: the E-step and the statistics the NEXT iteration starts from in one pass (updateSS runs at
    // the top of the next iteration on exactly these responsibilities, cluster.cpp:198-212)
    if (full && !opt.sparse) {
      nNk.resize(K);
      nxs.resize((size_t)K * D);
      nxxs.resize((size_t)K * std::max<size_t>(XX, 1));
      nNjk.resize((size_t)J * K);
      have_next = ctx.estep_suffstat_fused(K, A.data(), m.data(), c.data(), &Fz, opt.want_ll ? model.LLk.data() : nullptr,
                                           nNk.data(), nxs.data(), nxxs.data(), nNjk.data());
    }
    if (have_next) {
    } else if (opt.inc && opt.inc->on && full && !opt.sparse && opt.fixed_iters < 0 && ctx.dcache_eligible(K)) {
      int stale = 0, nre = -1;
      try {
        nre = ctx.estep_cache(K, A.data(), m.data(), c.data(), &Fz, opt.want_ll ? model.LLk.data() : nullptr, opt.inc->tol,
                              &stale);
      } catch (const lcc::CacheNoRoom& e) {
        // (the room check and the journal reservation are agreed on by all ranks: every rank gets here together)
        if (trace_phases) std::cerr << "[vbem] distance cache given up, ordinary E-step from here: " << e.what() << std::endl;
        ctx.dcache_release();
        opt.inc->on = false;
        opt.inc->no_room = true;
        opt.inc->synced = false;
        run_estep(ctx, model, K, &Fz, opt.want_ll ? model.LLk.data() : nullptr);
      }
      if (nre >= 0) {
        if (trace_phases)
          std::cerr << "[vbem] cached E-step: K " << K << ", recomputed " << nre << " (stale " << stale << ")" << std::endl;
        have_delta = ctx.delta_pending() == K;
        // most clusters' posteriors moved from one E-step to the next, twice in a row: the cache costs more than it
        // saves (once is what a full refresh of the statistics looks like, and what the first E-step after the cache
        // was left alone for a while looks like)
        if (opt.inc->synced && K >= 4 && stale * 10 > K * 7) ++opt.inc->bad;
        else opt.inc->bad = 0;
        opt.inc->synced = true;
        if (opt.inc->bad >= 2 && !inc_force) opt.inc->on = false;
      }
    } else {
      run_estep(ctx, model, K, &Fz, opt.want_ll ? model.LLk.data() : nullptr);
      if (opt.inc) opt.inc->synced = false;
    }
    const auto t3 = now();

    if (ctx.overlap_pending()) ctx.run_overlap();  // (an E-step path without the hook, or one that threw CacheNoRoom first)
    if (ctx.group_sharded()) Fw = ctx.allreduce_value(Fw);  // other ranks hold the other groups' weights
    F = Fc + Fw + Fz;
    if (opt.trace) opt.trace->push_back(F);
    if (trace_phases)
      std::cerr << "[vbem] suffstat+weights " << ms(t0, t1) << " ms, M-step+pack " << ms(t1, t2) << " ms, E-step "
                << ms(t2, t3) << " ms (free energy terms " << fe_ms << " ms under it), fenergy " << ms(t3, now()) << " ms"
                << std::endl;
    if (ctx.timing_enabled()) ctx.timing_host_phases(ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    ++done;

    if (opt.fixed_iters >= 0) {
      again = done < opt.fixed_iters;
    } else {
      if ((F - Fold) / std::abs(Fold) > lch::FENGYDEL) throw std::runtime_error("Free energy increase!");  // :229-230
      if (opt.verbose) std::cout << '-' << std::flush;
      again = (std::abs((Fold - F) / Fold) > lch::CONVERGE) && ((i++ < opt.maxit) || (opt.maxit < 0));  // :235-236
    }
  } while (again);
  if (opt.capture_final) {
    opt.capture_final->K = 0;
    if (have_delta && chain < CHAIN_CAP) {
      opt.capture_final->K = K;
      opt.capture_final->Nk = Nk;
      opt.capture_final->xs = xs;
      opt.capture_final->xxs = xxs;
      opt.capture_final->Njk = Njk;
      opt.capture_final->chain = chain;
      opt.capture_final->pending = true;
    }
  }
  return F;
}

bool finish_stats(lcc::Context& ctx, StatsBlock& s) {
  if (s.K < 1) return false;
  if (!s.pending) return true;
  s.pending = false;
  static const bool inc_force = test_on("LC_SPLIT_DELTA_FORCE");
  const int K = s.K;
  std::vector<double> dN(K), dx(s.xs.size()), dxx(s.xxs.size()), dNj(s.Njk.size());
  if (ctx.delta_pending() != K || s.chain >= CHAIN_CAP ||
      !ctx.delta_suffstat(K, inc_force ? 2.0 : 0.5, dN.data(), dx.data(), dxx.data(), dNj.data())) {
    s.K = 0;
    return false;
  }
  for (int k = 0; k < K; ++k) s.Nk[k] += dN[k];
  for (size_t t = 0; t < dx.size(); ++t) s.xs[t] += dx[t];
  for (size_t t = 0; t < dxx.size(); ++t) s.xxs[t] += dxx[t];
  for (size_t t = 0; t < dNj.size(); ++t) s.Njk[t] += dNj[t];
  ++s.chain;
  return true;
}

// Data term of the split ordering (cluster.cpp:407-410): LLk[k] = sum_n q_nk (log q~_nk - c_jk).
// vbem's iterations skip it; it is produced here by repeating the last E-step (same parameters,
// so qZ is rewritten with identical values) with the per-cluster reduction switched on.
static void data_loglik(lcc::Context& ctx, Model& model) {
  const int K = (int)model.clusters.size();
  double Fz = 0.0;
  model.LLk.assign(K, 0.0);
  run_estep(ctx, model, K, &Fz, model.LLk.data());
}

// ---------------------------------------------------------------------------
// cluster.cpp:505-552
// ---------------------------------------------------------------------------
bool prune_clusters(lcc::Context& ctx, Model& model, bool verbose) {
  const int K = (int)model.clusters.size(), J = ctx.J();
  std::vector<int> keep;
  for (int k = 0; k < K; ++k)
    if (!(model.clusters[k].N() < lch::ZEROCUTOFF)) keep.push_back(k);
  if ((int)keep.size() == K) return false;
  if (verbose) std::cout << '*' << std::flush;
  static const bool trace_phases = env_on("LC_TRACE_PHASES");
  if (trace_phases) std::cerr << "[prune] " << keep.size() << " of " << K << " clusters kept" << std::endl;
  if (keep.empty()) throw std::runtime_error("all clusters are empty");
  std::vector<ClusterAny> nc;
  std::vector<double> nll;
  for (int k : keep) {
    nc.push_back(std::move(model.clusters[k]));
    nll.push_back(k < (int)model.LLk.size() ? model.LLk[k] : 0.0);
  }
  model.clusters.swap(nc);
  model.LLk.swap(nll);
  ctx.qz_keep_columns(keep);
  ctx.dcache_invalidate();  // (rare: the next E-step recomputes every column)
  model.final_stats.K = 0;  // (their columns no longer line up)
  const int nK = (int)keep.size();
  std::vector<double> Njk((size_t)J * nK);
  ctx.colsums(Njk.data());
  for (int j = 0; j < J; ++j) model.weights[j].update(Njk.data() + (size_t)j * nK, nK);  // :546
  return true;
}

// ---------------------------------------------------------------------------
// cluster.cpp:366-495
// ---------------------------------------------------------------------------
static bool split_gr(lcc::Context& ctx, Model& model, std::vector<int>& tally, double F, const ClusterOptions& opt,
                     IncState* inc) {
  const int J = ctx.J(), K = (int)model.clusters.size(), D = ctx.D();
  if (K >= opt.maxclusters && opt.maxclusters >= 0) return false;
  tally.resize(K, 0);

  // cluster free energies and data likelihoods (:391-415).  The data term
  // sum_n q_nk Eloglike_k(x_n) = const_k * sum_n q_nk + LLk (from the last E-step).
  std::vector<double> Njk((size_t)J * K);
  ctx.colsums(Njk.data());
  std::vector<GreedOrder> ord(K);
  std::vector<double> wt(K);
  for (int k = 0; k < K; ++k) {
    ord[k].k = k;
    ord[k].tally = tally[k];
    ord[k].Fk = model.clusters[k].fenergy();
    const double cst = model.clusters[k].eloglike_const();
    double wterm = 0.0;
    for (int j = 0; j < J; ++j) wterm += (model.weights[j].Elogpi[k] + cst) * Njk[(size_t)j * K + k];
    wt[k] = wterm;
  }
  if (ctx.group_sharded()) ctx.allreduce_values(wt.data(), K);  // sum over the groups of all ranks
  for (int k = 0; k < K; ++k) ord[k].Fk -= wt[k] + model.LLk[k];
  std::sort(ord.begin(), ord.end(), greedcomp);  // :418

  const double prior = model.clusters[0].prior();
  lcc::RowSelection sel;
  std::vector<double> njs;
  std::vector<double> eigv;
  // Statistics of the round's converged qZ (K columns), known after the first candidate's full-data iteration: a
  // candidate changes two columns of qZ (auglabels moves mass from column k to the new column K), so every later
  // candidate recomputes those two only.  Not in sparse mode (the masks depend on all columns).
  static const bool no_incremental = test_on("LC_SPLIT_FULL_STATS");
  // ... known already when the round's VBEM could follow the rows its last E-step moved (cluster())
  StatsBlock round_stats;
  if (model.final_stats.K == K && !opt.sparse && !no_incremental) round_stats = model.final_stats;
  const size_t XX = ClusterAny::xx_size(model.ckind, D), XS = std::max<size_t>(XX, 1);
  // Gauss-Wishart, dense: a candidate's E-steps go through the context's distance cache IN PLACE (only the split
  // cluster's two halves, and whatever else its two iterations move, are recomputed); a journal puts the cache back
  // when the candidate is rejected
  const bool cache_trials = inc && model.ckind == lch::C_GAUSSWISH && !opt.sparse;

  static const bool trace_phases = env_on("LC_TRACE_PHASES");
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto msec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  for (const GreedOrder& o : ord) {
    const int k = o.k;
    ++tally[k];
    if (model.clusters[k].N() < 4) continue;  // :432
    const auto t0 = now();

    // partobs + splitobs per group (:438-453), on the device: ordered compaction of the rows with
    // q_k > 0.5, device-to-device gather into a fresh context, projection on the principal axis
    // Row-sharded runs: every rank gathers its own selected rows; the counts that steer the
    // control flow (Mtot, scount, N_k) are all-reduced, so all ranks take the same branches.
    ctx.select_rows(k, 0.5, sel);
    const int64_t Mtot = (int64_t)std::llround(ctx.allreduce_value((double)sel.M));
    lcc::Context sub(ctx.device(), ctx.stream());
    sub.inherit_comm(ctx);
    sub.set_data_gather(ctx, sel);
    {
      const ClusterAny& cl = model.clusters[k];
      if (cl.kind == lch::C_GAUSSWISH) {  // project on the principal axis of iW (distributions.cpp:373-385)
        lch::eigpower(cl.gw.iW, D, eigv);
        sub.qz_init_split(cl.gw.m.data(), eigv.data());
      } else if (cl.kind == lch::C_NORMGAMMA) {  // threshold the widest dimension at its mean (:495-505)
        eigv.assign(D, 0.0);
        eigv[(size_t)cl.ng.split_axis()] = 1.0;
        sub.qz_init_split(cl.ng.m.data(), eigv.data());
      } else {  // rate-weighted sum against its mean over the group's rows (:575-581)
        eigv.resize(D);
        for (int d = 0; d < D; ++d) eigv[(size_t)d] = cl.eg.a * cl.eg.ib[(size_t)d];
        sub.qz_init_split_mean(eigv.data());
      }
    }
    const auto t1 = now();
    njs.assign((size_t)J * 2, 0.0);
    sub.colsums(njs.data());
    double sc = 0.0;
    for (int j = 0; j < J; ++j) sc += njs[(size_t)j * 2];
    if (ctx.group_sharded()) sc = ctx.allreduce_value(sc);
    const int64_t scount = (int64_t)std::llround(sc);
    if (scount < 2 || scount > Mtot - 2) continue;  // :456

    // refine the split on the selected observations (:459-462)
    Model ms;
    ms.wkind = model.wkind;
    ms.ckind = model.ckind;
    {
      VbemOptions vo;
      vo.clusterprior = prior;
      vo.maxit = (int)lch::SPLITITER;
      vo.sparse = opt.sparse;
      vo.nthreads = opt.nthreads;
      vbem(sub, ms, vo);
      if (anyempty(ms.clusters)) continue;  // :464
    }
    const auto t2 = now();

    // auglabels (:468-470, comutils.cpp:75-104) on a copy of qZ
    ctx.qz_clone_to_alt();
    ctx.qz_swap_alt();
    ctx.qz_split_from(sub, sel, k);
    const auto t3 = now();

    // free energy of the split with all data (:473)
    VbemOptions vo;
    vo.clusterprior = prior;
    vo.maxit = 1;
    vo.sparse = opt.sparse;
    vo.nthreads = opt.nthreads;
    StatsBlock first;  // statistics of the augmented qZ (K + 1 columns) for the first of the two iterations
    StatsBlock fin;    // ... and of the responsibilities its last E-step leaves (next round's start when it is accepted)
    const bool incremental = !opt.sparse && !no_incremental;
    double Fsplit;
    bool journal = false;
    try {
      if (incremental && round_stats.K == K) {
        // auglabels moved the mass of the selected rows labelled for the new cluster from column k to column K
        // (comutils.cpp:75-104): the statistics of that mass, T, come from the sub-problem's own rows (1/K of the
        // data, one column); column K's are T, column k's are the round's minus T, the others are the round's
        std::vector<double> n1(1), x1((size_t)D), xx1(XS), nj1((size_t)J);
        sub.qz_gather_column(ctx, sel, K);
        if (model.ckind == lch::C_GAUSSWISH) sub.suffstat(nullptr, n1.data(), x1.data(), xx1.data(), nj1.data());
        else sub.suffstat_diag(nullptr, n1.data(), x1.data(), XX ? xx1.data() : nullptr, nj1.data());
        first.K = K + 1;
        first.chain = round_stats.chain;
        first.Nk.assign(round_stats.Nk.begin(), round_stats.Nk.end());
        first.Nk.push_back(n1[0]);
        first.Nk[(size_t)k] = std::max(0.0, first.Nk[(size_t)k] - n1[0]);
        first.xs.assign(round_stats.xs.begin(), round_stats.xs.end());
        first.xs.resize((size_t)(K + 1) * D, 0.0);
        first.xxs.assign(round_stats.xxs.begin(), round_stats.xxs.begin() + (size_t)K * XS);
        first.xxs.resize((size_t)(K + 1) * XS, 0.0);
        for (int d = 0; d < D; ++d) {
          first.xs[(size_t)K * D + d] = x1[(size_t)d];
          first.xs[(size_t)k * D + d] -= x1[(size_t)d];
        }
        for (size_t e = 0; e < XX; ++e) {
          first.xxs[(size_t)K * XX + e] = xx1[e];
          first.xxs[(size_t)k * XX + e] -= xx1[e];
        }
        first.Njk.assign((size_t)J * (K + 1), 0.0);
        for (int j = 0; j < J; ++j) {
          for (int c = 0; c < K; ++c) first.Njk[(size_t)j * (K + 1) + c] = round_stats.Njk[(size_t)j * K + c];
          first.Njk[(size_t)j * (K + 1) + K] = nj1[(size_t)j];
          first.Njk[(size_t)j * (K + 1) + k] = std::max(0.0, first.Njk[(size_t)j * (K + 1) + k] - nj1[(size_t)j]);
        }
        vo.preset = &first;
      } else if (incremental) {
        vo.capture = &first;
      }
      if (cache_trials && inc->on) {
        ctx.dcache_journal_begin();
        journal = true;
        vo.inc = inc;
        vo.capture_final = &fin;
      }
      Fsplit = vbem(ctx, ms, vo);
    } catch (...) {
      if (journal) ctx.dcache_rollback();
      ctx.qz_swap_alt();
      throw;
    }
    if (vo.capture && first.K == K + 1) {
      // the converged qZ's statistics from this candidate's: columns other than k are untouched, column k gave its
      // moved mass to column K (q_t[:,k] = q_aug[:,k] + q_aug[:,K] row by row, and the statistics are linear in q)
      round_stats.K = K;
      round_stats.chain = first.chain;
      round_stats.Nk.assign(first.Nk.begin(), first.Nk.begin() + K);
      round_stats.xs.assign(first.xs.begin(), first.xs.begin() + (size_t)K * D);
      round_stats.xxs.assign(first.xxs.begin(), first.xxs.begin() + (size_t)K * XS);
      round_stats.Njk.assign((size_t)J * K, 0.0);
      round_stats.Nk[(size_t)k] += first.Nk[(size_t)K];
      for (int d = 0; d < D; ++d) round_stats.xs[(size_t)k * D + d] += first.xs[(size_t)K * D + d];
      for (size_t e = 0; e < XX; ++e) round_stats.xxs[(size_t)k * XX + e] += first.xxs[(size_t)K * XX + e];
      for (int j = 0; j < J; ++j) {
        for (int c = 0; c < K; ++c) round_stats.Njk[(size_t)j * K + c] = first.Njk[(size_t)j * (K + 1) + c];
        round_stats.Njk[(size_t)j * K + k] += first.Njk[(size_t)j * (K + 1) + K];
      }
    }
    if (anyempty(ms.clusters)) {  // :476
      if (journal) ctx.dcache_rollback();
      ctx.qz_swap_alt();
      continue;
    }
    if (opt.verbose) std::cout << '=' << std::flush;
    if (trace_phases)
      std::cerr << "[split k=" << k << " M=" << sel.M << "] select+gather+init " << msec(t0, t1) << " ms, refine "
                << msec(t1, t2) << " ms, clone+auglabels " << msec(t2, t3) << " ms, full-data iteration " << msec(t3, now())
                << " ms" << std::endl;
    if ((Fsplit < F) && (std::abs((F - Fsplit) / F) > lch::CONVERGE)) {  // :484-489
      tally[k] = 0;
      if (journal) ctx.dcache_journal_end();
      finish_stats(ctx, fin);
      model.next_stats = std::move(fin);  // (K = 0 when they could not be had from the moved rows)
      return true;  // the augmented qZ is now the current one
    }
    if (journal) ctx.dcache_rollback();
    ctx.qz_swap_alt();
  }
  return false;
}

// ---------------------------------------------------------------------------
// cluster.cpp:564-629
// ---------------------------------------------------------------------------
double cluster(lcc::Context& ctx, Model& model, const ClusterOptions& opt) {
  if (opt.nthreads < 1) throw std::invalid_argument("Must specify at least one thread for execution!");
  ctx.qz_fill(1, 1.0);  // :583-585
  std::vector<int> tally;
  bool issplit = true;
  double F = 0.0;
  // Gauss-Wishart, dense: E-steps through the distance cache, statistics from the moved rows (VbemOptions::inc);
  // LC_SPLIT_NO_DELTA=1 restores full passes everywhere but the two-column statistics of the split candidates
  static const bool no_delta = test_on("LC_SPLIT_NO_DELTA") || test_on("LC_SPLIT_NO_DCACHE");
  static const double delta_tol = [] {
    const char* e = lck::test_switch("LC_SPLIT_DELTA_TOL");
    return e ? std::atof(e) : 0x1p-50;
  }();
  IncState inc;
  inc.tol = delta_tol;
  const bool inc_allowed = !no_delta && model.ckind == lch::C_GAUSSWISH && !opt.sparse;
  int round_no = 0, failed_rounds = 0, skip_rounds = 0, backoff = 4;
  bool inc_tried = false;
  model.final_stats.K = 0;
  model.next_stats.K = 0;
  ctx.dcache_invalidate();
  struct CacheRelease {  // the cache is this call's: hand its memory back on every way out
    lcc::Context& c;
    ~CacheRelease() { c.dcache_release(); }
  } cache_release{ctx};
  while (issplit) {
    std::vector<double> tr;
    VbemOptions vo;
    // A round may switch the cache off for itself (its clusters overlap too much for it to pay).  After two such rounds
    // in a row the next ones do not even try -- 4 of them, then 8, ... -- and the memory goes back meanwhile.
    if (round_no > 0) {
      if (inc_tried && !inc.on && !inc.no_room) {
        if (++failed_rounds >= 2) {
          skip_rounds = backoff;
          backoff *= 2;
          failed_rounds = 0;
          ctx.dcache_release();
        }
      } else if (inc_tried) {
        failed_rounds = 0;
        backoff = 4;
      }
    }
    ++round_no;
    inc_tried = inc_allowed && !inc.no_room && skip_rounds == 0;
    if (skip_rounds > 0) --skip_rounds;
    inc.on = inc_tried;
    inc.bad = 0;
    if (inc_allowed) {
      vo.inc = &inc;
      vo.capture_final = &model.final_stats;
      if (model.next_stats.K == ctx.K()) vo.preset = &model.next_stats;  // the accepted candidate's
    }
    model.final_stats.K = 0;
    vo.clusterprior = opt.clusterprior;
    vo.maxit = -1;
    vo.sparse = opt.sparse;
    vo.verbose = opt.verbose;
    vo.nthreads = opt.nthreads;
    vo.trace = &tr;
    // the split ordering needs the data term LL_k of the converged responsibilities (cluster.cpp:407-410): it comes out
    // of the last iteration's E-step (a by-product of the normalisation sweep) instead of a full extra pass per round
    static const bool ll_extra_pass = test_on("LC_LL_EXTRA_PASS");  // (tests: the round-1 behaviour)
    vo.want_ll = !ll_extra_pass;
    F = vbem(ctx, model, vo);
    model.next_stats.K = 0;
    finish_stats(ctx, model.final_stats);  // (before pruning touches the responsibilities)
    if (opt.trace) opt.trace->emplace_back((int)model.clusters.size(), tr);
    int nkeep = 0;
    for (const auto& cl : model.clusters) nkeep += !(cl.N() < lch::ZEROCUTOFF);
    static const bool trace_phases = env_on("LC_TRACE_PHASES");
    const auto c0 = std::chrono::steady_clock::now();
    if (!vo.want_ll && !(nkeep >= opt.maxclusters && opt.maxclusters >= 0)) data_loglik(ctx, model);  // split_gr will need it
    prune_clusters(ctx, model, opt.verbose);
    const auto c1 = std::chrono::steady_clock::now();
    if (opt.verbose) std::cout << '<' << std::flush;
    issplit = split_gr(ctx, model, tally, F, opt, inc_allowed ? &inc : nullptr);
    if (trace_phases)
      std::cerr << "[cluster K=" << model.clusters.size() << "] data_loglik+prune "
                << std::chrono::duration<double, std::milli>(c1 - c0).count() << " ms, split search "
                << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c1).count() << " ms"
                << std::endl;
    if (opt.verbose) std::cout << '>' << std::endl;
  }
  if (opt.verbose) {
    std::cout << "Finished!" << std::endl;
    std::cout << "Number of clusters = " << model.clusters.size() << std::endl;
    std::cout << "Free energy = " << F << std::endl;
  }
  return F;
}

}  // namespace lce
