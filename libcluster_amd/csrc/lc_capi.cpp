// extern "C" boundary (include/libcluster_hip.h).  Nothing throws across it.
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <new>
#include <algorithm>
#include <string>
#include <sched.h>
#include <atomic>
#include <thread>

#include "../../include/libcluster_hip.h"
#include "lc_ctx.hpp"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cstdlib>

#include "lc_engine.hpp"
#include "lc_topic.hpp"
#include "lc_host.hpp"

struct lc_ctx {
  lcc::Context impl;
  lc_ctx(int device, hipStream_t s) : impl(device, s) {}
};

struct lc_model {
  std::unique_ptr<lc_ctx> owned_ctx;  // set by lc_learn
  lc_ctx* ctx = nullptr;              // context holding qZ (owned or borrowed)
  lce::Model model;
  std::vector<std::pair<int, std::vector<double>>> rounds;
  int D = 0;
  // One process driving several GPUs (LIBCLUSTER_GPUS, lc_learn): every shard's context holds a block of rows
  // (single-matrix learners) or whole groups (GMC family); `ctx` stays null and the accessors walk the shards.
  struct Shard {
    std::unique_ptr<lc_ctx> ctx;
    hipStream_t stream = nullptr;
    int device = 0;
    std::vector<int> groups;  // global group ids in local order (whole-group sharding)
    int64_t row0 = 0;         // first global row (row sharding)
    ~Shard() {
      ctx.reset();
      if (stream) {
        (void)hipSetDevice(device);
        (void)hipStreamDestroy(stream);
      }
    }
  };
  std::vector<std::unique_ptr<Shard>> shards;
  bool rows_sharded = false;
  std::vector<int64_t> Nj;  // global group sizes
};

struct lc_tmodel {
  std::unique_ptr<lc_ctx> ctx;  // one group per document; holds qZ
  lce::TopicData data;
  lce::TopicModel model;
  std::vector<double> W;        // I_tot x Dt (MCM)
  std::vector<lce::TopicRound> rounds;
  std::vector<int> doc0;        // first document of every group (J + 1)
  int D = 0;
};

namespace {
// LC_BACKTRACE=1: print the native call stack on SIGSEGV / SIGABRT (debugging aid; symbols need -rdynamic or addr2line)
void segv_handler(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "libcluster_hip: fatal signal, native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct BacktraceInstaller {
  BacktraceInstaller() {
    if (std::getenv("LC_BACKTRACE")) {
      signal(SIGSEGV, segv_handler);
      signal(SIGABRT, segv_handler);
    }
  }
} g_backtrace_installer;

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

template <typename F>
int guarded(F f) {
  try {
    f();
    return LC_OK;
  } catch (const lcc::HipFailure& e) {
    return fail(LC_EHIP, e.what());
  } catch (const std::invalid_argument& e) {
    return fail(LC_EINVAL, e.what());
  } catch (const std::domain_error& e) {
    return fail(LC_EDOMAIN, e.what());
  } catch (const std::bad_alloc&) {
    return fail(LC_ERUNTIME, "out of host memory");
  } catch (const std::exception& e) {
    return fail(LC_ERUNTIME, e.what());
  } catch (...) {
    return fail(LC_ERUNTIME, "unknown error");
  }
}

void need(const void* p, const char* what) {
  if (!p) throw std::invalid_argument(std::string(what) + " must not be NULL");
}

// ---- lc_learn: algorithm set-up shared by the one-GPU and the sharded path ------------------------------------
bool algo_single(int algo) {
  return algo == LC_ALGO_VDP || algo == LC_ALGO_BGMM || algo == LC_ALGO_DGMM || algo == LC_ALGO_BEMM;
}

void algo_configure(int algo, lce::Model& model, bool verbose, bool sparse) {
  const char* sp = sparse ? "(sparse) " : "";
  switch (algo) {
    case LC_ALGO_VDP:
      if (verbose) std::cout << "Learning VDP..." << std::endl;  // cluster.cpp:647-648
      model.wkind = lch::W_STICKBREAK;
      break;
    case LC_ALGO_BGMM:
      if (verbose) std::cout << "Learning Bayesian GMM..." << std::endl;  // :678-679
      model.wkind = lch::W_DIRICHLET;
      break;
    case LC_ALGO_DGMM:
      if (verbose) std::cout << "Learning Bayesian diagonal GMM..." << std::endl;  // :708-709
      model.wkind = lch::W_DIRICHLET;
      model.ckind = lch::C_NORMGAMMA;
      break;
    case LC_ALGO_BEMM:
      if (verbose) std::cout << "Learning Bayesian EMM..." << std::endl;  // :745-746
      model.wkind = lch::W_DIRICHLET;
      model.ckind = lch::C_EXPGAMMA;
      break;
    case LC_ALGO_GMC:
      if (verbose) std::cout << "Learning " << sp << "GMC..." << std::endl;  // :775-779
      model.wkind = lch::W_GDIRICHLET;
      break;
    case LC_ALGO_SGMC:
      if (verbose) std::cout << "Learning " << sp << "Symmetric GMC..." << std::endl;  // :799-803
      model.wkind = lch::W_DIRICHLET;
      break;
    case LC_ALGO_DGMC:
      if (verbose) std::cout << "Learning " << sp << "Diagonal GMC..." << std::endl;  // :825-829
      model.wkind = lch::W_GDIRICHLET;
      model.ckind = lch::C_NORMGAMMA;
      break;
    default:
      if (verbose) std::cout << "Learning " << sp << "Exponential GMC..." << std::endl;  // :866-868
      model.wkind = lch::W_GDIRICHLET;
      model.ckind = lch::C_EXPGAMMA;
      break;
  }
}

// LIBCLUSTER_GPUS = N | "all": learn*() shard their observations over N GPUs of this node, one host thread and one
// context per GPU, statistics summed with RCCL (SURVEY 5 "Config / flags": GPU selection cannot go into the frozen
// learnBGMM / learnVDP / learnGMC signatures).  LIBCLUSTER_GPUS_SAME_DEVICE=1 places every shard on `device` with the
// host-staged transport (exercises the sharded path on a one-GPU machine; RCCL refuses two ranks on one GPU).
int requested_gpus(bool* same_device) {
  const char* e = std::getenv("LIBCLUSTER_GPUS");
  *same_device = false;
  if (!e || !*e) return 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return 1;  // (the one-GPU path reports the missing device)
  const char* sd = std::getenv("LIBCLUSTER_GPUS_SAME_DEVICE");
  *same_device = sd && *sd && !(sd[0] == '0' && sd[1] == 0);
  int n = std::string(e) == "all" ? ndev : std::atoi(e);
  if (n < 1) throw std::invalid_argument("LIBCLUSTER_GPUS must be a positive number or \"all\"");
  if (!*same_device && n > ndev) n = ndev;
  return std::min(n, 64);
}

// whole groups to shards, largest first onto the least loaded (per-group counts N_jk stay local)
std::vector<std::vector<int>> assign_groups(int J, const int64_t* Nj, int W) {
  std::vector<int> order((size_t)J);
  for (int j = 0; j < J; ++j) order[(size_t)j] = j;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return Nj[a] > Nj[b]; });
  std::vector<int64_t> load((size_t)W, 0);
  std::vector<std::vector<int>> mine((size_t)W);
  for (int j : order) {
    int r = 0;
    for (int t = 1; t < W; ++t)
      if (load[(size_t)t] < load[(size_t)r]) r = t;
    load[(size_t)r] += Nj[j];
    mine[(size_t)r].push_back(j);
  }
  for (auto& v : mine) std::sort(v.begin(), v.end());
  return mine;
}

double learn_sharded(int algo, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t rs, int64_t cs,
                     double wprior, const double* wprior_j, double clusterprior, int maxclusters, int sparse, int verbose, unsigned nthreads,
                     int device, int W, bool same_device, lc_model* m) {
  const bool single = algo_single(algo);
  if (single) W = (int)std::max<int64_t>(1, std::min<int64_t>(W, Nj[0] / 64));  // at least a few row groups per shard
  else W = std::min(W, J);
  algo_configure(algo, m->model, verbose != 0, sparse != 0);
  m->rows_sharded = single;
  m->Nj.assign(Nj, Nj + J);
  std::vector<std::vector<int>> groups;
  if (!single) groups = assign_groups(J, Nj, W);
  std::vector<int> devices((size_t)W);
  for (int r = 0; r < W; ++r) devices[(size_t)r] = same_device ? device : r;
  const char* force = std::getenv("LIBCLUSTER_COMM");
  const bool host = same_device || (force && std::string(force) == "host");
  std::vector<std::shared_ptr<lcm::Comm>> comms = host ? lcm::host_init_local(W) : lcm::rccl_init_all(devices);
  if (verbose) std::cout << "Sharding over " << W << " GPU(s), " << comms[0]->kind() << " all-reduce" << std::endl;

  std::vector<lce::Model> models((size_t)W, m->model);
  std::vector<std::vector<std::pair<int, std::vector<double>>>> traces((size_t)W);
  std::vector<std::exception_ptr> errs((size_t)W);
  std::vector<double> Fs((size_t)W, 0.0);
  m->shards.clear();
  for (int r = 0; r < W; ++r) m->shards.emplace_back(new lc_model::Shard());
  const int64_t N0 = single ? Nj[0] : 0;
  std::atomic<int> first_fail{-1};  // the shard whose failure made the others fail (they only see aborted collectives)
  // every shard thread -- and the M-step pool it starts (thread_local per calling thread, lc_engine.cpp; its workers
  // inherit the mask) -- keeps to its slice of the CPUs this process may use.  ALL shards run on short-lived threads and
  // the caller only joins: its own mask, and the masks of the persistent pool workers it may already own, are never
  // narrowed (a shard on the caller's thread left that thread's pool on 1 / W of the cores for later calls).
  cpu_set_t all_cpus;
  CPU_ZERO(&all_cpus);
  const bool have_mask = sched_getaffinity(0, sizeof(all_cpus), &all_cpus) == 0 && CPU_COUNT(&all_cpus) >= 2 * W;
  auto pin = [&](int r) {
    if (!have_mask) return;
    const int per = CPU_COUNT(&all_cpus) / W;
    cpu_set_t mine;
    CPU_ZERO(&mine);
    int seen = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &all_cpus)) {
        if (seen >= r * per && seen < (r + 1) * per) CPU_SET(c, &mine);
        ++seen;
      }
    (void)sched_setaffinity(0, sizeof(mine), &mine);  // (0 = the calling thread)
  };
  auto work = [&](int r) {
    try {
      pin(r);
      lc_model::Shard& sh = *m->shards[(size_t)r];
      sh.device = devices[(size_t)r];
      if (hipSetDevice(sh.device) != hipSuccess) throw lcc::HipFailure("hipSetDevice failed for a shard");
      if (hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking) != hipSuccess)
        throw lcc::HipFailure("hipStreamCreate failed for a shard");
      sh.ctx.reset(new lc_ctx(sh.device, sh.stream));
      lcc::Context& ctx = sh.ctx->impl;
      ctx.set_comm(comms[(size_t)r]);
      lce::Model& model = models[(size_t)r];
      if (single) {
        const int64_t base = N0 / W, rem = N0 % W;
        const int64_t lo = r * base + std::min<int64_t>(r, rem), n = base + (r < rem ? 1 : 0);
        sh.row0 = lo;
        const double* xp = Xj[0] + lo * rs;
        ctx.set_data(1, &xp, &n, D, rs, cs);
        model.weights.emplace_back(model.wkind, wprior);  // vecweights(1, weights), cluster.cpp:653/684/715/752
      } else {
        sh.groups = groups[(size_t)r];
        std::vector<const double*> xp;
        std::vector<int64_t> nn;
        for (int j : sh.groups) {
          xp.push_back(Xj[j]);
          nn.push_back(Nj[j]);
        }
        ctx.set_data((int)xp.size(), xp.data(), nn.data(), D, rs, cs);
        ctx.set_group_sharded(true);
        if (wprior_j)  // the priors the caller's weight objects carry (weights.resize(J, W()) keeps them, cluster.cpp:192)
          for (int j : sh.groups) model.weights.emplace_back(model.wkind, wprior_j[j]);
      }
      lce::ClusterOptions co;
      co.clusterprior = clusterprior;
      co.maxclusters = maxclusters;
      co.sparse = !single && sparse;  // the single-matrix learners pass sparse=false, :657/:688/:719/:756
      co.verbose = verbose != 0 && r == 0;
      co.nthreads = std::max(1u, nthreads / (unsigned)W);
      co.trace = &traces[(size_t)r];
      Fs[(size_t)r] = lce::cluster(ctx, model, co);
      ctx.synchronize();
      lcc::cache_release_thread();  // this thread ends here: its cached blocks may serve other threads from now on
    } catch (...) {
      errs[(size_t)r] = std::current_exception();
      int none = -1;
      first_fail.compare_exchange_strong(none, r);  // (the shards that fail AFTER this one report the abort below)
      for (auto& c : comms) c->abort();  // the other shards fail in their next collective instead of waiting
      // the owner tag's contract is "let go after a sync": copies or kernels of this shard may still target its blocks
      // (page-locked blocks are shared across devices); errors of the sync itself are of no interest here
      (void)hipSetDevice(devices[(size_t)r]);
      (void)hipDeviceSynchronize();
      lcc::cache_release_thread();
    }
  };
  std::vector<std::thread> th;
  for (int r = 0; r < W; ++r) th.emplace_back(work, r);
  for (auto& t : th) t.join();
  if (const int f = first_fail.load(); f >= 0) {  // the root cause, not a secondary "communicator was aborted"
    m->shards.clear();
    std::rethrow_exception(errs[(size_t)f]);
  }
  // every shard ran the same M-steps on the same reduced statistics: clusters, F and the rounds are identical; the
  // group weights live with the shard that holds the group
  lce::Model out = std::move(models[0]);
  if (!single) {
    std::vector<lch::WeightState> w;
    w.reserve((size_t)J);
    for (int j = 0; j < J; ++j) w.emplace_back(out.wkind, wprior_j ? wprior_j[j] : lch::ALPHA1PRIOR);
    for (int r = 0; r < W; ++r) {
      lce::Model& mr = r == 0 ? out : models[(size_t)r];
      for (size_t l = 0; l < m->shards[(size_t)r]->groups.size(); ++l)
        w[(size_t)m->shards[(size_t)r]->groups[l]] = mr.weights[l];
    }
    out.weights = std::move(w);
  }
  m->model = std::move(out);
  m->rounds = std::move(traces[0]);
  return Fs[0];
}
}  // namespace

extern "C" {

const char* lc_last_error(void) { return g_err.c_str(); }
int lc_version(void) { return 300; }
const char* lc_statistics_kernel_name(int D, int K) {
  const int DP = D <= 128 ? lck::padded_dim(D) : 0;
  return DP > 0 ? lck::suffstat_kernel_name(DP, K, lck::estep_active_width(D, DP)) : "suffstat_kernel";
}
#ifndef LC_SOURCE_HASH
#define LC_SOURCE_HASH "unknown"
#endif
const char* lc_source_hash(void) { return LC_SOURCE_HASH; }

int lc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

double lc_const_converge(void) { return lch::CONVERGE; }
double lc_const_fengydel(void) { return lch::FENGYDEL; }
double lc_const_zerocutoff(void) { return lch::ZEROCUTOFF; }
int lc_const_splititer(void) { return (int)lch::SPLITITER; }

// ---------------------------------------------------------------------------
int lc_ctx_create(int device, void* stream, lc_ctx** out) {
  return guarded([&] {
    need(out, "out");
    *out = new lc_ctx(device, (hipStream_t)stream);
  });
}

int lc_ctx_destroy(lc_ctx* ctx) {
  return guarded([&] { delete ctx; });
}

int lc_ctx_set_stream(lc_ctx* ctx, void* stream) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.set_stream((hipStream_t)stream);
  });
}

int lc_ctx_synchronize(lc_ctx* ctx) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.synchronize();
  });
}

int lc_ctx_dims(lc_ctx* ctx, int* J, int* D, int64_t* Ntotal, int* K) {
  return guarded([&] {
    need(ctx, "ctx");
    if (J) *J = ctx->impl.J();
    if (D) *D = ctx->impl.D();
    if (Ntotal) *Ntotal = ctx->impl.Ntotal();
    if (K) *K = ctx->impl.K();
  });
}

int lc_ctx_set_data(lc_ctx* ctx, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t rs, int64_t cs) {
  return guarded([&] {
    need(ctx, "ctx");
    need(Xj, "Xj");
    need(Nj, "Nj");
    ctx->impl.set_data(J, Xj, Nj, D, rs, cs);
  });
}

int lc_ctx_synth(lc_ctx* ctx, int64_t N, int D, int K, const double* mu, const double* L, uint64_t seed,
                 int64_t row_offset, double hard) {
  return guarded([&] {
    need(ctx, "ctx");
    need(mu, "mu");
    need(L, "L");
    ctx->impl.synth(N, D, K, mu, L, seed, row_offset, hard);
  });
}

int lc_ctx_synth_groups(lc_ctx* ctx, int J, const int64_t* Nj, int D, int K, const double* mu, const double* L,
                        const double* cdf, uint64_t seed, const int64_t* group_ids, double hard) {
  return guarded([&] {
    need(ctx, "ctx");
    need(Nj, "Nj");
    need(mu, "mu");
    need(L, "L");
    ctx->impl.synth_groups(J, Nj, D, K, mu, L, cdf, seed, group_ids, 0, hard);
  });
}

int lc_ctx_set_sharding(lc_ctx* ctx, int whole_groups) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.set_group_sharded(whole_groups != 0);
  });
}

int lc_ctx_set_skip_zero(lc_ctx* ctx, int on) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.set_skip_zero(on != 0);
  });
}

int lc_ctx_get_rows(lc_ctx* ctx, int j, int64_t row0, int64_t n, double* out) {
  return guarded([&] {
    need(ctx, "ctx");
    need(out, "out");
    ctx->impl.get_rows(j, row0, n, out);
  });
}

int lc_ctx_set_qz(lc_ctx* ctx, int j, const double* q, int K, int64_t rs, int64_t cs) {
  return guarded([&] {
    need(ctx, "ctx");
    need(q, "q");
    ctx->impl.qz_set(j, q, K, rs, cs);
  });
}

int lc_ctx_get_qz(lc_ctx* ctx, int j, double* q, int64_t rs, int64_t cs) {
  return guarded([&] {
    need(ctx, "ctx");
    need(q, "q");
    ctx->impl.qz_get(j, q, rs, cs);
  });
}

int lc_ctx_get_qz_rows(lc_ctx* ctx, int j, int64_t row0, int64_t n, double* q, int64_t rs, int64_t cs) {
  return guarded([&] {
    need(ctx, "ctx");
    need(q, "q");
    ctx->impl.qz_get_rows(j, row0, n, q, rs, cs);
  });
}

int lc_ctx_fill_qz(lc_ctx* ctx, int K, double value) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.qz_fill(K, value);
    ctx->impl.synchronize();
  });
}

// ---------------------------------------------------------------------------
int lc_estep(lc_ctx* ctx, int K, const double* A, const double* m, const double* c, double* Fz, double* LLk) {
  return guarded([&] {
    need(ctx, "ctx");
    need(A, "A");
    need(m, "m");
    need(c, "c");
    ctx->impl.estep(K, A, m, c, Fz, LLk);
  });
}

int lc_estep_posterior(lc_ctx* ctx, int K, const double* nu, const double* beta, const double* m, const double* iW,
                       const double* logdW, const double* Elogpi, const unsigned char* active, double* Fz,
                       double* LLk) {
  return guarded([&] {
    need(ctx, "ctx");
    need(nu, "nu");
    need(beta, "beta");
    need(m, "m");
    need(iW, "iW");
    need(logdW, "logdW");
    need(Elogpi, "Elogpi");
    if (K < 1) throw std::invalid_argument("K must be >= 1");
    const int D = ctx->impl.D(), J = ctx->impl.J();
    std::vector<double> A((size_t)K * D * D), c((size_t)J * K), cst(K);
    for (int k = 0; k < K; ++k) {
      std::vector<double> L(iW + (size_t)k * D * D, iW + (size_t)(k + 1) * D * D);
      // mahaldist's PD check, probutils.cpp:131-132
      if (!lch::cholesky(L, D)) throw std::invalid_argument("Matrix A is not positive definite");
      std::vector<double> Li = lch::tril_inverse(L, D);
      const double s = std::sqrt(nu[k]);
      for (size_t t = 0; t < Li.size(); ++t) A[(size_t)k * D * D + t] = s * Li[t];
      double sumpsi = 0.0;
      for (int d = 1; d <= D; ++d) sumpsi += lch::digamma((nu[k] + 1 - d) / 2);
      cst[k] = 0.5 * (sumpsi + logdW[k] - D * (1.0 / beta[k] + std::log(lch::PI)));  // distributions.cpp:360-364
    }
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K; ++k) {
        double v = Elogpi[(size_t)j * K + k] + cst[k];
        if (active && !active[(size_t)j * K + k]) v = -std::numeric_limits<double>::infinity();
        c[(size_t)j * K + k] = v;
      }
    std::vector<double> ll(K);
    ctx->impl.estep(K, A.data(), m, c.data(), Fz, ll.data());
    if (LLk) {
      std::vector<double> Njk((size_t)J * K), nks((size_t)K, 0.0);
      ctx->impl.colsums(Njk.data());
      for (int k = 0; k < K; ++k)
        for (int j = 0; j < J; ++j) nks[(size_t)k] += Njk[(size_t)j * K + k];
      // whole groups per rank: the counts above are this rank's groups only, while ll is already summed over ranks
      if (ctx->impl.group_sharded()) ctx->impl.allreduce_values(nks.data(), K);
      for (int k = 0; k < K; ++k) LLk[k] = ll[k] + cst[k] * nks[(size_t)k];
    }
  });
}

int lc_eloglike(lc_ctx* ctx, int K, const double* nu, const double* beta, const double* m, const double* iW,
                const double* logdW) {
  return guarded([&] {
    need(ctx, "ctx");
    need(nu, "nu");
    need(beta, "beta");
    need(m, "m");
    need(iW, "iW");
    need(logdW, "logdW");
    if (K < 1) throw std::invalid_argument("K must be >= 1");
    const int D = ctx->impl.D(), J = ctx->impl.J();
    std::vector<double> A((size_t)K * D * D), c((size_t)J * K);
    for (int k = 0; k < K; ++k) {
      std::vector<double> L(iW + (size_t)k * D * D, iW + (size_t)(k + 1) * D * D);
      if (!lch::cholesky(L, D)) throw std::invalid_argument("Matrix A is not positive definite");
      std::vector<double> Li = lch::tril_inverse(L, D);
      const double s = std::sqrt(nu[k]);
      for (size_t t = 0; t < Li.size(); ++t) A[(size_t)k * D * D + t] = s * Li[t];
      double sumpsi = 0.0;
      for (int d = 1; d <= D; ++d) sumpsi += lch::digamma((nu[k] + 1 - d) / 2);
      const double cst = 0.5 * (sumpsi + logdW[k] - D * (1.0 / beta[k] + std::log(lch::PI)));
      for (int j = 0; j < J; ++j) c[(size_t)j * K + k] = cst;
    }
    ctx->impl.estep(K, A.data(), m, c.data(), nullptr, nullptr, /*raw=*/true);
  });
}

int lc_mahaldist(lc_ctx* ctx, const double* mu, const double* A, double* dist) {
  return guarded([&] {
    need(ctx, "ctx");
    need(mu, "mu");
    need(A, "A");
    need(dist, "dist");
    const int D = ctx->impl.D(), J = ctx->impl.J();
    std::vector<double> L(A, A + (size_t)D * D);
    // probutils.cpp:128-132 (LDLT with a positive diagonal <=> Cholesky succeeds)
    if (!lch::cholesky(L, D)) throw std::invalid_argument("Matrix A is not positive definite");
    const std::vector<double> Li = lch::tril_inverse(L, D);  // (x-mu) A^-1 (x-mu)^T = || L^-1 (x-mu) ||^2
    std::vector<double> c((size_t)J, 0.0);
    ctx->impl.estep(1, Li.data(), mu, c.data(), nullptr, nullptr, /*raw=*/true);  // column 0 <- -0.5 d^2
    int64_t o = 0;
    for (int j = 0; j < J; ++j) {
      const int64_t n = ctx->impl.N(j);
      if (n > 0) ctx->impl.qz_get_column(j, 0, dist + o);
      for (int64_t r = 0; r < n; ++r) dist[o + r] *= -2.0;
      o += n;
    }
  });
}

int lc_suffstat(lc_ctx* ctx, const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.suffstat(smask, Nk, xs, xxs, Njk);
  });
}

int lc_suffstat_diag(lc_ctx* ctx, const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.suffstat_diag(smask, Nk, xs, xxs, Njk);
  });
}

int lc_estep_diag(lc_ctx* ctx, int K, const double* a, const double* w2, const double* w1, const double* c, int raw,
                  double* Fz, double* LLk) {
  return guarded([&] {
    need(ctx, "ctx");
    need(a, "a");
    need(w2, "w2");
    need(w1, "w1");
    need(c, "c");
    ctx->impl.estep_diag(K, a, w2, w1, c, Fz, LLk, raw != 0);
  });
}

int lc_colsums(lc_ctx* ctx, double* Njk) {
  return guarded([&] {
    need(ctx, "ctx");
    need(Njk, "Njk");
    ctx->impl.colsums(Njk);
  });
}

int lc_ctx_set_allreduce(lc_ctx* ctx, lc_allreduce_fn fn, void* user) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.set_allreduce(fn, user);
  });
}

int lc_comm_unique_id(void* id) {
  return guarded([&] {
    need(id, "id");
    lcm::rccl_unique_id(id);
  });
}

int lc_comm_rccl_available(void) { return lcm::rccl_available() ? 1 : 0; }

int lc_ctx_comm_init_rccl(lc_ctx* ctx, const void* id, int rank, int world) {
  return guarded([&] {
    need(ctx, "ctx");
    need(id, "id");
    ctx->impl.set_comm(lcm::rccl_init_rank(id, rank, world, ctx->impl.device()));
  });
}

int lc_ctx_comm_init_host(lc_ctx* ctx, const char* name, int rank, int world) {
  return guarded([&] {
    need(ctx, "ctx");
    need(name, "name");
    ctx->impl.set_comm(lcm::host_init_shm(name, rank, world));
  });
}

int lc_ctx_comm_free(lc_ctx* ctx) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.synchronize();
    ctx->impl.set_comm(nullptr);
  });
}

int lc_ctx_comm_info(lc_ctx* ctx, int* rank, int* world, const char** kind) {
  return guarded([&] {
    need(ctx, "ctx");
    const auto& c = ctx->impl.comm();
    if (rank) *rank = c ? c->rank() : 0;
    if (world) *world = c ? c->world() : 1;
    if (kind) *kind = c ? c->kind() : (ctx->impl.distributed() ? "hook" : "none");
  });
}

int lc_ctx_allreduce(lc_ctx* ctx, double* values, int n) {
  return guarded([&] {
    need(ctx, "ctx");
    need(values, "values");
    ctx->impl.allreduce_values(values, n);
  });
}

int lc_trim_cache(void) {
  return guarded([&] { lcc::trim_cache(); });
}

int lc_ctx_timing_enable(lc_ctx* ctx, int on) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.timing_enable(on != 0);
  });
}

int lc_ctx_timing_reset(lc_ctx* ctx) {
  return guarded([&] {
    need(ctx, "ctx");
    ctx->impl.timing_reset();
  });
}

int lc_ctx_timing_get(lc_ctx* ctx, double* estep_ms, int64_t* estep_calls, double* suffstat_ms,
                      int64_t* suffstat_calls) {
  return guarded([&] {
    need(ctx, "ctx");
    const lcc::KernelTimes t = ctx->impl.timing_get();
    if (estep_ms) *estep_ms = t.estep_ms;
    if (estep_calls) *estep_calls = t.estep_calls;
    if (suffstat_ms) *suffstat_ms = t.suffstat_ms;
    if (suffstat_calls) *suffstat_calls = t.suffstat_calls;
  });
}

int lc_ctx_timing_get_fused(lc_ctx* ctx, double* fused_ms, int64_t* fused_calls) {
  return guarded([&] {
    need(ctx, "ctx");
    const lcc::KernelTimes t = ctx->impl.timing_get();
    if (fused_ms) *fused_ms = t.fused_ms;
    if (fused_calls) *fused_calls = t.fused_calls;
  });
}

int lc_ctx_timing_get_all(lc_ctx* ctx, double* out, int n) {
  return guarded([&] {
    need(ctx, "ctx");
    need(out, "out");
    const lcc::KernelTimes t = ctx->impl.timing_get();
    const double v[LC_TIMING_FIELDS] = {t.estep_ms, (double)t.estep_calls, t.suffstat_ms, (double)t.suffstat_calls,
                                        t.fused_ms, (double)t.fused_calls, t.allreduce_ms, (double)t.allreduce_calls,
                                        t.host_stats_ms, t.host_mstep_ms, t.host_estep_ms, t.host_fenergy_ms,
                                        (double)t.host_iters, (double)t.estep_diag_mfma_calls};
    for (int i = 0; i < n && i < LC_TIMING_FIELDS; ++i) out[i] = v[i];
  });
}

// ---------------------------------------------------------------------------
int lc_vbem(lc_ctx* ctx, lc_model** model, int wkind, int ckind, double wprior, double clusterprior, int maxit,
            int sparse, int fixed_iters, int verbose, unsigned nthreads, double* F, int* niter, double* Ftrace,
            int ntrace) {
  return guarded([&] {
    need(ctx, "ctx");
    need(model, "model");
    if (wkind < 0 || wkind > 2) throw std::invalid_argument("unknown weight kind");
    if (ckind < 0 || ckind > 2) throw std::invalid_argument("unknown cluster kind");
    if (nthreads < 1) throw std::invalid_argument("Must specify at least one thread for execution!");
    std::unique_ptr<lc_model> fresh;
    lc_model* m = *model;
    if (!m) {
      fresh.reset(new lc_model());
      m = fresh.get();
      m->model.wkind = wkind;
      m->model.ckind = ckind;
      m->D = ctx->impl.D();
      for (int j = 0; j < ctx->impl.J(); ++j) m->model.weights.emplace_back(wkind, wprior);
    } else if (m->model.wkind != wkind || m->model.ckind != ckind) {
      throw std::invalid_argument("model was created with other distribution kinds");
    }
    m->ctx = ctx;
    std::vector<double> tr;
    lce::VbemOptions vo;
    vo.clusterprior = clusterprior;
    vo.maxit = maxit;
    vo.sparse = sparse != 0;
    vo.verbose = verbose != 0;
    vo.fixed_iters = fixed_iters;
    vo.nthreads = nthreads;
    vo.trace = &tr;
    const double f = lce::vbem(ctx->impl, m->model, vo);
    m->rounds.emplace_back((int)m->model.clusters.size(), tr);
    if (F) *F = f;
    if (niter) *niter = (int)tr.size();
    if (Ftrace)
      for (int i = 0; i < ntrace && i < (int)tr.size(); ++i) Ftrace[i] = tr[(size_t)i];
    if (fresh) *model = fresh.release();
  });
}

int lc_prune(lc_ctx* ctx, lc_model* model, int verbose, int* removed) {
  return guarded([&] {
    need(ctx, "ctx");
    need(model, "model");
    const int K0 = (int)model->model.clusters.size();
    if (K0 != ctx->impl.K()) throw std::invalid_argument("model and qZ have different numbers of clusters");
    lce::prune_clusters(ctx->impl, model->model, verbose != 0);
    if (removed) *removed = K0 - (int)model->model.clusters.size();
  });
}

int lc_learn(int algo, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t rs, int64_t cs,
             double wprior, double clusterprior, int maxclusters, int sparse, int verbose, unsigned nthreads,
             int device, lc_model** out, double* F) {
  return lc_learn_w(algo, J, Xj, Nj, D, rs, cs, wprior, nullptr, clusterprior, maxclusters, sparse, verbose, nthreads,
                    device, out, F);
}

int lc_learn_w(int algo, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t rs, int64_t cs,
               double wprior, const double* wprior_j, double clusterprior, int maxclusters, int sparse, int verbose,
               unsigned nthreads, int device, lc_model** out, double* F) {
  return guarded([&] {
    need(Xj, "Xj");
    need(Nj, "Nj");
    need(out, "out");
    if (algo < 0 || algo > LC_ALGO_EGMC) throw std::invalid_argument("unknown algorithm");
    const bool single = algo_single(algo);
    if (single && J != 1) throw std::invalid_argument("this algorithm takes a single observation matrix");
    if (nthreads < 1) throw std::invalid_argument("Must specify at least one thread for execution!");  // cluster.cpp:576-577
    if (algo == LC_ALGO_BEMM || algo == LC_ALGO_EGMC) {  // cluster.cpp:742-743, 862-864 (before anything is printed)
      for (int j = 0; j < J; ++j) {
        need(Xj[j], "Xj[j]");
        for (int64_t i = 0; i < Nj[j]; ++i)
          for (int d = 0; d < D; ++d)
            if (Xj[j][i * rs + d * cs] < 0) throw std::invalid_argument("X has to be in the range [0, inf)!");
      }
    }
    std::unique_ptr<lc_model> m(new lc_model());
    m->D = D;
    bool same_device = false;
    const int ngpu = requested_gpus(&same_device);
    // LIBCLUSTER_FORCE_SHARDED=1 (tests): take the sharded path with a single shard too (ncclCommInitAll on one device)
    static const bool force_sharded = std::getenv("LIBCLUSTER_FORCE_SHARDED") != nullptr;
    if ((ngpu > 1 || (force_sharded && std::getenv("LIBCLUSTER_GPUS"))) && (single ? Nj[0] >= 128 : J >= 2)) {
      for (int j = 0; j < J; ++j) need(Xj[j], "Xj[j]");
      const double f = learn_sharded(algo, J, Xj, Nj, D, rs, cs, wprior, single ? nullptr : wprior_j, clusterprior,
                                     maxclusters, sparse, verbose, nthreads, device, ngpu, same_device, m.get());
      if (F) *F = f;
      *out = m.release();
      return;
    }
    m->owned_ctx.reset(new lc_ctx(device, nullptr));
    m->ctx = m->owned_ctx.get();
    lcc::Context& ctx = m->ctx->impl;
    ctx.set_data(J, Xj, Nj, D, rs, cs);
    algo_configure(algo, m->model, verbose != 0, sparse != 0);
    if (single) m->model.weights.emplace_back(m->model.wkind, wprior);  // vecweights(1, weights), :653/:684/:715/:752
    else if (wprior_j)  // the caller's weight objects keep their priors (weights.resize(J, W()), cluster.cpp:192)
      for (int j = 0; j < J; ++j) m->model.weights.emplace_back(m->model.wkind, wprior_j[j]);
    lce::ClusterOptions co;
    co.clusterprior = clusterprior;
    co.maxclusters = maxclusters;
    co.sparse = !single && sparse;  // the single-matrix learners pass sparse=false, :657/:688/:719/:756
    co.verbose = verbose != 0;
    co.nthreads = nthreads;
    co.trace = &m->rounds;
    const double f = lce::cluster(ctx, m->model, co);
    if (F) *F = f;
    *out = m.release();
  });
}

int lc_cluster(lc_ctx* ctx, int wkind, int ckind, double wprior, double clusterprior, int maxclusters, int sparse,
               int verbose, unsigned nthreads, lc_model** out, double* F) {
  return guarded([&] {
    need(ctx, "ctx");
    need(out, "out");
    if (wkind < 0 || wkind > 2) throw std::invalid_argument("unknown weight kind");
    if (ckind < 0 || ckind > 2) throw std::invalid_argument("unknown cluster kind");
    if (nthreads < 1) throw std::invalid_argument("Must specify at least one thread for execution!");
    std::unique_ptr<lc_model> m(new lc_model());
    m->ctx = ctx;
    m->D = ctx->impl.D();
    m->model.wkind = wkind;
    m->model.ckind = ckind;
    if (ctx->impl.J() == 1 && wkind != lch::W_GDIRICHLET) m->model.weights.emplace_back(wkind, wprior);
    lce::ClusterOptions co;
    co.clusterprior = clusterprior;
    co.maxclusters = maxclusters;
    co.sparse = sparse != 0;
    co.verbose = verbose != 0;
    co.nthreads = nthreads;
    co.trace = &m->rounds;
    const double f = lce::cluster(ctx->impl, m->model, co);
    if (F) *F = f;
    *out = m.release();
  });
}

int lc_model_free(lc_model* m) {
  return guarded([&] { delete m; });
}

int lc_model_dims(lc_model* m, int* J, int* K, int* D) {
  return guarded([&] {
    need(m, "model");
    if (J) *J = (int)m->model.weights.size();
    if (K) *K = (int)m->model.clusters.size();
    if (D) *D = m->D;
  });
}

int lc_model_rounds(lc_model* m, int* nrounds) {
  return guarded([&] {
    need(m, "model");
    need(nrounds, "nrounds");
    *nrounds = (int)m->rounds.size();
  });
}

int lc_model_round(lc_model* m, int r, int* K, int* niter, double* F, int nF) {
  return guarded([&] {
    need(m, "model");
    if (r < 0 || r >= (int)m->rounds.size()) throw std::invalid_argument("round index out of range");
    const auto& rd = m->rounds[(size_t)r];
    if (K) *K = rd.first;
    if (niter) *niter = (int)rd.second.size();
    if (F)
      for (int i = 0; i < nF && i < (int)rd.second.size(); ++i) F[i] = rd.second[(size_t)i];
  });
}

int lc_model_get_qz(lc_model* m, int j, double* q, int64_t rs, int64_t cs) {
  return guarded([&] {
    need(m, "model");
    need(q, "q");
    if (!m->shards.empty()) {
      if (j < 0 || j >= (int)m->Nj.size()) throw std::invalid_argument("group index out of range");
      for (auto& sh : m->shards) {
        if (m->rows_sharded) {
          sh->ctx->impl.qz_get(0, q + sh->row0 * rs, rs, cs);
        } else {
          for (size_t l = 0; l < sh->groups.size(); ++l)
            if (sh->groups[l] == j) sh->ctx->impl.qz_get((int)l, q, rs, cs);
        }
      }
      return;
    }
    if (!m->ctx) throw std::invalid_argument("model has no context");
    m->ctx->impl.qz_get(j, q, rs, cs);
  });
}

int lc_model_get_qz_all(lc_model* m, double* q) {
  return guarded([&] {
    need(m, "model");
    need(q, "q");
    if (!m->shards.empty()) {
      const size_t K = m->model.clusters.size();
      if (m->rows_sharded) {
        for (auto& sh : m->shards) sh->ctx->impl.qz_get_all(q + (size_t)sh->row0 * K);
      } else {
        std::vector<int64_t> off(m->Nj.size() + 1, 0);
        for (size_t j = 0; j < m->Nj.size(); ++j) off[j + 1] = off[j] + m->Nj[j];
        std::vector<double> tmp;
        for (auto& sh : m->shards) {
          int64_t n = 0;
          for (int j : sh->groups) n += m->Nj[(size_t)j];
          tmp.resize((size_t)n * K);
          sh->ctx->impl.qz_get_all(tmp.data());
          int64_t at = 0;
          for (int j : sh->groups) {
            std::memcpy(q + (size_t)off[(size_t)j] * K, tmp.data() + (size_t)at * K, (size_t)m->Nj[(size_t)j] * K * sizeof(double));
            at += m->Nj[(size_t)j];
          }
        }
      }
      return;
    }
    if (!m->ctx) throw std::invalid_argument("model has no context");
    m->ctx->impl.qz_get_all(q);
  });
}

int lc_model_get_qz_all_colmajor(lc_model* m, double* const* q) {
  return guarded([&] {
    need(m, "model");
    need(q, "q");
    if (!m->shards.empty()) {
      for (auto& sh : m->shards) {
        if (m->rows_sharded) {  // a block of rows of the one N x K matrix: same columns, taller stride
          double* base = q[0] + sh->row0;
          const int64_t ld = m->Nj[0];
          sh->ctx->impl.qz_get_all_colmajor(&base, &ld);
        } else {
          std::vector<double*> ptrs;
          for (int j : sh->groups) ptrs.push_back(q[j]);
          sh->ctx->impl.qz_get_all_colmajor(ptrs.data());
        }
      }
      return;
    }
    if (!m->ctx) throw std::invalid_argument("model has no context");
    m->ctx->impl.qz_get_all_colmajor(q);
  });
}

int lc_ctx_get_qz_all_colmajor(lc_ctx* ctx, double* const* q) {
  return guarded([&] {
    need(ctx, "ctx");
    need(q, "q");
    ctx->impl.qz_get_all_colmajor(q);
  });
}

int lc_ctx_get_qz_all(lc_ctx* ctx, double* q) {
  return guarded([&] {
    need(ctx, "ctx");
    need(q, "q");
    ctx->impl.qz_get_all(q);
  });
}

int lc_model_weights(lc_model* m, int j, double* Elogweight, double* Nk) {
  return guarded([&] {
    need(m, "model");
    if (j < 0 || j >= (int)m->model.weights.size()) throw std::invalid_argument("group index out of range");
    const lch::WeightState& w = m->model.weights[(size_t)j];
    if (Elogweight) std::copy(w.Elogpi.begin(), w.Elogpi.end(), Elogweight);
    if (Nk) std::copy(w.Nk.begin(), w.Nk.end(), Nk);
  });
}

int lc_model_kinds(lc_model* m, int* wkind, int* ckind) {
  return guarded([&] {
    need(m, "model");
    if (wkind) *wkind = m->model.wkind;
    if (ckind) *ckind = m->model.ckind;
  });
}

int lc_model_cluster(lc_model* m, int k, double* N, double* mean, double* cov, double* nu, double* beta, double* iW,
                     double* logdW) {
  return guarded([&] {
    need(m, "model");
    if (k < 0 || k >= (int)m->model.clusters.size()) throw std::invalid_argument("cluster index out of range");
    const lch::ClusterAny& ca = m->model.clusters[(size_t)k];
    if (N) *N = ca.N();
    if (ca.kind == lch::C_GAUSSWISH) {
      const lch::GaussWishState& c = ca.gw;
      if (mean) std::copy(c.m.begin(), c.m.end(), mean);
      if (cov) {
        const std::vector<double> cv = c.getcov();
        std::copy(cv.begin(), cv.end(), cov);
      }
      if (nu) *nu = c.nu;
      if (beta) *beta = c.beta;
      if (iW) std::copy(c.iW.begin(), c.iW.end(), iW);
      if (logdW) *logdW = c.logdW;
    } else if (ca.kind == lch::C_NORMGAMMA) {
      // cov [D] = getcov() = L*nu (distributions.h:375, as the reference defines it); iW [D] = L; logdW = logL
      const lch::NormGammaState& c = ca.ng;
      if (mean) std::copy(c.m.begin(), c.m.end(), mean);
      if (cov)
        for (int d = 0; d < c.D; ++d) cov[d] = c.L[(size_t)d] * c.nu;
      if (nu) *nu = c.nu;
      if (beta) *beta = c.beta;
      if (iW) std::copy(c.L.begin(), c.L.end(), iW);
      if (logdW) *logdW = c.logL;
    } else {
      // mean [D] = getrate() = a*ib (distributions.h:433; the family has no getmean); nu = a; iW [D] = ib; logdW = logb
      const lch::ExpGammaState& c = ca.eg;
      if (mean)
        for (int d = 0; d < c.D; ++d) mean[d] = c.a * c.ib[(size_t)d];
      if (cov) throw std::invalid_argument("Exponential clusters have no covariance");
      if (nu) *nu = c.a;
      if (beta) *beta = 0.0;
      if (iW) std::copy(c.ib.begin(), c.ib.end(), iW);
      if (logdW) *logdW = c.logb;
    }
  });
}

int lc_model_fenergy(lc_model* m, double* Fw, double* Fc) {
  return guarded([&] {
    need(m, "model");
    if (Fw)
      for (size_t j = 0; j < m->model.weights.size(); ++j) Fw[j] = m->model.weights[j].fenergy();
    if (Fc)
      for (size_t k = 0; k < m->model.clusters.size(); ++k) Fc[k] = m->model.clusters[k].fenergy();
  });
}

// ---------------------------------------------------------------------------
double lc_digamma(double x) {  // poles (0, -1, -2, ...) give NaN here; inside the learners they raise LC_EDOMAIN
  try {
    return lch::digamma(x);
  } catch (...) {
    return std::numeric_limits<double>::quiet_NaN();
  }
}

int lc_weights_update(int wkind, double wprior, const double* Nk, int K, double* Elogweight, double* fenergy) {
  return guarded([&] {
    need(Nk, "Nk");
    if (wkind < 0 || wkind > 2) throw std::invalid_argument("unknown weight kind");
    if (K < 1) throw std::invalid_argument("K must be >= 1");
    lch::WeightState w(wkind, wprior);
    w.update(Nk, K);
    if (Elogweight) std::copy(w.Elogpi.begin(), w.Elogpi.end(), Elogweight);
    if (fenergy) *fenergy = w.fenergy();
  });
}

int lc_gw_mstep(double clustwidth, int D, double Ns, const double* xs, const double* xxs, double* nu, double* beta,
                double* m, double* iW, double* logdW, double* fenergy, double* A, double* eloglike_const) {
  return guarded([&] {
    need(xs, "xs");
    need(xxs, "xxs");
    lch::GaussWishState g(clustwidth, D);
    g.addstats(Ns, xs, xxs);
    g.update();
    if (nu) *nu = g.nu;
    if (beta) *beta = g.beta;
    if (m) std::copy(g.m.begin(), g.m.end(), m);
    if (iW) std::copy(g.iW.begin(), g.iW.end(), iW);
    if (logdW) *logdW = g.logdW;
    if (fenergy) *fenergy = g.fenergy();
    if (A) {
      const std::vector<double> a = g.whitener();
      std::copy(a.begin(), a.end(), A);
    }
    if (eloglike_const) *eloglike_const = g.eloglike_const();
  });
}

int lc_ng_mstep(double clustwidth, int D, double Ns, const double* xs, const double* xxs, double* nu, double* beta,
                double* m, double* L, double* logL, double* fenergy, double* eloglike_const) {
  return guarded([&] {
    need(xs, "xs");
    need(xxs, "xxs");
    lch::NormGammaState g(clustwidth, D);
    g.addstats(Ns, xs, xxs);
    g.update();
    if (nu) *nu = g.nu;
    if (beta) *beta = g.beta;
    if (m) std::copy(g.m.begin(), g.m.end(), m);
    if (L) std::copy(g.L.begin(), g.L.end(), L);
    if (logL) *logL = g.logL;
    if (fenergy) *fenergy = g.fenergy();
    if (eloglike_const) *eloglike_const = g.eloglike_const();
  });
}

int lc_eg_mstep(double obsmag, int D, double Ns, const double* xs, double* a, double* ib, double* logb,
                double* fenergy, double* eloglike_const) {
  return guarded([&] {
    need(xs, "xs");
    lch::ExpGammaState g(obsmag, D);
    g.addstats(Ns, xs, nullptr);
    g.update();
    if (a) *a = g.a;
    if (ib) std::copy(g.ib.begin(), g.ib.end(), ib);
    if (logb) *logb = g.logb;
    if (fenergy) *fenergy = g.fenergy();
    if (eloglike_const) *eloglike_const = g.eloglike_const();
  });
}

// ---------------------------------------------------------------------------
// learnSCM / learnMCM
// ---------------------------------------------------------------------------
int lc_learn_topic(int J, const int* Ij, const double* const* Xji, const int64_t* Nji, int D, int64_t rs, int64_t cs,
                   const double* const* Wj, int Dt, const double* const* qY0, double prior_t, double prior_k,
                   unsigned maxT, int maxK, int verbose, unsigned nthreads, int device, lc_tmodel** out, double* F) {
  return lc_learn_topic_dist(J, Ij, Xji, Nji, D, rs, cs, Wj, Dt, qY0, prior_t, prior_k, maxT, maxK, verbose, nthreads,
                             device, nullptr, nullptr, nullptr, out, F);
}

int lc_learn_topic_dist(int J, const int* Ij, const double* const* Xji, const int64_t* Nji, int D, int64_t rs,
                        int64_t cs, const double* const* Wj, int Dt, const double* const* qY0, double prior_t,
                        double prior_k, unsigned maxT, int maxK, int verbose, unsigned nthreads, int device,
                        void* stream, lc_allreduce_fn fn, void* user, lc_tmodel** out, double* F) {
  return guarded([&] {
    need(out, "out");
    const bool mcm = Wj != nullptr;
    std::unique_ptr<lc_tmodel> m(new lc_tmodel());
    lce::TopicData& d = m->data;
    // A rank that rejects its input must not leave the others waiting in the first collective: the communicator is set
    // up first, every rank validates and prepares locally, and the ranks agree on the outcome before anybody returns.
    m->D = D;
    if (fn) {  // whole groups (with all their documents) per rank: cluster statistics, N_tk, Fyz, Fz are summed
      m->ctx.reset(new lc_ctx(device, static_cast<hipStream_t>(stream)));
      m->ctx->impl.set_allreduce(fn, user);
      m->ctx->impl.set_group_sharded(true);
    }
    std::exception_ptr bad;
    try {
    need(Ij, "Ij");
    need(Xji, "Xji");
    need(Nji, "Nji");
    if (nthreads < 1) throw std::invalid_argument("Must specify at least one thread for execution!");
    if (J < 1) throw std::invalid_argument("need at least one group of observations");
    if (maxT < 1) throw std::invalid_argument("maxT must be at least 1");
    d.J = J;
    d.Ij.assign(Ij, Ij + J);
    m->doc0.assign(1, 0);
    for (int j = 0; j < J; ++j) {
      if (Ij[j] < 0) throw std::invalid_argument("negative document count");
      for (int i = 0; i < Ij[j]; ++i) d.doc_group.push_back(j);
      m->doc0.push_back(m->doc0.back() + Ij[j]);
    }
    d.Itot = (int)d.doc_group.size();
    if (d.Itot < 1) throw std::invalid_argument("need at least one document");
    if (mcm) {  // mcluster.cpp:548-556 checks W.size() / W[j].rows() against X: here they are implied by Ij
      if (Dt < 1) throw std::invalid_argument("W needs at least one column");
      m->W.resize((size_t)d.Itot * Dt);
      for (int j = 0; j < J; ++j) {
        if (Ij[j] > 0) need(Wj[j], "Wj[j]");
        if (Ij[j] > 0) std::copy(Wj[j], Wj[j] + (size_t)Ij[j] * Dt, m->W.begin() + (size_t)m->doc0[j] * Dt);
      }
      d.W = m->W.data();
      d.Dt = Dt;
    }
    // qY: |U(-1,1)| rows, normalised (scluster.cpp:519-521 / mcluster.cpp:559-561).  The reference draws from Eigen's
    // Random(), i.e. std::rand() per coefficient in column-major order; the same sequence is consumed here.
    m->model.T = (int)maxT;
    m->model.qY.assign((size_t)d.Itot * maxT, 0.0);
    for (int j = 0; j < J; ++j) {
      const int I = Ij[j];
      double* q = m->model.qY.data() + (size_t)m->doc0[j] * maxT;
      if (qY0) {
        if (I > 0) need(qY0[j], "qY0[j]");
        std::copy(qY0[j], qY0[j] + (size_t)I * maxT, q);
      } else {
        for (unsigned t = 0; t < maxT; ++t)
          for (int i = 0; i < I; ++i)
            q[(size_t)i * maxT + t] = std::abs(-1.0 + 2.0 * (double)std::rand() / (double)RAND_MAX);
        for (int i = 0; i < I; ++i) {
          double nrm = 0.0;
          for (unsigned t = 0; t < maxT; ++t) nrm += q[(size_t)i * maxT + t];
          for (unsigned t = 0; t < maxT; ++t)
            q[(size_t)i * maxT + t] = std::exp(std::log(q[(size_t)i * maxT + t]) - std::log(nrm));
        }
      }
    }
    } catch (...) {
      bad = std::current_exception();
    }
    if (!fn) {  // one rank: argument errors come before the device is touched, as in the reference
      if (bad) std::rethrow_exception(bad);
      m->ctx.reset(new lc_ctx(device, static_cast<hipStream_t>(stream)));
    } else if (m->ctx->impl.allreduce_value(bad ? 1.0 : 0.0) > 0.0) {
      if (bad) std::rethrow_exception(bad);
      throw std::invalid_argument("another rank rejected its input");
    }
    if (verbose) std::cout << (mcm ? "Learning MCM..." : "Learning SCM...") << std::endl;  // scluster.cpp:595-596
    const double docs = m->ctx->impl.allreduce_value((double)d.Itot);
    if (!mcm && (double)maxT > docs)  // scluster.cpp:531-533 (sic: no space before X)
      throw std::invalid_argument("maxT must be less than the number of documents ofX!");
    m->ctx->impl.set_data(d.Itot, Xji, Nji, D, rs, cs);
    lce::TopicOptions o;
    o.prior_t = prior_t;
    o.prior_k = prior_k;
    o.maxK = maxK;
    o.verbose = verbose != 0;
    o.nthreads = nthreads;
    const double f = lce::topic_cluster(m->ctx->impl, d, m->model, o, &m->rounds);
    if (F) *F = f;
    *out = m.release();
  });
}

int lc_tmodel_free(lc_tmodel* m) {
  return guarded([&] { delete m; });
}

int lc_tmodel_dims(lc_tmodel* m, int* J, int* Itot, int* T, int* K, int* D, int* Dt) {
  return guarded([&] {
    need(m, "model");
    if (J) *J = m->data.J;
    if (Itot) *Itot = m->data.Itot;
    if (T) *T = m->model.T;
    if (K) *K = (int)m->model.clusters.size();
    if (D) *D = m->D;
    if (Dt) *Dt = m->data.Dt;
  });
}

int lc_tmodel_get_qy(lc_tmodel* m, int j, double* qY) {
  return guarded([&] {
    need(m, "model");
    need(qY, "qY");
    if (j < 0 || j >= m->data.J) throw std::invalid_argument("group index out of range");
    const int T = m->model.T;
    std::copy(m->model.qY.begin() + (size_t)m->doc0[(size_t)j] * T, m->model.qY.begin() + (size_t)m->doc0[(size_t)j + 1] * T,
              qY);
  });
}

int lc_tmodel_get_qz(lc_tmodel* m, int doc, double* q, int64_t rs, int64_t cs) {
  return guarded([&] {
    need(m, "model");
    need(q, "q");
    m->ctx->impl.qz_get(doc, q, rs, cs);
  });
}

int lc_tmodel_get_qz_all(lc_tmodel* m, double* q) {
  return guarded([&] {
    need(m, "model");
    need(q, "q");
    m->ctx->impl.qz_get_all(q);
  });
}

int lc_tmodel_get_qz_all_colmajor(lc_tmodel* m, double* const* q) {
  return guarded([&] {
    need(m, "model");
    need(q, "q");
    m->ctx->impl.qz_get_all_colmajor(q);
  });
}

int lc_tmodel_weights(lc_tmodel* m, int level, int idx, double* Elogweight, double* Nk) {
  return guarded([&] {
    need(m, "model");
    const std::vector<lch::WeightState>& v = level == 0 ? m->model.weights_j : m->model.weights_t;
    if (level < 0 || level > 1) throw std::invalid_argument("level must be 0 (groups) or 1 (top-level clusters)");
    if (idx < 0 || idx >= (int)v.size()) throw std::invalid_argument("weight index out of range");
    const lch::WeightState& w = v[(size_t)idx];
    if (Elogweight) std::copy(w.Elogpi.begin(), w.Elogpi.end(), Elogweight);
    if (Nk) std::copy(w.Nk.begin(), w.Nk.end(), Nk);
  });
}

int lc_tmodel_cluster(lc_tmodel* m, int level, int idx, double* N, double* mean, double* cov, double* nu, double* beta,
                      double* iW, double* logdW, double* fenergy) {
  return guarded([&] {
    need(m, "model");
    if (level < 0 || level > 1) throw std::invalid_argument("level must be 0 (bottom) or 1 (top-level clusters)");
    const std::vector<lch::GaussWishState>& v = level == 0 ? m->model.clusters : m->model.clusters_t;
    if (idx < 0 || idx >= (int)v.size()) throw std::invalid_argument("cluster index out of range");
    const lch::GaussWishState& c = v[(size_t)idx];
    if (N) *N = c.N;
    if (mean) std::copy(c.m.begin(), c.m.end(), mean);
    if (cov) {
      const std::vector<double> cv = c.getcov();
      std::copy(cv.begin(), cv.end(), cov);
    }
    if (nu) *nu = c.nu;
    if (beta) *beta = c.beta;
    if (iW) std::copy(c.iW.begin(), c.iW.end(), iW);
    if (logdW) *logdW = c.logdW;
    if (fenergy) *fenergy = c.fenergy();
  });
}

int lc_tmodel_rounds(lc_tmodel* m, int* nrounds) {
  return guarded([&] {
    need(m, "model");
    need(nrounds, "nrounds");
    *nrounds = (int)m->rounds.size();
  });
}

int lc_tmodel_round(lc_tmodel* m, int r, int* T, int* K, int* niter, double* F, int nF) {
  return guarded([&] {
    need(m, "model");
    if (r < 0 || r >= (int)m->rounds.size()) throw std::invalid_argument("round index out of range");
    const lce::TopicRound& rd = m->rounds[(size_t)r];
    if (T) *T = rd.T;
    if (K) *K = rd.K;
    if (niter) *niter = (int)rd.F.size();
    if (F)
      for (int i = 0; i < nF && i < (int)rd.F.size(); ++i) F[i] = rd.F[(size_t)i];
  });
}

}  // extern "C"
