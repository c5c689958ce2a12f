// SANITIZER HARNESS (tools/sanitize_host.sh): drives the host-side concurrency of the library -- the M-step worker pool
// (lc_engine.cpp), the per-thread block cache and its ownership tags (lc_ctx.cpp), the heap- and shared-memory
// transports of the host-staged all-reduce incl. aborts and a left-over rendezvous object (lc_comm.cpp), and the
// failure path of the sharded learners (lc_capi.cpp learn_sharded) -- against the host-memory HIP stand-in of
// hip_host_stub.cpp.  No kernel runs here and no result of the data path is produced or checked; what is checked is
// that sums of known vectors come out right, that failures end in exceptions / status codes instead of hangs, and
// that ASan / UBSan / TSan stay silent.  The reference's whole discipline for the same loops is one `omp critical`
// and one `omp atomic` (src/cluster.cpp:77, 412).
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "libcluster_hip.h"
#include "lc_comm.hpp"
#include "lc_ctx.hpp"
#include "lc_engine.hpp"

#define CHECK(c)                                                                \
  do {                                                                          \
    if (!(c)) {                                                                 \
      std::fprintf(stderr, "host_hammer: CHECK failed line %d: %s\n", __LINE__, #c); \
      std::_Exit(3);                                                            \
    }                                                                           \
  } while (0)

namespace {

// ---- the M-step pool ---------------------------------------------------------------------------------------------
void hammer_pool() {
  const int callers = 6;
  std::atomic<long> total{0};
  std::vector<std::thread> th;
  for (int t = 0; t < callers; ++t)
    th.emplace_back([&, t] {
      std::mt19937 rng(100 + t);
      for (int it = 0; it < 150; ++it) {
        const int n = 1 + (int)(rng() % 97);
        const unsigned nt = 1 + rng() % 8;
        std::vector<long> slot((size_t)n, 0);
        const bool thrower = it % 17 == 5;
        bool caught = false;
        try {
          lce::parallel_chunks(n, nt, 1e7, [&](int c) {
            long s = 0;
            for (int i = 0; i <= c; ++i) s += i;
            slot[(size_t)c] = s;
            if (thrower && c == n / 2) throw std::runtime_error("item failed");
            if (c == 0 && it % 29 == 3)  // a nested call finds the pool busy and runs inline
              lce::parallel_chunks(5, 4, 1e7, [&](int d) { slot[0] += d - d; });
          });
        } catch (const std::runtime_error&) {
          caught = true;
        }
        CHECK(caught == thrower);
        long want = 0, got = 0;
        for (int c = 0; c < n; ++c) want += (long)c * (c + 1) / 2, got += slot[(size_t)c];
        // every item ran exactly once (after a failing item the pool still finishes the others, the inline path -- one
        // thread, or the pool busy -- stops at it: only the exception is promised then)
        if (!thrower) CHECK(want == got);
        total += got;
      }
    });  // (a caller thread ends here: its thread_local pool joins its workers)
  for (auto& x : th) x.join();
  std::printf("pool: %d caller threads x 150 jobs ok (checksum %ld)\n", callers, total.load());
}

// ---- block cache -------------------------------------------------------------------------------------------------
void hammer_cache() {
  const int nthr = 8;
  std::atomic<int> stop{0};
  std::thread trimmer([&] {
    while (!stop.load()) {
      std::this_thread::sleep_for(std::chrono::milliseconds(3));
      lcc::trim_cache();
    }
  });
  std::vector<std::thread> th;
  for (int t = 0; t < nthr; ++t)
    th.emplace_back([t] {
      CHECK(hipSetDevice(t % 2) == hipSuccess);
      std::mt19937 rng(7 + t);
      for (int it = 0; it < 400; ++it) {
        lcc::DevBuf<double> a, b;
        lcc::DevBuf<int> c;
        const size_t na = 64 + rng() % 5000, nb = 64 + rng() % 200000;
        a.reserve(na);
        {
          lcc::RelaxedFit relaxed;
          b.reserve(nb);
        }
        c.reserve(33);
        // a block handed to two owners at once would show as a torn pattern (and as a race under TSan)
        for (size_t i = 0; i < na; ++i) a.p[i] = (double)(t * 1000003 + it);
        for (size_t i = 0; i < nb; i += 97) b.p[i] = (double)t;
        lcc::PinnedBuf p;
        p.resize(128 + rng() % 4096);
        p[0] = (double)t;
        std::this_thread::yield();
        for (size_t i = 0; i < na; ++i) CHECK(a.p[i] == (double)(t * 1000003 + it));
        for (size_t i = 0; i < nb; i += 97) CHECK(b.p[i] == (double)t);
        CHECK(p[0] == (double)t);
        if (it % 50 == 49) lcc::cache_release_thread();  // "after a sync": everything above is synchronous here
      }
    });  // thread exit: ThreadTag's destructor lets go of the thread's blocks
  for (auto& x : th) x.join();
  stop = 1;
  trimmer.join();
  lcc::trim_cache();
  std::printf("cache: %d threads x 400 take / release rounds on 2 devices, concurrent trims ok\n", nthr);
}

// ---- host-staged all-reduce: threads ---------------------------------------------------------------------------------
void run_rank(lcm::Comm& comm, int rounds, unsigned seed) {
  const int r = comm.rank(), W = comm.world();
  std::mt19937 rng(seed);  // the same sequence of sizes on every rank
  std::vector<double> buf;
  for (int it = 0; it < rounds; ++it) {
    // (one message longer than a 16 MB slot: the piece loop)
    const size_t n = it == rounds / 2 ? ((size_t)2 << 20) + 12345 : 1 + rng() % 60000;
    buf.assign(n, 0.0);
    for (size_t i = 0; i < n; i += 7) buf[i] = (double)(r + 1) * (double)(i % 13 + it);
    comm.allreduce_sum(buf.data(), (int64_t)n, nullptr);
    const double tot = 0.5 * W * (W + 1);
    for (size_t i = 0; i < n; i += 7) CHECK(buf[i] == tot * (double)(i % 13 + it));
    if (n > 1) CHECK(buf[1] == 0.0);
  }
}

void hammer_local() {
  for (int W : {1, 2, 5, 8}) {
    auto comms = lcm::host_init_local(W);
    std::vector<std::thread> th;
    for (int r = 0; r < W; ++r) th.emplace_back([&, r] { run_rank(*comms[(size_t)r], 40, 99); });
    for (auto& x : th) x.join();
  }
  // one rank fails instead of joining a collective: the others must throw, not wait
  {
    const int W = 6;
    auto comms = lcm::host_init_local(W);
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    for (int r = 0; r < W; ++r)
      th.emplace_back([&, r] {
        try {
          run_rank(*comms[(size_t)r], 5, 3);
          if (r == 3) {
            for (auto& c : comms) c->abort();
            return;
          }
          std::vector<double> v(100, 1.0);
          comms[(size_t)r]->allreduce_sum(v.data(), 100, nullptr);
          CHECK(false && "a collective completed although one rank never joined");
        } catch (const std::runtime_error&) {
          ++failed;
        }
      });
    for (auto& x : th) x.join();
    CHECK(failed.load() == W - 1);
  }
  std::printf("local: worlds of 1, 2, 5, 8 threads x 40 all-reduces, abort of one rank ok\n");
}

// ---- host-staged all-reduce: processes -------------------------------------------------------------------------------
// returns the number of children that exited with status 0
int fork_ranks(int W, const std::function<int(int)>& body) {
  std::vector<pid_t> pids;
  for (int r = 0; r < W; ++r) {
    const pid_t p = fork();
    CHECK(p >= 0);
    if (p == 0) std::_Exit(body(r));
    pids.push_back(p);
  }
  int ok = 0;
  for (pid_t p : pids) {
    int st = 0;
    CHECK(waitpid(p, &st, 0) == p);
    if (WIFEXITED(st) && WEXITSTATUS(st) == 0) ++ok;
    else if (!(WIFEXITED(st) && WEXITSTATUS(st) == 7))
      std::fprintf(stderr, "host_hammer: a rank ended with status 0x%x\n", st);  // (7 = the expected failure exit)
  }
  return ok;
}

void hammer_shm() {
  const std::string base = "hammer_" + std::to_string((long)getpid());
  for (int W : {2, 5, 8}) {
    const std::string name = base + "_w" + std::to_string(W);
    const int ok = fork_ranks(W, [&](int r) {
      try {
        auto c = lcm::host_init_shm(name, r, W);
        CHECK(std::string(c->kind()) == "host-shm");
        run_rank(*c, 25, 11);
        return 0;
      } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: %s\n", r, e.what());
        return 1;
      }
    });
    CHECK(ok == W);
  }
  // one rank aborts after a few rounds: every other rank fails in its next collective (exit 7), nobody hangs
  {
    const int W = 5;
    const std::string name = base + "_abort";
    const auto t0 = std::chrono::steady_clock::now();
    const int ok = fork_ranks(W, [&](int r) {
      try {
        auto c = lcm::host_init_shm(name, r, W);
        run_rank(*c, 4, 5);
        if (r == 2) {
          c->abort();
          return 0;
        }
        std::vector<double> v(10, 1.0);
        c->allreduce_sum(v.data(), 10, nullptr);
        return 1;  // must not complete
      } catch (const std::runtime_error&) {
        return 7;
      }
    });
    CHECK(ok == 1);
    CHECK(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 60.0);
  }
  // a run that died during its rendezvous left an object behind: initialised, attached == world - 1 (one more attach
  // completes the DEAD run's count).  The new run's ranks > 0 start first and find it; rank 0 comes 300 ms later,
  // poisons and replaces it; everybody must end up on the new object and the sums must be right.
  {
    const int W = 4;
    const std::string name = base + "_stale";
    const std::string path = "/lc_comm_" + name;
    const int fd = shm_open(path.c_str(), O_CREAT | O_RDWR, 0600);
    CHECK(fd >= 0);
    const size_t bytes = 64 + (size_t)W * ((size_t)2 << 20) * sizeof(double);
    CHECK(ftruncate(fd, (off_t)bytes) == 0);
    void* p = mmap(nullptr, 64, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    CHECK(p != MAP_FAILED);
    close(fd);
    // Ctrl: magic, attached, count, gen, aborted, ready, world, pad, slot_doubles (lc_comm.cpp)
    uint32_t* w = static_cast<uint32_t*>(p);
    w[1] = (uint32_t)(W - 1);
    w[2] = w[3] = w[4] = w[5] = 0;
    w[6] = (uint32_t)W;
    *reinterpret_cast<uint64_t*>(w + 8) = (uint64_t)2 << 20;
    __atomic_store_n(&w[0], 0x4c43434du, __ATOMIC_RELEASE);
    munmap(p, 64);
    const int ok = fork_ranks(W, [&](int r) {
      try {
        if (r == 0) std::this_thread::sleep_for(std::chrono::milliseconds(300));
        auto c = lcm::host_init_shm(name, r, W);
        run_rank(*c, 10, 21);
        return 0;
      } catch (const std::exception& e) {
        std::fprintf(stderr, "stale-object run, rank %d: %s\n", r, e.what());
        return 1;
      }
    });
    CHECK(ok == W);
    shm_unlink(path.c_str());
  }
  std::printf("shm: worlds of 2, 5, 8 processes x 25 all-reduces, abort of one rank, left-over rendezvous object ok\n");
}

// ---- C ABI: host math, uploads through the packer, and the failure path of the sharded learners ---------------------------
void hammer_capi() {
  CHECK(lc_version() > 0);
  // M-steps and weight updates from several threads at once (the library keeps no shared state there)
  std::vector<std::thread> th;
  for (int t = 0; t < 4; ++t)
    th.emplace_back([t] {
      std::mt19937 rng(50 + t);
      std::normal_distribution<double> nd;
      for (int it = 0; it < 40; ++it) {
        const int D = 1 + (int)(rng() % 24), n = D + 5;
        std::vector<double> xs((size_t)D, 0.0), xxs((size_t)D * D, 0.0), row((size_t)D);
        for (int i = 0; i < n; ++i) {
          for (int d = 0; d < D; ++d) row[(size_t)d] = nd(rng) + d;
          for (int d = 0; d < D; ++d) {
            xs[(size_t)d] += row[(size_t)d];
            for (int e = 0; e < D; ++e) xxs[(size_t)d * D + e] += row[(size_t)d] * row[(size_t)e];
          }
        }
        double nu, beta, logdW, fe, cst;
        std::vector<double> m((size_t)D), iW((size_t)D * D), A((size_t)D * D);
        CHECK(lc_gw_mstep(1.0, D, (double)n, xs.data(), xxs.data(), &nu, &beta, m.data(), iW.data(), &logdW, &fe, A.data(),
                          &cst) == LC_OK);
        CHECK(std::isfinite(fe) && std::isfinite(logdW) && nu == D + n);
        std::vector<double> Nk((size_t)(2 + it % 9)), el(Nk.size());
        for (auto& v : Nk) v = 1.0 + (double)(rng() % 1000);
        double fw;
        for (int wk = 0; wk < 3; ++wk) CHECK(lc_weights_update(wk, 1.0, Nk.data(), (int)Nk.size(), el.data(), &fw) == LC_OK);
        CHECK(std::isfinite(lc_digamma(0.5 + it)));
      }
    });
  for (auto& x : th) x.join();

  // a context on the stand-in device: the upload packer (worker pool + two page-locked buffers) and its way back
  lc_ctx* ctx = nullptr;
  CHECK(lc_ctx_create(0, nullptr, &ctx) == LC_OK);
  const int J = 3, D = 5;
  const int64_t Nj[3] = {1000, 37, 2500};
  std::vector<std::vector<double>> X(J);
  const double* ptr[3];
  for (int j = 0; j < J; ++j) {
    X[(size_t)j].resize((size_t)Nj[j] * D);
    for (size_t i = 0; i < X[(size_t)j].size(); ++i) X[(size_t)j][i] = (double)(j * 100000 + (long)i);
    ptr[j] = X[(size_t)j].data();
  }
  CHECK(lc_ctx_set_data(ctx, J, ptr, Nj, D, D, 1) == LC_OK);
  std::vector<double> back((size_t)40 * D);
  CHECK(lc_ctx_get_rows(ctx, 2, 2460, 40, back.data()) == LC_OK);
  for (int i = 0; i < 40 * D; ++i) CHECK(back[(size_t)i] == X[2][(size_t)2460 * D + i]);
  // the first kernel launch fails on the stand-in: a status and a message, nothing torn down twice
  CHECK(lc_ctx_fill_qz(ctx, 3, 1.0) != LC_OK);
  CHECK(std::strlen(lc_last_error()) > 0);
  lc_ctx_destroy(ctx);

  // sharded learners: every shard thread fails in its first launch -- the abort / release / rethrow path of
  // learn_sharded with 1, 3 and 8 shard threads, row blocks and whole groups
  for (const char* gpus : {"1", "3", "8"}) {
    setenv("LIBCLUSTER_GPUS", gpus, 1);
    setenv("LIBCLUSTER_GPUS_SAME_DEVICE", "1", 1);
    setenv("LIBCLUSTER_FORCE_SHARDED", "1", 1);
    for (int algo : {LC_ALGO_BGMM, LC_ALGO_GMC}) {
      lc_model* model = nullptr;
      double F = 0;
      const int rc = algo == LC_ALGO_BGMM ? lc_learn(algo, 1, ptr + 2, Nj + 2, D, D, 1, 1.0, 1.0, -1, 0, 0, 4, 0, &model, &F)
                                           : lc_learn(algo, J, ptr, Nj, D, D, 1, 1.0, 1.0, -1, 0, 0, 4, 0, &model, &F);
      CHECK(rc == LC_EHIP && model == nullptr);
      CHECK(std::strstr(lc_last_error(), "HIP") != nullptr);
    }
  }
  unsetenv("LIBCLUSTER_GPUS");
  unsetenv("LIBCLUSTER_GPUS_SAME_DEVICE");
  unsetenv("LIBCLUSTER_FORCE_SHARDED");
  lc_trim_cache();
  std::printf("capi: concurrent M-steps, upload packer round trip, launch failure -> status, sharded failure path (1, 3, 8 shards) ok\n");
}

}  // namespace

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  setenv("LC_STUB_DEVICES", "2", 1);
  setenv("LC_COMM_TIMEOUT_S", "60", 0);
  const std::string what = argc > 1 ? argv[1] : "all";
  if (what == "shm" || what == "all") hammer_shm();  // first: forks while this process is still single-threaded
  if (what == "pool" || what == "all") hammer_pool();
  if (what == "cache" || what == "all") hammer_cache();
  if (what == "local" || what == "all") hammer_local();
  if (what == "capi" || what == "all") hammer_capi();
  std::printf("host_hammer %s OK\n", what.c_str());
  return 0;
}
