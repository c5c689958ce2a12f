// Two ranks (two processes) through the C ABI alone -- no Python, no torch: each rank holds half of the rows of one
// synthetic data set in its own context, the per-iteration statistics are summed by the library's own collective
// (lc_ctx_comm_init_host: host-staged, works with both ranks on one GPU; lc_ctx_comm_init_rccl: RCCL, which needs one
// GPU per rank and is reported as SKIP when the box has a single GPU), and the whole model selection
// (lc_cluster: VBEM + prune + split search) must reproduce the single-rank rounds, K and F.
//
// The loop being distributed is the reference's single-process one, src/cluster.cpp:207-223.
//
//   dist_test host|rccl [N D Ktrue]      exit 0 = pass (or SKIP, printed), 1 = failure
//
// The parent never touches the GPU: it forks the ranks first (fork after HIP initialisation is not supported).
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "libcluster_hip.h"

namespace {

struct Result {
  int ok = 0, K = 0, rounds = 0;
  double F = 0.0;
  char note[200] = {0};
};

void mixture(int D, int K, std::vector<double>& mu, std::vector<double>& L) {
  mu.assign((size_t)K * D, 0.0);
  L.assign((size_t)K * D * D, 0.0);
  unsigned s = 12345u;
  auto rnd = [&] {
    s = s * 1664525u + 1013904223u;
    return (double)(s >> 8) / (double)(1u << 24);
  };
  for (int k = 0; k < K; ++k) {
    for (int d = 0; d < D; ++d) mu[(size_t)k * D + d] = 8.0 * (rnd() - 0.5) + 6.0 * (k - 0.5 * K) * ((d + k) % 3 == 0);
    for (int i = 0; i < D; ++i)
      for (int j = 0; j <= i; ++j) L[((size_t)k * D + i) * D + j] = i == j ? 0.7 + 0.6 * rnd() : 0.3 * (rnd() - 0.5);
  }
}

#define CHECK_LC(expr)                                                              \
  do {                                                                              \
    if ((expr) != LC_OK) {                                                          \
      snprintf(res.note, sizeof(res.note), "%s: %s", #expr, lc_last_error());       \
      return res;                                                                   \
    }                                                                               \
  } while (0)

// transport: 0 = none (single rank), 1 = host, 2 = rccl
Result run_rank(int rank, int world, int transport, const std::string& tag, int64_t N, int D, int Kt) {
  Result res;
  if (N < 0) {  // probe: how many GPUs does this box have?
    res.K = lc_device_count();
    res.ok = 1;
    return res;
  }
  std::vector<double> mu, L;
  mixture(D, Kt, mu, L);
  lc_ctx* ctx = nullptr;
  CHECK_LC(lc_ctx_create(transport == 2 ? rank : 0, nullptr, &ctx));  // RCCL: one GPU per rank
  const int64_t base = N / world, rem = N % world;
  const int64_t lo = rank * base + (rank < rem ? rank : rem), n = base + (rank < rem ? 1 : 0);
  CHECK_LC(lc_ctx_synth(ctx, n, D, Kt, mu.data(), L.data(), 777, lo, 0.9));  // rows [lo, lo+n) of ONE Philox stream
  if (transport == 1) {
    CHECK_LC(lc_ctx_comm_init_host(ctx, tag.c_str(), rank, world));
  } else if (transport == 2) {
    // rank 0 draws the id and ships it through a file (any channel will do)
    const std::string path = "/tmp/lc_dist_test_" + tag + ".id";
    unsigned char id[LC_COMM_ID_BYTES];
    if (rank == 0) {
      CHECK_LC(lc_comm_unique_id(id));
      const std::string tmp = path + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) {
        snprintf(res.note, sizeof(res.note), "cannot write %s", tmp.c_str());
        return res;
      }
      fclose(f);
      rename(tmp.c_str(), path.c_str());
    } else {
      FILE* f = nullptr;
      for (int i = 0; i < 30000 && !(f = fopen(path.c_str(), "rb")); ++i) usleep(1000);
      if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) {
        snprintf(res.note, sizeof(res.note), "no unique id from rank 0");
        return res;
      }
      fclose(f);
    }
    if (lc_ctx_comm_init_rccl(ctx, id, rank, world) != LC_OK) {
      snprintf(res.note, sizeof(res.note), "SKIP %s", lc_last_error());
      res.ok = 2;
      if (rank == 0) unlink(path.c_str());
      return res;
    }
    if (rank == 0) unlink(path.c_str());
  }
  if (transport) {  // the collective itself: sum of (rank + 1) * i over the ranks
    double v[3] = {1.0 * (rank + 1), 2.0 * (rank + 1), -0.5 * (rank + 1)};
    CHECK_LC(lc_ctx_allreduce(ctx, v, 3));
    const double tot = 0.5 * world * (world + 1);
    if (v[0] != tot || v[1] != 2.0 * tot || v[2] != -0.5 * tot) {
      snprintf(res.note, sizeof(res.note), "all-reduce returned %g %g %g, expected %g %g %g", v[0], v[1], v[2], tot,
               2.0 * tot, -0.5 * tot);
      return res;
    }
  }
  lc_model* model = nullptr;
  double F = 0.0;
  CHECK_LC(lc_cluster(ctx, LC_W_DIRICHLET, LC_C_GAUSSWISH, 1.0, 1.0, -1, 0, 0, 4, &model, &F));
  int J = 0, K = 0, Dm = 0, nr = 0;
  CHECK_LC(lc_model_dims(model, &J, &K, &Dm));
  CHECK_LC(lc_model_rounds(model, &nr));
  res.F = F;
  res.K = K;
  res.rounds = nr;
  res.ok = 1;
  lc_model_free(model);
  if (transport) lc_ctx_comm_free(ctx);
  lc_ctx_destroy(ctx);
  return res;
}

// run `world` ranks as child processes; returns rank 0's result
Result run_world(int world, int transport, int64_t N, int D, int Kt) {
  const std::string tag = std::to_string((long)getpid()) + "_" + std::to_string(transport) + "_" + std::to_string(world);
  std::vector<int> fds((size_t)world);
  std::vector<pid_t> pids((size_t)world);
  for (int r = 0; r < world; ++r) {
    int p[2];
    if (pipe(p) != 0) exit(1);
    const pid_t pid = fork();
    if (pid == 0) {
      close(p[0]);
      Result res = run_rank(r, world, transport, tag, N, D, Kt);
      (void)!write(p[1], &res, sizeof(res));
      close(p[1]);
      _exit(res.ok ? 0 : 1);
    }
    close(p[1]);
    fds[(size_t)r] = p[0];
    pids[(size_t)r] = pid;
  }
  Result first;
  for (int r = 0; r < world; ++r) {
    Result res;
    const ssize_t got = read(fds[(size_t)r], &res, sizeof(res));
    close(fds[(size_t)r]);
    int st = 0;
    waitpid(pids[(size_t)r], &st, 0);
    if (got != (ssize_t)sizeof(res)) {
      res = Result();
      snprintf(res.note, sizeof(res.note), "rank %d died (status %d)", r, st);
    }
    if (r == 0 || (first.ok == 1 && res.ok != 1)) first = res;
  }
  return first;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "host";
  const int64_t N = argc > 2 ? atoll(argv[2]) : 40000;
  const int D = argc > 3 ? atoi(argv[3]) : 8;
  const int Kt = argc > 4 ? atoi(argv[4]) : 4;
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC (RCCL between processes on this pool)
  setenv("LC_COMM_TIMEOUT_S", "120", 0);
  const int transport = mode == "rccl" ? 2 : 1;
  const int ndev = run_world(1, 0, -1, D, Kt).K;
  printf("%d GPU(s) visible\n", ndev);
  const Result one = run_world(1, 0, N, D, Kt);
  if (one.ok != 1) {
    fprintf(stderr, "single rank failed: %s\n", one.note);
    return 1;
  }
  printf("1 rank : K = %d, rounds = %d, F = %.12g\n", one.K, one.rounds, one.F);
  if (transport == 2 && ndev < 2) {
    // RCCL refuses two ranks on one GPU: exercise the native RCCL path with a world of one (communicator set-up, the
    // all-reduce on the context's stream inside every EM iteration) and say what was left out
    const Result solo = run_world(1, 2, N, D, Kt);
    if (solo.ok != 1) {
      fprintf(stderr, "RCCL world of one failed: %s\n", solo.note);
      return 1;
    }
    if (solo.K != one.K || solo.rounds != one.rounds || solo.F != one.F) {
      fprintf(stderr, "RCCL world of one differs: F %.15g vs %.15g\n", solo.F, one.F);
      return 1;
    }
    printf("RCCL, world of one: identical rounds, K and F\nSKIP two RCCL ranks need two GPUs (this box has %d)\n", ndev);
    printf("dist_test rccl OK (world 1)\n");
    return 0;
  }
  const Result two = run_world(2, transport, N, D, Kt);
  if (two.ok == 2) {
    printf("%s (two RCCL ranks need two GPUs)\ndist_test %s SKIPPED\n", two.note, mode.c_str());
    return 0;
  }
  if (two.ok != 1) {
    fprintf(stderr, "two ranks (%s) failed: %s\n", mode.c_str(), two.note);
    return 1;
  }
  printf("2 ranks: K = %d, rounds = %d, F = %.12g  (%s all-reduce)\n", two.K, two.rounds, two.F, mode.c_str());
  if (two.K != one.K || two.rounds != one.rounds || one.K < Kt - 1) {
    fprintf(stderr, "model selection differs: K %d vs %d, rounds %d vs %d\n", one.K, two.K, one.rounds, two.rounds);
    return 1;
  }
  if (!(std::fabs(two.F - one.F) <= 1e-10 * std::fabs(one.F))) {
    fprintf(stderr, "F differs: %.15g vs %.15g\n", one.F, two.F);
    return 1;
  }
  printf("dist_test %s OK\n", mode.c_str());
  return 0;
}
