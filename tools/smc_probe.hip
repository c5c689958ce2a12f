// Probe: softmax_cached_kernel (the E-step from cached distances) alone, on a synthetic slab shaped like the
// model-selection run's (every row close to one cluster, far from the others): time per launch and bytes per second, for
// the main VBEM's form (LL_k wanted, moves reported, nothing changes between launches) and the plain form.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilibcluster_amd/csrc -Iinclude [-D...] -o tools/variants/smc_probe tools/smc_probe.hip
// Usage: smc_probe [N K]
#include "../libcluster_amd/csrc/lc_kernels_aux.hip"

#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void fill_slab(double* slab, int64_t NP, int K) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= NP) return;
  const unsigned h = (unsigned)(n * 2654435761u);
  const int own = (h >> 8) % K;
  for (int j = 0; j < K; ++j)
    slab[(int64_t)j * NP + n] = (j == own ? -32.0 : own == 5 && j == 6 ? -33.0 : -400.0 - 3.0 * j) - (h & 255) * (1.0 / 64);
}

int main(int argc, char** argv) {
  const int64_t NP = argc > 1 ? atoll(argv[1]) : 10000000;
  const int K = argc > 2 ? atoi(argv[2]) : 32;
  double *slab, *q, *dq, *amax, *ctab, *fz, *ll;
  CK(hipMalloc(&slab, NP * K * 8));
  CK(hipMalloc(&q, NP * K * 8));
  CK(hipMalloc(&dq, NP * K * 8));
  CK(hipMalloc(&amax, NP * 8));
  CK(hipMalloc(&ctab, K * 8));
  const int64_t grid = lck::softmax_cached_grid(NP);
  CK(hipMalloc(&fz, grid * 8));
  CK(hipMalloc(&ll, grid * K * 8));
  std::vector<double> c(K);
  for (int j = 0; j < K; ++j) c[j] = -3.0 - 0.01 * j;
  CK(hipMemcpy(ctab, c.data(), K * 8, hipMemcpyHostToDevice));
  CK(hipMemset(q, 0, NP * K * 8));
  fill_slab<<<(unsigned)((NP + 255) / 256), 256>>>(slab, NP, K);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int64_t* qh;
  CK(hipMalloc(&qh, NP * 8));
  for (int mode = 0; mode < 6; ++mode) {  // 0: plain (writes q), 1: moves reported, 2: moves + LL_k, 3: a split candidate's E-step, 4: 1 with row fingerprints, 5: 3 with them
    lck::CachedNormLaunch a{};
    a.dcache = slab; a.ldc = NP; a.fresh = nullptr; a.ldf = 0; a.colmap = nullptr; a.ctab = ctab; a.K = K;
    a.rginfo = nullptr; a.nrows = NP; a.NP = NP; a.qZ = q; a.ldq = NP; a.fz_part = fz;
    if (mode >= 1) { a.dq = dq; a.ldd = K; a.amax = amax; a.dq_tol = 8.9e-16; }
    if (mode >= 4) { a.qhash = qh; a.qhash_in = 1; }
    if (mode == 2) a.ll_part = ll;
    // mode 3: c_6 alternates between launches -- every row's (tiny) q_6 changes in its low bits, the rows of cluster 5
    // (1 / K of all; cluster 6 is their close second) move by O(1)
    double* ctab2;
    CK(hipMalloc(&ctab2, K * 8));
    c[6 % K] += 3.0;
    CK(hipMemcpy(ctab2, c.data(), K * 8, hipMemcpyHostToDevice));
    c[6 % K] -= 3.0;
    auto go = [&](int r) {
      a.ctab = (mode == 3 || mode == 5) && (r & 1) ? ctab2 : ctab;
      CK(lck::launch_softmax_cached(a, 0));
    };
    for (int w = 0; w < 2; ++w) go(w);
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) go(r);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)NP * K * 8 * 2;  // slab in + q out (plain) / slab in + old q in (nothing moved)
    printf("N=%lld K=%d mode=%d (%s): %.3f ms, %.2f TB/s\n", (long long)NP, K, mode,
           mode == 0 ? "plain: slab in, q out" : mode == 1 ? "moves reported, nothing moved: slab in, old q in" : mode == 2 ? "the same + LL_k" : "split candidate: + 2 columns out, 1/K rows moved", ms,
           bytes / ms / 1e9);
  }
  return 0;
}
