/* libcluster_hip.h -- C ABI of the MI355X (gfx950) implementation of
 * libcluster's variational E-step / sufficient-statistic hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference (dsteinberg/libcluster) interface it
 * replaces.  All functions return an lc_status; on failure lc_last_error()
 * (thread-local) holds the message.  The C++ facade (include/libcluster.h,
 * include/distributions.h) re-throws the reference's exception classes:
 *   LC_EINVAL  -> std::invalid_argument   LC_ERUNTIME -> std::runtime_error
 *   LC_EDOMAIN -> std::domain_error       LC_EHIP     -> std::runtime_error
 *
 * There is NO CPU fallback: without a HIP device every data-path call fails
 * with LC_EHIP.  Host-only entry points (lc_mstep_*, lc_weights_*, lc_digamma)
 * work anywhere.
 *
 * Matrices are described as (pointer, row_stride, col_stride) in elements, so
 * both Eigen storage orders (column-major default, row-major when the
 * reference is built with -DEIGEN_DEFAULT_TO_ROW_MAJOR, CMakeLists.txt:59-61)
 * are accepted without a copy on the caller's side.
 */
#ifndef LIBCLUSTER_HIP_H
#define LIBCLUSTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { LC_OK = 0, LC_EINVAL = 1, LC_ERUNTIME = 2, LC_EHIP = 3, LC_EDOMAIN = 4 } lc_status;

/* weight distribution kinds: include/distributions.h:163 (Dirichlet), :103 (StickBreak), :147 (GDirichlet) */
typedef enum { LC_W_DIRICHLET = 0, LC_W_STICKBREAK = 1, LC_W_GDIRICHLET = 2 } lc_weight_kind;
/* learners: include/libcluster.h:177 (learnVDP), :218 (learnBGMM), :356 (learnGMC), :409 (learnSGMC) */
typedef enum {
  LC_ALGO_VDP = 0,  /* learnVDP   cluster.cpp:636  StickBreak + GaussWish */
  LC_ALGO_BGMM = 1, /* learnBGMM  cluster.cpp:667  Dirichlet  + GaussWish */
  LC_ALGO_GMC = 2,  /* learnGMC   cluster.cpp:763  GDirichlet + GaussWish */
  LC_ALGO_SGMC = 3, /* learnSGMC  cluster.cpp:787  Dirichlet  + GaussWish */
  LC_ALGO_DGMM = 4, /* learnDGMM  cluster.cpp:697  Dirichlet  + NormGamma */
  LC_ALGO_BEMM = 5, /* learnBEMM  cluster.cpp:730  Dirichlet  + ExpGamma  */
  LC_ALGO_DGMC = 6, /* learnDGMC  cluster.cpp:811  GDirichlet + NormGamma */
  LC_ALGO_EGMC = 7  /* learnEGMC  cluster.cpp:842  GDirichlet + ExpGamma  */
} lc_algo;
/* cluster parameter distributions (distributions.h:264, 334, 406) */
typedef enum { LC_C_GAUSSWISH = 0, LC_C_NORMGAMMA = 1, LC_C_EXPGAMMA = 2 } lc_ckind;

typedef struct lc_ctx lc_ctx;     /* device-resident data set + qZ + workspaces */
typedef struct lc_model lc_model; /* weights + clusters (+ the context that holds qZ) */
typedef struct lc_tmodel lc_tmodel; /* two-level (SCM / MCM) model: qY, qZ, weights_j, weights_t, clusters(_t) */

const char* lc_last_error(void);
int lc_version(void);
/* Name of the device kernel a dense Gauss-Wishart statistics pass (updateSS, src/cluster.cpp:53-82) runs for D columns
 * and K clusters: "suffstat_kernel" (per-cluster form), "suffstat_feat_kernel" (feature GEMM) or "suffstat_quad_kernel"
 * (3 <= K <= 16 at D = 17 ... 64: four clusters in the four blocks of one matrix instruction) -- what a profiler will list;
 * bench.py prices that kernel. */
const char* lc_statistics_kernel_name(int D, int K);
/* hash of the sources this binary was built from (libcluster_amd/build.py::source_hash); the Python loader compares it
 * with the tree it sits in and rebuilds or refuses a stale binary */
const char* lc_source_hash(void);
/* number of visible HIP devices (0 on a host without a GPU; never fails) */
int lc_device_count(void);

/* ---- constants: include/libcluster.h:122-127, include/distributions.h:39-43 */
double lc_const_converge(void);   /* CONVERGE   = (double)1e-5f */
double lc_const_fengydel(void);   /* FENGYDEL   = CONVERGE/10   */
double lc_const_zerocutoff(void); /* ZEROCUTOFF = (double)0.1f  */
int lc_const_splititer(void);     /* SPLITITER  = 15            */

/* ======================================================================== *
 * Device context.  `stream` is a hipStream_t (NULL = the null stream).
 * ======================================================================== */
int lc_ctx_create(int device, void* stream, lc_ctx** out);
int lc_ctx_destroy(lc_ctx* ctx);
int lc_ctx_set_stream(lc_ctx* ctx, void* stream);
int lc_ctx_synchronize(lc_ctx* ctx);
int lc_ctx_dims(lc_ctx* ctx, int* J, int* D, int64_t* Ntotal, int* K);

/* Upload J groups of observations (the `const vMatrixXd& X` of
 * cluster.cpp:564-566; learnVDP/learnBGMM pass J = 1, cluster.cpp:651/682).
 * Element (n,d) of group j is Xj[j][n*row_stride + d*col_stride].
 * Any D for the diagonal / exponential families; the Gauss-Wishart entry points return LC_EINVAL beyond D = 1024. */
int lc_ctx_set_data(lc_ctx* ctx, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t row_stride,
                    int64_t col_stride);
/* Synthetic K-component full-covariance Gaussian mixture generated on the
 * device (bench workload): x = mu_z + L_z eps, z uniform,
 * Philox-4x32-10 keyed by `seed`, counter = row_offset + row.  Also writes the
 * initial qZ (true label = `hard`, rest (1-hard)/(K-1)).  mu: K x D, L: K x D x D
 * row-major lower Cholesky factors (host pointers). */
int lc_ctx_synth(lc_ctx* ctx, int64_t N, int D, int K, const double* mu, const double* L, uint64_t seed,
                 int64_t row_offset, double hard);
/* Grouped variant (GMC workload): J groups of Nj rows; cdf = J x K cumulative mixing proportions of
 * each group (NULL: uniform labels); group_ids = J global group ids used as Philox counters, so a group is
 * identical on whichever rank generates it (NULL: 0..J-1). */
int lc_ctx_synth_groups(lc_ctx* ctx, int J, const int64_t* Nj, int D, int K, const double* mu, const double* L,
                        const double* cdf, uint64_t seed, const int64_t* group_ids, double hard);
/* rows [row0,row0+n) of group j -> row-major n x D host buffer */
int lc_ctx_get_rows(lc_ctx* ctx, int j, int64_t row0, int64_t n, double* out);

/* qZ (the `vMatrixXd& qZ` of cluster.cpp:179): host <-> device, any strides. */
int lc_ctx_set_qz(lc_ctx* ctx, int j, const double* q, int K, int64_t row_stride, int64_t col_stride);
int lc_ctx_get_qz(lc_ctx* ctx, int j, double* q, int64_t row_stride, int64_t col_stride);
/* rows [row0,row0+n) of group j only (qZ[j].block(row0,0,n,K)) */
int lc_ctx_get_qz_rows(lc_ctx* ctx, int j, int64_t row0, int64_t n, double* q, int64_t row_stride, int64_t col_stride);
/* every group at once: q is [sum_j N_j x K] row-major, the groups' rows concatenated (one transfer) */
int lc_ctx_get_qz_all(lc_ctx* ctx, double* q);
/* every group at once, q[j] = the N_j x K matrix of group j in column-major order (Eigen's default: the `vMatrixXd& qZ`
 * of cluster.cpp:179 as the caller holds it); no transpose on either side */
int lc_ctx_get_qz_all_colmajor(lc_ctx* ctx, double* const* q);
int lc_ctx_fill_qz(lc_ctx* ctx, int K, double value); /* qZ[j].setOnes(N,1): cluster.cpp:583-585 */

/* ---- the hot path ------------------------------------------------------ */
/* vbexpectation (cluster.cpp:91-138) for ALL groups, from the posterior
 * hyper-parameters the reference keeps inside GaussWish (distributions.h:
 * 325-330) and the weights' Elogweight() (distributions.h:113/173):
 *   nu[K], beta[K], m[K*D], iW[K*D*D] row-major, logdW[K], Elogpi[J*K],
 *   active[J*K] (sparse mask: 0 => column zeroed, cluster.cpp:109-112/134-135)
 *   or NULL.  Overwrites the context's qZ with the new responsibilities.
 * Fz  <- -sum_n logZ_n  (cluster.cpp:137, summed over groups as in :221-223)
 * LLk <- K values sum_n q_nk * Eloglike_k(x_n) (data term of cluster.cpp:409-410) or NULL. */
int lc_estep_posterior(lc_ctx* ctx, int K, const double* nu, const double* beta, const double* m, const double* iW,
                       const double* logdW, const double* Elogpi, const unsigned char* active, double* Fz,
                       double* LLk);
/* Same kernel, kernel-level parameters: A[K*D*D] row-major lower-triangular
 * with nu*maha_k(x) = ||A_k (x - m_k)||^2, m[K*D], c[J*K] = Elogpi_jk +
 * 0.5*(sum psi + logdW - D/beta - D ln pi).  LLk here excludes the constant:
 * LLk = sum_n q_nk (log q~_nk - c_jk). */
int lc_estep(lc_ctx* ctx, int K, const double* A, const double* m, const double* c, double* Fz, double* LLk);
/* K x GaussWish::Eloglike(X) (distributions.cpp:356-370) for ALL groups: column k of the
 * context's qZ receives the expected log-likelihood of every observation under cluster k
 * (no weights, no normalisation); fetch it with lc_ctx_get_qz. */
int lc_eloglike(lc_ctx* ctx, int K, const double* nu, const double* beta, const double* m, const double* iW,
                const double* logdW);
/* probutils::mahaldist (probutils.cpp:113-138) on the context's observations: dist[n] = (x_n - mu) A^-1 (x_n - mu)^T
 * for every row of every group (concatenated); A is D x D row-major, symmetric positive definite, else LC_EINVAL
 * "Matrix A is not positive definite".  Overwrites the context's qZ (one scratch column). */
int lc_mahaldist(lc_ctx* ctx, const double* mu, const double* A, double* dist);
/* updateSS (cluster.cpp:53-82) + K x GaussWish::addobs (distributions.cpp:301-313)
 * for ALL groups on the current qZ: Nk[K], xs[K*D], xxs[K*D*D] (row-major),
 * Njk[J*K] (the returned `Njk` of every group).  smask[J*K]: sparse updates,
 * 0 => group j contributes nothing to cluster k (cluster.cpp:67-79); NULL = dense.
 * Any output pointer may be NULL. */
int lc_suffstat(lc_ctx* ctx, const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk);
/* The diagonal / exponential families (NormGamma, ExpGamma: distributions.cpp:418-590).
 * updateSS + K x NormGamma::addobs / ExpGamma::addobs (distributions.cpp:426-438, 533-542) for ALL groups:
 * Nk[K], xs[K*D] = sum_n q_nk x_n, xxs[K*D] = sum_n q_nk x_n.^2 (NULL for ExpGamma), Njk[J*K]. */
int lc_suffstat_diag(lc_ctx* ctx, const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk);
/* vbexpectation with clusters whose Eloglike is separable over dimensions:
 *   log q~_nk = c_jk + sum_d [ w2_kd (x_nd - a_kd)^2 + w1_kd x_nd ]         a, w2, w1: K*D each, c: J*K
 * NormGamma::Eloglike (distributions.cpp:483-492): a = m, w2 = -nu/(2L), w1 = 0,
 *   c = Elogpi + 0.5*(D*(psi(nu) - ln 2pi - 1/beta) - logL).
 * ExpGamma::Eloglike (distributions.cpp:568-572): a = 0, w2 = 0, w1 = -a*ib, c = Elogpi + D*psi(a) - logb.
 * raw != 0: no weights/normalisation, column k of qZ receives log q~ (the Eloglike columns).
 * Fz / LLk as lc_estep. */
int lc_estep_diag(lc_ctx* ctx, int K, const double* a, const double* w2, const double* w1, const double* c, int raw,
                  double* Fz, double* LLk);
/* qZ[j].colwise().sum() for every group (cluster.cpp:62, 546) */
int lc_colsums(lc_ctx* ctx, double* Njk);

/* ---- multi-GPU: rows are sharded one context per rank; the two per-iteration
 * reductions (packed suff-stats, [Fz; LLk]) call this hook on a device buffer.
 * fn must sum `count` doubles in place across ranks, ordered after work already
 * enqueued on `stream`; return 0 on success. */
typedef int (*lc_allreduce_fn)(void* user, void* device_buf, int64_t count, void* stream);
int lc_ctx_set_allreduce(lc_ctx* ctx, lc_allreduce_fn fn, void* user);
/* Native collectives (libcluster_amd/csrc/lc_comm.cpp) -- nothing in the reference corresponds: its loop over groups
 * (cluster.cpp:207-223) is single-process OpenMP; this is that loop distributed.  A context with a communicator sums
 * its packed statistics and [Fz; LL_k] over the ranks on its own stream; the communicator takes precedence over a hook.
 *   RCCL: rank 0 calls lc_comm_unique_id and ships the LC_COMM_ID_BYTES to the other ranks (any channel); every rank
 *         then calls lc_ctx_comm_init_rccl (ncclCommInitRank on the context's device; one rank per GPU;
 *         ncclAllReduce(ncclDouble, ncclSum) over xGMI).  librccl is bound on first use.
 *   host: staged through the POSIX shared-memory object "/lc_comm_<name>", summed in rank order by every rank -- any
 *         placement of the ranks of one node, including several ranks on one GPU (which RCCL refuses). */
#define LC_COMM_ID_BYTES 128
int lc_comm_rccl_available(void); /* 1 when librccl could be loaded */
int lc_comm_unique_id(void* id /* LC_COMM_ID_BYTES */);
int lc_ctx_comm_init_rccl(lc_ctx* ctx, const void* id, int rank, int world);
int lc_ctx_comm_init_host(lc_ctx* ctx, const char* name, int rank, int world);
int lc_ctx_comm_free(lc_ctx* ctx);
/* kind: "rccl" | "host-shm" | "host-local" | "hook" | "none" (static string) */
int lc_ctx_comm_info(lc_ctx* ctx, int* rank, int* world, const char** kind);
/* sum n host doubles over the ranks of the context's communicator / hook (identity without one) */
int lc_ctx_allreduce(lc_ctx* ctx, double* values, int n);
/* How a distributed run is sharded.  0 (default): every rank holds rows of the SAME groups (BGMM / VDP row
 * blocks; also GMC with every group split by rows) -- statistics and per-group counts are summed.
 * 1: every rank holds WHOLE, different groups (GMC) -- cluster statistics, Fz, LL_k and the
 * weights' free energy are summed, the per-group counts N_jk and the group weights stay local. */
int lc_ctx_set_sharding(lc_ctx* ctx, int whole_groups);
/* Statistics pass: skip every (4-row step, cluster) pair whose responsibilities are all exactly 0.0.  Their
 * contribution is exactly zero, so results are bit-identical; it pays off once qZ is mostly hard (late EM
 * iterations on separated data).  Off by default (the dense kernel has no test in its inner loop); always on in the
 * reference's `sparse` mode, whose point is to leave massless (group, cluster) pairs out (cluster.cpp:67-79). */
int lc_ctx_set_skip_zero(lc_ctx* ctx, int on);

/* Device and page-locked blocks released by contexts are cached for re-use (the split search builds a context per
 * attempt); this returns all of them to the driver. */
int lc_trim_cache(void);

/* ---- kernel timing (hipEvents on the context's stream) ------------------ */
int lc_ctx_timing_enable(lc_ctx* ctx, int on);
int lc_ctx_timing_reset(lc_ctx* ctx);
int lc_ctx_timing_get(lc_ctx* ctx, double* estep_ms, int64_t* estep_calls, double* suffstat_ms,
                      int64_t* suffstat_calls);
/* launches of the fused pass (small observations, D <= 16: vbexpectation + the next iteration's updateSS in one kernel) */
int lc_ctx_timing_get_fused(lc_ctx* ctx, double* fused_ms, int64_t* fused_calls);
/* Everything the context timed since the last reset, for the per-rank report of a multi-GPU run (the loop over groups
 * of src/cluster.cpp:207-223, one rank per GPU): out[0..LC_TIMING_FIELDS-1] =
 *   0 estep_ms  1 estep_calls  2 suffstat_ms  3 suffstat_calls  4 fused_ms  5 fused_calls
 *   6 allreduce_ms  7 allreduce_calls   (events around the exchange step: the sum + the wait for the slowest rank)
 *   8 host_stats_ms  9 host_mstep_ms  10 host_estep_ms  11 host_fenergy_ms  12 host_iters
 *     (host wall time per phase of the VBEM iterations: statistics pass + weights update, cluster M-step + parameter
 *      packing, E-step, free energy)
 *   13 estep_diag_mfma_calls  (separable families: how many of the estep_calls took estep_diag_mfma_kernel, the
 *      matrix-pipe form of NormGamma / ExpGamma::Eloglike, distributions.cpp:483-492, 568-581; the rest ran estep_diag_kernel) */
#define LC_TIMING_FIELDS 14
int lc_ctx_timing_get_all(lc_ctx* ctx, double* out, int n);

/* ======================================================================== *
 * Variational Bayes EM on a context (vbem, cluster.cpp:177-239).
 * The model is created on first use (*model == NULL) with `wkind` weights of
 * prior `wprior` and `ckind` clusters (lc_ckind) of prior `clusterprior`.
 * fixed_iters >= 0 runs exactly that many iterations without the convergence
 * and free-energy-increase tests (bench / parity harness: the reference has no
 * public fixed-K entry point).  Ftrace (may be NULL) gets
 * up to ntrace values of F, one per iteration; *niter the iterations run.
 * ======================================================================== */
int lc_vbem(lc_ctx* ctx, lc_model** model, int wkind, int ckind, double wprior, double clusterprior, int maxit,
            int sparse, int fixed_iters, int verbose, unsigned nthreads, double* F, int* niter, double* Ftrace,
            int ntrace);

/* prune_clusters (cluster.cpp:505-552) on a model lc_vbem produced and its context: clusters whose getN() is below
 * ZEROCUTOFF leave the model and their columns leave qZ; every group's weights are then updated with the remaining
 * columns' sums (not renormalised, cluster.cpp:546).  *removed (may be NULL) = how many went.  cluster() calls this
 * between vbem and the split search (cluster.cpp:606). */
int lc_prune(lc_ctx* ctx, lc_model* model, int verbose, int* removed);

/* learnVDP / learnBGMM / learnDGMM / learnBEMM / learnGMC / learnSGMC / learnDGMC / learnEGMC
 * (cluster.cpp:636-873; lc_algo): uploads X,
 * runs the model-selection loop, returns the model (which owns its context so
 * qZ can be fetched).  wprior: StickBreak concentration / Dirichlet alpha the
 * caller's `weights` argument carried (1.0 = default-constructed); ignored
 * for the multi-group learners (default-constructed GDirichlet / Dirichlet per group, cluster.cpp:192).
 * BEMM / EGMC: LC_EINVAL "X has to be in the range [0, inf)!" on a negative observation (cluster.cpp:742, 862). */
/* Environment (the frozen learn*() signatures have no room for it): LIBCLUSTER_GPUS = N | "all" shards the
 * observations over N GPUs of this node inside this one call -- row blocks for the single-matrix learners, whole
 * groups for the GMC family -- one host thread and one context per GPU, RCCL all-reduce of the statistics; results
 * (F, rounds, qZ, weights, clusters) are returned exactly as from one GPU.  `device` is then ignored (GPUs 0..N-1).
 * LIBCLUSTER_COMM = "host" (host-staged sum in rank order, any placement) | "rccl-gather" (ncclAllGather + the same
 * rank-order additions on every GPU: results independent of RCCL's ring order); default ncclAllReduce. */
int lc_learn(int algo, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t row_stride,
             int64_t col_stride, double wprior, double clusterprior, int maxclusters, int sparse, int verbose,
             unsigned nthreads, int device, lc_model** out, double* F);

/* The same with the prior of EVERY group's weight distribution (wprior_j[J], NULL = defaults): what the caller's
 * `std::vector<Dirichlet>& weights` carries into learnSGMC (vbem's weights.resize(J, W()) keeps existing elements,
 * cluster.cpp:192).  GDirichlet has no parameter (ignored there); single-matrix learners use `wprior`. */
int lc_learn_w(int algo, int J, const double* const* Xj, const int64_t* Nj, int D, int64_t row_stride,
               int64_t col_stride, double wprior, const double* wprior_j, double clusterprior, int maxclusters,
               int sparse, int verbose, unsigned nthreads, int device, lc_model** out, double* F);

/* The same model-selection loop (cluster(), cluster.cpp:564-629) on observations that already live in
 * a context (lc_ctx_set_data or lc_ctx_synth): nothing but the M-step statistics crosses PCIe, the
 * split search (partobs / splitobs / auglabels, cluster.cpp:438-470) runs on the device too.
 * wkind / wprior as in lc_vbem (GDirichlet and per-group Dirichlet ignore wprior for new groups).
 * The returned model borrows ctx (keep it alive while reading qZ through the model). */
int lc_cluster(lc_ctx* ctx, int wkind, int ckind, double wprior, double clusterprior, int maxclusters, int sparse,
               int verbose, unsigned nthreads, lc_model** out, double* F);

/* ---- model accessors ----------------------------------------------------- */
int lc_model_free(lc_model* m);
int lc_model_dims(lc_model* m, int* J, int* K, int* D);
int lc_model_rounds(lc_model* m, int* nrounds);                       /* vbem rounds of cluster() */
int lc_model_round(lc_model* m, int r, int* K, int* niter, double* F, int nF); /* F trace of round r */
int lc_model_get_qz(lc_model* m, int j, double* q, int64_t row_stride, int64_t col_stride);
int lc_model_get_qz_all(lc_model* m, double* q); /* all groups, [sum_j N_j x K] row-major */
int lc_model_get_qz_all_colmajor(lc_model* m, double* const* q); /* q[j]: N_j x K column-major (see lc_ctx_...) */
/* WeightDist::Elogweight() / getNk() of group j (K values each; may be NULL) */
int lc_model_weights(lc_model* m, int j, double* Elogweight, double* Nk);
int lc_model_kinds(lc_model* m, int* wkind, int* ckind);
/* Cluster k (any pointer may be NULL).
 * GaussWish: getN(), getmean() [D], getcov() [D*D row-major], posterior nu, beta, iW [D*D], logdW.
 * NormGamma: getN(), getmean() [D], cov [D] = getcov() = L*nu (distributions.h:375), nu, beta, iW [D] = L,
 *            logdW = logL.
 * ExpGamma:  getN(), mean [D] = getrate() = a*ib (distributions.h:433), cov must be NULL, nu = a, beta = 0,
 *            iW [D] = ib, logdW = logb. */
int lc_model_cluster(lc_model* m, int k, double* N, double* mean, double* cov, double* nu, double* beta, double* iW,
                     double* logdW);
int lc_model_fenergy(lc_model* m, double* Fw /*[J]*/, double* Fc /*[K]*/);

/* ======================================================================== *
 * Two-level models: learnSCM (libcluster.h:583-596, scluster.cpp:578-605) and
 * learnMCM (libcluster.h:661-676, mcluster.cpp:613-642).
 * X[j][i]: J groups with Ij[j] "documents" each; Xji holds the sum(Ij) document
 * matrices group-major, document d is Nji[d] x D with the given strides.  Every
 * document is one group of an internal context, so vbeZ (scluster.cpp:93-124 /
 * mcluster.cpp:100-135) is the E-step kernel and the bottom-level M-step is the
 * suff-stat kernel; qY and W (document level, small) stay on the host.
 * Wj: NULL selects the SCM (GDirichlet / Dirichlet(prior_t) / GaussWish(prior_k));
 *     else J pointers to (Ij[j] x Dt) row-major document observations: the MCM
 *     (GDirichlet / Dirichlet() / GaussWish(prior_t, Dt) / GaussWish(prior_k, D)).
 * qY0: NULL = the reference's random start, |U(-1,1)| rows normalised, drawn with
 *     std::rand() (Eigen's Random(), scluster.cpp:519-521); else J pointers to
 *     (Ij[j] x maxT) row-major initial assignments (additive: reproducible runs).
 * Errors: LC_EINVAL for nthreads < 1, and for the SCM maxT > number of documents
 * ("maxT must be less than the number of documents ofX!", scluster.cpp:531-533);
 * LC_ERUNTIME "Free energy increase!" (scluster.cpp:248-249).
 * ======================================================================== */
int lc_learn_topic(int J, const int* Ij, const double* const* Xji, const int64_t* Nji, int D, int64_t row_stride,
                   int64_t col_stride, const double* const* Wj, int Dt, const double* const* qY0, double prior_t,
                   double prior_k, unsigned maxT, int maxK, int verbose, unsigned nthreads, int device,
                   lc_tmodel** out, double* F);
/* The same on one process per GPU: every rank passes WHOLE groups (with all their documents); fn (see
 * lc_ctx_set_allreduce) sums the cluster statistics, N_tk, the document-level Gaussian statistics, Fyz / Fz and the
 * decision counts of the split search, so all ranks take the same decisions.  qY0 (if given) holds this rank's
 * documents.  stream: the HIP stream the context works on (the hook's collectives must be ordered with it). */
int lc_learn_topic_dist(int J, const int* Ij, const double* const* Xji, const int64_t* Nji, int D, int64_t row_stride,
                        int64_t col_stride, const double* const* Wj, int Dt, const double* const* qY0, double prior_t,
                        double prior_k, unsigned maxT, int maxK, int verbose, unsigned nthreads, int device,
                        void* stream, lc_allreduce_fn fn, void* user, lc_tmodel** out, double* F);
int lc_tmodel_free(lc_tmodel* m);
int lc_tmodel_dims(lc_tmodel* m, int* J, int* Itot, int* T, int* K, int* D, int* Dt);
int lc_tmodel_get_qy(lc_tmodel* m, int j, double* qY /* Ij[j] x T row-major */);
int lc_tmodel_get_qz(lc_tmodel* m, int doc, double* q, int64_t row_stride, int64_t col_stride); /* Nji[doc] x K */
int lc_tmodel_get_qz_all(lc_tmodel* m, double* q); /* all documents, [sum N_ji x K] row-major */
int lc_tmodel_get_qz_all_colmajor(lc_tmodel* m, double* const* q); /* q[doc]: N_ji x K column-major, documents in (j, i) order */
/* level 0: weights_j[idx] (T values each); level 1: weights_t[idx] (K values each) */
int lc_tmodel_weights(lc_tmodel* m, int level, int idx, double* Elogweight, double* Nk);
/* level 0: bottom-level clusters[idx] (D); level 1: top-level clusters_t[idx] (Dt, MCM only) */
int lc_tmodel_cluster(lc_tmodel* m, int level, int idx, double* N, double* mean, double* cov, double* nu, double* beta,
                      double* iW, double* logdW, double* fenergy);
int lc_tmodel_rounds(lc_tmodel* m, int* nrounds);
int lc_tmodel_round(lc_tmodel* m, int r, int* T, int* K, int* niter, double* F, int nF);

/* ======================================================================== *
 * Host-only pieces of the path (no GPU needed; used by the C++ facade classes
 * and by tests of the host arithmetic).
 * ======================================================================== */
double lc_digamma(double x); /* boost::math::digamma in the reference (probutils.cpp:213) */
/* WeightDist::update(Nk) + Elogweight() + fenergy(): distributions.cpp:124-168, 186-196, 242-266 */
int lc_weights_update(int wkind, double wprior, const double* Nk, int K, double* Elogweight, double* fenergy);
/* GaussWish: clearobs + addobs-sums + update (distributions.cpp:316-337) from
 * reduced statistics; outputs the posterior, the free energy (:388-399), the
 * E-step whitener A (D*D) and Eloglike constant.  Any output may be NULL. */
int lc_gw_mstep(double clustwidth, int D, double Ns, const double* xs, const double* xxs, double* nu, double* beta,
                double* m, double* iW, double* logdW, double* fenergy, double* A, double* eloglike_const);
/* NormGamma: addobs sums -> update() (distributions.cpp:441-464), fenergy (:508-517), Eloglike constant (:486-488).
 * xs, xxs: D values each (sum q x, sum q x.^2). */
int lc_ng_mstep(double clustwidth, int D, double Ns, const double* xs, const double* xxs, double* nu, double* beta,
                double* m, double* L, double* logL, double* fenergy, double* eloglike_const);
/* ExpGamma: addobs sums -> update() (distributions.cpp:545-552), fenergy (:584-589), Eloglike constant (:570). */
int lc_eg_mstep(double obsmag, int D, double Ns, const double* xs, double* a, double* ib, double* logb,
                double* fenergy, double* eloglike_const);

#ifdef __cplusplus
}
#endif
#endif /* LIBCLUSTER_HIP_H */
