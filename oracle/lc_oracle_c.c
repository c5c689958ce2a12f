/* CPU port of the reference's E-step / suff-stat arithmetic, for TIMING the
 * CPU baseline beside the GPU path (bench.py `cpu_baseline`, kind "port") and
 * as a second checker.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 * PARITY UNPINNED: the reference cannot be built here (needs Eigen + Boost);
 * this file restates its arithmetic and is validated against the numpy oracle
 * (tests/test_oracle_c.py), which in turn is pinned by scikit-learn/scipy.
 *
 * Structure follows the reference, one pass over X per cluster:
 *   lco_estep     vbexpectation   src/cluster.cpp:91-138
 *                 Eloglike        src/distributions.cpp:356-370
 *                 mahaldist       src/probutils.cpp:113-138 (factor, solve A y = (x-mu), dot)
 *                 logsumexp       src/probutils.cpp:141-150
 *   lco_suffstat  updateSS        src/cluster.cpp:53-82
 *                 addobs          src/distributions.cpp:301-313 (qX = diag(q) X; N, sum, (qX)^T X full DxD GEMM)
 * The reference runs these on ONE thread for learnBGMM/learnVDP (its OpenMP
 * loops are over groups, cluster.cpp:207-223, and Eigen's threading is off,
 * CMakeLists.txt:70).  `nthreads` > 1 here chunks ROWS over threads -- the
 * "what OpenMP could do" baseline the 10x target is quoted against.
 * Rows are processed in tiles of RB so the compiler vectorises across rows
 * (the reference gets its SIMD from Eigen's blocked triangular solves / GEMM).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC lc_oracle_c.c -o _build/liblc_oracle_c.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RB 32 /* rows per tile: 4 independent AVX-512 accumulators hide the FMA latency */

/* lower Cholesky of the row-major D x D matrix A into L (row-major); 0 on success */
static int chol(const double* A, int D, double* L) {
  memset(L, 0, sizeof(double) * (size_t)D * D);
  for (int j = 0; j < D; ++j) {
    double d = A[(size_t)j * D + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * D + k] * L[(size_t)j * D + k];
    if (!(d > 0.0)) return 1;
    const double ljj = sqrt(d);
    L[(size_t)j * D + j] = ljj;
    for (int i = j + 1; i < D; ++i) {
      double s = A[(size_t)i * D + j];
      for (int k = 0; k < j; ++k) s -= L[(size_t)i * D + k] * L[(size_t)j * D + k];
      L[(size_t)i * D + j] = s / ljj;
    }
  }
  return 0;
}

/* E-step.  X: N x D row-major.  Posterior per cluster: nu, beta, m[K*D],
 * iW[K*D*D], logdW, sumpsi (= sum_d digamma((nu+1-d)/2), computed by the
 * caller -- Boost.Math in the reference, scipy in the harness), Elogpi[K].
 * qZ: N x K row-major out.  Returns -sum_n logZ_n through Fz.  Returns 0, or 1
 * if some iW is not positive definite (probutils.cpp:131-132). */
int lco_estep(const double* X, int64_t N, int D, int K, const double* nu, const double* beta, const double* m,
              const double* iW, const double* logdW, const double* sumpsi, const double* Elogpi, double* qZ,
              double* Fz, int nthreads) {
  double* Ls = (double*)malloc(sizeof(double) * (size_t)K * D * D);
  if (!Ls) return 2;
  for (int k = 0; k < K; ++k)
    if (chol(iW + (size_t)k * D * D, D, Ls + (size_t)k * D * D)) {
      free(Ls);
      return 1;
    }
  const double lnpi = log(3.14159265358979323846264338327950288);
  double fz = 0.0;
  const int64_t ntile = (N + RB - 1) / RB;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : fz) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t t = 0; t < ntile; ++t) {
    const int64_t r0 = t * RB;
    const int nr = (int)((N - r0) < RB ? (N - r0) : RB);
    double xm[128][RB], y[128][RB], z[128][RB];
    double logq[RB];
    for (int k = 0; k < K; ++k) { /* one "Eloglike" per cluster, cluster.cpp:120-121 */
      const double* L = Ls + (size_t)k * D * D;
      const double* mk = m + (size_t)k * D;
      for (int d = 0; d < D; ++d) /* X_mu = (X - mu)^T, probutils.cpp:135 */
        for (int r = 0; r < RB; ++r) xm[d][r] = r < nr ? X[(size_t)(r0 + r) * D + d] - mk[d] : 0.0;
      /* Aldl.solve(X_mu): forward then backward substitution, probutils.cpp:136 */
      for (int i = 0; i < D; ++i) {
        double s[RB];
        for (int r = 0; r < RB; ++r) s[r] = xm[i][r];
        for (int j = 0; j < i; ++j) {
          const double l = L[(size_t)i * D + j];
          for (int r = 0; r < RB; ++r) s[r] -= l * y[j][r];
        }
        const double inv = 1.0 / L[(size_t)i * D + i];
        for (int r = 0; r < RB; ++r) y[i][r] = s[r] * inv;
      }
      for (int i = D - 1; i >= 0; --i) {
        double s[RB];
        for (int r = 0; r < RB; ++r) s[r] = y[i][r];
        for (int j = i + 1; j < D; ++j) {
          const double l = L[(size_t)j * D + i];
          for (int r = 0; r < RB; ++r) s[r] -= l * z[j][r];
        }
        const double inv = 1.0 / L[(size_t)i * D + i];
        for (int r = 0; r < RB; ++r) z[i][r] = s[r] * inv;
      }
      double maha[RB];
      for (int r = 0; r < RB; ++r) maha[r] = 0.0;
      for (int d = 0; d < D; ++d) /* (X_mu .* solve).colwise().sum(), :136-137 */
        for (int r = 0; r < RB; ++r) maha[r] += xm[d][r] * z[d][r];
      const double cst = sumpsi[k] + logdW[k] - D * (1.0 / beta[k] + lnpi); /* distributions.cpp:363 */
      for (int r = 0; r < nr; ++r) qZ[(size_t)(r0 + r) * K + k] = Elogpi[k] + 0.5 * (cst - nu[k] * maha[r]);
    }
    for (int r = 0; r < nr; ++r) { /* logsumexp + normalise, probutils.cpp:141-150, cluster.cpp:124-131 */
      double* row = qZ + (size_t)(r0 + r) * K;
      double mx = row[0];
      for (int k = 1; k < K; ++k) mx = row[k] > mx ? row[k] : mx;
      double se = 0.0;
      for (int k = 0; k < K; ++k) se += exp(row[k] - mx);
      const double lz = log(se) + mx;
      for (int k = 0; k < K; ++k) row[k] = exp(row[k] - lz);
      logq[r] = lz;
      fz += lz;
    }
    (void)logq;
  }
  free(Ls);
  *Fz = -fz;
  return 0;
}

/* Suff-stats.  qZ: N x K row-major.  Nk[K], xs[K*D], xxs[K*D*D] are
 * overwritten.  Per cluster: qX = diag(q_k) X (an N x D temporary in the
 * reference), N_s += sum q, x_s += colsum(qX), xx_s += qX^T X (the full D x D
 * product, as Eigen computes it).  Row-parallel structure: every thread owns a
 * contiguous block of rows and walks it in tiles of TB rows; a tile of X is
 * read ONCE and serves all K clusters (the per-thread accumulators, K records,
 * stream from L2/L3), so the pass is bound by the FMAs, not by re-reading X K
 * times as a literal per-cluster loop over all rows would be. */
#define TB 128 /* rows per tile: 64 KB of X at D = 64 stays in L2 across the K clusters */
int lco_suffstat(const double* X, const double* qZ, int64_t N, int D, int K, double* Nk, double* xs, double* xxs,
                 int nthreads) {
  const size_t SS = 1 + (size_t)D + (size_t)D * D;
  int nt = nthreads > 0 ? nthreads : 1;
  double* acc = (double*)calloc((size_t)nt * K * SS, sizeof(double));
  if (!acc) return 2;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
    const int tid = 0, nth = 1;
#endif
    const int64_t ntile = (N + TB - 1) / TB;
    const int64_t t0 = ntile * tid / nth, t1 = ntile * (tid + 1) / nth;
    double* my = acc + (size_t)tid * K * SS;
    double(*qx)[128] = (double(*)[128])malloc(sizeof(double) * TB * 128); /* qZkX = qZk.asDiagonal() * X, distributions.cpp:308 */
    for (int64_t t = t0; t < t1 && qx; ++t) {
      const int64_t r0 = t * TB;
      const int nr = (int)((N - r0) < TB ? (N - r0) : TB);
      for (int k = 0; k < K; ++k) { /* one "addobs" per cluster, cluster.cpp:75-79 */
        double* a = my + (size_t)k * SS;
        double* ax = a + 1;
        double* axx = a + 1 + D;
        for (int r = 0; r < nr; ++r) {
          const double q = qZ[(size_t)(r0 + r) * K + k];
          const double* x = X + (size_t)(r0 + r) * D;
          a[0] += q;
          for (int i = 0; i < D; ++i) {
            qx[r][i] = q * x[i];
            ax[i] += qx[r][i];
          }
        }
        /* xx_s += qZkX^T * X (:312): 4 rows of the accumulator at a time share every load of x */
        int i = 0;
        for (; i + 4 <= D; i += 4) {
          double* w0 = axx + (size_t)i * D;
          double* w1 = w0 + D;
          double* w2 = w1 + D;
          double* w3 = w2 + D;
          for (int r = 0; r < nr; ++r) {
            const double v0 = qx[r][i], v1 = qx[r][i + 1], v2 = qx[r][i + 2], v3 = qx[r][i + 3];
            const double* x = X + (size_t)(r0 + r) * D;
#pragma omp simd
            for (int j = 0; j < D; ++j) {
              const double xj = x[j];
              w0[j] += v0 * xj;
              w1[j] += v1 * xj;
              w2[j] += v2 * xj;
              w3[j] += v3 * xj;
            }
          }
        }
        for (; i < D; ++i) {
          double* row = axx + (size_t)i * D;
          for (int r = 0; r < nr; ++r) {
            const double v = qx[r][i];
            const double* x = X + (size_t)(r0 + r) * D;
#pragma omp simd
            for (int j = 0; j < D; ++j) row[j] += v * x[j];
          }
        }
      }
    }
    free(qx);
  }
  for (int k = 0; k < K; ++k) {
    Nk[k] = 0.0;
    memset(xs + (size_t)k * D, 0, sizeof(double) * D);
    memset(xxs + (size_t)k * D * D, 0, sizeof(double) * (size_t)D * D);
  }
  /* fold the per-thread records in thread order (deterministic), clusters in parallel */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
  for (int k = 0; k < K; ++k) {
    for (int t = 0; t < nt; ++t) {
      const double* a = acc + ((size_t)t * K + k) * SS;
      Nk[k] += a[0];
      for (int i = 0; i < D; ++i) xs[(size_t)k * D + i] += a[1 + i];
      for (size_t i = 0; i < (size_t)D * D; ++i) xxs[(size_t)k * D * D + i] += a[1 + D + i];
    }
  }
  free(acc);
  return 0;
}

int lco_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
