// Reductions, column sums, fills; the split-search data passes (partobs / splitobs / auglabels); the synthetic mixture generator
// (one translation unit per kernel family; the file header of lc_kernels_estep.hip maps kernels to the reference)
#include "lc_device.hpp"

namespace lck {

// ===========================================================================
// small helpers
// ===========================================================================
// out[e] = sum_c partial[c][e] in a fixed order.  Block = 16 consecutive elements x 16 part lanes: a part lane
// adds every 16th record with four independent accumulators, then the lanes are folded by a tree in LDS.
// (One thread per element walking all records serially was latency-bound: 1.5 ms for 4096 records.)
__global__ void __launch_bounds__(256) reduce_partials_kernel(const double* __restrict__ partial, int nparts, int64_t n,
                                                              double* __restrict__ out) {
  __shared__ double sh[16][17];
  const int ex = threadIdx.x & 15, py = threadIdx.x >> 4;
  const int64_t e = (int64_t)blockIdx.x * 16 + ex;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e < n) {
    const double* p = partial + e;
    int c = py;
    for (; c + 48 < nparts; c += 64) {
      s0 += p[(int64_t)c * n];
      s1 += p[(int64_t)(c + 16) * n];
      s2 += p[(int64_t)(c + 32) * n];
      s3 += p[(int64_t)(c + 48) * n];
    }
    for (; c < nparts; c += 16) s0 += p[(int64_t)c * n];
  }
  sh[py][ex] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int w = 8; w > 0; w >>= 1) {
    if (py < w) sh[py][ex] += sh[py + w][ex];
    __syncthreads();
  }
  if (py == 0 && e < n) out[e] = sh[0][ex];
}

// few elements, many parts: one block per element, fixed-shape strided sum + tree
__global__ void __launch_bounds__(256) reduce_cols_kernel(const double* partial, int nparts, int64_t n, double* out) {
  __shared__ double sh[256];
  const int64_t e = blockIdx.x;
  double s = 0.0;
  for (int c = threadIdx.x; c < nparts; c += 256) s += partial[(int64_t)c * n + e];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[e] = sh[0];
}

// Sparse statistics: records exist only for (row chunk, active cluster) pairs; cluster k sums the records listed in
// krec[kptr[k] .. kptr[k+1]) in list order (fixed => deterministic).  Same 16 x 16 tile as reduce_partials_kernel.
__global__ void __launch_bounds__(256) reduce_records_kernel(const double* __restrict__ partial, int64_t n,
                                                             const int* __restrict__ kptr, const int* __restrict__ krec,
                                                             double* __restrict__ out) {
  __shared__ double sh[16][17];
  const int ex = threadIdx.x & 15, py = threadIdx.x >> 4, k = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 16 + ex;
  const int b = kptr[k], en = kptr[k + 1];
  double s0 = 0.0, s1 = 0.0;
  if (e < n) {
    int c = b + py;
    for (; c + 16 < en; c += 32) {
      s0 += partial[(int64_t)krec[c] * n + e];
      s1 += partial[(int64_t)krec[c + 16] * n + e];
    }
    for (; c < en; c += 16) s0 += partial[(int64_t)krec[c] * n + e];
  }
  sh[py][ex] = s0 + s1;
  __syncthreads();
  for (int w = 8; w > 0; w >>= 1) {
    if (py < w) sh[py][ex] += sh[py + w][ex];
    __syncthreads();
  }
  if (py == 0 && e < n) out[(int64_t)k * n + e] = sh[0][ex];
}

hipError_t launch_reduce_records(const double* partial, int64_t n, int K, const int* kptr, const int* krec, double* out,
                                 hipStream_t stream) {
  if (n <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_records_kernel, dim3((unsigned)((n + 15) / 16), (unsigned)K), dim3(256), 0, stream, partial,
                     n, kptr, krec, out);
  return hipGetLastError();
}

// very many records of a few elements (the per-block F_z / LL_k partials of an E-step over 10^7 rows): 64 blocks
// per element sum contiguous record ranges into tmp[e][64] (each with the fixed-shape tree above), a second
// launch folds the 64.  Same summation order for a given (nparts, n) => deterministic.
__global__ void __launch_bounds__(256) reduce_cols_stage1_kernel(const double* partial, int nparts, int64_t n,
                                                                 double* tmp) {
  __shared__ double sh[256];
  const int64_t e = blockIdx.y;
  const int per = (nparts + 63) / 64, c0 = blockIdx.x * per;
  const int c1 = c0 + per < nparts ? c0 + per : nparts;
  double s = 0.0;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) s += partial[(int64_t)c * n + e];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) tmp[e * 64 + blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(64) reduce_cols_stage2_kernel(const double* tmp, double* out) {
  double v = tmp[(int64_t)blockIdx.x * 64 + threadIdx.x];
  v = wave_sum(v);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

hipError_t launch_reduce_partials(const double* partial, int nparts, int64_t n, double* out, hipStream_t stream,
                                  double* tmp) {
  if (n <= 0) return hipSuccess;
  if (tmp && nparts > 8192 && n <= REDUCE_TMP_ELEMS) {
    hipLaunchKernelGGL(reduce_cols_stage1_kernel, dim3(64, (unsigned)n), dim3(256), 0, stream, partial, nparts, n, tmp);
    hipLaunchKernelGGL(reduce_cols_stage2_kernel, dim3((unsigned)n), dim3(64), 0, stream, tmp, out);
  } else if (nparts > 512 && n <= 4096) {
    hipLaunchKernelGGL(reduce_cols_kernel, dim3((unsigned)n), dim3(256), 0, stream, partial, nparts, n, out);
  } else {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, partial,
                       nparts, n, out);
  }
  return hipGetLastError();
}

// one block per (k, j): fixed-shape tree => deterministic
__global__ void __launch_bounds__(256) group_colsum_kernel(const double* qZ, int64_t ldq, int K, const int64_t* goff,
                                                           double* out) {
  __shared__ double sh[256];
  const int k = blockIdx.x, j = blockIdx.y;
  const int64_t b = goff[j], e = goff[j + 1];
  double s = 0.0;
  for (int64_t r = b + threadIdx.x; r < e; r += 256) s += qZ[(int64_t)k * ldq + r];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(int64_t)j * K + k] = sh[0];
}

// many small groups (the documents of learnSCM / learnMCM): one block per group, wave w sums the columns
// w, w+4, ... with a fixed-shape reduction => deterministic
__global__ void __launch_bounds__(256) group_colsum_small_kernel(const double* qZ, int64_t ldq, int K,
                                                                 const int64_t* goff, double* out) {
  const int j = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = goff[j], e = goff[j + 1];
  for (int k = w; k < K; k += 4) {
    double s = 0.0;
    for (int64_t r = b + lane; r < e; r += 64) s += qZ[(int64_t)k * ldq + r];
    s = wave_sum(s);
    if (lane == 0) out[(int64_t)j * K + k] = s;
  }
}

// few groups with many rows each (one block per (k, j) would leave most of the chip idle: 1.5 ms per million rows at
// K = 32, J = 1): 64 row slices per (k, j), then a fixed-order sum of the 64 partials => still deterministic
constexpr int GCS_SLICES = 64;
__global__ void __launch_bounds__(256) group_colsum_slice_kernel(const double* qZ, int64_t ldq, int K,
                                                                 const int64_t* goff, double* tmp) {
  __shared__ double sh[256];
  const int k = blockIdx.x, j = blockIdx.y, sl = blockIdx.z;
  const int64_t b = goff[j], e = goff[j + 1], len = (e - b + GCS_SLICES - 1) / GCS_SLICES;
  const int64_t r0 = b + sl * len, r1 = r0 + len < e ? r0 + len : e;
  double s = 0.0;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) s += qZ[(int64_t)k * ldq + r];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) tmp[((int64_t)j * K + k) * GCS_SLICES + sl] = sh[0];
}
__global__ void __launch_bounds__(256) group_colsum_fold_kernel(const double* tmp, int n, double* out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  double s = 0.0;
  for (int i = 0; i < GCS_SLICES; ++i) s += tmp[(int64_t)t * GCS_SLICES + i];
  out[t] = s;
}

hipError_t launch_group_colsum(const double* qZ, int64_t ldq, int K, const int64_t* goff, int J, double* out,
                               hipStream_t stream, double* tmp, int64_t rows) {
  if (K <= 0 || J <= 0) return hipSuccess;
  if (tmp && (int64_t)J * K * GCS_SLICES <= (int64_t)REDUCE_TMP_ELEMS * 64 && rows >= (int64_t)J * 65536) {
    hipLaunchKernelGGL(group_colsum_slice_kernel, dim3((unsigned)K, (unsigned)J, GCS_SLICES), dim3(256), 0, stream, qZ,
                       ldq, K, goff, tmp);
    hipLaunchKernelGGL(group_colsum_fold_kernel, dim3((unsigned)((J * K + 255) / 256)), dim3(256), 0, stream, tmp, J * K,
                       out);
    return hipGetLastError();
  }
  if (J > 1024)
    hipLaunchKernelGGL(group_colsum_small_kernel, dim3((unsigned)J), dim3(256), 0, stream, qZ, ldq, K, goff, out);
  else
    hipLaunchKernelGGL(group_colsum_kernel, dim3((unsigned)K, (unsigned)J), dim3(256), 0, stream, qZ, ldq, K, goff,
                       out);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) fill_qz_kernel(double* qZ, int64_t ldq, int K, const int* rginfo,
                                                      int64_t nrows, int64_t NP, double value) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= NP) return;
  bool ok;
  if (rginfo)
    ok = (row & 15) < (rginfo[row >> 4] & 31);
  else
    ok = row < nrows;
  const double v = ok ? value : 0.0;
  for (int k = 0; k < K; ++k) qZ[(int64_t)k * ldq + row] = v;
}

hipError_t launch_fill_qz(double* qZ, int64_t ldq, int K, const int* rginfo, int64_t nrows, int64_t nrg, double value,
                          hipStream_t stream) {
  const int64_t NP = nrg * RG;
  if (NP <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_qz_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, stream, qZ, ldq, K, rginfo,
                     nrows, NP, value);
  return hipGetLastError();
}


// ===========================================================================
// split-search data passes: partobs / splitobs / auglabels
// ===========================================================================
// partobs (src/comutils.cpp:56-72) selects the rows with q_k > 0.5 in order.  Two passes over the
// column: per-block counts, then (after the host scans the ~N/1024 counts) an ordered compaction.
constexpr int SEL_ROWS = 1024;  // rows per 256-thread block, 4 consecutive rows per thread

__global__ void __launch_bounds__(256) select_count_kernel(const double* qcol, int64_t NP, double thresh,
                                                           int* counts) {
  __shared__ int sh[256];
  const int64_t r0 = (int64_t)blockIdx.x * SEL_ROWS + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (r0 + i < NP && qcol[r0 + i] > thresh) ++c;
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(256) select_compact_kernel(const double* qcol, int64_t NP, double thresh,
                                                             const int64_t* offsets, int64_t* idx) {
  __shared__ int sh[256];
  const int64_t r0 = (int64_t)blockIdx.x * SEL_ROWS + threadIdx.x * 4;
  bool f[4];
  int c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[i] = r0 + i < NP && qcol[r0 + i] > thresh;
    c += f[i] ? 1 : 0;
  }
  sh[threadIdx.x] = c;
  __syncthreads();
  // inclusive Hillis-Steele scan over the 256 per-thread counts
  for (int d = 1; d < 256; d <<= 1) {
    const int v = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  int64_t pos = offsets[blockIdx.x] + sh[threadIdx.x] - c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (f[i]) idx[pos++] = r0 + i;
}

hipError_t launch_select_count(const double* qcol, int64_t NP, double thresh, int* counts, hipStream_t stream) {
  if (NP <= 0) return hipSuccess;
  const unsigned nb = (unsigned)((NP + SEL_ROWS - 1) / SEL_ROWS);
  hipLaunchKernelGGL(select_count_kernel, dim3(nb), dim3(256), 0, stream, qcol, NP, thresh, counts);
  return hipGetLastError();
}

hipError_t launch_select_compact(const double* qcol, int64_t NP, double thresh, const int64_t* offsets, int64_t* idx,
                                 hipStream_t stream) {
  if (NP <= 0) return hipSuccess;
  const unsigned nb = (unsigned)((NP + SEL_ROWS - 1) / SEL_ROWS);
  hipLaunchKernelGGL(select_compact_kernel, dim3(nb), dim3(256), 0, stream, qcol, NP, thresh, offsets, idx);
  return hipGetLastError();
}
int select_blocks(int64_t NP) { return (int)((NP + SEL_ROWS - 1) / SEL_ROWS); }

// starts[j] = first position p with idx[p] >= goff[j]  (idx ascending), j = 0..J
__global__ void group_starts_kernel(const int64_t* idx, int64_t M, const int64_t* goff, int J, int64_t* starts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > J) return;
  const int64_t key = goff[j];
  int64_t lo = 0, hi = M;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (idx[mid] < key) lo = mid + 1; else hi = mid;
  }
  starts[j] = lo;
}

hipError_t launch_group_starts(const int64_t* idx, int64_t M, const int64_t* goff, int J, int64_t* starts,
                               hipStream_t stream) {
  hipLaunchKernelGGL(group_starts_kernel, dim3((unsigned)((J + 1 + 63) / 64)), dim3(64), 0, stream, idx, M, goff, J,
                     starts);
  return hipGetLastError();
}

// position p of the selection -> (group j, destination row in the gathered, re-padded layout)
__device__ __forceinline__ int64_t sel_dst_row(int64_t p, const int64_t* starts, const int64_t* goff_sub, int J) {
  int lo = 0, hi = J;  // largest j with starts[j] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= p) lo = mid; else hi = mid;
  }
  return goff_sub[lo] + (p - starts[lo]);
}

// Xk = X(rows idx): partobs' copy, device to device; one thread per (selected row, double2)
__global__ void __launch_bounds__(256) gather_rows_kernel(const double* X, int DP, const int64_t* idx, int64_t M,
                                                          const int64_t* starts, const int64_t* goff_sub, int J,
                                                          double* Xdst) {
  const int per = DP / 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * per) return;
  const int64_t p = t / per;
  const int c2 = (int)(t % per);
  const int64_t dst = sel_dst_row(p, starts, goff_sub, J);
  reinterpret_cast<double2*>(Xdst + dst * DP)[c2] = reinterpret_cast<const double2*>(X + idx[p] * DP)[c2];
}

// K columns of a column-major table for the selected rows, into the gathered layout; one thread per selected row
__global__ void __launch_bounds__(256) gather_cols_kernel(const double* src, int64_t lds, int K, const int64_t* idx,
                                                          int64_t M, const int64_t* starts, const int64_t* goff_sub, int J,
                                                          double* dst, int64_t ldd) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t s = idx[p], d = sel_dst_row(p, starts, goff_sub, J);
  for (int c = 0; c < K; ++c) dst[(int64_t)c * ldd + d] = src[(int64_t)c * lds + s];
}

// the same from a ROW-major table (row stride lds): one thread per (selected row, column) -- a row's K values are
// contiguous on the source side, the K columns of the gathered layout on the destination side
__global__ void __launch_bounds__(256) gather_rowmajor_kernel(const double* src, int64_t lds, int K, const int64_t* idx,
                                                              int64_t M, const int64_t* starts, const int64_t* goff_sub,
                                                              int J, double* dst, int64_t ldd) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * K) return;
  const int64_t p = t / K;
  const int c = (int)(t % K);
  const int64_t d = sel_dst_row(p, starts, goff_sub, J);
  dst[(int64_t)c * ldd + d] = src[idx[p] * lds + c];
}
hipError_t launch_gather_rowmajor(const double* src, int64_t lds, int K, const int64_t* idx, int64_t M,
                                  const int64_t* starts, const int64_t* goff_sub, int J, double* dst, int64_t ldd,
                                  hipStream_t stream) {
  if (M <= 0 || K <= 0) return hipSuccess;
  const int64_t n = M * K;
  hipLaunchKernelGGL(gather_rowmajor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, lds, K, idx, M,
                     starts, goff_sub, J, dst, ldd);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) gather_rowmajor_cols_kernel(const double* src, int64_t lds, const int* cols, int nc,
                                                                   const int64_t* idx, int64_t M, const int64_t* starts,
                                                                   const int64_t* goff_sub, int J, double* dst, int64_t ldd) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * nc) return;
  const int64_t p = t / nc;
  const int c = (int)(t % nc);
  const int64_t d = sel_dst_row(p, starts, goff_sub, J);
  dst[(int64_t)c * ldd + d] = src[idx[p] * lds + cols[c]];
}
hipError_t launch_gather_rowmajor_cols(const double* src, int64_t lds, const int* cols, int nc, const int64_t* idx, int64_t M,
                                       const int64_t* starts, const int64_t* goff_sub, int J, double* dst, int64_t ldd,
                                       hipStream_t stream) {
  if (M <= 0 || nc <= 0) return hipSuccess;
  const int64_t n = M * nc;
  hipLaunchKernelGGL(gather_rowmajor_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, lds, cols, nc,
                     idx, M, starts, goff_sub, J, dst, ldd);
  return hipGetLastError();
}

hipError_t launch_gather_cols(const double* src, int64_t lds, int K, const int64_t* idx, int64_t M, const int64_t* starts,
                              const int64_t* goff_sub, int J, double* dst, int64_t ldd, hipStream_t stream) {
  if (M <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, src, lds, K, idx, M,
                     starts, goff_sub, J, dst, ldd);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const double* X, int DP, const int64_t* idx, int64_t M, const int64_t* starts,
                              const int64_t* goff_sub, int J, double* Xdst, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  const int64_t n = M * (DP / 2);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, X, DP, idx, M, starts,
                     goff_sub, J, Xdst);
  return hipGetLastError();
}

// splitobs + the initial split responsibilities (cluster.cpp:446-449).  mode 0 (GaussWish
// distributions.cpp:373-385, NormGamma :495-505): q0 = (sum_d (x_d - m_d) v_d >= 0); mode 1 (first pass of
// ExpGamma :575-581): q0 = sum_d x_d v_d (the projection itself, for the per-group mean); mode 2 (second
// pass): q0 = (q0 > thr[group]).  q1 = 1 - q0 in modes 0 and 2; pad rows 0.  mv = [m(DP), v(DP)].
// Eight lanes per row: lane t of a row takes the column pairs 2 t + 16 j (a row's eight 16-byte loads are 128 contiguous
// bytes), the partial sums meet through three shuffles.  (One lane per row walked its 512-byte row 8 bytes at a time,
// every load instruction touching 64 cache lines: 0.82 ms for 300k rows, 0.2 TB/s -- 5 % of a model-selection run.)
__global__ void __launch_bounds__(256) split_init_kernel(const double* X, int DP, int D, int64_t NP, const int* rginfo,
                                                         int64_t nrows, const double* mv, double* q, int64_t ldq,
                                                         int mode, const double* thr) {
  const int t = threadIdx.x & 7;
  const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool inb = row < NP;
  int grp = 0;
  bool ok = false;
  if (inb) {
    if (rginfo) {
      const int info = rginfo[row >> 4];
      grp = info >> 5;
      ok = (int)(row & 15) < (info & 31);
    } else {
      ok = row < nrows;
    }
  }
  double q0 = 0.0, q1 = 0.0;
  if (mode == 2) {
    if (ok) {
      q0 = q[row] > thr[grp] ? 1.0 : 0.0;
      q1 = 1.0 - q0;
    }
  } else {
    double s = 0.0;
    if (ok) {  // (DP is a multiple of 16 and the pad columns of X are zero: whole column pairs, no tail)
      const double* xr = X + row * DP;
      for (int d = 2 * t; d < DP; d += 16) {
        const double2 x = *reinterpret_cast<const double2*>(xr + d);
        const double m0 = mode == 0 ? mv[d] : 0.0, m1 = mode == 0 ? mv[d + 1] : 0.0;
        const double v0 = d < D ? mv[DP + d] : 0.0, v1 = d + 1 < D ? mv[DP + d + 1] : 0.0;
        s += (x.x - m0) * v0;
        s += (x.y - m1) * v1;
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (ok) {
      if (mode == 0) {
        q0 = s >= 0.0 ? 1.0 : 0.0;
        q1 = 1.0 - q0;
      } else {
        q0 = s;
      }
    }
  }
  if (inb && t == 0) {
    q[row] = q0;
    q[ldq + row] = q1;
  }
}

hipError_t launch_split_init(const double* X, int DP, int D, int64_t NP, const int* rginfo, int64_t nrows,
                             const double* mv, double* q, int64_t ldq, int mode, const double* thr,
                             hipStream_t stream) {
  if (NP <= 0) return hipSuccess;
  hipLaunchKernelGGL(split_init_kernel, dim3((unsigned)((NP + 31) / 32)), dim3(256), 0, stream, X, DP, D, NP, rginfo,
                     nrows, mv, q, ldq, mode, thr);
  return hipGetLastError();
}

// auglabels (src/comutils.cpp:75-104) straight from the refined sub-problem: every selected row whose
// second refined responsibility exceeds 0.5 moves its column-k mass to the new column K
__global__ void __launch_bounds__(256) aug_from_sub_kernel(double* q, int64_t ldq, int k, int K, const int64_t* idx,
                                                           int64_t M, const int64_t* starts, const int64_t* goff_sub,
                                                           int J, const double* qsub1, int64_t* qhash) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t sub = sel_dst_row(p, starts, goff_sub, J);
  if (qsub1[sub] > 0.5) {
    const int64_t r = idx[p];
    q[(int64_t)K * ldq + r] = q[(int64_t)k * ldq + r];
    q[(int64_t)k * ldq + r] = 0.0;
    if (qhash) qhash[r] = QHASH_NONE;  // (the row no longer is what its fingerprint says)
  }
}

hipError_t launch_aug_from_sub(double* q, int64_t ldq, int k, int K, const int64_t* idx, int64_t M,
                               const int64_t* starts, const int64_t* goff_sub, int J, const double* qsub1,
                               hipStream_t stream, int64_t* qhash) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(aug_from_sub_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, q, ldq, k, K, idx, M,
                     starts, goff_sub, J, qsub1, qhash);
  return hipGetLastError();
}


// ===========================================================================
// synthetic mixture (bench workload)
// ===========================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u01(uint32_t a, uint32_t b) {
  // 53-bit uniform in (0,1)
  const uint64_t v = (((uint64_t)a << 32) | b) >> 11;
  return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// 16 rows (one row-group => one group of observations) per 256-thread block; eps staged in LDS.
// Philox counter = (group id << 40) + row inside the group (+ row_offset), so any shard of any group can
// be regenerated independently.  Labels: uniform, or by inverse CDF of the group's mixing proportions.
__global__ void __launch_bounds__(256) synth_kernel(SynthLaunch a) {
  extern __shared__ double eps[];  // [16][DP]
  __shared__ int zlab[16];
  const int DP = a.DP, D = a.D, K = a.K;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  int grp = 0, nvalid;
  if (a.rginfo) {
    const int info = a.rginfo[blockIdx.x];
    grp = info >> 5;
    nvalid = info & 31;
  } else {
    const int64_t rem = a.nrows - row0;
    nvalid = rem >= 16 ? 16 : (rem > 0 ? (int)rem : 0);
  }
  const int64_t ingrp0 = row0 - (a.goff ? a.goff[grp] : 0);  // row inside its group
  const uint64_t gid = a.gids ? (uint64_t)a.gids[grp] : (uint64_t)(a.group_base + grp);
  const uint64_t gbase = (gid << 40) + (uint64_t)(a.row_offset + ingrp0);
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const int npair = (D + 1) / 2;
  for (int t = threadIdx.x; t < 16 * npair; t += 256) {
    const int r = t / npair, p = t % npair;
    const uint64_t g = gbase + (uint64_t)r;
    uint32_t o[4];
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)p, 1u, k0, k1, o);
    const double u1 = u01(o[0], o[1]), u2 = u01(o[2], o[3]);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925 * u2, &sn, &cs);
    eps[r * DP + 2 * p] = rad * cs;
    if (2 * p + 1 < D) eps[r * DP + 2 * p + 1] = rad * sn;
  }
  if (threadIdx.x < 16) {
    const uint64_t g = gbase + (uint64_t)threadIdx.x;
    uint32_t o[4];
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 0u, 2u, k0, k1, o);
    int z;
    if (a.cdf) {
      const double u = u01(o[0], o[1]);
      const double* c = a.cdf + (int64_t)grp * K;
      z = K - 1;
      for (int k = 0; k < K - 1; ++k)
        if (u < c[k]) {
          z = k;
          break;
        }
    } else {
      z = (int)(o[0] % (uint32_t)K);
    }
    zlab[threadIdx.x] = z;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 16 * DP; t += 256) {
    const int r = t / DP, i = t % DP;
    const int64_t row = row0 + r;
    if (row >= a.NP) continue;
    double v = 0.0;
    if (r < nvalid && i < D) {
      const int z = zlab[r];
      const double* Lz = a.L + ((int64_t)z * D + i) * D;
      v = a.mu[(int64_t)z * D + i];
      for (int j = 0; j <= i; ++j) v += Lz[j] * eps[r * DP + j];
    }
    a.X[row * DP + i] = v;
  }
  if (a.qZ) {
    for (int t = threadIdx.x; t < 16 * K; t += 256) {
      const int k = t / 16, r = t % 16;
      const int64_t row = row0 + r;
      if (row >= a.NP) continue;
      double q = 0.0;
      if (r < nvalid) q = K == 1 ? 1.0 : (k == zlab[r] ? a.hard : (1.0 - a.hard) / (K - 1));
      a.qZ[(int64_t)k * a.ldq + row] = q;
    }
  }
}

hipError_t launch_synth(const SynthLaunch& a, hipStream_t stream) {
  if (a.NP <= 0) return hipSuccess;
  const size_t shmem = (size_t)16 * a.DP * sizeof(double);
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((a.NP + 15) / 16)), dim3(256), shmem, stream, a);
  return hipGetLastError();
}


// qT[row * K + k] = qZ[k * ldq + row]: the device-side half of handing responsibilities back in the caller's row-major
// layout.  One block = 64 rows x up to 64 columns through LDS: 512-byte column reads, row segments of up to 512 bytes
// written (one contiguous 64 * K block when K <= 64).
__global__ void __launch_bounds__(256) transpose_qz_kernel(const double* __restrict__ qZ, int64_t ldq, int K,
                                                           int64_t NP, double* __restrict__ qT) {
  __shared__ double tile[64][65];
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int k0 = blockIdx.y * 64, kc = K - k0 < 64 ? K - k0 : 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = w; k < kc; k += 4) {
    const int64_t row = row0 + lane;
    tile[lane][k] = row < NP ? qZ[(int64_t)(k0 + k) * ldq + row] : 0.0;
  }
  __syncthreads();
  const int64_t left = NP - row0;
  const int n = (int)(left < 64 ? left : 64) * kc;
  for (int t = threadIdx.x; t < n; t += 256) {
    const int r = t / kc, c = t - r * kc;
    qT[(row0 + r) * K + k0 + c] = tile[r][c];
  }
}
hipError_t launch_transpose_qz(const double* qZ, int64_t ldq, int K, int64_t NP, double* qT, hipStream_t stream) {
  if (NP <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(transpose_qz_kernel, dim3((unsigned)((NP + 63) / 64), (unsigned)((K + 63) / 64)), dim3(256), 0,
                     stream, qZ, ldq, K, NP, qT);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Model selection: the E-step from cached distances (Context::estep_cache).
// A cluster whose posterior has not changed in any bit since its column of -0.5 d^2_k(x_n) was computed keeps that
// column (estep_kernel in raw mode recomputes the others).  This kernel adds the constants c_jk, which DO change with
// the weights, and normalises: logsumexp in the reference's operation order (probutils.cpp:141-150), q = exp(x - logZ)
// (cluster.cpp:130-131), F_z and (optionally) LL_k partials.  One lane = one row; its K values stay in registers
// (KT >= K, statically unrolled: all loads of a row are in flight together, no LDS, full occupancy) -- the pass is a
// pure stream of 8 K bytes per row in and, where something changed, out.
// ---------------------------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(256) softmax_cached_kernel(CachedNormLaunch a) {
  __shared__ double fzw[4];
  __shared__ double llw[4 * KT];
  __shared__ double etab[64];  // 2^(j / 64) for exp_nonpos
  __shared__ unsigned long long cmask[2];
  const int tid = threadIdx.x, K = a.K;
  fill_exp_table(etab, tid, 256);
  if (tid < 2) cmask[tid] = 0ull;
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 256 + tid;
  const bool inb = row < a.NP;
  const int64_t rr = inb ? row : 0;  // (out-of-range lanes load row 0 and write nothing)
  int grp = 0;
  bool ok = false;
  if (inb) {
    if (a.rginfo) {
      const int info = a.rginfo[row >> 4];
      grp = info >> 5;
      ok = (int)(row & 15) < (info & 31);
    } else {
      ok = row < a.nrows;
    }
  }
  const double* crow = a.ctab + (int64_t)grp * K;
  double v[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    v[j] = -INFINITY;
    if (j < K) {
      const int cm = a.colmap ? a.colmap[j] : j;
      const double* col = cm >= 0 ? a.dcache + (int64_t)cm * a.ldc : a.fresh + (int64_t)(-cm - 1) * a.ldf;
      v[j] = col[rr];
    }
  }
  double mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < KT; ++j)
    if (j < K) {
      v[j] += crow[j];
      mx = fmax(mx, v[j]);
    }
  if (a.rmax && inb) {  // the row's largest log q~ and its cluster (BoundSelectLaunch)
    int am = 0;
#pragma unroll
    for (int j = 0; j < KT; ++j)
      if (j < K) am = v[j] == mx ? j : am;
    a.rmax[row] = mx;
    a.ramax[row] = am;
  }
  // (a.dq, a.ll_part: launch-uniform)
  // Without LL_k: ONE exponential per entry -- e = exp(x - max) replaces x in its register and q = e / sum(e) (the same
  // sum and logZ; q within 2 ulp of exp(x - logZ), far inside the tolerance of the moved-row test below, and the same
  // bits for the same inputs, which is what "unchanged rows are not written" relies on).  With LL_k the log value is
  // still needed behind the sum, so the second exponential stays.
  const bool onexp = a.ll_part == nullptr;
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < KT; ++j)
    if (j < K) {
      const double e = exp_nonpos(v[j] - mx, etab);
      s += e;
      if (onexp) v[j] = e;
    }
  const double logZ = log(s) + mx;
  const double inv = rcp_pos(s);
#pragma unroll
  for (int j = 0; j < KT; ++j)
    if (j < K) {
      const double lq = v[j];
      double q = onexp ? lq * inv : exp_nonpos(lq - logZ, etab);
      // In the moved-row sweeps (a.dq: cluster()'s split search with the distance cache; never a plain E-step, which is
      // cluster.cpp:130-131 to the letter like the other E-step kernels): responsibilities below 2^-300 are stored as
      // zero -- a deliberate deviation from the reference, DESIGN 4.4.  A row that belongs to one cluster has q = 1 there and
      // e^-(hundreds) everywhere else; those specks change in their last bits with every change of any weight, so that
      // every row "changed" in every sweep of a split candidate (all K old values read, all K new ones written: 3 x 2.6
      // GB per sweep at K = 33) although nothing above 1e-90 moved.  Flushed, such a row comes out bit for bit as it
      // was, its fingerprint matches and the sweep touches nothing of it.  The mass dropped from any statistic is below
      // 1e-90 of the row's; the reference itself loses everything below e^-745 (exp underflow, probutils.cpp:146).
      if (!ok || (a.dq && q < 0x1p-300)) q = 0.0;
      v[j] = q;
      if (a.ll_part) {  // the data term of the split ordering (cluster.cpp:407-410), as estep_kernel's sweep forms it
        const double ll = wave_sum(q > 0.0 ? q * (lq - crow[j]) : 0.0);
        if ((tid & 63) == 0) llw[(tid >> 6) * KT + j] = ll;
      }
    }
  if (a.dq) {
    // Also report the move away from the responsibilities being overwritten -- and write only what changed: a row
    // whose K values all come out bit for bit as they were is not written at all, its q_new - q_old only when some
    // |.| exceeds dq_tol (delta_suffstat never looks at the other rows).  Between the candidates of a split round
    // almost every row is of the first kind.
    // Fingerprint of the row's new values (qhash_step, lc_device.hpp).  Equal to the stored fingerprint of the old
    // values: the row is unchanged (up to a 2^-64 coincidence) and its old values are not read at all.
    int64_t h = 0;
    bool same = false;
    if (a.qhash) {
      uint64_t acc = QHASH_SEED;
#pragma unroll
      for (int j = 0; j < KT; ++j)
        if (j < K) acc = qhash_step(acc, v[j], j);
      h = qhash_finish(acc);
      if (a.qhash_in && inb) same = a.qhash[row] == h;
    }
    // A row that did change: its old values pass through eight registers at a time; every entry that differs is
    // written at once (with the specks flushed, a row of the cluster being split changes in two or three columns, not in
    // all K: 33 partial-line writes per such row were a third of a candidate's sweep), and the row's differences go to
    // the row-major table in the same pass (delta_suffstat reads them only where amax > dq_tol).
    double am = 0.0;
    unsigned long long mlo = 0ull, mhi = 0ull;  // columns in which this row moved at all (a.colmask)
    if (!same && inb) {
      // (all K old values in ONE batch of loads: nearly every wave holds a changed row or two, and with eight registers at
      //  a time it walked five dependent memory round trips for them -- 1.83 against 1.23 ms for a candidate's sweep at
      //  K = 33, tools/smc_probe.hip; the 2 KT extra registers cost the unchanged-rows stream 8 %)
      constexpr int OB = KT;
#pragma unroll
      for (int jb = 0; jb < KT; jb += OB) {
        double o[OB];
#pragma unroll
        for (int u = 0; u < OB; ++u)
          if (jb + u < KT && jb + u < K) o[u] = a.qZ[(int64_t)(jb + u) * a.ldq + row];
#pragma unroll
        for (int u = 0; u < OB; ++u)
          if (jb + u < KT && jb + u < K) {
            const double dd = v[jb + u] - o[u];
            am = fmax(am, fabs(dd));
            if (dd != 0.0) {
              a.qZ[(int64_t)(jb + u) * a.ldq + row] = v[jb + u];
              if (jb + u < 64) mlo |= 1ull << ((jb + u) & 63);
              else mhi |= 1ull << ((jb + u) & 63);
            }
            a.dq[row * a.ldd + jb + u] = dd;
          }
      }
      if (a.colmask) {
        if (mlo) atomicOr(&cmask[0], mlo);
        if (mhi) atomicOr(&cmask[1], mhi);
      }
    }
    if (inb) {
      a.amax[row] = am;
      if (a.qhash && !same) a.qhash[row] = h;
    }
  } else if (inb) {
#pragma unroll
    for (int j = 0; j < KT; ++j)
      if (j < K) a.qZ[(int64_t)j * a.ldq + row] = v[j];
  }
  const double fz = wave_sum(ok ? logZ : 0.0);
  if ((tid & 63) == 0) fzw[tid >> 6] = fz;
  __syncthreads();
  if (tid == 0) a.fz_part[blockIdx.x] = -(fzw[0] + fzw[1] + fzw[2] + fzw[3]);  // cluster.cpp:137 returns -sum(logZ)
  if (a.ll_part)
    for (int j = tid; j < K; j += 256)
      a.ll_part[(int64_t)blockIdx.x * K + j] = llw[j] + llw[KT + j] + llw[2 * KT + j] + llw[3 * KT + j];
  if (a.colmask && tid < 2) {  // (an OR: the order of the blocks does not matter; most blocks find their bits set already)
    const unsigned long long v = cmask[tid];
    if (v && (__hip_atomic_load(&a.colmask[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & v) != v) atomicOr(&a.colmask[tid], v);
  }
}
// Tests (libcluster_hip_testhooks.so, LC_TEST_VERIFY_QHASH): does every stored fingerprint describe the row it belongs
// to?  Counts the rows whose fingerprint is neither QHASH_NONE nor that of the K values in the buffer.
__global__ void __launch_bounds__(256) qhash_verify_kernel(const double* __restrict__ qZ, int64_t ldq, int K, int64_t NP,
                                                           const int64_t* __restrict__ qhash, unsigned long long* bad) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= NP) return;
  const int64_t stored = qhash[row];
  if (stored == QHASH_NONE) return;
  uint64_t acc = QHASH_SEED;
  for (int j = 0; j < K; ++j) acc = qhash_step(acc, qZ[(int64_t)j * ldq + row], j);
  if (qhash_finish(acc) != stored) atomicAdd(bad, 1ull);
}
hipError_t launch_qhash_verify(const double* qZ, int64_t ldq, int K, int64_t NP, const int64_t* qhash,
                               unsigned long long* bad, hipStream_t stream) {
  if (NP <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(qhash_verify_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, stream, qZ, ldq, K, NP, qhash, bad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Model selection: bring a split trial's working copy of the responsibilities back in line with the original, row by row
// (Context::qz_clone_to_alt, round 6).  After a rejected candidate the copy differs from the original in the candidate's own
// rows only; both buffers carry row fingerprints (softmax_cached_kernel), so a row is copied -- its K values, its
// fingerprint, zeros in the copy's columns K ... Kdst-1 -- exactly when the fingerprints differ or either is unknown.  Equal
// fingerprints = equal rows is the premise the moved-row sweeps already stand on (qhash_step; a zero entry leaves a
// fingerprint as it is, so a row that is unchanged has zeros in the copy's extra column).  Lanes = consecutive rows of
// column-major buffers: the copies coalesce whenever neighbouring rows differ together.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) qz_resync_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t ldq, int K,
                                                        int Kdst, int64_t NP, int64_t* __restrict__ dhash,
                                                        const int64_t* __restrict__ shash) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= NP) return;
  const int64_t hs = shash[row], hd = dhash[row];
  if (hs == hd && hs != QHASH_NONE) return;
  for (int j = 0; j < K; ++j) dst[(int64_t)j * ldq + row] = src[(int64_t)j * ldq + row];
  for (int j = K; j < Kdst; ++j) dst[(int64_t)j * ldq + row] = 0.0;
  dhash[row] = hs;
}
hipError_t launch_qz_resync(double* dst, const double* src, int64_t ldq, int K, int Kdst, int64_t NP, int64_t* dhash,
                            const int64_t* shash, hipStream_t stream) {
  if (NP <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(qz_resync_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, stream, dst, src, ldq, K, Kdst, NP, dhash,
                     shash);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Model selection: the rows a recomputed column matters for (BoundSelectLaunch, lc_kernels.h)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bound_select_kernel(BoundSelectLaunch a) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= a.NP) return;
  const int j = a.ramax[row];
  bool skip = j >= 0 && j < a.K && a.usable[j] != 0;
  double d2[BOUND_MAX_COLS];
#pragma unroll
  for (int t = 0; t < BOUND_MAX_COLS; ++t)
    if (t < a.ncol) d2[t] = -2.0 * a.ref[t][row];  // (every reference is read before any destination is written)
  double ub[BOUND_MAX_COLS];  // -0.5 s^2: an upper bound of the new column's raw value -0.5 d_new^2
  if (skip) {
    const double low = a.rmax[row] + a.dcj[j] - a.T;  // the row's largest log q~ among the unchanged columns, less the margin
#pragma unroll
    for (int t = 0; t < BOUND_MAX_COLS; ++t)
      if (t < a.ncol) {
        const double s = fmax(a.sigma[t] * sqrt(fmax(d2[t], 0.0)) - a.bnorm[t], 0.0);
        ub[t] = -0.5 * s * s;
        skip = skip && (a.cnew[t] + ub[t] < low);  // (NaN anywhere: false -> the row is recomputed)
      }
  }
  a.need[row] = skip ? 0.0 : 1.0;
  if (skip) {
    // A skipped row keeps the BOUND, not -inf: c + ub < max - T, so the sweep's exponential still comes out as
    // exactly 0.0 (or is flushed, moved-row mode) -- the same bits -- and the column stays a sound reference for the next
    // bounded pass: d_ref >= s holds for the stored value, where -inf would lock the row out however far the cluster moves.
#pragma unroll
    for (int t = 0; t < BOUND_MAX_COLS; ++t)
      if (t < a.ncol) a.dest[t][row] = ub[t];
  }
}
hipError_t launch_bound_select(const BoundSelectLaunch& a, hipStream_t stream) {
  if (a.NP <= 0 || a.ncol <= 0) return hipSuccess;
  if (a.ncol > BOUND_MAX_COLS || a.K > BOUND_MAX_K) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bound_select_kernel, dim3((unsigned)((a.NP + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) gather_rows_plain_kernel(const double* X, int DP, const int64_t* idx, int64_t M, double* Xdst) {
  const int per = DP / 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * per) return;
  const int64_t p = t / per;
  const int c2 = (int)(t % per);
  reinterpret_cast<double2*>(Xdst + p * DP)[c2] = reinterpret_cast<const double2*>(X + idx[p] * DP)[c2];
}
hipError_t launch_gather_rows_plain(const double* X, int DP, const int64_t* idx, int64_t M, double* Xdst, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  const int64_t n = M * (DP / 2);
  hipLaunchKernelGGL(gather_rows_plain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, X, DP, idx, M, Xdst);
  return hipGetLastError();
}
struct ScatterDest {
  double* p[BOUND_MAX_COLS];
};
__global__ void __launch_bounds__(256) scatter_cols_kernel(const double* src, int64_t lds, int ncol, ScatterDest d, const int64_t* idx,
                                                           int64_t M) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t r = idx[p];
#pragma unroll
  for (int t = 0; t < BOUND_MAX_COLS; ++t)
    if (t < ncol) d.p[t][r] = src[(int64_t)t * lds + p];
}
hipError_t launch_scatter_cols(const double* src, int64_t lds, int ncol, double* const* dest_host, const int64_t* idx, int64_t M,
                               hipStream_t stream) {
  if (M <= 0 || ncol <= 0) return hipSuccess;
  if (ncol > BOUND_MAX_COLS) return hipErrorInvalidValue;
  ScatterDest d{};
  for (int t = 0; t < ncol; ++t) d.p[t] = dest_host[t];
  hipLaunchKernelGGL(scatter_cols_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, src, lds, ncol, d, idx, M);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The sum over ranks in rank order (LIBCLUSTER_COMM=rccl-gather, lc_comm.cpp): one thread per element, W strided reads.
// __fadd_rn-style plain additions, never contracted: the same bits as the host transport's loop.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rank_order_sum_kernel(const double* __restrict__ g, int world, int64_t count,
                                                             double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = g[i];
  for (int r = 1; r < world; ++r) s = __dadd_rn(s, g[(int64_t)r * count + i]);
  out[i] = s;
}
hipError_t launch_rank_order_sum(const double* gathered, int world, int64_t count, double* out, hipStream_t stream) {
  if (count <= 0 || world < 1) return hipSuccess;
  hipLaunchKernelGGL(rank_order_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, gathered, world,
                     count, out);
  return hipGetLastError();
}

int64_t softmax_cached_grid(int64_t NP) { return (NP + 255) / 256; }
int softmax_cached_max_k() { return 72; }
hipError_t launch_softmax_cached(const CachedNormLaunch& a, hipStream_t stream) {
  const int64_t grid = softmax_cached_grid(a.NP);
  if (grid <= 0 || a.K <= 0) return hipSuccess;
  if (a.K > softmax_cached_max_k()) return hipErrorInvalidValue;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, stream, a);
    return hipGetLastError();
  };
  if (a.K <= 8) return go(softmax_cached_kernel<8>);
  if (a.K <= 16) return go(softmax_cached_kernel<16>);
  if (a.K <= 24) return go(softmax_cached_kernel<24>);
  if (a.K <= 32) return go(softmax_cached_kernel<32>);
  if (a.K <= 36) return go(softmax_cached_kernel<36>);  // (a round's candidates run at K + 1: 33 columns, not 40)
  if (a.K <= 40) return go(softmax_cached_kernel<40>);
  if (a.K <= 56) return go(softmax_cached_kernel<56>);
  return go(softmax_cached_kernel<72>);
}

}  // namespace lck
