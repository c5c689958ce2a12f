// gfx950 (MI355X / CDNA4) kernels for the libcluster variational E-step and
// sufficient-statistic accumulation.  fp64 throughout.
//
// What they replace in the reference (dsteinberg/libcluster):
//   estep_kernel     vbexpectation            src/cluster.cpp:91-138
//                    -> GaussWish::Eloglike   src/distributions.cpp:356-370
//                    -> probutils::mahaldist  src/probutils.cpp:113-138
//                    -> probutils::logsumexp  src/probutils.cpp:141-150
//   suffstat_kernel  updateSS                 src/cluster.cpp:53-82
//                    -> GaussWish::addobs     src/distributions.cpp:301-313
//
// Matrix instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks,
// 512 flop, 16 cycles/SIMD = 32 flop/clk/SIMD; measured 73-74 TFLOP/s on
// MI355X, vs 47-49 for v_mfma_f64_16x16x4_f64 -- profiles/r01_mfma_f64_probe.log).
// Lane layout (probed, tools/mfma_f64_4x4_layout.hip), lane = lo2 + 4*blk + 16*hi:
//   A[i][k] of block blk : i = lo2, k = hi
//   B[k][j] of block blk : j = lo2, k = hi
//   D[i][j] of block blk : j = lo2, i = hi
#include "lc_kernels.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace lck {

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// sum over the 4 lanes {l, l^16, l^32, l^48}
__device__ __forceinline__ double sum_over_hi(double v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// sum over the 16 lanes of a DPP row (same hi)
__device__ __forceinline__ double sum_over_lo4(double v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  v = sum_over_lo4(v);
  return sum_over_hi(v);
}

// ===========================================================================
// E-step
// ===========================================================================
// One wave owns R row-groups of 16 rows and keeps their X fragments in
// registers for the whole cluster loop (X is read from HBM exactly once).
// Per cluster k the host supplies A_k = sqrt(nu_k) * chol(iW_k)^-1 (lower
// triangular), b_k = A_k m_k and c_jk, so that
//     log q~[n,k] = c_jk - 0.5 * || A_k x_n - b_k ||^2
// which equals E_logZ(k) + Eloglike_k(x_n) of cluster.cpp:120-121.
// A_k arrives as 4x4 tiles in consumption order (only tiles on or below the
// diagonal), staged through LDS with register double-buffering; one tile read
// feeds R MFMAs.  MFMA block b <-> rows 4b..4b+3 of the row-group, so
//   A operand = tile A_k[4it+lo2][4jt+hi]   (same for all 4 blocks)
//   B operand = x[row = lane&15][4jt + hi]
//   D         = y[4it+hi][row = lane&15]
// log q~ is written to the qZ buffer as scratch, then normalised in place in
// the same arithmetic order as the reference: max, sum exp(x-max), log+max,
// exp(x - logZ).
// n-th read of a cluster's parameter stream (for it: -b[it], tile(it,0), ..., tile(it,it)):
// .jt < 0: element of the -b vector (offset in doubles from Pb), else of tile (it,jt) (from Pt)
struct RdInfo {
  int it, jt, off;
};
__host__ __device__ constexpr RdInfo rd_info(int n) {
  int it = 0;
  while (n >= it + 2) {
    n -= it + 2;
    ++it;
  }
  return n == 0 ? RdInfo{it, -1, 4 * it} : RdInfo{it, n - 1, (it * (it + 1) / 2 + (n - 1)) * 16};
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int DP, int R, int WAVES, bool SPARSE>
__global__ void __launch_bounds__(WAVES * 64, 2) estep_kernel(EstepLaunch a) {
  constexpr int NT = DP / 4;
  constexpr int NTILES = NT * (NT + 1) / 2;
  constexpr int NREAD = NTILES + NT;  // LDS reads per cluster
  constexpr int PF = 6;               // reads in flight ahead of their use
  constexpr int PS = NTILES * 16 + DP;
  constexpr int NTHR = WAVES * 64;
  constexpr int NV2 = PS / 2;
  constexpr int NPRE = (NV2 + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* pbuf = lds;              // [2][PS]
  double* llw = lds + 2 * PS;      // [WAVES][K]
  double* fzw = llw + WAVES * a.K; // [WAVES]
  int* klist = reinterpret_cast<int*>(fzw + WAVES);  // sparse mode: [K] active clusters of this block, [K] flags,
  int* kflag = klist + a.K;                          // [WAVES*R] groups of the block's row groups, [1] count
  int* bgrp = kflag + a.K;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4;
  const int K = a.K;
  const int64_t rg0 = ((int64_t)blockIdx.x * WAVES + wave) * R;

  double xf[R][NT];
  int grp[R];
  bool rowok[R], rgok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t rg = rg0 + r;
    rgok[r] = rg < a.nrg;
    int info = 0;
    if (rgok[r]) {
      if (a.rginfo) {
        info = a.rginfo[rg];
      } else {
        const int64_t rem = a.nrows - rg * RG;
        info = rem >= RG ? RG : (rem > 0 ? (int)rem : 0);
      }
    }
    grp[r] = info >> 5;
    rowok[r] = lo4 < (info & 31);
    const double* xr = a.X + ((rgok[r] ? rg : 0) * RG + lo4) * DP + hi;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) xf[r][jt] = xr[4 * jt];
  }

  // register double-buffer for the next cluster's parameter record
  double pre[NPRE][2];
#define LC_GLOAD(kk)                                                                          \
  {                                                                                           \
    const double2* src_ = reinterpret_cast<const double2*>(a.params + (int64_t)(kk) * PS);   \
    _Pragma("unroll") for (int i_ = 0; i_ < NPRE; ++i_) {                                     \
      const int idx_ = tid + i_ * NTHR;                                                       \
      const double2 v_ = src_[idx_ < NV2 ? idx_ : NV2 - 1];                                   \
      pre[i_][0] = v_.x;                                                                      \
      pre[i_][1] = v_.y;                                                                      \
    }                                                                                         \
  }
#define LC_LSTORE(bb)                                                                         \
  {                                                                                           \
    double2* dst_ = reinterpret_cast<double2*>(pbuf + (bb) * PS);                             \
    _Pragma("unroll") for (int i_ = 0; i_ < NPRE; ++i_) {                                     \
      const int idx_ = tid + i_ * NTHR;                                                       \
      if (idx_ < NV2) dst_[idx_] = make_double2(pre[i_][0], pre[i_][1]);                      \
    }                                                                                         \
  }

  // Sparse mode (cluster.cpp:109-112, 134-135): the block walks only the clusters that are active (c_jk > -inf)
  // for at least one of its row groups -- no parameter staging, barrier or MFMA for the others; their columns
  // get log q~ = -inf directly.  Waves whose own row groups are all inactive for a listed cluster skip it too.
  int nact = K;
  if constexpr (SPARSE) {
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) bgrp[wave * R + r] = rgok[r] ? grp[r] : -1;
    }
    __syncthreads();
    for (int k = tid; k < K; k += NTHR) {
      int f = 0;
      for (int g = 0; g < WAVES * R; ++g)
        if (bgrp[g] >= 0 && a.ctab[(int64_t)bgrp[g] * K + k] != -INFINITY) f = 1;
      kflag[k] = f;
    }
    __syncthreads();
    if (tid == 0) {
      int n = 0;
      for (int k = 0; k < K; ++k)
        if (kflag[k]) klist[n++] = k;
      bgrp[WAVES * R] = n;
    }
    __syncthreads();
    nact = bgrp[WAVES * R];
    for (int k = 0; k < K; ++k) {
      if (kflag[k]) continue;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (rgok[r] && hi == (k & 3)) a.qZ[(int64_t)k * a.ldq + (rg0 + r) * RG + lo4] = -INFINITY;
    }
  }

  if (nact > 0) {
    LC_GLOAD(SPARSE ? klist[0] : 0);
    LC_LSTORE(0);
  }
  __syncthreads();

  double mx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mx[r] = -INFINITY;

  for (int ii = 0; ii < nact; ++ii) {
    const int k = SPARSE ? klist[ii] : ii;
    const int buf = ii & 1;
    if (ii + 1 < nact) LC_GLOAD(SPARSE ? klist[ii + 1] : ii + 1);
    const double* P = pbuf + buf * PS;
    const double* Pt = P + (lane & 3) + 4 * hi;  // this lane's element of every 4x4 tile
    const double* Pb = P + NTILES * 16 + hi;     // this lane's element of every 4-vector of -b
    // The cluster's parameters are consumed as one linear stream of LDS reads
    // (for it: -b[it], tile(it,0..it)), software-pipelined PF reads ahead so no
    // MFMA ever waits on the read issued just before it.
    double ring[PF];
    static_for<PF>([&](auto ic) {
      constexpr RdInfo ri = rd_info(ic);
      ring[ic] = ri.jt < 0 ? Pb[ri.off] : Pt[ri.off];
    });
    double d2[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) d2[r] = 0.0;
    // sparse mode (cluster.cpp:109-112): a cluster that is inactive (c_jk = -inf) for the groups of ALL of this
    // wave's row groups needs no Mahalanobis term -- its log q~ is -inf whatever the distance
    bool wave_active = true;
    if constexpr (SPARSE) {
      wave_active = false;
#pragma unroll
      for (int r = 0; r < R; ++r)
        wave_active = wave_active || (rgok[r] && a.ctab[(int64_t)grp[r] * K + k] != -INFINITY);
    }
    if (wave_active)
    static_for<NREAD>([&](auto nc) {
      constexpr int n = nc;
      constexpr RdInfo ri = rd_info(n);
      const double v = ring[n % PF];
      if constexpr (n + PF < NREAD) {
        constexpr RdInfo rn = rd_info(n + PF);
        ring[n % PF] = rn.jt < 0 ? Pb[rn.off] : Pt[rn.off];
      }
      if constexpr (ri.jt < 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = v;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mfma4(v, xf[r][ri.jt], acc[r]);
        if constexpr (ri.jt == ri.it) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            d2[r] = fma(acc[r], acc[r], d2[r]);
            // pin the running sum here: otherwise LLVM sinks the whole fma chain below the
            // MFMA stream and keeps every row's accumulator alive (96 extra VGPRs, spills)
            asm volatile("" : "+v"(d2[r]));
          }
        }
      }
    });
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // sum over the four hi lanes on the matrix pipe: D[i][j] = sum_k 1 * B[k][j] leaves the
      // total in every lane (B[k=hi][j=row] is exactly where the partial sums live)
      const double dd = mfma4(1.0, d2[r], 0.0);
      const double lq = a.ctab[(int64_t)grp[r] * K + k] - 0.5 * dd;
      mx[r] = fmax(mx[r], lq);
      if (rgok[r] && hi == (k & 3)) a.qZ[(int64_t)k * a.ldq + (rg0 + r) * RG + lo4] = lq;
    }
    if (ii + 1 < nact) LC_LSTORE(buf ^ 1);
    __syncthreads();
  }

#undef LC_GLOAD
#undef LC_LSTORE

  if (a.raw) return;  // GaussWish::Eloglike: leave c_k - 0.5 d^2 in qZ, no normalisation

  // ---- normalise (probutils.cpp:141-150, cluster.cpp:124-131) -------------
  double logZ[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double s = 0.0;
    if (rgok[r]) {
      const double* qp = a.qZ + (rg0 + r) * RG + lo4;
#pragma unroll 8
      for (int k = hi; k < K; k += 4) s += exp(qp[(int64_t)k * a.ldq] - mx[r]);
    }
    s = sum_over_hi(s);
    logZ[r] = log(s) + mx[r];
  }
  for (int kb = 0; kb < K; kb += 4) {
    const int k = kb + hi;
    double ll = 0.0;
    if (k < K) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (rgok[r]) {
          double* qp = a.qZ + (int64_t)k * a.ldq + (rg0 + r) * RG + lo4;
          const double lq = *qp;
          double q = exp(lq - logZ[r]);
          if (!rowok[r]) q = 0.0;
          *qp = q;
          if (a.ll_part && q > 0.0) ll += q * (lq - a.ctab[(int64_t)grp[r] * K + k]);
        }
      }
    }
    if (a.ll_part) {  // wave-uniform
      ll = sum_over_lo4(ll);
      if (k < K && lo4 == 0) llw[wave * K + k] = ll;
    }
  }
  double fz = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (rgok[r] && rowok[r] && hi == 0) fz += logZ[r];
  fz = wave_sum(fz);
  if (lane == 0) fzw[wave] = fz;
  __syncthreads();
  if (a.ll_part)
    for (int k = tid; k < K; k += NTHR) {
      double s = 0.0;
      for (int w = 0; w < WAVES; ++w) s += llw[w * K + k];
      a.ll_part[(int64_t)blockIdx.x * K + k] = s;
    }
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < WAVES; ++w) s += fzw[w];
    a.fz_part[blockIdx.x] = -s;  // cluster.cpp:137 returns -sum(logZ)
  }
}

template <int DP>
struct EstepCfg;
template <>
struct EstepCfg<16> { static constexpr int R = 4, WAVES = 4; };
template <>
struct EstepCfg<32> { static constexpr int R = 4, WAVES = 4; };
template <>
struct EstepCfg<64> { static constexpr int R = 4, WAVES = 4; };
template <>
struct EstepCfg<128> { static constexpr int R = 2, WAVES = 8; };

template <int DP>
static int rows_per_block_t() { return EstepCfg<DP>::R * EstepCfg<DP>::WAVES * RG; }

int estep_rows_per_block(int DP) {
  switch (DP) {
    case 16: return rows_per_block_t<16>();
    case 32: return rows_per_block_t<32>();
    case 64: return rows_per_block_t<64>();
    case 128: return rows_per_block_t<128>();
  }
  return -1;
}

int64_t estep_grid(int DP, int64_t nrg) {
  const int64_t rgpb = estep_rows_per_block(DP) / RG;
  return (nrg + rgpb - 1) / rgpb;
}

template <int DP, bool SPARSE>
static hipError_t launch_estep_s(const EstepLaunch& a, hipStream_t stream) {
  constexpr int R = EstepCfg<DP>::R, WAVES = EstepCfg<DP>::WAVES;
  const size_t shmem = (size_t)(2 * pstride(DP) + WAVES * a.K + WAVES) * sizeof(double) +
                       (size_t)(2 * a.K + WAVES * R + 2) * sizeof(int);
  auto kern = estep_kernel<DP, R, WAVES, SPARSE>;
  static size_t attr_set = 0;  // largest dynamic-LDS size already granted
  if (shmem > 64 * 1024 && shmem > attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    attr_set = shmem;
  }
  const int64_t grid = estep_grid(DP, a.nrg);
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), shmem, stream, a);
  return hipGetLastError();
}

template <int DP>
static hipError_t launch_estep_t(const EstepLaunch& a, hipStream_t stream) {
  return a.sparse ? launch_estep_s<DP, true>(a, stream) : launch_estep_s<DP, false>(a, stream);
}

hipError_t launch_estep(const EstepLaunch& a, hipStream_t stream) {
  switch (a.DP) {
    case 16: return launch_estep_t<16>(a, stream);
    case 32: return launch_estep_t<32>(a, stream);
    case 64: return launch_estep_t<64>(a, stream);
    case 128: return launch_estep_t<128>(a, stream);
  }
  return hipErrorInvalidValue;
}

// ===========================================================================
// Sufficient statistics
// ===========================================================================
// S_k = sum_n q_nk x_n x_n^T is a GEMM whose reduction dimension is the data
// rows.  One wave owns CPW clusters and streams over a chunk of rows, four
// rows per step, with the (symmetric, lower-triangular 16x16-blocked)
// accumulators in registers for the whole chunk.  For every needed pair of
// 16-wide feature blocks (JBp >= JB) and every rotation s of the four 4-wide
// sub-blocks inside JBp the wave issues
//     D += A(x[., 16*JBp + 4*((blk+s)&3) + lo2]) * B(q_k * x[., 16*JB + 4*blk + lo2])
// so that MFMA block blk computes the 4x4 tile (i-tile (blk+s)&3, j-tile blk).
// Off-diagonal 16x16 blocks need s=0..3, diagonal ones s=0..2 (symmetry).
//
// fp64 VALU and fp64 MFMA share the issue pipe on gfx950 (a VALU block between
// MFMA streams is not hidden by the other resident wave:
// tools/mfma_issue_probe.hip, 97% -> 81% of peak), so the rotated operands are
// NOT formed with DPP moves: the workgroup stages BR rows of X in LDS (every
// wave of the group needs the same rows) and each wave reads all four
// rotations of a fragment straight from LDS with rotated addresses (LGKM
// path).  Row stride DP+16 doubles keeps the two rows a ds_read_b64 half-wave
// touches on disjoint banks.  q columns are staged per wave the same way.
// What remains on the VALU per cluster and step: NB multiplies (q*x), NB adds
// (s_k) and one add (N_k) against NACC MFMAs.
// Each (chunk, cluster) writes one partial record; launch_reduce_partials sums
// chunks in fixed order.
template <int NB>
struct SSAcc { static constexpr int N = NB * 3 + NB * (NB - 1) / 2 * 4; };

constexpr int SS_BR = 32;  // rows staged per batch

// SKIP: a (4-row step, cluster) pair whose four responsibilities are all exactly 0.0 contributes exactly nothing;
// the sparse mode (cluster.cpp:67-79: groups without mass in a cluster are left out) launches this variant so that
// "sparse" saves the work it saves in the reference.  The dense variant carries no test in its inner loop.
template <int DP, int CPW, bool SKIP>
__global__ void __launch_bounds__(256, (DP <= 64 ? 2 : 1)) suffstat_kernel(SuffstatLaunch a) {
  constexpr int NB = DP / 16;
  constexpr int NACC = SSAcc<NB>::N;
  constexpr int BR = SS_BR;
  constexpr int LD = DP + 16;            // padded LDS row stride (doubles)
  constexpr int XBUF = BR * LD;          // doubles per X buffer
  constexpr int NV2 = BR * DP / 2;       // double2 elements per staged batch
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int nwaves = 4, nthr = 256;  // always launched with 4 waves; surplus waves only help staging
  double* xbuf = lds;                              // [2][BR][LD]
  double* qbuf = lds + 2 * XBUF;                   // [2][nwaves][CPW][BR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  const int K = a.K;
  // Workgroup -> (row chunk, cluster slice).  Every slice of a chunk re-reads the
  // same X rows, so the slices of one chunk are placed back-to-back on ONE XCD
  // (block b runs on XCD b % 8, each XCD has its own L2): seq = b / 8 walks
  // (chunk-in-XCD, slice) with the slice fastest.  Pure speed; any placement is correct.
  int chunk, slice;
  {
    const int nslice = a.nslice, nchunks = a.nchunks;
    const int b = blockIdx.x;
    const int full = (nchunks / 8) * 8;
    if (b < full * nslice) {
      const int xcd = b & 7, seq = b >> 3;
      chunk = (seq / nslice) * 8 + xcd;
      slice = seq % nslice;
    } else {
      const int t = b - full * nslice;
      chunk = full + t / nslice;
      slice = t % nslice;
    }
  }
  const int kbase = (slice * nwaves + wave) * CPW;   // may be >= K: the wave still helps staging
  int nk = kbase >= K ? 0 : ((K - kbase) < CPW ? (K - kbase) : CPW);
  int64_t r0 = (int64_t)chunk * a.chunk_rows;
  int64_t r1 = (r0 + a.chunk_rows) < a.NP ? (r0 + a.chunk_rows) : a.NP;
  int kidx[CPW];      // cluster of accumulator set c
  int64_t recidx[CPW];  // its partial record
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    kidx[c] = kbase + c;
    recidx[c] = (int64_t)chunk * K + kbase + c;
  }
  if (a.items) {
    // sparse work list: one block = (row range inside ONE group, up to 4*CPW clusters of that group's active
    // list); work is proportional to the active (row, cluster) pairs, records exist only for those pairs
    const SSItem it = a.items[blockIdx.x];
    r0 = it.r0;
    r1 = it.r1;
    const int first = wave * CPW;
    nk = first >= it.kcnt ? 0 : ((it.kcnt - first) < CPW ? (it.kcnt - first) : CPW);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      kidx[c] = c < nk ? a.klist[it.kofs + first + c] : 0;
      recidx[c] = it.rec0 + first + c;
    }
  }

  double acc[CPW][NACC];
  double sacc[CPW][NB];
  double nacc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    nacc[c] = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[c][i] = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) sacc[c][i] = 0.0;
  }

  // ---- staging: registers hold the next batch while the current one is consumed
  constexpr int NPRE_MAX = (NV2 + nthr - 1) / nthr, npre = NPRE_MAX;
  double pre[NPRE_MAX][2];
  double qpre[CPW];
  auto gload = [&](int64_t b0) {
#pragma unroll
    for (int i = 0; i < NPRE_MAX; ++i) {
      if (i < npre) {
        const int idx = tid + i * nthr;          // double2 index inside the batch: row-major [BR][DP/2]
        const int row = idx / (DP / 2), c2 = idx % (DP / 2);
        double2 v = make_double2(0.0, 0.0);
        if (idx < NV2 && b0 + row < r1) v = *reinterpret_cast<const double2*>(a.X + (b0 + row) * DP + 2 * c2);
        pre[i][0] = v.x;
        pre[i][1] = v.y;
      }
    }
    // q: lanes 0..BR-1 of each wave fetch that wave's CPW columns (coalesced)
    int g = 0;
    const int64_t qrow = b0 + lane;
    const bool qok = lane < BR && qrow < r1;
    if (a.smask && qok) g = a.rginfo[qrow >> 4] >> 5;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      double q = 0.0;
      if (c < nk && qok) {
        q = a.qZ[(int64_t)kidx[c] * a.ldq + qrow];
        if (a.smask && !a.smask[(int64_t)g * K + kidx[c]]) q = 0.0;
      }
      qpre[c] = q;
    }
  };
  auto lstore = [&](int buf) {
    double* xb = xbuf + buf * XBUF;
#pragma unroll
    for (int i = 0; i < NPRE_MAX; ++i) {
      if (i < npre) {
        const int idx = tid + i * nthr;
        const int row = idx / (DP / 2), c2 = idx % (DP / 2);
        if (idx < NV2) *reinterpret_cast<double2*>(xb + row * LD + 2 * c2) = make_double2(pre[i][0], pre[i][1]);
      }
    }
    if (lane < BR) {
      double* qb = qbuf + ((buf * nwaves + wave) * CPW) * BR;
#pragma unroll
      for (int c = 0; c < CPW; ++c) qb[c * BR + lane] = qpre[c];
    }
  };

  if (r0 < r1) {
    gload(r0);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t b0 = r0; b0 < r1; b0 += BR, buf ^= 1) {
    const bool more = b0 + BR < r1;
    if (more) gload(b0 + BR);
    if (nk > 0) {
      const double* xb = xbuf + buf * XBUF + hi * LD + lo2;
      const double* qb = qbuf + ((buf * nwaves + wave) * CPW) * BR + hi;
      if constexpr (DP > 64) {
        // One wave per SIMD (the accumulators need > 256 registers): nothing else hides the LDS
        // latency, so the step loop is software-pipelined by hand, unrolled by two with two register
        // sets (unrotated fragments + q of a step) that swap roles -- no copies.  A step fetches its
        // rotated fragments at the top (first needed after the unrotated MFMAs) and the unrotated
        // fragments and q of the NEXT step; only the first step of a batch waits on LDS.  All BR/4
        // steps run (rows past the chunk end were staged as zeros with q = 0); the last step's prefetch
        // reads one step past the tile: inside the LDS allocation, never used.
        double xA[NB], qA[CPW], xB[NB], qB[CPW];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) xA[jb] = xb[16 * jb + 4 * blk];
#pragma unroll
        for (int c = 0; c < CPW; ++c) qA[c] = qb[c * BR];
        auto step = [&](int st, const double (&x0)[NB], const double (&q)[CPW], double (&x0n)[NB],
                        double (&qn)[CPW]) {
          double xr[NB][4];
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) {
            xr[jb][0] = x0[jb];
#pragma unroll
            for (int s2 = 1; s2 < 4; ++s2) xr[jb][s2] = xb[st * 4 * LD + 16 * jb + 4 * ((blk + s2) & 3)];
          }
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) x0n[jb] = xb[(st + 1) * 4 * LD + 16 * jb + 4 * blk];
#pragma unroll
          for (int c = 0; c < CPW; ++c) qn[c] = qb[c * BR + (st + 1) * 4];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            if (SKIP && __builtin_amdgcn_ballot_w64(q[c] != 0.0) == 0) continue;
            double qx[NB];
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
              qx[jb] = q[c] * xr[jb][0];
              sacc[c][jb] += qx[jb];
            }
            nacc[c] += q[c];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {  // unrotated operands first
              int idx = 0;
#pragma unroll
              for (int jbp = 0; jbp < NB; ++jbp)
#pragma unroll
                for (int jb = 0; jb <= jbp; ++jb)
#pragma unroll
                  for (int s2 = 0; s2 < 4; ++s2)
                    if (s2 < 3 || jb < jbp) {
                      if ((pass == 0) == (s2 == 0)) acc[c][idx] = mfma4(xr[jbp][s2], qx[jb], acc[c][idx]);
                      ++idx;
                    }
            }
          }
        };
        static_assert((BR / 4) % 2 == 0, "step loop is unrolled by two");
#pragma unroll 1
        for (int st = 0; st < BR / 4; st += 2) {
          step(st, xA, qA, xB, qB);
          step(st + 1, xB, qB, xA, qA);
        }
      } else {
        const int nstep = (int)(((r1 - b0) < BR ? (r1 - b0) : BR) / 4);
        for (int st = 0; st < nstep; ++st) {
          if (SKIP) {  // nothing to do for any of this wave's clusters: do not even fetch the fragments
            bool any = false;
#pragma unroll
            for (int c = 0; c < CPW; ++c) any = any || qb[c * BR + st * 4] != 0.0;
            if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
          }
          // fragments: xr[jb][s] = x[row 4*st+hi][16*jb + 4*((blk+s)&3) + lo2]
          double xr[NB][4];
  #pragma unroll
          for (int jb = 0; jb < NB; ++jb)
  #pragma unroll
            for (int s = 0; s < 4; ++s) xr[jb][s] = xb[st * 4 * LD + 16 * jb + 4 * ((blk + s) & 3)];
  #pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const double q = qb[c * BR + st * 4];
            if (SKIP && __builtin_amdgcn_ballot_w64(q != 0.0) == 0) continue;
            double qx[NB];
  #pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
              qx[jb] = q * xr[jb][0];
              sacc[c][jb] += qx[jb];
            }
            nacc[c] += q;
            int idx = 0;
  #pragma unroll
            for (int jbp = 0; jbp < NB; ++jbp) {
  #pragma unroll
              for (int jb = 0; jb <= jbp; ++jb) {
  #pragma unroll
                for (int s = 0; s < 4; ++s) {
                  if (s < 3 || jb < jbp) {
                    acc[c][idx] = mfma4(xr[jbp][s], qx[jb], acc[c][idx]);
                    ++idx;
                  }
                }
              }
            }
          }
        }
      }
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
  if (nk == 0) return;

  const int64_t SS = 1 + (int64_t)DP + (int64_t)DP * DP;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (c < nk) {
      double* out = a.partial + recidx[c] * SS;
      const double nsum = sum_over_hi(nacc[c]);
      if (lane == 0) out[0] = nsum;
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const double s = sum_over_hi(sacc[c][jb]);
        if (hi == 0) out[1 + 16 * jb + lo4] = s;
      }
      double* S = out + 1 + DP;
      int idx = 0;
#pragma unroll
      for (int jbp = 0; jbp < NB; ++jbp) {
#pragma unroll
        for (int jb = 0; jb <= jbp; ++jb) {
          const int ns = jb < jbp ? 4 : 3;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s < ns) {
              const int ti = (blk + s) & 3, tj = blk;
              const int gi = 16 * jbp + 4 * ti + hi, gj = 16 * jb + 4 * tj + lo2;
              const bool diag = jb == jbp;
              // diagonal 16x16 blocks: s=0 gives the diagonal tiles, s=1 every pair {t,t+1 mod 4}
              // once, s=2 the pairs {0,2},{1,3} twice (keep the lower copy); s=3 is never issued
              const bool wr = !diag || ti == tj || s == 1 || ti > tj;
              const double v = acc[c][idx];
              if (wr) {
                S[(int64_t)gi * DP + gj] = v;
                if (!diag || ti != tj) S[(int64_t)gj * DP + gi] = v;
              }
              ++idx;
            }
          }
        }
      }
    }
  }
}

template <int DP>
struct SSCfg;
template <>
struct SSCfg<16> { static constexpr int CPW = 4; };
template <>
struct SSCfg<32> { static constexpr int CPW = 4; };
template <>
struct SSCfg<64> { static constexpr int CPW = 2; };
template <>
struct SSCfg<128> { static constexpr int CPW = 1; };

static int ss_cpw(int DP, int K) {
  if (const char* e = getenv("LC_SS_CPW")) {  // tuning knob
    const int v = atoi(e);
    if (v == 1 || (v == 2 && DP <= 64) || (v == 4 && DP <= 32)) return v;
  }
  int cpw = DP == 16 ? SSCfg<16>::CPW : DP == 32 ? SSCfg<32>::CPW : DP == 64 ? SSCfg<64>::CPW : SSCfg<128>::CPW;
  // few clusters: spread them over more waves instead of stacking them in one
  while (cpw > 1 && (K + cpw - 1) / cpw < 4 && (K + cpw / 2 - 1) / (cpw / 2) <= 4) cpw /= 2;
  return cpw;
}

int suffstat_clusters_per_block(int DP, int K) { return 4 * ss_cpw(DP, K); }

int suffstat_plan(int DP, int64_t NP, int K, int64_t* chunk_rows) {
  const int cpw = ss_cpw(DP, K);
  const int kwaves = (K + cpw - 1) / cpw;  // waves needed to cover the clusters
  // aim for ~8 waves per CU on 256 CUs, at least 256 rows per chunk
  int64_t want = (256 * 8 + kwaves - 1) / kwaves;
  // four blocks per resident slot: the hardware back-fills slots as blocks retire, which evens out the
  // per-CU / per-XCD speed differences (measured 25.7 -> 24.9 ms at N=10M, D=64, K=32), while the partial
  // records (chunks x K x (1 + DP + DP^2) doubles) stay below 1 GiB
  int rounds = 4;
  if (const char* e = getenv("LC_SS_ROUNDS")) {  // tuning knob
    const int v = atoi(e);
    if (v >= 1 && v <= 16) rounds = v;
  }
  const int64_t rec = (int64_t)K * (1 + DP + (int64_t)DP * DP) * 8;
  const int64_t cap = ((int64_t)1 << 30) / rec;
  if (want * rounds <= cap) want *= rounds;
  else if (want < cap) want = cap;
  int64_t maxchunks = (NP + 255) / 256;
  if (want > maxchunks) want = maxchunks;
  if (want < 1) want = 1;
  int64_t rows = (NP + want - 1) / want;
  rows = (rows + SS_BR - 1) / SS_BR * SS_BR;  // whole staging batches
  *chunk_rows = rows;
  return (int)((NP + rows - 1) / rows);
}

template <int DP, int CPW, bool SKIP>
static hipError_t launch_ss_s(const SuffstatLaunch& a, hipStream_t stream) {
  const int kwaves = (a.K + CPW - 1) / CPW;
  const int wpb = 4;
  const int nslice = (kwaves + wpb - 1) / wpb;
  SuffstatLaunch b = a;
  b.nslice = nslice;
  const size_t shmem = (size_t)(2 * SS_BR * (DP + 16) + 2 * wpb * CPW * SS_BR) * sizeof(double);
  auto kern = suffstat_kernel<DP, CPW, SKIP>;
  static bool attr_set = false;
  if (shmem > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const unsigned grid = a.items ? (unsigned)a.nitems : (unsigned)(a.nchunks * nslice);
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(wpb * 64), shmem, stream, b);
  return hipGetLastError();
}

template <int DP, int CPW>
static hipError_t launch_ss_t(const SuffstatLaunch& a, hipStream_t stream) {
  // skip_zero: 1 = skipping variant, -1 = dense even with a mask, 0 = skipping iff a mask is given
  const bool skip = a.skip_zero > 0 || (a.skip_zero == 0 && a.smask);
  return skip ? launch_ss_s<DP, CPW, true>(a, stream) : launch_ss_s<DP, CPW, false>(a, stream);
}

hipError_t launch_suffstat(const SuffstatLaunch& a, hipStream_t stream) {
  if (a.K <= 0 || a.nchunks <= 0) return hipSuccess;
  const int cpw = ss_cpw(a.DP, a.K);
  switch (a.DP) {
    case 16:
      return cpw == 4 ? launch_ss_t<16, 4>(a, stream) : cpw == 2 ? launch_ss_t<16, 2>(a, stream)
                                                                : launch_ss_t<16, 1>(a, stream);
    case 32:
      return cpw == 4 ? launch_ss_t<32, 4>(a, stream) : cpw == 2 ? launch_ss_t<32, 2>(a, stream)
                                                                : launch_ss_t<32, 1>(a, stream);
    case 64:
      return cpw == 2 ? launch_ss_t<64, 2>(a, stream) : launch_ss_t<64, 1>(a, stream);
    case 128:
      return launch_ss_t<128, 1>(a, stream);
  }
  return hipErrorInvalidValue;
}

// ===========================================================================
// Diagonal / exponential cluster families
// ===========================================================================
// NormGamma::Eloglike (src/distributions.cpp:483-492) and ExpGamma::Eloglike (:568-572) inside the same
// vbexpectation (cluster.cpp:91-138):  log q~[n,k] = c_jk + sum_d ( w2_kd (x_nd - a_kd)^2 + w1_kd x_nd ).
// O(N K D) fp64 VALU operations against 8(D+K) bytes per row: at D = 64, K = 32 the two limits are about equal
// (the fp64 vector rate equals the fp64 MFMA rate on this part, and the difference form (x - a)^2 is not
// bilinear, so there is nothing for the matrix pipe to do here).  The design therefore minimises everything
// that is not one of the 3 (NormGamma) / 1 (ExpGamma) operations per (row, cluster, dimension):
//  * a 256-thread block owns 64 rows, staged once in LDS (coalesced load; odd row stride => conflict-free
//    per-lane row reads); lane = row;
//  * wave w owns the cluster tiles {w, w+4, ...} of KT clusters: the tile index is wave-uniform, so the
//    parameters arrive through the scalar cache as SGPR operands (no vector loads, no LDS traffic), and one
//    LDS read of x feeds 3*KT operations;
//  * log q~ stays in registers up to K = 4*DIAG_MAXT*KT clusters; the row maximum and sum cross the four waves
//    through 2 KB of LDS; beyond that the columns are re-read from L2.
// MODE 0: general (a, w2, w1); 1: w1 == 0 (NormGamma); 2: a == w2 == 0 (ExpGamma).
constexpr int DIAG_MAXT = 4;

template <int MODE, int KT>
__device__ __forceinline__ void diag_tile(const double* __restrict__ xr, const double* __restrict__ PA,
                                          const double* __restrict__ PW2, const double* __restrict__ PW1, int k0,
                                          int K, int DP, double (&acc)[KT]) {
  const double* pa[KT];
  const double* p2[KT];
  const double* p1[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int k = k0 + j < K ? k0 + j : K - 1;  // clamped: the caller discards clusters >= K
    pa[j] = PA + (int64_t)k * DP;
    p2[j] = PW2 + (int64_t)k * DP;
    p1[j] = PW1 + (int64_t)k * DP;
    acc[j] = 0.0;
  }
  for (int d = 0; d < DP; d += 4) {
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = xr[d + u];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE != 2) {
          const double t = x[u] - pa[j][d + u];
          acc[j] = fma(t * p2[j][d + u], t, acc[j]);
        }
        if (MODE != 1) acc[j] = fma(p1[j][d + u], x[u], acc[j]);
      }
    }
  }
}

template <int MODE, int KT, bool REG>
__global__ void __launch_bounds__(256)
    estep_diag_kernel(const double* __restrict__ X, const double* __restrict__ PA, const double* __restrict__ PW2,
                      const double* __restrict__ PW1, const double* __restrict__ ctab, const int* __restrict__ rginfo,
                      double* __restrict__ qZ, double* __restrict__ fz_part, double* __restrict__ ll_part, int DP, int K,
                      int64_t NP, int64_t nrows, int64_t ldq, int raw) {
  extern __shared__ double lds[];
  const int LD = DP + 1;
  double* xt = lds;             // [64][LD]
  double* red = lds + 64 * LD;  // [4][64]
  double* llw = red + 256;      // [K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  {
    const int c2 = DP >> 1, sh = __builtin_ctz(c2);  // double2 columns per row (DP is a power of two)
    const double2* X2 = reinterpret_cast<const double2*>(X);
    for (int idx = tid; idx < 64 * c2; idx += 256) {
      const int r = idx >> sh, c = idx & (c2 - 1);
      double2 v = make_double2(0.0, 0.0);
      if (row0 + r < NP) v = X2[(row0 + r) * c2 + c];
      xt[r * LD + 2 * c] = v.x;
      xt[r * LD + 2 * c + 1] = v.y;
    }
  }
  __syncthreads();
  const int64_t row = row0 + lane;
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
  const double* xr = xt + lane * LD;
  const double* crow = ctab + (int64_t)grp * K;
  const int ntiles = (K + KT - 1) / KT;
  const double NINF = -INFINITY;
  double mx = NINF;
  double lq[REG ? DIAG_MAXT : 1][KT], dt[REG ? DIAG_MAXT : 1][KT];

  // ---- pass 1: log q~ for this wave's tiles ----
  if (REG) {
#pragma unroll
    for (int i = 0; i < DIAG_MAXT; ++i) {
      const int tile = w + 4 * i;
#pragma unroll
      for (int j = 0; j < KT; ++j) lq[i][j] = NINF, dt[i][j] = 0.0;
      if (tile < ntiles) {
        double acc[KT];
        diag_tile<MODE, KT>(xr, PA, PW2, PW1, tile * KT, K, DP, acc);
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int k = tile * KT + j;
          if (k < K) {
            const double v = crow[k] + acc[j];
            lq[i][j] = v;
            dt[i][j] = acc[j];
            mx = fmax(mx, v);
            if (raw && inb) qZ[(int64_t)k * ldq + row] = v;
          }
        }
      }
    }
  } else {
    for (int tile = w; tile < ntiles; tile += 4) {
      double acc[KT];
      diag_tile<MODE, KT>(xr, PA, PW2, PW1, tile * KT, K, DP, acc);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int k = tile * KT + j;
        if (k < K) {
          const double v = crow[k] + acc[j];
          mx = fmax(mx, v);
          if (inb) qZ[(int64_t)k * ldq + row] = v;
        }
      }
    }
  }
  if (raw) return;

  // ---- row maximum and sum of exponentials across the four waves (logsumexp, probutils.cpp:141-150) ----
  red[w * 64 + lane] = mx;
  __syncthreads();
  mx = fmax(fmax(red[lane], red[64 + lane]), fmax(red[128 + lane], red[192 + lane]));
  __syncthreads();
  double se = 0.0;
  if (REG) {
#pragma unroll
    for (int i = 0; i < DIAG_MAXT; ++i)
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        lq[i][j] = exp(lq[i][j] - mx);  // exp(-inf) = 0 for the slots past K
        se += lq[i][j];
      }
  } else if (inb) {
    for (int tile = w; tile < ntiles; tile += 4)
      for (int j = 0; j < KT; ++j) {
        const int k = tile * KT + j;
        if (k < K) se += exp(qZ[(int64_t)k * ldq + row] - mx);
      }
  }
  red[w * 64 + lane] = se;
  __syncthreads();
  se = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
  const double logZ = log(se) + mx;
  const double inv = 1.0 / se;

  // ---- pass 2: q = exp(log q~ - logZ), data term of the split ordering ----
  if (REG) {
#pragma unroll
    for (int i = 0; i < DIAG_MAXT; ++i) {
      const int tile = w + 4 * i;
      if (tile < ntiles) {
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int k = tile * KT + j;
          if (k < K) {
            const double q = ok ? lq[i][j] * inv : 0.0;
            if (inb) qZ[(int64_t)k * ldq + row] = q;
            if (ll_part) {
              const double ll = wave_sum(q > 0.0 ? q * dt[i][j] : 0.0);
              if (lane == 0) llw[k] = ll;
            }
          }
        }
      }
    }
  } else {
    for (int tile = w; tile < ntiles; tile += 4)
      for (int j = 0; j < KT; ++j) {
        const int k = tile * KT + j;
        if (k < K) {
          double ll = 0.0;
          if (inb) {
            double* qp = qZ + (int64_t)k * ldq + row;
            const double v = *qp;
            const double q = ok ? exp(v - mx) * inv : 0.0;
            *qp = q;
            if (ll_part && q > 0.0) ll = q * (v - crow[k]);
          }
          if (ll_part) {
            ll = wave_sum(ll);
            if (lane == 0) llw[k] = ll;
          }
        }
      }
  }
  double fz = (w == 0 && ok) ? logZ : 0.0;
  fz = wave_sum(fz);
  __syncthreads();
  if (tid == 0) fz_part[blockIdx.x] = -fz;
  if (ll_part)
    for (int k = tid; k < K; k += 256) ll_part[(int64_t)blockIdx.x * K + k] = llw[k];
}

template <int MODE, int KT, bool REG>
static hipError_t launch_ed_t(const DiagEstepLaunch& a, int64_t grid, size_t shmem, hipStream_t stream) {
  auto kern = estep_diag_kernel<MODE, KT, REG>;
  static size_t attr_set = 0;
  if (shmem > 64 * 1024 && shmem > attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return e;
    attr_set = shmem;
  }
  const double* PA = a.params;
  const double* PW2 = PA + (int64_t)a.K * a.DP;
  const double* PW1 = PW2 + (int64_t)a.K * a.DP;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), shmem, stream, a.X, PA, PW2, PW1, a.ctab, a.rginfo, a.qZ,
                     a.fz_part, a.ll_part, a.DP, a.K, a.nrg * RG, a.nrows, a.ldq, a.raw);
  return hipGetLastError();
}

template <int MODE>
static hipError_t launch_ed_m(const DiagEstepLaunch& a, int64_t grid, size_t shmem, hipStream_t stream) {
  // clusters per tile: every wave should have work (K >= 4*KT), registers hold 4*DIAG_MAXT*KT columns
  if (a.K <= 4) return launch_ed_t<MODE, 1, true>(a, grid, shmem, stream);
  if (a.K <= 8) return launch_ed_t<MODE, 2, true>(a, grid, shmem, stream);
  if (a.K <= 4 * DIAG_MAXT * 4) return launch_ed_t<MODE, 4, true>(a, grid, shmem, stream);
  return launch_ed_t<MODE, 4, false>(a, grid, shmem, stream);
}

hipError_t launch_estep_diag(const DiagEstepLaunch& a, hipStream_t stream) {
  const int64_t grid = (a.nrg * RG + 63) / 64;
  if (grid <= 0) return hipSuccess;
  const size_t shmem = (size_t)(64 * (a.DP + 1) + 256 + a.K) * sizeof(double);
  switch (a.mode) {
    case 1:
      return launch_ed_m<1>(a, grid, shmem, stream);
    case 2:
      return launch_ed_m<2>(a, grid, shmem, stream);
    default:
      return launch_ed_m<0>(a, grid, shmem, stream);
  }
}

// NormGamma::addobs (distributions.cpp:426-438) / ExpGamma::addobs (:533-542):
//   N_k = sum_n q_nk,  x_s[k][d] = sum_n q_nk x_nd,  xx_s[k][d] = sum_n q_nk x_nd^2.
// These are plain GEMMs  Q^T X  and  Q^T X.^2  (reduction over rows; bilinear, so the matrix pipe applies
// without any cancellation concern), 2 N K D MACs against 8(D+K) bytes per row.  A VALU formulation is
// LDS-issue-bound (every FMA pair needs a broadcast q operand from LDS); v_mfma_f64_4x4x4_4b shares each
// operand fragment over 4 x 4 outputs, so one ds_read feeds 16 MACs per lane instead of 1.
//   One MFMA: A[i][k] = q[row k][cluster i] (the same fragment in all four blocks),
//             B_b[k][j] = x[row k][dim 16 jb + 4 b + j]  (b = MFMA block) -> 4 clusters x 16 dims x 4 rows.
//   A wave owns 16 clusters (CT = 4 cluster tiles) x all DP dims x {x, x^2}: CT*NB*2 accumulators.
//   A block = 4 waves = up to 64 clusters ("slice"); with fewer than 3 cluster groups of 16 the waves also
//   split the 4-row steps of a batch (RS = 2 or 4 row classes, each writing its own partial record).
// X batches (BR rows) and the slice's q columns are staged through LDS, next batch in flight in registers,
// the same scheme as suffstat_kernel.
constexpr int SD_QMAX = 64;  // clusters per block
template <int DP, bool SECOND>
__global__ void __launch_bounds__(256, 2) suffstat_diag_kernel(DiagStatLaunch a) {
  constexpr int NB = DP / 16, CT = 4;
  constexpr int BR = DP <= 64 ? 32 : 16;
  constexpr int LD = DP + 16, XBUF = BR * LD;
  constexpr int NV2 = BR * DP / 2, NPRE = (NV2 + 255) / 256;   // double2 per thread and batch
  constexpr int NQ = SD_QMAX * BR / 256;                       // q elements per thread and batch
  constexpr int QLD = BR + 4;                                  // padded q column stride: conflict-free A fragments
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* xbuf = lds;                // [2][BR][LD]
  double* qbuf = lds + 2 * XBUF;     // [2][SD_QMAX][QLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo2 = lane & 3, blk = (lane >> 2) & 3, hi = lane >> 4, lo4 = lane & 15;
  const int K = a.K, RS = a.rsplit;
  const int chunk = blockIdx.x / a.nslice, slice = blockIdx.x % a.nslice;
  const int kb0 = slice * SD_QMAX;
  const int kc = (K - kb0) < SD_QMAX ? (K - kb0) : SD_QMAX;     // clusters of this block
  const int group = RS == 1 ? wave : RS == 2 ? (wave & 1) : 0;  // 16-cluster group of this wave
  const int rcls = RS == 1 ? 0 : RS == 2 ? (wave >> 1) : wave;  // row class (steps st = rcls mod RS)
  const bool active = group * 16 < kc;
  const int64_t r0 = (int64_t)chunk * a.chunk_rows;
  const int64_t r1 = (r0 + a.chunk_rows) < a.NP ? (r0 + a.chunk_rows) : a.NP;

  double acc1[CT][NB], acc2[CT][NB], nacc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    nacc[c] = 0.0;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) acc1[c][jb] = acc2[c][jb] = 0.0;
  }

  double pre[NPRE][2], qpre[NQ];
  auto gload = [&](int64_t b0) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * 256;  // double2 index inside the batch, row-major [BR][DP/2]
      const int row = idx / (DP / 2), c2 = idx % (DP / 2);
      double2 v = make_double2(0.0, 0.0);
      if (idx < NV2 && b0 + row < r1) v = *reinterpret_cast<const double2*>(a.X + (b0 + row) * DP + 2 * c2);
      pre[i][0] = v.x;
      pre[i][1] = v.y;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + i * 256;  // [SD_QMAX][BR]
      const int kk = idx / BR, r = idx % BR;
      double q = 0.0;
      if (kk < kc && b0 + r < r1) {
        q = a.qZ[(int64_t)(kb0 + kk) * a.ldq + b0 + r];
        if (a.smask && !a.smask[(int64_t)(a.rginfo[(b0 + r) >> 4] >> 5) * K + kb0 + kk]) q = 0.0;
      }
      qpre[i] = q;
    }
  };
  auto lstore = [&](int buf) {
    double* xb = xbuf + buf * XBUF;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DP / 2), c2 = idx % (DP / 2);
      if (idx < NV2) *reinterpret_cast<double2*>(xb + row * LD + 2 * c2) = make_double2(pre[i][0], pre[i][1]);
    }
    double* qb = qbuf + buf * SD_QMAX * QLD;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + i * 256;
      qb[(idx / BR) * QLD + idx % BR] = qpre[i];
    }
  };

  if (r0 < r1) {
    gload(r0);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t b0 = r0; b0 < r1; b0 += BR, buf ^= 1) {
    const bool more = b0 + BR < r1;
    if (more) gload(b0 + BR);
    if (active) {
      const double* xb = xbuf + buf * XBUF + hi * LD + 4 * blk + lo2;
      const double* qb = qbuf + buf * SD_QMAX * QLD + (group * 16 + lo2) * QLD + hi;
      // rows past the chunk end were staged as zeros with q = 0, so every step runs
      for (int st = rcls; st < BR / 4; st += RS) {
        double xf[NB], x2[NB];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          xf[jb] = xb[st * 4 * LD + 16 * jb];
          if (SECOND) x2[jb] = xf[jb] * xf[jb];
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const double q = qb[4 * c * QLD + st * 4];
          nacc[c] += q;
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) {
            acc1[c][jb] = mfma4(q, xf[jb], acc1[c][jb]);
            if (SECOND) acc2[c][jb] = mfma4(q, x2[jb], acc2[c][jb]);
          }
        }
      }
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
  if (!active) return;

  // output lane (lo2, blk, hi) of accumulator (c, jb): cluster 4 c + hi, dimension 16 jb + 4 blk + lo2
  const int64_t SS = 1 + 2 * (int64_t)DP;
  double* rec = a.partial + ((int64_t)(chunk * RS + rcls) * K + kb0 + group * 16) * SS;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    // N_k: this lane summed q[row class hi][cluster 4 c + lo2] (replicated over blk)
    const double n = sum_over_hi(nacc[c]);
    if (hi == 0 && blk == 0 && group * 16 + 4 * c + lo2 < kc) rec[(int64_t)(4 * c + lo2) * SS] = n;
    if (group * 16 + 4 * c + hi < kc) {
      double* out = rec + (int64_t)(4 * c + hi) * SS;
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        out[1 + 16 * jb + lo4] = acc1[c][jb];
        out[1 + DP + 16 * jb + lo4] = SECOND ? acc2[c][jb] : 0.0;
      }
    }
  }
}

int suffstat_diag_rsplit(int K) {  // row classes per block: waves left over by the cluster groups split the rows
  const int groups = ((K < SD_QMAX ? K : SD_QMAX) + 15) / 16;
  return groups >= 3 ? 1 : groups == 2 ? 2 : 4;
}

template <int DP>
static hipError_t launch_sd_t(const DiagStatLaunch& a, hipStream_t stream) {
  constexpr int BR = DP <= 64 ? 32 : 16;
  const size_t shmem = (size_t)(2 * BR * (DP + 16) + 2 * SD_QMAX * (BR + 4)) * sizeof(double);
  static bool attr_set = false;
  if (shmem > 64 * 1024 && !attr_set) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(suffstat_diag_kernel<DP, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(suffstat_diag_kernel<DP, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e1 != hipSuccess) return e1;
    if (e2 != hipSuccess) return e2;
    attr_set = true;
  }
  const dim3 grid((unsigned)(a.nchunks * a.nslice));
  if (a.second)
    hipLaunchKernelGGL((suffstat_diag_kernel<DP, true>), grid, dim3(256), shmem, stream, a);
  else
    hipLaunchKernelGGL((suffstat_diag_kernel<DP, false>), grid, dim3(256), shmem, stream, a);
  return hipGetLastError();
}

hipError_t launch_suffstat_diag(const DiagStatLaunch& a0, hipStream_t stream) {
  if (a0.K <= 0 || a0.nchunks <= 0) return hipSuccess;
  if (a0.chunk_rows % 32) return hipErrorInvalidValue;
  DiagStatLaunch a = a0;
  a.nslice = (a.K + SD_QMAX - 1) / SD_QMAX;
  a.rsplit = suffstat_diag_rsplit(a.K);
  switch (a.DP) {
    case 16:
      return launch_sd_t<16>(a, stream);
    case 32:
      return launch_sd_t<32>(a, stream);
    case 64:
      return launch_sd_t<64>(a, stream);
    case 128:
      return launch_sd_t<128>(a, stream);
  }
  return hipErrorInvalidValue;
}

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

hipError_t launch_group_colsum(const double* qZ, int64_t ldq, int K, const int64_t* goff, int J, double* out,
                               hipStream_t stream) {
  if (K <= 0 || J <= 0) return hipSuccess;
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
__global__ void __launch_bounds__(256) split_init_kernel(const double* X, int DP, int D, int64_t NP, const int* rginfo,
                                                         int64_t nrows, const double* mv, double* q, int64_t ldq,
                                                         int mode, const double* thr) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= NP) return;
  int grp = 0;
  bool ok;
  if (rginfo) {
    const int info = rginfo[row >> 4];
    grp = info >> 5;
    ok = (int)(row & 15) < (info & 31);
  } else {
    ok = row < nrows;
  }
  double q0 = 0.0, q1 = 0.0;
  if (ok) {
    if (mode == 2) {
      q0 = q[row] > thr[grp] ? 1.0 : 0.0;
      q1 = 1.0 - q0;
    } else {
      double s = 0.0;
      if (mode == 0) {
        for (int d = 0; d < D; ++d) s += (X[row * DP + d] - mv[d]) * mv[DP + d];
        q0 = s >= 0.0 ? 1.0 : 0.0;
        q1 = 1.0 - q0;
      } else {
        for (int d = 0; d < D; ++d) s += X[row * DP + d] * mv[DP + d];
        q0 = s;
      }
    }
  }
  q[row] = q0;
  q[ldq + row] = q1;
}

hipError_t launch_split_init(const double* X, int DP, int D, int64_t NP, const int* rginfo, int64_t nrows,
                             const double* mv, double* q, int64_t ldq, int mode, const double* thr,
                             hipStream_t stream) {
  if (NP <= 0) return hipSuccess;
  hipLaunchKernelGGL(split_init_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, stream, X, DP, D, NP, rginfo,
                     nrows, mv, q, ldq, mode, thr);
  return hipGetLastError();
}

// auglabels (src/comutils.cpp:75-104) straight from the refined sub-problem: every selected row whose
// second refined responsibility exceeds 0.5 moves its column-k mass to the new column K
__global__ void __launch_bounds__(256) aug_from_sub_kernel(double* q, int64_t ldq, int k, int K, const int64_t* idx,
                                                           int64_t M, const int64_t* starts, const int64_t* goff_sub,
                                                           int J, const double* qsub1) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t sub = sel_dst_row(p, starts, goff_sub, J);
  if (qsub1[sub] > 0.5) {
    const int64_t r = idx[p];
    q[(int64_t)K * ldq + r] = q[(int64_t)k * ldq + r];
    q[(int64_t)k * ldq + r] = 0.0;
  }
}

hipError_t launch_aug_from_sub(double* q, int64_t ldq, int k, int K, const int64_t* idx, int64_t M,
                               const int64_t* starts, const int64_t* goff_sub, int J, const double* qsub1,
                               hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(aug_from_sub_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, q, ldq, k, K, idx, M,
                     starts, goff_sub, J, qsub1);
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

}  // namespace lck
