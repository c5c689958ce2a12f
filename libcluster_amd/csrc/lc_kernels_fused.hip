// Small observations (D <= 16): one persistent pass that does the E-step of iteration t AND the sufficient statistics
// iteration t + 1 starts from.
//
//   vbexpectation  src/cluster.cpp:91-138  (-> GaussWish::Eloglike distributions.cpp:356-370, mahaldist probutils.cpp:113-138,
//                                              logsumexp probutils.cpp:141-150)
//   updateSS       src/cluster.cpp:53-82   (-> GaussWish::addobs distributions.cpp:301-313)
//
// The reference runs updateSS at the top of iteration t + 1 on the qZ of iteration t (cluster.cpp:198-223); the
// statistics are a pure function of (X, qZ), so producing them right behind the E-step in the same pass is the same
// arithmetic.  At D = 16 the two separate kernels are 0.1 ms launches bound by latency, not by a pipe (a block's work
// per barrier is a few dozen MFMAs); here
//   * a block keeps the parameters of ALL clusters in LDS (1.4 KB each): no staging loop, no barrier per cluster;
//   * it walks tiles of 256 rows: the tile of X is staged in LDS once and serves both halves, q goes to HBM once and to
//     an LDS table the statistics half reads back (the statistics never re-read q or X from memory);
//   * the statistics accumulators live in registers across all the tiles of the block: one partial record per
//     (block, cluster) at the very end, folded by reduce_partials_kernel in fixed order.
// Layouts are those of estep_kernel (row-owning scheme: lane (lo4, hi) owns row 16 hi + lo4 of its wave's 64 rows)
// and of suffstat_kernel (MFMA block blk computes tile ((blk + s) & 3, blk) of the one 16 x 16 block, s = 0..2).
#include "lc_device.hpp"

#include <algorithm>

namespace lck {

constexpr int FUSED_KMAX = 16;  // clusters per block-resident parameter set (four per wave in the statistics half)
constexpr int FUSED_ROWS = 256; // rows per tile

// CPW: clusters per wave in the statistics half (K <= 4 CPW)
template <int DP, int CPW>
__global__ void __launch_bounds__(256, 2) fused_small_kernel(FusedLaunch a) {
  static_assert(DP == 16, "one 16 x 16 block of S_k (the general blocking lives in suffstat_kernel)");
  constexpr int NT = DP / 4;
  constexpr int NTILES = NT * (NT + 1) / 2;
  constexpr int NREAD = NTILES + NT;
  constexpr int PF = 6;
  constexpr int PS = NTILES * 16 + DP;
  // row stride of the staged tile (36 dwords).  E-step half: a half-wave reads rows lo4 = 0..15 at two columns -- 36 lo4
  // mod 64 are sixteen different multiples of 4: conflict-free.  Statistics half: a half-wave reads all 16 columns of TWO
  // rows, which must lie 32 banks apart: rows r and r + 8 do (8 x 36 = 4 x 64 + 32), consecutive rows do not (the bank
  // conflicts of the round-2 profile) -- so a four-row step takes rows {t, t + 8, t + 4, t + 12} of a 16-row block (any
  // four rows serve as the reduction index of the MFMA, as long as q is read for the same rows)
  constexpr int LD = DP + 2;
  constexpr int R = 4;
  static_assert(4 * CPW <= FUSED_KMAX, "statistics accumulators");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int K = a.K;
  double* xt = lds;                    // [256][LD]
  double* par = xt + FUSED_ROWS * LD;  // [K][PS]
  double* qt = par + (size_t)K * PS;   // [K][256]: log q~, then q, of the tile's rows
  double* llw = qt + (size_t)K * 256;  // [4][K]
  double* fzw = llw + 4 * K;           // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  for (int i = tid; i < K * PS; i += 256) par[i] = a.params[i];
  for (int i = tid; i < 4 * K; i += 256) llw[i] = 0.0;
  // statistics half: wave w owns clusters w, w + 4, w + 8, w + 12
  double acc[CPW][3], sacc[CPW], nacc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    sacc[c] = nacc[c] = 0.0;
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[c][s] = 0.0;
  }
  double fz = 0.0;
  const int64_t NP = a.nrg * RG;
  const int64_t ntile = (NP + FUSED_ROWS - 1) / FUSED_ROWS;
  // register double-buffer for the next tile of X (coalesced double2 pieces; rows past the end as zeros)
  constexpr int C2 = DP / 2, NPRE = FUSED_ROWS * C2 / 256;
  double2 pre[NPRE];
  auto fetch = [&](int64_t tile) {
    const int64_t r0 = tile * FUSED_ROWS;
    const int64_t left = tile < ntile ? NP - r0 : 0;
    const int lim = (int)(left < FUSED_ROWS ? left : FUSED_ROWS) * C2;
    const double2* X2 = reinterpret_cast<const double2*>(a.X) + (tile < ntile ? r0 : 0) * C2;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * 256;
      pre[i] = idx < lim ? X2[idx] : make_double2(0.0, 0.0);
    }
  };
  fetch(blockIdx.x);
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t row0 = tile * FUSED_ROWS;
    // ---- the tile of X -> LDS (its loads were issued a whole tile ago)
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / C2, c2 = idx % C2;
      xt[row * LD + 2 * c2] = pre[i].x;
      xt[row * LD + 2 * c2 + 1] = pre[i].y;
    }
    __syncthreads();
    fetch(tile + gridDim.x);  // in flight during both halves of this tile

    // ---- E-step half: this wave's 64 rows as four row groups
    const int64_t rg0 = tile * (FUSED_ROWS / RG) + wave * R;
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
      const double* xr = xt + (wave * 64 + r * 16 + lo4) * LD + hi;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) xf[r][jt] = xr[4 * jt];
    }
    bool myok = false, myrow = false;
    int mygrp = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (hi == r) myok = rgok[r], myrow = rowok[r], mygrp = grp[r];
    double mymx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const double* P = par + (size_t)k * PS;
      const double* Pt = P + (lane & 3) + 4 * hi;  // this lane's element of every 4x4 tile
      const double* Pb = P + NTILES * 16 + hi;     // this lane's element of every 4-vector of -b
      double ring[PF];
      static_for<PF>([&](auto ic) {
        constexpr RdInfo ri = rd_info(ic);
        ring[ic] = ri.jt < 0 ? Pb[ri.off] : Pt[ri.off];
      });
      double d2[R], acc1[R];
#pragma unroll
      for (int r = 0; r < R; ++r) d2[r] = 0.0;
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
          for (int r = 0; r < R; ++r) acc1[r] = v;  // y starts at -b: y = A x - b
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) acc1[r] = mfma4(v, xf[r][ri.jt], acc1[r]);
          if constexpr (ri.jt == ri.it) {
#pragma unroll
            for (int r = 0; r < R; ++r) d2[r] = fma(acc1[r], acc1[r], d2[r]);
          }
        }
      });
      double lqsel = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double dd = mfma4(1.0, d2[r], 0.0);  // sum over the four hi lanes, total in every lane
        const double lq = a.ctab[(int64_t)grp[r] * K + k] - 0.5 * dd;
        if (hi == r) lqsel = lq;
      }
      mymx = fmax(mymx, lqsel);
      qt[k * 256 + tid] = lqsel;  // (this lane's own slot: no barrier needed before it reads it back)
    }
    // logsumexp and normalisation in the reference's order: max, sum exp(x - max), log + max, exp(x - logZ)
    {
      // (without LL_k: ONE exponential per entry -- e = exp(log q~ - max) goes back into the lane's slot and q = e / sum(e);
      //  the same sum and logZ, q within 2 ulp of exp(log q~ - logZ))
      double s = 0.0;
      const bool onexp = !a.want_ll;
      for (int k = 0; k < K; ++k) {
        const double e = exp(qt[k * 256 + tid] - mymx);
        s += e;
        if (onexp) qt[k * 256 + tid] = e;
      }
      const double logZ = log(s) + mymx;
      const double inv = 1.0 / s;
      double* qp = a.qZ + row0 + tid;
      for (int k = 0; k < K; ++k) {
        const double lq = qt[k * 256 + tid];
        double q = onexp ? lq * inv : exp(lq - logZ);
        if (!myok || !myrow) q = 0.0;
        if (myok) qp[(int64_t)k * a.ldq] = q;
        qt[k * 256 + tid] = q;
        if (a.want_ll) {  // wave-uniform
          const double ll = wave_sum(q > 0.0 ? q * (lq - a.ctab[(int64_t)mygrp * K + k]) : 0.0);
          if (lane == 0) llw[wave * K + k] += ll;
        }
      }
      if (myok && myrow) fz += logZ;
    }
    __syncthreads();

    // ---- statistics half: 64 four-row steps over the tile, this wave's clusters
    {
      const int rsub = 8 * (hi & 1) + 4 * (hi >> 1);  // this lane's row of a step inside its 16-row block (see LD)
      const double* xb = xt + rsub * LD + lo2;
      const double* qb = qt + rsub;
#pragma unroll 4
      for (int st = 0; st < FUSED_ROWS / 4; ++st) {
        const int rbase = (st >> 2) * 16 + (st & 3);
        double xr[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) xr[s] = xb[rbase * LD + 4 * ((blk + s) & 3)];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const int k = wave + 4 * c;
          if (k < K) {  // wave-uniform
            const double q = qb[k * 256 + rbase];
            const double qx = q * xr[0];
            sacc[c] += qx;
            nacc[c] += q;
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[c][s] = mfma4(xr[s], qx, acc[c][s]);
          }
        }
      }
    }
    __syncthreads();  // the next tile overwrites xt and qt
  }

  // ---- one partial record per (block, cluster): [N_k, s_k(DP), S_k(DP x DP)]
  const int64_t SS = 1 + DP + DP * DP;
  double* rec = a.partial + (int64_t)blockIdx.x * (K * SS + 1 + K);
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int k = wave + 4 * c;
    if (k < K) {
      double* out = rec + (int64_t)k * SS;
      const double nsum = sum_over_hi(nacc[c]);
      if (lane == 0) out[0] = nsum;
      const double ssum = sum_over_hi(sacc[c]);
      if (hi == 0) out[1 + lo4] = ssum;
      double* S = out + 1 + DP;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        // s = 0: the diagonal tiles; s = 1: every pair {t, t + 1 mod 4} once; s = 2: the pairs {0,2}, {1,3} twice
        // (the lower copy is kept)
        const int ti = (blk + s) & 3, tj = blk;
        const int gi = 4 * ti + hi, gj = 4 * tj + lo2;
        if (ti == tj || s == 1 || ti > tj) {
          S[gi * DP + gj] = acc[c][s];
          if (ti != tj) S[gj * DP + gi] = acc[c][s];
        }
      }
    }
  }
  fz = wave_sum(fz);
  if (lane == 0) fzw[wave] = fz;
  __syncthreads();
  for (int k = tid; k < K; k += 256)
    rec[K * SS + 1 + k] = a.want_ll ? llw[k] + llw[K + k] + llw[2 * K + k] + llw[3 * K + k] : 0.0;
  if (tid == 0) rec[K * SS] = -(fzw[0] + fzw[1] + fzw[2] + fzw[3]);  // cluster.cpp:137 returns -sum(logZ)
}

static size_t fused_lds_bytes(int DP, int K) {
  const int NT = DP / 4, PS = NT * (NT + 1) / 2 * 16 + DP;
  return ((size_t)FUSED_ROWS * (DP + 2) + (size_t)K * PS + (size_t)K * 256 + 4 * K + 4) * sizeof(double);
}

// does this shape have a fused path?  (a property of (DP, K) alone: every rank of a distributed run must take the same
// branch whatever its share of the rows, an empty share included)
bool fused_eligible(int DP, int K) {
  static const bool off = getenv("LC_FUSED_SMALL") && atoi(getenv("LC_FUSED_SMALL")) == 0;
  if (off || DP != 16 || K < 1 || K > FUSED_KMAX) return false;
  // the pass keeps a 256-row tile, all K parameter records and a K x 256 table in LDS (93 KB at K = 16): the answer must
  // also hold on the device at hand -- and be the same on every rank, so it is asked of the architecture the library is
  // built for (gfx950: 160 KB per workgroup), never of "whichever device is current"
  constexpr size_t GFX950_LDS_PER_BLOCK = 160 * 1024;
  return fused_lds_bytes(DP, K) <= GFX950_LDS_PER_BLOCK;
}

// number of persistent blocks (= partial records per cluster, fz / ll partial slots)
int fused_plan(int DP, int64_t nrg, int K) {
  if (!fused_eligible(DP, K) || nrg <= 0) return 0;
  // compute units of the CURRENT device (a process may hold contexts on several)
  static int cus_of[16] = {};
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
  if (dev >= 0) cus = cus_of[dev];
  if (!cus) {
    hipDeviceProp_t p;
    if (dev >= 0 && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
    if (dev >= 0) cus_of[dev] = cus;
  }
  const int64_t ntile = (nrg * RG + FUSED_ROWS - 1) / FUSED_ROWS;
  const size_t lds = fused_lds_bytes(DP, K);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  return (int)std::min<int64_t>(ntile, (int64_t)cus * per_cu);
}

hipError_t launch_fused(const FusedLaunch& a, hipStream_t stream) {
  if (a.DP != 16 || a.grid <= 0) return hipErrorInvalidValue;
  const size_t shmem = fused_lds_bytes(a.DP, a.K);
  static LdsGrant grants[3];
  auto go = [&](auto kern, LdsGrant& g) {
    if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, g); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)a.grid), dim3(256), shmem, stream, a);
    return hipGetLastError();
  };
  if (a.K <= 4) return go(fused_small_kernel<16, 1>, grants[0]);
  if (a.K <= 8) return go(fused_small_kernel<16, 2>, grants[1]);
  return go(fused_small_kernel<16, 4>, grants[2]);
}

}  // namespace lck
