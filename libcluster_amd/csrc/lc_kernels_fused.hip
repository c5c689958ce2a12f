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
// Layouts: the E-step half is estep_kernel's row-owning scheme (lane (lo4, hi) owns row 16 hi + lo4 of its wave's 64
// rows); the statistics half is the feature GEMM of suffstat_feat_kernel (round 3: the cluster index inside the MFMA --
// A = q[row][cluster quad], B = x_i x_j for one 4 x 4 patch of the 16 x 16 matrix, one multiply per patch for ALL
// cluster quads; s_k and N_k are the features x_i * 1 and 1 * 1 with a column of ones in the staged tile).  Twelve tiles
// (10 patches + s_k + N_k) x NQ cluster quads per wave; the four waves split the tile's 64 four-row steps and meet once,
// at the very end, through LDS in wave order.
#include "lc_device.hpp"

#include <algorithm>

namespace lck {

constexpr int FUSED_KMAX = 16;  // clusters per block-resident parameter set (four per wave in the statistics half)
constexpr int FUSED_ROWS = 256; // rows per tile
constexpr int FUSED_QS = FUSED_ROWS + 2;  // q table: 258 doubles per cluster = 4 banks between consecutive clusters

// CPW: cluster quads of the statistics half (K <= 4 CPW)
// ONEGRP: one group, known at compile time (no row-group table).  As a run-time choice the table read and the computed
// `info` end in one register, which the compiler guards with `s_waitcnt vmcnt(0)` in BOTH paths -- four full drains of
// the just-issued prefetch at the head of every tile.
// GRP 0 = that instance; 1: several groups, the J x K table c_jk waits in LDS (J x K <= FUSED_CT_CAP); 2: several groups, the
// table is read from global memory inside the cluster loop (each read then waits for the prefetch as well).  Compile-time
// because a run-time choice between an LDS read and a global read ends in a combined `vmcnt(0) lgkmcnt(0)` at the merge.
template <int DP, int CPW, int GRP = 2>
__global__ void __launch_bounds__(256, 2) fused_small_kernel(FusedLaunch a_) {
  constexpr bool ONEGRP = GRP == 0, CTLDS = GRP != 2;
  FusedLaunch a = a_;
  if constexpr (ONEGRP) a.rginfo = nullptr;
  static_assert(DP == 16, "one 16 x 16 block of S_k (the general blocking lives in suffstat_kernel)");
  constexpr int NT = DP / 4;
  constexpr int NTILES = NT * (NT + 1) / 2;
  constexpr int NREAD = NTILES + NT;
  constexpr int PF = 6;
  constexpr int PS = NTILES * 16 + DP;
  // row stride of the staged tile (36 dwords).  E-step half: a half-wave reads rows lo4 = 0..15 at two columns -- 36 lo4
  // mod 64 are sixteen different multiples of 4: conflict-free.  Statistics half (feature form): a half-wave reads up to
  // sixteen columns of two rows that lie 8 rows apart (288 dwords = 32 banks: disjoint), and q[row][cluster] with
  // consecutive clusters 4 banks apart (FUSED_QS): conflict-free as well.  Column DP of every row holds 1.0 (s_k, N_k).
  constexpr int LD = DP + 2;
  constexpr int R = 4;
  static_assert(4 * CPW <= FUSED_KMAX, "statistics accumulators");
  constexpr int NQ = CPW, NTL = 12;    // cluster quads; feature tiles: 10 patches (ia <= ja), s_k, N_k
  constexpr int QS = FUSED_QS;         // row stride of the q table: consecutive clusters 8 banks apart
  constexpr int ONE = DP;              // column of the staged tile that holds 1.0
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int K = a.K;
  double* xt = lds;                    // [256][LD]
  double* par = xt + FUSED_ROWS * LD;  // [K][PS]
  double* qt = par + (size_t)K * PS;   // [4 NQ][QS]: log q~, then q, of the tile's rows (clusters >= K: zeros)
  double* llw = qt + (size_t)4 * NQ * QS;  // [4][K]
  double* fzw = llw + 4 * K;           // [4]
  // One group: the K constants c_k wait in LDS.  Read from global memory inside the cluster loop they are vector loads
  // issued AFTER the next tile's prefetch, and the vector-memory counter retires in order: every wait for a constant was
  // a wait for the prefetch (an HBM round trip per tile, exposed).  With several groups the table is still read there.
  double* ctl = fzw + 4;               // [FUSED_CT_CAP]: the table c_jk when it fits (J x K entries), else unused
  const int ctrows = ONEGRP ? 1 : a.ngroups;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  for (int i = tid; i < K * PS; i += 256) par[i] = a.params[i];
  for (int i = tid; i < 4 * K; i += 256) llw[i] = 0.0;
  if constexpr (CTLDS)
    for (int i = tid; i < ctrows * K; i += 256) ctl[i] = a.ctab[i];
  for (int i = tid; i < 4 * NQ * QS; i += 256) qt[i] = 0.0;
  xt[tid * LD + ONE] = 1.0;  // (the staging below writes columns 0 .. DP - 1 only)
  double acc[NTL][NQ];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int c = 0; c < NQ; ++c) acc[t][c] = 0.0;
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
      // (one 16-byte store: its 8-lane groups fill one row's 128 bytes; two 8-byte stores put two rows, 36 dwords apart,
      //  into a 16-lane group: a 2-way conflict on the 32 write banks)
      *reinterpret_cast<double2*>(xt + row * LD + 2 * c2) = pre[i];
    }
    __syncthreads();

    // ---- E-step half: this wave's 64 rows as four row groups
    // (the row groups' table entries are read BEFORE the next tile's prefetch goes out: they are waited for at once, and
    //  the vector-memory counter retires in order)
    const int64_t rg0 = tile * (FUSED_ROWS / RG) + wave * R;
    double xf[R][NT];
    int grp[R];
    bool rowok[R], rgok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t rg = rg0 + r;
      rgok[r] = rg < a.nrg;
      int info = 0;
      if constexpr (ONEGRP) {
        const int64_t rem = a.nrows - rg * RG;
        info = !rgok[r] ? 0 : rem >= RG ? RG : (rem > 0 ? (int)rem : 0);
      } else if (rgok[r]) {
        if (a.rginfo) {
          info = a.rginfo[rg];
        } else {
          const int64_t rem = a.nrows - rg * RG;
          info = rem >= RG ? RG : (rem > 0 ? (int)rem : 0);
        }
      }
      if constexpr (!ONEGRP) asm volatile("" : "+v"(info));  // (the table entry has to be HERE: its wait stands in front of the prefetch)
      grp[r] = info >> 5;
      rowok[r] = lo4 < (info & 31);
    }
    if constexpr (!ONEGRP) __builtin_amdgcn_sched_barrier(0);
    fetch(tile + gridDim.x);  // in flight during both halves of this tile
#pragma unroll
    for (int r = 0; r < R; ++r) {
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
        double cjk;
        if constexpr (ONEGRP) cjk = ctl[k];
        else if constexpr (CTLDS) cjk = ctl[grp[r] * K + k];
        else cjk = a.ctab[(int64_t)grp[r] * K + k];
        const double lq = cjk - 0.5 * dd;
        if (hi == r) lqsel = lq;
      }
      mymx = fmax(mymx, lqsel);
      qt[k * QS + tid] = lqsel;  // (this lane's own slot: no barrier needed before it reads it back)
    }
    // logsumexp and normalisation in the reference's order: max, sum exp(x - max), log + max, exp(x - logZ)
    {
      // (without LL_k: ONE exponential per entry -- e = exp(log q~ - max) goes back into the lane's slot and q = e / sum(e);
      //  the same sum and logZ, q within 2 ulp of exp(log q~ - logZ))
      double s = 0.0;
      const bool onexp = !a.want_ll;
      for (int k = 0; k < K; ++k) {
        const double e = exp(qt[k * QS + tid] - mymx);
        s += e;
        if (onexp) qt[k * QS + tid] = e;
      }
      const double logZ = log(s) + mymx;
      const double inv = 1.0 / s;
      double* qp = a.qZ + row0 + tid;
      for (int k = 0; k < K; ++k) {
        const double lq = qt[k * QS + tid];
        double q = onexp ? lq * inv : exp(lq - logZ);
        if (!myok || !myrow) q = 0.0;
        if (myok) qp[(int64_t)k * a.ldq] = q;
        qt[k * QS + tid] = q;
        if (a.want_ll) {  // wave-uniform
          double cjk;
          if constexpr (ONEGRP) cjk = ctl[k];
          else if constexpr (CTLDS) cjk = ctl[mygrp * K + k];
          else cjk = a.ctab[(int64_t)mygrp * K + k];
          const double ll = wave_sum(q > 0.0 ? q * (lq - cjk) : 0.0);
          if (lane == 0) llw[wave * K + k] += ll;
        }
      }
      if (myok && myrow) fz += logZ;
    }
    __syncthreads();

    // ---- statistics half: this wave's 16 of the tile's 64 four-row steps, all feature tiles, all cluster quads.
    // Tile t < 10 is the patch (ia, ja): x[row][4 ia + lo2] * x[row][4 ja + blk] -- two lane-dependent base pointers and
    // compile-time column offsets; t = 10: x[row][4 blk + lo2] * 1 (s_k); t = 11: 1 * 1 (N_k).
    {
      // a step's four rows are {t, t + 8, t + 4, t + 12} of a 16-row block: the two rows of a half-wave then lie 32 banks
      // apart (8 x 36 dwords), which every fragment of the step needs -- the 8-bank patch operands and the 32-bank s_k
      // tile alike -- and in the q table 16 dwords apart, between the 4-dword steps of consecutive clusters (FUSED_QS)
      const int rsub = 8 * (hi & 1) + 4 * (hi >> 1);
      const double* xu = xt + rsub * LD + lo2;        // + 4 ia
      const double* xw = xt + rsub * LD + blk;        // + 4 ja
      const double* xs = xt + rsub * LD + 4 * blk + lo2;
      const double* x1 = xt + rsub * LD + ONE;
      const double* qb = qt + lo2 * QS + rsub;
#pragma unroll 2
      for (int s4 = 0; s4 < FUSED_ROWS / 16; ++s4) {
        const int st = 4 * s4 + wave, rbase = (st >> 2) * 16 + (st & 3), ro = rbase * LD;
        double qa[NQ];
#pragma unroll
        for (int c = 0; c < NQ; ++c) qa[c] = qb[4 * c * QS + rbase];
        const double one = x1[ro];
        static_for<NTL>([&](auto tc) {
          constexpr int t = tc;
          double p;
          if constexpr (t < 10) {
            constexpr int ja = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3, ia = t - ja * (ja + 1) / 2;
            p = xu[ro + 4 * ia] * xw[ro + 4 * ja];
          } else if constexpr (t == 10) {
            p = xs[ro];
          } else {
            p = one;
          }
#pragma unroll
          for (int c = 0; c < NQ; ++c) acc[t][c] = mfma4(qa[c], p, acc[t][c]);
        });
      }
    }
    __syncthreads();  // the next tile overwrites xt and qt
  }

  // ---- one partial record per (block, cluster): [N_k, s_k(DP), S_k(DP x DP)].  The four waves' accumulators meet in
  // LDS in wave order (fixed: deterministic); then every thread writes its share of the 12 x NQ x 64 entries.
  const int64_t SS = 1 + DP + DP * DP;
  double* rec = a.partial + (int64_t)blockIdx.x * (K * SS + 1 + K);
  {
    double* red = xt;  // (the tile is not needed any more: 12 x NQ x 64 doubles <= 24.6 KB)
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
          for (int c = 0; c < NQ; ++c) {
            double* r = red + (t * NQ + c) * 64 + lane;
            *r = w == 0 ? acc[t][c] : *r + acc[t][c];
          }
      }
      __syncthreads();
    }
    for (int e = tid; e < NTL * NQ * 64; e += 256) {
      const int l = e & 63, tc = e >> 6, t = tc / NQ, c = tc % NQ;
      const int h = l >> 4, b = (l >> 2) & 3, lo = l & 3, k = 4 * c + h;
      if (k >= K) continue;
      double* out = rec + (int64_t)k * SS;
      const double v = red[e];
      if (t < 10) {
        int ja = 0;
        while ((ja + 1) * (ja + 2) / 2 <= t) ++ja;
        const int ia = t - ja * (ja + 1) / 2, gi = 4 * ia + lo, gj = 4 * ja + b;
        out[1 + DP + gi * DP + gj] = v;
        out[1 + DP + gj * DP + gi] = v;
      } else if (t == 10) {
        out[1 + 4 * b + lo] = v;
      } else if (b == 0 && lo == 0) {
        out[0] = v;
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
  const int NQ = K <= 4 ? 1 : K <= 8 ? 2 : 4;  // the instance launch_fused picks
  return ((size_t)FUSED_ROWS * (DP + 2) + (size_t)K * PS + (size_t)4 * NQ * FUSED_QS + 4 * K + 4 + FUSED_CT_CAP) * sizeof(double);
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
  static LdsGrant grants[9];
  auto go = [&](auto kern, LdsGrant& g) {
    if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, g); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)a.grid), dim3(256), shmem, stream, a);
    return hipGetLastError();
  };
  if (!a.rginfo) {
    if (a.K <= 4) return go(fused_small_kernel<16, 1, 0>, grants[3]);
    if (a.K <= 8) return go(fused_small_kernel<16, 2, 0>, grants[4]);
    return go(fused_small_kernel<16, 4, 0>, grants[5]);
  }
  if ((int64_t)a.ngroups * a.K <= FUSED_CT_CAP) {
    if (a.K <= 4) return go(fused_small_kernel<16, 1, 1>, grants[6]);
    if (a.K <= 8) return go(fused_small_kernel<16, 2, 1>, grants[7]);
    return go(fused_small_kernel<16, 4, 1>, grants[8]);
  }
  if (a.K <= 4) return go(fused_small_kernel<16, 1, 2>, grants[0]);
  if (a.K <= 8) return go(fused_small_kernel<16, 2, 2>, grants[1]);
  return go(fused_small_kernel<16, 4, 2>, grants[2]);
}

}  // namespace lck
