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
// GRP 0: one group, known at compile time (no row-group table; as a run-time choice the table read and the computed
// `info` end in one register, which the compiler guards with `s_waitcnt vmcnt(0)` in BOTH paths -- full drains of the
// just-issued prefetch at the head of every tile); 1: several groups, the J x K table c_jk waits in LDS (J x K <=
// FUSED_CT_CAP); 2: several groups, the table is read from global memory inside the cluster loop (each read then waits
// for the prefetch as well).  Compile-time because a run-time choice between an LDS read and a global read ends in a
// combined `vmcnt(0) lgkmcnt(0)` at the merge.
// WANT_LL: also the split-ordering data term LL_k (cluster()'s rounds; the VBEM iterations run the plain instance, whose
// normalisation takes one exponential per entry and has no per-cluster wave reductions).
//
// Round 4: the E-step half is ONE instruction stream over all clusters.  The round-3 loop issued a cluster's seven LDS
// reads at its head and waited for them (an exposed LDS round trip per cluster), and ended every cluster in four
// exec-masked blocks `if (hi == r) { read c_k; wait; fma }` (four more): ~ 4 000 of a wave's 25 000 cycles per tile.
// Now (i) the parameter ring runs SEVEN reads ahead ACROSS cluster boundaries (NREAD = 14 = 2 x 7: the ring slots line
// up from cluster to cluster), fenced as in estep_kernel; (ii) a cluster's tail -- the squares of its last tile row, the
// sum over the four `hi` lanes and the choice of the lane's own row group -- is done under the NEXT cluster's MFMAs:
// four chained MFMAs  t = sum_r A_r d2_r + c  with selector operands A_r[i][.] = -1/2 [i == r]  leave
// log q~ = c - d^2 / 2 of row group `hi` in lane (lo4, hi) directly: no select, no branch, no VALU; (iii) c_k is read at
// the head of the cluster it belongs to and rides into the chain as its C operand.
// NTA: tile rows of the whitener / 4-column blocks of X that are not identically zero: 4, 2 for D <= 8, 1 for D <= 4 (the layout is
// the DP = 16 one either way -- `Xcat`'s two columns are padded eightfold -- but the tiles of rows and columns 8 ... 15 are
// zeros there: the E-step half then walks 5 of a cluster's 14 reads (20 + 4 MFMAs instead of 40 + 4) and the statistics
// half 5 of its 12 feature tiles; what is left out is written as zeros).
template <int DP, int CPW, int GRP, bool WANT_LL, int NTA = 4>
__global__ void __launch_bounds__(256, CPW == 4 ? 1 : 2) fused_small_kernel(FusedLaunch a_) {
  constexpr bool ONEGRP = GRP == 0, CTLDS = GRP != 2;
  static_assert(NTA == 1 || NTA == 2 || NTA == 4, "active tile rows");
  FusedLaunch a = a_;
  if constexpr (ONEGRP) a.rginfo = nullptr;
  static_assert(DP == 16, "one 16 x 16 block of S_k (the general blocking lives in suffstat_kernel)");
  constexpr int NT = DP / 4;
  constexpr int NTILES = NT * (NT + 1) / 2;
  constexpr int NREAD = NTA * (NTA + 1) / 2 + NTA;  // reads of the first NTA tile rows: a prefix of the cluster's stream
  constexpr int PF = NTA == 4 ? 7 : NTA == 2 ? 5 : 2;
  static_assert(NREAD % PF == 0, "the ring's slots must line up from one cluster to the next");
  constexpr int PS = NTILES * 16 + DP;
  // row stride of the staged tile (36 dwords).  E-step half: a half-wave reads rows lo4 = 0..15 at two columns -- 36 lo4
  // mod 64 are sixteen different multiples of 4: conflict-free.  Statistics half (feature form): a half-wave reads up to
  // sixteen columns of two rows that lie 8 rows apart (288 dwords = 32 banks: disjoint), and q[row][cluster] with
  // consecutive clusters 4 banks apart (FUSED_QS): conflict-free as well.  Column DP of every row holds 1.0 (s_k, N_k).
  constexpr int LD = DP + 2;
  constexpr int R = 4;
  static_assert(4 * CPW <= FUSED_KMAX, "statistics accumulators");
  constexpr int NQ = CPW, NTL = 12;    // cluster quads; feature tiles: 10 patches (ia <= ja), s_k, N_k
  // active feature tiles: the patches with ja < NTA (the first NTA (NTA + 1) / 2 of the enumeration), then s_k and N_k
  // N_k: with LL_k the twelfth feature tile (1 * 1); in the plain instance the sweep adds every responsibility it forms
  // to a per-lane sum instead (one add per entry in a phase that is VALU anyway, against two MFMAs per 4-row step: 8 % of
  // the statistics half at D = 16, 20 % at D <= 8), folded once at the end of the kernel
  constexpr bool NK_MFMA = WANT_LL;
  constexpr int NPA = NTA * (NTA + 1) / 2, NTLA = NPA + 1 + (NK_MFMA ? 1 : 0);
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
  double* etab = ctl + FUSED_CT_CAP;   // [64]: 2^(j / 64) for exp_nonpos
  const int ctrows = ONEGRP ? 1 : a.ngroups;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: its arithmetic belongs on the scalar unit)
  const int lo4 = lane & 15, hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  double acc[NTLA][NQ];  // [patches | s_k | N_k]
#pragma unroll
  for (int t = 0; t < NTLA; ++t)
#pragma unroll
    for (int c = 0; c < NQ; ++c) acc[t][c] = 0.0;
  // selector operands of the lane-sum chain: A_r[i = lo2][k = hi] = -1/2 [lo2 == r]
  double selA[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    selA[r] = lo2 == r ? -0.5 : 0.0;
    asm volatile("" : "+v"(selA[r]));  // (four registers for the whole kernel, not a compare + select per use)
  }
  double fz = 0.0;
  double nk[NK_MFMA ? 1 : 4 * NQ];  // this lane's share of N_k (plain instance)
#pragma unroll
  for (int i = 0; i < (NK_MFMA ? 1 : 4 * NQ); ++i) nk[i] = 0.0;
  const int64_t NP = a.nrg * RG;
  const int64_t ntile = (NP + FUSED_ROWS - 1) / FUSED_ROWS;
  // register double-buffer for the next tile of X (coalesced double2 pieces; rows past the end as zeros)
  constexpr int C2 = DP / 2, NPRE = FUSED_ROWS * C2 / 256;
  double2 pre[NPRE];
  auto fetch = [&](int64_t tile) {
    const int64_t r0 = tile * FUSED_ROWS;
    const int64_t left = tile < ntile ? NP - r0 : 0;
    const double2* X2 = reinterpret_cast<const double2*>(a.X) + (tile < ntile ? r0 : 0) * C2;
    if (left >= FUSED_ROWS) {  // (uniform) a whole tile: no per-load bounds
#pragma unroll
      for (int i = 0; i < NPRE; ++i) pre[i] = X2[tid + i * 256];
    } else {
      const int lim = (int)left * C2;
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int idx = tid + i * 256;
        pre[i] = idx < lim ? X2[idx] : make_double2(0.0, 0.0);
      }
    }
  };
  // Which tiles are this block's.  With two blocks per CU the SIMD's arbiter serves the wave that arrived first: the
  // first block of a CU walks a tile in ~ 13 us, the second in ~ 20 (profiles/r05_fused_timeline.log), and with equal
  // shares the first ones finish 20-30 us early and leave the others alone on a half-empty pipe.  Blocks are dispatched
  // in index order, the first grid / 2 one per CU: those take the larger share (a.yshare per mille of a CU's tiles go to
  // the second block) -- a fixed function of (blockIdx, grid, nrg), so every run sums the same rows in the same block,
  // and if the placement is ever different the only loss is the balance.
  int64_t tfirst = blockIdx.x, tstride = gridDim.x, tcount = tfirst < ntile ? (ntile - tfirst + tstride - 1) / tstride : 0;
  {
    const int64_t G = gridDim.x, H = G / 2;
    if (a.yshare > 0 && (G & 1) == 0 && ntile >= 4 * G) {
      const int64_t xy = ntile * 2 * a.yshare / (1000 * G);  // tiles of a second block
      const int64_t Te = ntile - H * xy;                     // tiles 0 .. Te - 1: dealt to the first blocks in turn
      tstride = H;
      if ((int64_t)blockIdx.x < H) {
        tfirst = blockIdx.x;
        tcount = tfirst < Te ? (Te - tfirst + H - 1) / H : 0;
      } else {
        tfirst = Te + ((int64_t)blockIdx.x - H);
        tcount = xy;
      }
    }
  }
  fetch(tcount > 0 ? tfirst : ntile);
  // the block's parameters: every load in flight at once, behind the first tile's (the round-4 loop waited for each of its
  // up to eleven loads in turn -- five L2 round trips in front of the first tile: 4 us of a 40 ... 135 us launch)
  constexpr int NPV = (4 * CPW * PS + 255) / 256;
  double pv[NPV];
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int idx = tid + i * 256;
    pv[i] = idx < K * PS ? a.params[idx] : 0.0;
  }
  for (int i = tid; i < 4 * K; i += 256) llw[i] = 0.0;
  if constexpr (CTLDS)
    for (int i = tid; i < ctrows * K; i += 256) ctl[i] = a.ctab[i];
  for (int i = tid; i < 4 * NQ * QS; i += 256) qt[i] = 0.0;
  xt[tid * LD + ONE] = 1.0;  // (the staging below writes columns 0 .. DP - 1 only)
  fill_exp_table(etab, tid, 256);
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    const int idx = tid + i * 256;
    if (idx < K * PS) par[idx] = pv[i];
  }
  __syncthreads();
  // n-th read of cluster `kk`'s parameter stream relative to the running pointers of the current cluster
  const double* Pt = par + (lane & 3) + 4 * hi;  // this lane's element of every 4x4 tile (cluster 0)
  const double* Pb = par + NTILES * 16 + hi;     // this lane's element of every 4-vector of -b (cluster 0)
  double* const xstage = xt + (tid / C2) * LD + 2 * (tid % C2);
  for (int64_t ti = 0; ti < tcount; ++ti) {
    const int64_t tile = tfirst + ti * tstride;
    const int64_t row0 = tile * FUSED_ROWS;
    // ---- the tile of X -> LDS (its loads were issued a whole tile ago)
    // (one 16-byte store per piece: its 8-lane groups fill one row's 128 bytes; two 8-byte stores put two rows, 36 dwords
    //  apart, into a 16-lane group: a 2-way conflict on the 32 write banks)
#pragma unroll
    for (int i = 0; i < NPRE; ++i) *reinterpret_cast<double2*>(xstage + i * (256 / C2) * LD) = pre[i];
    __syncthreads();
    // ---- E-step half: this wave's 64 rows as four row groups; lane (lo4, hi) owns row 16 hi + lo4 = row `tid` of the tile
    // (a table entry is read BEFORE the next tile's prefetch goes out: it is waited for at once, and the vector-memory
    //  counter retires in order)
    const int64_t left = NP - row0;  // padded rows from here on (a multiple of 16)
    const bool myok = tid < left;    // this lane's row group exists
    bool myrow;
    int mygrp = 0;
    if constexpr (ONEGRP) {
      myrow = row0 + tid < a.nrows;
    } else {
      int info = 0;
      if (myok) {
        const int64_t rg = row0 / RG + (tid >> 4);
        if (a.rginfo) {
          info = a.rginfo[rg];
        } else {
          const int64_t rem = a.nrows - rg * RG;
          info = rem >= RG ? RG : (rem > 0 ? (int)rem : 0);
        }
      }
      asm volatile("" : "+v"(info));  // (the table entry has to be HERE: its wait stands in front of the prefetch)
      mygrp = info >> 5;
      myrow = lo4 < (info & 31);
      __builtin_amdgcn_sched_barrier(0);
    }
    fetch(ti + 1 < tcount ? tile + tstride : ntile);  // in flight during both halves of this tile
    double xf[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double* xr = xt + (wave * 64 + r * 16 + lo4) * LD + hi;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) xf[r][jt] = jt < NTA ? xr[4 * jt] : 0.0;
    }
    // the constant of cluster k for this lane's row group
    auto cjk_of = [&](int k) -> double {
      if constexpr (ONEGRP) return ctl[k];
      else if constexpr (CTLDS) return ctl[mygrp * K + k];
      else return a.ctab[(int64_t)mygrp * K + k];
    };
    double ring[PF];
    static_for<PF>([&](auto ic) {
      constexpr RdInfo ri = rd_info(ic);
      ring[ic] = ri.jt < 0 ? Pb[ri.off] : Pt[ri.off];
    });
    // carried from cluster k - 1 into cluster k's pass: its last tile row (not yet squared), its partial squared norms
    // and its constant (-inf before the first cluster: that pass's "log q~" is -inf and changes nothing)
    double accP[R], d2P[R], cP = -INFINITY, mymx = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) accP[r] = 0.0, d2P[r] = 0.0;
    for (int k = 0; k < K; ++k) {
      const double* Ptk = Pt + (size_t)k * PS;
      const double* Pbk = Pb + (size_t)k * PS;
      const double cK = cjk_of(k);
      double* const qslot = qt + (size_t)(k > 0 ? k - 1 : 0) * QS + tid;  // (k = 0 writes -inf where cluster 0 lands next)
      double d2[R], accs[2][R], t = 0.0;
      if constexpr (NTA == 1) {  // (no tile row is squared inside the pass: the only one waits for the next cluster)
#pragma unroll
        for (int r = 0; r < R; ++r) d2[r] = 0.0;
      }
      static_for<NREAD>([&](auto nc) {
        constexpr int n = nc;
        constexpr RdInfo ri = rd_info(n);
        constexpr int set = ri.it & 1;
        const double v = ring[n % PF];
        {  // the read PF ahead: of this cluster, or already of the next one
          constexpr int m = n + PF;
          constexpr RdInfo rn = rd_info(m < NREAD ? m : m - NREAD);
          constexpr int over = m < NREAD ? 0 : PS;
          ring[n % PF] = rn.jt < 0 ? Pbk[rn.off + over] : Ptk[rn.off + over];
        }
        if constexpr (ri.jt < 0) {
#pragma unroll
          for (int r = 0; r < R; ++r) accs[set][r] = v;  // y starts at -b: y = A x - b
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) accs[set][r] = mfma4(v, xf[r][ri.jt], accs[set][r]);
        }
        // under those MFMAs: the previous cluster's tail and this cluster's deferred squares
        if constexpr (n == 1) {
#pragma unroll
          for (int r = 0; r < R; ++r) d2P[r] = fma(accP[r], accP[r], d2P[r]);
        }
        // the four links of the previous cluster's lane-sum chain (NTA = 4: behind reads 3, 4, 6, 7; NTA = 2: 2, 3, 4, 4;
        // NTA = 1: all behind read 1, the only tile)
        constexpr int L0 = NTA == 4 ? 3 : NTA == 2 ? 2 : 1, L1 = NTA == 4 ? 4 : NTA == 2 ? 3 : 1, L2 = NTA == 4 ? 6 : NTA == 2 ? 4 : 1,
                      L3 = NTA == 4 ? 7 : NTA == 2 ? 4 : 1;
        if constexpr (n == L0) t = mfma4(selA[0], d2P[0], cP);
        if constexpr (n == L1) t = mfma4(selA[1], d2P[1], t);
        if constexpr (n == L2) t = mfma4(selA[2], d2P[2], t);
        if constexpr (n == L3) t = mfma4(selA[3], d2P[3], t);
        // tile row it - 1 is squared behind tile (it, 1), under the MFMAs of row it (the last row waits for the next cluster)
        if constexpr (ri.jt == 1 && ri.it >= 1) {
          constexpr int pset = (ri.it - 1) & 1;
#pragma unroll
          for (int r = 0; r < R; ++r)
            d2[r] = ri.it == 1 ? accs[pset][r] * accs[pset][r] : fma(accs[pset][r], accs[pset][r], d2[r]);
        }
        if constexpr (n == NREAD - (NTA == 4 ? 2 : 1)) {  // log q~ of cluster k - 1 for this lane's row: its own slot (no barrier before it reads it back)
          mymx = fmax(mymx, t);  // (fmax, not the asm max_raw: t comes straight out of an MFMA, and the compiler only counts the
                                 //  wait states between an MFMA and its reader for instructions it emitted itself)
          *qslot = t;
        }
        // within a step: its VALU instructions as ONE group ahead of the ring read and the MFMAs (hipcc otherwise splits a row's
        // four squares around the MFMAs; next to the matrix pipe VALU work is paid per switch: tools/mfma_batch_probe.hip).
        // K = 16, N = 4M: 1.092 -> 1.06 ms; level at K <= 8 (gpurun_out/r05i)
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int r = 0; r < R; ++r) accP[r] = accs[(NTA - 1) & 1][r], d2P[r] = d2[r];  // the last tile row is squared under the next cluster
      cP = cK;
    }
    {  // the last cluster's tail
#pragma unroll
      for (int r = 0; r < R; ++r) d2P[r] = fma(accP[r], accP[r], d2P[r]);
      double t = mfma4(selA[0], d2P[0], cP);
#pragma unroll
      for (int r = 1; r < R; ++r) t = mfma4(selA[r], d2P[r], t);
      mymx = fmax(mymx, t);
      qt[(size_t)(K - 1) * QS + tid] = t;
    }
    // logsumexp and normalisation in the reference's order: max, sum exp(x - max), log + max, exp(x - logZ)
    {
      double* const ql = qt + tid;
      double* const qp = a.qZ + row0 + tid;
      const bool live = myok && myrow;
      if constexpr (!WANT_LL) {
        // ONE exponential per entry -- e = exp(log q~ - max) stays in registers and q = e / sum(e); the same sum and
        // logZ, q within a few ulp of exp(log q~ - logZ).  Four entries at a time: four independent polynomial chains
        // (the table exponential and the Newton reciprocal of lc_device.hpp: this sweep is bound by fp64 issue).
        double e[4 * NQ], s = 0.0;
        static_for<NQ>([&](auto cc) {
          constexpr int c = cc;
          if (4 * c < K) {  // (uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int kk = 4 * c + j;
              const double x = exp_nonpos(ql[kk * QS] - mymx, etab);
              e[kk] = kk < K ? x : 0.0;  // (slots of clusters >= K hold zeros, not log q~)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s += e[4 * c + j];
          }
        });
        const double inv = rcp_pos(s);
        static_for<NQ>([&](auto cc) {
          constexpr int c = cc;
          if (4 * c < K) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int kk = 4 * c + j;
              const double q = live ? e[kk] * inv : 0.0;
              if constexpr (!NK_MFMA) nk[kk] += q;  // (slots of clusters >= K add zeros)
              if (kk < K) {
                if (myok) qp[(int64_t)kk * a.ldq] = q;
                ql[kk * QS] = q;
              }
            }
          }
        });
        if (live) fz += log(s) + mymx;
      } else {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += exp(ql[k * QS] - mymx);
        const double logZ = log(s) + mymx;
        for (int k = 0; k < K; ++k) {
          const double lq = ql[k * QS];
          double q = exp(lq - logZ);
          if (!live) q = 0.0;
          if (myok) qp[(int64_t)k * a.ldq] = q;
          ql[k * QS] = q;
          const double ll = wave_sum(q > 0.0 ? q * (lq - cjk_of(k)) : 0.0);
          if (lane == 0) llw[wave * K + k] += ll;
        }
        if (live) fz += logZ;
      }
    }
    __syncthreads();

    // ---- statistics half: this wave's 16 of the tile's 64 four-row steps, all feature tiles, all cluster quads.
    // Tile t < 10 is the patch (ia, ja): x[row][4 ia + lo2] * x[row][4 ja + blk] -- two lane-dependent base pointers and
    // compile-time column offsets; t = 10: x[row][4 blk + lo2] * 1 (s_k); t = 11: 1 * 1 (N_k).
    {
      // a step's four rows are {t, t + 8, t + 4, t + 12} of a 16-row block: the two rows of a half-wave then lie 32 banks
      // apart (8 x 36 dwords), which every fragment of the step needs -- the 8-bank patch operands and the 32-bank s_k
      // tile alike -- and in the q table 16 dwords apart, between the 4-dword steps of consecutive clusters (FUSED_QS).
      // Step s4 of wave w is row block s4, rows w + {0, 8, 4, 12}: every address is a per-lane base + a compile-time offset.
      const int rsub = 8 * (hi & 1) + 4 * (hi >> 1) + wave;
      const double* xu = xt + rsub * LD + lo2;        // + 4 ia
      const double* xw = xt + rsub * LD + blk;        // + 4 ja
      const double* xs = xt + rsub * LD + 4 * blk + lo2;
      const double* x1 = xt + rsub * LD + ONE;
      const double* qb = qt + lo2 * QS + rsub;
      // software pipeline: the operands of step s4 + 1 (q of the cluster quads, the four x fragments of either role, the
      // s_k fragment and the ones) are read while step s4's products and MFMAs issue -- left to itself hipcc reads them
      // right in front of their use and every step starts with an exposed LDS round trip
      struct StepOps {
        double qa[NQ], u[NTA], w[NTA], s, one = 0.0;
      };
      auto load = [&](auto sc, StepOps& o) {
        constexpr int s4 = decltype(sc)::value, ro = s4 * 16 * LD;
#pragma unroll
        for (int c = 0; c < NQ; ++c) o.qa[c] = qb[4 * c * QS + s4 * 16];
#pragma unroll
        for (int i = 0; i < NTA; ++i) o.u[i] = xu[ro + 4 * i], o.w[i] = xw[ro + 4 * i];
        o.s = xs[ro];
        if constexpr (NK_MFMA) o.one = x1[ro];
      };
      StepOps cur;
      load(std::integral_constant<int, 0>{}, cur);
      static_for<FUSED_ROWS / 16>([&](auto sc) {
        constexpr int s4 = sc;
        StepOps nxt = cur;
        if constexpr (s4 + 1 < FUSED_ROWS / 16) load(std::integral_constant<int, s4 + 1>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
        // ALL of the step's products first, then all of its MFMAs: next to the matrix pipe a VALU instruction is paid per
        // switch between the two kinds, not per instruction (tools/mfma_batch_probe.hip: ~ 12 clocks each when they stand
        // alone between MFMAs, ~ 5 in a group of eight)
        double p[NTLA];
        static_for<NTLA>([&](auto tc) {
          constexpr int t = tc;
          if constexpr (t < NPA) {
            constexpr int ja = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3, ia = t - ja * (ja + 1) / 2;
            p[t] = cur.u[ia] * cur.w[ja];
          } else if constexpr (t == NPA) {
            p[t] = cur.s;
          } else {
            p[t] = cur.one;
          }
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<NTLA>([&](auto tc) {
          constexpr int t = tc;
#pragma unroll
          for (int c = 0; c < NQ; ++c) acc[t][c] = mfma4(cur.qa[c], p[t], acc[t][c]);
        });
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      });
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
            // record tile t: patch t (t < 10), s_k (10), N_k (11) <- accumulator slot; the tiles left out are zeros
            const int slot = t < NPA ? t : t == 10 ? NPA : t == 11 && NK_MFMA ? NPA + 1 : -1;
            const double v = slot >= 0 ? acc[slot >= 0 ? slot : 0][c] : 0.0;
            double* r = red + (t * NQ + c) * 64 + lane;
            *r = w == 0 ? v : *r + v;
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
      } else if (NK_MFMA && b == 0 && lo == 0) {
        out[0] = v;
      }
    }
  }
  if constexpr (!NK_MFMA) {  // N_k: lanes (fixed butterfly), then the four waves in wave order (llw is unused without LL_k)
#pragma unroll
    for (int kk = 0; kk < 4 * NQ; ++kk) {
      const double v = wave_sum(nk[kk]);
      if (lane == 0 && kk < K) llw[wave * K + kk] = v;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) rec[(int64_t)k * SS] = ((llw[k] + llw[K + k]) + llw[2 * K + k]) + llw[3 * K + k];
    __syncthreads();
  }
  fz = wave_sum(fz);
  if (lane == 0) fzw[wave] = fz;
  __syncthreads();
  for (int k = tid; k < K; k += 256)
    rec[K * SS + 1 + k] = WANT_LL ? llw[k] + llw[K + k] + llw[2 * K + k] + llw[3 * K + k] : 0.0;
  if (tid == 0) rec[K * SS] = -(fzw[0] + fzw[1] + fzw[2] + fzw[3]);  // cluster.cpp:137 returns -sum(logZ)
}

static size_t fused_lds_bytes(int DP, int K) {
  const int NT = DP / 4, PS = NT * (NT + 1) / 2 * 16 + DP;
  const int NQ = K <= 4 ? 1 : K <= 8 ? 2 : 4;  // the instance launch_fused picks
  return ((size_t)FUSED_ROWS * (DP + 2) + (size_t)K * PS + (size_t)4 * NQ * FUSED_QS + 4 * K + 4 + FUSED_CT_CAP + 64) * sizeof(double);
}

// does this shape have a fused path?  (a property of (DP, K) alone: every rank of a distributed run must take the same
// branch whatever its share of the rows, an empty share included)
bool fused_eligible(int DP, int K) {
  static const bool off = test_switch("LC_FUSED_SMALL") && atoi(test_switch("LC_FUSED_SMALL")) == 0;  // (tests: the two separate kernels)
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
  const int cus = current_device_cus();
  const int64_t ntile = (nrg * RG + FUSED_ROWS - 1) / FUSED_ROWS;
  const size_t lds = fused_lds_bytes(DP, K);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  return (int)std::min<int64_t>(ntile, (int64_t)cus * per_cu);
}

template <int CPW, int GRP>
static hipError_t launch_fused_t(const FusedLaunch& a, hipStream_t stream, size_t shmem) {
  auto go = [&](auto kern, LdsGrant& g) {
    if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, g); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)a.grid), dim3(256), shmem, stream, a);
    return hipGetLastError();
  };
  static LdsGrant grants[6];
  static const bool full_only = test_switch("LC_FUSED_FULL") != nullptr;  // (tests: the full-width instance at every D)
  if (a.D <= 4 && !full_only)  // (columns 4 ... 15 are zeros: one tile of the whitener, one patch of S_k -- the reference's own test data is D = 2)
    return a.want_ll ? go(fused_small_kernel<16, CPW, GRP, true, 1>, grants[5]) : go(fused_small_kernel<16, CPW, GRP, false, 1>, grants[4]);
  if (a.D <= 8 && !full_only)  // (columns 8 ... 15 of the padded layout are zeros: the half-width instance)
    return a.want_ll ? go(fused_small_kernel<16, CPW, GRP, true, 2>, grants[3]) : go(fused_small_kernel<16, CPW, GRP, false, 2>, grants[2]);
  return a.want_ll ? go(fused_small_kernel<16, CPW, GRP, true>, grants[1]) : go(fused_small_kernel<16, CPW, GRP, false>, grants[0]);
}

hipError_t launch_fused(const FusedLaunch& a_, hipStream_t stream) {
  FusedLaunch a = a_;
  if (a.DP != 16 || a.grid <= 0) return hipErrorInvalidValue;
  {
    // two blocks per CU on every CU (fused_plan): the second block of a CU takes 41 % of its tiles (measured pace ratio
    // 13 : 20 us per tile at K = 8; same-box A/B in profiles/r05_fused_variants.log)
    static const int ys = test_switch("LC_FUSED_YSHARE") ? atoi(test_switch("LC_FUSED_YSHARE")) : 410;
    const int cus = current_device_cus();
    a.yshare = a.grid == 2 * cus ? ys : 0;
  }
  const size_t shmem = fused_lds_bytes(a.DP, a.K);
  if (!a.rginfo) {
    if (a.K <= 4) return launch_fused_t<1, 0>(a, stream, shmem);
    if (a.K <= 8) return launch_fused_t<2, 0>(a, stream, shmem);
    return launch_fused_t<4, 0>(a, stream, shmem);
  }
  if ((int64_t)a.ngroups * a.K <= FUSED_CT_CAP) {
    if (a.K <= 4) return launch_fused_t<1, 1>(a, stream, shmem);
    if (a.K <= 8) return launch_fused_t<2, 1>(a, stream, shmem);
    return launch_fused_t<4, 1>(a, stream, shmem);
  }
  if (a.K <= 4) return launch_fused_t<1, 2>(a, stream, shmem);
  if (a.K <= 8) return launch_fused_t<2, 2>(a, stream, shmem);
  return launch_fused_t<4, 2>(a, stream, shmem);
}

}  // namespace lck
