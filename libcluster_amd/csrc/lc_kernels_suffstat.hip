// Sufficient statistics of the Gauss-Wishart clusters: updateSS (src/cluster.cpp:53-82) -> GaussWish::addobs (src/distributions.cpp:301-313)
// (one translation unit per kernel family; the file header of lc_kernels_estep.hip maps kernels to the reference)
#include "lc_device.hpp"

// widths at which the feature-GEMM form of the statistics pass is the default for more than 16 clusters (measured:
// tools/ssfeat_check.py; LC_SS_FEAT=2 selects it wherever it exists, 0 nowhere)
#ifndef LC_SS_FEAT_MINK_ACTIVE
#define LC_SS_FEAT_MINK_ACTIVE 5
#endif
#ifndef LC_SS_FEAT_WIDTHS
#define LC_SS_FEAT_WIDTHS(DP) ((DP) >= 32 && (DP) <= 128)  // (every padded width: 80, 96, 112 fit their registers as 8-wave blocks, round 5)
#endif

namespace lck {

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
// HALF: 0 = the whole tile triangle in one launch; 1 / 2 = the first / second half of the 16x16 block pairs (balanced by
// MFMA count).  At D = 128 the 136 accumulators per cluster of the full triangle leave room for ONE wave per SIMD, and
// a lone wave cannot hide its own LDS latency; two half launches (68 / 69 accumulators, both re-read X, which the
// MFMA-bound kernel can afford) run two waves per SIMD with the simple step loop.  N_k and s_k ride with half 1.
template <int NB>
__host__ __device__ constexpr bool ss_in_half(int half, int jbp, int jb) {
  if (half == 0) return true;
  const bool first = NB == 8   ? (jbp < 5 || (jbp == 5 && jb < 3))    // 67 | 69 MFMAs
                     : NB == 7 ? (jbp < 4 || (jbp == 4 && jb < 4))    // 52 | 53
                     : NB == 6 ? (jbp < 4 || (jbp == 4 && jb < 1))    // 40 | 38
                               : (jbp == 1 || jbp == 2);              // 18 | 18 (NB = 4)
  return first == (half == 1);
}
// PAN (observations wider than 128 columns, DP = 64 only): the statistics of a wide X are assembled from 64-column
// panels, one launch per panel pair (P >= Q).  1 = diagonal pair (P, P): the triangle loop of the narrow kernel on
// columns [colA, colA + 64) of rows with stride ldx; N_k rides with panel 0, s_k with every diagonal pair.
// 2 = off-diagonal pair: A operands (all four rotations) from panel colA, q x from panel colB, the full 16 x 4 set of
// MFMAs, block written to (P, Q) and mirrored to (Q, P).  Records are DPW wide.
template <int PAN>
constexpr int ss_batch_rows() { return PAN == 2 ? 24 : SS_BR; }  // (two panels per batch: 24 rows keep two blocks per CU)
template <int DP, int CPW, bool SKIP, int HALF, int PAN = 0, bool RSP = false>
__global__ void __launch_bounds__(256, (DP <= 64 ? (HALF != 0 ? 3 : 2) : (HALF != 0 || DP <= 80 ? 2 : 1)))
    suffstat_kernel(SuffstatLaunch a) {
  static_assert(PAN == 0 || (DP == 64 && HALF == 0), "panel variants are built on the D = 64 kernel");
  constexpr int NB = DP / 16;
  constexpr int NACC = PAN == 2 ? NB * NB * 4 : SSAcc<NB>::N;
  constexpr int BR = ss_batch_rows<PAN>();
  constexpr int LD = lds_row_stride(DP);  // padded LDS row stride (doubles)
  constexpr int NPB = PAN == 2 ? 2 : 1;  // column panels staged per batch
  constexpr bool SORD = DP == 64;        // rotation-major MFMA order in the step loop (see there)
  // step loop with the next step's unrotated fragments and q prefetched (see there); -2...4 % at every width but
  // 16 (+20 %: four clusters per wave on 10 MFMAs each leave nothing to hide the extra reads behind)
  constexpr bool PFETCH = DP >= 32 && (DP <= 80 || HALF != 0);
  constexpr int XBUF = NPB * BR * LD;    // doubles per X buffer
  constexpr int NV2 = BR * DP / 2;       // double2 elements per staged batch and panel
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int nwaves = 4, nthr = 256;  // always launched with 4 waves; surplus waves only help staging
  double* xbuf = lds;                              // [2][NPB][BR][LD]
  double* qbuf = lds + 2 * XBUF;                   // [2][nwaves][CPW][BR]
  const int64_t ldx = PAN ? a.ldx : DP;            // row stride of X
  const int DPW = PAN ? a.DPW : DP;                // record width
  const int colA = PAN ? a.colA : 0, colB = PAN == 2 ? a.colB : colA;
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
  // RSP (row split, launched for a last slice that fills only one or two waves): the idle waves share the rows of
  // the active ones -- wave w works on the clusters of wave w % f, 4-row steps st = rcls mod RS, and writes its own record
  int wslot = wave, rcls = 0;
  if constexpr (RSP) {
    slice += a.slice0;
    const int f = nwaves / a.rs;
    wslot = wave % f;
    rcls = wave / f;
  }
  const int KR = a.KR;                                // records per chunk (K, or K + extra row-split records)
  const int kbase = (slice * nwaves + wslot) * CPW;   // may be >= K: the wave still helps staging
  int nk = kbase >= K ? 0 : ((K - kbase) < CPW ? (K - kbase) : CPW);
  int64_t r0 = (int64_t)chunk * a.chunk_rows;
  int64_t r1 = (r0 + a.chunk_rows) < a.NP ? (r0 + a.chunk_rows) : a.NP;
  int kidx[CPW];      // cluster of accumulator set c
  int64_t recidx[CPW];  // its partial record
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    kidx[c] = kbase + c;
    recidx[c] = (int64_t)chunk * KR + kbase + c;
    if (RSP && rcls > 0) recidx[c] = (int64_t)chunk * KR + K + (rcls - 1) * a.nklast + (kbase + c - a.klast0);
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
  double pre[NPB][NPRE_MAX][2];
  double qpre[CPW];
  auto gload = [&](int64_t b0) {
#pragma unroll
    for (int pn = 0; pn < NPB; ++pn) {
#pragma unroll
      for (int i = 0; i < NPRE_MAX; ++i) {
        if (i < npre) {
          const int idx = tid + i * nthr;          // double2 index inside the batch: row-major [BR][DP/2]
          const int row = idx / (DP / 2), c2 = idx % (DP / 2);
          double2 v = make_double2(0.0, 0.0);
          if (idx < NV2 && b0 + row < r1)
            v = *reinterpret_cast<const double2*>(a.X + (b0 + row) * ldx + (pn == 0 ? colA : colB) + 2 * c2);
          pre[pn][i][0] = v.x;
          pre[pn][i][1] = v.y;
        }
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
    for (int pn = 0; pn < NPB; ++pn) {
#pragma unroll
      for (int i = 0; i < NPRE_MAX; ++i) {
        if (i < npre) {
          const int idx = tid + i * nthr;
          const int row = idx / (DP / 2), c2 = idx % (DP / 2);
          if (idx < NV2)
            *reinterpret_cast<double2*>(xb + pn * BR * LD + row * LD + 2 * c2) = make_double2(pre[pn][i][0], pre[pn][i][1]);
        }
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
      if constexpr (DP > 80 && HALF == 0) {
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
      } else if constexpr (PAN == 2) {
        const int nstep = (int)(((r1 - b0) < BR ? (r1 - b0) : BR) / 4);
        for (int st = 0; st < nstep; ++st) {
          if (SKIP) {
            bool any = false;
#pragma unroll
            for (int c = 0; c < CPW; ++c) any = any || qb[c * BR + st * 4] != 0.0;
            if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
          }
          // A side: all four rotations of panel colA; B side: the unrotated fragments of panel colB
          // (read in the order of use: the B side first, then the A side block row by block row, so that the
          // MFMAs of block row 0 start on a partial lgkmcnt wait)
          double xr[NB][4], xq[NB];
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) xq[jb] = xb[BR * LD + st * 4 * LD + 16 * jb + 4 * blk];
#pragma unroll
          for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) xr[jb][s2] = xb[st * 4 * LD + 16 * jb + 4 * ((blk + s2) & 3)];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const double q = qb[c * BR + st * 4];
            if (SKIP && __builtin_amdgcn_ballot_w64(q != 0.0) == 0) continue;
            double qx[NB];
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) qx[jb] = q * xq[jb];
#pragma unroll
            for (int jbp = 0; jbp < NB; ++jbp)
#pragma unroll
              for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                  acc[c][(jbp * NB + jb) * 4 + s2] = mfma4(xr[jbp][s2], qx[jb], acc[c][(jbp * NB + jb) * 4 + s2]);
          }
        }
      } else if constexpr (RSP) {
        const int nstep = (int)(((r1 - b0) < BR ? (r1 - b0) : BR) / 4);
        for (int st = rcls; st < nstep; st += a.rs) {
          double xr[NB][4];
#pragma unroll
          for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) xr[jb][s2] = xb[st * 4 * LD + 16 * jb + 4 * ((blk + s2) & 3)];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const double q = qb[c * BR + st * 4];
            double qx[NB];
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
              qx[jb] = q * xr[jb][0];
              if (HALF != 2) sacc[c][jb] += qx[jb];
            }
            if (HALF != 2) nacc[c] += q;
            int idx = 0;
#pragma unroll
            for (int jbp = 0; jbp < NB; ++jbp)
#pragma unroll
              for (int jb = 0; jb <= jbp; ++jb)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                  if (s2 < 3 || jb < jbp) {
                    if (ss_in_half<NB>(HALF, jbp, jb)) acc[c][idx] = mfma4(xr[jbp][s2], qx[jb], acc[c][idx]);
                    ++idx;
                  }
          }
        }
      } else if constexpr (PFETCH && !SKIP) {
        // Dense: the unrotated fragments and q of step st + 1 are fetched under the MFMAs of step st (12 registers at
        // D = 64), so a step starts its rotation-0 MFMAs at once and its own rotated fragments arrive under them.
        // All BR/4 steps run (rows past the chunk end were staged as zeros with q = 0).
        double x0[NB], qv[CPW];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) x0[jb] = xb[16 * jb + 4 * blk];
#pragma unroll
        for (int c = 0; c < CPW; ++c) qv[c] = qb[c * BR];
#pragma unroll 1
        for (int st = 0; st < BR / 4; ++st) {
          double xr[NB][4], x0n[NB], qn[CPW];
#pragma unroll
          for (int s2 = 1; s2 < 4; ++s2)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) xr[jb][s2] = xb[st * 4 * LD + 16 * jb + 4 * ((blk + s2) & 3)];
          const int sn = st + 1 < BR / 4 ? st + 1 : st;  // (the last step re-reads its own operands; not used)
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) {
            xr[jb][0] = x0[jb];
            x0n[jb] = xb[sn * 4 * LD + 16 * jb + 4 * blk];
          }
#pragma unroll
          for (int c = 0; c < CPW; ++c) qn[c] = qb[c * BR + sn * 4];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            double qx[NB];
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
              qx[jb] = qv[c] * xr[jb][0];
              if (HALF != 2) sacc[c][jb] += qx[jb];
            }
            if (HALF != 2) nacc[c] += qv[c];
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
              int idx = 0;
#pragma unroll
              for (int jbp = 0; jbp < NB; ++jbp)
#pragma unroll
                for (int jb = 0; jb <= jbp; ++jb)
#pragma unroll
                  for (int s2 = 0; s2 < 4; ++s2)
                    if (s2 < 3 || jb < jbp) {
                      if (s2 == sp && ss_in_half<NB>(HALF, jbp, jb)) acc[c][idx] = mfma4(xr[jbp][s2], qx[jb], acc[c][idx]);
                      ++idx;
                    }
            }
          }
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) x0[jb] = x0n[jb];
#pragma unroll
          for (int c = 0; c < CPW; ++c) qv[c] = qn[c];
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
          for (int t = 0; t < 4 * NB; ++t) {
            const int jb = SORD ? t % NB : t / 4, s = SORD ? t / NB : t % 4;
            xr[jb][s] = xb[st * 4 * LD + 16 * jb + 4 * ((blk + s) & 3)];
          }
  #pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const double q = qb[c * BR + st * 4];
            if (SKIP && __builtin_amdgcn_ballot_w64(q != 0.0) == 0) continue;
            double qx[NB];
  #pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
              qx[jb] = q * xr[jb][0];
              if (HALF != 2) sacc[c][jb] += qx[jb];
            }
            if (HALF != 2) nacc[c] += q;
            // D = 64: rotation-major order -- the unrotated fragments (the first reads of the step) are consumed
            // first, so the compiler can start the MFMAs on partial lgkmcnt waits while the rotated fragments are
            // still in flight (23.5 -> 22.9 ms at the north-star shape; the other widths are 2-5 % slower that way)
  #pragma unroll
            for (int sp = 0; sp < (SORD ? 4 : 1); ++sp) {
              int idx = 0;
  #pragma unroll
              for (int jbp = 0; jbp < NB; ++jbp) {
  #pragma unroll
                for (int jb = 0; jb <= jbp; ++jb) {
  #pragma unroll
                  for (int s = 0; s < 4; ++s) {
                    if (s < 3 || jb < jbp) {
                      if ((!SORD || s == sp) && ss_in_half<NB>(HALF, jbp, jb))
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
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
  if (nk == 0) return;

  const int64_t SS = 1 + (int64_t)DPW + (int64_t)DPW * DPW;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (c < nk) {
      double* out = a.partial + recidx[c] * SS;
      if (HALF != 2 && PAN != 2) {
        if (PAN == 0 || colA == 0) {
          const double nsum = sum_over_hi(nacc[c]);
          if (lane == 0) out[0] = nsum;
        }
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          const double s = sum_over_hi(sacc[c][jb]);
          if (hi == 0) out[1 + colA + 16 * jb + lo4] = s;
        }
      }
      double* S = out + 1 + DPW;
      if constexpr (PAN == 2) {
#pragma unroll
        for (int jbp = 0; jbp < NB; ++jbp)
#pragma unroll
          for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int gi = colA + 16 * jbp + 4 * ((blk + s) & 3) + hi, gj = colB + 16 * jb + 4 * blk + lo2;
              const double v = acc[c][(jbp * NB + jb) * 4 + s];
              S[(int64_t)gi * DPW + gj] = v;
              S[(int64_t)gj * DPW + gi] = v;
            }
      } else {
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
              const int gi = colA + 16 * jbp + 4 * ti + hi, gj = colA + 16 * jb + 4 * tj + lo2;
              const bool diag = jb == jbp;
              // diagonal 16x16 blocks: s=0 gives the diagonal tiles, s=1 every pair {t,t+1 mod 4}
              // once, s=2 the pairs {0,2},{1,3} twice (keep the lower copy); s=3 is never issued
              const bool wr = (!diag || ti == tj || s == 1 || ti > tj) && ss_in_half<NB>(HALF, jbp, jb);
              const double v = acc[c][idx];
              if (wr) {
                S[(int64_t)gi * DPW + gj] = v;
                if (!diag || ti != tj) S[(int64_t)gj * DPW + gi] = v;
              }
              ++idx;
            }
          }
        }
      }
      }
    }
  }
}

// ===========================================================================
// Sufficient statistics as a feature GEMM (32 <= D <= 128, more than 16 clusters, dense)
// ===========================================================================
// suffstat_kernel forms q_k x per cluster: 9 fp64 VALU instructions per 36 MFMAs, and an fp64 VALU instruction inside a
// stream of fp64 MFMAs costs the matrix pipe about 11 clocks (tools/mfma_mix_probe.hip) -- its ceiling is ~81 % of the
// pipe, and it sits there.  Here the cluster index moves into the MFMA:  T = Q^T Phi(x), Phi(x) = [x_i x_j (i <= j) | x_i | 1]:
//   A operand = q[row 4 st + hi][cluster k0 + 4 c + lo2]     (one LDS read per cluster quad and step, all tiles share it)
//   B operand = x[row][4 ia + lo2] * x[row][4 ja + blk]      (ONE multiply per 16 features, shared by ALL cluster quads)
//   D[i = hi][j = lo2] of block blk = S_{k0 + 4c + hi}[4 ia + lo2][4 ja + blk]
// i.e. one v_mul_f64 per NQ MFMAs (NQ = cluster quads per launch, up to 8: 1 : 8 against 1 : 4) and nothing else on the
// VALU: s_k and N_k are the features x_i * 1 and 1 * 1 (a column of ones rides in the staged X rows).  A "tile" is 16
// features: the 4 x 4 patches (ia <= ja) of the symmetric matrix, DP / 16 tiles of s_k, one of N_k (141 at D = 64); a
// wave owns up to 9 tiles x NQ quads (<= 72 accumulators), nslice blocks of four waves cover a row chunk (as the cluster
// slices of suffstat_kernel do).  Every tile reads its two operand fragments itself (LDS broadcast reads cost the
// matrix pipe nothing; no operand sharing to schedule).  More than 32 clusters: one launch per range of <= 32.
// Same partial records, same reduction, deterministic; the patch on the diagonal computes both halves of its 4 x 4
// block from commuted products, so S_k comes out exactly symmetric.
// (DC: the ACTIVE width, estep_active_width in lc_kernels.h -- the patches of the columns past it are products of zeros
//  and are not dealt out at all; the s_k tiles cover whole 16-column blocks, their idle columns read the staged zeros)
__host__ __device__ constexpr int ft_tiles(int DC) { return (DC / 4) * (DC / 4 + 1) / 2 + (DC + 15) / 16 + 1; }
// tiles per wave: as many as the accumulators allow (72 doubles; 64 for 8 quads at D = 80, 96, 112, where 72 spill) -- 9 tiles
// with 8 cluster quads, 12 with 6, 14 with 5: a launch with fewer quads keeps its MFMAs per step (and has fewer blocks
// re-staging the same rows)
__host__ __device__ constexpr int ft_tpw_max(int DP, int NQ) {
  const int t = (NQ >= 8 && DP > 64 && DP < 128 ? 64 : 72) / NQ;
  return t < 14 ? t : 14;  // (two address registers per tile)
}
// Waves per block.  The waves of a block share one staged batch, so 8 waves (one block per CU, longer batches) halve
// the staging instructions and barriers per MFMA against 4 waves (two blocks per CU): D = 64, N = 10M, K = 32:
// 22.03 -> 21.70 ms.  With 8 waves the tiles are dealt out unevenly -- the first TILES % NW of a chunk's NW = waves x
// nslice waves take one tile more than the others (ft_tpw is the larger count) -- so no wave carries idle tiles (D = 128:
// 537 tiles = 25 waves x 9 + 39 x 8; with 9 everywhere 64 waves would carry 39 idle ones: 73.0 -> 77.2 ms).
#ifndef LC_FT_WAVES64
#define LC_FT_WAVES64 8
#endif
#ifndef LC_FT_WAVES128
#define LC_FT_WAVES128 8
#endif
#ifndef LC_FT_BR
#define LC_FT_BR 32
#endif
#ifndef LC_FT_BR64
#define LC_FT_BR64 48
#endif
#ifndef LC_FT_BR128
#define LC_FT_BR128 32
#endif
// tiles whose products are formed together in front of their MFMAs (suffstat_feat_kernel's step loop)
// (round 5, K = 20 / 40 at D = 64: threes and fours for the launches of <= 6 quads measure the same as pairs, gpurun_out/r05y2)
__host__ __device__ constexpr int ft_group(int DP) { return DP >= 48 && DP <= 112 ? 2 : 1; }
// (round 5: D = 80, 96, 112 as 8-wave blocks too -- half the staging registers per thread, and with them every instance
//  but <80, 6> free of scratch, where the 4-wave instances spilled 64 ... 136 bytes into the step loop: the feature GEMM
//  now runs at these widths, N = 4M, K = 32: D = 96 22.9 -> 19.1 ms (0.67 -> 0.81 of the fp64 peak), D = 80 15.0 -> 13.7,
//  D = 112 (N = 3M) 21.9 -> 19.9; gpurun_out/r05l)
__host__ __device__ constexpr int ft_waves(int DP) { return DP == 64 ? LC_FT_WAVES64 : DP == 128 ? LC_FT_WAVES128 : DP > 64 ? 8 : 4; }
__host__ __device__ constexpr int ft_nslice(int DP, int DC, int NQ) {  // blocks per row chunk
  return (ft_tiles(DC) + ft_waves(DP) * ft_tpw_max(DP, NQ) - 1) / (ft_waves(DP) * ft_tpw_max(DP, NQ));
}
__host__ __device__ constexpr int ft_tpw(int DP, int DC, int NQ) {  // tiles of the fuller waves
  return (ft_tiles(DC) + ft_waves(DP) * ft_nslice(DP, DC, NQ) - 1) / (ft_waves(DP) * ft_nslice(DP, DC, NQ));
}
// (160 KB of LDS per CU: two blocks of 4 waves or one of 8)
__host__ __device__ constexpr int ft_batch_rows(int DP) {
  return DP == 128 && ft_waves(DP) == 8 ? LC_FT_BR128 : DP > 96 ? 24 : DP == 64 && ft_waves(DP) == 8 ? LC_FT_BR64 : LC_FT_BR;
}
// row stride of the staged q quads (32 or 64 cluster columns + 4): the rows of a half-wave 8 banks apart
__host__ __device__ constexpr int ft_qld(int NQ) { return NQ > 8 ? 68 : 36; }
// cluster range of one launch: 32 (up to 8 quads), or -- D = 128 -- 64 in ONE pass over X (16 quads, 4 tiles per wave:
// one multiply per 16 MFMAs; config 5's K = 64 used to take two launches of 8 quads, each re-reading and re-staging X)
inline int ft_range(int DP, int K) { return DP == 128 && K % 64 == 0 ? 64 : DP == 128 && K % 64 >= 57 ? 64 : 32; }
inline bool ss_feat_eligible(int DP, int K, int DC) {
  // (tests, libcluster_hip_testhooks.so only: 0 off, 1 where it wins, 2 everywhere it exists)
  static const int mode = test_switch("LC_SS_FEAT") ? atoi(test_switch("LC_SS_FEAT")) : 1;
  if (mode == 0 || DP < 32 || DP > 128) return false;
  if (mode == 2) return true;
  // active width below the padded one (round 6): this kernel skips the idle patches (D = 23: 24 of 39 tiles), the per-cluster
  // kernel works in 16 x 16 blocks and cannot -- from two cluster quads up the feature GEMM is the faster one there
  if (DC > 0 && DC < DP && K >= LC_SS_FEAT_MINK_ACTIVE) return true;
  // few clusters (one launch of <= 4 quads): the per-cluster kernel holds its own up to K = 12 everywhere and at D = 64 /
  // 128 up to 16 (tools/ss_smallk_probe.py, round 5: D = 64, K = 16 0.685 against 0.672 of the peak); a full fourth quad
  // pays at the widths whose per-cluster instances are the weak ones: D = 96, K = 16 0.64 -> 0.75, D = 32 0.56 -> 0.61
  if (K <= 16) return K >= 13 && (DP == 32 || DP == 80 || DP == 96 || DP == 112);
  // measured (tools/ssfeat_check.py, MI355X): it wins where EVERY launch carries 7 or 8 cluster quads (one multiply per
  // 7-8 MFMAs): K = 28..32, 60..64, ...; a remainder launch with few quads costs a whole pass over X at a poor ratio
  // (K = 33: 6.8 against 5.8 ms at N = 2M), and with 5-6 quads the two kernels are level
  // round 4, with chunk counts that fill the last round of resident blocks (suffstat_plan): at D <= 64 a remainder launch
  // of 1 ... 7 quads pays as soon as it carries a whole quad's worth of clusters (D = 64: K = 36 +2 %, 40 +9 %, 48 +2 %,
  // 56 +17 %; D = 48, K = 48 +12 %; D = 32, K = 40 +21 %; K = 33 still loses 12 %); D = 128 keeps the round-3 rule
  // (K = 40 level, 48 -5 %) next to the 64-cluster ranges
  const int rem = K % 32;
  if (!LC_SS_FEAT_WIDTHS(DP)) return false;
  if (DP == 128) return rem == 0 || rem >= 28 || (ft_range(DP, K) == 64 && K >= 57);
  return rem == 0 || rem >= 4;
}
template <int DP, int DC, int NQ>
__global__ void __launch_bounds__(64 * ft_waves(DP), 2) suffstat_feat_kernel(SuffstatLaunch a) {
  static_assert(DC % 4 == 0 && DC <= DP && DC > DP - 16, "active width");
  constexpr int FTW = ft_waves(DP);
  constexpr int BR = ft_batch_rows(DP), LD = lds_row_stride(DP), QLD = ft_qld(NQ), XBUF = BR * LD, QBUF = BR * QLD;
  constexpr int TPW = ft_tpw(DP, DC, NQ), TILES = ft_tiles(DC), NPATCH = (DC / 4) * (DC / 4 + 1) / 2, NSL = ft_nslice(DP, DC, NQ);
  constexpr int NLIN = (DC + 15) / 16;  // s_k tiles
  constexpr int ONE = DP;  // column of the staged rows that holds 1.0
  static_assert(LD > DP, "the staged rows need a spare column");
  static_assert(TPW * NQ <= 72, "accumulators");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* xbuf = lds;              // [2][BR][LD]
  double* qbuf = lds + 2 * XBUF;   // [2][BR][QLD]   q[row][cluster - k0], clusters past the range are zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  const int k0 = a.klast0;                                  // first cluster of this launch's range
  const int KC = a.K - k0 < 4 * NQ ? a.K - k0 : 4 * NQ;     // clusters in the range
  int chunk, slice;
  {  // (chunk, slice) placement as in suffstat_kernel: the slices of a chunk back-to-back on one XCD
    const int nslice = NSL, nchunks = a.nchunks, b = blockIdx.x, full = (nchunks / 8) * 8;
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
  const int64_t r0 = (int64_t)chunk * a.chunk_rows;
  const int64_t r1 = (r0 + a.chunk_rows) < a.NP ? (r0 + a.chunk_rows) : a.NP;
  // this wave's tiles and, per lane, where their two operand fragments sit in a staged batch (in doubles, step 0)
  // 8 waves: NFULL waves of TPW tiles, then TPW - 1 (two instances of the batch body).  4 waves: TPW everywhere, the
  // last waves of a chunk carry idle tiles -- the second instance costs more there than the idle tiles do (D = 128,
  // 4 waves: 72.9 -> 77.2 ms with the uneven deal; both bodies compete for the instruction cache)
  constexpr bool UNEVEN = FTW == 8;
  constexpr int NW = FTW * NSL, TPWL = UNEVEN ? TILES / NW : TPW, NFULL = UNEVEN ? TILES - TPWL * NW : 0;
  static_assert(TPWL >= 1 && (NFULL == 0 ? TPWL : TPWL + 1) == TPW, "tile deal");
  const int gw = slice * FTW + wave;
  const int T0 = UNEVEN ? gw * TPWL + (gw < NFULL ? gw : NFULL) : gw * TPW;
  const int nt = UNEVEN ? TPWL + (gw < NFULL ? 1 : 0) : (TILES - T0 < TPW ? (TILES - T0 > 0 ? TILES - T0 : 0) : TPW);
  const double* pu[TPW];
  const double* pw[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int T = T0 + t;
    int cu = ONE, cw = ONE;  // N_k tile (and the idle ones): 1 * 1
    if (T < NPATCH) {
      int ja = 0;
      while ((ja + 1) * (ja + 2) / 2 <= T) ++ja;
      const int ia = T - ja * (ja + 1) / 2;
      cu = 4 * ia + lo2;
      cw = 4 * ja + blk;
    } else if (T < NPATCH + NLIN) {
      cu = 16 * (T - NPATCH) + 4 * blk + lo2;
    }
    pu[t] = xbuf + hi * LD + cu;
    pw[t] = xbuf + hi * LD + cw;
  }
  double acc[TPW][NQ];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int c = 0; c < NQ; ++c) acc[t][c] = 0.0;

  // ---- staging: registers hold the next batch while the current one is consumed.
  // X batch: thread -> 16 bytes of NPRE consecutive rows (row group tid / TPR, column pair tid % TPR): one address
  // register on either side, the rows are immediate offsets.  q batch: thread -> (cluster tid % QCL, row quad tid / QCL),
  // QCL = 32 or 64 cluster columns: clusters fastest, so that the 16-lane groups of a ds_write_b64 fill one LDS row (rows
  // fastest would put all the row quads of a group on one bank)
  constexpr int NTHR = 64 * FTW;
  constexpr int TPR = DP / 2, NV2 = BR * TPR, NPRE = (NV2 + NTHR - 1) / NTHR;  // double2 per row / per batch / per thread
  constexpr int RPP = NTHR / TPR;                                               // rows covered by one pass of the block
  static_assert(NTHR % TPR == 0 || NPRE * NTHR >= NV2, "staging");
  constexpr int QCL = NQ > 8 ? 64 : 32;
  static_assert(BR / 4 <= NTHR / QCL && 2 * BR <= NTHR, "q staging / ones column");
  double pre[NPRE][2], qpre[4];
  const int qcl = tid % QCL, qrq = tid / QCL;
  const bool qthr = qrq < BR / 4;
  const int xr0 = tid / TPR, xc2 = tid % TPR;  // (DP = 80, 112: 256 is not a multiple of TPR -- the generic index below)
  double* const qdst = qbuf + (4 * qrq) * QLD + qcl;
  auto gload = [&](int64_t b0) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      int row, c2;
      if constexpr (NTHR % TPR == 0) {
        row = xr0 + i * RPP, c2 = xc2;
      } else {
        const int idx = tid + i * NTHR;
        row = idx / TPR, c2 = idx % TPR;
      }
      double2 v = make_double2(0.0, 0.0);
      if (row < BR && b0 + row < r1) v = *reinterpret_cast<const double2*>(a.X + (b0 + row) * DP + 2 * c2);
      pre[i][0] = v.x;
      pre[i][1] = v.y;
    }
    const int64_t qrow = b0 + 4 * qrq;
#pragma unroll
    for (int i = 0; i < 4; ++i) qpre[i] = 0.0;
    if (qthr && qcl < KC && qrow < r1) {  // (a row quad lies inside the chunk or outside it: chunk_rows is a multiple of 4)
      const double2* qp = reinterpret_cast<const double2*>(a.qZ + (int64_t)(k0 + qcl) * a.ldq + qrow);
      const double2 v0 = qp[0], v1 = qp[1];
      qpre[0] = v0.x, qpre[1] = v0.y, qpre[2] = v1.x, qpre[3] = v1.y;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      int row, c2;
      if constexpr (NTHR % TPR == 0) {
        row = xr0 + i * RPP, c2 = xc2;
      } else {
        const int idx = tid + i * NTHR;
        row = idx / TPR, c2 = idx % TPR;
      }
      if (row < BR) *reinterpret_cast<double2*>(xbuf + buf * XBUF + row * LD + 2 * c2) = make_double2(pre[i][0], pre[i][1]);
    }
    if (qthr && qcl < 4 * NQ) {
#pragma unroll
      for (int i = 0; i < 4; ++i) qdst[buf * QBUF + i * QLD] = qpre[i];
    }
  };
  if (tid < 2 * BR) xbuf[(tid / BR) * XBUF + (tid % BR) * LD + ONE] = 1.0;
  if (r0 < r1) {
    gload(r0);
    lstore(0);
  }
  __syncthreads();
  const double* pq = qbuf + hi * QLD + lo2;
  // One batch from buffer B, for a wave of NT tiles (two instances: the fuller waves and the others).  All BR / 4 steps
  // run (rows past the chunk end were staged as zeros with q = 0): nothing in the loop depends on run-time counts.  The next tile's fragments -- behind the last tile the next step's first tile and its
  // q quads -- are issued BEFORE this tile's MFMAs and arrive under them (two register sets that swap roles; the fences
  // keep hipcc from sinking the reads to their uses).
  // 16 quads: two sets of q quads would be 64 registers next to 128 of accumulators -- ONE set, refilled in place during
  // the step's last tile (quad c's register takes the next step's quad c right behind the MFMA that read it)
  constexpr bool ONEQ = NQ > 8;
  auto batch = [&](auto bsel, auto ntsel) {
    constexpr int B = decltype(bsel)::value, XO = B * XBUF, QO = B * QBUF, NT = decltype(ntsel)::value;
    if constexpr (!ONEQ && ft_group(DP) > 1) {
      // Tiles in groups of G: the group's products first, then its G x NQ MFMAs.  Next to the matrix pipe a VALU
      // instruction is paid per switch between the two kinds, not per instruction (tools/mfma_batch_probe.hip: ~ 12 clocks
      // for a lone multiply between MFMAs, 8 each in pairs, 5.7 in fours); the fragments of the next group are read into
      // the registers the multiplies have just freed and arrive under the MFMAs.  ONE set of q quads, refilled in place
      // during the last tile of a step (as the 16-quad instance does): the second set's 16 registers pay for the group.
      // Measured (N = 10M, D = 64, K = 32; gpurun_out/r05e): pairs 21.61 -> 21.22 ms, threes 21.30, fours 21.51, fives 21.79
      // -- the multiplies were a third of what the step loses, and larger groups expose the fragment reads; D = 48 gains
      // 1 % with pairs, D = 80 ... 112 1-2 %, D = 32 and D = 128 nothing (ft_group).
      constexpr int G = ft_group(DP), TOT = (BR / 4) * NT;
      double qa[NQ], u[G], w[G], pp[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (g < TOT && g < NT) {
          u[g] = pu[g][XO];
          w[g] = pw[g][XO];
        }
#pragma unroll
      for (int c = 0; c < NQ; ++c) qa[c] = pq[QO + 4 * c];
#pragma unroll
      for (int st = 0; st < BR / 4; ++st) {
        // (groups never straddle a step: the step's q quads change there)
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += G) {
          const int ng = NT - t0 < G ? NT - t0 : G;
#pragma unroll
          for (int g = 0; g < G; ++g)
            if (g < ng) pp[g] = u[g] * w[g];
          // the next group: the rest of this step, or the head of the next one
          const int nt0 = t0 + G < NT ? t0 + G : 0, nst = t0 + G < NT ? st : st + 1;
          if (nst < BR / 4) {
            const int nng = NT - nt0 < G ? NT - nt0 : G;
#pragma unroll
            for (int g = 0; g < G; ++g)
              if (g < nng) {
                u[g] = pu[nt0 + g][XO + nst * 4 * LD];
                w[g] = pw[nt0 + g][XO + nst * 4 * LD];
              }
          }
          const bool refill = nst != st && nst < BR / 4;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < G; ++g)
            if (g < ng) {
              if (refill && g + 1 == ng) {
#pragma unroll
                for (int c4 = 0; c4 < NQ; c4 += 4) {
#pragma unroll
                  for (int c = c4; c < c4 + 4 && c < NQ; ++c) acc[t0 + g][c] = mfma4(qa[c], pp[g], acc[t0 + g][c]);
#pragma unroll
                  for (int c = c4; c < c4 + 4 && c < NQ; ++c) qa[c] = pq[QO + nst * 4 * QLD + 4 * c];
                  __builtin_amdgcn_sched_barrier(0);
                }
              } else {
#pragma unroll
                for (int c = 0; c < NQ; ++c) acc[t0 + g][c] = mfma4(qa[c], pp[g], acc[t0 + g][c]);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
    double qa[ONEQ ? 1 : 2][NQ], u[2], w[2];
    u[0] = pu[0][XO];
    w[0] = pw[0][XO];
#pragma unroll
    for (int c = 0; c < NQ; ++c) qa[0][c] = pq[QO + 4 * c];
#pragma unroll
    for (int st = 0; st < BR / 4; ++st) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int cur = (st * NT + t) & 1, nxt = cur ^ 1;
        const double p = u[cur] * w[cur];
        const bool refill = t + 1 == NT && st + 1 < BR / 4;
        if (t + 1 < NT) {
          u[nxt] = pu[t + 1][XO + st * 4 * LD];
          w[nxt] = pw[t + 1][XO + st * 4 * LD];
        } else if (st + 1 < BR / 4) {
          u[nxt] = pu[0][XO + (st + 1) * 4 * LD];
          w[nxt] = pw[0][XO + (st + 1) * 4 * LD];
          if constexpr (!ONEQ) {
#pragma unroll
            for (int c = 0; c < NQ; ++c) qa[(st + 1) & 1][c] = pq[QO + (st + 1) * 4 * QLD + 4 * c];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ONEQ) {
#pragma unroll
          for (int c4 = 0; c4 < NQ; c4 += 4) {
#pragma unroll
            for (int c = c4; c < c4 + 4; ++c) acc[t][c] = mfma4(qa[0][c], p, acc[t][c]);
            if (refill) {
#pragma unroll
              for (int c = c4; c < c4 + 4; ++c) qa[0][c] = pq[QO + (st + 1) * 4 * QLD + 4 * c];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int c = 0; c < NQ; ++c) acc[t][c] = mfma4(qa[st & 1][c], p, acc[t][c]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  for (int64_t b0 = r0; b0 < r1; b0 += 2 * BR) {  // two batches per trip: the buffer index is a compile-time constant
    const bool more1 = b0 + BR < r1, more2 = b0 + 2 * BR < r1;
    if (more1) gload(b0 + BR);
    if (NFULL == 0 || nt == TPW) batch(std::integral_constant<int, 0>{}, std::integral_constant<int, TPW>{});
    else batch(std::integral_constant<int, 0>{}, std::integral_constant<int, TPWL>{});
    if (more1) lstore(1);
    __syncthreads();
    if (!more1) break;
    if (more2) gload(b0 + 2 * BR);
    if (NFULL == 0 || nt == TPW) batch(std::integral_constant<int, 1>{}, std::integral_constant<int, TPW>{});
    else batch(std::integral_constant<int, 1>{}, std::integral_constant<int, TPWL>{});
    if (more2) lstore(0);
    __syncthreads();
  }

  // ---- partial records: [N_k, s_k[DP], S_k[DP x DP]] per (chunk, cluster), as suffstat_kernel writes them
  const int64_t SS = 1 + (int64_t)DP + (int64_t)DP * DP;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t >= nt) continue;
    const int T = T0 + t;
    int ia = 0, ja = 0;
    if (T < NPATCH) {
      while ((ja + 1) * (ja + 2) / 2 <= T) ++ja;
      ia = T - ja * (ja + 1) / 2;
    }
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int kk = 4 * c + hi;
      if (kk >= KC) continue;
      double* out = a.partial + ((int64_t)chunk * a.KR + k0 + kk) * SS;
      const double v = acc[t][c];
      if (T < NPATCH) {
        const int gi = 4 * ia + lo2, gj = 4 * ja + blk;
        double* S = out + 1 + DP;
        S[(int64_t)gi * DP + gj] = v;
        S[(int64_t)gj * DP + gi] = v;
      } else if (T < NPATCH + NLIN) {
        out[1 + 16 * (T - NPATCH) + 4 * blk + lo2] = v;
      } else if (blk == 0 && lo2 == 0) {
        out[0] = v;
      }
    }
  }
}

template <int DP, int DC, int NQ>
static hipError_t launch_ss_feat_q(const SuffstatLaunch& b, hipStream_t stream) {
  constexpr int BR = ft_batch_rows(DP), LD = lds_row_stride(DP);
  const size_t shmem = (size_t)(2 * BR * LD + 2 * BR * ft_qld(NQ)) * sizeof(double);
  auto kern = suffstat_feat_kernel<DP, DC, NQ>;
  static LdsGrant grant;
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grant); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(b.nchunks * ft_nslice(DP, DC, NQ))), dim3(64 * ft_waves(DP)), shmem, stream, b);
  return hipGetLastError();
}
template <int DP, int DC>
static hipError_t launch_ss_feat_d(const SuffstatLaunch& a, hipStream_t stream) {
  // cluster ranges: as many full ranges of 32 (8 quads: one multiply per 8 MFMAs) as there are, the remainder in a last
  // range of its own -- near-equal ranges (K = 48 as 24 + 24) put BOTH launches at the poor 5-6 quad ratio.  Every range
  // re-reads X, which an MFMA-bound pass affords.
  SuffstatLaunch b = a;
  if (b.KR < a.K) b.KR = a.K;
  const int range = ft_range(DP, a.K);
  for (int k0 = 0; k0 < a.K; k0 += range) {
    b.klast0 = k0;
    const int nq = ((a.K - k0 < range ? a.K - k0 : range) + 3) / 4;
    hipError_t e = hipErrorInvalidValue;
    if constexpr (DP == 128) {
      if (nq > 8) {
        e = launch_ss_feat_q<DP, DC, 16>(b, stream);
        if (e != hipSuccess) return e;
        continue;
      }
    }
    switch (nq) {
      case 1: case 2: e = launch_ss_feat_q<DP, DC, 2>(b, stream); break;
      case 3: case 4: e = launch_ss_feat_q<DP, DC, 4>(b, stream); break;
      case 5: e = launch_ss_feat_q<DP, DC, 5>(b, stream); break;
      case 6: e = launch_ss_feat_q<DP, DC, 6>(b, stream); break;
      case 7: e = launch_ss_feat_q<DP, DC, 7>(b, stream); break;
      case 8: e = launch_ss_feat_q<DP, DC, 8>(b, stream); break;
    }
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
template <int DP>
static hipError_t launch_ss_feat_w(const SuffstatLaunch& a, hipStream_t stream) {
  if constexpr (DP <= 48) {
    if (a.DC == DP - 4) return launch_ss_feat_d<DP, DP - 4>(a, stream);
    if (a.DC == DP - 12) return launch_ss_feat_d<DP, DP - 12>(a, stream);
  }
  if (a.DC == DP - 8) return launch_ss_feat_d<DP, DP - 8>(a, stream);
  if (a.DC != 0 && a.DC != DP) return hipErrorInvalidValue;
  return launch_ss_feat_d<DP, DP>(a, stream);
}
static hipError_t launch_ss_feat(const SuffstatLaunch& a, hipStream_t stream) {
  switch (a.DP) {
    case 32: return launch_ss_feat_w<32>(a, stream);
    case 48: return launch_ss_feat_w<48>(a, stream);
    case 64: return launch_ss_feat_w<64>(a, stream);
    case 80: return launch_ss_feat_w<80>(a, stream);
    case 96: return launch_ss_feat_w<96>(a, stream);
    case 112: return launch_ss_feat_w<112>(a, stream);
    case 128: return launch_ss_feat_w<128>(a, stream);
  }
  return hipErrorInvalidValue;
}

// ===========================================================================
// Sufficient statistics for FEW clusters (K <= 16) at D <= 64: four clusters in the four blocks of one MFMA (round 6)
// ===========================================================================
// Few clusters leave both forms above short of operands to share: suffstat_kernel forms q_k x per cluster and 16 x 16
// block (D = 32: five fp64 VALU instructions per ten MFMAs), the feature GEMM shares one product among NQ <= 4 cluster
// quads and reads two LDS fragments for it (NQ = 2: the LDS port is the bound).  Here the MFMA's four blocks are the four
// CLUSTERS of a quad and its output tile a 4 x 4 patch of the scatter matrix:
//   A operand = x[row 4 st + hi][4 ia + lo2]                        (the same in all four blocks: one LDS read per ia and step)
//   B operand = q[row 4 st + hi][cluster 4 c + blk] * x[row][4 ja + lo2]   (ONE v_mul_f64 per ja and step, used by ja + 1 patches;
//                                                                    s_k adds the same product up on the VALU)
//   D[i = hi][j = lo2] of block blk = S_{4c + blk}[4 ia + hi][4 ja + lo2]
// -- NT multiplies, NT + 1 additions and NT + 1 LDS reads per NT (NT + 1) / 2 MFMAs and step (D = 48: 25 : 78, D = 24: 13 : 21), no
// rotated reads, cross-lane sums in the epilogue only.  A quad's work list (patches ia <= ja in ja-major order, each ja followed by its s_k
// tile, N_k last) is cut into NPART contiguous parts of equal length, one wave each: K <= 4 runs as 4 parts, K <= 8 as 2 x 2,
// 12 as 3 x 4 (three blocks per row chunk), 16 as 4 x 1 (D <= 32) or 4 x 2; D = 64 always in 4 parts (39 accumulators).  Same staging, chunks, partial records and
// fixed-order reduction as the other two kernels; a diagonal patch stores its lower triangle only (and mirrors it): the two
// halves come from differently rounded products, and the host reads the lower triangle.  Dense only.
struct SqItem {
  int kind, ia, ja;  // kind 0: patch (ia, ja); 1: s_k tile of ja; 2: N_k
};
__host__ __device__ constexpr int sq_items(int DC) { return (DC / 4) * (DC / 4 + 1) / 2 + DC / 4 + 1; }
__host__ __device__ constexpr SqItem sq_item(int DC, int n) {
  for (int ja = 0; ja < DC / 4; ++ja) {
    if (n <= ja) return SqItem{0, n, ja};
    n -= ja + 1;
    if (n == 0) return SqItem{1, 0, ja};
    n -= 1;
  }
  return SqItem{2, 0, 0};
}
__host__ __device__ constexpr int sq_part_lo(int DC, int NPART, int p) { return (sq_items(DC) * p + NPART - 1) / NPART; }
__host__ __device__ constexpr int sq_part_max(int DC, int NPART) {
  int m = 0;
  for (int p = 0; p < NPART; ++p) {
    const int n = sq_part_lo(DC, NPART, p + 1) - sq_part_lo(DC, NPART, p);
    m = n > m ? n : m;
  }
  return m;
}
constexpr int SQ_BR = 32, SQ_QLD = 18;  // (rows per staged batch: 16 measures 1-7 % slower, 64 3-20 % -- the blocks a CU holds)
inline int sq_npart(int DP, int K) {  // parts per cluster quad (the instances launch_ss_quad has)
  const int nq = (K + 3) / 4;
  if (DP > 48) return 4;  // (D = 64: two parts per quad need more than 256 registers)
  return nq == 2 ? 2 : nq == 4 ? (DP <= 32 ? 1 : 2) : 4;
}
inline int sq_nslice(int DP, int K) { return (((K + 3) / 4) * sq_npart(DP, K) + 3) / 4; }
inline bool ss_quad_eligible(int DP, int K) {
  // (tests, libcluster_hip_testhooks.so only: 0 never, 1 wherever an instance exists)
  static const int mode = test_switch("LC_SS_QUAD") ? atoi(test_switch("LC_SS_QUAD")) : -1;
  if (mode == 0 || DP < 32 || DP > 64 || K < 1 || K > 16) return false;
  if (mode == 1) return true;
  // a quad's four MFMA blocks are four clusters whether they exist or not: K = 2 does twice the work of the per-cluster
  // kernel and loses (D = 64: 1.30 against 0.98 ms at N = 4M; cluster()'s two-cluster sub-problems: 403 launches, 107 against
  // 73 ms); from three clusters on the form wins although blocks idle (D = 48, K = 5: 1.76 against 2.04 ms, D = 32, K = 6:
  // 1.10 against 1.28; gpurun_out/r06i)
  return K >= 3;
}
template <int DP, int DC, int NPART>
__global__ void __launch_bounds__(256, 2) suffstat_quad_kernel(SuffstatLaunch a) {
  static_assert(DC % 4 == 0 && DC <= DP && DC > DP - 16, "active width");
  constexpr int NT = DC / 4, BR = SQ_BR, LD = lds_row_stride(DP), QLD = SQ_QLD, XBUF = BR * LD, QBUF = BR * QLD, NTHR = 256;
  constexpr int NACC = sq_part_max(DC, NPART);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* xbuf = lds;             // [2][BR][LD]
  double* qbuf = lds + 2 * XBUF;  // [2][BR][QLD]: q[row][cluster], clusters past K zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  const int K = a.K, NQ = (K + 3) / 4;
  int chunk, slice;
  {  // (chunk, slice) placement as in suffstat_kernel: the slices of a chunk back-to-back on one XCD
    const int nslice = a.nslice, nchunks = a.nchunks, b = blockIdx.x, full = (nchunks / 8) * 8;
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
  const int64_t r0 = (int64_t)chunk * a.chunk_rows;
  const int64_t r1 = (r0 + a.chunk_rows) < a.NP ? (r0 + a.chunk_rows) : a.NP;
  const int unit = slice * 4 + wave, quad = unit / NPART, part = unit % NPART;
  const bool busy = quad < NQ;

  // ---- staging (as suffstat_feat_kernel): registers hold the next batch while the current one is consumed
  constexpr int TPR = DP / 2, NV2 = BR * TPR, NPRE = (NV2 + NTHR - 1) / NTHR;
  double pre[NPRE][2], qpre[4];
  const int qcl = tid % 16, qrq = tid / 16;
  const bool qthr = qrq < BR / 4;
  auto gload = [&](int64_t b0) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * NTHR;
      const int row = idx / TPR, c2 = idx % TPR;
      double2 v = make_double2(0.0, 0.0);
      if (row < BR && b0 + row < r1) v = *reinterpret_cast<const double2*>(a.X + (b0 + row) * DP + 2 * c2);
      pre[i][0] = v.x;
      pre[i][1] = v.y;
    }
    const int64_t qrow = b0 + 4 * qrq;
#pragma unroll
    for (int i = 0; i < 4; ++i) qpre[i] = 0.0;
    if (qthr && qcl < K && qrow < r1) {  // (a row quad lies inside the chunk or outside it: chunk_rows is a multiple of 4)
      const double2* qp = reinterpret_cast<const double2*>(a.qZ + (int64_t)qcl * a.ldq + qrow);
      const double2 v0 = qp[0], v1 = qp[1];
      qpre[0] = v0.x, qpre[1] = v0.y, qpre[2] = v1.x, qpre[3] = v1.y;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * NTHR;
      const int row = idx / TPR, c2 = idx % TPR;
      if (row < BR) *reinterpret_cast<double2*>(xbuf + buf * XBUF + row * LD + 2 * c2) = make_double2(pre[i][0], pre[i][1]);
    }
    if (qthr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) qbuf[buf * QBUF + (4 * qrq + i) * QLD + qcl] = qpre[i];
    }
  };

  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const double* px = xbuf + hi * LD + lo2;
  const double* pq = qbuf + hi * QLD + 4 * (busy ? quad : 0) + blk;

  // one part's work on one staged batch (buffer B): all BR / 4 steps run (rows past the chunk end were staged as zeros with
  // q = 0).  The next step's fragments are read before this step's MFMAs (two register sets); the products stand together
  // in front of them (a VALU instruction next to the matrix pipe is paid per switch).
  auto batch = [&](auto bsel, auto psel) {
    constexpr int B = decltype(bsel)::value, P = decltype(psel)::value, XO = B * XBUF, QO = B * QBUF;
    constexpr int LO = sq_part_lo(DC, NPART, P), HI = sq_part_lo(DC, NPART, P + 1);
    constexpr int JA0 = sq_item(DC, LO).kind == 2 ? NT - 1 : sq_item(DC, LO).ja;
    constexpr int JA1 = sq_item(DC, HI - 1).kind == 2 ? NT - 1 : sq_item(DC, HI - 1).ja;  // fragments 0 .. JA1, products JA0 .. JA1
    double xa[2][JA1 + 1], qa[2];
#pragma unroll
    for (int t = 0; t <= JA1; ++t) xa[0][t] = px[XO + 4 * t];
    qa[0] = pq[QO];
#pragma unroll
    for (int st = 0; st < BR / 4; ++st) {
      const int cur = st & 1, nxt = cur ^ 1;
      double pr[JA1 - JA0 + 1];
#pragma unroll
      for (int j = JA0; j <= JA1; ++j) pr[j - JA0] = qa[cur] * xa[cur][j];
      if (st + 1 < BR / 4) {
#pragma unroll
        for (int t = 0; t <= JA1; ++t) xa[nxt][t] = px[XO + (st + 1) * 4 * LD + 4 * t];
        qa[nxt] = pq[QO + (st + 1) * 4 * QLD];
      }
      // s_k and N_k on the VALU, next to the products (one v_add_f64 each: as MFMA tiles they were 7 of 28 matrix
      // instructions at D = 24, 13 of 91 at D = 48); the four hi lanes' partial sums meet in the epilogue
      static_for<HI - LO>([&](auto ic) {
        constexpr SqItem it = sq_item(DC, LO + ic);
        if constexpr (it.kind == 1) acc[ic] += pr[it.ja - JA0];
        else if constexpr (it.kind == 2) acc[ic] += qa[cur];
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<HI - LO>([&](auto ic) {
        constexpr SqItem it = sq_item(DC, LO + ic);
        if constexpr (it.kind == 0) acc[ic] = mfma4(xa[cur][it.ia], pr[it.ja - JA0], acc[ic]);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto run = [&](auto psel) {
    if (r0 < r1) {
      gload(r0);
      lstore(0);
    }
    __syncthreads();
    for (int64_t b0 = r0; b0 < r1; b0 += 2 * BR) {  // two batches per trip: the buffer index is a compile-time constant
      const bool more1 = b0 + BR < r1, more2 = b0 + 2 * BR < r1;
      if (more1) gload(b0 + BR);
      if (busy) batch(std::integral_constant<int, 0>{}, psel);
      if (more1) lstore(1);
      __syncthreads();
      if (!more1) break;
      if (more2) gload(b0 + 2 * BR);
      if (busy) batch(std::integral_constant<int, 1>{}, psel);
      if (more2) lstore(0);
      __syncthreads();
    }
    if (!busy) return;
    // ---- partial records: [N_k, s_k[DP], S_k[DP x DP]] per (chunk, cluster), as the other kernels write them
    constexpr int P = decltype(psel)::value, LO = sq_part_lo(DC, NPART, P), HI = sq_part_lo(DC, NPART, P + 1);
    const int kk = 4 * quad + blk;
    if (kk >= K) return;
    const int64_t SS = 1 + (int64_t)DP + (int64_t)DP * DP;
    double* out = a.partial + ((int64_t)chunk * a.KR + kk) * SS;
    double* S = out + 1 + DP;
    static_for<HI - LO>([&](auto ic) {
      constexpr SqItem it = sq_item(DC, LO + ic);
      const double v = acc[ic];
      if constexpr (it.kind == 0) {
        const int gi = 4 * it.ia + hi, gj = 4 * it.ja + lo2;
        if constexpr (it.ia == it.ja) {
          if (hi >= lo2) S[(int64_t)gi * DP + gj] = v;   // lower triangle of the diagonal patch ...
          if (hi > lo2) S[(int64_t)gj * DP + gi] = v;    // ... and its mirror image
        } else {
          S[(int64_t)gi * DP + gj] = v;
          S[(int64_t)gj * DP + gi] = v;
        }
      } else if constexpr (it.kind == 1) {
        const double t = sum_over_hi(v);
        if (hi == 0) out[1 + 4 * it.ja + lo2] = t;
      } else {
        const double t = sum_over_hi(v);
        if (hi == 0 && lo2 == 0) out[0] = t;
      }
    });
  };
  if constexpr (NPART == 1) {
    run(std::integral_constant<int, 0>{});
  } else if constexpr (NPART == 2) {
    if (part == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
  } else {
    static_assert(NPART == 4, "parts per quad");
    if (part == 0) run(std::integral_constant<int, 0>{});
    else if (part == 1) run(std::integral_constant<int, 1>{});
    else if (part == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
  }
}

template <int DP, int DC, int NPART>
static hipError_t launch_ss_quad_p(const SuffstatLaunch& a, hipStream_t stream) {
  const size_t shmem = (size_t)(2 * SQ_BR * lds_row_stride(DP) + 2 * SQ_BR * SQ_QLD) * sizeof(double);
  auto kern = suffstat_quad_kernel<DP, DC, NPART>;
  static LdsGrant grant;
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grant); e != hipSuccess) return e;
  if (a.occ_out) {  // (suffstat_plan: how many blocks of this instance a CU holds)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, shmem) != hipSuccess || per_cu < 1) per_cu = 2;
    *a.occ_out = per_cu;
    return hipSuccess;
  }
  SuffstatLaunch b = a;
  b.nslice = sq_nslice(DP, a.K);
  if (b.KR < a.K) b.KR = a.K;
  hipLaunchKernelGGL(kern, dim3((unsigned)(b.nchunks * b.nslice)), dim3(256), shmem, stream, b);
  return hipGetLastError();
}
template <int DP, int DC>
static hipError_t launch_ss_quad_d(const SuffstatLaunch& a, hipStream_t stream) {
  const int np = sq_npart(DP, a.K);  // (only the instances sq_npart can ask for exist)
  if constexpr (DP <= 32) {
    if (np == 1) return launch_ss_quad_p<DP, DC, 1>(a, stream);
  }
  if constexpr (DP <= 48) {
    if (np == 2) return launch_ss_quad_p<DP, DC, 2>(a, stream);
  }
  return np == 4 ? launch_ss_quad_p<DP, DC, 4>(a, stream) : hipErrorInvalidValue;
}
template <int DP>
static hipError_t launch_ss_quad_w(const SuffstatLaunch& a, hipStream_t stream) {
  if constexpr (DP <= 48) {
    if (a.DC == DP - 4) return launch_ss_quad_d<DP, DP - 4>(a, stream);
    if (a.DC == DP - 12) return launch_ss_quad_d<DP, DP - 12>(a, stream);
  }
  if (a.DC == DP - 8) return launch_ss_quad_d<DP, DP - 8>(a, stream);
  if (a.DC != 0 && a.DC != DP) return hipErrorInvalidValue;
  return launch_ss_quad_d<DP, DP>(a, stream);
}
static hipError_t launch_ss_quad(const SuffstatLaunch& a, hipStream_t stream) {
  switch (a.DP) {
    case 32: return launch_ss_quad_w<32>(a, stream);
    case 48: return launch_ss_quad_w<48>(a, stream);
    case 64: return launch_ss_quad_w<64>(a, stream);
  }
  return hipErrorInvalidValue;
}

template <int DP>
struct SSCfg;
template <>
struct SSCfg<16> { static constexpr int CPW = 4; };
template <>
struct SSCfg<32> { static constexpr int CPW = 4; };
template <>
struct SSCfg<48> { static constexpr int CPW = 2; };  // (four clusters per wave spill at the 256-register budget)
template <>
struct SSCfg<64> { static constexpr int CPW = 2; };
template <>
struct SSCfg<80> { static constexpr int CPW = 1; };
template <>
struct SSCfg<96> { static constexpr int CPW = 1; };
template <>
struct SSCfg<128> { static constexpr int CPW = 1; };

static int ss_cpw(int DP, int K) {
  int cpw = DP == 16 ? SSCfg<16>::CPW : DP == 32 ? SSCfg<32>::CPW : DP == 48 ? SSCfg<48>::CPW : DP == 64 ? SSCfg<64>::CPW
                                                                                             : SSCfg<128>::CPW;  // (96, 128)
  // few clusters: spread them over more waves instead of stacking them in one
  while (cpw > 1 && (K + cpw - 1) / cpw < 4 && (K + cpw / 2 - 1) / (cpw / 2) <= 4) cpw /= 2;
  return cpw;
}

// row classes for a last slice with `active` of its four waves in use
static int ss_row_classes(int active) { return active == 1 ? 4 : active == 2 ? 2 : 1; }

int suffstat_extra_records(int DP, int K, bool skip_or_items, int* klast0, int DC) {
  if (klast0) *klast0 = K;
  if (DP > 128 || skip_or_items || K < 1) return 0;
  if (ss_quad_eligible(DP, K)) return 0;      // (whole quads of clusters per wave: no ragged last slice)
  if (ss_feat_eligible(DP, K, DC)) return 0;  // (the feature-GEMM kernel covers any K of its range with the same 16 waves)
  const int cpw = ss_cpw(DP, K), kwaves = (K + cpw - 1) / cpw, nslice = (kwaves + 3) / 4;
  const int rs = ss_row_classes(kwaves - (nslice - 1) * 4);
  if (rs == 1) return 0;
  const int k0 = (nslice - 1) * 4 * cpw;
  if (klast0) *klast0 = k0;
  return (rs - 1) * (K - k0);
}

// rec[(klast0 + e % nlast) * SS + i] += rec[(K + e) * SS + i], e = 0 .. extra-1 in order (deterministic)
__global__ void __launch_bounds__(256) fold_extra_kernel(double* rec, int64_t SS, int K, int klast0, int extra) {
  const int nlast = K - klast0;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)nlast * SS) return;
  const int kk = (int)(t / SS);
  const int64_t i = t % SS;
  double v = rec[(int64_t)(klast0 + kk) * SS + i];
  for (int e = kk; e < extra; e += nlast) v += rec[(int64_t)(K + e) * SS + i];
  rec[(int64_t)(klast0 + kk) * SS + i] = v;
}
hipError_t launch_fold_extra(double* rec, int64_t SS, int K, int klast0, int extra, hipStream_t stream) {
  if (extra <= 0) return hipSuccess;
  const int64_t n = (int64_t)(K - klast0) * SS;
  hipLaunchKernelGGL(fold_extra_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rec, SS, K, klast0, extra);
  return hipGetLastError();
}

// which kernel a dense Gauss-Wishart statistics pass of this shape runs (bench.py names the kernel it prices)
const char* suffstat_kernel_name(int DP, int K, int DC) {
  if (ss_quad_eligible(DP, K)) return "suffstat_quad_kernel";
  return DP <= 128 && ss_feat_eligible(DP, K, DC) ? "suffstat_feat_kernel" : "suffstat_kernel";
}

int suffstat_clusters_per_block(int DP, int K) { return DP > 128 ? 4 : 4 * ss_cpw(DP, K); }

int suffstat_plan(int DP, int64_t NP, int K, int64_t* chunk_rows, int DC) {
  const int cpw = DP > 128 ? 1 : ss_cpw(DP, K);  // wide: panel launches, one cluster per wave
  const int kwaves = (K + cpw - 1) / cpw;  // waves needed to cover the clusters
  // aim for ~8 waves per CU on 256 CUs, at least 256 rows per chunk
  int64_t want = (256 * 8 + kwaves - 1) / kwaves;
  if (ss_quad_eligible(DP, K)) want = (256 * 8 / 4 + sq_nslice(DP, K) - 1) / sq_nslice(DP, K);  // blocks of four waves, nslice per chunk
  // four blocks per resident slot: the hardware back-fills slots as blocks retire, which evens out the
  // per-CU / per-XCD speed differences (measured 25.7 -> 24.9 ms at N=10M, D=64, K=32), while the partial
  // records (chunks x K x (1 + DP + DP^2) doubles) stay below 1 GiB
  const int rounds = 4;
  const int64_t rec = (int64_t)K * (1 + DP + (int64_t)DP * DP) * 8;
  // (wide records are large: allow 4 GiB of them so that the grid still covers the chip)
  const int64_t cap = ((int64_t)(DP > 128 ? 4 : 1) << 30) / rec;
  if (want * rounds <= cap) want *= rounds;
  else if (want < cap) want = cap;
  else if (want > cap && DP > 128) want = cap > 1 ? cap : 1;
  // at least 1024 rows per chunk where that still leaves two chunks per CU (a chunk's K records are written and read
  // back by the reduction: at 256 rows they are half of a two-cluster sub-problem's traffic), 256 rows otherwise
  int64_t maxchunks = std::max<int64_t>((NP + 1023) / 1024, std::min<int64_t>((NP + 255) / 256, 512));
  if (want > maxchunks) want = maxchunks;
  if (want < 1) want = 1;
  auto plan = [&](int64_t w, int64_t* rows_out) {
    int64_t rows = (NP + w - 1) / w;
    rows = (rows + SS_BR - 1) / SS_BR * SS_BR;  // whole staging batches
    *rows_out = rows;
    return (NP + rows - 1) / rows;
  };
  int64_t rows = 0;
  int64_t n = plan(want, &rows);
  // Feature-GEMM launches put nslice blocks on every chunk and one (8 waves) or two (4 waves) blocks on a CU: a grid
  // whose last round of resident blocks is half empty loses that much of the launch.  Config 5 (D = 128, K = 64: 17 blocks
  // per chunk) ran 127 chunks = 2159 blocks = 8.43 rounds of 256 -- 6 % of the pass idle; 120 chunks are 7.97 rounds.
  // Among the chunk counts down to 80 % of the wanted one, take the fullest last round (the larger count on ties).
  if (DP <= 128 && !ss_quad_eligible(DP, K) && ss_feat_eligible(DP, K, DC) && NP >= 64 * 1024) {
    const int range = ft_range(DP, K), nq = ((K < range ? K : range) + 3) / 4;
    const int nqi = nq > 8 ? 16 : nq > 4 ? nq : nq > 2 ? 4 : 2;  // the instance launch_ss_feat_d takes
    const int nslice = ft_nslice(DP, DC > 0 ? DC : DP, nqi);
    if (nslice > 0) {
      const int cus = current_device_cus();
      const int64_t slots = (int64_t)cus * (ft_waves(DP) == 8 ? 1 : 2);
      auto fill = [&](int64_t chunks) {
        const int64_t blocks = chunks * nslice, rounds = (blocks + slots - 1) / slots;
        return (double)blocks / (double)(rounds * slots);
      };
      double best = fill(n);
      for (int64_t w = want - 1; w >= 1 && w * 5 >= want * 4 && best < 0.985; --w) {
        int64_t r2 = 0;
        const int64_t n2 = plan(w, &r2);
        if (fill(n2) > best + 1e-9) best = fill(n2), n = n2, rows = r2;
      }
    }
  }
  // ... and the few-cluster kernel (sq_nslice blocks per chunk, 2 - 4 resident per CU): N = 5M, D = 48, K = 12 ran 684 chunks = 2052
  // blocks = 2.67 rounds of 768
  if (ss_quad_eligible(DP, K) && NP >= 64 * 1024) {
    static std::atomic<int> occ_cache[8 * 16 * 4];  // [layout][active-width step][parts]: resident blocks per CU, asked once
    const int dc = DC > 0 ? DC : DP, np = sq_npart(DP, K);
    const int slot = ((DP / 16) % 8) * 64 + (((DP - dc) / 4) % 16) * 4 + (np == 1 ? 0 : np == 2 ? 1 : 2);
    int per_cu = occ_cache[slot].load(std::memory_order_relaxed);
    if (per_cu <= 0) {
      SuffstatLaunch q{};
      q.DP = DP;
      q.DC = dc;
      q.K = K;
      q.nchunks = 1;
      q.occ_out = &per_cu;
      if (launch_ss_quad(q, nullptr) != hipSuccess || per_cu < 1) per_cu = 2;
      occ_cache[slot].store(per_cu, std::memory_order_relaxed);
    }
    const int nslice = sq_nslice(DP, K);
    const int64_t slots = (int64_t)current_device_cus() * per_cu;
    auto fill = [&](int64_t chunks) {
      const int64_t blocks = chunks * nslice, rounds = (blocks + slots - 1) / slots;
      return (double)blocks / (double)(rounds * slots);
    };
    // (chunk counts from 80 % to 125 % of the wanted one, the nearest first: three blocks per chunk on 768 slots have no
    //  full round between 512 and 768 chunks)
    double best = fill(n);
    for (int64_t d = 1; best < 0.985 && d * 4 <= want; ++d)
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int64_t w = want + sgn * d;
        if (w < 1 || (sgn < 0 && w * 5 < want * 4) || w > maxchunks || (sgn > 0 && w > cap)) continue;  // (cap: the partial records' memory)
        int64_t r2 = 0;
        const int64_t n2 = plan(w, &r2);
        if (fill(n2) > best + 1e-9) best = fill(n2), n = n2, rows = r2;
      }
  }
  *chunk_rows = rows;
  return (int)n;
}

template <int DP, int CPW, bool SKIP, int HALF, int PAN = 0, bool RSP = false>
static hipError_t launch_ss_k(const SuffstatLaunch& b, unsigned grid, hipStream_t stream) {
  const int wpb = 4;
  constexpr int BR = ss_batch_rows<PAN>();
  const size_t shmem = (size_t)(2 * (PAN == 2 ? 2 : 1) * BR * lds_row_stride(DP) + 2 * wpb * CPW * BR) * sizeof(double);
  auto kern = suffstat_kernel<DP, CPW, SKIP, HALF, PAN, RSP>;
  static LdsGrant grant;
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grant); e != hipSuccess) return e;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(wpb * 64), shmem, stream, b);
  return hipGetLastError();
}

template <int DP, int CPW, bool SKIP, int HALF, int PAN = 0>
static hipError_t launch_ss_h(const SuffstatLaunch& a, hipStream_t stream) {
  const int kwaves = (a.K + CPW - 1) / CPW;
  const int wpb = 4;
  const int nslice = (kwaves + wpb - 1) / wpb;
  SuffstatLaunch b = a;
  b.nslice = nslice;
  if (b.KR < a.K) b.KR = a.K;
  if constexpr (PAN == 0 && !SKIP && !(DP > 80 && HALF == 0)) {
    // ragged K: the last slice runs as its own launch with the idle waves sharing the rows (suffstat_extra_records)
    const int rs = ss_row_classes(kwaves - (nslice - 1) * wpb);
    if (!a.items && rs > 1 && b.KR > a.K) {
      b.nslice = nslice - 1;
      hipError_t e = launch_ss_k<DP, CPW, SKIP, HALF, PAN>(b, (unsigned)(a.nchunks * (nslice - 1)), stream);
      if (e != hipSuccess) return e;
      b.nslice = 1;
      b.slice0 = nslice - 1;
      b.rs = rs;
      b.klast0 = (nslice - 1) * wpb * CPW;
      b.nklast = a.K - b.klast0;
      return launch_ss_k<DP, CPW, SKIP, HALF, PAN, true>(b, (unsigned)a.nchunks, stream);
    }
  }
  return launch_ss_k<DP, CPW, SKIP, HALF, PAN>(b, a.items ? (unsigned)a.nitems : (unsigned)(a.nchunks * nslice), stream);
}

// Observations wider than 128 columns: one launch per pair of 64-column panels (P >= Q) over the same row chunks (and
// the same work list in sparse mode, built for four clusters per block: any clusters-per-wave serves it), all
// writing the same DP-wide records.  Two clusters per wave in the diagonal pairs (the D = 64 kernel as it is), one in
// the off-diagonal ones (64 accumulators per cluster).
template <bool SKIP>
static hipError_t launch_ss_wide(const SuffstatLaunch& a, hipStream_t stream) {
  SuffstatLaunch b = a;
  b.ldx = a.DP;
  b.DPW = a.DP;
  b.DP = 64;
  const int npan = a.DP / 64;
  for (int P = 0; P < npan; ++P)
    for (int Q = 0; Q <= P; ++Q) {
      b.colA = 64 * P;
      b.colB = 64 * Q;
      const hipError_t e = P == Q ? launch_ss_h<64, 2, SKIP, 0, 1>(b, stream) : launch_ss_h<64, 1, SKIP, 0, 2>(b, stream);
      if (e != hipSuccess) return e;
    }
  return hipSuccess;
}

template <int DP, int CPW, bool SKIP>
static hipError_t launch_ss_s(const SuffstatLaunch& a, hipStream_t stream) {
  if constexpr (DP > 80) {  // (at D = 64 the two-half variant is slower: 26.0 vs 23.6 ms; one launch at one wave per SIMD
                            //  measured 82.8 against 73.2 ms at D = 128: DESIGN 4.5.3)
    hipError_t e = launch_ss_h<DP, CPW, SKIP, 1>(a, stream);
    if (e != hipSuccess) return e;
    return launch_ss_h<DP, CPW, SKIP, 2>(a, stream);
  } else {
    return launch_ss_h<DP, CPW, SKIP, 0>(a, stream);
  }
}

template <int DP, int CPW>
static hipError_t launch_ss_t(const SuffstatLaunch& a, hipStream_t stream) {
  // skip_zero: 1 = skipping variant, -1 = dense even with a mask, 0 = skipping iff a mask is given
  const bool skip = a.skip_zero > 0 || (a.skip_zero == 0 && a.smask);
  return skip ? launch_ss_s<DP, CPW, true>(a, stream) : launch_ss_s<DP, CPW, false>(a, stream);
}

hipError_t launch_suffstat(const SuffstatLaunch& a, hipStream_t stream) {
  if (a.K <= 0 || a.nchunks <= 0) return hipSuccess;
  if (a.DP > 128) {
    if (a.DP % 64) return hipErrorInvalidValue;
    const bool skip = a.skip_zero > 0 || (a.skip_zero == 0 && a.smask);
    return skip ? launch_ss_wide<true>(a, stream) : launch_ss_wide<false>(a, stream);
  }
  if (ss_quad_eligible(a.DP, a.K) && !a.smask && !a.items && a.skip_zero <= 0 && a.KR <= a.K) return launch_ss_quad(a, stream);
  if (ss_feat_eligible(a.DP, a.K, a.DC) && !a.smask && !a.items && a.skip_zero <= 0 && a.KR <= a.K)
    return launch_ss_feat(a, stream);
  const int cpw = ss_cpw(a.DP, a.K);
  switch (a.DP) {
    case 16:
      return cpw == 4 ? launch_ss_t<16, 4>(a, stream) : cpw == 2 ? launch_ss_t<16, 2>(a, stream)
                                                                : launch_ss_t<16, 1>(a, stream);
    case 32:
      return cpw == 4 ? launch_ss_t<32, 4>(a, stream) : cpw == 2 ? launch_ss_t<32, 2>(a, stream)
                                                                : launch_ss_t<32, 1>(a, stream);
    case 48:
      return cpw == 2 ? launch_ss_t<48, 2>(a, stream) : launch_ss_t<48, 1>(a, stream);
    case 64:
      return cpw == 2 ? launch_ss_t<64, 2>(a, stream) : launch_ss_t<64, 1>(a, stream);
    case 80:
      return launch_ss_t<80, 1>(a, stream);
    case 96:
      return launch_ss_t<96, 1>(a, stream);
    case 112:
      return launch_ss_t<112, 1>(a, stream);
    case 128:
      return launch_ss_t<128, 1>(a, stream);
  }
  return hipErrorInvalidValue;
}


}  // namespace lck
