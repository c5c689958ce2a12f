// E-step and statistics of the diagonal (NormGamma) and exponential (ExpGamma) cluster families (src/distributions.cpp:418-590)
// (one translation unit per kernel family; the file header of lc_kernels_estep.hip maps kernels to the reference)
#include "lc_device.hpp"

#include <algorithm>

namespace lck {

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
                                          int K, int DPS, int DC, double (&acc)[KT]) {
  // PA / PW2 / PW1 already point at the first dimension of this chunk; DPS = parameter row stride, DC = chunk length
  const double* pa[KT];
  const double* p2[KT];
  const double* p1[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int k = k0 + j < K ? k0 + j : K - 1;  // clamped: the caller discards clusters >= K
    pa[j] = PA + (int64_t)k * DPS;
    p2[j] = PW2 + (int64_t)k * DPS;
    p1[j] = PW1 + (int64_t)k * DPS;
    acc[j] = 0.0;
  }
  for (int d = 0; d < DC; d += 4) {
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
    // the 64 x DP tile is one contiguous piece of X (row length = tile width): element idx of the tile is element
    // row0 * DP/2 + idx of X as double2; rows past NP are staged as zeros
    const int c2 = DP >> 1;  // double2 columns per row
    const unsigned inv = ((1u << 20) + c2 - 1) / c2;  // idx / c2 == (idx * inv) >> 20 for idx < 64 * c2 <= 4096
    const double2* T2 = reinterpret_cast<const double2*>(X) + row0 * c2;
    const int64_t left = NP - row0;
    const int lim = (int)(left < 64 ? left : 64) * c2;
    for (int idx = tid; idx < 64 * c2; idx += 256) {
      const int r = (int)(((unsigned)idx * inv) >> 20);
      double2 v = make_double2(0.0, 0.0);
      if (idx < lim) v = T2[idx];
      xt[2 * idx + r] = v.x;  // = xt[r * LD + 2 * c] with LD = 2 * c2 + 1
      xt[2 * idx + r + 1] = v.y;
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
        diag_tile<MODE, KT>(xr, PA, PW2, PW1, tile * KT, K, DP, DP, acc);
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int k = tile * KT + j;
          if (k < K) {
            const double v = crow[k] + acc[j];
            lq[i][j] = v;
            dt[i][j] = acc[j];
            mx = fmax(mx, v);
            if (raw && inb) (qZ + (int64_t)k * ldq + row0)[lane] = v;
          }
        }
      }
    }
  } else {
    for (int tile = w; tile < ntiles; tile += 4) {
      double acc[KT];
      diag_tile<MODE, KT>(xr, PA, PW2, PW1, tile * KT, K, DP, DP, acc);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int k = tile * KT + j;
        if (k < K) {
          const double v = crow[k] + acc[j];
          mx = fmax(mx, v);
          if (inb) (qZ + (int64_t)k * ldq + row0)[lane] = v;
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
    for (int i = 0; i < DIAG_MAXT; ++i) {
      if (w + 4 * i < ntiles) {  // wave-uniform: no exponentials for tile slots this wave does not own
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          lq[i][j] = exp(lq[i][j] - mx);  // exp(-inf) = 0 for the slots past K
          se += lq[i][j];
        }
      }
    }
  } else if (inb) {
    for (int tile = w; tile < ntiles; tile += 4)
      for (int j = 0; j < KT; ++j) {
        const int k = tile * KT + j;
        if (k < K) se += exp((qZ + (int64_t)k * ldq + row0)[lane] - mx);
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
            if (inb) (qZ + (int64_t)k * ldq + row0)[lane] = q;
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
            double* qp = qZ + (int64_t)k * ldq + row0 + lane;
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

// The same kernel for wide observations (DP > 128): see the chunk loop.  Kept apart from the kernel above, whose
// single-chunk body the compiler schedules markedly better (3.8 vs 6.5 ms at D = 64 when both shared one body).
template <int MODE, int KT, bool REG>
__global__ void __launch_bounds__(256)
    estep_diag_wide_kernel(const double* __restrict__ X, const double* __restrict__ PA, const double* __restrict__ PW2,
                      const double* __restrict__ PW1, const double* __restrict__ ctab, const int* __restrict__ rginfo,
                      double* __restrict__ qZ, double* __restrict__ fz_part, double* __restrict__ ll_part, int DP, int K,
                      int64_t NP, int64_t nrows, int64_t ldq, int raw) {
  extern __shared__ double lds[];
  // wide observations (DP > 128, the separable families have no limit on D): the row tile is staged in chunks of
  // 128 dimensions and the per-cluster sums run over the chunks
  const int LD = 129;
  double* xt = lds;             // [64][LD]
  double* red = lds + 64 * LD;  // [4][64]
  double* llw = red + 256;      // [K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * 64;
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
  if (REG) {
#pragma unroll
    for (int i = 0; i < DIAG_MAXT; ++i)
#pragma unroll
      for (int j = 0; j < KT; ++j) dt[i][j] = 0.0;
  }

  // ---- pass 1: the data term of log q~ for this wave's tiles, chunk by chunk ----
  for (int d0 = 0; d0 < DP; d0 += 128) {
    if (d0 > 0) __syncthreads();  // every wave is done with the previous chunk
    const int DC = DP - d0 < 128 ? DP - d0 : 128;  // (DP is a multiple of 64: the last chunk may be a half one)
    {
      const int c2 = DC >> 1, sh = __builtin_ctz(c2);  // double2 columns per chunk row (DC is a power of two)
      const double2* X2 = reinterpret_cast<const double2*>(X + d0);
      for (int idx = tid; idx < 64 * c2; idx += 256) {
        const int r = idx >> sh, c = idx & (c2 - 1);
        double2 v = make_double2(0.0, 0.0);
        if (row0 + r < NP) v = X2[(row0 + r) * (DP >> 1) + c];
        xt[r * LD + 2 * c] = v.x;
        xt[r * LD + 2 * c + 1] = v.y;
      }
    }
    __syncthreads();
    if (REG) {
#pragma unroll
      for (int i = 0; i < DIAG_MAXT; ++i) {
        const int tile = w + 4 * i;
        if (tile < ntiles) {
          double acc[KT];
          diag_tile<MODE, KT>(xr, PA + d0, PW2 + d0, PW1 + d0, tile * KT, K, DP, DC, acc);
#pragma unroll
          for (int j = 0; j < KT; ++j) dt[i][j] += acc[j];
        }
      }
    } else {
      for (int tile = w; tile < ntiles; tile += 4) {
        double acc[KT];
        diag_tile<MODE, KT>(xr, PA + d0, PW2 + d0, PW1 + d0, tile * KT, K, DP, DC, acc);
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int k = tile * KT + j;
          if (k < K && inb) {
            double* qp = qZ + (int64_t)k * ldq + row;
            *qp = (d0 == 0 ? crow[k] : *qp) + acc[j];  // running sum of the data term on top of c_jk
          }
        }
      }
    }
  }
  if (REG) {
#pragma unroll
    for (int i = 0; i < DIAG_MAXT; ++i) {
      const int tile = w + 4 * i;
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        lq[i][j] = NINF;
        const int k = tile * KT + j;
        if (tile < ntiles && k < K) {
          const double v = crow[k] + dt[i][j];
          lq[i][j] = v;
          mx = fmax(mx, v);
          if (raw && inb) qZ[(int64_t)k * ldq + row] = v;
        }
      }
    }
  } else if (!raw) {
    for (int tile = w; tile < ntiles; tile += 4)
      for (int j = 0; j < KT; ++j) {
        const int k = tile * KT + j;
        if (k < K && inb) mx = fmax(mx, qZ[(int64_t)k * ldq + row]);
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

// ===========================================================================
// The same E-step on the matrix pipe
// ===========================================================================
// Around a fixed centre mu_d (x' = x - mu, a' = a - mu) the separable log-likelihood is bilinear in the features
// phi(x) = [x'^2, x']:
//     sum_d w2 (x - a)^2 + w1 x  =  sum_d w2 x'^2 + (w1 - 2 w2 a') x'  +  sum_d (w2 a'^2 + w1 mu)
// i.e. one N x NF . NF x K product (NF = 2 DP, or DP when w2 == 0: the exponential family is linear in x) plus a
// per-cluster constant -- 2 NF MACs per (row, cluster) on v_mfma_f64_4x4x4_4b instead of 3 D fp64 VALU instructions,
// which leaves the kernel bound by HBM (X once, q once).  The expansion cancels where |x'| is large against a
// cluster's width, so lc_ctx.cpp takes this path only when max |w2| x'^2 over the clusters' 6-sigma ranges is small
// (the centre is the mean of the cluster centres); otherwise the difference form above runs (estep_diag_kernel).
// Layout as in estep_kernel: MFMA block b <-> rows 4b..4b+3 of a 16-row group,
//   A operand = tile W[cluster 4it+lo2][feature 4jt+hi] (LDS, staged once per persistent block: all K clusters fit),
//   B operand = phi[row = lane&15][4jt+hi] (registers),  D = y[cluster 4it+hi][row].
// log q~ of a row group lives in registers (K <= 4*KTM clusters: KTM values per lane); max / sum cross the four hi
// lanes with two shuffles.  Reference operation order of logsumexp and of the normalisation (probutils.cpp:141-150,
// cluster.cpp:130-131).
// KTM > 0: K <= 4 KTM and log q~ stays in registers (KTM x R values per lane); KTM == 0: any K, per-lane LDS slots.
// PLAIN: the instance the VBEM iterations use -- normalised responsibilities, no split-ordering term: `raw` and
// `ll_part` are compile-time constants there (they are tested per cluster tile and row group otherwise: several hundred
// scalar and exec-mask branches per tile of the generic instance)
// ONEGRP (with PLAIN): one group, known at compile time -- no row-group table.  With the table as a run-time choice the
// two ways of forming a row group's `info` end in the same register, and the compiler protects that register with a
// `s_waitcnt vmcnt(0)` in the table-less path too: a full drain (the previous tile's stores, the prefetch just issued) at
// the head of every tile.
// GRP: 0 = that instance; 1 (PLAIN too): several groups, the J x K table c_jk waits in LDS (J x K <= EDM_CT_CAP); 2: the table
// (if any) is read from global memory at the head of every cluster tile.
template <int NT, bool QUAD, int R, int KTM, bool PLAIN = false, int GRP = 2>
__global__ void __launch_bounds__(256, 2)
    estep_diag_mfma_kernel(const double* __restrict__ X, const double* __restrict__ Wt, const double* __restrict__ mu,
                           const double* __restrict__ constk, const double* __restrict__ ctab,
                           const int* __restrict__ rginfo_, double* __restrict__ qZ, double* __restrict__ fz_part,
                           double* __restrict__ ll_part_, int K, int64_t nrg, int64_t nrows, int64_t ldq, int raw_,
                           int64_t nslots, double* __restrict__ sink, int ngroups) {
  constexpr bool ONEGRP = GRP == 0, CTLDS = GRP == 1;
  static_assert(GRP == 2 || PLAIN, "the one-group and LDS-table instances are plain ones");
  const int* const rginfo = ONEGRP ? nullptr : rginfo_;
  const int raw = PLAIN ? 0 : raw_;
  double* const ll_part = PLAIN ? nullptr : ll_part_;
  constexpr int DP = NT * 4;
  constexpr int NTF = QUAD ? 2 * NT : NT;  // feature tiles per cluster tile
  constexpr bool REGS = KTM > 0;
  constexpr int NLQ = REGS ? KTM : 1;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int KT = (K + 3) / 4;
  double* wl = lds;                           // [KT][NTF][16]
  double* cst = wl + (size_t)KT * NTF * 16;   // [4 KT] per-cluster constant
  double* mul = cst + 4 * KT;                 // [DP] centre
  double* fzw = mul + DP;                     // [4]
  double* lls = fzw + 4;                      // [KT][256]: running q * data term per lane (split-ordering term)
  double* lqs = lls + (size_t)KT * 256;       // [KT][R][256] (KTM == 0): every lane's own log q~ slots
  double* ctl = lqs + (KTM == 0 ? (size_t)KT * R * 256 : 0);  // [EDM_CT_CAP] (GRP == 1): the table c_jk
  double* etab = ctl + (CTLDS ? EDM_CT_CAP : 0);              // [64]: 2^(j / 64) for exp_nonpos
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4;
  {
    const double2* src = reinterpret_cast<const double2*>(Wt);
    double2* dst = reinterpret_cast<double2*>(wl);
    for (int i = tid; i < KT * NTF * 8; i += 256) dst[i] = src[i];
    // One group (no row-group table): c_jk is one row of K values -- folded into the staged constants.  The table read
    // in tile_const is a vector load at the head of every cluster tile, and the vector-memory counter retires in order:
    // waiting for it is waiting for the next tile's rows issued just before it (an HBM round trip per cluster tile --
    // the kernel's "half of the wave time waiting").  With several groups the table is still read there.
    // (clusters past K -- the padding of the last cluster tile -- carry -inf: they drop out of max and sum by themselves)
    for (int i = tid; i < 4 * KT; i += 256) cst[i] = i < K ? constk[i] + (rginfo ? 0.0 : ctab[i]) : -INFINITY;
    for (int i = tid; i < DP; i += 256) mul[i] = QUAD ? mu[i] : 0.0;
    if constexpr (CTLDS)
      for (int i = tid; i < ngroups * K; i += 256) ctl[i] = ctab[i];
    for (int it = 0; it < KT; ++it) lls[it * 256 + tid] = 0.0;
    fill_exp_table(etab, tid, 256);
  }
  __syncthreads();
  const double* Pt = wl + (lane & 3) + 4 * hi;  // this lane's element of every 4x4 tile
  double* lqme = lqs + tid;
  double fz = 0.0;
  const int64_t ntile = (nrg + 4 * R - 1) / (4 * R);  // tiles of 4 waves x R row groups
  // the next tile's rows are in flight while the current one is computed (registers: R x NT fragments)
  double xn[R][NT];
  int infon[R];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // (wave-uniform: its arithmetic belongs on the scalar unit)
  const int lane_x = lo4 * DP + 4 * hi;                     // this lane's offset inside a row group's 16 x DP block
  auto fetch = [&](int64_t tile) {
    if constexpr (ONEGRP) {
      // one group: everything about a tile's row groups but the lane's own offset is uniform -- a scalar base per row
      // group and a 32-bit lane offset (the general form below costs ~ 40 VALU instructions of 64-bit address and
      // bounds arithmetic per row group and tile)
      const int64_t rg0 = (tile * 4 + wave_u) * R;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t rg = rg0 + r;
        const bool ok = tile < ntile && rg < nrg;
        const int64_t rem = nrows - rg * RG;
        infon[r] = !ok ? -1 : rem >= RG ? RG : (rem > 0 ? (int)rem : 0);
        const double* xr = X + (ok ? rg : 0) * (int64_t)(RG * DP) + lane_x;
#pragma unroll
        for (int q = 0; q < NT / 4; ++q) {
          const double2* p2 = reinterpret_cast<const double2*>(xr + 16 * q);
          const double2 v0 = p2[0], v1 = p2[1];
          xn[r][4 * q] = v0.x, xn[r][4 * q + 1] = v0.y, xn[r][4 * q + 2] = v1.x, xn[r][4 * q + 3] = v1.y;
        }
      }
      return;
    }
    const int64_t rg0 = (tile * 4 + wave) * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t rg = rg0 + r;
      const bool ok = tile < ntile && rg < nrg;
      int info = -1;  // -1: no such row group
      if (ok) {
        if (rginfo) {
          info = rginfo[rg];
        } else {
          const int64_t rem = nrows - rg * RG;
          info = rem >= RG ? RG : (rem > 0 ? (int)rem : 0);
        }
      }
      infon[r] = info;
      // feature tile jt = 4 q + jr stands for the columns {16 q + jr + 4 h}: lane hi holds columns 16 q + 4 hi + (0..3) --
      // four contiguous doubles per (row, q), two 16-byte loads, a row's 128-byte line consumed by its four hi lanes at
      // once (contiguous groups 4 jt + hi take 8-byte loads 32 bytes apart: four times the cache-line requests).  The
      // weight tiles are packed with the same column map (lc_ctx.cpp, estep_diag).
      const double* xr = X + ((ok ? rg : 0) * RG + lo4) * DP + 4 * hi;
#pragma unroll
      for (int q = 0; q < NT / 4; ++q) {
        const double2* p2 = reinterpret_cast<const double2*>(xr + 16 * q);
        const double2 v0 = p2[0], v1 = p2[1];
        xn[r][4 * q] = v0.x, xn[r][4 * q + 1] = v0.y, xn[r][4 * q + 2] = v1.x, xn[r][4 * q + 3] = v1.y;
      }
    }
  };
  static_assert(NT % 4 == 0, "column groups of sixteen");
  fetch(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t rg0 = (tile * 4 + wave) * R;
    double f[R][NT];  // x' (then x'^2: the linear half of the weights is applied first, the features are squared in place)
    int grp[R];
    bool rowok[R], rgok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      rgok[r] = infon[r] >= 0;
      const int info = rgok[r] ? infon[r] : 0;
      grp[r] = info >> 5;
      rowok[r] = lo4 < (info & 31);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) f[r][jt] = xn[r][jt];
    }
    if constexpr (QUAD) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const double m = mul[16 * (jt / 4) + 4 * hi + (jt % 4)];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r][jt] -= m;
      }
    }
    fetch(tile + gridDim.x);
    double mx[R], lq[NLQ][R];
#pragma unroll
    for (int r = 0; r < R; ++r) mx[r] = -INFINITY;
    // one cluster tile (4 clusters) and one half of the features: out += W_tile[half] . f for the wave's R row groups.
    // The weight-tile reads run PFH ahead of their MFMAs ACROSS cluster tiles (the ring is handed from one call to the
    // next: the last PFH reads of tile `it` are the first of tile it + 1 of the same half, NTF tiles further on), fenced
    // so that hipcc does not sink them to their uses -- a ring restarted per half tile exposed an LDS round trip sixteen
    // times per 32-row tile.  The accumulators start from `out` (the MFMA's C operand): no add behind the chain.
    constexpr int PFH = NT % 8 == 0 ? 8 : 4;  // a divisor of NT: the ring's slots line up from one cluster tile to the next
    static_assert(NT % PFH == 0 && PFH <= NT, "ring slots");
    auto ring_start = [&](int it, int half, double (&ring)[PFH]) {
      const double* Pi = Pt + ((size_t)it * NTF + (size_t)half * NT) * 16;
      static_for<PFH>([&](auto ic) { ring[ic] = Pi[ic * 16]; });
    };
    auto half_tile = [&](int it, int half, double (&out)[R], double (&ring)[PFH]) {
      const double* Pi = Pt + ((size_t)it * NTF + (size_t)half * NT) * 16;
      static_for<NT>([&](auto jc) {
        constexpr int jt = jc;
        const double v = ring[jt % PFH];
        constexpr int m = jt + PFH;  // (past this tile: the same half of the next cluster tile; past the last one: unused)
        ring[jt % PFH] = Pi[(m < NT ? m : NTF + (m - NT)) * 16];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = mfma4(v, f[r][jt], out[r]);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // constants of a cluster tile: E[log weight] + per-cluster constant; clusters past K (padding of the last tile)
    // drop out as -inf
    auto tile_const = [&](int it, double (&out)[R]) {
      const int k = 4 * it + hi;
      const double ck = cst[k];  // (-inf for k >= K)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if constexpr (ONEGRP) out[r] = ck;
        else if constexpr (CTLDS) out[r] = k < K ? ctl[grp[r] * K + k] + ck : -INFINITY;
        else out[r] = k < K ? (rginfo ? ctab[(int64_t)grp[r] * K + k] + ck : ck) : -INFINITY;
      }
    };
    auto tile_done = [&](int it, double (&v)[R]) {
      const int k = 4 * it + hi;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        mx[r] = fmax(mx[r], v[r]);  // (v is an MFMA result: see max_raw)
        if (raw && k < K && rgok[r]) qZ[(int64_t)k * ldq + (rg0 + r) * RG + lo4] = v[r];
      }
    };
    // Two cluster tiles at a time (REGS): four accumulation chains (2 tiles x R row groups).  With R = 2 chains an MFMA's
    // C operand is the result of the MFMA two instructions back, and hipcc fills the gap with `s_nop 1` behind every pair
    // (6 % of the stream); the second ring of weight reads costs 16 registers.
    // (two rings of PFP = 4 reads: a step is four MFMAs now, so four steps ahead is as far ahead in time as eight were)
    constexpr int PFP = 4;
    auto pair_start = [&](int it, int half, double (&r0)[PFP], double (&r1)[PFP]) {
      const double* P0 = Pt + ((size_t)it * NTF + (size_t)half * NT) * 16;
      static_for<PFP>([&](auto ic) { r0[ic] = P0[ic * 16], r1[ic] = P0[(NTF + ic) * 16]; });
    };
    auto half_pair = [&](int it, int half, double (&o0)[R], double (&o1)[R], double (&r0)[PFP], double (&r1)[PFP]) {
      const double* P0 = Pt + ((size_t)it * NTF + (size_t)half * NT) * 16;
      const double* P1 = P0 + (size_t)NTF * 16;
      static_for<NT>([&](auto jc) {
        constexpr int jt = jc;
        const double v0 = r0[jt % PFP], v1 = r1[jt % PFP];
        constexpr int m = jt + PFP;  // (past these tiles: the same half of the next PAIR; past the last one: unused)
        r0[jt % PFP] = P0[(m < NT ? m : 2 * NTF + (m - NT)) * 16];
        r1[jt % PFP] = P1[(m < NT ? m : 2 * NTF + (m - NT)) * 16];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) o0[r] = mfma4(v0, f[r][jt], o0[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) o1[r] = mfma4(v1, f[r][jt], o1[r]);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // the half of the weights that multiplies x' (QUAD: tiles NT..2NT-1; otherwise the only half)
    // (where it pays and fits: the VBEM iterations' instances and the one-row-group widths, whose single chain stalls the
    //  most; the instances that carry LL_k / raw output as well already spill without the second ring)
    constexpr bool PAIR = REGS && (PLAIN || R == 1);
    double ring[PFH], pr0[PFP], pr1[PFP];
    if constexpr (REGS && !PAIR) {
      ring_start(0, QUAD ? 1 : 0, ring);
#pragma unroll
      for (int it = 0; it < KTM; ++it) {
        if (it < KT) {  // block-uniform
          tile_const(it, lq[it]);
          half_tile(it, QUAD ? 1 : 0, lq[it], ring);
          if constexpr (!QUAD) tile_done(it, lq[it]);
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) lq[it][r] = -INFINITY;
        }
      }
    } else if constexpr (REGS) {
      static_assert(KTM % 2 == 0 && NT % PFP == 0, "cluster tiles in pairs");
      pair_start(0, QUAD ? 1 : 0, pr0, pr1);
#pragma unroll
      for (int it = 0; it < KTM; it += 2) {
        if (it + 1 < KT) {  // block-uniform
          tile_const(it, lq[it]);
          tile_const(it + 1, lq[it + 1]);
          half_pair(it, QUAD ? 1 : 0, lq[it], lq[it + 1], pr0, pr1);
          if constexpr (!QUAD) {
            tile_done(it, lq[it]);
            tile_done(it + 1, lq[it + 1]);
          }
        } else if (it < KT) {  // (an odd last tile, on its own ring)
          tile_const(it, lq[it]);
          ring_start(it, QUAD ? 1 : 0, ring);
          half_tile(it, QUAD ? 1 : 0, lq[it], ring);
          if constexpr (!QUAD) tile_done(it, lq[it]);
#pragma unroll
          for (int r = 0; r < R; ++r) lq[it + 1][r] = -INFINITY;
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) lq[it][r] = -INFINITY, lq[it + 1][r] = -INFINITY;
        }
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < KT; ++it) {
        tile_const(it, lq[0]);
        ring_start(it, QUAD ? 1 : 0, ring);  // (any K: the ring restarts per cluster tile here)
        half_tile(it, QUAD ? 1 : 0, lq[0], ring);
        if constexpr (!QUAD) tile_done(it, lq[0]);
        if (QUAD || !raw) {
#pragma unroll
          for (int r = 0; r < R; ++r) lqme[(it * R + r) * 256] = lq[0][r];
        }
      }
    }
    if constexpr (QUAD) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) f[r][jt] *= f[r][jt];
      if constexpr (REGS && !PAIR) {
        ring_start(0, 0, ring);
#pragma unroll
        for (int it = 0; it < KTM; ++it) {
          if (it < KT) {
            half_tile(it, 0, lq[it], ring);
            tile_done(it, lq[it]);
          }
        }
      } else if constexpr (REGS) {
        pair_start(0, 0, pr0, pr1);
#pragma unroll
        for (int it = 0; it < KTM; it += 2) {
          if (it + 1 < KT) {
            half_pair(it, 0, lq[it], lq[it + 1], pr0, pr1);
            tile_done(it, lq[it]);
            tile_done(it + 1, lq[it + 1]);
          } else if (it < KT) {
            ring_start(it, 0, ring);
            half_tile(it, 0, lq[it], ring);
            tile_done(it, lq[it]);
          }
        }
      } else {
#pragma unroll 1
        for (int it = 0; it < KT; ++it) {
#pragma unroll
          for (int r = 0; r < R; ++r) lq[0][r] = lqme[(it * R + r) * 256];
          ring_start(it, 0, ring);
          half_tile(it, 0, lq[0], ring);
          tile_done(it, lq[0]);
          if (!raw) {
#pragma unroll
            for (int r = 0; r < R; ++r) lqme[(it * R + r) * 256] = lq[0][r];
          }
        }
      }
    }
    if (raw) continue;
    // logsumexp (probutils.cpp:141-150): max, sum exp(x - max), log + max; q = exp(x - max) / sum = exp(x - logZ)
    // (cluster.cpp:130-131) with ONE exponential per entry -- e is kept (registers or the lane's slot) and scaled
    if constexpr (PLAIN && REGS) {
      // The VBEM iterations' instance: everything but the MFMAs is fp64 issue (an fp64 VALU instruction costs the matrix
      // pipe ~ 11 clocks), so the sweep spends as few as it can -- the table exponential (16 instructions
      // for 26), a Newton reciprocal (5 for 12), and ONE logarithm per lane instead of one per row group: log Z is only
      // needed for F_z, the four `hi` lanes of a row all hold its sum, so lane (lo4, hi) takes row group hi's.
      double mm[R], se[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double m = mx[r];
        m = max_raw(m, __shfl_xor(m, 16));
        m = max_raw(m, __shfl_xor(m, 32));
        mm[r] = m;
        se[r] = 0.0;
      }
      // cluster tiles in pairs: 2 R exponentials per basic block -- independent chains that fill each other's latency
      // (one block per (tile, row group) is a 21-instruction dependent chain with an LDS wait in the middle, at two waves
      // per SIMD).  A tile past the last one holds -inf (exp = 0): the odd tile of the last pair needs no test.
#pragma unroll
      for (int it = 0; it < KTM; it += 2) {
        if (it < KT) {  // block-uniform
#pragma unroll
          for (int j = 0; j < 2 && it + j < KTM; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const double e = exp_nonpos(lq[it + j][r] - mm[r], etab);
              se[r] += e;
              lq[it + j][r] = e;
            }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) se[r] = sum_over_hi(se[r]);
      {
        double sl = se[0], ml = mm[0];
        bool okl = rgok[0] && rowok[0];
#pragma unroll
        for (int r = 1; r < R; ++r)
          if (hi == r) sl = se[r], ml = mm[r], okl = rgok[r] && rowok[r];
        const double lz = log(sl) + ml;
        if (hi < R && okl) fz += lz;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double inv = rcp_pos(se[r]);
#pragma unroll
        for (int it = 0; it < KTM; ++it) {
          const int k = 4 * it + hi;
          const bool kin = k < K;
          double q = lq[it][r] * inv;
          if (!rowok[r] || !kin) q = 0.0;
          // (unconditional store: see the general path below)
          double* dst = kin && rgok[r] ? qZ + ((int64_t)k * ldq + (rg0 + r) * RG + lo4) : sink + tid;
          *dst = q;
        }
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double m = mx[r];
      m = fmax(m, __shfl_xor(m, 16));
      m = fmax(m, __shfl_xor(m, 32));
      double se = 0.0;
      if constexpr (REGS) {
#pragma unroll
        for (int it = 0; it < KTM; ++it) {
          if (it < KT) {
            const double e = exp(lq[it][r] - m);
            se += e;
            lq[it][r] = ll_part ? lq[it][r] : e;      // with the split-ordering term the log value is still needed
          }
        }
      } else {
        for (int it = 0; it < KT; ++it) {
          const double v = lqme[(it * R + r) * 256];
          const double e = exp(v - m);
          se += e;
          if (!ll_part) lqme[(it * R + r) * 256] = e;
        }
      }
      se = sum_over_hi(se);
      const double logZ = log(se) + m;
      const double inv = 1.0 / se;
      // The store is UNCONDITIONAL: lanes of padding clusters (k >= K, cluster tiles past the last one included) and of
      // row groups past the end write a zero to their slot of `sink`.  A store under a branch is one the compiler cannot
      // count on, and the vector-memory counter retires in order: the wait for the prefetched rows at the head of the
      // next tile then became "until at most the 8 newest operations are outstanding" -- i.e. until all 16 stores of
      // this tile had reached memory.  Counted, they stay in flight under the next tile's MFMAs.
      auto finish = [&](int it, double v) {
        const int k = 4 * it + hi;
        const bool kin = k < K;
        double q = ll_part ? exp(v - logZ) : v * inv;
        if (!rowok[r] || !kin) q = 0.0;
        double* dst = kin && rgok[r] ? qZ + ((int64_t)k * ldq + (rg0 + r) * RG + lo4) : sink + tid;
        *dst = q;
        if (ll_part && kin && q > 0.0) lls[it * 256 + tid] += q * (v - ctab[(int64_t)grp[r] * K + k]);
      };
      if constexpr (REGS) {
#pragma unroll
        for (int it = 0; it < KTM; ++it) finish(it, lq[it][r]);
      } else {
        for (int it = 0; it < KT; ++it) finish(it, lqme[(it * R + r) * 256]);
      }
      if (rgok[r] && rowok[r] && hi == 0) fz += logZ;
    }
  }
  if (raw) return;
  fz = wave_sum(fz);
  if (lane == 0) fzw[wave] = fz;
  if (ll_part) {
    for (int it = 0; it < KT; ++it) {
      const double v = sum_over_lo4(lls[it * 256 + tid]);  // cluster 4 it + hi, summed over the 16 rows' lanes
      if (lo4 == 0) lls[it * 256 + tid] = v;               // lanes 0, 16, 32, 48 of every wave hold the wave's sums
    }
  }
  __syncthreads();
  // partial slots are sized by the caller (nslots >= gridDim.x): this block fills its own and zeroes its share of the rest
  if (ll_part)
    for (int k = tid; k < K; k += 256) {
      const int it = k >> 2, h = k & 3;
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += lls[it * 256 + w * 64 + 16 * h];
      ll_part[(int64_t)blockIdx.x * K + k] = v;
      for (int64_t sl = blockIdx.x + gridDim.x; sl < nslots; sl += gridDim.x) ll_part[sl * K + k] = 0.0;
    }
  if (tid == 0) {
    fz_part[blockIdx.x] = -(fzw[0] + fzw[1] + fzw[2] + fzw[3]);
    for (int64_t sl = blockIdx.x + gridDim.x; sl < nslots; sl += gridDim.x) fz_part[sl] = 0.0;
  }
}

static size_t edm_lds_bytes(int NT, int NTF, int KT, int R, bool slots, bool table = false) {
  return ((size_t)KT * NTF * 16 + 4 * KT + 4 * NT + 4 + (size_t)KT * 256 + (slots ? (size_t)KT * R * 256 : 0) +
          (table ? EDM_CT_CAP : 0) + 64) * sizeof(double);
}

template <int NT, bool QUAD, int R, int KTM>
static hipError_t launch_edm_k(const DiagEstepLaunch& a, hipStream_t stream) {
  const int KT = (a.K + 3) / 4, NTF = QUAD ? 2 * NT : NT;
  size_t shmem = edm_lds_bytes(NT, NTF, KT, R, KTM == 0);
#ifndef LC_EDM_PLAIN_QUAD
#define LC_EDM_PLAIN_QUAD 1  // (with unconditional stores the plain instance wins for both: linear features 10 %, with the quadratic half 6.5 %; it lost 4 % there before)
#endif
  const bool plain = KTM > 0 && !a.raw && !a.ll_part && (!QUAD || LC_EDM_PLAIN_QUAD);
  const bool onegrp = plain && !a.rginfo;
  const bool tabled = plain && a.rginfo && (int64_t)a.ngroups * a.K <= EDM_CT_CAP &&
                      edm_lds_bytes(NT, NTF, KT, R, KTM == 0, true) <= 150 * 1024;
  if (tabled) shmem = edm_lds_bytes(NT, NTF, KT, R, KTM == 0, true);
  if (!a.sink) return hipErrorInvalidValue;
  auto kern = onegrp   ? estep_diag_mfma_kernel<NT, QUAD, R, KTM, (KTM > 0), (KTM > 0 ? 0 : 2)>
              : tabled ? estep_diag_mfma_kernel<NT, QUAD, R, KTM, (KTM > 0), (KTM > 0 ? 1 : 2)>
              : plain  ? estep_diag_mfma_kernel<NT, QUAD, R, KTM, (KTM > 0)>
                       : estep_diag_mfma_kernel<NT, QUAD, R, KTM>;
  static LdsGrant grants[4];
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grants[onegrp ? 3 : tabled ? 2 : plain ? 1 : 0]);
      e != hipSuccess)
    return e;
  const int64_t ntile = (a.nrg + 4 * R - 1) / (4 * R);
  const int64_t nslots = estep_diag_grid(a.nrg);
  const int cus = current_device_cus();
  // persistent blocks (the weights are staged once per block): as many as are resident, a few tiles each
  const int per_cu = shmem <= 80 * 1024 ? 2 : 1;
  int64_t grid = std::min<int64_t>(std::min<int64_t>(ntile, nslots), (int64_t)cus * per_cu);
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), shmem, stream, a.X, a.wt, a.mu, a.constk, a.ctab, a.rginfo,
                     a.qZ, a.fz_part, a.ll_part, a.K, a.nrg, a.nrows, a.ldq, a.raw, nslots, a.sink, a.ngroups);
  return hipGetLastError();
}

template <int NT, bool QUAD>
static hipError_t launch_edm_q(const DiagEstepLaunch& a, hipStream_t stream) {
  // row groups per wave by the register budget (R x NTF feature fragments of two registers each)
  constexpr int R = NT <= 16 ? 2 : 1;  // two sets of R x NT fragments live (current tile, next tile in flight)
  if (a.K <= 32) return launch_edm_k<NT, QUAD, R, 8>(a, stream);
  return launch_edm_k<NT, QUAD, R, 0>(a, stream);
}

template <int NT>
static hipError_t launch_edm_t(const DiagEstepLaunch& a, hipStream_t stream) {
  return a.mode == 2 ? launch_edm_q<NT, false>(a, stream) : launch_edm_q<NT, true>(a, stream);
}

// doubles of the packed weight tiles lc_ctx.cpp uploads for the matrix-pipe path; 0: this shape has no such path
int64_t estep_diag_mfma_weights(int DP, int K, int mode) {
  if (DP > 128 || DP % 16 || K < 1) return 0;
  const int NT = DP / 4, NTF = mode == 2 ? NT : 2 * NT, KT = (K + 3) / 4;
  const int R = NT <= 16 ? 2 : 1;
  // the weights of all clusters, every lane's split-ordering sums and (K > 32) its log q~ slots must fit in the CU's LDS
  if (edm_lds_bytes(NT, NTF, KT, R, K > 32) > 150 * 1024) return 0;
  return (int64_t)KT * NTF * 16;
}

static hipError_t launch_estep_diag_mfma(const DiagEstepLaunch& a, hipStream_t stream) {
  switch (a.DP) {
    case 16: return launch_edm_t<4>(a, stream);
    case 32: return launch_edm_t<8>(a, stream);
    case 48: return launch_edm_t<12>(a, stream);
    case 64: return launch_edm_t<16>(a, stream);
    case 80: return launch_edm_t<20>(a, stream);
    case 96: return launch_edm_t<24>(a, stream);
    case 112: return launch_edm_t<28>(a, stream);
    case 128: return launch_edm_t<32>(a, stream);
  }
  return hipErrorInvalidValue;
}

template <int MODE, int KT, bool REG>
static hipError_t launch_ed_t(const DiagEstepLaunch& a, int64_t grid, size_t shmem, hipStream_t stream) {
  auto kern = a.DP > 128 ? estep_diag_wide_kernel<MODE, KT, REG> : estep_diag_kernel<MODE, KT, REG>;
  static LdsGrant grants[2];  // per kernel (narrow, wide)
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grants[a.DP > 128 ? 1 : 0]); e != hipSuccess)
    return e;
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
  if (a.wt) return launch_estep_diag_mfma(a, stream);
  const int DC = a.DP < 128 ? a.DP : 128;
  const size_t shmem = (size_t)(64 * (DC + 1) + 256 + a.K) * sizeof(double);
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
// RS (row classes: 1, 2 or 4) is a template parameter so that a wave's steps of a batch are a compile-time list: the
// operands of its next step (NB x fragments, CT q fragments) are read while the current step's MFMAs issue (fenced;
// hipcc reads them right in front of their use otherwise, and every step of 32 MFMAs started with an exposed LDS round
// trip: DGMM statistics 2.05 ms at 58 % of the pipe).
template <int DP, bool SECOND, int RS>
__global__ void __launch_bounds__(256, 2) suffstat_diag_kernel(DiagStatLaunch a) {
  constexpr int NB = DP / 16, CT = 4;
  constexpr int BR = DP <= 64 ? 32 : 16;
  constexpr int LD = lds_row_stride(DP), XBUF = BR * LD;
  constexpr int NV2 = BR * DP / 2, NPRE = (NV2 + 255) / 256;   // double2 per thread and batch
  constexpr int QCAP = SD_QMAX / RS;                           // clusters a block with RS row classes can hold (suffstat_diag_rsplit)
  constexpr int NQ = QCAP * BR / 256;                          // q elements per thread and batch
  static_assert(NQ >= 1 && NQ * 256 == QCAP * BR, "whole q columns per thread");
  constexpr int QLD = BR + 4;                                  // padded q column stride: conflict-free A fragments
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* xbuf = lds;                // [2][BR][LD]
  double* qbuf = lds + 2 * XBUF;     // [2][SD_QMAX][QLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo2 = lane & 3, blk = (lane >> 2) & 3, hi = lane >> 4, lo4 = lane & 15;
  const int K = a.K;
  const int chunk = blockIdx.x / a.nslice, slice = blockIdx.x % a.nslice;
  const int kb0 = slice * SD_QMAX;
  const int kc = (K - kb0) < SD_QMAX ? (K - kb0) : SD_QMAX;     // clusters of this block
  const int group = RS == 1 ? wave : RS == 2 ? (wave & 1) : 0;  // 16-cluster group of this wave
  const int rcls = RS == 1 ? 0 : RS == 2 ? (wave >> 1) : wave;  // row class (steps st = rcls mod RS)
  const bool active = group * 16 < kc;
  const int64_t r0 = (int64_t)chunk * a.chunk_rows;
  const int64_t r1 = (r0 + a.chunk_rows) < a.NP ? (r0 + a.chunk_rows) : a.NP;

  double acc1[CT][NB], acc2[CT][NB];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) acc1[c][jb] = acc2[c][jb] = 0.0;

  // Two batches in flight (DP <= 96: the second register set spills next to 128 accumulator registers beyond): with
  // one, a block asked for batch b + 1 at the top of batch b and needed it ~ 1 us of MFMAs later -- about one HBM round
  // trip under load, so the kernel sat at 3.8 TB/s waiting for its single batch.  Set (b & 1) now receives batch b + 2
  // while batch b is multiplied out of LDS and batch b + 1 moves from the other set into the other LDS buffer; the batch
  // loop is unrolled twice so that set and buffer are compile-time names (registers, not scratch).
  constexpr bool DEEP = DP <= 96;
  constexpr int NSET = DEEP ? 2 : 1, AHEAD = DEEP ? 2 : 1;
  double pre[NSET][NPRE][2], qpre[NSET][NQ];
  // N_k is summed where q is staged (NQ adds per thread and batch) and not next to the MFMAs, where the A fragment is
  // replicated over the four MFMA blocks and every lane added CT values per 4-row step: four times the additions, at
  // the fp64 VALU's price in matrix-pipe time (16 of a wave's 48 non-MFMA fp64 instructions per batch at D = 64)
  // (where NQ more registers would spill -- a block of more than 32 clusters at D >= 64, D >= 112 -- the sum stays in the
  //  step loop)
  constexpr bool NSTAGE = DP <= 96 && !(DP >= 64 && RS == 1);
  double nq[NSTAGE ? NQ : 1], nacc[NSTAGE ? 1 : CT];
#pragma unroll
  for (int i = 0; i < (NSTAGE ? NQ : 1); ++i) nq[i] = 0.0;
#pragma unroll
  for (int c = 0; c < (NSTAGE ? 1 : CT); ++c) nacc[c] = 0.0;
  // Staging addresses = a UNIFORM base per batch (scalar registers) + per-lane offsets that are fixed for the whole
  // chunk.  Formed inside the batch loop -- (b0 + row) * ldx, (kb0 + kk) * ldq + b0 + r, idx / BR, the bounds -- they
  // were ~ 150 VALU instructions per thread and batch, 37 per 32-MFMA step: the kernel sat at 59 % of the pipe for them
  // (PMC: 146 M non-MFMA VALU instructions per launch at D = 64, K = 32 against 160 M MFMAs).
  int xoff[NPRE], xlds[NPRE], xrow[NPRE];
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int idx = tid + i * 256;  // double2 index inside the batch, row-major [BR][DP/2]
    const int row = idx / (DP / 2), c2 = idx % (DP / 2);
    xrow[i] = idx < NV2 ? row : BR;  // (BR: never inside a batch)
    xoff[i] = row * (int)a.ldx + 2 * c2;
    xlds[i] = row * LD + 2 * c2;
  }
  static_assert(256 % BR == 0, "a thread stages the same row of every q column it takes");
  const int qr = tid % BR, qk0 = tid / BR;                  // q element i of this thread: cluster qk0 + i * (256 / BR), row qr
  const int64_t qoff = (int64_t)qk0 * a.ldq + qr;
  const int qlds0 = qk0 * QLD + qr;
  // The loads of a batch are straight-line code without a use of what they return: a load under a branch (a row bound, a
  // cluster bound, "is there another batch") makes the number of loads in flight a run-time quantity, and a select on a
  // loaded value in front of the MFMAs is a wait for it -- either way the compiler ends up at `s_waitcnt vmcnt(0)` before
  // the older set is used: one batch in flight again.  So every load is issued always -- a lane outside the batch (or a
  // batch outside the chunk: left <= 0) reads a valid stand-in address -- and the zeros for such lanes are chosen when the
  // set is stored to LDS, a batch later.
  auto rows_left = [&](int64_t b0) {
    const int64_t left64 = r1 - b0;
    return left64 < BR ? (left64 > 0 ? (int)left64 : 0) : BR;  // rows of this batch inside the chunk
  };
  // Whole batches (all but the last ones of a chunk) are loaded without any per-batch address arithmetic on the vector
  // unit: X as  uniform base + a 32-bit byte offset fixed per lane, q as  uniform base + a 64-bit offset fixed per lane
  // that already holds the stand-in of a lane without a cluster (one add per load).  Before this the selects and shifts
  // of the general form were 40 of a wave's 50 32-bit VALU instructions per batch, next to 128 MFMAs.
  // (Not where the offsets' registers would spill: the instances that keep N_k in the step loop.)
  constexpr bool WHOLE = NSTAGE;
  unsigned xbyte[WHOLE ? NPRE : 1];
  int64_t qfix[WHOLE ? NQ : 1];  // element offset from qZ + (first row of the batch)
  if constexpr (WHOLE) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) xbyte[i] = xrow[i] < BR ? (unsigned)xoff[i] * 8u : 0u;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
      qfix[i] = qk0 + i * (256 / BR) < kc ? (int64_t)(kb0 + i * (256 / BR)) * a.ldq + qoff : (int64_t)qr;
  }
  auto gload = [&](auto whole, auto sc, int64_t b0) {
    constexpr int S = decltype(sc)::value;
    if constexpr (WHOLE && decltype(whole)::value) {
      const char* xc = reinterpret_cast<const char*>(a.X + b0 * a.ldx + a.col0);  // uniform
      const double* qc = a.qZ + b0;                                                // uniform
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const double2 v = *reinterpret_cast<const double2*>(xc + xbyte[i]);
        pre[S][i][0] = v.x;
        pre[S][i][1] = v.y;
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) qpre[S][i] = qc[qfix[i]];
    } else {
      const int left = rows_left(b0);
      const double* xb = a.X + (left > 0 ? b0 * a.ldx : 0) + a.col0;  // uniform
      const int64_t qb0 = left > 0 ? b0 : 0;                          // uniform
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const double2 v = *reinterpret_cast<const double2*>(xb + (xrow[i] < left ? xoff[i] : 0));
        pre[S][i][0] = v.x;
        pre[S][i][1] = v.y;
      }
      const bool qrow = qr < left;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        // (an offset chosen per lane, not a pointer: hipcc turns `*(in ? p : q)` with a uniform q into a branch around a
        //  scalar load of *q and a copy into the destination -- which waits for every vector load in flight)
        const bool in = qrow && qk0 + i * (256 / BR) < kc;
        // (the stand-in stays inside the smallest buffer there is -- 16 padded rows of one cluster -- and differs by lane)
        const int64_t off = in ? (int64_t)(kb0 + i * (256 / BR)) * a.ldq + qb0 + qoff : (int64_t)(qr & 15);
        qpre[S][i] = a.qZ[off];
      }
    }
  };
  // (b0: first row of the batch being stored -- the sparse mask is looked up here, where its two dependent loads delay
  //  nothing but the sparse runs themselves)
  auto lstore = [&](auto sc, int buf, int64_t b0) {
    constexpr int S = decltype(sc)::value;
    const int left = rows_left(b0);
    double* xb = xbuf + buf * XBUF;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      // (a compile-time "always" wherever it is one: a set register whose store stands under a branch counts as never
      //  waited for, and the next write to it -- hipcc reuses them as address temporaries -- waits for everything.
      //  A row past the end of the chunk keeps what its stand-in address held -- observations of the chunk's first rows --
      //  and its q is zero: 0 * x adds nothing as long as x is finite.  A non-finite observation poisons the statistics
      //  of every cluster here exactly as it does in the reference, where its responsibilities are NaN in every column:
      //  cluster.cpp:120-131 on a NaN / Inf row)
      if ((i + 1) * 256 <= NV2 || xrow[i] < BR)
        *reinterpret_cast<double2*>(xb + xlds[i]) = make_double2(pre[S][i][0], pre[S][i][1]);
    }
    const bool qrow = qr < left;
    int64_t g = 0;
    if (a.smask && qrow) g = a.rginfo[(b0 + qr) >> 4] >> 5;
    double* qb = qbuf + buf * SD_QMAX * QLD + qlds0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int kk = qk0 + i * (256 / BR);
      bool in = qrow && kk < kc;
      if (a.smask && in) in = a.smask[g * K + kb0 + kk] != 0;
      const double q = in ? qpre[S][i] : 0.0;
      qb[i * (256 / BR) * QLD] = q;
      if constexpr (NSTAGE) nq[i] += q;  // N_k: every responsibility passes through exactly one thread's hands here
    }
  };
  // the MFMAs of one batch out of LDS buffer `buf`
  auto multiply = [&](int buf) {
    const double* xb = xbuf + buf * XBUF + hi * LD + 4 * blk + lo2;
    const double* qb = qbuf + buf * SD_QMAX * QLD + (group * 16 + lo2) * QLD + hi;
    // rows past the chunk end were staged as zeros with q = 0, so every step runs: steps rcls, rcls + RS, ...
    constexpr int NST = BR / 4 / RS;
    const double* xs = xb + rcls * 4 * LD;
    const double* qs = qb + rcls * 4;
    constexpr bool PIPE = DP <= 96;  // (the second operand set spills next to 128 accumulator registers at D = 112, 128)
    double xf[PIPE ? 2 : 1][NB], qv[PIPE ? 2 : 1][CT];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) xf[0][jb] = xs[16 * jb];
#pragma unroll
    for (int c = 0; c < CT; ++c) qv[0][c] = qs[4 * c * QLD];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int cur = PIPE ? (i & 1) : 0, nxt = PIPE ? (cur ^ 1) : 0;
      if (PIPE && i + 1 < NST) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) xf[nxt][jb] = xs[(i + 1) * RS * 4 * LD + 16 * jb];
#pragma unroll
        for (int c = 0; c < CT; ++c) qv[nxt][c] = qs[4 * c * QLD + (i + 1) * RS * 4];
      }
      if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
      double x2[NB];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
        if (SECOND) x2[jb] = xf[cur][jb] * xf[cur][jb];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const double q = qv[cur][c];
        if constexpr (!NSTAGE) nacc[c] += q;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          acc1[c][jb] = mfma4(q, xf[cur][jb], acc1[c][jb]);
          if (SECOND) acc2[c][jb] = mfma4(q, x2[jb], acc2[c][jb]);
        }
      }
      if constexpr (PIPE) {
        __builtin_amdgcn_sched_barrier(0);
      } else if (i + 1 < NST) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) xf[0][jb] = xs[(i + 1) * RS * 4 * LD + 16 * jb];
#pragma unroll
        for (int c = 0; c < CT; ++c) qv[0][c] = qs[4 * c * QLD + (i + 1) * RS * 4];
      }
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, NSET - 1>;

  using Whole = std::true_type;
  using Any = std::false_type;
  if (r0 < r1) {
    gload(Any{}, Set0{}, r0);
    lstore(Set0{}, 0, r0);
    if constexpr (DEEP) gload(Any{}, Set1{}, r0 + BR);
  }
  __syncthreads();
  // batch b0 sits in LDS buffer `buf`; batch b0 + BR is in flight into (DEEP) or about to be asked for from (otherwise)
  // the registers; `into` receives batch b0 + AHEAD BR, `from` is stored to the other LDS buffer behind the MFMAs.
  // No branch around the loads or the stores, and the loops take batches in pairs with the odd one behind them: two
  // paths that meet with different numbers of loads in flight leave the compiler one safe count for the next wait --
  // zero.  (A batch past the end of the chunk is loaded from stand-in addresses and stored as zeros that nobody reads.)
  auto batch = [&](auto whole, auto into, auto from, int64_t b0, int buf) {
    gload(whole, into, b0 + AHEAD * BR);
    if (active) multiply(buf);
    lstore(from, buf ^ 1, b0 + BR);
    __syncthreads();
  };
  int64_t b0 = r0;
  if constexpr (WHOLE) {
    for (; b0 + (AHEAD + 2) * BR <= r1; b0 += 2 * BR) {  // both batches this pair asks for are whole
      batch(Whole{}, Set0{}, Set1{}, b0, 0);
      batch(Whole{}, Set1{}, Set0{}, b0 + BR, 1);
    }
  }
  for (; b0 + BR < r1; b0 += 2 * BR) {
    batch(Any{}, Set0{}, Set1{}, b0, 0);
    batch(Any{}, Set1{}, Set0{}, b0 + BR, 1);
  }
  if (b0 < r1 && active) multiply(0);
  // record layout: [N_k | x_s (DPT) | xx_s (DPT)]; a launch over the column block [col0, col0 + DP) fills its part
  const int64_t SS = 1 + 2 * (int64_t)a.DPT;
  // N_k of the chunk: the BR threads that staged a cluster's rows sit in consecutive lanes; the sum goes to the record of
  // row class 0, the other row classes' records carry zero (the fold adds all of them)
  if (NSTAGE && a.col0 == 0) {
#pragma unroll
    for (int i = 0; i < (NSTAGE ? NQ : 1); ++i) {
      double v = nq[i];
#pragma unroll
      for (int m = BR / 2; m > 0; m >>= 1) v += __shfl_xor(v, m);
      const int kk = qk0 + i * (256 / BR);
      if (qr == 0 && kk < kc) {
#pragma unroll
        for (int rc = 0; rc < RS; ++rc) a.partial[((int64_t)(chunk * RS + rc) * K + kb0 + kk) * SS] = rc == 0 ? v : 0.0;
      }
    }
  }
  if (!active) return;

  // output lane (lo2, blk, hi) of accumulator (c, jb): cluster 4 c + hi, dimension 16 jb + 4 blk + lo2
  double* rec = a.partial + ((int64_t)(chunk * RS + rcls) * K + kb0 + group * 16) * SS;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if constexpr (!NSTAGE) {
      // N_k: this lane summed q[row class hi][cluster 4 c + lo2] (replicated over blk)
      const double n = sum_over_hi(nacc[c]);
      if (a.col0 == 0 && hi == 0 && blk == 0 && group * 16 + 4 * c + lo2 < kc) rec[(int64_t)(4 * c + lo2) * SS] = n;
    }
    if (group * 16 + 4 * c + hi < kc) {
      double* out = rec + (int64_t)(4 * c + hi) * SS;
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        out[1 + a.col0 + 16 * jb + lo4] = acc1[c][jb];
        out[1 + a.DPT + a.col0 + 16 * jb + lo4] = SECOND ? acc2[c][jb] : 0.0;
      }
    }
  }
}

int suffstat_diag_rsplit(int K) {  // row classes per block: waves left over by the cluster groups split the rows
  const int groups = ((K < SD_QMAX ? K : SD_QMAX) + 15) / 16;
  return groups >= 3 ? 1 : groups == 2 ? 2 : 4;
}

template <int DP, int RS>
static hipError_t launch_sd_r(const DiagStatLaunch& a, hipStream_t stream) {
  constexpr int BR = DP <= 64 ? 32 : 16;
  const size_t shmem = (size_t)(2 * BR * lds_row_stride(DP) + 2 * SD_QMAX * (BR + 4)) * sizeof(double);
  static LdsGrant grants[2];
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(suffstat_diag_kernel<DP, true, RS>), shmem, grants[0]);
      e != hipSuccess)
    return e;
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(suffstat_diag_kernel<DP, false, RS>), shmem, grants[1]);
      e != hipSuccess)
    return e;
  const dim3 grid((unsigned)(a.nchunks * a.nslice));
  if (a.second)
    hipLaunchKernelGGL((suffstat_diag_kernel<DP, true, RS>), grid, dim3(256), shmem, stream, a);
  else
    hipLaunchKernelGGL((suffstat_diag_kernel<DP, false, RS>), grid, dim3(256), shmem, stream, a);
  return hipGetLastError();
}
template <int DP>
static hipError_t launch_sd_t(const DiagStatLaunch& a, hipStream_t stream) {
  return a.rsplit == 1 ? launch_sd_r<DP, 1>(a, stream) : a.rsplit == 2 ? launch_sd_r<DP, 2>(a, stream) : launch_sd_r<DP, 4>(a, stream);
}

hipError_t launch_suffstat_diag(const DiagStatLaunch& a0, hipStream_t stream) {
  if (a0.K <= 0 || a0.nchunks <= 0) return hipSuccess;
  if (a0.chunk_rows % 32) return hipErrorInvalidValue;
  DiagStatLaunch a = a0;
  a.nslice = (a.K + SD_QMAX - 1) / SD_QMAX;
  a.rsplit = suffstat_diag_rsplit(a.K);
  a.ldx = a0.DP;
  a.DPT = a0.DP;
  if (a0.DP > 128) {  // wide observations: one launch per block of 128 columns (q is re-read; X is read once in total)
    if (a0.DP % 64) return hipErrorInvalidValue;
    for (int c0 = 0; c0 < a0.DP; c0 += 128) {
      a.col0 = c0;
      a.DP = a0.DP - c0 < 128 ? 64 : 128;  // (a multiple of 64: the last block may be a half one)
      hipError_t e = a.DP == 128 ? launch_sd_t<128>(a, stream) : launch_sd_t<64>(a, stream);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  switch (a.DP) {
    case 16:
      return launch_sd_t<16>(a, stream);
    case 32:
      return launch_sd_t<32>(a, stream);
    case 48:
      return launch_sd_t<48>(a, stream);
    case 64:
      return launch_sd_t<64>(a, stream);
    case 80:
      return launch_sd_t<80>(a, stream);
    case 96:
      return launch_sd_t<96>(a, stream);
    case 112:
      return launch_sd_t<112>(a, stream);
    case 128:
      return launch_sd_t<128>(a, stream);
  }
  return hipErrorInvalidValue;
}


}  // namespace lck
