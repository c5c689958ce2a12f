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
#include "lc_device.hpp"

// The cluster loop's schedule is fixed by hand (DESIGN 4.1 has the measurements of every alternative; git keeps their
// code): a scheduling fence after every step of the LDS parameter stream (hipcc otherwise sinks the ring's reads to
// just before their use), ES_PF reads in flight ahead of their use, the squares of a tile row taken under the next
// row's first MFMAs (two accumulator sets), c_jk fetched before the tile stream, the next cluster's record brought
// from global memory straight into the other LDS buffer (global_load_lds_dwordx4: no staging registers, no ds_write),
// a cluster's log q~ stored at the top of the NEXT cluster's pass (the s_waitcnt vmcnt(0) in front of the barrier then
// finds the stores long retired), and -- dense k-sliced scheme -- c_jk loads / log q~ stores in the scalar-base form.

namespace lck {

constexpr int ES_PF = 4;  // LDS reads in flight ahead of their use (2...10 are within 1 %)

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
// diagonal), double-buffered in LDS; one tile read feeds R MFMAs.  MFMA block
// b <-> rows 4b..4b+3 of the row-group, so
//   A operand = tile A_k[4it+lo2][4jt+hi]   (same for all 4 blocks)
//   B operand = x[row = lane&15][4jt + hi]
//   D         = y[4it+hi][row = lane&15]
// log q~ is written to the qZ buffer as scratch (or waits in LDS, lq_lds), then
// normalised in place in the same arithmetic order as the reference: max,
// sum exp(x-max), log+max, exp(x - logZ).
// blocks per CU the register budget is set for.  D = 64: three row groups per wave (96 VGPRs of X fragments) at
// three waves per SIMD measured 22.2 ms against 22.6 for four row groups at two waves per SIMD (N=10M, K=32).
template <int DP, int R>
struct EstepOcc { static constexpr int BLOCKS = DP == 64 && R == 3 ? 3 : 2; };

// DC <= DP: the active width (estep_active_width, lc_kernels.h) -- the row stride of X stays DP, the tile rows walked and
// the parameter record are those of DC columns.
template <int DP, int DC, int R, int WAVES, bool SPARSE>
__global__ void __launch_bounds__(WAVES * 64, (EstepOcc<DP, R>::BLOCKS)) estep_kernel(EstepLaunch a) {
  static_assert(DC % 4 == 0 && DC <= DP && DC > DP - 16, "active width");
  constexpr int NT = DC / 4;
  constexpr int NTILES = NT * (NT + 1) / 2;
  constexpr int NREAD = NTILES + NT;  // LDS reads per cluster
  constexpr int PF = ES_PF;
  constexpr int PS = NTILES * 16 + DC;
  constexpr int NTHR = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* pbuf = lds;              // [2][PS]
  double* llw = lds + 2 * PS;      // [WAVES][K]
  double* fzw = llw + WAVES * a.K; // [WAVES]
  // lq_lds (four row groups per wave, where every lane owns one row outright, and K small enough): log q~ waits in
  // LDS for the normalisation, [K][threads], instead of making a round trip through the qZ buffer
  double* etab = fzw + WAVES;      // [64]: 2^(j / 64) for exp_nonpos (the normalisation sweep)
  double* lql = etab + 64;
  const bool lqm = R == 4 && a.lq_lds != 0;
  int* klist = reinterpret_cast<int*>(lql + (lqm ? (size_t)a.K * NTHR : 0));  // sparse mode: [K] active clusters of this block, [K] flags,
  int* kflag = klist + a.K;                          // [WAVES*R] groups of the block's row groups, [1] count
  int* bgrp = kflag + a.K;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4;
  const int K = a.K;
  const int64_t rg0 = ((int64_t)blockIdx.x * WAVES + wave) * R;
  fill_exp_table(etab, tid, NTHR);  // (read after the cluster loop's barriers)
  constexpr bool ROWLANES = R == 4;  // four row groups per wave: lane (lo4, hi) can own row group hi outright

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
  bool myok = false;  // ROWLANES: this lane's row group exists
  int owngrp = 0;     // ROWLANES: ... and its group
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (hi == r) myok = rgok[r], owngrp = grp[r];
  // ROWLANES: selector operands of the epilogue's chain (A[i][k] = 1 for i == r: D[i = hi][j] += sum_k d2_r[k][j] lands in
  // the lanes of row group hi == r only) -- four registers for the whole kernel, as fused_small_kernel keeps them
  double selA[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    selA[r] = (lane & 3) == r ? 1.0 : 0.0;
    asm volatile("" : "+v"(selA[r]));
  }

  // The next cluster's record, global -> LDS without staging registers: 16 bytes per lane and instruction land at LDS
  // address M0 + 16 * lane; a round of the block moves NTHR * 16 bytes.  M0 is written here behind the compiler's back (it
  // refuses M0 in a clobber list: "reserved register"); gfx950 code needs M0 for nothing else in this kernel, and
  // tools/check_isa.py asserts exactly that on the generated ISA: every instruction that names m0 is one of these pairs.
  const unsigned dvoff = (unsigned)(wave * 1024 + lane * 16);
  const unsigned dlds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pbuf + (unsigned)(wave * 1024));
  auto dma_record = [&](int kk, int bb) {
    const char* src = reinterpret_cast<const char*>(a.params + (int64_t)__builtin_amdgcn_readfirstlane(kk) * PS);
#pragma unroll
    for (int i = 0; i < (PS * 8 + NTHR * 16 - 1) / (NTHR * 16); ++i) {
      if (dvoff + i * (NTHR * 16) < (unsigned)(PS * 8))
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
                     ::"v"(dvoff), "s"(src + i * (NTHR * 16)), "s"(dlds0 + bb * (PS * 8) + i * (NTHR * 16)) : "memory");
    }
  };

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
      if constexpr (ROWLANES) {
        if (lqm) lql[k * NTHR + tid] = -INFINITY;
        else if (myok) a.qZ[(int64_t)k * a.ldq + (rg0 + hi) * RG + lo4] = -INFINITY;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (rgok[r] && hi == (k & 3)) a.qZ[(int64_t)k * a.ldq + (rg0 + r) * RG + lo4] = -INFINITY;
      }
    }
  }

  if (nact > 0) {
    dma_record(SPARSE ? klist[0] : 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  double mx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mx[r] = -INFINITY;

  // SADDR (dense k-sliced scheme): c_jk loads and log q~ stores in the scalar-base form (global_load / store ...
  // voffset, saddr): the cluster index lives in SGPRs, no 64-bit VALU address arithmetic in the loop
  constexpr bool SADDR = !SPARSE && !ROWLANES;
  unsigned coff[R];              // SADDR: byte offset of the c_jk row of row group r's group (J K 8 < 4 GB)
  const unsigned qoff = lo4 * 8u;  // SADDR: this lane's byte offset inside a 16-row piece of a qZ column
  // SADDR: byte address of the wave's first row in column 0 of qZ (wave-uniform)
  const char* qwave = reinterpret_cast<const char*>(a.qZ) +
                      (((int64_t)blockIdx.x * WAVES + __builtin_amdgcn_readfirstlane(wave)) * R * RG) * 8;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    coff[r] = (unsigned)grp[r] * (unsigned)K * 8u;
    if constexpr (SADDR) asm volatile("" : "+v"(coff[r]));  // (kept: hipcc would redo the multiply in every pass of the loop)
  }
  double lqprev[R];
  int kprev = -1;
  auto flush_lq = [&]() {  // the previous cluster's log q~: stored one pass late (see the top of the file)
    if constexpr (SADDR) {
      if (kprev >= 0) {
        const char* qk = qwave + (int64_t)kprev * a.ldq * 8;
        const bool mine = hi == (kprev & 3);
        if (rgok[0] && mine) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(qoff), "v"(lqprev[0]), "s"(qk) : "memory");
        if constexpr (R > 1) {
          if (rgok[1] && mine) asm volatile("global_store_dwordx2 %0, %1, %2 offset:128" ::"v"(qoff), "v"(lqprev[1]), "s"(qk) : "memory");
        }
        if constexpr (R > 2) {
          if (rgok[2] && mine) asm volatile("global_store_dwordx2 %0, %1, %2 offset:256" ::"v"(qoff), "v"(lqprev[R > 2 ? 2 : 0]), "s"(qk) : "memory");
        }
      }
    } else if constexpr (ROWLANES) {
      if (kprev >= 0 && !lqm && myok) a.qZ[(int64_t)kprev * a.ldq + (rg0 + hi) * RG + lo4] = lqprev[0];
    } else {
      if (kprev >= 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (rgok[r] && hi == (kprev & 3)) a.qZ[(int64_t)kprev * a.ldq + (rg0 + r) * RG + lo4] = lqprev[r];
      }
    }
  };
  for (int ii = 0; ii < nact; ++ii) {
    const int k = SPARSE ? klist[ii] : ii;
    const int buf = ii & 1;
    flush_lq();
    if (ii + 1 < nact) dma_record(SPARSE ? klist[ii + 1] : ii + 1, buf ^ 1);
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
    double d2[R], acc[2][R];
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
    double cv[R];  // c_jk of this cluster, fetched before the tile stream
    if constexpr (SADDR) {
      const double* ck = a.ctab + k;  // (scalar)
#pragma unroll
      for (int r = 0; r < R; ++r) asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(cv[r]) : "v"(coff[r]), "s"(ck) : "memory");
    } else if constexpr (ROWLANES) {
      cv[0] = a.ctab[(int64_t)owngrp * K + k];  // (the lane's own row group's constant: one load, not R)
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) cv[r] = a.ctab[(int64_t)grp[r] * K + k];
    }
    auto square = [&](int set) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        d2[r] = fma(acc[set][r], acc[set][r], d2[r]);
        // pin the running sum here: otherwise LLVM sinks the whole fma chain below the
        // MFMA stream and keeps every row's accumulator alive (96 extra VGPRs, spills)
        asm volatile("" : "+v"(d2[r]));
      }
    };
    if (wave_active) {
    static_for<NREAD>([&](auto nc) {
      constexpr int n = nc;
      constexpr RdInfo ri = rd_info(n);
      constexpr int set = ri.it & 1;
      const double v = ring[n % PF];
      if constexpr (n + PF < NREAD) {
        constexpr RdInfo rn = rd_info(n + PF);
        ring[n % PF] = rn.jt < 0 ? Pb[rn.off] : Pt[rn.off];
      }
      if constexpr (ri.jt < 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[set][r] = v;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[set][r] = mfma4(v, xf[r][ri.jt], acc[set][r]);
        if constexpr (ri.it > 0 && ri.jt == 1) square(set ^ 1);  // the row above, six MFMAs after its last one
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    square((NT - 1) & 1);
    }
    double lqsel = 0.0;
    // The asm loads of c_jk are invisible to hipcc's waitcnt pass (the LDS-direct record with them): waited for here, in
    // one asm statement that also names cv[] as in-out operands, so that no copy of a not-yet-loaded register can be
    // scheduled above it.  tools/check_isa.py asserts that the kernel has no scratch and that nothing reads these
    // registers between the loads and this wait.
    if constexpr (SADDR) {
      if constexpr (R == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2])::"memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(cv[0]), "+v"(cv[R > 1 ? 1 : 0])::"memory");
    }
    if constexpr (ROWLANES) {
      // Lane (lo4, hi) wants row group hi's distance: four CHAINED selector MFMAs leave sum_k d2_r[k][row] in the lanes with
      // hi == r and nothing else (round 6; four independent sums + four selects + four multiply-adds before).  The sum of a
      // row group is the one mfma4(1.0, d2[r], 0.0) forms -- the chain adds exact zeros to it: the same bits.
      double t = mfma4(selA[0], d2[0], 0.0);
#pragma unroll
      for (int r = 1; r < R; ++r) t = mfma4(selA[r], d2[r], t);
      lqsel = cv[0] - 0.5 * t;
      mx[0] = fmax(mx[0], lqsel);  // (one maximum per cluster; fmax: lqsel is one instruction behind an MFMA result -- see max_raw)
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        // sum over the four hi lanes on the matrix pipe: D[i][j] = sum_k 1 * B[k][j] leaves the
        // total in every lane (B[k=hi][j=row] is exactly where the partial sums live)
        const double dd = mfma4(1.0, d2[r], 0.0);
        const double lq = cv[r] - 0.5 * dd;
        mx[r] = max_raw(mx[r], lq);
        lqprev[r] = lq;
      }
    }
    kprev = k;
    // R == 4: lane (lo4, hi) keeps row group hi -- ONE 512-byte store per cluster column instead of four
    // 128-byte ones, and the normalisation below needs no cross-lane sums
    if constexpr (ROWLANES) {
      if (lqm) lql[k * NTHR + tid] = lqsel;
      else lqprev[0] = lqsel;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record has landed (this wave's part; the barrier covers the rest)
    __syncthreads();
  }

  flush_lq();
  if constexpr (SADDR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (asm stores: re-read below by the same lanes)

  if (a.raw) return;  // GaussWish::Eloglike: leave c_k - 0.5 d^2 in qZ, no normalisation

  // ---- normalise (probutils.cpp:141-150, cluster.cpp:124-131) -------------
  double fz = 0.0;
  if constexpr (ROWLANES) {
    // every lane owns one row (row group hi, row lo4) and walks all K columns it wrote itself
    const double mymx = mx[0];
    int mygrp = grp[0];
    bool myrow = rowok[0];
#pragma unroll
    for (int r = 1; r < R; ++r)
      if (hi == r) mygrp = grp[r], myrow = rowok[r];
    double* qp = a.qZ + (rg0 + hi) * RG + lo4;
    double s = 0.0;
    if (lqm && !a.ll_part) {  // (launch-uniform)
      // log q~ waits in LDS and the split-ordering term is not asked for: ONE exponential per entry, as fused_small_kernel
      // and the separable families' sweeps do -- e = exp(log q~ - max) goes back into the lane's slot and q = e / sum(e):
      // the same sum and logZ, q within two ulp of exp(log q~ - logZ) (cluster.cpp:130-131).  At D = 23 the second
      // exponential and its loop were 25 of the 90 VALU instructions a cluster costs next to 88 MFMAs.
      if (myok) {
#pragma unroll 8
        for (int k = 0; k < K; ++k) {
          const double e = exp_nonpos(lql[k * NTHR + tid] - mymx, etab);
          s += e;
          lql[k * NTHR + tid] = e;
        }
        const double inv = rcp_pos(s);
#pragma unroll 8
        for (int k = 0; k < K; ++k) {
          double q = lql[k * NTHR + tid] * inv;
          if (!myrow) q = 0.0;
          qp[(int64_t)k * a.ldq] = q;
        }
      }
      fz = (myok && myrow) ? log(s) + mymx : 0.0;
    } else {
    if (myok) {
#pragma unroll 8
      for (int k = 0; k < K; ++k) s += exp_nonpos((lqm ? lql[k * NTHR + tid] : qp[(int64_t)k * a.ldq]) - mymx, etab);
    }
    const double logZ = log(s) + mymx;
    for (int k = 0; k < K; ++k) {
      double ll = 0.0;
      if (myok) {
        const double lq = lqm ? lql[k * NTHR + tid] : qp[(int64_t)k * a.ldq];
        double q = exp_nonpos(lq - logZ, etab);
        if (!myrow) q = 0.0;
        qp[(int64_t)k * a.ldq] = q;
        if (a.ll_part && q > 0.0) ll = q * (lq - a.ctab[(int64_t)mygrp * K + k]);
      }
      if (a.ll_part) {  // wave-uniform
        ll = wave_sum(ll);
        if (lane == 0) llw[wave * K + k] = ll;
      }
    }
    fz = (myok && myrow) ? logZ : 0.0;
    }
  } else {
  double logZ[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double s = 0.0;
    if (rgok[r]) {
      const double* qp = a.qZ + (rg0 + r) * RG + lo4;
#pragma unroll 8
      for (int k = hi; k < K; k += 4) s += exp_nonpos(qp[(int64_t)k * a.ldq] - mx[r], etab);
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
          double q = exp_nonpos(lq - logZ[r], etab);
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
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (rgok[r] && rowok[r] && hi == 0) fz += logZ[r];
  }
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

// ===========================================================================
// E-step for observations wider than 128 columns
// ===========================================================================
// Same arithmetic, same operand layout; what changes is what stays resident.  A cluster's whitener no longer fits in
// LDS (268 KB at D = 256) and a row group's X fragments no longer fit in registers, so the lower-triangular A_k is cut
// into 64 x 64 blocks (I, J <= I) that stream through a two-deep LDS ring in row-major order -- one chunk = the 256
// 4x4 tiles of a block (a diagonal block reads only the WIDE_NLOW tiles that reach its diagonal) followed by -b_I -- and a wave keeps
//   * the 16 tile-row accumulators of block row I for its R row groups (y_I = sum_J A_IJ x_J - b_I), and
//   * the X fragments of column panel J only, re-read from L2 for every chunk (the loads return in order, so the
//     first MFMAs start as soon as the first fragments are back).
// After the diagonal chunk (J = I) the 16 accumulators are squared into the running distance.  Normalisation as in
// estep_kernel (k-sliced scheme).  lc_ctx.cpp packs the chunks (wide_chunk_stride doubles each, K * NCH of them).
// Column groups: tile column jt = 4 q + jr of a 64-column panel stands for the columns {16 q + jr + 4 h : h = 0..3},
// so that lane hi holds columns 16 q + 4 hi + (0..3) for jr = 0..3 -- FOUR CONTIGUOUS doubles per (row, q), fetched with
// two 16-byte loads, a row's 128-byte line consumed by its four hi lanes at once.  (With contiguous groups 4 jt + hi a
// lane's sixteen values lie 32 bytes apart: 32 eight-byte loads per chunk, every line of the wave's 16 KB working set
// fetched from L2 four times over -- the kernel sat at 0.52 of the peak waiting on those loads.)  lc_ctx.cpp packs the
// whitener's tiles with the same column map (wide_col).  A diagonal block keeps 160 of its 256 tiles instead of 136.
__host__ __device__ constexpr int wide_col(int jt, int h) { return 16 * (jt / 4) + 4 * h + (jt % 4); }
__host__ __device__ constexpr bool wide_low(int it, int jt) { return wide_col(jt, 0) <= 4 * it + 3; }  // any entry on / below the diagonal
// n-th tile read of a chunk -> 16 * it + jt.  Tile rows go in pairs (it = 2 (m / 32) + (m & 1), jt = (m % 32) / 2 over
// m = 0..255) so that 2R independent accumulator chains alternate; the tiles a diagonal block needs come first, the
// others after them: a diagonal block stops after the first part.
__host__ __device__ constexpr int wide_read(int n) {
  int c = 0;
  for (int part = 0; part < 2; ++part)
    for (int m = 0; m < 256; ++m) {
      const int it = 2 * (m / 32) + (m & 1), jt = (m % 32) / 2;
      if (wide_low(it, jt) == (part == 0)) {
        if (c == n) return it * 16 + jt;
        ++c;
      }
    }
  return -1;
}
__host__ __device__ constexpr int wide_nlow() {
  int c = 0;
  for (int it = 0; it < 16; ++it)
    for (int jt = 0; jt < 16; ++jt) c += wide_low(it, jt) ? 1 : 0;
  return c;
}
constexpr int WIDE_NLOW = wide_nlow();
// NPX > 0 (round 5): the instance for exactly NPX column panels (D <= 256) keeps ALL of its rows' X fragments in registers
// -- 2 * 16 * NPX doubles per lane, which takes one block per CU and the wave's full register file -- and selects the
// panel of a chunk by a switch over NPX copies of the chunk body: nothing but the whitener stream is read inside the
// cluster loop.  (The streaming instance re-read 64 KB of X per block and chunk: 83 GB per launch from beyond L2 at
// N = 1M, D = 256, K = 16, under which the part held 2.03 GHz.)
template <int R, int WAVES, int NPX>
__global__ void __launch_bounds__(WAVES * 64, NPX ? 1 : 2) estep_wide_kernel(EstepLaunch a) {
  constexpr int CHS = WIDE_CHUNK;  // 256 tiles x 16 + 64
  constexpr int NTHR = WAVES * 64;
  constexpr int NV2 = CHS / 2;
  constexpr int NPRE = (NV2 + NTHR - 1) / NTHR;
  constexpr int PF = NPX ? 8 : 6;  // tile reads in flight (one wave per SIMD in the resident instances: nobody else covers them)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* pbuf = lds;               // [2][CHS]
  double* llw = lds + 2 * CHS;      // [WAVES][K]
  double* fzw = llw + WAVES * a.K;  // [WAVES]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo4 = lane & 15, hi = lane >> 4;
  const int K = a.K, DPW = a.DP, NPAN = DPW / 64, NCH = NPAN * (NPAN + 1) / 2;
  const int64_t rg0 = ((int64_t)blockIdx.x * WAVES + wave) * R;

  const double* xbase[R];
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
    xbase[r] = a.X + ((rgok[r] ? rg : 0) * RG + lo4) * (int64_t)DPW + 4 * hi;
  }

  double pre[NPRE][2];
  auto gload = [&](int64_t g) {
    const double2* src = reinterpret_cast<const double2*>(a.params + g * CHS);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * NTHR;
      const double2 v = src[idx < NV2 ? idx : NV2 - 1];
      pre[i][0] = v.x;
      pre[i][1] = v.y;
    }
  };
  auto lstore = [&](int b) {
    double2* dst = reinterpret_cast<double2*>(pbuf + b * CHS);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < NV2) dst[idx] = make_double2(pre[i][0], pre[i][1]);
    }
  };

  // resident instances: the staging of the whitener stream rides inside the tile loop.  Chunk g + 1 goes from global memory
  // STRAIGHT into the other ring slot (global_load_lds_dwordx4 as in estep_kernel: 16 bytes per lane land at M0 + 16 * lane,
  // no staging registers, no address arithmetic on the VALU, no ds_write), one instruction every twelve tiles of chunk g --
  // everybody left that slot before the barrier that ended chunk g - 1, and a vmcnt(0) in front of this chunk's barrier
  // finds the loads long retired.  (In bulk at the chunk's end, with one wave per SIMD and nothing to hide them behind,
  // load addresses, stores and their wait were 10 % of the launch: profiles/r05_wide_resident_probe.log.)
  // Same-box A/Bs (profiles/r05_wide_resident_probe.log): against staging through registers inside the tile loop -4.5 % at
  // D = 256; for the STREAMING instance, whose fragment loads share the vector-memory counter with it, +2 ... 7 % -- that
  // one keeps its bulk register staging.
  constexpr bool DMA = NPX > 0;
  const unsigned dvoff = (unsigned)(wave * 1024 + lane * 16);
  const unsigned dlds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pbuf + (unsigned)(wave * 1024));
  auto dma_part = [&](int64_t g, int bb, auto ic) {
    constexpr int i = decltype(ic)::value;
    const char* src = reinterpret_cast<const char*>(a.params + g * CHS) + i * (NTHR * 16);
    const unsigned dv = dvoff, dl = dlds0 + bb * (CHS * 8) + i * (NTHR * 16);  // (copies: hipcc does not capture asm operands of a generic lambda)
    if ((i + 1) * (NTHR * 16) <= CHS * 8 || dv + i * (NTHR * 16) < (unsigned)(CHS * 8))  // (only the last round is partial)
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dv), "s"(src), "s"(dl) : "memory");
  };
  const int64_t total = (int64_t)K * NCH;
  if constexpr (DMA) {
    static_for<NPRE>([&](auto ic) { dma_part(0, 0, ic); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    gload(0);
    lstore(0);
  }
  __syncthreads();

  double xres[NPX ? NPX : 1][R][16];  // xres[p][r][4 q + jr] = x[row][64 p + 16 q + 4 hi + jr]
  if constexpr (NPX > 0) {
#pragma unroll
    for (int p = 0; p < NPX; ++p)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double2* p2 = reinterpret_cast<const double2*>(xbase[r] + 64 * p + 16 * q);
          const double2 v0 = p2[0], v1 = p2[1];
          xres[p][r][4 * q] = v0.x, xres[p][r][4 * q + 1] = v0.y, xres[p][r][4 * q + 2] = v1.x, xres[p][r][4 * q + 3] = v1.y;
        }
    // opaque from here on: hipcc otherwise sinks the loads back into the chunk bodies, which is the streaming instance
    // (volatile loads would do too -- one at a time, each waited for: 128 round trips at the head of every block)
#pragma unroll
    for (int p = 0; p < NPX; ++p)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          // (the first two panels in VGPRs, the others in AGPRs, where an MFMA can read its B operand as well: a hint
          //  that spares the chunk bodies hipcc's copies of a whole panel from one file to the other)
          if (p < 2) asm volatile("" : "+v"(xres[p][r][j]));
          else asm volatile("" : "+a"(xres[p][r][j]));
        }
  }
  double mx[R], d2[R], acc[16][R];
#pragma unroll
  for (int r = 0; r < R; ++r) mx[r] = -INFINITY, d2[r] = 0.0;
  int I = 0, J = 0, k = 0;
  for (int64_t g = 0; g < total; ++g) {
    const int buf = (int)(g & 1);
    if constexpr (!DMA) {
      if (g + 1 < total) gload(g + 1);
    }
    const int64_t g1 = g + 1 < total ? g + 1 : total - 1;  // (resident instances: the chunk fetched during this one)
    auto chunk = [&](const double (&xf)[R][16]) __attribute__((always_inline)) {
    const double* P = pbuf + buf * CHS;
    const double* Pt = P + (lane & 3) + 4 * hi;  // this lane's element of every 4x4 tile
    if (J == 0) {
      const double* Pb = P + 4096 + hi;  // -b_I: the accumulators of a block row start there (y = A x - b)
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const double v = Pb[4 * it];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[it][r] = v;
      }
    }
    // tile reads PF ahead of their use; the reads that run past the lower part of a diagonal block are not used
    double ring[PF];
    static_for<PF>([&](auto ic) {
      constexpr int t0 = wide_read(ic);
      ring[ic] = Pt[16 * t0];
    });
    auto reads = [&](auto first_c, auto count_c) {
      static_for<decltype(count_c)::value>([&](auto nc) {
        constexpr int n = decltype(first_c)::value + nc, t = wide_read(n), it = t / 16, jt = t % 16;
        const double v = ring[n % PF];
        if constexpr (n + PF < 256) {
          constexpr int tn = wide_read(n + PF);
          ring[n % PF] = Pt[16 * tn];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[it][r] = mfma4(v, xf[r][jt], acc[it][r]);
        if constexpr (DMA) {
          static_assert(NPRE <= 9 && WIDE_NLOW >= 12 * NPRE, "staging slots of the tile loop");
          if constexpr (n % 12 == 6 && n / 12 < NPRE) dma_part(g1, buf ^ 1, std::integral_constant<int, n / 12>{});
          // (one wave per SIMD: nobody else covers a tile read that hipcc moves next to its use)
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    };
    reads(std::integral_constant<int, 0>{}, std::integral_constant<int, WIDE_NLOW>{});
    if (J != I) reads(std::integral_constant<int, WIDE_NLOW>{}, std::integral_constant<int, 256 - WIDE_NLOW>{});
    };
    if constexpr (NPX == 0) {
      double xf[R][16];  // xf[r][4 q + jr] = x[row][64 J + 16 q + 4 hi + jr]
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double2* p2 = reinterpret_cast<const double2*>(xbase[r] + 64 * J + 16 * q);
          const double2 v0 = p2[0], v1 = p2[1];
          xf[r][4 * q] = v0.x, xf[r][4 * q + 1] = v0.y, xf[r][4 * q + 2] = v1.x, xf[r][4 * q + 3] = v1.y;
        }
      chunk(xf);
    } else {
      static_for<NPX>([&](auto pc) {
        if (J == decltype(pc)::value) chunk(xres[decltype(pc)::value]);
      });
    }
    if (J == I) {  // block row complete
#pragma unroll
      for (int it = 0; it < 16; ++it)
#pragma unroll
        for (int r = 0; r < R; ++r) d2[r] = fma(acc[it][r], acc[it][r], d2[r]);
      if (I == NPAN - 1) {  // cluster complete
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double dd = mfma4(1.0, d2[r], 0.0);  // sum over the four hi lanes, total in every lane
          const double lq = a.ctab[(int64_t)grp[r] * K + k] - 0.5 * dd;
          mx[r] = fmax(mx[r], lq);
          if (rgok[r] && hi == (k & 3)) a.qZ[(int64_t)k * a.ldq + (rg0 + r) * RG + lo4] = lq;
          d2[r] = 0.0;
        }
      }
    }
    if constexpr (!DMA) {
      if (g + 1 < total) lstore(buf ^ 1);
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (J == I) {
      J = 0;
      if (++I == NPAN) I = 0, ++k;
    } else {
      ++J;
    }
  }
  if (a.raw) return;

  // ---- normalise (probutils.cpp:141-150, cluster.cpp:124-131), as in estep_kernel ----
  double logZ[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double s = 0.0;
    if (rgok[r]) {
      const double* qp = a.qZ + (rg0 + r) * RG + lo4;
#pragma unroll 8
      for (int kk = hi; kk < K; kk += 4) s += exp(qp[(int64_t)kk * a.ldq] - mx[r]);
    }
    s = sum_over_hi(s);
    logZ[r] = log(s) + mx[r];
  }
  for (int kb = 0; kb < K; kb += 4) {
    const int kk = kb + hi;
    double ll = 0.0;
    if (kk < K) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (rgok[r]) {
          double* qp = a.qZ + (int64_t)kk * a.ldq + (rg0 + r) * RG + lo4;
          const double lq = *qp;
          double q = exp(lq - logZ[r]);
          if (!rowok[r]) q = 0.0;
          *qp = q;
          if (a.ll_part && q > 0.0) ll += q * (lq - a.ctab[(int64_t)grp[r] * K + kk]);
        }
      }
    }
    if (a.ll_part) {  // wave-uniform
      ll = sum_over_lo4(ll);
      if (kk < K && lo4 == 0) llw[wave * K + kk] = ll;
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
    for (int kk = tid; kk < K; kk += NTHR) {
      double s = 0.0;
      for (int w = 0; w < WAVES; ++w) s += llw[w * K + kk];
      a.ll_part[(int64_t)blockIdx.x * K + kk] = s;
    }
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < WAVES; ++w) s += fzw[w];
    a.fz_part[blockIdx.x] = -s;  // cluster.cpp:137 returns -sum(logZ)
  }
}
constexpr int WIDE_R = 2, WIDE_WAVES = 4;

static hipError_t launch_estep_wide(const EstepLaunch& a, hipStream_t stream) {
  if (a.DP % 64) return hipErrorInvalidValue;
  const size_t shmem = (size_t)(2 * WIDE_CHUNK + WIDE_WAVES * a.K + WIDE_WAVES) * sizeof(double);
  const int64_t grid = estep_grid(a);
  if (grid <= 0) return hipSuccess;
  static LdsGrant grants[3];
  auto go = [&](auto kern, LdsGrant& grant) {
    if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grant); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WIDE_WAVES * 64), shmem, stream, a);
    return hipGetLastError();
  };
  const char* sw = lck::test_switch("LC_WIDE_STREAM");  // tests: the streaming instance at every width
#ifdef LC_WIDE_FORCE_STREAM  // (tools/variants.py: the A/B of the two instances on one box)
  const bool stream_only = true;
  (void)sw;
#else
  const bool stream_only = sw && *sw && *sw != '0';
#endif
  if (a.DP == 192 && !stream_only) return go(estep_wide_kernel<WIDE_R, WIDE_WAVES, 3>, grants[1]);
  if (a.DP == 256 && !stream_only) return go(estep_wide_kernel<WIDE_R, WIDE_WAVES, 4>, grants[2]);
  return go(estep_wide_kernel<WIDE_R, WIDE_WAVES, 0>, grants[0]);
}

template <int DP>
struct EstepCfg;
template <>
struct EstepCfg<16> { static constexpr int R = 4, WAVES = 4; };
template <>
struct EstepCfg<32> { static constexpr int R = 4, WAVES = 4; };  // (8-wave blocks: 15-30 % slower at every narrow shape, 2-wave blocks 0-20 %: round 6)
template <>
struct EstepCfg<48> { static constexpr int R = 4, WAVES = 4; };
template <>
struct EstepCfg<64> { static constexpr int R = 3, WAVES = 4; };
template <>
struct EstepCfg<80> { static constexpr int R = 3, WAVES = 4; };
template <>
struct EstepCfg<96> { static constexpr int R = 3, WAVES = 4; };  // (5.04 vs 5.43 ms for R = 2 / WAVES = 8 at N = 1M, K = 32)
template <>
struct EstepCfg<112> { static constexpr int R = 2, WAVES = 8; };  // (three row groups spill here)
template <>
struct EstepCfg<128> { static constexpr int R = 2, WAVES = 8; };

template <int DP>
static int rows_per_block_t() { return EstepCfg<DP>::R * EstepCfg<DP>::WAVES * RG; }

int estep_rows_per_block(int DP) {
  switch (DP) {
    case 16: return rows_per_block_t<16>();
    case 32: return rows_per_block_t<32>();
    case 48: return rows_per_block_t<48>();
    case 64: return rows_per_block_t<64>();
    case 80: return rows_per_block_t<80>();
    case 96: return rows_per_block_t<96>();
    case 112: return rows_per_block_t<112>();
    case 128: return rows_per_block_t<128>();
  }
  if (DP > 128 && DP % 64 == 0) return WIDE_R * WIDE_WAVES * RG;
  return -1;
}

// log q~ waits in LDS up to this block size (LC_ES_LQLDS_KB, test-hooks library: A/B).  Round 6: 40 -> 80 KB, two blocks per CU.
// With ONE exponential per entry on that path the table pays although the occupancy halves: D = 32, K = 16 2.16 -> 2.00 ms
// (N = 6M), K = 32 4.07 -> 3.86; D = 48, K = 12 2.68 -> 2.51 (N = 5M), K = 28 5.75 -> 5.45; D = 23, K = 24 2.26 -> 2.13 (gpurun_out/r06m)
constexpr size_t ES_LQ_LDS_CAP = 80 * 1024;
static size_t estep_lq_cap() {
  static const size_t cap = test_switch("LC_ES_LQLDS_KB") ? (size_t)atoi(test_switch("LC_ES_LQLDS_KB")) * 1024 : ES_LQ_LDS_CAP;
  return cap;
}
static size_t estep_lds_base(int DC, int K, int R, int WAVES) {
  return (size_t)(2 * pstride(DC) + WAVES * K + WAVES + 64) * sizeof(double) + (size_t)(2 * K + WAVES * R + 2) * sizeof(int);
}
// D = 64 and 80 (three row groups per wave by default): the FOUR-row-group scheme with its log q~ table in LDS, one
// exponential per entry and the selector-chain epilogue wins where that table fits next to the two parameter records (D = 64:
// K <= 21, D = 80: K <= 12) from six clusters on -- D = 64, N = 4M: K = 8 2.30 -> 2.25 ms, K = 12 3.35 -> 3.25, K = 16 4.31 -> 4.19
// (0.82 -> 0.85 of the peak), K = 20 5.31 -> 5.19; K = 4 loses 2 % and keeps three (gpurun_out/r06r).  Dense, normalising
// launches only.  ONE decision for the launch and for the grid the caller sizes its partial sums by (estep_grid).
static bool estep_four_groups(const EstepLaunch& a) {
  static const bool off = test_switch("LC_ES_R4") && atoi(test_switch("LC_ES_R4")) == 0;  // (tests: the default scheme everywhere)
  if (off || (a.DP != 64 && a.DP != 80) || a.raw || a.sparse || a.K < 6) return false;
  const int DC = a.DC > 0 ? a.DC : a.DP;
  return estep_lds_base(DC, a.K, 4, 4) + (size_t)a.K * 4 * 64 * sizeof(double) <= estep_lq_cap();
}
int64_t estep_grid(const EstepLaunch& a) {
  const int64_t rgpb = (a.DP <= 128 && estep_four_groups(a) ? 4 * 4 * RG : estep_rows_per_block(a.DP)) / RG;
  return (a.nrg + rgpb - 1) / rgpb;
}

template <int DP, int DC, bool SPARSE, int R = EstepCfg<DP>::R, int WAVES = EstepCfg<DP>::WAVES>
static hipError_t launch_estep_s(const EstepLaunch& a, hipStream_t stream) {
  size_t shmem = estep_lds_base(DC, a.K, R, WAVES);
  EstepLaunch b = a;
  if (R == 4 && !a.raw && shmem + (size_t)a.K * WAVES * 64 * sizeof(double) <= estep_lq_cap()) {
    b.lq_lds = 1;  // log q~ stays in LDS until the normalisation
    shmem += (size_t)a.K * WAVES * 64 * sizeof(double);
  }
  auto kern = estep_kernel<DP, DC, R, WAVES, SPARSE>;
  static LdsGrant grant;  // largest dynamic-LDS size already granted, per device
  if (hipError_t e = grant_dynamic_lds(reinterpret_cast<const void*>(kern), shmem, grant); e != hipSuccess) return e;
  const int64_t grid = estep_grid(a);
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), shmem, stream, b);
  return hipGetLastError();
}

template <int DP>
static hipError_t launch_estep_t(const EstepLaunch& a, hipStream_t stream) {
  if constexpr (DP == 64 || DP == 80) {
    if (estep_four_groups(a)) {
      if (a.DC == DP - 8) return launch_estep_s<DP, DP - 8, false, 4, 4>(a, stream);
      if (a.DC == 0 || a.DC == DP) return launch_estep_s<DP, DP, false, 4, 4>(a, stream);
    }
  }
  if constexpr (DP >= 32) {
    if (a.DC == DP - 8) return a.sparse ? launch_estep_s<DP, DP - 8, true>(a, stream) : launch_estep_s<DP, DP - 8, false>(a, stream);
  }
  if constexpr (DP == 32 || DP == 48) {  // (active widths in steps of four columns at the two narrowest layouts)
    if (a.DC == DP - 4) return a.sparse ? launch_estep_s<DP, DP - 4, true>(a, stream) : launch_estep_s<DP, DP - 4, false>(a, stream);
    if (a.DC == DP - 12) return a.sparse ? launch_estep_s<DP, DP - 12, true>(a, stream) : launch_estep_s<DP, DP - 12, false>(a, stream);
  }
  if (a.DC != 0 && a.DC != DP) return hipErrorInvalidValue;
  return a.sparse ? launch_estep_s<DP, DP, true>(a, stream) : launch_estep_s<DP, DP, false>(a, stream);
}

hipError_t launch_estep(const EstepLaunch& a, hipStream_t stream) {
  if (a.DP > 128) return launch_estep_wide(a, stream);  // -inf entries of ctab need no special handling there
  switch (a.DP) {
    case 16: return launch_estep_t<16>(a, stream);
    case 32: return launch_estep_t<32>(a, stream);
    case 48: return launch_estep_t<48>(a, stream);
    case 64: return launch_estep_t<64>(a, stream);
    case 80: return launch_estep_t<80>(a, stream);
    case 96: return launch_estep_t<96>(a, stream);
    case 112: return launch_estep_t<112>(a, stream);
    case 128: return launch_estep_t<128>(a, stream);
  }
  return hipErrorInvalidValue;
}


}  // namespace lck
