// Device-side helpers shared by the kernel translation units.
//
// Matrix instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks, 512 flop, 16 cycles/SIMD = 32
// flop/clk/SIMD; measured 73-77 TFLOP/s on MI355X, vs 47-49 for v_mfma_f64_16x16x4_f64 --
// profiles/r01_mfma_f64_probe.log).  Lane layout (probed, tools/mfma_f64_4x4_layout.hip),
// lane = lo2 + 4*blk + 16*hi:
//   A[i][k] of block blk : i = lo2, k = hi
//   B[k][j] of block blk : j = lo2, k = hi
//   D[i][j] of block blk : j = lo2, i = hi
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "lc_kernels.h"

namespace lck {

// Compute units of the CURRENT device (a process may hold contexts on several, and learn_sharded drives its shards from
// concurrent threads: one relaxed atomic slot per device, filled by whoever asks first -- every writer stores the same
// value).  256 when the query fails.
inline int current_device_cus() {
  static std::atomic<int> cus_of[32];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = -1;
  int cus = dev >= 0 ? cus_of[dev].load(std::memory_order_relaxed) : 0;
  if (cus <= 0) {
    hipDeviceProp_t p;
    if (dev >= 0 && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
    if (dev >= 0) cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// max without the canonicalising self-max hipcc puts in front of fmax (three v_max_f64 per call where one does;
// the operands here are never signalling NaNs).  NOT for an operand that an MFMA has just written: the hazard recogniser
// does not look inside inline asm, so the wait states an MFMA result needs before a VALU read are not inserted in front
// of it -- use fmax there (seen as NaN rows out of the half-width fused instance, whose last chain link sits right in
// front of the max).
__device__ __forceinline__ double max_raw(double a, double b) {
  // ("+v": the result goes into the register that holds `a` -- with a separate "=v" output the allocator may hand the asm a
  //  freshly freed register that an in-flight MFMA still reads as SrcC or is about to write, and no wait states are
  //  inserted for asm in that direction either; tools/check_isa.py looks at both directions)
  asm("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));
  return a;
}

// exp(x) for x <= 0 (log q~ - max in the normalisation sweeps; -inf allowed), 1 ulp: 2^(n / 64) from a 64-entry table the
// kernel keeps in LDS (fill_exp_table) times a degree-5 polynomial on |r| <= ln 2 / 128 -- 16 VALU instructions and one
// LDS read against the 26 of the library exp, in kernels whose normalisation is bound by fp64 issue (one exponential per
// responsibility: K of them per row).  Reference: probutils::logsumexp / vbexpectation's exp() calls
// (src/probutils.cpp:146, src/cluster.cpp:130-131); the result differs from libm's by at most one unit in the last place.
static __constant__ double LC_EXP2_TAB[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
__device__ __forceinline__ void fill_exp_table(double* etab, int tid, int nthreads) {
  for (int i = tid; i < 64; i += nthreads) etab[i] = LC_EXP2_TAB[i];
}
__device__ __forceinline__ double exp_nonpos(double x, const double* etab) {
  // (exp(-750) = 0; keeps n finite for x = -inf.  fmax, not the asm max_raw: the argument is the result of a subtraction,
  //  which hipcc knows to be canonical -- one v_max_f64 -- and inline asm counts as convergent, which keeps the sweeps'
  //  run-time loops from being unrolled)
  x = fmax(x, -750.0);
  const double n = __builtin_rint(x * 0x1.71547652b82fep+6);      // 64 / ln 2
  double r = fma(n, -0x1.62e42fef00000p-7, x);                    // ln 2 / 64, 33 significant bits: n * C1 is exact
  r = fma(n, -0x1.473de6af278edp-40, r);
  double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  const double p = fma(q, r * r, r);                               // e^r - 1
  const int ni = (int)n;
  const double T = etab[ni & 63];
  return ldexp(fma(T, p, T), ni >> 6);
}
// 1 / s for s > 0, finite (a row's sum of exponentials, >= 1): hardware estimate + two Newton steps, < 1 ulp off the
// correctly rounded quotient -- 5 instructions against the 12 of an IEEE division
__device__ __forceinline__ double rcp_pos(double s) {
  double y = __builtin_amdgcn_rcp(s);
  double e = fma(-s, y, 1.0);
  y = fma(y, e, y);
  e = fma(-s, y, 1.0);
  return fma(y, e, y);
}

// Fingerprint of one row of responsibilities (softmax_cached_kernel writes and compares them; qhash_verify_kernel
// recomputes them in the tests): a multiplicative chain over the NON-ZERO entries,
//   h <- (h ^ (bits(q_j) + C (j + 1))) * G;  h ^= h >> 32      (mod 2^64, G odd)
// A zero entry leaves h as it is, so K columns and K + 1 columns with q_K = 0 agree; every other entry is mixed in
// together with its column, so a label that moves from one column to another changes the fingerprint.  (Round 4 summed
// bits(q_j) * G (2 j + 1): LINEAR in the bit patterns -- two rows collided whenever sum_j (2 j + 1) dbits_j = 0, e.g.
// column 0 up by 3 ulp and column 1 down by 1, which few-ulp moves between the sweeps of a split round do produce.
// In the chain a change of any entry reaches every later step through a product and an xor: no small integer relation
// between the entries' moves survives.)
constexpr uint64_t QHASH_SEED = 0x243F6A8885A308D3ull;
__device__ __forceinline__ uint64_t qhash_step(uint64_t acc, double q, int j) {
  const uint64_t b = (uint64_t)__double_as_longlong(q);  // (q >= +0: a zero is the all-zero pattern)
  uint64_t t = (acc ^ (b + 0xD6E8FEB86659FD93ull * (uint64_t)(j + 1))) * 0x9E3779B97F4A7C15ull;
  t ^= t >> 32;
  return b ? t : acc;
}
__device__ __forceinline__ int64_t qhash_finish(uint64_t acc) {
  const int64_t h = (int64_t)acc;
  return h == QHASH_NONE ? 1 : h;
}

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

}  // namespace lck
