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

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "lc_kernels.h"

namespace lck {

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// max without the canonicalising self-max hipcc puts in front of fmax (three v_max_f64 per call where one does;
// the operands here are never signalling NaNs)
__device__ __forceinline__ double max_raw(double a, double b) {
  asm("v_max_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
  return a;
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
