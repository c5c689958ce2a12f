// Probe: v_mfma_f64_16x16x4_f64 fragment layout + sustained rate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_probe tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* C) {
  // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1e-9;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


template <int NACC>
__global__ void __launch_bounds__(256) rate4_kernel(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA + interleaved independent VALU fma (same wave)
template <int NACC, int NV>
__global__ void __launch_bounds__(256) mixed_kernel(double* out, int iters, double seed, long long* clk) {
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  v4d acc[NACC];
  double va[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NV; ++i) va[i] = i;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) va[j] = __builtin_fma(a, va[j], 1e-9);
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < NV; ++i) s += va[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  // ---- layout
  std::vector<double> A(64), B(64), C(256), R(256, 0.0);
  for (int i = 0; i < 64; ++i) { A[i] = sin(1.0 + i * 0.37); B[i] = cos(2.0 + i * 0.91) + 0.01 * i; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
  double *dA, *dB, *dC;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dC);
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(C[i] - R[i]));
  printf("layout max_abs_err=%.3e (%s)\n", err, err < 1e-12 ? "OK" : "MISMATCH");
  // ---- rate
  double* out; CK(hipMalloc(&out, (size_t)256 * 8192 * 8));
  const int iters = 4096;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    int blocks = p.multiProcessorCount * bpc;
    {
      float ms = time_ms([&] { rate_kernel<1><<<blocks, 256>>>(out, iters, 0.5); }, 5);
      double fl = (double)blocks * 4 * iters * 1 * 2048.0;
      printf("mfma_f64 NACC=1 blocks/CU=%d: %.3f ms  %.2f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", bpc, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 1.0 * bpc));
    }
    {
      float ms = time_ms([&] { rate_kernel<2><<<blocks, 256>>>(out, iters, 0.5); }, 5);
      double fl = (double)blocks * 4 * iters * 2 * 2048.0;
      printf("mfma_f64 NACC=2 blocks/CU=%d: %.3f ms  %.2f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", bpc, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 2.0 * bpc));
    }
    {
      float ms = time_ms([&] { rate_kernel<4><<<blocks, 256>>>(out, iters, 0.5); }, 5);
      double fl = (double)blocks * 4 * iters * 4 * 2048.0;
      printf("mfma_f64 NACC=4 blocks/CU=%d: %.3f ms  %.2f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", bpc, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * bpc));
    }
    {
      float ms = time_ms([&] { rate_kernel<8><<<blocks, 256>>>(out, iters, 0.5); }, 5);
      double fl = (double)blocks * 4 * iters * 8 * 2048.0;
      printf("mfma_f64 NACC=8 blocks/CU=%d: %.3f ms  %.2f TFLOP/s\n", bpc, ms, fl / ms * 1e-9);
    }
  }
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    int blocks = p.multiProcessorCount * bpc;
    float ms = time_ms([&] { fma_kernel<8><<<blocks, 256>>>(out, iters, 0.5); }, 5);
    double fl = (double)blocks * 256 * iters * 8 * 2.0;
    printf("v_fma_f64 NACC=8 blocks/CU=%d: %.3f ms  %.2f TFLOP/s\n", bpc, ms, fl / ms * 1e-9);
  }

  {
    long long* clk; CK(hipMalloc(&clk, 16)); long long h[2];
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
      int blocks = p.multiProcessorCount * bpc;
      float ms = time_ms([&] { mixed_kernel<4, 0><<<blocks, 256>>>(out, iters, 0.5, clk); }, 5);
      CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      double fl = (double)blocks * 4 * iters * 4 * 2048.0;
      printf("mixed<4,0> blocks/CU=%d: %.3f ms  mfma %.2f TF  clk64=%lld wall=%lld (ratio %.3f => %.0f MHz if wall=100MHz) cyc/mfma/wave=%.1f\n", bpc, ms, fl / ms * 1e-9, h[0], h[1], (double)h[0] / h[1], 100.0 * h[0] / h[1], (double)h[0] / (iters * 4.0));
    }
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      int blocks = p.multiProcessorCount * bpc;
      float ms = time_ms([&] { mixed_kernel<4, 1><<<blocks, 256>>>(out, iters, 0.5, clk); }, 5);
      CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      double fl = (double)blocks * 4 * iters * 4 * 2048.0, fv = (double)blocks * 256 * iters * 4 * 1 * 2.0;
      printf("mixed<4,1> blocks/CU=%d: %.3f ms  mfma %.2f TF + valu %.2f TF  cyc/mfma/wave=%.1f\n", bpc, ms, fl / ms * 1e-9, fv / ms * 1e-9, (double)h[0] / (iters * 4.0));
      ms = time_ms([&] { mixed_kernel<4, 4><<<blocks, 256>>>(out, iters, 0.5, clk); }, 5);
      CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      fv = (double)blocks * 256 * iters * 4 * 4 * 2.0;
      printf("mixed<4,4> blocks/CU=%d: %.3f ms  mfma %.2f TF + valu %.2f TF  cyc/mfma/wave=%.1f\n", bpc, ms, fl / ms * 1e-9, fv / ms * 1e-9, (double)h[0] / (iters * 4.0));
      ms = time_ms([&] { mixed_kernel<4, 8><<<blocks, 256>>>(out, iters, 0.5, clk); }, 5);
      CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      fv = (double)blocks * 256 * iters * 4 * 8 * 2.0;
      printf("mixed<4,8> blocks/CU=%d: %.3f ms  mfma %.2f TF + valu %.2f TF  cyc/mfma/wave=%.1f\n", bpc, ms, fl / ms * 1e-9, fv / ms * 1e-9, (double)h[0] / (iters * 4.0));
    }
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      int blocks = p.multiProcessorCount * bpc;
      float ms = time_ms([&] { rate4_kernel<8><<<blocks, 256>>>(out, iters, 0.5); }, 5);
      double fl = (double)blocks * 4 * iters * 8 * (2.0 * 4 * 4 * 4 * 4);
      printf("mfma_f64_4x4x4 NACC=8 blocks/CU=%d: %.3f ms  %.2f TFLOP/s\n", bpc, ms, fl / ms * 1e-9);
    }
  }
  return 0;
}
