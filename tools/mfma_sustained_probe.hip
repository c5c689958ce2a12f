// Probe: what fp64 matrix rate does an MI355X SUSTAIN over tens of milliseconds?  (tools/mfma_f64_probe.hip measures
// launches of under a millisecond, before power management reacts, on constant operands.)
// A wave streams v_mfma_f64_4x4x4_4b with 12 independent accumulators; operands are either constants or random
// doubles (switching activity = power); optionally one ds_read_b64 per LDSPER MFMAs, as the E-step has.  Reports, per
// launch of >= 20 ms: TFLOP/s, the shader clock (clock64 against the 100 MHz wall clock) and the resulting cycles per
// MFMA and SIMD -- 16 is the pipe's limit.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_sustained_probe tools/mfma_sustained_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int LDSPER>
__global__ void __launch_bounds__(256) stream_kernel(const double* in, double* out, int iters, long long* clk) {
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = in[i];
  __syncthreads();
  double b[16], acc[12];
#pragma unroll
  for (int i = 0; i < 16; ++i) b[i] = in[2048 + i * 256 + threadIdx.x];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0;
  double a = in[threadIdx.x];
  const double* lp = lds + (threadIdx.x & 15);
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if constexpr (LDSPER > 0) {
        if (j % LDSPER == 0) a = lp[((it * 16 + j) * 16) & 2047];
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[(i + j) & 15], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] *= 1e-30;  // keep the values finite (12 VALU per 192 MFMAs)
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int LDSPER>
static void run(const char* name, const double* din, double* dout, long long* dclk, int wavesPerSimd, int iters, int reps) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * wavesPerSimd;  // 4 waves per block: one per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream_kernel<LDSPER>, dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[2]; CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
    const double mf = (double)blocks * 4 * iters * 192.0;
    const double ghz = (double)h[0] / h[1] * 0.1;
    printf("%-34s waves/SIMD=%d launch %d: %7.2f ms  %6.2f TFLOP/s  clock %.3f GHz  %.2f cycles per MFMA and SIMD\n", name, wavesPerSimd, r, ms,
           mf * 512 / ms / 1e9, ghz, (double)h[0] / ((double)iters * 192.0 * wavesPerSimd));
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000, reps = argc > 2 ? atoi(argv[2]) : 6;
  std::vector<double> h(2048 + 16 * 256);
  double *dc, *dr, *dout; long long* dclk;
  CK(hipMalloc(&dc, h.size() * 8)); CK(hipMalloc(&dr, h.size() * 8)); CK(hipMalloc(&dout, 8 << 20)); CK(hipMalloc(&dclk, 16));
  for (auto& v : h) v = 1.0;
  CK(hipMemcpy(dc, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  std::mt19937_64 g(1); std::normal_distribution<double> nd(0.0, 1.0);
  for (auto& v : h) v = nd(g);
  CK(hipMemcpy(dr, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  for (int w = 2; w <= 3; ++w) {
    run<0>("constant operands, no LDS", dc, dout, dclk, w, iters, reps);
    run<0>("random operands, no LDS", dr, dout, dclk, w, iters, reps);
    run<3>("random operands, 1 LDS read / 3", dr, dout, dclk, w, iters, reps);
  }
  return 0;
}
