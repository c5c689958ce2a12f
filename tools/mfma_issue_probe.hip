// Probe: what keeps v_mfma_f64_4x4x4_4b_f64 below peak in a suffstat-like loop?
// Variants (template V): 0 = 72 independent accumulators, fixed A/B operands
//                        1 = varying A/B operand registers (16 A regs x 8 B regs)
//                        2 = 1 + bunched VALU block per iteration (46 DPP movs + 18 fp64 ops)
//                        3 = 2 but A operands rotate through DPP results (true dependence VALU->MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int V, int WPS>
__global__ void __launch_bounds__(256, WPS) k(double* out, int iters, const double* in) {
  double acc[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) acc[i] = 0;
  double x[4], q[2];
  for (int i = 0; i < 4; ++i) x[i] = in[threadIdx.x + 256 * i];
  q[0] = in[threadIdx.x + 1024]; q[1] = in[threadIdx.x + 1280];
  double sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nacc[2] = {0, 0};
  for (int it = 0; it < iters; ++it) {
    double A[16], B[8];
    if (V >= 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        A[4 * j] = x[j];
        A[4 * j + 1] = dpp<0x124>(x[j]);
        A[4 * j + 2] = dpp<0x128>(x[j]);
        A[4 * j + 3] = dpp<0x12C>(x[j]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { B[4 * c + j] = q[c] * x[j]; sacc[4 * c + j] += B[4 * c + j]; }
        nacc[c] += q[c];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) A[j] = x[j & 3] + (V == 1 ? j : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) B[j] = q[j & 1] + (V == 1 ? j : 0);
    }
#pragma unroll
    for (int i = 0; i < 72; ++i) {
      const int ai = V == 0 ? 0 : (i % 16), bi = V == 0 ? 0 : ((i / 9) % 8);
      acc[i] = mfma4(A[ai], B[bi], acc[i]);
    }
    if (V >= 2) {  // make x/q change so nothing hoists
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = x[j] * 1.0000001;
    }
  }
  double s = nacc[0] + nacc[1];
#pragma unroll
  for (int i = 0; i < 72; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += sacc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V, int WPS> void run(const char* name, double* out, const double* in, int cus) {
  const int iters = 2000, blocks = cus * WPS;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V, WPS><<<blocks, 256>>>(out, iters, in);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) k<V, WPS><<<blocks, 256>>>(out, iters, in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  double fl = (double)blocks * 4 * iters * 72 * 512.0;
  printf("%-44s waves/SIMD=%d  %.3f ms  %.2f TFLOP/s  (%.1f%% of 78.6)  cyc/MFMA/SIMD=%.2f\n", name, WPS, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 78.6 * 100, ms * 1e-3 * 2.4e9 / (iters * 72.0 * WPS));
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  double *out, *in; CK(hipMalloc(&out, 256 * 1024 * 8)); CK(hipMalloc(&in, 2048 * 8));
  double h[2048]; for (int i = 0; i < 2048; ++i) h[i] = 0.5 + 1e-3 * (i % 97);
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  int cus = p.multiProcessorCount;
  run<0, 1>("V0 fixed operands", out, in, cus);
  run<0, 2>("V0 fixed operands", out, in, cus);
  run<1, 1>("V1 varying operand registers", out, in, cus);
  run<1, 2>("V1 varying operand registers", out, in, cus);
  run<2, 1>("V2 + bunched VALU block (DPP, mul, add)", out, in, cus);
  run<2, 2>("V2 + bunched VALU block (DPP, mul, add)", out, in, cus);
  return 0;
}
