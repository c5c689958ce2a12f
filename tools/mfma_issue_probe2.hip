// Probe 2: add suffstat's non-MFMA work to a 72-MFMA step one piece at a time.
//  V=0 MFMAs only (operands in registers, varying)
//  V=1 + operands re-read from LDS every step (16 x ds_read_b64 + q) 
//  V=2 + the 18 fp64 VALU ops of the step (q*x, s += qx, n += q)
//  V=3 + workgroup barrier every 8 steps
//  V=4 + global->register->LDS staging of the next batch (4 x 16B loads per thread per batch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
constexpr int LD = 80, BR = 32;
template <int V, int WPS>
__global__ void __launch_bounds__(256, WPS) k(double* out, int nbatch, const double* X, long ldx) {
  __shared__ double xs[2][BR * LD];
  __shared__ double qs[2][4][2][BR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
  double acc[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) acc[i] = 0;
  double sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nacc[2] = {0, 0};
  for (int i = tid; i < 2 * BR * LD; i += 256) (&xs[0][0])[i] = 0.001 * (i % 113);
  for (int i = tid; i < 2 * 4 * 2 * BR; i += 256) (&qs[0][0][0][0])[i] = 0.5 + 0.001 * (i % 7);
  __syncthreads();
  double xr[4][4], q[2];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int s = 0; s < 4; ++s) xr[jb][s] = xs[0][hi * LD + 16 * jb + 4 * ((blk + s) & 3) + lo2];
  q[0] = qs[0][wave][0][hi]; q[1] = qs[0][wave][1][hi];
  const double* xg = X + (size_t)blockIdx.x * nbatch * BR * 64;
  for (int b = 0; b < nbatch; ++b) {
    const int buf = b & 1;
    double2 pre[4];
    if (V >= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, row = idx / 32, c2 = idx % 32;
        pre[i] = *reinterpret_cast<const double2*>(xg + ((size_t)(b + 1) * BR + row) * 64 + 2 * c2);
      }
    }
    for (int st = 0; st < 8; ++st) {
      if (V >= 1) {
        const double* xb = &xs[buf][(st * 4 + hi) * LD + lo2];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
          for (int s = 0; s < 4; ++s) xr[jb][s] = xb[16 * jb + 4 * ((blk + s) & 3)];
        q[0] = qs[buf][wave][0][st * 4 + hi]; q[1] = qs[buf][wave][1][st * 4 + hi];
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        double qx[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          if (V >= 2) { qx[jb] = q[c] * xr[jb][0]; sacc[4 * c + jb] += qx[jb]; } else qx[jb] = xr[jb][(c + 1) & 3];
        }
        if (V >= 2) nacc[c] += q[c];
        int idx = 36 * c;
#pragma unroll
        for (int jbp = 0; jbp < 4; ++jbp)
#pragma unroll
          for (int jb = 0; jb <= jbp; ++jb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (s < 3 || jb < jbp) { acc[idx] = mfma4(xr[jbp][s], qx[jb], acc[idx]); ++idx; }
      }
    }
    if (V >= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256, row = idx / 32, c2 = idx % 32;
        *reinterpret_cast<double2*>(&xs[buf ^ 1][row * LD + 2 * c2]) = pre[i];
      }
    }
    if (V >= 3) __syncthreads();
  }
  double s = nacc[0] + nacc[1];
#pragma unroll
  for (int i = 0; i < 72; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += sacc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V, int WPS> void run(const char* name, double* out, const double* X, int cus) {
  const int nbatch = 200, blocks = cus * WPS;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V, WPS><<<blocks, 256>>>(out, nbatch, X, 64);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) k<V, WPS><<<blocks, 256>>>(out, nbatch, X, 64);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  double fl = (double)blocks * 4 * nbatch * 8 * 72 * 512.0;
  printf("%-52s waves/SIMD=%d  %.3f ms  %.2f TF  (%.1f%% of 78.6)  cyc/MFMA/SIMD=%.2f\n", name, WPS, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 78.6 * 100, ms * 1e-3 * 2.4e9 / (nbatch * 8 * 72.0 * WPS));
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double *out, *X; CK(hipMalloc(&out, 256 * 1024 * 8));
  size_t nx = (size_t)cus * 2 * 202 * BR * 64; CK(hipMalloc(&X, nx * 8)); CK(hipMemset(X, 0, nx * 8));
  run<0, 1>("V0 MFMA only", out, X, cus); run<0, 2>("V0 MFMA only", out, X, cus);
  run<1, 1>("V1 + LDS operand reads each step", out, X, cus); run<1, 2>("V1 + LDS operand reads each step", out, X, cus);
  run<2, 1>("V2 + 18 fp64 VALU ops", out, X, cus); run<2, 2>("V2 + 18 fp64 VALU ops", out, X, cus);
  run<3, 1>("V3 + barrier per 8 steps", out, X, cus); run<3, 2>("V3 + barrier per 8 steps", out, X, cus);
  run<4, 1>("V4 + global->LDS staging", out, X, cus); run<4, 2>("V4 + global->LDS staging", out, X, cus);
  return 0;
}
