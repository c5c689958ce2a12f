// What does one more instruction cost next to a stream of v_mfma_f64_4x4x4_4b?  72 independent MFMAs per iteration
// (suffstat_kernel's step for two clusters) in 18 groups of four; after each of the first NV groups ONE extra
// operation of type T is issued.  Prints cycles per MFMA per SIMD and the price of the extra operation in cycles.
//   T: 1 v_mul_f64   2 v_add_f64   3 v_fma_f64   4 2 x v_mov_b32 (64-bit copy)   5 2 x v_mov_b32_dpp (bank-masked)
//      6 2 x v_cndmask_b32   7 ds_read_b64 (used at once)   8 v_add_u32   9 v_mul_f64 feeding the NEXT group's MFMAs
//      10 ds_read_b64 whose value is used four groups later (software-pipelined, as the kernels do)
//      11 ds_read_b128 used four groups later (two operands per instruction)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_price.hip -o tools/bin/mfma_valu_price
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int T, int NV, int WPS>
__global__ void __launch_bounds__(256, WPS) k(double* out, int iters, double seed) {
  __shared__ __attribute__((aligned(16))) double lds[1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 1024; i += 256) lds[i] = 0.001 * i;
  __syncthreads();
  double acc[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) acc[i] = 0;
  double a[4], b[4], y[4], s[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = seed + lane * 0.01 + i, b[i] = seed * 0.5 + i, y[i] = b[i];
  double q = seed * 0.25;
  double ringv[4] = {1.0, 2.0, 3.0, 4.0};
  double2 ring2[4] = {{1.0, 2.0}, {3.0, 4.0}, {5.0, 6.0}, {7.0, 8.0}};
  int iv = lane;
  const bool sel = (lane & 8) != 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 18; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * g + j] = mfma4(a[j], T == 9 ? y[(g + j) & 3] : b[(g + j) & 3], acc[4 * g + j]);
      __builtin_amdgcn_sched_barrier(0);
      if (g < NV) {
        const int r = g & 3;
        if (T == 1) y[r] = q * a[r];
        if (T == 2) s[r] += a[r];
        if (T == 3) s[r] = __builtin_fma(q, a[r], s[r]);
        if (T == 4) { asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(((int*)&y[r])[0]), "=v"(((int*)&y[r])[1]) : "v"(((int*)&a[r])[0]), "v"(((int*)&a[r])[1])); }
        if (T == 5) {
          int lo = ((int*)&y[r])[0], hi = ((int*)&y[r])[1];
          lo = __builtin_amdgcn_update_dpp(lo, ((int*)&a[r])[0], 0xE4, 0xf, 0xc, false);
          hi = __builtin_amdgcn_update_dpp(hi, ((int*)&a[r])[1], 0xE4, 0xf, 0xc, false);
          ((int*)&y[r])[0] = lo;
          ((int*)&y[r])[1] = hi;
        }
        if (T == 6) y[r] = sel ? a[r] : b[r];
        if (T == 7) y[r] = lds[(lane + 64 * r + it) & 1023];
        if (T == 8) iv += lane + g;
        if (T == 9) y[(g + 1) & 3] = q * a[(g + 1) & 3];  // consumed by the MFMAs of the next group
        if (T == 10) {  // the value read now replaces an MFMA operand four groups from now (ring of four)
          b[(g + 0) & 3] = ringv[g & 3];
          ringv[g & 3] = lds[(lane + 64 * r + it) & 1023];
        }
        if (T == 11) {
          b[(g + 0) & 3] = ring2[g & 3].x + ring2[g & 3].y;
          ring2[g & 3] = *reinterpret_cast<const double2*>(&lds[(2 * lane + 128 * r + 2 * it) & 1022]);
        }
        if (T < 10) asm volatile("" : "+v"(y[r]), "+v"(s[r]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    q += 1e-9;
  }
  double t = q + iv;
#pragma unroll
  for (int i = 0; i < 72; ++i) t += acc[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) t += y[i] + s[i] + ringv[i] + ring2[i].x;
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

static double base_cyc[3];
template <int T, int NV, int WPS>
void run(const char* name, double* out, int cus) {
  const int iters = 2000, blocks = cus * WPS;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k<T, NV, WPS><<<blocks, 256>>>(out, iters, 1.0);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    k<T, NV, WPS><<<blocks, 256>>>(out, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double cyc_iter = best * 1e-3 * 2.4e9 / iters / WPS;  // SIMD cycles per iteration and wave (at 2.4 GHz)
  if (NV == 0) base_cyc[WPS] = cyc_iter;
  const double fl = (double)blocks * 4 * iters * 72 * 512.0;
  printf("%-34s NV=%2d waves/SIMD=%d  %.3f ms  %5.1f%% of 78.6 TF  cyc/MFMA=%.2f", name, NV, WPS, best, fl / best * 1e-9 / 78.6 * 100,
         cyc_iter / 72.0);
  if (NV > 0) printf("  price=%.1f cyc per extra op", (cyc_iter - base_cyc[WPS]) / NV);
  printf("\n");
}
template <int T>
void sweep(const char* name, double* out, int cus) {
  run<T, 9, 1>(name, out, cus);
  run<T, 18, 1>(name, out, cus);
  run<T, 9, 2>(name, out, cus);
  run<T, 18, 2>(name, out, cus);
}
int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double* out;
  CK(hipMalloc(&out, (size_t)cus * 2 * 256 * 8));
  run<0, 0, 1>("MFMA only", out, cus);
  run<0, 0, 2>("MFMA only", out, cus);
  sweep<1>("v_mul_f64", out, cus);
  sweep<2>("v_add_f64", out, cus);
  sweep<3>("v_fma_f64", out, cus);
  sweep<4>("2 x v_mov_b32", out, cus);
  sweep<5>("2 x v_mov_b32_dpp bank-masked", out, cus);
  sweep<6>("2 x v_cndmask_b32", out, cus);
  sweep<7>("ds_read_b64", out, cus);
  sweep<8>("v_add_u32", out, cus);
  sweep<9>("v_mul_f64 -> next MFMA operand", out, cus);
  sweep<10>("ds_read_b64, used 4 groups later", out, cus);
  sweep<11>("ds_read_b128, used 4 groups later", out, cus);
  return 0;
}
