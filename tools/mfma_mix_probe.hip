// Probe: what does each kind of instruction cost when it sits inside a stream of v_mfma_f64_4x4x4_4b?  Pure inline asm
// (the bare stream runs 16.25 clocks per MFMA = 98.7 % of the pipe, tools/mfma_bank_probe.hip); one 64-MFMA block per
// loop trip = 8 groups of 8 MFMAs over 8 accumulators, with a pattern of other instructions in front of every group.
// Prints clocks per MFMA and SIMD at 1, 2 and 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_mix_probe.bin tools/mfma_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define M(acc, a, b) "v_mfma_f64_4x4x4_4b_f64 " acc ", " a ", " b ", " acc "\n"
#define G8(a, b) M("v[64:65]", a, b) M("v[66:67]", a, b) M("v[68:69]", a, b) M("v[70:71]", a, b) M("v[72:73]", a, b) M("v[74:75]", a, b) M("v[76:77]", a, b) M("v[78:79]", a, b)
#define A0 "v[32:33]"
#define B0 "v[36:37]"
#define P "v[40:41]"
#define X8(S) S S S S S S S S
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory"

// patterns (one per group of 8 MFMAs); v50 = LDS byte address of this lane
#define PAT0 G8(A0, B0)
#define PAT1 "v_mul_f64 v[42:43], v[44:45], v[46:47]\n" G8(A0, B0)                                   // independent f64 VALU
#define PAT2 "ds_read_b64 v[44:45], %0\n ds_read_b64 v[46:47], %0 offset:640\n" G8(A0, B0)             // two LDS reads, never waited for
#define PAT3 "ds_read_b64 v[44:45], %0\n ds_read_b64 v[46:47], %0 offset:640\n v_mul_f64 v[42:43], v[36:37], v[32:33]\n" G8(A0, B0)
#define PAT4 "v_mul_f64 " P ", v[36:37], v[32:33]\n s_nop 1\n" G8(A0, P)                              // the MFMAs consume the product
#define PAT5 "s_waitcnt lgkmcnt(2)\n v_mul_f64 " P ", v[36:37], v[32:33]\n ds_read_b64 v[44:45], %0\n ds_read_b64 v[46:47], %0 offset:640\n s_nop 0\n" G8(A0, P)  // the feature-GEMM tile
#define PAT6 "s_nop 1\n" G8(A0, B0)
#define PAT7 "ds_read_b64 v[44:45], %0\n" M("v[64:65]", A0, B0) M("v[66:67]", A0, B0) M("v[68:69]", A0, B0) "ds_read_b64 v[46:47], %0 offset:128\n" M("v[70:71]", A0, B0) M("v[72:73]", A0, B0) M("v[74:75]", A0, B0) "ds_read_b64 v[48:49], %0 offset:256\n" M("v[76:77]", A0, B0) M("v[78:79]", A0, B0)  // E-step: a read per 3 MFMAs
#define PAT8 "v_fma_f64 v[42:43], v[64:65], v[64:65], v[42:43]\n" G8(A0, B0)                           // VALU reading an accumulator (E-step squares)
#define PAT9 "v_add_u32 v47, v46, v45\n v_add_u32 v48, v46, v45\n" G8(A0, B0)                          // two 32-bit VALU
#define PAT10 "s_add_u32 s20, s20, 4\n s_addc_u32 s21, s21, 0\n s_cmp_lt_u32 s20, 100\n" G8(A0, B0)   // scalar ALU
#define PAT11 "ds_read2_b64 v[44:47], %0 offset0:8 offset1:12\n" G8(A0, B0)                            // one ds_read2_b64

template <int CFG>
__global__ void __launch_bounds__(256) k(double* out, int iters, long long* clk) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(lds + (threadIdx.x & 15) + 80 * ((threadIdx.x >> 4) & 3));
  asm volatile("v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3e000000\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3e000000\n"
               "v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"
               "v_mov_b32 v64,0\n v_mov_b32 v65,0\n v_mov_b32 v66,0\n v_mov_b32 v67,0\n v_mov_b32 v68,0\n v_mov_b32 v69,0\n v_mov_b32 v70,0\n v_mov_b32 v71,0\n"
               "v_mov_b32 v72,0\n v_mov_b32 v73,0\n v_mov_b32 v74,0\n v_mov_b32 v75,0\n v_mov_b32 v76,0\n v_mov_b32 v77,0\n v_mov_b32 v78,0\n v_mov_b32 v79,0\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n" ::: CLOB, "s20", "s21");
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define RUN(N, PAT) if constexpr (CFG == N) asm volatile(X8(PAT) "s_waitcnt lgkmcnt(0)\n" :: "v"(addr) : CLOB, "s20", "s21", "scc");
    RUN(0, PAT0) RUN(1, PAT1) RUN(2, PAT2) RUN(3, PAT3) RUN(4, PAT4) RUN(5, PAT5) RUN(6, PAT6) RUN(7, PAT7) RUN(8, PAT8) RUN(9, PAT9) RUN(10, PAT10) RUN(11, PAT11)
  }
  const long long c1 = clock64();
  double s;
  asm volatile("v_add_f64 %0, v[64:65], v[42:43]" : "=v"(s) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int CFG>
static void run(const char* name, double* dout, long long* dclk) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("%-58s", name);
  for (int w = 1; w <= 3; ++w) {
    const int iters = 10000, blocks = p.multiProcessorCount * w;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<CFG>, dim3(blocks), dim3(256), 32768, 0, dout, 100, dclk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CFG>, dim3(blocks), dim3(256), 32768, 0, dout, iters, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // (waves are served oldest first: the first block's own clocks say little with several waves per SIMD -- wall time)
    printf("  %dw: %6.2f", w, ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * w));
  }
  printf("   2.4 GHz clocks per MFMA and SIMD (wall time)\n");
}

int main() {
  double* dout; long long* dclk;
  CK(hipMalloc(&dout, 8 << 20)); CK(hipMalloc(&dclk, 16));
  run<0>("bare stream", dout, dclk);
  run<1>("+ 1 independent v_mul_f64 per 8", dout, dclk);
  run<8>("+ 1 v_fma_f64 reading an accumulator per 8", dout, dclk);
  run<4>("+ 1 v_mul_f64 per 8 whose product the 8 MFMAs consume", dout, dclk);
  run<9>("+ 2 v_add_u32 per 8", dout, dclk);
  run<10>("+ 3 scalar ALU per 8", dout, dclk);
  run<6>("+ s_nop 1 per 8", dout, dclk);
  run<2>("+ 2 ds_read_b64 per 8", dout, dclk);
  run<11>("+ 1 ds_read2_b64 per 8", dout, dclk);
  run<7>("+ 1 ds_read_b64 per 3 (E-step ratio)", dout, dclk);
  run<3>("+ 2 ds_read_b64 + 1 independent v_mul_f64 per 8", dout, dclk);
  run<5>("feature-GEMM tile: wait, mul, 2 reads, 8 dependent MFMAs", dout, dclk);
  return 0;
}
