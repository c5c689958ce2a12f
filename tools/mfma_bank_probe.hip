// Probe: does the register-file position of a v_mfma_f64_4x4x4_4b's operand pairs change its rate?  A 64-bit operand
// sits in an even-aligned pair v[2n:2n+1]; pairs alternate between two halves of the VGPR banks (n even / n odd).
// 64 MFMAs per loop trip over 8 accumulators, every combination of pair parity for A, B and the accumulators,
// 1 to 3 waves per SIMD; prints shader clocks per MFMA and SIMD (16 = the pipe's limit).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_bank_probe.bin tools/mfma_bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define M(acc, a, b) "v_mfma_f64_4x4x4_4b_f64 " acc ", " a ", " b ", " acc "\n"
// accumulators: parity 0 -> pairs 32,34,..,46 = v64,v68,...; parity 1 -> pairs 33,35,.. = v66,v70,...
#define ACC0(a, b) M("v[64:65]", a, b) M("v[68:69]", a, b) M("v[72:73]", a, b) M("v[76:77]", a, b) M("v[80:81]", a, b) M("v[84:85]", a, b) M("v[88:89]", a, b) M("v[92:93]", a, b)
#define ACC1(a, b) M("v[66:67]", a, b) M("v[70:71]", a, b) M("v[74:75]", a, b) M("v[78:79]", a, b) M("v[82:83]", a, b) M("v[86:87]", a, b) M("v[90:91]", a, b) M("v[94:95]", a, b)
#define ACCM(a, b) M("v[64:65]", a, b) M("v[66:67]", a, b) M("v[68:69]", a, b) M("v[70:71]", a, b) M("v[72:73]", a, b) M("v[74:75]", a, b) M("v[76:77]", a, b) M("v[78:79]", a, b)
#define A0 "v[32:33]"
#define A1 "v[34:35]"
#define B0 "v[36:37]"
#define B1 "v[38:39]"
#define X8(S) S S S S S S S S
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"

template <int CFG>
__global__ void __launch_bounds__(256) k(double* out, int iters, long long* clk) {
  asm volatile("v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n"
               "v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3e000000\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0x3e000000\n" ::: CLOB);
  asm volatile("v_mov_b32 v64,0\n v_mov_b32 v65,0\n v_mov_b32 v66,0\n v_mov_b32 v67,0\n v_mov_b32 v68,0\n v_mov_b32 v69,0\n v_mov_b32 v70,0\n v_mov_b32 v71,0\n"
               "v_mov_b32 v72,0\n v_mov_b32 v73,0\n v_mov_b32 v74,0\n v_mov_b32 v75,0\n v_mov_b32 v76,0\n v_mov_b32 v77,0\n v_mov_b32 v78,0\n v_mov_b32 v79,0\n"
               "v_mov_b32 v80,0\n v_mov_b32 v81,0\n v_mov_b32 v82,0\n v_mov_b32 v83,0\n v_mov_b32 v84,0\n v_mov_b32 v85,0\n v_mov_b32 v86,0\n v_mov_b32 v87,0\n"
               "v_mov_b32 v88,0\n v_mov_b32 v89,0\n v_mov_b32 v90,0\n v_mov_b32 v91,0\n v_mov_b32 v92,0\n v_mov_b32 v93,0\n v_mov_b32 v94,0\n v_mov_b32 v95,0\n" ::: CLOB);
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (CFG == 0) asm volatile(X8(ACC0(A0, B0)) ::: CLOB);
    if constexpr (CFG == 1) asm volatile(X8(ACC0(A0, B1)) ::: CLOB);
    if constexpr (CFG == 2) asm volatile(X8(ACC0(A1, B1)) ::: CLOB);
    if constexpr (CFG == 3) asm volatile(X8(ACC1(A0, B0)) ::: CLOB);
    if constexpr (CFG == 4) asm volatile(X8(ACC1(A0, B1)) ::: CLOB);
    if constexpr (CFG == 5) asm volatile(X8(ACCM(A0, B0)) ::: CLOB);
    if constexpr (CFG == 6) asm volatile(X8(ACCM(A0, B1)) ::: CLOB);
  }
  const long long c1 = clock64();
  double s;
  asm volatile("v_add_f64 %0, v[64:65], v[66:67]" : "=v"(s) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int CFG>
static void run(const char* name, double* dout, long long* dclk) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  for (int w = 1; w <= 3; ++w) {
    const int iters = 20000, blocks = p.multiProcessorCount * w;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<CFG>, dim3(blocks), dim3(256), 0, 0, dout, 100, dclk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CFG>, dim3(blocks), dim3(256), 0, 0, dout, iters, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, dclk, 8, hipMemcpyDeviceToHost));
    const double mf = (double)blocks * 4 * iters * 64.0;
    printf("%-34s waves/SIMD=%d: %7.2f ms %6.2f TFLOP/s  %.2f clock64 ticks per MFMA and SIMD\n", name, w, ms, mf * 512 / ms / 1e9,
           (double)h / ((double)iters * 64.0 * w));
  }
}

int main() {
  double* dout; long long* dclk;
  CK(hipMalloc(&dout, 8 << 20)); CK(hipMalloc(&dclk, 16));
  run<0>("A even, B even, acc even", dout, dclk);
  run<1>("A even, B odd,  acc even", dout, dclk);
  run<2>("A odd,  B odd,  acc even", dout, dclk);
  run<3>("A even, B even, acc odd", dout, dclk);
  run<4>("A even, B odd,  acc odd", dout, dclk);
  run<5>("A even, B even, acc mixed", dout, dclk);
  run<6>("A even, B odd,  acc mixed", dout, dclk);
  return 0;
}
