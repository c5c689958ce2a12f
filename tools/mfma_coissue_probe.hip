// Probe: do the fp64 matrix pipe and the fp64 vector ALU of a SIMD run side by side when the two instruction streams
// come from DIFFERENT waves?  (tools/mfma_mix_probe.hip priced a VALU instruction inside an MFMA stream of the SAME wave:
// ~ 11 clocks per v_mul/fma_f64 with 1, 2 or 3 such waves per SIMD.  Round 4's verdict asks whether waves specialised by
// role -- some only MFMAs, some only VALU -- would overlap instead.)
// One block per CU (LDS request > 80 KB), 256 * NW threads: waves 0..3 are role A (64 independent
// v_mfma_f64_4x4x4_4b per trip), waves 4..4 NW - 1 role B (64 VALU instructions per trip over 8 independent chains).
// Waves of a block go to the SIMDs round-robin, so every SIMD holds one A wave and NW - 1 B waves.
// Three launches per B kind: A alone, B alone, both.  Side by side: T(both) ~ max; one shared pipe: T(both) ~ sum.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_coissue_probe.bin tools/mfma_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define M(acc) "v_mfma_f64_4x4x4_4b_f64 " acc ", v[32:33], v[36:37], " acc "\n"
#define G8 M("v[64:65]") M("v[66:67]") M("v[68:69]") M("v[70:71]") M("v[72:73]") M("v[74:75]") M("v[76:77]") M("v[78:79]")
#define F(acc) "v_fma_f64 " acc ", v[32:33], v[36:37], " acc "\n"
#define F8 F("v[64:65]") F("v[66:67]") F("v[68:69]") F("v[70:71]") F("v[72:73]") F("v[74:75]") F("v[76:77]") F("v[78:79]")
#define S(acc) "v_fma_f32 " acc ", v32, v36, " acc "\n"
#define S8 S("v64") S("v66") S("v68") S("v70") S("v72") S("v74") S("v76") S("v78")
#define I(acc) "v_add_u32 " acc ", v32, " acc "\n"
#define I8 I("v64") I("v66") I("v68") I("v70") I("v72") I("v74") I("v76") I("v78")
#define X8(S_) S_ S_ S_ S_ S_ S_ S_ S_
#define CLOB "v32","v33","v36","v37","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"

// BK: 0 = v_fma_f64, 1 = v_fma_f32, 2 = v_add_u32;   roles: bit 0 = A runs, bit 1 = B runs
template <int BK>
__global__ void k(double* out, int itA, int itB, int roles) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6;
  asm volatile("v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3e000000\n"
               "v_mov_b32 v64,0\n v_mov_b32 v65,0\n v_mov_b32 v66,0\n v_mov_b32 v67,0\n v_mov_b32 v68,0\n v_mov_b32 v69,0\n v_mov_b32 v70,0\n v_mov_b32 v71,0\n"
               "v_mov_b32 v72,0\n v_mov_b32 v73,0\n v_mov_b32 v74,0\n v_mov_b32 v75,0\n v_mov_b32 v76,0\n v_mov_b32 v77,0\n v_mov_b32 v78,0\n v_mov_b32 v79,0\n" ::: CLOB);
  if (wave < 4) {
    if (roles & 1)
      for (int it = 0; it < itA; ++it) asm volatile(X8(G8) ::: CLOB);
  } else if (roles & 2) {
    for (int it = 0; it < itB; ++it) {
      if constexpr (BK == 0) asm volatile(X8(F8) ::: CLOB);
      if constexpr (BK == 1) asm volatile(X8(S8) ::: CLOB);
      if constexpr (BK == 2) asm volatile(X8(I8) ::: CLOB);
    }
  }
  double s;
  asm volatile("v_add_f64 %0, v[64:65], v[66:67]" : "=v"(s) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) lds[0] = s;
}

template <int BK>
static float launch(double* dout, int nw, int itA, int itB, int roles, int cus) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<BK>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipLaunchKernelGGL(k<BK>, dim3(cus), dim3(256 * nw), 96 * 1024, 0, dout, 50, 50, roles);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<BK>, dim3(cus), dim3(256 * nw), 96 * 1024, 0, dout, itA, itB, roles);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms;
}

template <int BK>
static void run(const char* name, double* dout, int cus) {
  for (int nw = 2; nw <= 4; ++nw) {  // 1 A wave + (nw - 1) B waves per SIMD
    const int itA = 20000;
    // B's trip count such that B alone takes about as long as A alone (measured below, then rescaled once)
    int itB = 20000;
    const float a = launch<BK>(dout, nw, itA, itB, 1, cus);
    float b = launch<BK>(dout, nw, itA, itB, 2, cus);
    itB = (int)(itB * a / b);
    b = launch<BK>(dout, nw, itA, itB, 2, cus);
    const float ab = launch<BK>(dout, nw, itA, itB, 3, cus);
    const double clkA = a * 1e-3 * 2.4e9 / (itA * 64.0), clkB = b * 1e-3 * 2.4e9 / ((double)itB * 64.0 * (nw - 1));
    printf("%-12s 1 MFMA wave + %d VALU waves/SIMD: A alone %7.3f ms (%5.2f clk/MFMA)  B alone %7.3f ms (%5.2f clk/instr/SIMD)  both %7.3f ms  => overlap %.2f (1 = side by side, 0 = one pipe)\n",
           name, nw - 1, a, clkA, b, clkB, ab, (a + b - ab) / (a < b ? a : b));
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  double* dout; CK(hipMalloc(&dout, 8 << 20));
  run<0>("v_fma_f64", dout, p.multiProcessorCount);
  run<1>("v_fma_f32", dout, p.multiProcessorCount);
  run<2>("v_add_u32", dout, p.multiProcessorCount);
  return 0;
}
