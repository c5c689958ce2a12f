// Probe: is the price of a VALU instruction next to the matrix pipe paid per INSTRUCTION or per SWITCH between the two
// kinds?  tools/mfma_mix_probe.hip: one v_mul_f64 in front of every 8 v_mfma_f64_4x4x4_4b costs ~ 11 clocks;
// tools/mfma_coissue_probe.hip: the same instruction from ANOTHER wave costs 4.5-5.5 (and never overlaps).  Here the ratio
// stays 1 VALU : 8 MFMAs and the grouping changes: 1 + 8, 2 + 16, 4 + 32, 8 + 64 per trip of 64 MFMAs, for
// independent v_mul_f64 and for v_fma_f64 reading an accumulator the MFMAs wrote.
// Prints 2.4 GHz clocks per MFMA and SIMD (wall time) at 1, 2, 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_batch_probe.bin tools/mfma_batch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define M(acc) "v_mfma_f64_4x4x4_4b_f64 " acc ", v[32:33], v[36:37], " acc "\n"
#define G8 M("v[64:65]") M("v[66:67]") M("v[68:69]") M("v[70:71]") M("v[72:73]") M("v[74:75]") M("v[76:77]") M("v[78:79]")
#define V1(d) "v_mul_f64 " d ", v[44:45], v[46:47]\n"
#define W1(d, acc) "v_fma_f64 " d ", " acc ", " acc ", " d "\n"
#define CLOB "v32","v33","v36","v37","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"

// KIND 0: independent v_mul_f64; 1: v_fma_f64 on an accumulator.  GRP: VALU instructions per group (1, 2, 4, 8); a trip = 64 MFMAs + 8 VALU
template <int KIND, int GRP>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
  asm volatile("v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3e000000\n"
               "v_mov_b32 v40,0\n v_mov_b32 v41,0\n v_mov_b32 v42,0\n v_mov_b32 v43,0\n v_mov_b32 v44,0\n v_mov_b32 v45,0x3ff00000\n v_mov_b32 v46,0\n v_mov_b32 v47,0x3ff00000\n"
               "v_mov_b32 v48,0\n v_mov_b32 v49,0\n v_mov_b32 v50,0\n v_mov_b32 v51,0\n v_mov_b32 v52,0\n v_mov_b32 v53,0\n v_mov_b32 v54,0\n v_mov_b32 v55,0\n v_mov_b32 v56,0\n v_mov_b32 v57,0\n"
               "v_mov_b32 v64,0\n v_mov_b32 v65,0\n v_mov_b32 v66,0\n v_mov_b32 v67,0\n v_mov_b32 v68,0\n v_mov_b32 v69,0\n v_mov_b32 v70,0\n v_mov_b32 v71,0\n"
               "v_mov_b32 v72,0\n v_mov_b32 v73,0\n v_mov_b32 v74,0\n v_mov_b32 v75,0\n v_mov_b32 v76,0\n v_mov_b32 v77,0\n v_mov_b32 v78,0\n v_mov_b32 v79,0\n" ::: CLOB);
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0 && GRP == 1) asm volatile(V1("v[40:41]") G8 V1("v[42:43]") G8 V1("v[48:49]") G8 V1("v[50:51]") G8 V1("v[52:53]") G8 V1("v[54:55]") G8 V1("v[56:57]") G8 V1("v[40:41]") G8 ::: CLOB);
    if constexpr (KIND == 0 && GRP == 2) asm volatile(V1("v[40:41]") V1("v[42:43]") G8 G8 V1("v[48:49]") V1("v[50:51]") G8 G8 V1("v[52:53]") V1("v[54:55]") G8 G8 V1("v[56:57]") V1("v[40:41]") G8 G8 ::: CLOB);
    if constexpr (KIND == 0 && GRP == 4) asm volatile(V1("v[40:41]") V1("v[42:43]") V1("v[48:49]") V1("v[50:51]") G8 G8 G8 G8 V1("v[52:53]") V1("v[54:55]") V1("v[56:57]") V1("v[40:41]") G8 G8 G8 G8 ::: CLOB);
    if constexpr (KIND == 0 && GRP == 8) asm volatile(V1("v[40:41]") V1("v[42:43]") V1("v[48:49]") V1("v[50:51]") V1("v[52:53]") V1("v[54:55]") V1("v[56:57]") V1("v[40:41]") G8 G8 G8 G8 G8 G8 G8 G8 ::: CLOB);
    if constexpr (KIND == 1 && GRP == 1) asm volatile(W1("v[40:41]", "v[64:65]") G8 W1("v[42:43]", "v[66:67]") G8 W1("v[48:49]", "v[68:69]") G8 W1("v[50:51]", "v[70:71]") G8 W1("v[52:53]", "v[72:73]") G8 W1("v[54:55]", "v[74:75]") G8 W1("v[56:57]", "v[76:77]") G8 W1("v[40:41]", "v[78:79]") G8 ::: CLOB);
    if constexpr (KIND == 1 && GRP == 8) asm volatile("s_nop 7\n s_nop 7\n" W1("v[40:41]", "v[64:65]") W1("v[42:43]", "v[66:67]") W1("v[48:49]", "v[68:69]") W1("v[50:51]", "v[70:71]") W1("v[52:53]", "v[72:73]") W1("v[54:55]", "v[74:75]") W1("v[56:57]", "v[76:77]") W1("v[40:41]", "v[78:79]") G8 G8 G8 G8 G8 G8 G8 G8 ::: CLOB);
    if constexpr (KIND == 1 && GRP == 4) asm volatile("s_nop 7\n s_nop 7\n" W1("v[40:41]", "v[64:65]") W1("v[42:43]", "v[66:67]") W1("v[48:49]", "v[68:69]") W1("v[50:51]", "v[70:71]") G8 G8 G8 G8 "s_nop 7\n s_nop 7\n" W1("v[52:53]", "v[72:73]") W1("v[54:55]", "v[74:75]") W1("v[56:57]", "v[76:77]") W1("v[40:41]", "v[78:79]") G8 G8 G8 G8 ::: CLOB);
  }
  double s;
  asm volatile("v_add_f64 %0, v[64:65], v[40:41]" : "=v"(s) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int GRP>
static void run(const char* name, double* dout) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("%-64s", name);
  for (int w = 1; w <= 3; ++w) {
    const int iters = 10000, blocks = p.multiProcessorCount * w;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, GRP>), dim3(blocks), dim3(256), 32768, 0, dout, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, GRP>), dim3(blocks), dim3(256), 32768, 0, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_mfma = ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * w);
    printf("  %dw: %6.2f (%5.1f/VALU)", w, per_mfma, (per_mfma - 16.25) * 8.0);
  }
  printf("\n");
}

int main() {
  double* dout; CK(hipMalloc(&dout, 8 << 20));
  printf("clocks per MFMA and SIMD at 2.4 GHz (and the clocks each VALU instruction adds over the bare stream's 16.25)\n");
  run<0, 1>("v_mul_f64, independent:   (1 VALU + 8 MFMA) x 8", dout);
  run<0, 2>("                          (2 VALU + 16 MFMA) x 4", dout);
  run<0, 4>("                          (4 VALU + 32 MFMA) x 2", dout);
  run<0, 8>("                          8 VALU + 64 MFMA", dout);
  run<1, 1>("v_fma_f64 on an accumulator: (1 VALU + 8 MFMA) x 8", dout);
  run<1, 4>("                          (wait, 4 VALU + 32 MFMA) x 2", dout);
  run<1, 8>("                          wait, 8 VALU + 64 MFMA", dout);
  return 0;
}
