// Can the host write kernel parameters straight into DEVICE memory (large BAR), skipping the copy command?
// hipExtMallocWithFlags(hipDeviceMallocFinegrained) and plain hipMalloc, each tried in a child process (a fault is an answer).
//   hipcc --offload-arch=gfx950 -O2 tools/bar_write_probe.hip -o tools/bar_write_probe.bin && tools/bar_write_probe.bin
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void sum_kernel(const double* p, int n, double* out) {  // every block sums the whole parameter block; block 0 reports
  __shared__ double part[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int i = 0; i < 256; ++i) t += part[i];
    if (blockIdx.x == 0) out[0] = t;
  }
}
static int run(int mode) {
  const int n = 1408;  // 11 KB
  double* d = nullptr;
  if (mode == 0) CK(hipExtMallocWithFlags((void**)&d, n * 8, hipDeviceMallocFinegrained));
  else if (mode == 1) CK(hipMalloc((void**)&d, n * 8));
  else CK(hipExtMallocWithFlags((void**)&d, n * 8, hipDeviceMallocUncached));
  double* out;
  CK(hipHostMalloc((void**)&out, 64, 0));
  std::vector<double> h(n);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  double worst = 0, tw = 0, tl = 0;
  for (int it = 0; it < 200; ++it) {
    for (int i = 0; i < n; ++i) h[i] = it + i * 1e-3;
    memset(out, 0, 64);
    auto t0 = std::chrono::steady_clock::now();
    memcpy(d, h.data(), n * 8);  // host stores into device memory
    __builtin_ia32_sfence();
    auto t1 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(sum_kernel, dim3(512), dim3(256), 0, st, d, n, out);
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    double want = 0;
    for (int i = 0; i < n; ++i) want += h[i];
    const double got = out[0];
    worst = std::max(worst, std::abs(got - want) / want);
    if (it >= 20) {
      tw += std::chrono::duration<double, std::micro>(t1 - t0).count();
      tl += std::chrono::duration<double, std::micro>(t2 - t1).count();
    }
  }
  // the same with a copy command
  double tc = 0;
  double* d2;
  CK(hipMalloc((void**)&d2, n * 8));
  double* hp;
  CK(hipHostMalloc((void**)&hp, n * 8, 0));
  for (int it = 0; it < 200; ++it) {
    memcpy(hp, h.data(), n * 8);
    auto t0 = std::chrono::steady_clock::now();
    CK(hipMemcpyAsync(d2, hp, n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sum_kernel, dim3(512), dim3(256), 0, st, d2, n, out);
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    if (it >= 20) tc += std::chrono::duration<double, std::micro>(t2 - t0).count();
  }
  printf("mode %d: host write of 11 KB %.2f us, launch + sync %.2f us, worst rel error %.1e  |  copy command + launch + sync %.2f us\n", mode, tw / 180,
         tl / 180, worst, tc / 180);
  return 0;
}
int main() {
  const char* names[3] = {"hipExtMallocWithFlags(Finegrained)", "hipMalloc", "hipExtMallocWithFlags(Uncached)"};
  for (int mode = 0; mode < 3; ++mode) {
    fflush(stdout);
    pid_t c = fork();
    if (c == 0) { int r = run(mode); fflush(stdout); _exit(r); }
    int stt = 0;
    waitpid(c, &stt, 0);
    if (WIFSIGNALED(stt)) printf("mode %d (%s): child died with signal %d -- not host-writable\n", mode, names[mode], WTERMSIG(stt));
    else printf("mode %d (%s): exit %d\n", mode, names[mode], WEXITSTATUS(stt));
  }
  return 0;
}
