// SANITIZER HARNESS ONLY -- never linked into libcluster_hip.so, never shipped, never used by tests that claim parity.
//
// A host-memory stand-in for the two dozen HIP runtime entry points the library's HOST code calls, so that the host
// translation units (lc_ctx.cpp, lc_comm.cpp, lc_engine.cpp, lc_topic.cpp, lc_capi.cpp and the host halves of the
// .hip files: launch planners, grids, LDS grants) can be built with -fsanitize=address,undefined / -fsanitize=thread
// and driven on a machine without a GPU (tools/sanitize_host.sh).  "Device" memory is host memory, copies are
// memcpy, streams and events are tokens -- and KERNELS DO NOT RUN: hipLaunchKernel fails with hipErrorNoDevice, so
// nothing here can produce a result of the data path.  What it lets the sanitizers see is the concurrency the host
// side added over the reference's `omp critical` / `omp atomic` (src/cluster.cpp:77, 412): the M-step worker pool,
// the per-thread block cache and its ownership tags, and the shared-memory rendezvous / barrier of the host-staged
// all-reduce.
//
// LC_STUB_DEVICES=n makes hipGetDeviceCount report n devices (default 0: lc_ctx_create fails with LC_EHIP exactly as
// on a GPU-less host, which is what tests/test_host.py::test_no_cpu_fallback expects of the sanitized library too).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace {
int stub_devices() {
  const char* e = std::getenv("LC_STUB_DEVICES");
  return e ? std::atoi(e) : 0;
}
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;  // what hipGetLastError reports (and clears): set by a launch
std::atomic<size_t> g_live_bytes{0};
}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) {
  *n = stub_devices();
  return *n > 0 ? hipSuccess : hipErrorNoDevice;
}
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= stub_devices()) return hipErrorInvalidDevice;
  t_device = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
  *d = t_device;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorNoDevice: return "no ROCm-capable device is detected (host stub)";
    case hipErrorOutOfMemory: return "out of memory (host stub)";
    default: return "HIP error (host stub)";
  }
}
hipError_t hipGetLastError(void) {
  const hipError_t e = t_last;
  t_last = hipSuccess;
  return e;
}

hipError_t hipMalloc(void** p, size_t n) {
  if (stub_devices() <= 0) return hipErrorNoDevice;
  *p = std::malloc(n ? n : 1);
  if (!*p) return hipErrorOutOfMemory;
  g_live_bytes += n;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) {
  if (stub_devices() <= 0) return hipErrorNoDevice;
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  *total_b = (size_t)8 << 30;
  *free_b = (size_t)6 << 30;
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
  if (n) std::memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
  if (n) std::memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) {
  if (n) std::memset(dst, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) {
  *s = reinterpret_cast<hipStream_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
  std::free(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = reinterpret_cast<hipEvent_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  std::free(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.0f;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* numBlocks, const void*, int, size_t) {  // (suffstat_plan's question)
  if (numBlocks) *numBlocks = 2;
  return hipSuccess;
}
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorNotSupported; }
hipError_t hipMemcpyFromSymbol(void*, const void*, size_t, size_t, hipMemcpyKind) { return hipErrorNotSupported; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* prop, int) {
  std::memset(prop, 0, sizeof(*prop));
  prop->multiProcessorCount = 256;
  prop->sharedMemPerBlock = 160 * 1024;
  prop->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  std::strcpy(prop->gcnArchName, "gfx950");
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
  *v = 256;
  return hipSuccess;
}

// the host halves of the .hip files (compiled with --cuda-host-only): registration is a no-op, launches fail
void** __hipRegisterFatBinary(const void*) {
  static void* handle[1] = {nullptr};
  return handle;
}
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, dim3*, dim3*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3*, dim3*, size_t*, hipStream_t*) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) {
  t_last = hipErrorNoDevice;
  return hipErrorNoDevice;
}

}  // extern "C"
