#include "lc_comm.hpp"

#include "lc_kernels.h"  // rank_order_sum

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>  // types and enums only: every call goes through the lazily bound table below
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>
#include <string>

namespace lcm {

namespace {

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("libcluster comm: " + what); }

void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) fail(std::string(what) + ": " + hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------------------------------------
// RCCL, bound on first use
// ---------------------------------------------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  std::string error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // multi-process GPU work on hosts whose driver only supports dmabuf IPC (hipIpcGetMemHandle fails otherwise)
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    // Where to look: an explicit path, the loader's search path, and then NEXT TO THE HIP RUNTIME THIS PROCESS
    // ALREADY USES (a pure C++ caller linked against /opt/rocm/lib/libamdhip64.so, a Python caller on the copy a
    // PyTorch wheel bundles: RCCL must come from the same ROCm tree as the runtime it is going to drive)
    std::vector<std::string> names;
    if (const char* e = std::getenv("LC_RCCL_LIBRARY"); e && *e) names.push_back(e);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    {
      Dl_info info{};
      if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
          dir.resize(slash + 1);
          names.push_back(dir + "librccl.so.1");
          names.push_back(dir + "librccl.so");
        }
      }
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& n : names) {
      api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
      api.error = dlerror();
    }
    if (!api.handle) return;
    auto sym = [&](const char* s) {
      void* p = dlsym(api.handle, s);
      if (!p && api.error.find("missing symbol") == std::string::npos) api.error = std::string("missing symbol ") + s;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommInitAll || !api.CommDestroy || !api.CommAbort ||
        !api.AllReduce || !api.AllGather || !api.GetErrorString) {
      dlclose(api.handle);
      api.handle = nullptr;
    }
  });
  return api;
}

RcclApi& rccl_or_throw() {
  RcclApi& a = rccl();
  if (!a.handle) fail("RCCL is not available (" + (a.error.empty() ? std::string("librccl.so.1 not found") : a.error) + ")");
  return a;
}

void nccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) fail(std::string(what) + ": " + rccl().GetErrorString(r));
}

// LIBCLUSTER_COMM=rccl-gather: the sum over ranks in a FIXED order.  ncclAllReduce adds in the order of its ring / tree
// (chosen per message size and topology), so the low bits of the statistics -- and, through `Fsplit < F`
// (cluster.cpp:473-479), in principle a split decision -- may depend on the number of GPUs.  Here every rank gathers all
// ranks' buffers (ncclAllGather: W x 1 - 8.5 MB, nothing at xGMI rates) and adds them itself, rank 0 first, with the
// very additions of the host transport (HostComm::allreduce_sum): an N-GPU run then equals the host-transport run and,
// with it, every other placement of the same shards, bit for bit.  Default stays ncclAllReduce (one pass, no W-fold buffer).
bool rccl_gather_mode() {
  const char* e = std::getenv("LIBCLUSTER_COMM");
  return e && std::string(e) == "rccl-gather";
}

class RcclComm final : public Comm {
 public:
  RcclComm(ncclComm_t c, int rank, int world, int device) : comm_(c), device_(device), gather_(rccl_gather_mode()) {
    rank_ = rank;
    world_ = world;
  }
  ~RcclComm() override {
    ncclComm_t c = comm_.exchange(nullptr);
    if (c) {
      (void)hipSetDevice(device_);
      (void)rccl().CommDestroy(c);
    }
    if (gbuf_) {
      (void)hipSetDevice(device_);
      (void)hipFree(gbuf_);
    }
  }
  void allreduce_sum(double* dbuf, int64_t count, hipStream_t stream) override {
    if (count <= 0) return;
    ncclComm_t c = comm_.load();
    if (!c) fail("communicator was aborted");
    if (gather_) {
      const size_t need = (size_t)world_ * (size_t)count;
      if (need > gcap_) {  // (grow-only; the stream may still be reading the old block)
        hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
        if (gbuf_) hip_ok(hipFree(gbuf_), "hipFree(gather buffer)");
        gbuf_ = nullptr;
        gcap_ = 0;
        hip_ok(hipMalloc(reinterpret_cast<void**>(&gbuf_), need * sizeof(double)), "hipMalloc(gather buffer)");
        gcap_ = need;
      }
      nccl_ok(rccl().AllGather(dbuf, gbuf_, (size_t)count, ncclDouble, c, stream), "ncclAllGather");
      hip_ok(lck::launch_rank_order_sum(gbuf_, world_, count, dbuf, stream), "rank_order_sum");
      return;
    }
    // in place, on the caller's stream: ordered with the kernels that produced dbuf and with the copy that reads it
    nccl_ok(rccl().AllReduce(dbuf, dbuf, (size_t)count, ncclDouble, ncclSum, c, stream), "ncclAllReduce");
  }
  void abort() noexcept override {  // ncclCommAbort ends collectives in flight: ranks blocked on them fail instead of hanging
    ncclComm_t c = comm_.exchange(nullptr);
    if (c) (void)rccl().CommAbort(c);
  }
  const char* kind() const override { return gather_ ? "rccl-gather" : "rccl"; }

 private:
  std::atomic<ncclComm_t> comm_{nullptr};
  int device_ = 0;
  bool gather_ = false;
  double* gbuf_ = nullptr;  // [world][count] in gather mode
  size_t gcap_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// host-staged: a region [Ctrl | world slots of SLOT doubles] in shared memory or on the heap
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t HOST_MAGIC = 0x4c43434du;          // "LCCM"
constexpr size_t SLOT_DOUBLES = (size_t)2 << 20;      // 16 MB per rank and piece (config 5's message is 8.5 MB)

struct Ctrl {
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> attached;
  std::atomic<uint32_t> count;
  std::atomic<uint32_t> gen;
  std::atomic<uint32_t> aborted;
  std::atomic<uint32_t> ready;  // set by rank 0 once every rank of ITS run is attached (just before the name is removed)
  uint32_t world;
  uint32_t pad0;
  uint64_t slot_doubles;
  char pad[64 - 6 * 4 - 4 - 4 - 8];
};
static_assert(sizeof(Ctrl) == 64, "control block is one cache line");
static_assert(std::atomic<uint32_t>::is_always_lock_free, "process-shared atomics must be lock-free");

struct Region {
  void* base = nullptr;
  size_t bytes = 0;
  bool shm = false;
  std::string shm_name;
  ~Region() {
    if (!base) return;
    if (shm) munmap(base, bytes);
    else std::free(base);
  }
  Ctrl* ctrl() const { return static_cast<Ctrl*>(base); }
  double* slot(int r) const { return reinterpret_cast<double*>(static_cast<char*>(base) + sizeof(Ctrl)) + (size_t)r * SLOT_DOUBLES; }
};

double timeout_seconds() {
  if (const char* e = std::getenv("LC_COMM_TIMEOUT_S")) {
    const double v = std::atof(e);
    if (v > 0) return v;
  }
  return 300.0;
}

class HostComm final : public Comm {
 public:
  HostComm(std::shared_ptr<Region> reg, int rank, int world) : reg_(std::move(reg)) {
    rank_ = rank;
    world_ = world;
  }
  ~HostComm() override {
    if (stage_) (void)hipHostFree(stage_);
  }
  void allreduce_sum(double* dbuf, int64_t count, hipStream_t stream) override {
    if (count <= 0) return;
    if (!stage_) hip_ok(hipHostMalloc(reinterpret_cast<void**>(&stage_), SLOT_DOUBLES * sizeof(double), hipHostMallocDefault),
                        "hipHostMalloc(staging)");
    for (int64_t off = 0; off < count; off += (int64_t)SLOT_DOUBLES) {
      const size_t n = (size_t)std::min<int64_t>((int64_t)SLOT_DOUBLES, count - off);
      hip_ok(hipMemcpyAsync(stage_, dbuf + off, n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy to host");
      hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
      std::memcpy(reg_->slot(rank_), stage_, n * sizeof(double));
      barrier();
      // every rank adds the slots in rank order: the same bits everywhere, whatever the placement
      const double* s0 = reg_->slot(0);
      std::memcpy(stage_, s0, n * sizeof(double));
      for (int r = 1; r < world_; ++r) {
        const double* s = reg_->slot(r);
        for (size_t i = 0; i < n; ++i) stage_[i] += s[i];
      }
      barrier();  // nobody overwrites its slot before everybody has read it
      hip_ok(hipMemcpyAsync(dbuf + off, stage_, n * sizeof(double), hipMemcpyHostToDevice, stream), "copy to device");
      hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");  // stage_ is reused by the next piece / call
    }
  }
  void abort() noexcept override { reg_->ctrl()->aborted.store(1, std::memory_order_release); }
  const char* kind() const override { return reg_->shm ? "host-shm" : "host-local"; }

 private:
  void barrier() {
    Ctrl* c = reg_->ctrl();
    const uint32_t g = c->gen.load(std::memory_order_acquire);
    if (c->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
      c->count.store(0, std::memory_order_relaxed);
      c->gen.store(g + 1, std::memory_order_release);
      return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = timeout_seconds();
    for (unsigned spin = 0; c->gen.load(std::memory_order_acquire) == g; ++spin) {
      if (c->aborted.load(std::memory_order_acquire)) fail("another rank failed (communicator aborted)");
      if (spin < 2000) continue;
      sched_yield();
      if ((spin & 1023) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
        c->aborted.store(1, std::memory_order_release);
        fail("timed out waiting for the other ranks (LC_COMM_TIMEOUT_S)");
      }
    }
    if (c->aborted.load(std::memory_order_acquire)) fail("another rank failed (communicator aborted)");
  }
  std::shared_ptr<Region> reg_;
  double* stage_ = nullptr;
};

size_t region_bytes(int world) { return sizeof(Ctrl) + (size_t)world * SLOT_DOUBLES * sizeof(double); }

void init_ctrl(Ctrl* c, int world) {
  c->attached.store(0);
  c->count.store(0);
  c->gen.store(0);
  c->aborted.store(0);
  c->ready.store(0);
  c->world = (uint32_t)world;
  c->slot_doubles = SLOT_DOUBLES;
  c->magic.store(HOST_MAGIC, std::memory_order_release);
}

}  // namespace

bool rccl_available(std::string* why) {
  RcclApi& a = rccl();
  if (!a.handle && why) *why = a.error;
  return a.handle != nullptr;
}

void rccl_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  nccl_ok(rccl_or_throw().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id128, &id, sizeof(id));
}

std::shared_ptr<Comm> rccl_init_rank(const void* id128, int rank, int world, int device) {
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rank / world out of range");
  RcclApi& a = rccl_or_throw();
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  hip_ok(hipSetDevice(device), "hipSetDevice");
  ncclComm_t c = nullptr;
  nccl_ok(a.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  return std::make_shared<RcclComm>(c, rank, world, device);
}

std::vector<std::shared_ptr<Comm>> rccl_init_all(const std::vector<int>& devices) {
  if (devices.empty()) throw std::invalid_argument("no devices");
  RcclApi& a = rccl_or_throw();
  std::vector<ncclComm_t> cs(devices.size(), nullptr);
  nccl_ok(a.CommInitAll(cs.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
  std::vector<std::shared_ptr<Comm>> out;
  for (size_t r = 0; r < devices.size(); ++r)
    out.push_back(std::make_shared<RcclComm>(cs[r], (int)r, (int)devices.size(), devices[r]));
  return out;
}

std::shared_ptr<Comm> host_init_shm(const std::string& name, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rank / world out of range");
  if (name.empty() || name.find('/') != std::string::npos) throw std::invalid_argument("bad communicator name");
  auto reg = std::make_shared<Region>();
  reg->shm = true;
  reg->shm_name = "/lc_comm_" + name;
  reg->bytes = region_bytes(world);
  const double limit = timeout_seconds();
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit; };
  // Which object does the name refer to right now?  (0: none.)  A run that crashed during its rendezvous leaves an
  // object behind that already carries the magic word: a rank > 0 may open THAT one before rank 0 has replaced it, and
  // would wait there for ranks that never come.  So a waiting rank keeps checking that the name still leads to the
  // object it is attached to, and starts over on the new one otherwise.
  auto current_inode = [&]() -> ino_t {
    const int f = shm_open(reg->shm_name.c_str(), O_RDONLY, 0600);
    if (f < 0) return 0;
    struct stat st;
    const ino_t ino = fstat(f, &st) == 0 ? st.st_ino : 0;
    close(f);
    return ino;
  };
  for (;;) {
    int fd = -1;
    ino_t mine = 0;
    if (rank == 0) {
      // a stale object of a crashed run: POISON it before the name goes (magic cleared, abort flag set), so that a rank
      // of this run that attached to it in the meantime -- it may even complete the dead run's attach count -- is sent
      // back to the name by its magic / inode re-check, or at the latest fails in its first barrier instead of waiting
      // out the timeout
      {
        const int old = shm_open(reg->shm_name.c_str(), O_RDWR, 0600);
        if (old >= 0) {
          struct stat st;
          if (fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(Ctrl)) {
            void* p = mmap(nullptr, sizeof(Ctrl), PROT_READ | PROT_WRITE, MAP_SHARED, old, 0);
            if (p != MAP_FAILED) {
              Ctrl* dead = static_cast<Ctrl*>(p);
              dead->aborted.store(1, std::memory_order_release);
              dead->magic.store(0, std::memory_order_release);
              munmap(p, sizeof(Ctrl));
            }
          }
          close(old);
        }
      }
      shm_unlink(reg->shm_name.c_str());
      fd = shm_open(reg->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) fail("shm_open(" + reg->shm_name + ") failed: " + std::strerror(errno));
      if (ftruncate(fd, (off_t)reg->bytes) != 0) {
        close(fd);
        shm_unlink(reg->shm_name.c_str());
        fail("ftruncate of the shared region failed: " + std::string(std::strerror(errno)));
      }
    } else {
      for (;;) {  // wait for rank 0 to create and size the object
        fd = shm_open(reg->shm_name.c_str(), O_RDWR, 0600);
        if (fd >= 0) {
          struct stat st;
          if (fstat(fd, &st) == 0 && (size_t)st.st_size >= reg->bytes) {
            mine = st.st_ino;
            break;
          }
          close(fd);
          fd = -1;
        }
        if (expired()) fail("timed out waiting for rank 0 to create " + reg->shm_name);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    }
    reg->base = mmap(nullptr, reg->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (reg->base == MAP_FAILED) {
      reg->base = nullptr;
      fail("mmap of the shared region failed: " + std::string(std::strerror(errno)));
    }
    Ctrl* c = reg->ctrl();
    bool stale = false;
    auto replaced = [&] {  // (ranks > 0) the name leads elsewhere AND this object is still waiting for ranks: a left-over
      const ino_t now = current_inode();
      return now != 0 && now != mine;
    };
    if (rank == 0) {
      init_ctrl(c, world);
    } else {
      int tick = 0;
      while (c->magic.load(std::memory_order_acquire) != HOST_MAGIC) {
        if (expired()) fail("timed out waiting for rank 0 to initialise " + reg->shm_name);
        if (++tick % 100 == 0 && replaced()) { stale = true; break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (!stale && (c->world != (uint32_t)world || c->slot_doubles != SLOT_DOUBLES)) {
        if (replaced()) stale = true;
        else fail("ranks disagree about the communicator size");
      }
    }
    if (!stale) {
      // the name is only needed for the rendezvous: once everybody is attached rank 0 removes it
      c->attached.fetch_add(1, std::memory_order_acq_rel);
      int tick = 0;
      while (c->attached.load(std::memory_order_acquire) < (uint32_t)world) {
        if (expired()) fail("timed out waiting for all ranks to attach to " + reg->shm_name);
        if (rank != 0 && ++tick % 100 == 0 && replaced()) { stale = true; break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
    }
    // The attach count alone does not say WHOSE run this is: a left-over object of a run that died during its
    // rendezvous with attached == world - 1 lets a rank > 0 through the loop above without a single look at the name.
    // So rank 0 confirms: `ready` is set only by the rank 0 that created the object, once all ranks of its run are
    // there -- a dead run's object never gets it (its rank 0 would have removed the name right afterwards), and the
    // ranks waiting here move on to the new object as soon as the name leads to it (or its magic word is cleared).
    if (!stale) {
      if (rank == 0) {
        c->ready.store(1, std::memory_order_release);
      } else {
        int tick = 0;
        while (c->ready.load(std::memory_order_acquire) == 0) {
          if (c->magic.load(std::memory_order_acquire) != HOST_MAGIC) { stale = true; break; }
          if (expired()) fail("timed out waiting for rank 0 to confirm the rendezvous on " + reg->shm_name);
          if (++tick % 100 == 0 && replaced()) { stale = true; break; }
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
      }
    }
    if (!stale) break;
    munmap(reg->base, reg->bytes);  // a left-over of a crashed run: attach to the object rank 0 has created since
    reg->base = nullptr;
  }
  if (rank == 0) shm_unlink(reg->shm_name.c_str());
  return std::make_shared<HostComm>(reg, rank, world);
}

std::vector<std::shared_ptr<Comm>> host_init_local(int world) {
  if (world < 1) throw std::invalid_argument("world must be >= 1");
  auto reg = std::make_shared<Region>();
  reg->bytes = region_bytes(world);
  reg->base = std::aligned_alloc(64, (reg->bytes + 63) / 64 * 64);
  if (!reg->base) throw std::bad_alloc();
  new (reg->base) Ctrl();
  init_ctrl(reg->ctrl(), world);
  std::vector<std::shared_ptr<Comm>> out;
  for (int r = 0; r < world; ++r) out.push_back(std::make_shared<HostComm>(reg, r, world));
  return out;
}

}  // namespace lcm
