// Collectives of the hot path: the one exchange step per EM iteration is the sum over ranks of the packed
// sufficient statistics (and of [Fz; LL_k]) -- SURVEY 8(e).  The reference has no counterpart: its loop over groups
// (src/cluster.cpp:207-223) is single-process OpenMP; this is the loop being distributed.
//
// Two transports behind one interface:
//   * RCCL  -- ncclAllReduce(ncclDouble, ncclSum) on the context's stream, one rank per GPU over xGMI; ranks are
//              processes (ncclCommInitRank from a broadcast unique id) or the threads of one process
//              (ncclCommInitAll: what LIBCLUSTER_GPUS=N selects inside learnBGMM / learnVDP / learnGMC).
//   * host  -- staged through host memory (POSIX shared memory between processes, the heap between threads), summed
//              in rank order by every rank: bit-identical results on all ranks for ANY placement, including several
//              ranks on one GPU, which RCCL refuses.  Latency-bound like the message itself (1 - 8.5 MB).
// librccl (573 MB) is bound lazily on first use so that single-GPU callers never page it in.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace lcm {

constexpr int UNIQUE_ID_BYTES = 128;  // NCCL_UNIQUE_ID_BYTES

class Comm {
 public:
  virtual ~Comm() = default;
  // Sum `count` doubles in place over all ranks.  dbuf is a device pointer on this rank's GPU; the operation is ordered
  // after the work already enqueued on `stream` and before anything enqueued on it afterwards.  Every rank receives
  // the same bits.  Throws std::runtime_error on failure.
  virtual void allreduce_sum(double* dbuf, int64_t count, hipStream_t stream) = 0;
  // Unblock the other ranks after a local failure (they fail instead of waiting for ever).
  virtual void abort() noexcept = 0;
  virtual const char* kind() const = 0;
  int rank() const { return rank_; }
  int world() const { return world_; }

 protected:
  int rank_ = 0, world_ = 1;
};

// ---- RCCL ------------------------------------------------------------------------------------------------------
bool rccl_available(std::string* why = nullptr);
void rccl_unique_id(void* id128);  // ncclGetUniqueId: one rank calls it and ships the 128 bytes to the others
std::shared_ptr<Comm> rccl_init_rank(const void* id128, int rank, int world, int device);
// one process driving several GPUs: one communicator per entry of `devices` (ncclCommInitAll)
std::vector<std::shared_ptr<Comm>> rccl_init_all(const std::vector<int>& devices);

// ---- host-staged -------------------------------------------------------------------------------------------------
// processes of one node, rendezvous on the POSIX shared-memory object "/lc_comm_<name>"
std::shared_ptr<Comm> host_init_shm(const std::string& name, int rank, int world);
// threads of one process
std::vector<std::shared_ptr<Comm>> host_init_local(int world);

}  // namespace lcm
