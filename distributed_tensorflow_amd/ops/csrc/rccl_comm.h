// Native RCCL communicator of the engine (collectives issued from C++, stream-ordered, usable as
// launch-plan entries and therefore capturable into a hipGraph).
//
// The reference's four transports (in-process NCCL for MirroredStrategy, CollectiveOps ring /
// NCCL for MultiWorkerMirrored, gRPC for the parameter server, MPI + NCCL for Horovod:
// `README.md:61-77`, `ps_server/log1.log:82-84`) map onto one communicator per job.  RCCL is
// loaded at run time (dlopen of the library PyTorch ships, so that a process holds one copy);
// the extension itself has no link-time dependency on it.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace dtf {

class RcclComm {
 public:
  // 128 opaque bytes from rank 0 that every rank passes to the constructor
  static std::string unique_id();
  static void load_library(const std::string& path);   // before anything else; "" = default search

  RcclComm(const std::string& id, int nranks, int rank);
  ~RcclComm();
  RcclComm(const RcclComm&) = delete;
  RcclComm& operator=(const RcclComm&) = delete;

  int rank() const { return rank_; }
  int size() const { return nranks_; }

  // element types: 0 bf16, 1 fp16, 2 fp32 (the engine's codes)
  void all_reduce(void* buf, size_t count, int dt, hipStream_t s);
  // reduce-scatter + all-gather in place: every rank reduces 1/size of the buffer and the
  // shards travel over all links at once (count must be a multiple of size)
  void all_reduce_direct(void* buf, size_t count, int dt, hipStream_t s);
  // picks by message size: one-shot all_reduce below `direct_min_bytes`, direct above
  void all_reduce_auto(void* buf, size_t count, int dt, hipStream_t s);
  void reduce_scatter(const void* send, void* recv, size_t recv_count, int dt, hipStream_t s);
  void all_gather(const void* send, void* recv, size_t send_count, int dt, hipStream_t s);
  void broadcast(void* buf, size_t count, int dt, int root, hipStream_t s);
  void reduce(void* buf, size_t count, int dt, int root, hipStream_t s);
  void send(const void* buf, size_t count, int dt, int peer, hipStream_t s);
  void recv(void* buf, size_t count, int dt, int peer, hipStream_t s);
  static void group_start();
  static void group_end();

  size_t direct_min_bytes = 1 << 20;
  uint64_t collectives = 0;       // issued so far (tests, logs)

 private:
  void* comm_ = nullptr;
  int nranks_ = 1, rank_ = 0;
};

}  // namespace dtf
