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

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>

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

  // All-link all-reduce for a full mesh of point-to-point links (xGMI inside a node: 7 links
  // per GPU): every rank owns 1/size of the buffer.  Phase 1, one group of size - 1 sends and
  // receives: rank r hands peer j its copy of shard j and receives j's copy of shard r into a
  // scratch slot - every link carries one shard each way at the same time.  A kernel adds the
  // size - 1 slots to the own shard.  Phase 2, a second group: the reduced shard goes to every
  // peer and theirs come back into place.  A ring all-reduce pushes 2 (size - 1) / size of the
  // buffer through ONE link per direction; this moves 2 / size of it through each of them.
  // fp32 only (gradient buffers); count must be a multiple of size.  The scratch buffer
  // ((size - 1) shards) is allocated at first use, so the first call must not be under capture.
  void all_reduce_mesh(void* buf, size_t count, int dt, hipStream_t s);

  // Failure handling.  RCCL reports a dead peer or a failed link asynchronously:
  // `async_error` polls ncclCommGetAsyncError and returns "" while the communicator is
  // healthy, else the error text - after aborting the communicator (ncclCommAbort), so that
  // kernels blocked in a collective return and the process can exit instead of hanging.
  std::string async_error();
  void abort();
  // Background poll every `interval_ms`: on an error the thread reports it on stderr, aborts the
  // communicator and - with a non-zero `exit_code` - ends the process with that code (a launcher
  // that watches its children then takes the whole job down: fail-fast instead of a hang).
  void start_watchdog(int interval_ms, int exit_code);
  void stop_watchdog();
  bool failed() const { return failed_.load(); }

  size_t direct_min_bytes = 1 << 20;
  uint64_t collectives = 0;       // issued so far (tests, logs)

 private:
  void* comm_ = nullptr;
  int nranks_ = 1, rank_ = 0;
  void* scratch_ = nullptr;
  size_t scratch_bytes_ = 0;
  std::mutex mutex_;              // abort / destroy against the watchdog
  std::atomic<bool> failed_{false};
  std::atomic<bool> stop_{false};
  std::thread watchdog_;
  std::string error_;
};

}  // namespace dtf
