// See rccl_comm.h.
#include "rccl_comm.h"

#include <dlfcn.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>

namespace dtf {
namespace {

// The slice of the RCCL C API used here (rccl.h: ncclUniqueId is 128 opaque bytes; ncclSum = 0;
// ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9).
struct UniqueId { char internal[128]; };
typedef void* comm_t;
typedef int result_t;
constexpr int kSum = 0;

struct Api {
  void* handle = nullptr;
  result_t (*GetUniqueId)(UniqueId*) = nullptr;
  result_t (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
  result_t (*CommDestroy)(comm_t) = nullptr;
  result_t (*CommAbort)(comm_t) = nullptr;
  result_t (*CommGetAsyncError)(comm_t, result_t*) = nullptr;
  const char* (*GetErrorString)(result_t) = nullptr;
  result_t (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*ReduceScatter)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  result_t (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*Reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*GroupStart)() = nullptr;
  result_t (*GroupEnd)() = nullptr;
};

Api g_api;
std::mutex g_mutex;
std::string g_path;

template <typename F>
void sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_api.handle, name));
  if (!f) throw std::runtime_error(std::string("RCCL symbol missing: ") + name);
}

Api& api() {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g_api.handle) return g_api;
  const char* candidates[] = {g_path.c_str(), "librccl.so", "librccl.so.1"};
  for (const char* c : candidates) {
    if (!c[0]) continue;
    g_api.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (g_api.handle) break;
  }
  if (!g_api.handle) throw std::runtime_error(std::string("cannot load RCCL: ") + dlerror());
  sym(g_api.GetUniqueId, "ncclGetUniqueId");
  sym(g_api.CommInitRank, "ncclCommInitRank");
  sym(g_api.CommDestroy, "ncclCommDestroy");
  sym(g_api.CommAbort, "ncclCommAbort");
  sym(g_api.CommGetAsyncError, "ncclCommGetAsyncError");
  sym(g_api.GetErrorString, "ncclGetErrorString");
  sym(g_api.AllReduce, "ncclAllReduce");
  sym(g_api.ReduceScatter, "ncclReduceScatter");
  sym(g_api.AllGather, "ncclAllGather");
  sym(g_api.Broadcast, "ncclBroadcast");
  sym(g_api.Reduce, "ncclReduce");
  sym(g_api.Send, "ncclSend");
  sym(g_api.Recv, "ncclRecv");
  sym(g_api.GroupStart, "ncclGroupStart");
  sym(g_api.GroupEnd, "ncclGroupEnd");
  return g_api;
}

void check(result_t r, const char* what) {
  if (r != 0) {
    const char* msg = g_api.GetErrorString ? g_api.GetErrorString(r) : "?";
    throw std::runtime_error(std::string("RCCL ") + what + " failed: " + msg);
  }
}

int nccl_type(int dt) {
  switch (dt) {
    case 0: return 9;   // bf16
    case 1: return 6;   // fp16
    case 2: return 7;   // fp32
    default: throw std::invalid_argument("RcclComm: unknown element type");
  }
}

size_t elem_size(int dt) { return dt == 2 ? 4 : 2; }

// own[i] += sum over `slots` scratch shards; 4 floats per thread, grid-stride
__global__ void mesh_reduce_kernel(float* __restrict__ own, const float* __restrict__ scratch,
                                   size_t shard, int slots) {
  const size_t n4 = shard / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(own)[i];
    for (int j = 0; j < slots; ++j) {
      const float4 b = reinterpret_cast<const float4*>(scratch + (size_t)j * shard)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(own)[i] = a;
  }
  if (blockIdx.x == 0)                               // ragged tail (shard not a multiple of 4)
    for (size_t i = n4 * 4 + threadIdx.x; i < shard; i += blockDim.x) {
      float a = own[i];
      for (int j = 0; j < slots; ++j) a += scratch[(size_t)j * shard + i];
      own[i] = a;
    }
}

}  // namespace

void RcclComm::load_library(const std::string& path) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g_api.handle) g_path = path;
}

std::string RcclComm::unique_id() {
  UniqueId id;
  std::memset(&id, 0, sizeof(id));
  check(api().GetUniqueId(&id), "ncclGetUniqueId");
  return std::string(id.internal, sizeof(id.internal));
}

RcclComm::RcclComm(const std::string& id, int nranks, int rank) : nranks_(nranks), rank_(rank) {
  if (id.size() != sizeof(UniqueId)) throw std::invalid_argument("RcclComm: unique id must be 128 bytes");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("RcclComm: bad rank / size");
  UniqueId u;
  std::memcpy(u.internal, id.data(), sizeof(u.internal));
  comm_t c = nullptr;
  check(api().CommInitRank(&c, nranks, u, rank), "ncclCommInitRank");
  comm_ = c;
}

RcclComm::~RcclComm() {
  stop_watchdog();
  std::lock_guard<std::mutex> lock(mutex_);
  if (comm_ && !failed_.load() && g_api.CommDestroy) g_api.CommDestroy(comm_);
  comm_ = nullptr;
  if (scratch_) (void)hipFree(scratch_);
}

std::string RcclComm::async_error() {
  std::lock_guard<std::mutex> lock(mutex_);
  if (failed_.load()) return error_;
  if (!comm_) return "";
  result_t state = 0;
  const result_t r = api().CommGetAsyncError(comm_, &state);
  if (r == 0 && state == 0) return "";
  const result_t bad = r != 0 ? r : state;
  error_ = std::string("RCCL asynchronous error on rank ") + std::to_string(rank_) + " of " +
           std::to_string(nranks_) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(bad) : "?");
  failed_.store(true);
  if (g_api.CommAbort) g_api.CommAbort(comm_);       // blocked collective kernels return
  comm_ = nullptr;
  return error_;
}

void RcclComm::abort() {
  std::lock_guard<std::mutex> lock(mutex_);
  if (comm_ && g_api.CommAbort) g_api.CommAbort(comm_);
  comm_ = nullptr;
  if (!failed_.exchange(true)) error_ = "communicator aborted";
}

void RcclComm::start_watchdog(int interval_ms, int exit_code) {
  if (watchdog_.joinable()) return;
  if (interval_ms < 1) interval_ms = 1;
  stop_.store(false);
  watchdog_ = std::thread([this, interval_ms, exit_code]() {
    while (!stop_.load()) {
      const std::string e = async_error();
      if (!e.empty()) {
        std::fprintf(stderr, "[dtf] %s; communicator aborted%s\n", e.c_str(),
                     exit_code ? ", exiting" : "");
        std::fflush(stderr);
        if (exit_code) _exit(exit_code);
        return;
      }
      for (int waited = 0; waited < interval_ms && !stop_.load(); waited += 20)
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  });
}

void RcclComm::stop_watchdog() {
  stop_.store(true);
  if (watchdog_.joinable()) watchdog_.join();
}

void RcclComm::all_reduce_mesh(void* buf, size_t count, int dt, hipStream_t s) {
  if (dt != 2) throw std::invalid_argument("all_reduce_mesh: fp32 buffers only");
  if (failed_.load()) throw std::runtime_error(error_);
  const size_t n = (size_t)nranks_;
  if (count % n) throw std::invalid_argument("all_reduce_mesh: count % size != 0");
  if (n == 1) return;
  const size_t shard = count / n;
  const size_t need = (n - 1) * shard * sizeof(float);
  if (need > scratch_bytes_) {
    if (scratch_) (void)hipFree(scratch_);
    scratch_ = nullptr;
    scratch_bytes_ = 0;
    if (hipMalloc(&scratch_, need) != hipSuccess)
      throw std::runtime_error("all_reduce_mesh: cannot allocate the scratch shards");
    scratch_bytes_ = need;
  }
  float* base = static_cast<float*>(buf);
  float* scratch = static_cast<float*>(scratch_);
  // slot of peer j in the scratch buffer: peers in rank order with the own rank left out
  auto slot = [&](int j) { return (size_t)(j < rank_ ? j : j - 1); };
  check(api().GroupStart(), "ncclGroupStart");
  for (int j = 0; j < nranks_; ++j) {
    if (j == rank_) continue;
    check(api().Send(base + (size_t)j * shard, shard, 7, j, comm_, s), "ncclSend");
    check(api().Recv(scratch + slot(j) * shard, shard, 7, j, comm_, s), "ncclRecv");
  }
  check(api().GroupEnd(), "ncclGroupEnd");
  float* own = base + (size_t)rank_ * shard;
  const int threads = 256;
  size_t blocks = (shard / 4 + threads - 1) / threads;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mesh_reduce_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, own, scratch,
                     shard, nranks_ - 1);
  check(api().GroupStart(), "ncclGroupStart");
  for (int j = 0; j < nranks_; ++j) {
    if (j == rank_) continue;
    check(api().Send(own, shard, 7, j, comm_, s), "ncclSend");
    check(api().Recv(base + (size_t)j * shard, shard, 7, j, comm_, s), "ncclRecv");
  }
  check(api().GroupEnd(), "ncclGroupEnd");
  collectives += 2;
}

void RcclComm::all_reduce(void* buf, size_t count, int dt, hipStream_t s) {
  if (failed_.load()) throw std::runtime_error(error_);
  check(api().AllReduce(buf, buf, count, nccl_type(dt), kSum, comm_, s), "ncclAllReduce");
  ++collectives;
}

void RcclComm::all_reduce_direct(void* buf, size_t count, int dt, hipStream_t s) {
  if (count % (size_t)nranks_) throw std::invalid_argument("all_reduce_direct: count % size != 0");
  const size_t shard = count / (size_t)nranks_;
  char* mine = static_cast<char*>(buf) + (size_t)rank_ * shard * elem_size(dt);
  check(api().ReduceScatter(buf, mine, shard, nccl_type(dt), kSum, comm_, s), "ncclReduceScatter");
  check(api().AllGather(mine, buf, shard, nccl_type(dt), comm_, s), "ncclAllGather");
  collectives += 2;
}

void RcclComm::all_reduce_auto(void* buf, size_t count, int dt, hipStream_t s) {
  if (nranks_ > 1 && count * elem_size(dt) >= direct_min_bytes && count % (size_t)nranks_ == 0)
    all_reduce_direct(buf, count, dt, s);
  else
    all_reduce(buf, count, dt, s);
}

void RcclComm::reduce_scatter(const void* send, void* recv, size_t recv_count, int dt, hipStream_t s) {
  check(api().ReduceScatter(send, recv, recv_count, nccl_type(dt), kSum, comm_, s), "ncclReduceScatter");
  ++collectives;
}

void RcclComm::all_gather(const void* send, void* recv, size_t send_count, int dt, hipStream_t s) {
  check(api().AllGather(send, recv, send_count, nccl_type(dt), comm_, s), "ncclAllGather");
  ++collectives;
}

void RcclComm::broadcast(void* buf, size_t count, int dt, int root, hipStream_t s) {
  check(api().Broadcast(buf, buf, count, nccl_type(dt), root, comm_, s), "ncclBroadcast");
  ++collectives;
}

void RcclComm::reduce(void* buf, size_t count, int dt, int root, hipStream_t s) {
  check(api().Reduce(buf, buf, count, nccl_type(dt), kSum, root, comm_, s), "ncclReduce");
  ++collectives;
}

void RcclComm::send(const void* buf, size_t count, int dt, int peer, hipStream_t s) {
  if (failed_.load()) throw std::runtime_error(error_);
  check(api().Send(buf, count, nccl_type(dt), peer, comm_, s), "ncclSend");
}

void RcclComm::recv(void* buf, size_t count, int dt, int peer, hipStream_t s) {
  if (failed_.load()) throw std::runtime_error(error_);
  check(api().Recv(buf, count, nccl_type(dt), peer, comm_, s), "ncclRecv");
}

void RcclComm::group_start() { check(api().GroupStart(), "ncclGroupStart"); }
void RcclComm::group_end() { check(api().GroupEnd(), "ncclGroupEnd"); }

}  // namespace dtf
