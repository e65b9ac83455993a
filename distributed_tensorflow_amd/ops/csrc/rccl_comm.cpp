// See rccl_comm.h.
#include "rccl_comm.h"

#include <dlfcn.h>

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
  if (comm_ && g_api.CommDestroy) g_api.CommDestroy(comm_);
}

void RcclComm::all_reduce(void* buf, size_t count, int dt, hipStream_t s) {
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
  check(api().Send(buf, count, nccl_type(dt), peer, comm_, s), "ncclSend");
}

void RcclComm::recv(void* buf, size_t count, int dt, int peer, hipStream_t s) {
  check(api().Recv(buf, count, nccl_type(dt), peer, comm_, s), "ncclRecv");
}

void RcclComm::group_start() { check(api().GroupStart(), "ncclGroupStart"); }
void RcclComm::group_end() { check(api().GroupEnd(), "ncclGroupEnd"); }

}  // namespace dtf
