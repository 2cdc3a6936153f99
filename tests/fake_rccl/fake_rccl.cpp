// TEST infrastructure: an in-process stand-in for librccl's point-to-point API, so that glx's RCCL
// transport (csrc/glx_comm.hip, RcclComm: groups of ncclSend / ncclRecv, message rounds, the count all-gather)
// can be exercised with world size > 1 on ONE GPU, where real RCCL refuses to run several ranks.
// Ranks are host threads of one process; every operation completes before the call returns:
//   ncclSend   posts {pointer, bytes} to the (me -> peer) queue once the caller's stream has drained,
//   ncclRecv   waits for the head of the (peer -> me) queue, checks the size, copies device-to-device,
//   a group    posts all its sends first, then serves its receives in call order, then waits until its sends
//              were consumed (the sender may reuse its buffers after ncclGroupEnd + a stream sync, as with RCCL),
//   ncclAllGather  is a rendezvous of all ranks of the communicator.
// Only peers that have something to exchange need to call -- like RCCL, unlike a barrier-based fabric.
// Loaded instead of librccl through GLX_RCCL_LIBRARY (tests/scripts/fake_rccl_check.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct SendDesc {
  const void* ptr;
  size_t bytes;
  bool consumed = false;
};

struct Fabric {
  int world = 0, joined = 0, left = 0;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::deque<SendDesc*>> q;  // [src * world + dst]
  // all-gather rendezvous
  std::vector<const void*> ag_ptr;
  int ag_arrived = 0;
  uint64_t ag_gen = 0;
  int ag_leaving = 0;
};

std::mutex g_m;
std::map<std::string, Fabric*> g_fabrics;
uint64_t g_next_id = 1;

struct FakeComm {
  Fabric* fab;
  int rank;
};

struct Op {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  FakeComm* comm;
  hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat16: case ncclBfloat16: return 2;
    default: return 1;
  }
}

const auto kPatience = std::chrono::seconds(120);

ncclResult_t run_group(std::vector<Op>& ops) {
  // 1. my sends become visible once my streams have drained
  std::vector<SendDesc*> mine;
  for (Op& o : ops) {
    if (!o.send) continue;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    Fabric* f = o.comm->fab;
    SendDesc* d = new SendDesc{o.ptr, o.bytes};
    mine.push_back(d);
    std::lock_guard<std::mutex> lk(f->m);
    f->q[(size_t)o.comm->rank * f->world + o.peer].push_back(d);
    f->cv.notify_all();
  }
  // 2. my receives, in call order
  for (Op& o : ops) {
    if (o.send) continue;
    Fabric* f = o.comm->fab;
    SendDesc* d = nullptr;
    {
      std::unique_lock<std::mutex> lk(f->m);
      auto& q = f->q[(size_t)o.peer * f->world + o.comm->rank];
      if (!f->cv.wait_for(lk, kPatience, [&] { return !q.empty(); })) return ncclSystemError;  // no matching send
      d = q.front();
      q.pop_front();
    }
    if (d->bytes != o.bytes) return ncclInvalidArgument;  // the two ends disagree about a message size
    if (o.bytes) {
      if (hipMemcpyAsync(o.ptr, d->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
      if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    std::lock_guard<std::mutex> lk(f->m);
    d->consumed = true;
    f->cv.notify_all();
  }
  // 3. my sends have been taken
  for (size_t i = 0, k = 0; i < ops.size(); ++i) {
    if (!ops[i].send) continue;
    Fabric* f = ops[i].comm->fab;
    SendDesc* d = mine[k++];
    std::unique_lock<std::mutex> lk(f->m);
    if (!f->cv.wait_for(lk, kPatience, [&] { return d->consumed; })) return ncclSystemError;  // nobody received it
    lk.unlock();
    delete d;
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_m);
  memset(id, 0, sizeof(*id));
  const uint64_t v = g_next_id++;
  memcpy(id->internal, "fake-rccl", 9);
  memcpy(id->internal + 16, &v, 8);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  Fabric* f = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_m);
    const std::string key(id.internal, sizeof(id.internal));
    auto it = g_fabrics.find(key);
    if (it == g_fabrics.end()) {
      f = new Fabric();
      f->world = nranks;
      f->q.resize((size_t)nranks * nranks);
      f->ag_ptr.assign((size_t)nranks, nullptr);
      g_fabrics[key] = f;
    } else {
      f = it->second;
    }
  }
  if (f->world != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  {
    std::unique_lock<std::mutex> lk(f->m);
    ++f->joined;
    f->cv.notify_all();
    if (!f->cv.wait_for(lk, kPatience, [&] { return f->joined >= f->world; })) return ncclSystemError;
  }
  *comm = reinterpret_cast<ncclComm_t>(new FakeComm{f, rank});
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete reinterpret_cast<FakeComm*>(comm);  // fabrics are leaked: a test process is short-lived
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "fake rccl: the two ends of a message disagree (size / world / rank)";
    case ncclSystemError: return "fake rccl: no peer showed up within 120 s";
    default: return "fake rccl: HIP error";
  }
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_group(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (peer < 0 || peer >= c->fab->world || peer == c->rank) return ncclInvalidArgument;
  t_ops.push_back(Op{true, const_cast<void*>(buf), count * type_bytes(t), peer, c, s});
  if (t_depth == 0) {
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
  }
  return ncclSuccess;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (peer < 0 || peer >= c->fab->world || peer == c->rank) return ncclInvalidArgument;
  t_ops.push_back(Op{false, buf, count * type_bytes(t), peer, c, s});
  if (t_depth == 0) {
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  Fabric* f = c->fab;
  const size_t bytes = count * type_bytes(t);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::unique_lock<std::mutex> lk(f->m);
    // the previous round must have been left by everybody
    if (!f->cv.wait_for(lk, kPatience, [&] { return f->ag_leaving == 0; })) return ncclSystemError;
    f->ag_ptr[(size_t)c->rank] = send;
    const uint64_t gen = f->ag_gen;
    if (++f->ag_arrived == f->world) {
      f->ag_arrived = 0;
      f->ag_leaving = f->world;
      ++f->ag_gen;
      f->cv.notify_all();
    } else if (!f->cv.wait_for(lk, kPatience, [&] { return f->ag_gen != gen; })) {
      return ncclSystemError;
    }
  }
  for (int q = 0; q < f->world; ++q) {
    if (hipMemcpyAsync(static_cast<char*>(recv) + (size_t)q * bytes, f->ag_ptr[(size_t)q], bytes, hipMemcpyDeviceToDevice, s) !=
        hipSuccess) {
      return ncclUnhandledCudaError;
    }
  }
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  std::lock_guard<std::mutex> lk(f->m);
  --f->ag_leaving;
  f->cv.notify_all();
  return ncclSuccess;
}

}  // extern "C"
