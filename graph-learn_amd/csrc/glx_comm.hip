// glx shard communicator: the transport under the distributed store (glx_dist.hip).
// Replaces the RPC layer the reference's DistributeRunner fans out over
// (graphlearn/src/core/runner/op_runner.h:86-152 RunInParallel -> one gRPC call per remote
// shard; service/client_impl.cc, rpc/) with RCCL point-to-point groups over xGMI:
// an all-to-all(v) is ncclGroupStart + P x (ncclSend, ncclRecv) + ncclGroupEnd on the
// caller's stream (SURVEY.md 8(e) "Collective API"), large exchanges are cut into rounds by
// pointer offsets (no staging copies), the self part is a device-to-device copy.
//
// Two more transports share the interface so that the same store code runs where RCCL
// cannot: ranks that are threads of one process (one-GPU test rig, single-process
// multi-GPU), and a host-staged one that hands pinned buffers to caller-supplied
// collectives (gloo / MPI).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "glx_comm.h"

namespace {

// ------------------------------------------------------------------ librccl --
// Loaded lazily with dlopen: libglx.so then loads (and everything single-GPU works) on a
// box without RCCL, and inside a torch process the already-loaded librccl.so.1 is reused.
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // GLX_RCCL_LIBRARY names another build of the library (the tests load an in-process stand-in through it)
    const char* names[] = {getenv("GLX_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (n == nullptr || *n == 0) continue;
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) return;
#define GLX_NCCL_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, #sym))
    GLX_NCCL_SYM(GetUniqueId, ncclGetUniqueId);
    GLX_NCCL_SYM(CommInitRank, ncclCommInitRank);
    GLX_NCCL_SYM(CommDestroy, ncclCommDestroy);
    GLX_NCCL_SYM(GroupStart, ncclGroupStart);
    GLX_NCCL_SYM(GroupEnd, ncclGroupEnd);
    GLX_NCCL_SYM(Send, ncclSend);
    GLX_NCCL_SYM(Recv, ncclRecv);
    GLX_NCCL_SYM(AllGather, ncclAllGather);
    GLX_NCCL_SYM(GetErrorString, ncclGetErrorString);
#undef GLX_NCCL_SYM
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd &&
             api.Send && api.Recv && api.AllGather && api.GetErrorString;
  });
  return &api;
}

#define GLX_NCCL(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t r__ = (expr);                                                                  \
    if (r__ != ncclSuccess) {                                                                   \
      glx_set_error("%s failed: %s (%s:%d)", #expr, rccl_api()->GetErrorString(r__), __FILE__, \
                    __LINE__);                                                                  \
      return GLX_INTERNAL;                                                                      \
    }                                                                                           \
  } while (0)

// Small pinned + device staging areas every transport needs for the count exchange.
struct CountStage {
  int64_t* h_pin = nullptr;
  int64_t* d_buf = nullptr;
  size_t cap = 0;  // int64 entries
  int ensure(size_t n) {
    if (n <= cap) return GLX_OK;
    release();
    size_t want = n < 1024 ? 1024 : n;
    GLX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_pin), want * 8, hipHostMallocDefault));
    GLX_HIP(hipMalloc(reinterpret_cast<void**>(&d_buf), want * 8));
    cap = want;
    return GLX_OK;
  }
  void release() {
    if (h_pin) (void)hipHostFree(h_pin);
    if (d_buf) (void)hipFree(d_buf);
    h_pin = nullptr;
    d_buf = nullptr;
    cap = 0;
  }
};

// ------------------------------------------------------------------- RCCL ----
struct RcclComm : glx_comm {
  ncclComm_t comm = nullptr;
  CountStage stage;
  ~RcclComm() override {
    GlxDeviceGuard guard(device);
    if (comm) (void)rccl_api()->CommDestroy(comm);
    stage.release();
  }

  int alltoallv(const GlxSeg* segs, int nseg, const int64_t* send_counts, const int64_t* send_offs,
                const int64_t* recv_counts, const int64_t* recv_offs, hipStream_t s) override {
    RcclApi* api = rccl_api();
    // self part: a plain device copy
    for (int j = 0; j < nseg; ++j) {
      const size_t eb = segs[j].elem_bytes;
      const size_t bytes = (size_t)send_counts[rank] * eb;
      if (bytes == 0) continue;
      GLX_REQUIRE(send_counts[rank] == recv_counts[rank], "self message size mismatch");
      GLX_HIP(hipMemcpyAsync(static_cast<char*>(segs[j].recv) + (size_t)recv_offs[rank] * eb,
                             static_cast<const char*>(segs[j].send) + (size_t)send_offs[rank] * eb, bytes,
                             hipMemcpyDeviceToDevice, s));
    }
    // rounds: message (p -> q) of c elements of eb bytes goes out in pieces of
    // step = max_message_bytes / eb elements; both ends derive the same pieces from c.
    int64_t rounds = 0;
    for (int j = 0; j < nseg; ++j) {
      const int64_t step = max_message_bytes / (int64_t)segs[j].elem_bytes > 0
                               ? max_message_bytes / (int64_t)segs[j].elem_bytes : 1;
      for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        const int64_t a = (send_counts[p] + step - 1) / step, b = (recv_counts[p] + step - 1) / step;
        if (a > rounds) rounds = a;
        if (b > rounds) rounds = b;
      }
    }
    last_rounds = rounds;
    for (int64_t r = 0; r < rounds; ++r) {
      GLX_NCCL(api->GroupStart());
      for (int j = 0; j < nseg; ++j) {
        const size_t eb = segs[j].elem_bytes;
        const int64_t step = max_message_bytes / (int64_t)eb > 0 ? max_message_bytes / (int64_t)eb : 1;
        for (int p = 0; p < world; ++p) {
          if (p == rank) continue;
          const int64_t lo = r * step;
          if (lo < send_counts[p]) {
            const int64_t n = send_counts[p] - lo < step ? send_counts[p] - lo : step;
            GLX_NCCL(api->Send(static_cast<const char*>(segs[j].send) + (size_t)(send_offs[p] + lo) * eb,
                               (size_t)n * eb, ncclInt8, p, comm, s));
          }
          if (lo < recv_counts[p]) {
            const int64_t n = recv_counts[p] - lo < step ? recv_counts[p] - lo : step;
            GLX_NCCL(api->Recv(static_cast<char*>(segs[j].recv) + (size_t)(recv_offs[p] + lo) * eb,
                               (size_t)n * eb, ncclInt8, p, comm, s));
          }
        }
      }
      GLX_NCCL(api->GroupEnd());
    }
    return GLX_OK;
  }

  int allgather_i64(const int64_t* d_vals, int nvals, int64_t* h_out, hipStream_t s) override {
    int rc = stage.ensure((size_t)world * nvals);
    if (rc != GLX_OK) return rc;
    if (world == 1) {
      GLX_HIP(hipMemcpyAsync(stage.h_pin, d_vals, (size_t)nvals * 8, hipMemcpyDeviceToHost, s));
    } else {
      GLX_NCCL(rccl_api()->AllGather(d_vals, stage.d_buf, (size_t)nvals, ncclInt64, comm, s));
      GLX_HIP(hipMemcpyAsync(stage.h_pin, stage.d_buf, (size_t)world * nvals * 8, hipMemcpyDeviceToHost, s));
    }
    GLX_HIP(hipStreamSynchronize(s));
    memcpy(h_out, stage.h_pin, (size_t)world * nvals * 8);
    return GLX_OK;
  }

  int barrier(hipStream_t s) override {
    int rc = stage.ensure((size_t)world + 1);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemsetAsync(stage.d_buf + world, 0, 8, s));
    std::vector<int64_t> sink((size_t)world);
    int64_t* d_one = stage.d_buf + world;
    if (world == 1) {
      GLX_HIP(hipStreamSynchronize(s));
      return GLX_OK;
    }
    GLX_NCCL(rccl_api()->AllGather(d_one, stage.d_buf, 1, ncclInt64, comm, s));
    GLX_HIP(hipStreamSynchronize(s));
    return GLX_OK;
  }
};

// ------------------------------------------------------- in-process fabric ---
struct LocalPost {
  const GlxSeg* segs;
  int nseg;
  const int64_t* send_counts;
  const int64_t* send_offs;
  const int64_t* h_vals;
  hipEvent_t ready;  // alltoallv: the poster's send buffers are complete when this fires
};

struct LocalFabric {
  int64_t key = 0;
  int world = 0;
  int refs = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  bool broken = false;
  LocalPost post[64];
  hipEvent_t done[64];  // alltoallv: rank q has read what it needed from its peers' send buffers when done[q] fires
  std::vector<std::pair<int, hipEvent_t>> retired;  // (device, event) of ranks that have left; under g_fabric_mtx

  // Generation barrier with a deadline: a rank that failed before the collective must not
  // hang its peers (and the GPU box) forever.
  int barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return fail();
    const uint64_t g = gen;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return GLX_OK;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(deadline_s()), [&] { return gen != g || broken; }) || broken) {
      broken = true;
      cv.notify_all();
      return fail();
    }
    return GLX_OK;
  }
  // 120 s unless GLX_LOCAL_COMM_TIMEOUT_S says otherwise (read once; test rigs that expect failures shorten it)
  static int deadline_s() {
    static const int s = [] {
      const char* e = getenv("GLX_LOCAL_COMM_TIMEOUT_S");
      const int v = e ? atoi(e) : 0;
      return v > 0 ? v : 120;
    }();
    return s;
  }
  static int fail() {
    glx_set_error("local communicator: a peer rank did not reach the collective within %d s", deadline_s());
    return GLX_UNAVAILABLE;
  }
};

std::mutex g_fabric_mtx;
std::map<int64_t, LocalFabric*> g_fabrics;

struct LocalComm : glx_comm {
  LocalFabric* fab = nullptr;
  CountStage stage;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  ~LocalComm() override {
    {
      GlxDeviceGuard guard(device);
      (void)hipDeviceSynchronize();  // peers' streams may still wait on this rank's events
      stage.release();
    }
    // This rank's events outlive it: a slower peer may not have ENQUEUED its wait on this rank's `done` event yet when
    // this rank has already left its last collective and is being torn down (the two host meetings of an exchange are
    // behind both, the peer's hipStreamWaitEvent loop is not) -- destroying the event here was a use-after-free in the
    // peer (round 6: one segmentation fault in a rank's last collective in ~84,000 fuzz cases, its peers already gone).
    // The fabric destroys them when its last rank has gone.
    std::lock_guard<std::mutex> g(g_fabric_mtx);
    if (fab) {
      if (ev_ready) fab->retired.emplace_back(device, ev_ready);
      if (ev_done) fab->retired.emplace_back(device, ev_done);
      if (--fab->refs == 0) {
        for (auto& e : fab->retired) {
          GlxDeviceGuard guard(e.first);
          (void)hipEventDestroy(e.second);
        }
        g_fabrics.erase(fab->key);
        delete fab;
      }
    }
  }

  // The ranks meet on the HOST twice (to see each other's posts, and to know every reader has enqueued its copies);
  // the streams never drain: a rank's copies wait for its peers' `ready` events, and its stream waits for every
  // peer's `done` event before anything later may touch the send buffers again -- the device-side order of a real
  // transport, without link time.
  int alltoallv(const GlxSeg* segs, int nseg, const int64_t* send_counts, const int64_t* send_offs,
                const int64_t* recv_counts, const int64_t* recv_offs, hipStream_t s) override {
    last_rounds = 1;
    if (!ev_ready) GLX_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
    if (!ev_done) GLX_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    GLX_HIP(hipEventRecord(ev_ready, s));
    fab->post[rank] = LocalPost{segs, nseg, send_counts, send_offs, nullptr, ev_ready};
    int rc = fab->barrier();
    if (rc != GLX_OK) return rc;
    hipError_t e = hipSuccess;
    bool mismatch = false;
    for (int q = 0; q < world && e == hipSuccess; ++q) {
      const LocalPost& from = fab->post[q];
      if (from.nseg != nseg || from.send_counts[rank] != recv_counts[q]) {
        mismatch = true;
        break;
      }
      bool waited = q == rank;
      for (int j = 0; j < nseg && e == hipSuccess; ++j) {
        const size_t eb = segs[j].elem_bytes;
        const size_t bytes = (size_t)recv_counts[q] * eb;
        if (bytes == 0) continue;
        if (!waited) {
          e = hipStreamWaitEvent(s, from.ready, 0);
          waited = true;
          if (e != hipSuccess) break;
        }
        e = hipMemcpyAsync(static_cast<char*>(segs[j].recv) + (size_t)recv_offs[q] * eb,
                           static_cast<const char*>(from.segs[j].send) + (size_t)from.send_offs[rank] * eb, bytes,
                           hipMemcpyDefault, s);
      }
    }
    if (e == hipSuccess) e = hipEventRecord(ev_done, s);
    fab->done[rank] = ev_done;
    rc = fab->barrier();  // every reader has enqueued its copies (the posts' host arrays may go now)
    GLX_REQUIRE(!mismatch, "local communicator: send / receive counts of two ranks disagree");
    GLX_HIP(e);
    if (rc != GLX_OK) return rc;
    for (int q = 0; q < world; ++q) {
      if (q != rank) GLX_HIP(hipStreamWaitEvent(s, fab->done[q], 0));
    }
    // (the events are re-recorded by this rank's NEXT call only after that call's first meeting, which no peer reaches
    // before it has enqueued the waits above)
    return GLX_OK;
  }

  int allgather_i64(const int64_t* d_vals, int nvals, int64_t* h_out, hipStream_t s) override {
    int rc = stage.ensure((size_t)nvals);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemcpyAsync(stage.h_pin, d_vals, (size_t)nvals * 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    fab->post[rank] = LocalPost{nullptr, nvals, nullptr, nullptr, stage.h_pin};
    rc = fab->barrier();
    if (rc != GLX_OK) return rc;
    bool mismatch = false;
    for (int q = 0; q < world; ++q) {
      if (fab->post[q].nseg != nvals) {
        mismatch = true;
        break;
      }
      memcpy(h_out + (size_t)q * nvals, fab->post[q].h_vals, (size_t)nvals * 8);
    }
    rc = fab->barrier();
    GLX_REQUIRE(!mismatch, "local communicator: ranks gathered different value counts");
    return rc;
  }

  int barrier(hipStream_t s) override {
    GLX_HIP(hipStreamSynchronize(s));
    return fab->barrier();
  }
};

// ------------------------------------------------------------ host-staged ----
struct HostBuf {
  char* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return GLX_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    GLX_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault));
    cap = want;
    return GLX_OK;
  }
  ~HostBuf() {
    if (p) (void)hipHostFree(p);
  }
};

struct CallbackComm : glx_comm {
  glx_host_alltoallv_fn a2a = nullptr;
  glx_host_allgather_fn gather = nullptr;
  void* user = nullptr;
  HostBuf hs, hr, hv;

  int alltoallv(const GlxSeg* segs, int nseg, const int64_t* send_counts, const int64_t* send_offs,
                const int64_t* recv_counts, const int64_t* recv_offs, hipStream_t s) override {
    last_rounds = 1;
    int64_t ns = 0, nr = 0;
    for (int p = 0; p < world; ++p) {
      ns += send_counts[p];
      nr += recv_counts[p];
    }
    for (int j = 0; j < nseg; ++j) {
      const size_t eb = segs[j].elem_bytes;
      int rc = hs.ensure((size_t)ns * eb);
      if (rc == GLX_OK) rc = hr.ensure((size_t)nr * eb);
      if (rc != GLX_OK) return rc;
      size_t at = 0;
      for (int p = 0; p < world; ++p) {
        const size_t bytes = (size_t)send_counts[p] * eb;
        if (bytes) {
          GLX_HIP(hipMemcpyAsync(hs.p + at, static_cast<const char*>(segs[j].send) + (size_t)send_offs[p] * eb,
                                 bytes, hipMemcpyDeviceToHost, s));
        }
        at += bytes;
      }
      GLX_HIP(hipStreamSynchronize(s));
      const int crc = a2a(user, hs.p, send_counts, hr.p, recv_counts, (int64_t)eb);
      if (crc != 0) {
        glx_set_error("host all-to-all callback failed with code %d", crc);
        return GLX_INTERNAL;
      }
      at = 0;
      for (int q = 0; q < world; ++q) {
        const size_t bytes = (size_t)recv_counts[q] * eb;
        if (bytes) {
          GLX_HIP(hipMemcpyAsync(static_cast<char*>(segs[j].recv) + (size_t)recv_offs[q] * eb, hr.p + at, bytes,
                                 hipMemcpyHostToDevice, s));
        }
        at += bytes;
      }
      GLX_HIP(hipStreamSynchronize(s));  // hr is reused by the next segment
    }
    return GLX_OK;
  }

  int allgather_i64(const int64_t* d_vals, int nvals, int64_t* h_out, hipStream_t s) override {
    int rc = hv.ensure((size_t)nvals * 8);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemcpyAsync(hv.p, d_vals, (size_t)nvals * 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    const int crc = gather(user, hv.p, h_out, (int64_t)nvals * 8);
    if (crc != 0) {
      glx_set_error("host all-gather callback failed with code %d", crc);
      return GLX_INTERNAL;
    }
    return GLX_OK;
  }

  int barrier(hipStream_t s) override {
    GLX_HIP(hipStreamSynchronize(s));
    int64_t one = 0;
    std::vector<int64_t> sink((size_t)world);
    const int crc = gather(user, &one, sink.data(), 8);
    if (crc != 0) {
      glx_set_error("host all-gather callback failed with code %d", crc);
      return GLX_INTERNAL;
    }
    return GLX_OK;
  }
};

int check_rank(int device, int rank, int world, glx_comm** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(world >= 1 && world <= 64, "world size must be in [1, 64]");
  GLX_REQUIRE(rank >= 0 && rank < world, "rank %d outside [0, %d)", rank, world);
  return glx_init_device(device);
}

}  // namespace

extern "C" int glx_comm_unique_id(void* id_out) {
  GLX_REQUIRE(id_out != nullptr, "id_out is NULL");
  static_assert(sizeof(ncclUniqueId) == GLX_UNIQUE_ID_BYTES, "ncclUniqueId size");
  RcclApi* api = rccl_api();
  if (!api->ok) {
    glx_set_error("librccl.so.1 could not be loaded (%s)", api->handle ? "missing symbols" : dlerror());
    return GLX_UNAVAILABLE;
  }
  ncclUniqueId id;
  GLX_NCCL(api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return GLX_OK;
}

extern "C" int glx_comm_init_rccl(int device, int rank, int world, const void* unique_id, glx_comm** out) {
  int rc = check_rank(device, rank, world, out);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(unique_id != nullptr, "unique_id is NULL");
  RcclApi* api = rccl_api();
  if (!api->ok) {
    glx_set_error("librccl.so.1 could not be loaded (%s)", api->handle ? "missing symbols" : dlerror());
    return GLX_UNAVAILABLE;
  }
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  RcclComm* c = new (std::nothrow) RcclComm();
  GLX_REQUIRE(c != nullptr, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->kind = GLX_COMM_RCCL;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    glx_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
    c->comm = nullptr;
    delete c;
    return GLX_UNAVAILABLE;
  }
  *out = c;
  return GLX_OK;
}

extern "C" int glx_comm_init_local(int64_t fabric_key, int device, int rank, int world, glx_comm** out) {
  int rc = check_rank(device, rank, world, out);
  if (rc != GLX_OK) return rc;
  LocalComm* c = new (std::nothrow) LocalComm();
  GLX_REQUIRE(c != nullptr, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->kind = GLX_COMM_LOCAL;
  bool mismatch = false;
  {
    std::lock_guard<std::mutex> g(g_fabric_mtx);
    auto it = g_fabrics.find(fabric_key);
    if (it == g_fabrics.end()) {
      LocalFabric* f = new LocalFabric();
      f->key = fabric_key;
      f->world = world;
      it = g_fabrics.emplace(fabric_key, f).first;
    }
    LocalFabric* f = it->second;
    if (f->world != world) {
      glx_set_error("fabric %lld was created with world size %d, not %d", (long long)fabric_key, f->world, world);
      mismatch = true;
    } else {
      ++f->refs;
      c->fab = f;
    }
  }
  if (mismatch) {
    // outside the lock: ~LocalComm takes g_fabric_mtx itself (it is not recursive); c->fab is still null, so the
    // fabric the OTHER ranks share is left alone
    delete c;
    return GLX_INVALID_ARGUMENT;
  }
  *out = c;
  return GLX_OK;
}

extern "C" int glx_comm_init_callbacks(int device, int rank, int world, glx_host_alltoallv_fn alltoallv,
                                       glx_host_allgather_fn allgather, void* user, glx_comm** out) {
  int rc = check_rank(device, rank, world, out);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(alltoallv != nullptr && allgather != nullptr, "NULL callback");
  CallbackComm* c = new (std::nothrow) CallbackComm();
  GLX_REQUIRE(c != nullptr, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->kind = GLX_COMM_CALLBACKS;
  c->a2a = alltoallv;
  c->gather = allgather;
  c->user = user;
  *out = c;
  return GLX_OK;
}

extern "C" void glx_comm_destroy(glx_comm* c) { delete c; }

extern "C" int glx_comm_info(const glx_comm* c, int* rank, int* world, int* device, int* transport) {
  GLX_REQUIRE(c != nullptr, "comm is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (device) *device = c->device;
  if (transport) *transport = c->kind;
  return GLX_OK;
}

extern "C" int64_t glx_comm_set_max_message_bytes(glx_comm* c, int64_t bytes) {
  if (!c) return 0;
  const int64_t prev = c->max_message_bytes;
  if (bytes > 0) c->max_message_bytes = bytes;
  return prev;
}

extern "C" int glx_exchange_v(glx_comm* c, const void* send, const int64_t* send_counts, void* recv,
                              const int64_t* recv_counts, int64_t elem_bytes, int ptr_kind, void* stream) {
  GLX_REQUIRE(c != nullptr, "comm is NULL");
  GLX_REQUIRE(send_counts && recv_counts, "NULL counts");
  GLX_REQUIRE(elem_bytes > 0, "elem_bytes must be positive");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GlxDeviceGuard guard(c->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", c->device);
  std::vector<int64_t> so((size_t)c->world + 1, 0), ro((size_t)c->world + 1, 0);
  for (int p = 0; p < c->world; ++p) {
    GLX_REQUIRE(send_counts[p] >= 0 && recv_counts[p] >= 0, "negative count");
    so[p + 1] = so[p] + send_counts[p];
    ro[p + 1] = ro[p] + recv_counts[p];
  }
  GLX_REQUIRE((so[c->world] == 0 || send) && (ro[c->world] == 0 || recv), "NULL data pointer");
  if (ptr_kind == GLX_PTR_DEVICE) {
    GlxSeg seg{send, recv, (size_t)elem_bytes};
    return c->alltoallv(&seg, 1, send_counts, so.data(), recv_counts, ro.data(), glx_stream(stream));
  }
  hipStream_t s = glx_host_call_stream(stream, c->device);
  const size_t sb = (size_t)so[c->world] * elem_bytes, rb = (size_t)ro[c->world] * elem_bytes;
  GlxTemp ds, dr;
  GLX_HIP(hipMalloc(&ds.p, sb ? sb : 256));
  GLX_HIP(hipMalloc(&dr.p, rb ? rb : 256));
  if (sb) GLX_HIP(hipMemcpyAsync(ds.p, send, sb, hipMemcpyHostToDevice, s));
  GlxSeg seg{ds.p, dr.p, (size_t)elem_bytes};
  int rc = c->alltoallv(&seg, 1, send_counts, so.data(), recv_counts, ro.data(), s);
  hipError_t e = hipSuccess;
  if (rc == GLX_OK && rb) e = hipMemcpyAsync(recv, dr.p, rb, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipStreamSynchronize(s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(e);
  GLX_HIP(e2);
  return GLX_OK;
}

extern "C" int glx_comm_allgather_i64(glx_comm* c, const int64_t* vals, int32_t nvals, int64_t* out, int ptr_kind,
                                      void* stream) {
  GLX_REQUIRE(c != nullptr, "comm is NULL");
  GLX_REQUIRE(vals && out && nvals > 0, "bad arguments");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GlxDeviceGuard guard(c->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", c->device);
  if (ptr_kind == GLX_PTR_DEVICE) {
    hipStream_t s = glx_stream(stream);
    std::vector<int64_t> h((size_t)c->world * nvals);
    int rc = c->allgather_i64(vals, nvals, h.data(), s);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemcpyAsync(out, h.data(), h.size() * 8, hipMemcpyHostToDevice, s));
    GLX_HIP(hipStreamSynchronize(s));
    return GLX_OK;
  }
  hipStream_t s = glx_host_call_stream(stream, c->device);
  GlxTemp d;
  GLX_HIP(hipMalloc(&d.p, (size_t)nvals * 8));
  GLX_HIP(hipMemcpyAsync(d.p, vals, (size_t)nvals * 8, hipMemcpyHostToDevice, s));
  return c->allgather_i64(d.as<int64_t>(), nvals, out, s);
}

extern "C" int glx_comm_barrier(glx_comm* c, void* stream) {
  GLX_REQUIRE(c != nullptr, "comm is NULL");
  GlxDeviceGuard guard(c->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", c->device);
  return c->barrier(glx_stream(stream));
}
