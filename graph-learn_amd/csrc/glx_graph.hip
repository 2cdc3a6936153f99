// glx graph storage: device CSR (64-bit row_ptr, interleaved {nbr, eid} slots),
// per-row alias tables built once on the device, optional id->row hash map.
// Replaces graphlearn/src/core/graph/storage/{memory_adj_matrix.cc:159-225,
// auto_indexing.cc:21-33} + the per-request AliasMethod::Build of
// edge_weight_sampler.cc:78-92 (alias_method.cc:57-107).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <vector>

#include <map>

#include "glx_common.h"

// ------------------------------------------------------------------ errors --
static thread_local char g_err[512] = "";

void glx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* glx_last_error(void) { return g_err; }
extern "C" int glx_abi_version(void) { return GLX_ABI_VERSION; }

GlxSideKnobs& glx_side_knobs() {
  static GlxSideKnobs k;
  static std::once_flag once;
  std::call_once(once, [] {
    if (getenv("GLX_COND_SEQUENTIAL")) k.cond_sequential = 1;
    if (getenv("GLX_DIST_NO_BITMAP")) k.dist_no_bitmap = 1;
    if (const char* e = getenv("GLX_FILTER_SPAN_CAP")) k.filter_span_cap = atoll(e);
    if (const char* e = getenv("GLX_FILTER_DEDUP_MIN_ROWS")) k.filter_dedup_min_rows = atoll(e);
    if (getenv("GLX_IDMAP_HASH_ONLY")) k.idmap_hash_only = 1;
    if (const char* e = getenv("GLX_RESOLVE_IDS")) k.resolve_ids = atoll(e);
    if (const char* e = getenv("GLX_RESOLVE_BLOCKS")) k.resolve_blocks = atoll(e);
    if (const char* e = getenv("GLX_RESOLVE_SET_SHARE")) k.resolve_set_share = atoll(e);
    if (const char* e = getenv("GLX_RESOLVE_PEEK")) k.resolve_peek = atoll(e);
    if (const char* e = getenv("GLX_RESOLVE_OWN_FIRST")) k.resolve_own_first = atoll(e);
  });
  return k;
}

extern "C" int glx_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    if (count) *count = 0;
    glx_set_error("no usable HIP device (%s); glx has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return GLX_UNAVAILABLE;
  }
  if (count) *count = n;
  return GLX_OK;
}

namespace {
// The ranges pinned through glx_host_register.  Host-pointer calls write straight into a caller buffer only when it
// lies in one of THESE (glx_mapped_ptr): what the runtime itself reports about an arbitrary pointer also covers
// ranges other code registered -- and, after unregistrations of ranges that shared pages, was seen to be stale.
std::mutex g_pinned_mtx;
std::map<uintptr_t, size_t> g_pinned;
}  // namespace

extern "C" int glx_host_register(void* p, uint64_t bytes) {
  GLX_REQUIRE(p != nullptr && bytes > 0, "bad buffer");
  int n = 0;
  int rc = glx_device_count(&n);
  if (rc != GLX_OK) return rc;
  // Ranges of the malloc heap are refused (round 6; GLX_HOST_REGISTER_HEAP=1 lifts it): registered heap ranges beside heap
  // memory marked MADV_HUGEPAGE make later pageable copies of the process fault on ROCm 7.0 (include/glx.h,
  // scripts/r06/repro/hostreg_pageable.hip).  Only the brk heap can be told apart here; the rule is the header's.
  {
    static const bool allow_heap = [] {
      const char* e = getenv("GLX_HOST_REGISTER_HEAP");
      return e && atoi(e) != 0;
    }();
    uintptr_t lo = 0, hi = 0;
    if (!allow_heap) {
      if (FILE* f = fopen("/proc/self/maps", "r")) {
        char line[512];
        while (fgets(line, sizeof(line), f)) {
          unsigned long a = 0, b = 0;
          if (strstr(line, "[heap]") && sscanf(line, "%lx-%lx", &a, &b) == 2) {
            lo = a;
            hi = b;
            break;
          }
        }
        fclose(f);
      }
    }
    const uintptr_t at = reinterpret_cast<uintptr_t>(p);
    GLX_REQUIRE(!(at < hi && at + bytes > lo),
                "glx_host_register: the range lies in the malloc heap; register private anonymous mappings of their own "
                "(mmap, 2 MiB aligned) instead -- see include/glx.h (GLX_HOST_REGISTER_HEAP=1 overrides)");
  }
  hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    glx_set_error("hipHostRegister(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? GLX_RESOURCE_EXHAUSTED : GLX_UNAVAILABLE;
  }
  std::lock_guard<std::mutex> g(g_pinned_mtx);
  g_pinned[reinterpret_cast<uintptr_t>(p)] = (size_t)bytes;
  return GLX_OK;
}

extern "C" int glx_host_unregister(void* p) {
  GLX_REQUIRE(p != nullptr, "bad buffer");
  {
    std::lock_guard<std::mutex> g(g_pinned_mtx);
    g_pinned.erase(reinterpret_cast<uintptr_t>(p));
  }
  hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    glx_set_error("hipHostUnregister failed: %s", hipGetErrorString(e));
    return GLX_UNAVAILABLE;
  }
  return GLX_OK;
}

namespace {
struct HostGate {
  std::mutex m;
  std::condition_variable cv;
  int in_flight = 0;
};
HostGate g_host_gate[64];
int host_call_limit() {
  static const int limit = [] {
    const char* e = getenv("GLX_HOST_CALL_CONCURRENCY");
    return e ? atoi(e) : 12;
  }();
  return limit;
}
}  // namespace

GlxHostCallSlot::GlxHostCallSlot(int dev) : device(dev) {
  const int limit = host_call_limit();
  if (limit <= 0 || device < 0 || device >= 64) {
    device = -1;
    return;
  }
  HostGate& g = g_host_gate[device];
  std::unique_lock<std::mutex> lk(g.m);
  g.cv.wait(lk, [&] { return g.in_flight < limit; });
  ++g.in_flight;
}

GlxHostCallSlot::~GlxHostCallSlot() {
  if (device < 0) return;
  HostGate& g = g_host_gate[device];
  {
    std::lock_guard<std::mutex> lk(g.m);
    --g.in_flight;
  }
  g.cv.notify_one();
}

void* glx_mapped_ptr(const void* host_ptr, size_t bytes) {
  static const bool off = [] {
    const char* e = getenv("GLX_HOST_ZERO_COPY");  // "0": always stage through a copy (A/B knob)
    return e && atoi(e) == 0;
  }();
  if (off || host_ptr == nullptr) return nullptr;
  {
    const uintptr_t a = reinterpret_cast<uintptr_t>(host_ptr);
    std::lock_guard<std::mutex> g(g_pinned_mtx);
    auto it = g_pinned.upper_bound(a);
    if (it == g_pinned.begin()) return nullptr;
    --it;
    if (a >= it->first + it->second || bytes > it->first + it->second - a) return nullptr;  // not (wholly) in a range glx_host_register pinned
  }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, const_cast<void*>(host_ptr), 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return dev;
}

int glx_init_device(int device) {
  static std::mutex mtx;
  static bool done[64] = {false};
  std::lock_guard<std::mutex> g(mtx);
  if (device < 0 || device >= 64) {
    glx_set_error("bad device index %d", device);
    return GLX_INVALID_ARGUMENT;
  }
  if (done[device]) return GLX_OK;
  int n = 0;
  int rc = glx_device_count(&n);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(device < n, "device %d out of range (%d visible)", device, n);
  done[device] = true;
  return GLX_OK;
}

// Per-(thread, device, stream, slot) grow-only workspaces.  Work on one stream
// is ordered, so consecutive calls on a stream may reuse the same buffer; calls
// on different streams or from different host threads never share one (the
// reference calls Process() concurrently from its thread pool).  This avoids
// the stream-ordered allocator entirely: hipMallocAsync on the null stream
// handed out blocks whose pageable H2D copies raced with the first kernel on
// the ROCm 7.2 runtime (observed as all-zero inputs on the first call after a
// pool reuse).  Growth frees with hipFree, which synchronises the device, so a
// buffer is never released under a running kernel.
namespace {
struct WsKey {
  int device;
  hipStream_t stream;
  int slot;
  bool operator==(const WsKey& o) const {
    return device == o.device && stream == o.stream && slot == o.slot;
  }
};
struct WsBuf {
  WsKey key;
  void* p;
  size_t cap;
};
struct WsCache {
  std::vector<WsBuf> bufs;
  ~WsCache() {
    // Best effort: at process exit the runtime may already be gone.
    for (auto& b : bufs) {
      if (b.p && hipSetDevice(b.key.device) == hipSuccess) (void)hipFree(b.p);
    }
    (void)hipGetLastError();
  }
};
thread_local WsCache g_ws;
}  // namespace

namespace {
struct ScratchPlan {
  int mode = GLX_SCRATCH_NORMAL;
  std::vector<size_t> sizes;  // aligned sizes of the dry run, in call order
  size_t cursor = 0, offset = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
};
thread_local ScratchPlan g_scratch_plan;
thread_local const uint64_t* g_cc_dev = nullptr;
}  // namespace

void glx_scratch_mode(int mode, char* arena, size_t arena_bytes) {
  ScratchPlan& sp = g_scratch_plan;
  sp.mode = mode;
  sp.cursor = 0;
  sp.offset = 0;
  sp.arena = arena;
  sp.arena_bytes = arena_bytes;
  if (mode == GLX_SCRATCH_RECORD) sp.sizes.clear();
}

size_t glx_scratch_recorded_bytes() {
  size_t total = 0;
  for (size_t b : g_scratch_plan.sizes) total += b;
  return total;
}

void glx_capture_set_cc_dev(const uint64_t* p) { g_cc_dev = p; }
const uint64_t* glx_capture_cc_dev() { return g_cc_dev; }

int glx_scratch_alloc(void** p, size_t bytes, hipStream_t s, int slot) {
  int dev = 0;
  GLX_HIP(hipGetDevice(&dev));
  if (bytes == 0) bytes = 256;
  ScratchPlan& sp = g_scratch_plan;
  if (sp.mode == GLX_SCRATCH_REPLAY) {
    // the same call sequence as the recorded dry run, served out of the plan's own arena
    const size_t want = (bytes + 255) & ~(size_t)255;
    GLX_REQUIRE(sp.cursor < sp.sizes.size() && want <= sp.sizes[sp.cursor] && sp.offset + sp.sizes[sp.cursor] <= sp.arena_bytes,
                "request plan: workspace request %zu does not match the recorded dry run", sp.cursor);
    *p = sp.arena + sp.offset;
    sp.offset += sp.sizes[sp.cursor++];
    return GLX_OK;
  }
  if (sp.mode == GLX_SCRATCH_RECORD) sp.sizes.push_back((bytes + 255) & ~(size_t)255);
  WsKey key{dev, s, slot};
  WsBuf* hit = nullptr;
  for (auto& b : g_ws.bufs) {
    if (b.key == key) {
      hit = &b;
      break;
    }
  }
  if (!hit) {
    g_ws.bufs.push_back(WsBuf{key, nullptr, 0});
    hit = &g_ws.bufs.back();
  }
  if (hit->cap < bytes) {
    if (hit->p) GLX_HIP(hipFree(hit->p));
    hit->p = nullptr;
    hit->cap = 0;
    size_t want = bytes + bytes / 4;
    if (want < (1u << 20)) want = 1u << 20;
    want = (want + 255) & ~(size_t)255;
    GLX_HIP(hipMalloc(&hit->p, want));
    hit->cap = want;
  }
  *p = hit->p;
  return GLX_OK;
}

// Gives a slot's cached workspace back to the device when it has grown beyond `keep_bytes`
// (a rare huge request must not pin tens of GB per calling thread).  Synchronises `s` first.
void glx_scratch_trim(hipStream_t s, int slot, size_t keep_bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  WsKey key{dev, s, slot};
  for (auto& b : g_ws.bufs) {
    if (b.key == key && b.cap > keep_bytes) {
      (void)hipStreamSynchronize(s);
      (void)hipFree(b.p);
      b.p = nullptr;
      b.cap = 0;
    }
  }
}

void glx_scratch_free(void* p, hipStream_t s) {
  (void)p;
  (void)s;  // workspaces are cached per thread/stream; nothing to release per call
}

// ------------------------------------------------------- per-thread streams --
namespace {
struct ThreadStreams {
  hipStream_t s[64] = {nullptr};
  ~ThreadStreams() {
    for (int d = 0; d < 64; ++d) {
      if (s[d] && hipSetDevice(d) == hipSuccess) (void)hipStreamDestroy(s[d]);
    }
    (void)hipGetLastError();
  }
};
thread_local ThreadStreams g_thread_streams;
}  // namespace

hipStream_t glx_thread_stream(int device) {
  if (device < 0 || device >= 64) return nullptr;
  hipStream_t& st = g_thread_streams.s[device];
  if (!st) {
    // the caller holds a GlxDeviceGuard for `device`
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      st = nullptr;  // fall back to the null stream
    }
  }
  return st;
}

// ---------------------------------------------------------------- profiling --
namespace {
struct TimedLaunch {
  int kind;
  hipEvent_t start, stop;
};
struct ProfileState {
  bool on = false;
  std::vector<TimedLaunch> launches;
  std::vector<hipEvent_t> free_events;
};
thread_local ProfileState g_prof;

hipEvent_t prof_event() {
  if (!g_prof.free_events.empty()) {
    hipEvent_t e = g_prof.free_events.back();
    g_prof.free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

GlxKernelTimer::GlxKernelTimer(int kind, hipStream_t stream) : s(stream) {
  if (!g_prof.on) return;
  if (g_prof.launches.size() >= (1u << 20)) return;  // never collected: stop recording
  TimedLaunch t{kind, prof_event(), prof_event()};
  if (!t.start || !t.stop) return;
  if (hipEventRecord(t.start, s) != hipSuccess) return;
  g_prof.launches.push_back(t);
  slot = (int)g_prof.launches.size() - 1;
}

void GlxKernelTimer::stop() {
  if (slot >= 0) (void)hipEventRecord(g_prof.launches[slot].stop, s);
}

bool glx_profile_suspend(bool suspend) {
  static thread_local bool saved = false;
  if (suspend) {
    saved = g_prof.on;
    g_prof.on = false;
    return saved;
  }
  g_prof.on = saved;
  return saved;
}

extern "C" int glx_profile_enable(int on) {
  g_prof.on = on != 0;
  return GLX_OK;
}

extern "C" int glx_profile_collect(int kind, float* ms_out, int32_t cap, int32_t* count) {
  GLX_REQUIRE(count != nullptr, "count is NULL");
  int32_t n = 0;
  std::vector<TimedLaunch> keep;
  for (auto& t : g_prof.launches) {
    if (t.kind != kind) {
      keep.push_back(t);
      continue;
    }
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(t.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t.start, t.stop);
    if (e == hipSuccess && ms_out && n < cap) ms_out[n] = ms;
    if (e == hipSuccess) ++n;
    g_prof.free_events.push_back(t.start);
    g_prof.free_events.push_back(t.stop);
  }
  g_prof.launches.swap(keep);
  (void)hipGetLastError();
  *count = n < cap || !ms_out ? n : cap;
  return GLX_OK;
}

// ------------------------------------------------------------------ id map --
__global__ void glx_idmap_fill_kernel(int64_t* keys, uint64_t cap) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) keys[i] = GLX_EMPTY_KEY;
}

// Ids are unique per storage (AutoIndex inserts once per new id); if a caller
// passes duplicates the smallest row wins (first-insertion order).
__global__ void glx_idmap_insert_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t* keys,
                                        int32_t* vals, uint64_t mask) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  int64_t id = ids[r];
  if (id == GLX_EMPTY_KEY) return;
  uint64_t h = glx_mix64((uint64_t)id) & mask;
  while (true) {
    unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long*>(&keys[h]),
                  (unsigned long long)GLX_EMPTY_KEY, (unsigned long long)id);
    if ((int64_t)prev == GLX_EMPTY_KEY) {
      atomicMin(&vals[h], (int32_t)r);
      return;
    }
    if ((int64_t)prev == id) {
      atomicMin(&vals[h], (int32_t)r);
      return;
    }
    h = (h + 1) & mask;
  }
}

__global__ void glx_fill_i32_kernel(int32_t* p, uint64_t n, int32_t v) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

int glx_idmap_build(const int64_t* d_ids, int64_t num_rows, GlxIdMapStorage* out, hipStream_t s) {
  uint64_t cap = 64;
  while (cap < (uint64_t)num_rows * 2) cap <<= 1;
  GLX_HIP(hipMalloc(&out->keys, cap * sizeof(int64_t)));
  GLX_HIP(hipMalloc(&out->vals, cap * sizeof(int32_t)));
  out->cap = cap;
  int blocks = (int)((cap + 255) / 256 < 4096 ? (cap + 255) / 256 : 4096);
  glx_idmap_fill_kernel<<<blocks, 256, 0, s>>>(out->keys, cap);
  glx_fill_i32_kernel<<<blocks, 256, 0, s>>>(out->vals, cap, INT32_MAX);
  if (num_rows > 0) {
    glx_idmap_insert_kernel<<<(unsigned)((num_rows + 255) / 256), 256, 0, s>>>(
        d_ids, num_rows, out->keys, out->vals, cap - 1);
  }
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// flag[0] |= 1 when ids is not ids[0] + i * (ids[1] - ids[0])
__global__ void glx_idmap_affine_kernel(const int64_t* __restrict__ ids, int64_t n, int* flag) {
  const int64_t base = ids[0], step = ids[1] - ids[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (ids[i] != base + i * step) {
      *flag = 1;
      return;
    }
  }
}

int glx_idmap_build_auto(const int64_t* d_ids, int64_t num_rows, GlxIdMapStorage* out, hipStream_t s) {
  if (num_rows >= 2 && glx_side_knobs().idmap_hash_only.load(std::memory_order_relaxed) <= 0) {
    int64_t first[2] = {0, 0};
    GLX_HIP(hipMemcpyAsync(first, d_ids, 16, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    const int64_t step = first[1] - first[0];
    // (the last id must not overflow: base + (n - 1) * step computed in 128 bits)
    const __int128 last = (__int128)first[0] + (__int128)(num_rows - 1) * (__int128)step;
    if (step > 0 && first[1] > first[0] && last <= (__int128)INT64_MAX) {
      GlxTemp flag;
      GLX_HIP(hipMalloc(&flag.p, sizeof(int)));
      GLX_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), s));
      glx_idmap_affine_kernel<<<(unsigned)((num_rows + 255) / 256 < 4096 ? (num_rows + 255) / 256 : 4096), 256, 0, s>>>(d_ids, num_rows, flag.as<int>());
      int h = 1;
      GLX_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
      GLX_HIP(hipStreamSynchronize(s));
      if (h == 0) {
        out->keys = nullptr;
        out->vals = nullptr;
        out->cap = 0;
        out->base = first[0];
        out->step = step;
        return GLX_OK;
      }
    }
  }
  return glx_idmap_build(d_ids, num_rows, out, s);
}

void glx_idmap_free(GlxIdMapStorage* m) {
  if (m->keys) (void)hipFree(m->keys);
  if (m->vals) (void)hipFree(m->vals);
  m->keys = nullptr;
  m->vals = nullptr;
  m->cap = 0;
  m->base = m->step = 0;
}

// ------------------------------------------------------------ CSR kernels --
// flag bits: 1 = row_ptr not monotone / bad ends; 2 = a row degree >= 2^31.
__global__ void glx_check_row_ptr_kernel(const int64_t* __restrict__ row_ptr, int64_t V, int64_t E,
                                         int* flag) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r == 0 && (row_ptr[0] != 0 || row_ptr[V] != E)) atomicOr(flag, 1);
  if (r < V && row_ptr[r + 1] < row_ptr[r]) atomicOr(flag, 1);
  if (r < V && row_ptr[r + 1] - row_ptr[r] >= (int64_t)INT32_MAX) atomicOr(flag, 2);
}

// SoA (col[], eid[]) -> 16-byte {nbr, eid} slots: one dwordx4 gather per draw.
__global__ void glx_pack_adj_kernel(const int64_t* __restrict__ col, const int64_t* __restrict__ eid,
                                    int64_t E, GlxAdj* __restrict__ adj) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < E; i += stride) adj[i] = GlxAdj{col[i], eid[i]};
}

// AliasMethod::Build (alias_method.cc:57-107), one lane per row, bit-identical
// to the serial reference: LIFO low/high stacks (kept in `stk` as {prob, index}
// pairs: low grows up from the row's first slot, high grows down from its last;
// |low|+|high| <= deg always), sum accumulated in double then narrowed to float
// (:73), float arithmetic without contraction (-ffp-contract=off).
__global__ void glx_alias_build_kernel(const int64_t* __restrict__ row_ptr,
                                       const float* __restrict__ weight, int64_t V,
                                       GlxAlias* __restrict__ out, GlxAlias* __restrict__ stk) {
  int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (row >= V) return;
  const int64_t s = row_ptr[row];
  const int32_t count = (int32_t)(row_ptr[row + 1] - s);
  if (count == 0 || count > kAliasLaneRowMax) return;  // longer rows: glx_alias_build_wave_kernel
  glx_alias_build_row_dev(weight + s, count, out + s, stk + s);
}

// Rows longer than kAliasLaneRowMax: one wave per row (glx_alias_build_row_wave).
__global__ __launch_bounds__(256) void glx_alias_build_wave_kernel(const int64_t* __restrict__ row_ptr,
                                                                   const float* __restrict__ weight, int64_t V,
                                                                   GlxAlias* __restrict__ out,
                                                                   GlxAlias* __restrict__ stk) {
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= V) return;
  const int64_t s = row_ptr[row];
  const int32_t count = (int32_t)(row_ptr[row + 1] - s);
  if (count <= kAliasLaneRowMax) return;
  glx_alias_build_row_wave(weight + s, count, out + s, stk + s);
}

// Packs {prob, (nbr, eid) of the slot, (nbr, eid) of its alias} per slot; *bad is set
// when an edge id does not fit an int32 (the records are then discarded).
__global__ void glx_pack_ew_kernel(const int64_t* __restrict__ row_ptr, const GlxAdj* __restrict__ adj,
                                   const GlxAlias* __restrict__ alias, int64_t V, GlxEwRec* __restrict__ out,
                                   int* bad) {
  // one wave per row: lanes stride over the row's slots
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= V) return;
  const int64_t s = row_ptr[row], e = row_ptr[row + 1];
  for (int64_t i = s + lane; i < e; i += 64) {
    const GlxAlias a = alias[i];
    const GlxAdj self = adj[i];
    const GlxAdj other = adj[s + a.alias];
    if (self.eid > INT32_MAX || self.eid < INT32_MIN || other.eid > INT32_MAX || other.eid < INT32_MIN) *bad = 1;
    out[i] = GlxEwRec{a.prob, (int32_t)self.eid, (int32_t)other.eid, 0, self.nbr, other.nbr};
  }
}

__global__ void glx_unpack_alias_kernel(const GlxAlias* __restrict__ tab, int64_t E,
                                        float* __restrict__ prob, int32_t* __restrict__ alias) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < E; i += stride) {
    GlxAlias a = tab[i];
    prob[i] = a.prob;
    alias[i] = a.alias;
  }
}

__global__ void glx_degrees_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr,
                                   const int64_t* __restrict__ src, int64_t n,
                                   int64_t* __restrict__ deg) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t row = glx_row_of(map, src[i]);
  deg[i] = row < 0 ? 0 : row_ptr[row + 1] - row_ptr[row];
}

static inline unsigned grid_for(int64_t n, int cap = 8192) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b < cap ? b : cap);
}

// ----------------------------------------------------------------- create --
void glx_graph_free(glx_graph* g) {
  if (!g) return;
  if (g->row_ptr) (void)hipFree(g->row_ptr);
  if (g->adj) (void)hipFree(g->adj);
  if (g->weight) (void)hipFree(g->weight);
  if (g->alias) (void)hipFree(g->alias);
  if (g->alias_indeg) (void)hipFree(g->alias_indeg);
  if (g->nbr_sorted) (void)hipFree(g->nbr_sorted);
  if (g->slot_sorted) (void)hipFree(g->slot_sorted);
  if (g->dst_count) (void)hipFree(g->dst_count);
  glx_idmap_free(&g->dst_map);
  if (g->ew) (void)hipFree(g->ew);
  if (g->ts) (void)hipFree(g->ts);
  glx_idmap_free(&g->idmap);
  delete g;
}

static int graph_create_impl(glx_graph* g, const int64_t* row_ptr, const int64_t* col,
                             const int64_t* eid, const float* weight, const int64_t* ids,
                             int ptr_kind, hipStream_t s) {
  const int64_t V = g->num_rows, E = g->num_edges;
  const hipMemcpyKind kind = ptr_kind == GLX_PTR_HOST ? hipMemcpyHostToDevice
                                                      : hipMemcpyDeviceToDevice;
  GLX_HIP(hipMalloc(&g->row_ptr, (size_t)(V + 1) * sizeof(int64_t)));
  GLX_HIP(hipMalloc(&g->adj, (size_t)(E > 0 ? E : 1) * sizeof(GlxAdj)));
  GLX_HIP(hipMemcpyAsync(g->row_ptr, row_ptr, (size_t)(V + 1) * sizeof(int64_t), kind, s));

  GlxTemp flag_buf, stage, ids_buf;
  GLX_HIP(hipMalloc(&flag_buf.p, sizeof(int)));
  int* d_flag = flag_buf.as<int>();
  GLX_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), s));
  glx_check_row_ptr_kernel<<<(unsigned)((V + 256) / 256), 256, 0, s>>>(g->row_ptr, V, E, d_flag);
  // Validate before any kernel trusts row_ptr for addressing.
  int flag = 0;
  GLX_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  GLX_REQUIRE((flag & 1) == 0, "row_ptr must start at 0, end at num_edges and be non-decreasing");
  GLX_REQUIRE((flag & 2) == 0, "a row has 2^31 or more neighbours (row-local indices are int32, as in the reference)");

  // Stage col/eid on the device (host input) then pack them into 16-byte slots.
  const int64_t* d_col = col;
  const int64_t* d_eid = eid;
  if (ptr_kind == GLX_PTR_HOST && E > 0) {
    GLX_HIP(hipMalloc(&stage.p, (size_t)E * 2 * sizeof(int64_t)));
    int64_t* tmp = stage.as<int64_t>();
    GLX_HIP(hipMemcpyAsync(tmp, col, (size_t)E * sizeof(int64_t), kind, s));
    GLX_HIP(hipMemcpyAsync(tmp + E, eid, (size_t)E * sizeof(int64_t), kind, s));
    d_col = tmp;
    d_eid = tmp + E;
  }
  if (E > 0) glx_pack_adj_kernel<<<grid_for(E), 256, 0, s>>>(d_col, d_eid, E, g->adj);

  if (weight) {
    GLX_HIP(hipMalloc(&g->weight, (size_t)(E > 0 ? E : 1) * sizeof(float)));
    if (E > 0) GLX_HIP(hipMemcpyAsync(g->weight, weight, (size_t)E * sizeof(float), kind, s));
  }
  const int64_t* d_ids = nullptr;
  if (ids) {
    d_ids = ids;
    if (ptr_kind == GLX_PTR_HOST) {
      GLX_HIP(hipMalloc(&ids_buf.p, (size_t)(V > 0 ? V : 1) * sizeof(int64_t)));
      GLX_HIP(hipMemcpyAsync(ids_buf.p, ids, (size_t)V * sizeof(int64_t), kind, s));
      d_ids = ids_buf.as<int64_t>();
    }
  }
  return glx_graph_finalize(g, d_ids, s);
}

int glx_alias_build_launch(const int64_t* row_ptr, const float* weight, int64_t V, int64_t E,
                           GlxAlias* out, hipStream_t s) {
  if (E <= 0 || V <= 0) return GLX_OK;
  GlxTemp stk_buf;
  GLX_HIP(hipMalloc(&stk_buf.p, (size_t)E * sizeof(GlxAlias)));
  glx_alias_build_kernel<<<(unsigned)((V + 63) / 64), 64, 0, s>>>(row_ptr, weight, V, out,
                                                                  stk_buf.as<GlxAlias>());
  glx_alias_build_wave_kernel<<<(unsigned)((V * 64 + 255) / 256), 256, 0, s>>>(row_ptr, weight, V, out,
                                                                               stk_buf.as<GlxAlias>());
  GLX_HIP(hipStreamSynchronize(s));  // stk_buf is released on return
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// The per-row alias tables over g->weight, and the packed EdgeWeight records on top of them.
static int glx_graph_build_alias(glx_graph* g, hipStream_t s) {
  const int64_t V = g->num_rows, E = g->num_edges;
  {
    GLX_HIP(hipMalloc(&g->alias, (size_t)(E > 0 ? E : 1) * sizeof(GlxAlias)));
    int rc = glx_alias_build_launch(g->row_ptr, g->weight, V, E, g->alias, s);
    if (rc != GLX_OK) return rc;
    const char* env = getenv("GLX_EW_PACKED");  // "0" disables the packed fast path
    if (E > 0 && !(env && env[0] == '0')) {
      GlxTemp bad;
      GLX_HIP(hipMalloc(&bad.p, sizeof(int)));
      GLX_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), s));
      GLX_HIP(hipMalloc(&g->ew, (size_t)E * sizeof(GlxEwRec)));
      glx_pack_ew_kernel<<<(unsigned)((V * 64 + 255) / 256), 256, 0, s>>>(g->row_ptr, g->adj, g->alias, V, g->ew,
                                                                         bad.as<int>());
      int h_bad = 0;
      GLX_HIP(hipMemcpyAsync(&h_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
      GLX_HIP(hipStreamSynchronize(s));
      if (h_bad) {
        (void)hipFree(g->ew);
        g->ew = nullptr;
      }
    }
  }
  return GLX_OK;
}

int glx_graph_finalize(glx_graph* g, const int64_t* d_ids, hipStream_t s) {
  const int64_t V = g->num_rows;
  if (g->weight) {
    int rc = glx_graph_build_alias(g, s);
    if (rc != GLX_OK) return rc;
  }
  if (d_ids) {
    int rc = glx_idmap_build(d_ids, V, &g->idmap, s);
    if (rc != GLX_OK) return rc;
  }
  GLX_HIP(hipStreamSynchronize(s));
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

extern "C" int glx_graph_create(int device, int64_t num_rows, int64_t num_edges,
                                const int64_t* row_ptr, const int64_t* col, const int64_t* eid,
                                const float* weight, const int64_t* ids, int ptr_kind, void* stream,
                                glx_graph** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(num_rows >= 0 && num_edges >= 0, "negative sizes");
  GLX_REQUIRE(num_rows < INT32_MAX, "num_rows must be < 2^31 (row indices are int32 in the id map)");
  GLX_REQUIRE(row_ptr && (num_edges == 0 || (col && eid)), "row_ptr/col/eid must not be NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  glx_graph* g = new (std::nothrow) glx_graph();
  GLX_REQUIRE(g != nullptr, "out of host memory");
  memset(static_cast<void*>(g), 0, sizeof(*g));
  g->device = device;
  g->num_rows = num_rows;
  g->num_edges = num_edges;
  rc = graph_create_impl(g, row_ptr, col, eid, weight, ids, ptr_kind, glx_stream(stream));
  if (rc != GLX_OK) {
    glx_graph_free(g);
    return rc;
  }
  *out = g;
  return GLX_OK;
}

__global__ void glx_fill_float_kernel(float* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

extern "C" int glx_graph_enable_default_weight(glx_graph* g, float default_weight, void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  if (g->weight) return GLX_OK;  // a weighted type keeps its weights
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = glx_stream(stream);
  const int64_t E = g->num_edges;
  GLX_HIP(hipMalloc(&g->weight, (size_t)(E > 0 ? E : 1) * sizeof(float)));
  if (E > 0) glx_fill_float_kernel<<<grid_for(E), 256, 0, s>>>(g->weight, E, default_weight);
  int rc = glx_graph_build_alias(g, s);
  GLX_HIP(hipStreamSynchronize(s));
  GLX_HIP(hipGetLastError());
  return rc;
}

extern "C" void glx_graph_destroy(glx_graph* g) {
  if (!g) return;
  GlxDeviceGuard guard(g->device);
  glx_graph_free(g);
}

extern "C" int glx_graph_info(const glx_graph* g, int64_t* num_rows, int64_t* num_edges,
                              int* weighted, int* has_id_map, int* device) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  if (num_rows) *num_rows = g->num_rows;
  if (num_edges) *num_edges = g->num_edges;
  if (weighted) *weighted = g->alias != nullptr;
  if (has_id_map) *has_id_map = g->idmap.any();
  if (device) *device = g->device;
  return GLX_OK;
}

extern "C" int glx_graph_export_alias(const glx_graph* g, float* prob, int32_t* alias,
                                      int ptr_kind, void* stream) {
  GLX_REQUIRE(g && prob && alias, "NULL argument");
  GLX_REQUIRE(g->alias != nullptr, "graph has no weights, hence no alias table");
  GlxDeviceGuard guard(g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  const int64_t E = g->num_edges;
  if (E == 0) return GLX_OK;
  if (ptr_kind == GLX_PTR_DEVICE) {
    glx_unpack_alias_kernel<<<grid_for(E), 256, 0, s>>>(g->alias, E, prob, alias);
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  float* d_prob = nullptr;
  int32_t* d_alias = nullptr;
  GLX_HIP(hipMalloc(&d_prob, (size_t)E * sizeof(float)));
  GLX_HIP(hipMalloc(&d_alias, (size_t)E * sizeof(int32_t)));
  glx_unpack_alias_kernel<<<grid_for(E), 256, 0, s>>>(g->alias, E, d_prob, d_alias);
  GLX_HIP(hipMemcpyAsync(prob, d_prob, (size_t)E * sizeof(float), hipMemcpyDeviceToHost, s));
  GLX_HIP(hipMemcpyAsync(alias, d_alias, (size_t)E * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  (void)hipFree(d_prob);
  (void)hipFree(d_alias);
  return GLX_OK;
}

extern "C" int glx_graph_degrees(const glx_graph* g, const int64_t* src, int64_t n,
                                 int64_t* deg_out, int ptr_kind, void* stream) {
  GLX_REQUIRE(g && (n == 0 || (src && deg_out)), "NULL argument");
  GLX_REQUIRE(n >= 0, "negative n");
  if (n == 0) return GLX_OK;
  GlxDeviceGuard guard(g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  if (ptr_kind == GLX_PTR_DEVICE) {
    glx_degrees_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(g->map(), g->row_ptr, src, n,
                                                                   deg_out);
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  int64_t* d = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), (size_t)n * 2 * sizeof(int64_t), s, 0);
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipMemcpyAsync(d, src, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, s));
  glx_degrees_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(g->map(), g->row_ptr, d, n, d + n);
  GLX_HIP(hipMemcpyAsync(deg_out, d + n, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  glx_scratch_free(d, s);
  return GLX_OK;
}
