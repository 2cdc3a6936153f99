// Status, flags, constants and the vector-backed Tensor of the glx host layer.
#include <cstdlib>
#include <mutex>
#include <new>
#include <sys/mman.h>

#include <unordered_map>
#include <unordered_set>
#include <string>
#include <vector>
#include <cstring>

#include "glx.h"
#include "graphlearn/config.h"
#include "graphlearn/constants.h"
#include "graphlearn/status.h"
#include "graphlearn/tensor.h"

namespace graphlearn {

// ----------------------------------------------------------------- status --
std::string Status::ToString() const {
  if (ok()) return "OK";
  return "code " + std::to_string(static_cast<int>(code_)) + ": " + msg_;
}

namespace error {
Status FromGlx(int code) {
  if (code == GLX_OK) return Status::OK();
  return Status(static_cast<Code>(code), glx_last_error());
}
}  // namespace error

// ------------------------------------------------------------------ flags --
int32_t gPaddingMode = 1;
int64_t gDefaultNeighborId = 0;
float gDefaultFloatAttribute = 0.0f;
float gDefaultWeight = 0.0f;
int64_t gDefaultIntAttribute = 0;
std::string gDefaultStringAttribute = "";
int64_t gDefaultLabel = -1;
int64_t gDefaultTimestamp = -1;
int32_t gIgnoreInvalid = 1;
int32_t gSamplingRetryTimes = 5;
int32_t gDefaultFullNbrNum = 100;
int32_t gShuffleBufferSize = 10240;
int32_t gDeployMode = 0;
int32_t gClientId = 0;
int32_t gClientCount = 1;
int32_t gServerCount = 1;
int32_t gTimeout = 60;
int32_t gTapeCapacity = 10;
int32_t gDatasetCapacity = 10;
int32_t gTrackerMode = 1;
int64_t gSamplingSeed = 0;
int32_t gDeviceId = 0;

void SetGlobalFlagPaddingMode(int32_t v) { gPaddingMode = v; }
void SetGlobalFlagDefaultNeighborId(int64_t v) { gDefaultNeighborId = v; }
void SetGlobalFlagDefaultFloatAttribute(float v) { gDefaultFloatAttribute = v; }
void SetGlobalFlagDefaultWeight(float v) { gDefaultWeight = v; }
void SetGlobalFlagDefaultIntAttribute(int64_t v) { gDefaultIntAttribute = v; }
void SetGlobalFlagDefaultStringAttribute(const std::string& v) { gDefaultStringAttribute = v; }
void SetGlobalFlagDefaultLabel(int64_t v) { gDefaultLabel = v; }
void SetGlobalFlagDefaultTimestamp(int64_t v) { gDefaultTimestamp = v; }
void SetGlobalFlagIgnoreInvalid(int32_t v) { gIgnoreInvalid = v; }
void SetGlobalFlagSamplingRetryTimes(int32_t v) { gSamplingRetryTimes = v; }
void SetGlobalFlagDefaultFullNbrNum(int32_t v) { gDefaultFullNbrNum = v; }
void SetGlobalFlagShuffleBufferSize(int32_t v) { gShuffleBufferSize = v; }
void SetGlobalFlagDeployMode(int32_t v) { gDeployMode = v; }
void SetGlobalFlagClientId(int32_t v) { gClientId = v; }
void SetGlobalFlagClientCount(int32_t v) { gClientCount = v; }
void SetGlobalFlagServerCount(int32_t v) { gServerCount = v; }
void SetGlobalFlagTimeout(int32_t v) { gTimeout = v; }
void SetGlobalFlagTapeCapacity(int32_t v) { gTapeCapacity = v; }
void SetGlobalFlagDatasetCapacity(int32_t v) { gDatasetCapacity = v; }
void SetGlobalFlagTrackerMode(int32_t v) { gTrackerMode = v; }
int32_t GetGlobalFlagTrackerMode() { return gTrackerMode; }
namespace {
std::mutex g_unused_mtx;
std::unordered_map<std::string, std::string>& UnusedFlags() {
  static auto* m = new std::unordered_map<std::string, std::string>;
  return *m;
}
}  // namespace
void SetGlobalFlagUnused(const char* name, int64_t v) { SetGlobalFlagUnused(name, std::to_string(v)); }
void SetGlobalFlagUnused(const char* name, const std::string& v) {
  std::lock_guard<std::mutex> g(g_unused_mtx);
  UnusedFlags()[name] = v;
}
void SetGlobalFlagSamplingSeed(int64_t v) { gSamplingSeed = v; }
void SetGlobalFlagDeviceId(int32_t v) { gDeviceId = v; }

// -------------------------------------------------------------- constants --
// Key strings: the reference's values (service/constants.cc:20-72), see constants.h.
#define GLX_DEFINE_TENSOR_KEY(name, wire) const char* name = wire;
GLX_TENSOR_KEYS(GLX_DEFINE_TENSOR_KEY)
#undef GLX_DEFINE_TENSOR_KEY

// ----------------------------------------------------------------- tensor --
// Storage of the numeric tensors.  A response of the device path is one large block
// written by a single device->host copy, so two things matter on the host side:
// (1) no value-initialisation of memory that is about to be overwritten, and
// (2) no fresh mmap + page faults per request, and (3) the block should be PINNED: a
// device->host copy into pageable memory bounces through the runtime's staging buffers (one
// extra pass over every response byte, serialised between the pool threads), a copy into
// registered memory is a single DMA straight into the response.  BlockPool recycles large
// blocks (>= 64 KiB, power-of-two classes, at most kPoolCap bytes pinned in all; anonymous mappings of the pool's own,
// never the malloc heap -- see Take()) across requests and
// registers each block with the GPU runtime ONCE, when it is first obtained
// (glx_host_register; pinning costs milliseconds, recycling a pinned block nothing);
// small tensors use the ordinary heap.  Without a GPU runtime registration fails and the
// block simply stays pageable.
namespace {
class BlockPool {
public:
  static BlockPool& Get() {
    static BlockPool* pool = new BlockPool;  // never destroyed: tensors may outlive static teardown
    return *pool;
  }
  static constexpr size_t kMinBytes = 64u << 10;
  static constexpr size_t kPoolCap = 4ull << 30;
  static int ClassOf(size_t bytes) {
    int c = 18;
    while ((1ull << c) < bytes) ++c;
    return c;
  }
  void* Take(size_t bytes) {
    const int c = ClassOf(bytes);
    bool pin = false;
    {
      std::lock_guard<std::mutex> g(mtx_);
      if (c < 48 && !free_[c].empty()) {
        void* p = free_[c].back();
        free_[c].pop_back();
        return p;
      }
      if (c < 48 && pinned_ + (1ull << c) <= kPoolCap) {
        pinned_ += 1ull << c;
        pin = true;
      }
    }
    if (!pin) {  // beyond the pool's cap: an ordinary heap block, freed on return
      void* p = nullptr;
      if (posix_memalign(&p, 4096, 1ull << c) != 0) throw std::bad_alloc();
      return p;
    }
    // The pool's blocks live in anonymous mappings of their OWN, 2 MiB aligned and whole 2 MiB granules -- never in the
    // malloc heap (round 6).  A process that has hipHostRegister-ed ranges of its heap and ALSO holds heap memory
    // marked MADV_HUGEPAGE (numpy does that to every array of 4 MiB or more) gets "an illegal memory access" from
    // later pageable host-to-device copies on ROCm 7.0 -- reproduced without any glx code in
    // scripts/r06/repro/hostreg_pageable.hip (registered ranges cut from the heap: a fault within ~50 rounds; cut from
    // mappings of their own: none in 600); a test that registered numpy page ranges killed one run in ~10 of this repo's GPU
    // suite that way.  The pool's former posix_memalign blocks stayed out of the brk heap only while glibc served their
    // size as mmap chunks -- its threshold grows with every larger mmapped chunk freed -- so their placement is explicit
    // now (the old pool did not fault in 1,200 rounds of scripts/r06/pool_pageable_stress.py: a precaution, not a repair).
    // Small classes are carved from 2 MiB slabs that are registered whole, ONCE (pinning costs milliseconds per call).
    constexpr size_t kGranule = 2u << 20;
    const size_t bytes_c = 1ull << c;
    std::lock_guard<std::mutex> g(map_mtx_);
    if (bytes_c < kGranule) {
      if (slab_left_ < bytes_c) {
        slab_ = MapAligned(kGranule, kGranule);
        slab_left_ = kGranule;
        if (pin_) (void)glx_host_register(slab_, kGranule);  // best effort: an unregistered block is only slower
      }
      void* p = slab_;
      slab_ = static_cast<char*>(slab_) + bytes_c;
      slab_left_ -= bytes_c;
      std::lock_guard<std::mutex> g2(mtx_);
      owned_.insert(p);
      return p;
    }
    void* p = MapAligned(bytes_c, kGranule);
    if (pin_) (void)glx_host_register(p, bytes_c);
    {
      std::lock_guard<std::mutex> g2(mtx_);
      owned_.insert(p);
    }
    return p;
  }
  // A pinned block is parked for ever, never unregistered and freed: pageable copies from addresses that WERE
  // registered once and have been recycled by the heap since were seen to fault in the runtime (ROCm 7.0; the
  // parity suite hit it through a test that registered and released numpy buffers).  So the pool pins at most
  // kPoolCap bytes over the life of the process and hands out ordinary heap blocks beyond that.
  void Give(void* p, size_t bytes) {
    const int c = ClassOf(bytes);
    {
      std::lock_guard<std::mutex> g(mtx_);
      if (owned_.count(p)) {
        free_[c].push_back(p);
        return;
      }
    }
    std::free(p);
  }

private:
  // `bytes` (a multiple of `align`) of zero pages in a private anonymous mapping, aligned to `align`; never unmapped
  static void* MapAligned(size_t bytes, size_t align) {
    void* m = mmap(nullptr, bytes + align, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(m) + align - 1) & ~(uintptr_t)(align - 1);
    // give the unaligned head and tail back: the pool's mappings must not share a granule with anything else
    if (lo > reinterpret_cast<uintptr_t>(m)) (void)munmap(m, lo - reinterpret_cast<uintptr_t>(m));
    const uintptr_t end = reinterpret_cast<uintptr_t>(m) + bytes + align;
    if (end > lo + bytes) (void)munmap(reinterpret_cast<void*>(lo + bytes), end - (lo + bytes));
    // never a transparent huge page: a collapse would migrate pages the GPU runtime has mapped (the boxes measured run
    // THP in `madvise` mode, where nobody asks for these ranges; this covers hosts that run it in `always` mode)
    (void)madvise(reinterpret_cast<void*>(lo), bytes, MADV_NOHUGEPAGE);
    return reinterpret_cast<void*>(lo);
  }
  std::mutex map_mtx_;  // the slab cursor and the mappings (taken before mtx_)
  void* slab_ = nullptr;
  size_t slab_left_ = 0;
  std::mutex mtx_;
  std::vector<void*> free_[48];
  std::unordered_set<void*> owned_;  // the pool's blocks (in use or parked; pinned unless pinning is off)
  size_t pinned_ = 0;                // their bytes
  // GLX_HOST_PINNED_RESPONSES=0 keeps response blocks pageable (A/B knob)
  const bool pin_ = !(std::getenv("GLX_HOST_PINNED_RESPONSES") && std::atoi(std::getenv("GLX_HOST_PINNED_RESPONSES")) == 0);
};

template <class T>
struct PoolAlloc {
  typedef T value_type;
  PoolAlloc() = default;
  template <class U>
  PoolAlloc(const PoolAlloc<U>&) {}
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= BlockPool::kMinBytes) return static_cast<T*>(BlockPool::Get().Take(bytes));
    return static_cast<T*>(::operator new(bytes));
  }
  void deallocate(T* p, size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= BlockPool::kMinBytes) BlockPool::Get().Give(p, bytes);
    else ::operator delete(p);
  }
  template <class U>
  void construct(U* p) { ::new (static_cast<void*>(p)) U; }  // default-init: no zero fill
  template <class U, class... Args>
  void construct(U* p, Args&&... args) { ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...); }
  template <class U>
  bool operator==(const PoolAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const PoolAlloc<U>&) const { return false; }
};
}  // namespace

struct Tensor::Impl {
  DataType type = kUnknown;
  std::vector<int32_t, PoolAlloc<int32_t>> i32;
  std::vector<int64_t, PoolAlloc<int64_t>> i64;
  std::vector<float, PoolAlloc<float>> f32;
  std::vector<double, PoolAlloc<double>> f64;
  std::vector<std::string> str;
};

std::shared_ptr<const void> Tensor::Owner() const { return impl_; }

Tensor::Tensor() : impl_(std::make_shared<Impl>()) {}
Tensor::Tensor(DataType dtype) : impl_(std::make_shared<Impl>()) { impl_->type = dtype; }
Tensor::Tensor(DataType dtype, int32_t capacity) : impl_(std::make_shared<Impl>()) {
  impl_->type = dtype;
  if (capacity > 0) {
    switch (dtype) {
      case kInt32: impl_->i32.reserve(capacity); break;
      case kInt64: impl_->i64.reserve(capacity); break;
      case kFloat: impl_->f32.reserve(capacity); break;
      case kDouble: impl_->f64.reserve(capacity); break;
      case kString: impl_->str.reserve(capacity); break;
      default: break;
    }
  }
}
DataType Tensor::DType() const { return impl_->type; }
int32_t Tensor::Size() const {
  switch (impl_->type) {
    case kInt32: return (int32_t)impl_->i32.size();
    case kInt64: return (int32_t)impl_->i64.size();
    case kFloat: return (int32_t)impl_->f32.size();
    case kDouble: return (int32_t)impl_->f64.size();
    case kString: return (int32_t)impl_->str.size();
    default: return 0;
  }
}
void Tensor::Resize(int32_t size) {
  switch (impl_->type) {
    case kInt32: impl_->i32.resize(size); break;
    case kInt64: impl_->i64.resize(size); break;
    case kFloat: impl_->f32.resize(size); break;
    case kDouble: impl_->f64.resize(size); break;
    case kString: impl_->str.resize(size); break;
    default: break;
  }
}
void Tensor::AddInt32(int32_t v) { impl_->i32.push_back(v); }
void Tensor::AddInt64(int64_t v) { impl_->i64.push_back(v); }
void Tensor::AddFloat(float v) { impl_->f32.push_back(v); }
void Tensor::AddDouble(double v) { impl_->f64.push_back(v); }
void Tensor::AddString(const std::string& v) { impl_->str.push_back(v); }
void Tensor::AddInt32(const int32_t* b, const int32_t* e) { impl_->i32.insert(impl_->i32.end(), b, e); }
void Tensor::AddInt64(const int64_t* b, const int64_t* e) { impl_->i64.insert(impl_->i64.end(), b, e); }
void Tensor::AddFloat(const float* b, const float* e) { impl_->f32.insert(impl_->f32.end(), b, e); }
void Tensor::AddDouble(const double* b, const double* e) { impl_->f64.insert(impl_->f64.end(), b, e); }
void Tensor::SetInt32(int32_t i, int32_t v) { impl_->i32[i] = v; }
void Tensor::SetInt64(int32_t i, int64_t v) { impl_->i64[i] = v; }
void Tensor::SetFloat(int32_t i, float v) { impl_->f32[i] = v; }
int32_t Tensor::GetInt32(int32_t i) const { return impl_->i32[i]; }
int64_t Tensor::GetInt64(int32_t i) const { return impl_->i64[i]; }
float Tensor::GetFloat(int32_t i) const { return impl_->f32[i]; }
double Tensor::GetDouble(int32_t i) const { return impl_->f64[i]; }
const std::string& Tensor::GetString(int32_t i) const { return impl_->str[i]; }
const int32_t* Tensor::GetInt32() const { return impl_->i32.data(); }
const int64_t* Tensor::GetInt64() const { return impl_->i64.data(); }
const float* Tensor::GetFloat() const { return impl_->f32.data(); }
const double* Tensor::GetDouble() const { return impl_->f64.data(); }
int32_t* Tensor::MutableInt32() { return impl_->i32.data(); }
int64_t* Tensor::MutableInt64() { return impl_->i64.data(); }
float* Tensor::MutableFloat() { return impl_->f32.data(); }
void Tensor::Swap(Tensor& right) { std::swap(impl_, right.impl_); }

}  // namespace graphlearn
