// glx shard exchange helpers: device-side HashPartitioner + Stitcher.
// Replaces graphlearn/src/core/partition/hash_partitioner.h:33-92 (shard =
// llabs(id) % P, stable inside a shard, Sticker = original indices) and
// graphlearn/src/core/partition/stitcher.h:67-107 (scatter rows back by
// sticker).  These run between the two RCCL all-to-alls of a sharded request;
// the exchange itself is done by the caller (torch.distributed / RCCL).
#include "glx_common.h"

namespace {

constexpr int kTile = 2048;       // ids per workgroup (8 passes of 256) ...
constexpr int kTileSmall = 512;   // ... and for requests up to kSmallIds: a 64 K-id request on 32 workgroups was eight
constexpr int64_t kSmallIds = 1 << 18;  // dependent round trips long (44 us; the 1.6 M-id one beside it took 34)
constexpr int kMaxShards = 64;
constexpr int64_t kOwnScanCells = 32768;  // tile histogram cells (buckets x tiles) up to which the scatter kernel scans them itself

__device__ __forceinline__ int32_t shard_of(int64_t id, int32_t P) {
  // llabs(id) % P (hash_partitioner.h:90-92)
  uint64_t a = id < 0 ? (uint64_t)0 - (uint64_t)id : (uint64_t)id;
  if (a <= 0xffffffffull) return (int32_t)((uint32_t)a % (uint32_t)P);  // the 32-bit remainder is a quarter of the 64-bit one
  return (int32_t)(a % (uint64_t)P);
}

// With a divert map the request has one bucket more: ids the map knows go to bucket P - 1 (rows a local replica
// serves, glx_dist.hip), everything else to llabs(id) % (P - 1).
__device__ __forceinline__ int32_t bucket_of(int64_t id, int32_t P, const GlxIdMap& divert, const GlxMember& member) {
  if (divert.keys == nullptr && divert.step == 0) return shard_of(id, P);
  if (member.bits != nullptr) {  // the divert map's ids, one bit each
    const bool in = id >= 0 && id <= member.max && ((member.bits[id >> 6] >> (id & 63)) & 1ull);
    return in ? P - 1 : shard_of(id, P - 1);
  }
  return glx_row_of(divert, id) >= 0 ? P - 1 : shard_of(id, P - 1);
}

// block_counts is shard-major: [P][nblocks].  bucket_cache (or null): the bucket of every id, written here and read by
// the scatter kernel -- with a divert map a bucket costs a hash probe, which is then paid once per id, not twice.
__global__ __launch_bounds__(256) void glx_part_count_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                             int32_t P, int64_t nblocks, int32_t tile, GlxIdMap divert,
                                                             GlxMember member, int64_t* __restrict__ block_counts,
                                                             uint8_t* __restrict__ bucket_cache) {
  __shared__ int32_t cnt[kMaxShards];
  if (threadIdx.x < kMaxShards) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = blockIdx.x * (int64_t)tile;
  for (int it = 0; it < tile / 256; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    if (i < n) {
      const int32_t b = bucket_of(ids[i], P, divert, member);
      if (bucket_cache) bucket_cache[i] = (uint8_t)b;
      atomicAdd(&cnt[b], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < P) block_counts[(int64_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];
}

// Exclusive scan of the flattened [P * nblocks] counts (single workgroup), plus
// per-shard totals.
__global__ __launch_bounds__(1024) void glx_part_scan_kernel(int64_t* __restrict__ block_counts,
                                                             int64_t nblocks, int32_t P,
                                                             int64_t* __restrict__ counts, GlxPartitionTail tail) {
  __shared__ int64_t wave_sum[16];
  __shared__ int64_t carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int64_t total = (int64_t)P * nblocks;
  for (int64_t base = 0; base < total; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = i < total ? block_counts[i] : 0;
    int64_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int64_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wave_sum[w];
    const int64_t carry = carry_s;
    if (i < total) block_counts[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  // shard totals from the scanned offsets
  if (threadIdx.x < P) {
    const int64_t p = threadIdx.x;
    const int64_t begin = block_counts[p * nblocks];
    const int64_t end = (p + 1 < P) ? block_counts[(p + 1) * nblocks] : carry_s;
    if (tail.n == 0 || p < tail.at) counts[p] = end - begin;
  }
  if ((int)threadIdx.x < tail.n) counts[tail.at + threadIdx.x] = tail.v[threadIdx.x];
}

// An empty request: its bucket sizes (all zero) and its tail, in one launch.
__global__ void glx_part_empty_kernel(int64_t* __restrict__ counts, int32_t P, GlxPartitionTail tail) {
  const int t = threadIdx.x;
  if (t < P && (tail.n == 0 || t < tail.at)) counts[t] = 0;
  if (t < tail.n) counts[tail.at + t] = tail.v[t];
}

// kOwnScan: block_off holds the raw per-block counts of glx_part_count_kernel and every block derives its own
// offsets from them (sum of the whole buckets before its bucket + its bucket's counts in the blocks before it) --
// O(P * nblocks) L2 reads per block, which for requests of a few million ids is cheaper than a third kernel in the
// chain: a one-workgroup scan between two launches waits for a CU behind whatever else runs (0.7 ms instead of
// 15 us beside a segmented reduce).  Block 0 also writes the bucket totals.  Otherwise block_off is the scanned table.
template <bool kOwnScan>
__global__ __launch_bounds__(256) void glx_part_scatter_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                               int32_t P, int64_t nblocks, int32_t tile, GlxIdMap divert,
                                                               GlxMember member,
                                                               const int64_t* __restrict__ block_off,
                                                               const uint8_t* __restrict__ bucket_cache,
                                                               int64_t* __restrict__ bucketed,
                                                               int64_t* __restrict__ order,
                                                               int64_t* __restrict__ counts, GlxPartitionTail tail) {
  __shared__ int64_t run[kMaxShards];       // next output position per shard for this block
  __shared__ int32_t wave_cnt[4][kMaxShards];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (kOwnScan) {
    __shared__ int64_t before[kMaxShards], total[kMaxShards];
    for (int32_t p = wid; p < P; p += 4) {  // one wave per bucket at a time
      int64_t b = 0, t = 0;
      for (int64_t j = lane; j < nblocks; j += 64) {
        const int64_t c = block_off[(int64_t)p * nblocks + j];
        t += c;
        if (j < (int64_t)blockIdx.x) b += c;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        b += __shfl_xor(b, off);
        t += __shfl_xor(t, off);
      }
      if (lane == 0) {
        before[p] = b;
        total[p] = t;
      }
    }
    __syncthreads();
    if (threadIdx.x < P) {
      int64_t base = 0;
      for (int32_t q = 0; q < (int32_t)threadIdx.x; ++q) base += total[q];
      run[threadIdx.x] = base + before[threadIdx.x];
      if (blockIdx.x == 0 && (tail.n == 0 || (int)threadIdx.x < tail.at)) counts[threadIdx.x] = total[threadIdx.x];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail.n) counts[tail.at + threadIdx.x] = tail.v[threadIdx.x];
  } else if (threadIdx.x < P) {
    run[threadIdx.x] = block_off[(int64_t)threadIdx.x * nblocks + blockIdx.x];
  }
  __syncthreads();
  const int64_t base = blockIdx.x * (int64_t)tile;
  for (int it = 0; it < tile / 256; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool valid = i < n;
    const int64_t id = valid ? ids[i] : 0;
    const int32_t sh = !valid ? -1 : (bucket_cache ? (int32_t)bucket_cache[i] : bucket_of(id, P, divert, member));
    int32_t my_rank = 0;
    for (int32_t p = 0; p < P; ++p) {
      const uint64_t b = __ballot(sh == p);
      if (sh == p) my_rank = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[wid][p] = __popcll(b);
    }
    __syncthreads();
    if (valid) {
      int64_t pos = run[sh] + my_rank;
      for (int w = 0; w < wid; ++w) pos += wave_cnt[w][sh];
      bucketed[pos] = id;
      order[pos] = i;
    }
    __syncthreads();
    if (threadIdx.x < P) {
      run[threadIdx.x] += wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] +
                          wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void glx_stitch_kernel(const T* __restrict__ in,
                                                         const int64_t* __restrict__ order, int64_t n,
                                                         int32_t width, T* __restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = n * width;
  if (t >= total) return;
  const int64_t i = t / width;
  const int32_t c = (int32_t)(t - i * width);
  out[order[i] * width + c] = in[t];
}

}  // namespace

static int partition_impl(int device, const int64_t* ids, int64_t n, int32_t num_buckets, GlxIdMap divert,
                          GlxMember member, int64_t* bucketed, int64_t* order, int64_t* counts, hipStream_t s,
                          const GlxPartitionTail& tail = GlxPartitionTail{{0}, 0, 0}) {
  if (n == 0) {
    glx_part_empty_kernel<<<1, 64, 0, s>>>(counts, num_buckets, tail);
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  const int32_t tile = n <= kSmallIds ? kTileSmall : kTile;
  const int64_t nblocks = (n + tile - 1) / tile;
  int64_t* block_counts = nullptr;
  const size_t cells_b = (size_t)num_buckets * nblocks * sizeof(int64_t);
  // a bucket that costs a hash probe is remembered (one byte per id); a bitmap or arithmetic test is cheaper recomputed
  const bool probe = divert.keys != nullptr && member.bits == nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&block_counts), cells_b + (probe ? (size_t)n : 0), s, 1);
  if (rc != GLX_OK) return rc;
  uint8_t* bucket_cache = probe ? reinterpret_cast<uint8_t*>(block_counts) + cells_b : nullptr;
  glx_part_count_kernel<<<(unsigned)nblocks, 256, 0, s>>>(ids, n, num_buckets, nblocks, tile, divert, member, block_counts,
                                                          bucket_cache);
  if ((int64_t)num_buckets * nblocks <= kOwnScanCells) {
    glx_part_scatter_kernel<true><<<(unsigned)nblocks, 256, 0, s>>>(ids, n, num_buckets, nblocks, tile, divert, member,
                                                                   block_counts, bucket_cache, bucketed, order, counts, tail);
  } else {
    glx_part_scan_kernel<<<1, 1024, 0, s>>>(block_counts, nblocks, num_buckets, counts, tail);
    glx_part_scatter_kernel<false><<<(unsigned)nblocks, 256, 0, s>>>(ids, n, num_buckets, nblocks, tile, divert, member,
                                                                    block_counts, bucket_cache, bucketed, order, counts,
                                                                    GlxPartitionTail{{0}, 0, 0});
  }
  hipError_t e = hipGetLastError();
  glx_scratch_free(block_counts, s);
  GLX_HIP(e);
  return GLX_OK;
}

extern "C" int glx_partition(int device, const int64_t* ids, int64_t n, int32_t num_shards,
                             int64_t* bucketed, int64_t* order, int64_t* counts, void* stream) {
  GLX_REQUIRE(num_shards >= 1 && num_shards <= kMaxShards, "num_shards must be in [1, %d]", kMaxShards);
  GLX_REQUIRE(n >= 0, "negative n");
  GLX_REQUIRE(counts != nullptr && (n == 0 || (ids && bucketed && order)), "NULL data pointer");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  return partition_impl(device, ids, n, num_shards, GlxIdMap{nullptr, nullptr, 0, 0}, GlxMember{nullptr, -1}, bucketed, order, counts,
                        glx_stream(stream));
}

// The partition of a request that a local replica serves in part: buckets 0 .. num_shards - 1 as glx_partition,
// bucket num_shards = the ids `divert` knows (counts has num_shards + 1 entries).  Device already selected.
int glx_partition_divert(int device, const int64_t* ids, int64_t n, int32_t num_shards, GlxIdMap divert, GlxMember member,
                         int64_t* bucketed, int64_t* order, int64_t* counts, hipStream_t s) {
  GLX_REQUIRE(num_shards >= 1 && num_shards + 1 <= kMaxShards, "num_shards must be in [1, %d)", kMaxShards);
  GLX_REQUIRE(divert.keys != nullptr || divert.step > 0, "no divert map");
  return partition_impl(device, ids, n, num_shards + 1, divert, member, bucketed, order, counts, s);
}

int glx_partition_tail(int device, const int64_t* ids, int64_t n, int32_t num_shards, GlxIdMap divert, GlxMember member,
                       int64_t* bucketed, int64_t* order, int64_t* counts, const GlxPartitionTail& tail, hipStream_t s) {
  const bool diverted = divert.keys != nullptr || divert.step > 0;
  GLX_REQUIRE(num_shards >= 1 && num_shards + (diverted ? 1 : 0) <= kMaxShards, "num_shards must be in [1, %d)", kMaxShards);
  GLX_REQUIRE(tail.n >= 0 && tail.n <= 12 && (tail.n == 0 || tail.at >= 0), "a partition tail holds at most 12 values");
  return partition_impl(device, ids, n, num_shards + (diverted ? 1 : 0), divert, member, bucketed, order, counts, s, tail);
}

template <typename T>
static int stitch_impl(int device, const T* in, const int64_t* order, int64_t n, int32_t width, T* out,
                       void* stream) {
  GLX_REQUIRE(n >= 0 && width >= 1, "bad sizes");
  if (n == 0) return GLX_OK;
  GLX_REQUIRE(in && order && out, "NULL data pointer");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  const int64_t total = n * width;
  glx_stitch_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, glx_stream(stream)>>>(in, order, n,
                                                                                       width, out);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

extern "C" int glx_stitch_i64(int device, const int64_t* in, const int64_t* order, int64_t n,
                              int32_t width, int64_t* out, void* stream) {
  return stitch_impl<int64_t>(device, in, order, n, width, out, stream);
}

extern "C" int glx_stitch_f32(int device, const float* in, const int64_t* order, int64_t n,
                              int32_t width, float* out, void* stream) {
  return stitch_impl<float>(device, in, order, n, width, out, stream);
}
