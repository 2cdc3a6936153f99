// glx conditional negative sampling: replaces ConditionalNegativeSampler
// (graphlearn/src/core/operator/sampler/conditional_negative_sampler.cc:37-161) over ConditionTable
// (condition_table.cc:65-148) and AttributeNodesMap (attribute_nodes_map.h:74-127).
//
// Condition table (built once per type, on the device): per selected attribute column the candidates are radix-sorted by
// (attribute key, candidate position) -- the groups of AttributeNodesMap::Insert, members in candidate order --
// run-length encoded into groups, and every group gets its AliasMethod table over the members' weights (CreateAM) from
// the per-row alias builder of the graph storage (glx_alias_build_launch: bit-identical to alias_method.cc:57-107).
// The default table is one more alias row over all candidates.
//
// The reference's exclusion set (nbr_set) is declared before the row loop and never cleared, so row i rejects the
// neighbours and dst ids of rows 0..i -- and, with `unique`, everything accepted so far.
//   * Without `unique` the set at row i is known up front: pass 1 records, per id, the FIRST row that inserts it; pass 2
//     samples one row per wave, "in the set" = "first row <= mine".  Should a row exhaust its default-sampling retries
//     -- where the reference drops the set for every later row -- the request is replayed sequentially.
//   * With `unique` the rows depend on each other by definition: one wave walks them in order.
// Inside a row either way:
//   * the set is an open-addressing table in HBM sized for everything the request can insert; lanes insert the
//     neighbours of src i with atomicCAS and probe with device-scope atomic loads (L2-coherent inside the wave);
//   * the candidates of a block are evaluated one per lane; acceptance in the reference's order is a ballot + prefix
//     popcount, `unique` adds an in-block duplicate test (an id equal to an earlier candidate of the block that was not
//     in the set is a repeat);
//   * AttributeNodesMap::Sample's retry schedule is kept exactly, including its last block of which only the first
//     entry is looked at (attribute_nodes_map.h:109-125).
// The reference's fill loop is dead code (it derives "how many do I have" from the static response shape); here it runs
// as written.
#include <stdlib.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <vector>

#include "glx_common.h"

struct glx_cond_table {
  int device;
  int64_t num_ids;
  int32_t num_cols;
  int64_t* ids;          // [U] candidates
  int64_t* member;       // [num_cols][U] candidate ids grouped by key, candidate order inside a group
  GlxAlias* member_tab;  // [num_cols][U] alias tables, one per group
  int64_t* group_key;    // [num_cols][U] sorted distinct keys (first num_groups[c] entries used)
  int64_t* group_off;    // [num_cols][U + 1] member offsets of the groups
  int64_t* num_groups;   // [num_cols] (device)
  GlxAlias* default_tab; // [U] alias table over all candidates
};

namespace {

#define GLX_ROCPRIM_C(call)                               \
  do {                                                    \
    size_t bytes__ = 0;                                   \
    GLX_HIP(call(nullptr, bytes__));                      \
    GlxTemp tmp__;                                        \
    GLX_HIP(hipMalloc(&tmp__.p, bytes__ ? bytes__ : 16)); \
    GLX_HIP(call(tmp__.p, bytes__));                      \
    GLX_HIP(hipStreamSynchronize(s));                     \
  } while (0)

inline unsigned grid_of(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b);
}

__global__ void glx_cond_iota_kernel(int64_t* p, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

__global__ void glx_cond_gather_kernel(const int64_t* __restrict__ pos, const int64_t* __restrict__ ids,
                                       const float* __restrict__ w, int64_t n, int64_t* __restrict__ member,
                                       float* __restrict__ member_w) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  member[i] = ids[pos[i]];
  member_w[i] = w ? w[pos[i]] : 1.0f;
}

__global__ void glx_cond_fill_f32_kernel(float* p, int64_t n, float v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void glx_cond_fill_i64_kernel(int64_t* p, int64_t n, int64_t v) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += step) p[i] = v;
}

__global__ void glx_cond_set2_kernel(int64_t* p, int64_t a, int64_t b) {
  p[0] = a;
  p[1] = b;
}

// ---- the exclusion set -------------------------------------------------------------------------
__device__ __forceinline__ void set_insert(int64_t* tab, uint64_t mask, int64_t v) {
  if (v == GLX_EMPTY_KEY) return;
  uint64_t h = glx_mix64((uint64_t)v) & mask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {  // bounded: a full table drops the id instead of spinning
    const unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long*>(&tab[h]), (unsigned long long)GLX_EMPTY_KEY, (unsigned long long)v);
    if ((int64_t)prev == GLX_EMPTY_KEY || (int64_t)prev == v) return;
    h = (h + 1) & mask;
  }
}

__device__ __forceinline__ bool set_has(const int64_t* tab, uint64_t mask, int64_t v) {
  if (v == GLX_EMPTY_KEY) return false;
  uint64_t h = glx_mix64((uint64_t)v) & mask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    const int64_t k = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == v) return true;
    if (k == GLX_EMPTY_KEY) return false;
    h = (h + 1) & mask;
  }
  return false;
}

// ---- the exclusion set of a request WITHOUT `unique`, as a function of the row --------------------------------
// Without `unique` nothing a row samples enters the set, so what row i rejects is known before any row samples: the
// neighbours and dst ids of rows 0..i (all dst ids when batch_share).  One table maps an id to the FIRST row that
// inserts it; "in the set at row i" = "first row <= i".  Rows then sample independently, one wave each -- unless a
// row exhausts its default-sampling retries, where the reference drops the whole set for every later row
// (nbr_set.clear()): such a request is replayed by the sequential kernel.
struct FirstEnt {
  int64_t key;
  int64_t row;
};

__device__ __forceinline__ void first_insert(FirstEnt* tab, uint64_t mask, int64_t v, int64_t row) {
  if (v == GLX_EMPTY_KEY) return;
  uint64_t h = glx_mix64((uint64_t)v) & mask;
  while (true) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[h].key),
                                              (unsigned long long)GLX_EMPTY_KEY, (unsigned long long)v);
    if ((int64_t)prev == GLX_EMPTY_KEY || (int64_t)prev == v) {
      atomicMin(reinterpret_cast<long long*>(&tab[h].row), (long long)row);
      return;
    }
    h = (h + 1) & mask;
  }
}

__device__ __forceinline__ bool first_has(const FirstEnt* tab, uint64_t mask, int64_t v, int64_t row) {
  if (v == GLX_EMPTY_KEY) return false;
  uint64_t h = glx_mix64((uint64_t)v) & mask;
  while (true) {
    const FirstEnt e = tab[h];  // complete before the sampling kernel starts: plain loads
    if (e.key == v) return e.row <= row;
    if (e.key == GLX_EMPTY_KEY) return false;
    h = (h + 1) & mask;
  }
}

struct CondArgs {
  const int64_t* ids;
  const int64_t* member;
  const GlxAlias* member_tab;
  const int64_t* group_key;
  const int64_t* group_off;
  const int64_t* num_groups;
  const GlxAlias* default_tab;
  int64_t U;
  int32_t ncols;
  const int32_t* num_c;  // [ncols] (device): (int32)(count * props[c])
  // neighbours of src (may be absent)
  GlxIdMap map;
  const int64_t* row_ptr;
  const GlxAdj* adj;
  int has_graph;
  const int64_t* src;
  const int64_t* dst;
  const int64_t* dst_keys;
  int32_t batch, count;
  int batch_share, unique;
  int32_t retry;
  int64_t default_nbr;
  uint64_t seed, cc;
  int64_t* set;
  uint64_t set_mask;
  // parallel rows (no `unique`): first-insertion table + "a row wanted to drop the set" flag
  FirstEnt* first;
  uint64_t first_mask;
  int* replay;
  int64_t* out;
};

// Evaluates up to `look` candidates of one block (draws first_draw + j) from an alias row `tab` of `n` members whose
// ids are members[0..n) and accepts them in order against the set.  Returns through got/taken.
template <bool PAR>
__device__ __forceinline__ void cond_block(const CondArgs& a, int lane, int32_t row, uint32_t first_draw, int32_t look,
                                           const int64_t* members, const GlxAlias* tab, int64_t n, int32_t want,
                                           int64_t* orow, int32_t& got, int32_t& taken) {
  for (int32_t j0 = 0; j0 < look && got < want; j0 += 64) {
    const int32_t j = j0 + lane;
    const bool active = j < look;
    int64_t item = GLX_EMPTY_KEY;
    bool ok = false;
    if (active) {
      const uint64_t u = glx_draw64(a.seed, a.cc, (uint32_t)row, first_draw + (uint32_t)j);
      item = members[glx_alias_pick(u, n, tab)];
      ok = PAR ? !first_has(a.first, a.first_mask, item, row) : !set_has(a.set, a.set_mask, item);
    }
    if (!PAR && a.unique) {
      // an id equal to an earlier candidate of this chunk: that one was not in the set either, so it was accepted (or
      // the row was already full) and this one is a repeat
      bool dup = false;
      for (int t = 0; t < 64; ++t) {
        const int64_t it = __shfl(item, t);
        const bool act = __shfl((int)active, t) != 0;
        if (t < lane && act && it == item) dup = true;
      }
      ok = ok && !dup;
    }
    const uint64_t m = __ballot(ok);
    const int32_t rank = (int32_t)__popcll(m & ((1ull << lane) - 1ull));
    const bool fin = ok && rank < want - got;
    if (fin) {
      const int32_t pos = taken + rank;
      if (pos < a.count) orow[pos] = item;
      if (!PAR && a.unique) set_insert(a.set, a.set_mask, item);
    }
    int32_t acc = (int32_t)__popcll(m);
    if (acc > want - got) acc = want - got;
    got += acc;
    taken += acc;
  }
}

// One request row: the condition columns, then the default sampler, then default ids.
template <bool PAR>
__device__ __forceinline__ void cond_row(const CondArgs& a, int lane, int32_t i) {
  int64_t* orow = a.out + (int64_t)i * a.count;
  int32_t taken = 0;
  uint32_t base = 0;
  // an empty candidate set (a condition table of an edge type without edges) has no group tables at all
  for (int32_t c = 0; a.U > 0 && c < a.ncols; ++c) {
    const int32_t n = a.num_c[c];
    if (n <= 0) continue;
    const int64_t key = a.dst_keys[(int64_t)i * a.ncols + c];
    const int64_t* gk = a.group_key + (int64_t)c * a.U;
    const int64_t* go = a.group_off + (int64_t)c * (a.U + 1);
    const int64_t G = a.num_groups[c];
    int64_t lo = 0, hi = G;
    while (lo < hi) {  // every lane the same search
      const int64_t mid = (lo + hi) >> 1;
      if (gk[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (key != GLX_EMPTY_KEY && lo < G && gk[lo] == key) {
      const int64_t g0 = go[lo], gn = go[lo + 1] - g0;
      const int64_t* members = a.member + (int64_t)c * a.U + g0;
      const GlxAlias* tab = a.member_tab + (int64_t)c * a.U + g0;
      int32_t got = 0;
      for (int32_t blk = 0; blk < a.retry && got < n; ++blk) {
        const int32_t look = (blk == a.retry - 1) ? 1 : n;  // the last block: its first entry only
        cond_block<PAR>(a, lane, i, base + (uint32_t)(blk * n), look, members, tab, gn, n, orow, got, taken);
      }
    }
    base += (uint32_t)a.retry * (uint32_t)n;
  }
  if (taken > a.count) taken = a.count;
  // default sampling (conditional_negative_sampler.cc:128-152, as written)
  if (a.U > 0) {
    int32_t retry_times = a.retry + 1, blk = 0;
    bool last = false;
    while (taken < a.count && !last) {
      if (--retry_times <= 0) {  // nbr_set.clear()
        if (PAR) {  // the set changes for every later row: this request must be replayed row by row
          if (lane == 0) *a.replay = 1;
          return;
        }
        for (uint64_t h = lane; h <= a.set_mask; h += 64) {
          __hip_atomic_store(&a.set[h], (int64_t)GLX_EMPTY_KEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (retry_times < 0) last = true;
      int32_t got = taken;
      cond_block<PAR>(a, lane, i, base + (uint32_t)(blk * a.count), last ? 1 : a.count, a.ids, a.default_tab, a.U, a.count,
                      orow, got, taken);
      taken = got;
      ++blk;
    }
  }
  for (int32_t j = taken + lane; j < a.count; j += 64) orow[j] = a.default_nbr;
}

// Sequential rows (the set evolves: `unique`, or a replay after a dropped set): ONE wave walks the request.
__global__ __launch_bounds__(64) void glx_cond_sample_kernel(CondArgs a) {
  const int lane = threadIdx.x;
  if (a.batch_share) {
    for (int32_t i = lane; i < a.batch; i += 64) set_insert(a.set, a.set_mask, a.dst[i]);
  }
  for (int32_t i = 0; i < a.batch; ++i) {
    if (!a.batch_share) {
      if (a.has_graph) {
        const int64_t r = glx_row_of(a.map, a.src[i]);
        if (r >= 0) {
          const int64_t s0 = a.row_ptr[r], s1 = a.row_ptr[r + 1];
          for (int64_t e = s0 + lane; e < s1; e += 64) set_insert(a.set, a.set_mask, a.adj[e].nbr);
        }
      }
      if (lane == 0) set_insert(a.set, a.set_mask, a.dst[i]);
    }
    cond_row<false>(a, lane, i);
  }
}

// Parallel rows (no `unique`): pass 1 records the first row that inserts every id, pass 2 samples one row per wave.
__global__ __launch_bounds__(256) void glx_cond_first_kernel(CondArgs a) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= a.batch) return;
  if (a.batch_share) {
    if (lane == 0) first_insert(a.first, a.first_mask, a.dst[i], -1);  // in the set before row 0 samples
    return;
  }
  if (a.has_graph) {
    const int64_t r = glx_row_of(a.map, a.src[i]);
    if (r >= 0) {
      const int64_t s0 = a.row_ptr[r], s1 = a.row_ptr[r + 1];
      for (int64_t e = s0 + lane; e < s1; e += 64) first_insert(a.first, a.first_mask, a.adj[e].nbr, i);
    }
  }
  if (lane == 0) first_insert(a.first, a.first_mask, a.dst[i], i);
}

__global__ __launch_bounds__(256) void glx_cond_sample_rows_kernel(CondArgs a) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (i >= a.batch) return;
  cond_row<true>(a, threadIdx.x & 63, (int32_t)i);
}

__global__ void glx_cond_fill_first_kernel(FirstEnt* p, int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += step) p[i] = FirstEnt{GLX_EMPTY_KEY, INT64_MAX};
}

__global__ void glx_cond_degsum_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ src,
                                       int32_t batch, unsigned long long* total) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const int64_t r = glx_row_of(map, src[i]);
  if (r >= 0) atomicAdd(total, (unsigned long long)(row_ptr[r + 1] - row_ptr[r]));
}

void free_cond(glx_cond_table* t) {
  if (!t) return;
  GlxDeviceGuard guard(t->device);
  for (void* p : {(void*)t->ids, (void*)t->member, (void*)t->member_tab, (void*)t->group_key, (void*)t->group_off,
                  (void*)t->num_groups, (void*)t->default_tab}) {
    if (p) (void)hipFree(p);
  }
  delete t;
}

struct CondOwner {
  glx_cond_table* t = nullptr;
  ~CondOwner() { free_cond(t); }
};

}  // namespace

extern "C" int glx_cond_table_create(int device, int64_t num_ids, const int64_t* ids, const float* weights, int32_t num_cols,
                                     const int64_t* cand_keys, int ptr_kind, void* stream, glx_cond_table** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(num_ids >= 0 && num_ids < INT32_MAX, "num_ids must be in [0, 2^31)");
  GLX_REQUIRE(num_cols >= 0 && num_cols <= 64, "num_cols must be in [0, 64]");
  GLX_REQUIRE(num_ids == 0 || ids != nullptr, "ids is NULL");
  GLX_REQUIRE(num_ids == 0 || num_cols == 0 || cand_keys != nullptr, "cand_keys is NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  hipStream_t s = glx_stream(stream);
  CondOwner own;
  own.t = new glx_cond_table();
  memset(static_cast<void*>(own.t), 0, sizeof(*own.t));
  glx_cond_table* t = own.t;
  t->device = device;
  t->num_ids = num_ids;
  t->num_cols = num_cols;
  const int64_t U = num_ids;
  if (U > 0) {
    const hipMemcpyKind kind = ptr_kind == GLX_PTR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const size_t nc = (size_t)(num_cols > 0 ? num_cols : 1);
    GLX_HIP(hipMalloc(&t->ids, (size_t)U * 8));
    GLX_HIP(hipMemcpyAsync(t->ids, ids, (size_t)U * 8, kind, s));
    GlxTemp w, keys, keys_s, pos, pos_s, member_w, cnt, nruns, row2;
    if (weights) {
      GLX_HIP(hipMalloc(&w.p, (size_t)U * 4));
      GLX_HIP(hipMemcpyAsync(w.p, weights, (size_t)U * 4, kind, s));
    }
    GLX_HIP(hipMalloc(&t->member, nc * (size_t)U * 8));
    GLX_HIP(hipMalloc(&t->member_tab, nc * (size_t)U * sizeof(GlxAlias)));
    GLX_HIP(hipMalloc(&t->group_key, nc * (size_t)U * 8));
    GLX_HIP(hipMalloc(&t->group_off, nc * ((size_t)U + 1) * 8));
    GLX_HIP(hipMalloc(&t->num_groups, nc * 8));
    GLX_HIP(hipMemsetAsync(t->num_groups, 0, nc * 8, s));
    GLX_HIP(hipMalloc(&t->default_tab, (size_t)U * sizeof(GlxAlias)));
    GLX_HIP(hipMalloc(&keys.p, (size_t)U * 8));
    GLX_HIP(hipMalloc(&keys_s.p, (size_t)U * 8));
    GLX_HIP(hipMalloc(&pos.p, (size_t)U * 8));
    GLX_HIP(hipMalloc(&pos_s.p, (size_t)U * 8));
    GLX_HIP(hipMalloc(&member_w.p, (size_t)U * 4));
    GLX_HIP(hipMalloc(&cnt.p, ((size_t)U + 1) * 8));
    GLX_HIP(hipMalloc(&nruns.p, 8));
    GLX_HIP(hipMalloc(&row2.p, 16));
    for (int32_t c = 0; c < num_cols; ++c) {
      GLX_HIP(hipMemcpyAsync(keys.p, cand_keys + (size_t)c * U, (size_t)U * 8, kind, s));
      glx_cond_iota_kernel<<<grid_of(U), 256, 0, s>>>(pos.as<int64_t>(), U);
      // stable: candidate order survives inside a key
#define SORTK(tmp, bytes) \
  rocprim::radix_sort_pairs(tmp, bytes, keys.as<int64_t>(), keys_s.as<int64_t>(), pos.as<int64_t>(), pos_s.as<int64_t>(), (size_t)U, 0, 64, s)
      GLX_ROCPRIM_C(SORTK);
#undef SORTK
      int64_t* gkey = t->group_key + (size_t)c * U;
      int64_t* goff = t->group_off + (size_t)c * (U + 1);
      GLX_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)U + 1) * 8, s));
#define RLE(tmp, bytes) \
  rocprim::run_length_encode(tmp, bytes, keys_s.as<int64_t>(), (size_t)U, gkey, cnt.as<int64_t>(), nruns.as<int64_t>(), s)
      GLX_ROCPRIM_C(RLE);
#undef RLE
      int64_t G = 0;
      GLX_HIP(hipMemcpyAsync(&G, nruns.p, 8, hipMemcpyDeviceToHost, s));
      GLX_HIP(hipStreamSynchronize(s));
      GLX_HIP(hipMemcpyAsync(t->num_groups + c, nruns.p, 8, hipMemcpyDeviceToDevice, s));
#define SCANG(tmp, bytes) \
  rocprim::exclusive_scan(tmp, bytes, cnt.as<int64_t>(), goff, (int64_t)0, (size_t)G + 1, rocprim::plus<int64_t>(), s)
      GLX_ROCPRIM_C(SCANG);
#undef SCANG
      glx_cond_gather_kernel<<<grid_of(U), 256, 0, s>>>(pos_s.as<int64_t>(), t->ids, w.as<float>(), U,
                                                      t->member + (size_t)c * U, member_w.as<float>());
      rc = glx_alias_build_launch(goff, member_w.as<float>(), G, U, t->member_tab + (size_t)c * U, s);
      if (rc != GLX_OK) return rc;
      GLX_HIP(hipStreamSynchronize(s));
    }
    // default table: one alias row over all candidates (AliasMethod(weights) / AliasMethod(ids.Size()))
    if (!weights) {
      GLX_HIP(hipMalloc(&w.p, (size_t)U * 4));
      glx_cond_fill_f32_kernel<<<grid_of(U), 256, 0, s>>>(w.as<float>(), U, 1.0f);
    }
    glx_cond_set2_kernel<<<1, 1, 0, s>>>(row2.as<int64_t>(), 0, U);
    rc = glx_alias_build_launch(row2.as<int64_t>(), w.as<float>(), 1, U, t->default_tab, s);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipGetLastError());
    GLX_HIP(hipStreamSynchronize(s));
  }
  *out = own.t;
  own.t = nullptr;
  return GLX_OK;
}

extern "C" void glx_cond_table_destroy(glx_cond_table* t) { free_cond(t); }

extern "C" int glx_cond_negative_sample(const glx_cond_table* t, const glx_graph* g, const int64_t* src, const int64_t* dst,
                                        const int64_t* dst_keys, const float* props, int32_t batch, int32_t count,
                                        int batch_share, int unique, int32_t retry_times, int64_t default_neighbor_id,
                                        uint64_t seed, uint64_t call_counter, int64_t* out, int ptr_kind, void* stream) {
  GLX_REQUIRE(t != nullptr, "condition table is NULL");
  GLX_REQUIRE(batch >= 0 && count >= 0, "negative batch / count");
  GLX_REQUIRE((int64_t)batch * count <= INT32_MAX, "batch * count exceeds int32 (tensor.h:47)");
  GLX_REQUIRE(retry_times >= 0, "negative retry_times");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE(g == nullptr || g->device == t->device, "graph and condition table live on different devices");
  if (batch == 0 || count == 0) return GLX_OK;
  GLX_REQUIRE(src && dst && out, "NULL data pointer");
  GLX_REQUIRE(t->num_cols == 0 || (dst_keys && props), "NULL dst_keys / props");
  GlxDeviceGuard guard(t->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", t->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, t->device) : glx_stream(stream);
  const int32_t ncols = t->num_cols;
  // staged inputs
  GlxTemp d_src, d_dst, d_keys, d_out, d_num, d_set, d_total;
  const int64_t* p_src = src;
  const int64_t* p_dst = dst;
  const int64_t* p_keys = dst_keys;
  int64_t* p_out = out;
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMalloc(&d_src.p, (size_t)batch * 8));
    GLX_HIP(hipMalloc(&d_dst.p, (size_t)batch * 8));
    GLX_HIP(hipMalloc(&d_keys.p, (size_t)batch * (size_t)(ncols > 0 ? ncols : 1) * 8));
    GLX_HIP(hipMalloc(&d_out.p, (size_t)batch * count * 8));
    GLX_HIP(hipMemcpyAsync(d_src.p, src, (size_t)batch * 8, hipMemcpyHostToDevice, s));
    GLX_HIP(hipMemcpyAsync(d_dst.p, dst, (size_t)batch * 8, hipMemcpyHostToDevice, s));
    if (ncols > 0) GLX_HIP(hipMemcpyAsync(d_keys.p, dst_keys, (size_t)batch * ncols * 8, hipMemcpyHostToDevice, s));
    p_src = d_src.as<int64_t>();
    p_dst = d_dst.as<int64_t>();
    p_keys = d_keys.as<int64_t>();
    p_out = d_out.as<int64_t>();
  }
  std::vector<int32_t> num_c((size_t)(ncols > 0 ? ncols : 1), 0);
  int64_t by_columns = 0;
  for (int32_t c = 0; c < ncols; ++c) {
    GLX_REQUIRE(props[c] >= 0.0f && props[c] <= 1.0f, "props[%d] = %g is not a proportion", c, (double)props[c]);
    num_c[(size_t)c] = (int32_t)((float)count * props[c]);  // neg_num * props[i]
    by_columns += num_c[(size_t)c];
  }
  // the columns share one row of `count` slots; with `unique` every accepted id also enters the exclusion set, which is
  // sized for count ids per row -- proportions that add up to more than one would overfill both (the Python wrapper
  // refused them already; the C entry point must too: a full set makes its probe loop spin)
  GLX_REQUIRE(by_columns <= count, "the column proportions ask for %lld of %d negatives per row (sum(props) > 1)",
              (long long)by_columns, count);
  GLX_HIP(hipMalloc(&d_num.p, num_c.size() * 4));
  GLX_HIP(hipMemcpyAsync(d_num.p, num_c.data(), num_c.size() * 4, hipMemcpyHostToDevice, s));
  // everything the request can insert: its dst ids, the neighbours of its src ids, every accepted id
  unsigned long long deg_total = 0;
  const bool with_graph = g != nullptr && !batch_share && g->num_edges > 0;
  if (with_graph) {
    GLX_HIP(hipMalloc(&d_total.p, 8));
    GLX_HIP(hipMemsetAsync(d_total.p, 0, 8, s));
    glx_cond_degsum_kernel<<<grid_of(batch), 256, 0, s>>>(g->map(), g->row_ptr, p_src, batch,
                                                        d_total.as<unsigned long long>());
    GLX_HIP(hipMemcpyAsync(&deg_total, d_total.p, 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  uint64_t want = (uint64_t)batch * (uint64_t)(count + 1) + deg_total;
  uint64_t cap = 64;
  while (cap < 2 * want + 2) cap <<= 1;
  CondArgs a;
  memset(static_cast<void*>(&a), 0, sizeof(a));
  a.ids = t->ids;
  a.member = t->member;
  a.member_tab = t->member_tab;
  a.group_key = t->group_key;
  a.group_off = t->group_off;
  a.num_groups = t->num_groups;
  a.default_tab = t->default_tab;
  a.U = t->num_ids;
  a.ncols = ncols;
  a.num_c = d_num.as<int32_t>();
  if (with_graph) {
    a.map = g->map();
    a.row_ptr = g->row_ptr;
    a.adj = g->adj;
    a.has_graph = 1;
  }
  a.src = p_src;
  a.dst = p_dst;
  a.dst_keys = p_keys;
  a.batch = batch;
  a.count = count;
  a.batch_share = batch_share ? 1 : 0;
  a.unique = unique ? 1 : 0;
  a.retry = retry_times;
  a.default_nbr = default_neighbor_id;
  a.seed = seed;
  a.cc = call_counter;
  a.out = p_out;
  const unsigned row_grid = (unsigned)(((int64_t)batch * 64 + 255) / 256);
  int replay = 1;
  const bool force_seq = glx_side_knobs().cond_sequential.load(std::memory_order_relaxed) > 0;  // A/B and test knob
  if (!unique && !force_seq) {
    // rows are independent given the first-insertion table: one wave per row
    GlxTemp d_first, d_flag;
    GLX_HIP(hipMalloc(&d_first.p, (size_t)cap * sizeof(FirstEnt)));
    GLX_HIP(hipMalloc(&d_flag.p, sizeof(int)));
    GLX_HIP(hipMemsetAsync(d_flag.p, 0, sizeof(int), s));
    {
      int64_t blocks = ((int64_t)cap + 255) / 256;
      glx_cond_fill_first_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, s>>>(d_first.as<FirstEnt>(), (int64_t)cap);
    }
    a.first = d_first.as<FirstEnt>();
    a.first_mask = cap - 1;
    a.replay = d_flag.as<int>();
    glx_cond_first_kernel<<<row_grid, 256, 0, s>>>(a);
    glx_cond_sample_rows_kernel<<<row_grid, 256, 0, s>>>(a);
    GLX_HIP(hipGetLastError());
    GLX_HIP(hipMemcpyAsync(&replay, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  if (replay) {
    // `unique` (every accepted id joins the set), or a row of the parallel pass wanted to drop the set: row by row
    GLX_HIP(hipMalloc(&d_set.p, (size_t)cap * 8));
    int64_t blocks = ((int64_t)cap + 255) / 256;
    glx_cond_fill_i64_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, s>>>(d_set.as<int64_t>(), (int64_t)cap,
                                                                                      GLX_EMPTY_KEY);
    a.first = nullptr;
    a.replay = nullptr;
    a.set = d_set.as<int64_t>();
    a.set_mask = cap - 1;
    glx_cond_sample_kernel<<<1, 64, 0, s>>>(a);
  }
  GLX_HIP(hipGetLastError());
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMemcpyAsync(out, p_out, (size_t)batch * count * 8, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));  // the temporaries are released on return
  return GLX_OK;
}
