// glx negative samplers: RandomNegativeSampler, InDegreeNegativeSampler,
// SoftInDegreeNegativeSampler, NodeWeightNegativeSampler on the device.
// Replaces graphlearn/src/core/operator/sampler/random_negative_sampler.cc:30-63,
// in_degree_negative_sampler.cc:29-135 and node_weight_negative_sampler.cc:29-110.
//
// All four draw from ONE global candidate list: the edge type's distinct destination
// ids in first-appearance order with their in-degrees (TopoStatics::Add,
// topo_statics.cc:32-55) or a node type's ids with their weights.  The weighted ones use
// one AliasMethod table over the whole list (AliasMethodFactory::LookupOrCreate, built
// once per type), with the reference's biased draw (alias_method.cc:117-121).  The
// strict ones reject candidates found in an exclusion set -- the source's neighbours
// (in-degree) or the request's own ids (node-weight) -- and reproduce the reference's
// retry schedule exactly: candidates come in blocks of `count` draws, rejected ones are
// skipped, after three blocks the exclusion set is dropped (kRetryTimes = 3), and
// accepted candidates keep their draw order.
#include <string.h>  // rocprim's texture_cache_iterator uses memset

#include <rocprim/rocprim.hpp>

#include <vector>

#include "glx_common.h"

struct glx_negative {
  int device;
  int64_t num_ids;
  int64_t* ids;     // [U] candidates
  GlxAlias* table;  // [U] alias table over the candidates' weights, or nullptr (uniform)
};

namespace {

#define GLX_ROCPRIM(call)                                 \
  do {                                                    \
    size_t bytes__ = 0;                                   \
    GLX_HIP(call(nullptr, bytes__));                      \
    GlxTemp tmp__;                                        \
    GLX_HIP(hipMalloc(&tmp__.p, bytes__ ? bytes__ : 16)); \
    GLX_HIP(call(tmp__.p, bytes__));                      \
    GLX_HIP(hipStreamSynchronize(s));                     \
  } while (0)

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

__global__ void glx_neg_split_adj_kernel(const GlxAdj* __restrict__ adj, int64_t E, int64_t* __restrict__ nbr,
                                         int64_t* __restrict__ eid) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= E) return;
  const GlxAdj a = adj[i];
  nbr[i] = a.nbr;
  eid[i] = a.eid;
}

__global__ void glx_neg_to_float_kernel(const int64_t* __restrict__ in, int64_t n, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

__global__ void glx_neg_fill_kernel(int64_t* __restrict__ out, int64_t n, int64_t value) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = value;
}

struct MinOp {
  __host__ __device__ int64_t operator()(int64_t a, int64_t b) const { return a < b ? a : b; }
};

__device__ __forceinline__ bool glx_sorted_contains(const int64_t* __restrict__ a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int64_t x = a[mid];
    if (x == v) return true;
    if (x < v) lo = mid + 1;
    else hi = mid;
  }
  return false;
}

struct NegArgs {
  const int64_t* ids;
  const GlxAlias* table;
  int64_t num_ids;
  const int64_t* src;
  int32_t batch, count;
  uint64_t seed, cc;
  int64_t* out;
  // GLX_NEG_EXCLUDE_NEIGHBORS
  GlxIdMap map;
  const int64_t* row_ptr;
  const int64_t* nbr_sorted;
  // GLX_NEG_EXCLUDE_BATCH
  const int64_t* batch_sorted;
  // ... whose exclusion set is ONE object for the whole request in the reference (node_weight_negative_sampler.cc:68:
  // `sets` is built before the row loop), so the `sets.clear()` of the first row that exhausts its retries (:80) also
  // frees every LATER row from it.  first_dropped: the smallest such row index (atomicMin; batch = none), found by the
  // main kernel; glx_negative_after_drop_kernel then redoes the rows behind it without a set.
  unsigned long long* first_dropped;
  // rows of a partitioned request (glx_dist_negative_sample): row r draws from the stream of rng_rows[r], its index in
  // the ORIGINAL request; nullptr = r itself
  const int64_t* rng_rows;
};

// One group of W lanes (W = 8, 16, 32 or 64: the smallest that covers `count`, so a
// wavefront serves 64 / W request rows) per row.  Draw d of row i is word d of the stream
// (seed, cc, i); block b of the reference's retry loop uses draws [b * count, (b + 1) * count).
template <int MODE, int W>
__global__ __launch_bounds__(256) void glx_negative_kernel(NegArgs a) {
  constexpr int kGroups = 64 / W;  // per wavefront
  const int lane = threadIdx.x & 63;
  const int sub = lane & (W - 1);
  const int grp = lane / W;
  const int64_t row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (int64_t)kGroups + grp;
  const bool live = row < a.batch;
  const int64_t* ex = nullptr;
  int64_t exn = 0;
  if (live && MODE == GLX_NEG_EXCLUDE_NEIGHBORS) {
    const int64_t r = glx_row_of(a.map, a.src[row]);
    if (r >= 0) {
      ex = a.nbr_sorted + a.row_ptr[r];
      exn = a.row_ptr[r + 1] - a.row_ptr[r];
    }
  } else if (live && MODE == GLX_NEG_EXCLUDE_BATCH) {
    ex = a.batch_sorted;
    exn = a.batch;
  }
  const int32_t n = a.count;
  const uint64_t group_mask = W == 64 ? ~0ull : (((1ull << W) - 1ull) << (grp * W));
  const uint32_t stream_row = (live && a.rng_rows) ? (uint32_t)a.rng_rows[row] : (uint32_t)row;
  int32_t taken = live ? 0 : n;
  // every lane of the wavefront runs the same trip count; finished groups just idle
  for (int32_t blk = 0; blk < 4; ++blk) {
    const bool strict = MODE != GLX_NEG_EXCLUDE_NONE && blk < 3;  // the 4th block drops the set
    if (MODE == GLX_NEG_EXCLUDE_BATCH && blk == 3 && live && taken < n && sub == 0) {
      atomicMin(a.first_dropped, (unsigned long long)row);  // this row clears the request's set
    }
    for (int32_t base = 0; base < n; base += W) {
      if (__ballot(taken < n) == 0) return;
      const int32_t j = base + sub;
      bool ok = false;
      int64_t item = 0;
      if (taken < n && j < n) {
        const uint64_t u = glx_draw64(a.seed, a.cc, stream_row, (uint32_t)(blk * n + j));
        const int64_t idx = a.table ? (int64_t)glx_alias_pick(u, a.num_ids, a.table)
                                    : (int64_t)glx_bounded(u, (uint64_t)a.num_ids);
        item = a.ids[idx];
        ok = !strict || !glx_sorted_contains(ex, exn, item);
      }
      const uint64_t m = __ballot(ok) & group_mask;
      const int32_t pos = taken + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (ok && pos < n) a.out[row * (int64_t)n + pos] = item;
      taken += (int32_t)__popcll(m);
    }
  }
}

// Rows behind the first one that dropped the request's set (see NegArgs::first_dropped): no exclusion at all, so the
// first block's `count` draws are the answer.  One thread per output slot.
__global__ void glx_negative_after_drop_kernel(NegArgs a) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.batch * a.count;
  if (t >= total) return;
  const int64_t row = t / a.count;
  if ((unsigned long long)row <= *a.first_dropped) return;
  const int32_t j = (int32_t)(t - row * a.count);
  const uint32_t stream_row = a.rng_rows ? (uint32_t)a.rng_rows[row] : (uint32_t)row;
  const uint64_t u = glx_draw64(a.seed, a.cc, stream_row, (uint32_t)j);
  const int64_t idx = a.table ? (int64_t)glx_alias_pick(u, a.num_ids, a.table) : (int64_t)glx_bounded(u, (uint64_t)a.num_ids);
  a.out[t] = a.ids[idx];
}

template <int MODE>
void launch_negative(const NegArgs& a, hipStream_t s) {
  const int32_t n = a.count;
  const int w = n <= 8 ? 8 : n <= 16 ? 16 : n <= 32 ? 32 : 64;
  const int64_t rows_per_block = 4 * (64 / w);
  const unsigned grid = (unsigned)((a.batch + rows_per_block - 1) / rows_per_block);
  switch (w) {
    case 8: glx_negative_kernel<MODE, 8><<<grid, 256, 0, s>>>(a); break;
    case 16: glx_negative_kernel<MODE, 16><<<grid, 256, 0, s>>>(a); break;
    case 32: glx_negative_kernel<MODE, 32><<<grid, 256, 0, s>>>(a); break;
    default: glx_negative_kernel<MODE, 64><<<grid, 256, 0, s>>>(a); break;
  }
}

int build_table(glx_negative* t, const float* d_weights, hipStream_t s) {
  // One distribution over all candidates: AliasMethod::Build is a serial algorithm, so
  // this load-time step runs on the host (a single GPU lane would take seconds for 10^7
  // entries); the table then lives in HBM.
  const int64_t U = t->num_ids;
  std::vector<float> w((size_t)U);
  GLX_HIP(hipMemcpyAsync(w.data(), d_weights, (size_t)U * sizeof(float), hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  std::vector<GlxAlias> tab((size_t)U);
  std::vector<int32_t> stk((size_t)U);
  glx_alias_build_row(w.data(), (int32_t)U, tab.data(), stk.data(), stk.data() + U - 1);
  GLX_HIP(hipMalloc(&t->table, (size_t)U * sizeof(GlxAlias)));
  GLX_HIP(hipMemcpyAsync(t->table, tab.data(), (size_t)U * sizeof(GlxAlias), hipMemcpyHostToDevice, s));
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

void free_table(glx_negative* t) {
  if (!t) return;
  if (t->ids) (void)hipFree(t->ids);
  if (t->table) (void)hipFree(t->table);
  delete t;
}

struct Owner {  // frees a half-built table on early returns
  glx_negative* t = nullptr;
  ~Owner() { free_table(t); }
};

}  // namespace

extern "C" int glx_negative_create(int device, int64_t num_ids, const int64_t* ids, const float* weights,
                                   int ptr_kind, void* stream, glx_negative** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(num_ids >= 0 && num_ids < INT32_MAX, "num_ids must be in [0, 2^31)");
  GLX_REQUIRE(num_ids == 0 || ids != nullptr, "ids is NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  hipStream_t s = glx_stream(stream);
  Owner own;
  own.t = new glx_negative{device, num_ids, nullptr, nullptr};
  if (num_ids > 0) {
    const hipMemcpyKind kind = ptr_kind == GLX_PTR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    GLX_HIP(hipMalloc(&own.t->ids, (size_t)num_ids * sizeof(int64_t)));
    GLX_HIP(hipMemcpyAsync(own.t->ids, ids, (size_t)num_ids * sizeof(int64_t), kind, s));
    if (weights) {
      GlxTemp w;
      GLX_HIP(hipMalloc(&w.p, (size_t)num_ids * sizeof(float)));
      GLX_HIP(hipMemcpyAsync(w.p, weights, (size_t)num_ids * sizeof(float), kind, s));
      rc = build_table(own.t, w.as<float>(), s);
      if (rc != GLX_OK) return rc;
    }
    GLX_HIP(hipStreamSynchronize(s));
  }
  *out = own.t;
  own.t = nullptr;
  return GLX_OK;
}

extern "C" int glx_negative_from_graph(const glx_graph* g, int by_in_degree, void* stream, glx_negative** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = glx_stream(stream);
  const int64_t E = g->num_edges;
  Owner own;
  own.t = new glx_negative{g->device, 0, nullptr, nullptr};
  if (E > 0) {
    const size_t n = (size_t)E;
    // (dst, edge id) sorted by dst; per distinct dst: smallest edge id (= first appearance,
    // edge ids are insertion indices) and run length (= in-degree)
    GlxTemp nbr, eid, nbr_s, eid_s, uniq, first, cnt, nruns;
    GLX_HIP(hipMalloc(&nbr.p, n * 8));
    GLX_HIP(hipMalloc(&eid.p, n * 8));
    GLX_HIP(hipMalloc(&nbr_s.p, n * 8));
    GLX_HIP(hipMalloc(&eid_s.p, n * 8));
    glx_neg_split_adj_kernel<<<grid_for(E), 256, 0, s>>>(g->adj, E, nbr.as<int64_t>(), eid.as<int64_t>());
#define SORT1(tmp, bytes)                                                                                     \
  rocprim::radix_sort_pairs(tmp, bytes, nbr.as<int64_t>(), nbr_s.as<int64_t>(), eid.as<int64_t>(), eid_s.as<int64_t>(), \
                            n, 0, 64, s)
    GLX_ROCPRIM(SORT1);
#undef SORT1
    GLX_HIP(hipMalloc(&uniq.p, n * 8));
    GLX_HIP(hipMalloc(&first.p, n * 8));
    GLX_HIP(hipMalloc(&cnt.p, n * 8));
    GLX_HIP(hipMalloc(&nruns.p, 8));
#define FIRST(tmp, bytes)                                                                                          \
  rocprim::reduce_by_key(tmp, bytes, nbr_s.as<int64_t>(), eid_s.as<int64_t>(), n, uniq.as<int64_t>(), first.as<int64_t>(), \
                         nruns.as<int64_t>(), MinOp(), rocprim::equal_to<int64_t>(), s)
    GLX_ROCPRIM(FIRST);
#undef FIRST
#define RLE(tmp, bytes)                                                                                 \
  rocprim::run_length_encode(tmp, bytes, nbr_s.as<int64_t>(), n, nbr.as<int64_t>(), cnt.as<int64_t>(), \
                             nruns.as<int64_t>(), s)
    GLX_ROCPRIM(RLE);  // nbr is scratch from here on
#undef RLE
    int64_t U = 0;
    GLX_HIP(hipMemcpyAsync(&U, nruns.p, 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    GLX_REQUIRE(U < INT32_MAX, "more than 2^31 distinct destination ids");
    own.t->num_ids = U;
    // order the distinct ids (and their counts) by first appearance
    GlxTemp key_out, cnt_o;
    GLX_HIP(hipMalloc(&own.t->ids, (size_t)U * 8));
    GLX_HIP(hipMalloc(&key_out.p, (size_t)U * 8));
    GLX_HIP(hipMalloc(&cnt_o.p, (size_t)U * 8));
#define SORT2(tmp, bytes)                                                                                   \
  rocprim::radix_sort_pairs(tmp, bytes, first.as<int64_t>(), key_out.as<int64_t>(), uniq.as<int64_t>(), own.t->ids, \
                            (size_t)U, 0, 64, s)
    GLX_ROCPRIM(SORT2);
#undef SORT2
    if (by_in_degree) {
#define SORT3(tmp, bytes)                                                                                        \
  rocprim::radix_sort_pairs(tmp, bytes, first.as<int64_t>(), key_out.as<int64_t>(), cnt.as<int64_t>(), cnt_o.as<int64_t>(), \
                            (size_t)U, 0, 64, s)
      GLX_ROCPRIM(SORT3);
#undef SORT3
      GlxTemp w;
      GLX_HIP(hipMalloc(&w.p, (size_t)U * 4));
      glx_neg_to_float_kernel<<<grid_for(U), 256, 0, s>>>(cnt_o.as<int64_t>(), U, w.as<float>());
      int rc = build_table(own.t, w.as<float>(), s);
      if (rc != GLX_OK) return rc;
    }
    GLX_HIP(hipStreamSynchronize(s));
  }
  *out = own.t;
  own.t = nullptr;
  return GLX_OK;
}

extern "C" void glx_negative_destroy(glx_negative* t) {
  if (!t) return;
  GlxDeviceGuard guard(t->device);
  free_table(t);
}

extern "C" int glx_negative_info(const glx_negative* t, int64_t* num_ids, int* weighted) {
  GLX_REQUIRE(t != nullptr, "table is NULL");
  if (num_ids) *num_ids = t->num_ids;
  if (weighted) *weighted = t->table != nullptr;
  return GLX_OK;
}

extern "C" int glx_negative_export(const glx_negative* t, int64_t* ids, float* prob, int32_t* alias, void* stream) {
  GLX_REQUIRE(t != nullptr, "table is NULL");
  GlxDeviceGuard guard(t->device);
  hipStream_t s = glx_host_call_stream(stream, t->device);
  const size_t U = (size_t)t->num_ids;
  if (U == 0) return GLX_OK;
  if (ids) GLX_HIP(hipMemcpyAsync(ids, t->ids, U * 8, hipMemcpyDeviceToHost, s));
  if (prob || alias) {
    GLX_REQUIRE(t->table != nullptr, "the table is uniform: it has no alias entries");
    std::vector<GlxAlias> tab(U);
    GLX_HIP(hipMemcpyAsync(tab.data(), t->table, U * sizeof(GlxAlias), hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < U; ++i) {
      if (prob) prob[i] = tab[i].prob;
      if (alias) alias[i] = tab[i].alias;
    }
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

__global__ void glx_neg_iota_u32_kernel(uint32_t* p, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (uint32_t)i;
}

extern "C" int glx_graph_enable_negative(glx_graph* g, void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  if (g->nbr_sorted) return GLX_OK;
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = glx_stream(stream);
  const int64_t E = g->num_edges, V = g->num_rows;
  int64_t* sorted = nullptr;
  uint32_t* slots = nullptr;
  GLX_HIP(hipMalloc(&sorted, (size_t)(E > 0 ? E : 1) * 8));
  GlxTemp own, own2;
  own.p = sorted;
  GLX_HIP(hipMalloc(&slots, (size_t)(E > 0 ? E : 1) * 4));
  own2.p = slots;
  if (E > 0) {
    GlxTemp nbr, eid, iota;
    GLX_REQUIRE(E < (int64_t)UINT32_MAX, "the id-sorted row index supports up to 2^32 - 1 edges per GPU");
    GLX_HIP(hipMalloc(&nbr.p, (size_t)E * 8));
    GLX_HIP(hipMalloc(&eid.p, (size_t)E * 8));
    GLX_HIP(hipMalloc(&iota.p, (size_t)E * 4));
    glx_neg_split_adj_kernel<<<grid_for(E), 256, 0, s>>>(g->adj, E, nbr.as<int64_t>(), eid.as<int64_t>());
    glx_neg_iota_u32_kernel<<<grid_for(E), 256, 0, s>>>(iota.as<uint32_t>(), E);
    // every row's neighbour ids in ascending order, each with the CSR slot it sits in: the membership test of
    // strict negative sampling is a binary search, an id == value filter finds its hits the same way
#define SEGSORT(tmp, bytes)                                                                                    \
  rocprim::segmented_radix_sort_pairs(tmp, bytes, nbr.as<int64_t>(), sorted, iota.as<uint32_t>(), slots,    \
                                      (unsigned int)E, (unsigned int)V, g->row_ptr, g->row_ptr + 1, 0, 64, s)
    GLX_ROCPRIM(SEGSORT);
#undef SEGSORT
  }
  own2.p = nullptr;
  g->slot_sorted = slots;
  own.p = nullptr;
  g->nbr_sorted = sorted;
  return GLX_OK;
}

// glx_dist.hip: strict in-degree sampling for the rows an owner received (device pointers; rng_rows = their indices in
// the requester's request).  Device already selected by the caller.
int glx_negative_sample_rows_device(const glx_negative* t, const glx_graph* g, const int64_t* src, const int64_t* rng_rows,
                                    int32_t batch, int32_t count, int64_t default_neighbor_id, uint64_t seed,
                                    uint64_t call_counter, int64_t* out, hipStream_t s) {
  GLX_REQUIRE(t != nullptr && g != nullptr, "NULL table / graph");
  GLX_REQUIRE(g->nbr_sorted != nullptr, "call glx_graph_enable_negative(graph) first");
  const int64_t total = (int64_t)batch * count;
  if (total == 0) return GLX_OK;
  if (t->num_ids == 0) {
    glx_neg_fill_kernel<<<grid_for(total), 256, 0, s>>>(out, total, default_neighbor_id);
  } else {
    NegArgs a{};
    a.ids = t->ids;
    a.table = t->table;
    a.num_ids = t->num_ids;
    a.src = src;
    a.batch = batch;
    a.count = count;
    a.seed = seed;
    a.cc = call_counter;
    a.out = out;
    a.map = g->map();
    a.row_ptr = g->row_ptr;
    a.nbr_sorted = g->nbr_sorted;
    a.rng_rows = rng_rows;
    launch_negative<GLX_NEG_EXCLUDE_NEIGHBORS>(a, s);
  }
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

extern "C" int glx_negative_sample(const glx_negative* t, int exclude, const glx_graph* g, const int64_t* src,
                                   int32_t batch, int32_t count, int64_t default_neighbor_id, uint64_t seed,
                                   uint64_t call_counter, int64_t* out, int ptr_kind, void* stream) {
  GLX_REQUIRE(t != nullptr, "table is NULL");
  GLX_REQUIRE(exclude >= GLX_NEG_EXCLUDE_NONE && exclude <= GLX_NEG_EXCLUDE_BATCH, "unknown exclusion mode %d", exclude);
  GLX_REQUIRE(batch >= 0 && count >= 0, "negative batch or count");
  GLX_REQUIRE((int64_t)batch * count <= INT32_MAX, "batch * count exceeds 2^31 - 1");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  const int64_t total = (int64_t)batch * count;
  if (total == 0) return GLX_OK;
  GLX_REQUIRE(out != nullptr, "out is NULL");
  GLX_REQUIRE(exclude == GLX_NEG_EXCLUDE_NONE || src != nullptr, "src is NULL");
  if (exclude == GLX_NEG_EXCLUDE_NEIGHBORS) {
    GLX_REQUIRE(g != nullptr, "strict in-degree sampling needs the edge type's graph");
    GLX_REQUIRE(g->nbr_sorted != nullptr, "call glx_graph_enable_negative(graph) first");
    GLX_REQUIRE(g->device == t->device, "graph and table live on different devices");
  }
  GlxDeviceGuard guard(t->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", t->device);
  const bool host = ptr_kind == GLX_PTR_HOST;
  hipStream_t s = host ? glx_host_call_stream(stream, t->device) : glx_stream(stream);
  // scratch: [out (host calls)] [src (host calls)] [sorted batch (node-weight)]
  const bool need_src = exclude != GLX_NEG_EXCLUDE_NONE;
  const size_t out_bytes = host ? (size_t)total * 8 : 0;
  const size_t src_bytes = host && need_src ? (size_t)batch * 8 : 0;
  const size_t sort_bytes = exclude == GLX_NEG_EXCLUDE_BATCH ? (size_t)batch * 8 + 8 : 0;  // + first_dropped
  char* scratch = nullptr;
  if (out_bytes + src_bytes + sort_bytes > 0) {
    int rc = glx_scratch_alloc(reinterpret_cast<void**>(&scratch), out_bytes + src_bytes + sort_bytes, s, 0);
    if (rc != GLX_OK) return rc;
  }
  struct Release {
    void* p;
    hipStream_t s;
    ~Release() {
      if (p) glx_scratch_free(p, s);
    }
  } release{scratch, s};
  int64_t* d_out = host ? reinterpret_cast<int64_t*>(scratch) : out;
  const int64_t* d_src = src;
  if (host && need_src) {
    int64_t* p = reinterpret_cast<int64_t*>(scratch + out_bytes);
    GLX_HIP(hipMemcpyAsync(p, src, src_bytes, hipMemcpyHostToDevice, s));
    d_src = p;
  }
  if (t->num_ids == 0) {
    // no candidates at all (random_negative_sampler.cc:50-54): the default neighbour id
    glx_neg_fill_kernel<<<grid_for(total), 256, 0, s>>>(d_out, total, default_neighbor_id);
  } else {
    NegArgs a{};
    a.ids = t->ids;
    a.table = t->table;
    a.num_ids = t->num_ids;
    a.src = d_src;
    a.batch = batch;
    a.count = count;
    a.seed = seed;
    a.cc = call_counter;
    a.out = d_out;
    if (exclude == GLX_NEG_EXCLUDE_NEIGHBORS) {
      a.map = g->map();
      a.row_ptr = g->row_ptr;
      a.nbr_sorted = g->nbr_sorted;
    } else if (exclude == GLX_NEG_EXCLUDE_BATCH) {
      // the request's own ids, ascending (per request: temp storage from the workspace cache)
      int64_t* sorted = reinterpret_cast<int64_t*>(scratch + out_bytes + src_bytes);
      size_t tmp_bytes = 0;
      GLX_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_src, sorted, (size_t)batch, 0, 64, s));
      void* tmp = nullptr;
      int rc = glx_scratch_alloc(&tmp, tmp_bytes ? tmp_bytes : 16, s, 1);
      if (rc != GLX_OK) return rc;
      hipError_t e = rocprim::radix_sort_keys(tmp, tmp_bytes, d_src, sorted, (size_t)batch, 0, 64, s);
      glx_scratch_free(tmp, s);
      GLX_HIP(e);
      a.batch_sorted = sorted;
      a.first_dropped = reinterpret_cast<unsigned long long*>(sorted + batch);
      glx_neg_fill_kernel<<<1, 64, 0, s>>>(reinterpret_cast<int64_t*>(a.first_dropped), 1, (int64_t)batch);
    }
    GlxKernelTimer timer(GLX_KERNEL_SAMPLE, s);
    switch (exclude) {
      case GLX_NEG_EXCLUDE_NONE: launch_negative<GLX_NEG_EXCLUDE_NONE>(a, s); break;
      case GLX_NEG_EXCLUDE_NEIGHBORS: launch_negative<GLX_NEG_EXCLUDE_NEIGHBORS>(a, s); break;
      default:
        launch_negative<GLX_NEG_EXCLUDE_BATCH>(a, s);
        glx_negative_after_drop_kernel<<<grid_for(total), 256, 0, s>>>(a);
        break;
    }
    timer.stop();
  }
  GLX_HIP(hipGetLastError());
  if (host) {
    GLX_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  return GLX_OK;
}
