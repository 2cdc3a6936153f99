// glx device-side storage build: raw edge list -> CSR, entirely on the GPU.
// Replaces the host path LocalGraph::UpdateEdges -> MemoryGraphStorage::Add
// (local_graph.cc:50-64), MemoryAdjMatrix::Build/Sort (memory_adj_matrix.cc:60-66,
// 105-125) and CompressedMemoryAdjMatrix::Build (:169-189).
//
// Pipeline (all streaming / radix passes, HBM-bound):
//   1. [weighted + sort] stable radix sort of edge indices by weight, descending
//   2. stable radix sort by source id  -> order = (src, weight desc, insertion)
//   3. run-length encode the sorted sources -> row ids + degrees
//   4. exclusive scan of the degrees -> row_ptr
//   5. gather dst / edge id / weight through the permutation into 16-byte slots
//   6. alias tables + id map (glx_graph_finalize)
#include <string.h>

#include <new>

#include <rocprim/rocprim.hpp>

#include "glx_common.h"

namespace {

__global__ void glx_iota_kernel(int64_t* p, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = i;
}

template <typename T>
__global__ void glx_gather_kernel(const T* __restrict__ in, const int64_t* __restrict__ perm, int64_t n,
                                  T* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[perm[i]];
}

__global__ void glx_gather_adj_kernel(const int64_t* __restrict__ dst, const int64_t* __restrict__ eids,
                                      const int64_t* __restrict__ perm, int64_t n,
                                      GlxAdj* __restrict__ adj) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t e = perm[i];
    // edge id = insertion index (memory_edge_storage.cc:53-57) unless the caller
    // supplies global ids for a shard's subset of the edges
    adj[i] = GlxAdj{dst[e], eids ? eids[e] : e};
  }
}

__global__ void glx_set_last_kernel(int64_t* row_ptr, int64_t V, int64_t E) { row_ptr[V] = E; }

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b < 8192 ? b : 8192);
}

#define GLX_ROCPRIM(call)                                                      \
  do {                                                                         \
    size_t bytes__ = 0;                                                        \
    GLX_HIP(call(nullptr, bytes__));                                           \
    GlxTemp tmp__;                                                             \
    GLX_HIP(hipMalloc(&tmp__.p, bytes__ ? bytes__ : 16));                      \
    GLX_HIP(call(tmp__.p, bytes__));                                           \
    GLX_HIP(hipStreamSynchronize(s)); /* tmp__ is freed right after */         \
  } while (0)

int build_impl(glx_graph* g, const int64_t* src, const int64_t* dst, const float* weight,
               const int64_t* edge_ids, const int64_t* timestamp, int order, int ptr_kind, hipStream_t s) {
  const int64_t E = g->num_edges;
  const hipMemcpyKind kind = ptr_kind == GLX_PTR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if (E == 0) {
    g->num_rows = 0;
    GLX_HIP(hipMalloc(&g->row_ptr, sizeof(int64_t)));
    GLX_HIP(hipMemsetAsync(g->row_ptr, 0, sizeof(int64_t), s));
    GLX_HIP(hipMalloc(&g->adj, sizeof(GlxAdj)));
    if (weight) GLX_HIP(hipMalloc(&g->weight, sizeof(float)));
    GlxTemp dummy_ids;
    GLX_HIP(hipMalloc(&dummy_ids.p, sizeof(int64_t)));
    return glx_graph_finalize(g, dummy_ids.as<int64_t>(), s);
  }
  // stage the edge list on the device
  GlxTemp h_src, h_dst, h_w, h_eid, h_ts;
  const int64_t* d_ts = timestamp;
  const int64_t* d_eid = edge_ids;
  const int64_t* d_src = src;
  const int64_t* d_dst = dst;
  const float* d_w = weight;
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMalloc(&h_src.p, (size_t)E * 8));
    GLX_HIP(hipMalloc(&h_dst.p, (size_t)E * 8));
    GLX_HIP(hipMemcpyAsync(h_src.p, src, (size_t)E * 8, kind, s));
    GLX_HIP(hipMemcpyAsync(h_dst.p, dst, (size_t)E * 8, kind, s));
    d_src = h_src.as<int64_t>();
    d_dst = h_dst.as<int64_t>();
    if (edge_ids) {
      GLX_HIP(hipMalloc(&h_eid.p, (size_t)E * 8));
      GLX_HIP(hipMemcpyAsync(h_eid.p, edge_ids, (size_t)E * 8, kind, s));
      d_eid = h_eid.as<int64_t>();
    }
    if (weight) {
      GLX_HIP(hipMalloc(&h_w.p, (size_t)E * 4));
      GLX_HIP(hipMemcpyAsync(h_w.p, weight, (size_t)E * 4, kind, s));
      d_w = h_w.as<float>();
    }
    if (timestamp) {
      GLX_HIP(hipMalloc(&h_ts.p, (size_t)E * 8));
      GLX_HIP(hipMemcpyAsync(h_ts.p, timestamp, (size_t)E * 8, kind, s));
      d_ts = h_ts.as<int64_t>();
    }
  }
  GlxTemp perm_a, perm_b, keys_a, keys_b;
  GLX_HIP(hipMalloc(&perm_a.p, (size_t)E * 8));
  GLX_HIP(hipMalloc(&perm_b.p, (size_t)E * 8));
  int64_t* pa = perm_a.as<int64_t>();
  int64_t* pb = perm_b.as<int64_t>();
  glx_iota_kernel<<<grid_for(E), 256, 0, s>>>(pa, E);
  const size_t n = (size_t)E;
  if (timestamp && order == GLX_ORDER_TIMESTAMP_ASC) {
    // pass 1 for timestamped types: timestamp ascending, stable (MemoryAdjMatrix::SortByTimestamp,
    // memory_adj_matrix.cc:129-148; it takes precedence over the weight order, :60-66)
    GlxTemp tk;
    GLX_HIP(hipMalloc(&tk.p, (size_t)E * 8));
    int64_t* ts_sorted = tk.as<int64_t>();
#define SORT_T(tmp, bytes) rocprim::radix_sort_pairs(tmp, bytes, d_ts, ts_sorted, pa, pb, n, 0, 64, s)
    GLX_ROCPRIM(SORT_T);
#undef SORT_T
    int64_t* t = pa; pa = pb; pb = t;
  } else if (weight && order == GLX_ORDER_WEIGHT_DESC) {
    // pass 1: weight descending, stable (ties keep insertion order)
    GlxTemp wk;
    GLX_HIP(hipMalloc(&wk.p, (size_t)E * 4));
    float* w_sorted = wk.as<float>();
#define SORT_W(tmp, bytes) \
  rocprim::radix_sort_pairs_desc(tmp, bytes, d_w, w_sorted, pa, pb, n, 0, 32, s)
    GLX_ROCPRIM(SORT_W);
#undef SORT_W
    int64_t* t = pa; pa = pb; pb = t;
  }
  // pass 2: source id ascending, stable
  GLX_HIP(hipMalloc(&keys_a.p, (size_t)E * 8));
  GLX_HIP(hipMalloc(&keys_b.p, (size_t)E * 8));
  int64_t* ka = keys_a.as<int64_t>();
  int64_t* kb = keys_b.as<int64_t>();
  glx_gather_kernel<int64_t><<<grid_for(E), 256, 0, s>>>(d_src, pa, E, ka);
#define SORT_S(tmp, bytes) rocprim::radix_sort_pairs(tmp, bytes, ka, kb, pa, pb, n, 0, 64, s)
  GLX_ROCPRIM(SORT_S);
#undef SORT_S
  const int64_t* sorted_src = kb;
  const int64_t* perm = pb;
  // rows = runs of equal source id
  GlxTemp uniq, counts, nruns;
  GLX_HIP(hipMalloc(&uniq.p, (size_t)E * 8));
  GLX_HIP(hipMalloc(&counts.p, (size_t)E * 8));
  GLX_HIP(hipMalloc(&nruns.p, sizeof(int64_t)));
#define RLE(tmp, bytes)                                                                        \
  rocprim::run_length_encode(tmp, bytes, sorted_src, n, uniq.as<int64_t>(), counts.as<int64_t>(), \
                             nruns.as<int64_t>(), s)
  GLX_ROCPRIM(RLE);
#undef RLE
  int64_t V = 0;
  GLX_HIP(hipMemcpyAsync(&V, nruns.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  GLX_REQUIRE(V < INT32_MAX, "more than 2^31 distinct source ids");
  g->num_rows = V;
  GLX_HIP(hipMalloc(&g->row_ptr, (size_t)(V + 1) * 8));
#define SCAN(tmp, bytes)                                                                          \
  rocprim::exclusive_scan(tmp, bytes, counts.as<int64_t>(), g->row_ptr, (int64_t)0, (size_t)V,   \
                          rocprim::plus<int64_t>(), s)
  GLX_ROCPRIM(SCAN);
#undef SCAN
  glx_set_last_kernel<<<1, 1, 0, s>>>(g->row_ptr, V, E);
  GLX_HIP(hipMalloc(&g->adj, (size_t)E * sizeof(GlxAdj)));
  glx_gather_adj_kernel<<<grid_for(E), 256, 0, s>>>(d_dst, d_eid, perm, E, g->adj);
  if (weight) {
    GLX_HIP(hipMalloc(&g->weight, (size_t)E * 4));
    glx_gather_kernel<float><<<grid_for(E), 256, 0, s>>>(d_w, perm, E, g->weight);
  }
  if (timestamp) {
    // GetEdgeTimestamp(edge id) of every slot, for timestamp filters (filter.cc:136-141)
    GLX_HIP(hipMalloc(&g->ts, (size_t)E * 8));
    glx_gather_kernel<int64_t><<<grid_for(E), 256, 0, s>>>(d_ts, perm, E, g->ts);
  }
  GLX_HIP(hipGetLastError());
  return glx_graph_finalize(g, uniq.as<int64_t>(), s);
}

}  // namespace

extern "C" int glx_graph_build(int device, int64_t num_edges, const int64_t* src, const int64_t* dst,
                               const float* weight, const int64_t* edge_ids, int sort_by_weight,
                               int ptr_kind, void* stream, glx_graph** out) {
  return glx_graph_build_ordered(device, num_edges, src, dst, weight, edge_ids, nullptr,
                                 sort_by_weight ? GLX_ORDER_WEIGHT_DESC : GLX_ORDER_INSERTION, ptr_kind, stream, out);
}

extern "C" int glx_graph_build_ordered(int device, int64_t num_edges, const int64_t* src, const int64_t* dst,
                                       const float* weight, const int64_t* edge_ids, const int64_t* timestamp,
                                       int order, int ptr_kind, void* stream, glx_graph** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  GLX_REQUIRE(order >= GLX_ORDER_INSERTION && order <= GLX_ORDER_TIMESTAMP_ASC, "unknown row order %d", order);
  GLX_REQUIRE(order != GLX_ORDER_TIMESTAMP_ASC || num_edges == 0 || timestamp != nullptr,
              "GLX_ORDER_TIMESTAMP_ASC needs timestamps");
  *out = nullptr;
  GLX_REQUIRE(num_edges >= 0, "negative num_edges");
  GLX_REQUIRE(num_edges == 0 || (src && dst), "src/dst must not be NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  glx_graph* g = new (std::nothrow) glx_graph();
  GLX_REQUIRE(g != nullptr, "out of host memory");
  memset(static_cast<void*>(g), 0, sizeof(*g));
  g->device = device;
  g->num_edges = num_edges;
  rc = build_impl(g, src, dst, weight, edge_ids, timestamp, order, ptr_kind, glx_stream(stream));
  if (rc != GLX_OK) {
    glx_graph_free(g);
    return rc;
  }
  *out = g;
  return GLX_OK;
}
