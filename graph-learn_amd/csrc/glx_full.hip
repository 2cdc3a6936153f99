// glx: the further samplers of the registry.
//   * in-degree alias tables for InDegreeSampler (in_degree_sampler.cc:33-114,
//     GraphStorage::GetInDegree / TopoStatics::Add, topo_statics.cc:33-69);
//   * FullSampler (full_sampler.cc:28-97): ragged "all neighbours" response.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "glx_common.h"

namespace {

__global__ void glx_extract_nbr_kernel(const GlxAdj* __restrict__ adj, int64_t E, int64_t* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < E; i += stride) out[i] = adj[i].nbr;
}

// weight of slot i = float(in-degree of its neighbour id)
__global__ void glx_indeg_weight_kernel(const GlxAdj* __restrict__ adj, int64_t E, GlxIdMap uniq_map,
                                        const int64_t* __restrict__ counts, float* __restrict__ w) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < E; i += stride) {
    const int64_t r = glx_row_of(uniq_map, adj[i].nbr);
    w[i] = r < 0 ? 0.0f : (float)(int32_t)counts[r];  // static_cast<float>(GetInDegree(id)), :86-87
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b < 8192 ? b : 8192);
}

#define GLX_ROCPRIM(call)                                 \
  do {                                                    \
    size_t bytes__ = 0;                                   \
    GLX_HIP(call(nullptr, bytes__));                      \
    GlxTemp tmp__;                                        \
    GLX_HIP(hipMalloc(&tmp__.p, bytes__ ? bytes__ : 16)); \
    GLX_HIP(call(tmp__.p, bytes__));                      \
    GLX_HIP(hipStreamSynchronize(s));                     \
  } while (0)

// ---- FullSampler ------------------------------------------------------------
__global__ void glx_full_sizes_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr,
                                      const int64_t* __restrict__ src, int32_t batch, int32_t max_limit,
                                      int32_t* __restrict__ degrees, int64_t* __restrict__ deg64) {
  int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const int64_t row = glx_row_of(map, src[i]);
  int64_t deg = row < 0 ? 0 : row_ptr[row + 1] - row_ptr[row];
  // GetTruncatedSize, full_sampler.cc:89-96
  if (max_limit > 0 && max_limit < deg) deg = max_limit;
  degrees[i] = (int32_t)deg;
  deg64[i] = deg;
}

// One wave per request row; lanes stride over the row's (nbr, eid) slots.
__global__ __launch_bounds__(256) void glx_full_copy_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr,
                                                            const GlxAdj* __restrict__ adj,
                                                            const int64_t* __restrict__ src, int32_t batch,
                                                            const int64_t* __restrict__ offsets,
                                                            int64_t* __restrict__ nbr_out,
                                                            int64_t* __restrict__ eid_out) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= batch) return;
  const int64_t row = glx_row_of(map, src[i]);
  if (row < 0) return;
  const int64_t start = row_ptr[row];
  const int64_t o0 = offsets[i];
  const int64_t take = offsets[i + 1] - o0;
  for (int64_t j = lane; j < take; j += 64) {
    const GlxAdj r = adj[start + j];
    nbr_out[o0 + j] = r.nbr;
    eid_out[o0 + j] = r.eid;
  }
}

__global__ void glx_set_i64_kernel(int64_t* p, int64_t v) { *p = v; }

int full_sizes_device(const glx_graph* g, const int64_t* d_src, int32_t batch, int32_t max_limit,
                      int32_t* d_degrees, int64_t* d_offsets, hipStream_t s) {
  // d_offsets[0..batch) temporarily holds the int64 degrees, then is scanned in place.
  glx_full_sizes_kernel<<<(unsigned)((batch + 255) / 256), 256, 0, s>>>(g->map(), g->row_ptr, d_src, batch,
                                                                       max_limit, d_degrees, d_offsets + 1);
  glx_set_i64_kernel<<<1, 1, 0, s>>>(d_offsets, 0);
#define SCAN(tmp, bytes) \
  rocprim::inclusive_scan(tmp, bytes, d_offsets + 1, d_offsets + 1, (size_t)batch, rocprim::plus<int64_t>(), s)
  GLX_ROCPRIM(SCAN);
#undef SCAN
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

}  // namespace

// Installs the InDegreeSampler tables from (destination id, in-degree) pairs: uniq[U] distinct ids
// (device), counts[U] their in-degrees -- this shard's own (glx_graph_enable_in_degree) or summed over
// all shards (glx_dist_enable_in_degree).  Replaces tables installed earlier.
int glx_graph_install_in_degree(glx_graph* g, const int64_t* d_uniq, const int64_t* d_counts, int64_t U,
                                hipStream_t s) {
  const int64_t V = g->num_rows, E = g->num_edges;
  GLX_REQUIRE(U < INT32_MAX, "more than 2^31 distinct neighbour ids");
  GlxAlias* table = nullptr;
  GLX_HIP(hipMalloc(&table, (size_t)(E > 0 ? E : 1) * sizeof(GlxAlias)));
  GlxTemp own;  // frees the table on early returns
  own.p = table;
  GlxIdMapStorage um;
  int64_t* keep = nullptr;
  if (E > 0) {
    GlxTemp w;
    int rc = glx_idmap_build(d_uniq, U, &um, s);
    if (rc != GLX_OK) return rc;
    hipError_t e = hipMalloc(&w.p, (size_t)E * 4);
    if (e == hipSuccess) {
      glx_indeg_weight_kernel<<<grid_for(E), 256, 0, s>>>(g->adj, E, GlxIdMap{um.keys, um.vals, um.cap - 1, U}, d_counts,
                                                         w.as<float>());
      rc = glx_alias_build_launch(g->row_ptr, w.as<float>(), V, E, table, s);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess || rc != GLX_OK) glx_idmap_free(&um);
    GLX_HIP(e);
    GLX_HIP(e2);
    if (rc != GLX_OK) return rc;
    // keep destination id -> in-degree for glx_graph_in_degrees (GetInDegree, topo_statics.cc:62-69)
    hipError_t e3 = hipMalloc(&keep, (size_t)(U > 0 ? U : 1) * 8);
    if (e3 == hipSuccess) e3 = hipMemcpyAsync(keep, d_counts, (size_t)U * 8, hipMemcpyDeviceToDevice, s);
    if (e3 == hipSuccess) e3 = hipStreamSynchronize(s);
    if (e3 != hipSuccess) {
      if (keep) (void)hipFree(keep);
      glx_idmap_free(&um);
      GLX_HIP(e3);
    }
  }
  // swap in (a handle is immutable while it is shared; this runs right after creation)
  if (g->alias_indeg) (void)hipFree(g->alias_indeg);
  if (g->dst_count) (void)hipFree(g->dst_count);
  glx_idmap_free(&g->dst_map);
  own.p = nullptr;
  g->alias_indeg = table;
  g->dst_map = um;
  g->dst_count = keep;
  g->num_dst = E > 0 ? U : 0;
  return GLX_OK;
}

// This shard's distinct destination ids (ascending) and how often each occurs.
int glx_graph_dst_counts(const glx_graph* g, GlxTemp* uniq, GlxTemp* counts, int64_t* U, hipStream_t s) {
  const int64_t E = g->num_edges;
  *U = 0;
  const size_t n = (size_t)(E > 0 ? E : 1);
  GlxTemp keys, sorted, nruns;
  GLX_HIP(hipMalloc(&keys.p, n * 8));
  GLX_HIP(hipMalloc(&sorted.p, n * 8));
  GLX_HIP(hipMalloc(&uniq->p, n * 8));
  GLX_HIP(hipMalloc(&counts->p, n * 8));
  GLX_HIP(hipMalloc(&nruns.p, 8));
  if (E == 0) return GLX_OK;
  glx_extract_nbr_kernel<<<grid_for(E), 256, 0, s>>>(g->adj, E, keys.as<int64_t>());
#define SORT(tmp, bytes) rocprim::radix_sort_keys(tmp, bytes, keys.as<int64_t>(), sorted.as<int64_t>(), n, 0, 64, s)
  GLX_ROCPRIM(SORT);
#undef SORT
#define RLE(tmp, bytes)                                                                                    \
  rocprim::run_length_encode(tmp, bytes, sorted.as<int64_t>(), n, uniq->as<int64_t>(), counts->as<int64_t>(), \
                             nruns.as<int64_t>(), s)
  GLX_ROCPRIM(RLE);
#undef RLE
  GLX_HIP(hipMemcpyAsync(U, nruns.p, 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

extern "C" int glx_graph_enable_in_degree(glx_graph* g, void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  if (g->alias_indeg) return GLX_OK;
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = glx_stream(stream);
  GlxTemp uniq, counts;
  int64_t U = 0;
  int rc = glx_graph_dst_counts(g, &uniq, &counts, &U, s);
  if (rc != GLX_OK) return rc;
  return glx_graph_install_in_degree(g, uniq.as<int64_t>(), counts.as<int64_t>(), U, s);
}

namespace {
__global__ void glx_in_degrees_kernel(GlxIdMap map, const int64_t* __restrict__ count, const int64_t* __restrict__ ids,
                                      int64_t n, int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t at = glx_row_of(map, ids[i]);
  out[i] = at < 0 ? 0 : count[at];
}
}  // namespace

extern "C" int glx_graph_in_degrees(const glx_graph* g, const int64_t* ids, int64_t n, int64_t* deg_out, int ptr_kind,
                                    void* stream) {
  GLX_REQUIRE(g && (n == 0 || (ids && deg_out)), "NULL argument");
  GLX_REQUIRE(n >= 0, "negative n");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE(g->alias_indeg != nullptr, "call glx_graph_enable_in_degree(graph) first");
  if (n == 0) return GLX_OK;
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  const bool host = ptr_kind == GLX_PTR_HOST;
  hipStream_t s = host ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  const int64_t* d_ids = ids;
  int64_t* d_out = deg_out;
  int64_t* scratch = nullptr;
  if (host) {
    int rc = glx_scratch_alloc(reinterpret_cast<void**>(&scratch), (size_t)n * 16, s, 0);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemcpyAsync(scratch, ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    d_ids = scratch;
    d_out = scratch + n;
  }
  if (g->num_dst == 0) {
    GLX_HIP(hipMemsetAsync(d_out, 0, (size_t)n * 8, s));
  } else {
    const GlxIdMap map{g->dst_map.keys, g->dst_map.vals, g->dst_map.cap - 1, g->num_dst};
    glx_in_degrees_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(map, g->dst_count, d_ids, n, d_out);
  }
  hipError_t e = hipGetLastError();
  if (host && e == hipSuccess) e = hipMemcpyAsync(deg_out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s);
  if (host && e == hipSuccess) e = hipStreamSynchronize(s);
  if (scratch) glx_scratch_free(scratch, s);
  GLX_HIP(e);
  return GLX_OK;
}

extern "C" int glx_sample_full_sizes(const glx_graph* g, const int64_t* src, int32_t batch,
                                     int32_t max_limit, int32_t* degrees_out, int64_t* offsets_out,
                                     int ptr_kind, void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(batch >= 0, "negative batch");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE(offsets_out != nullptr && (batch == 0 || (src && degrees_out)), "NULL data pointer");
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  if (batch == 0) {
    if (ptr_kind == GLX_PTR_HOST) offsets_out[0] = 0;
    else glx_set_i64_kernel<<<1, 1, 0, s>>>(offsets_out, 0);
    return GLX_OK;
  }
  if (ptr_kind == GLX_PTR_DEVICE) return full_sizes_device(g, src, batch, max_limit, degrees_out, offsets_out, s);
  char* d = nullptr;
  const size_t bytes = (size_t)batch * 8 + ((size_t)batch + 1) * 8 + (size_t)batch * 4;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), bytes, s, 0);
  if (rc != GLX_OK) return rc;
  int64_t* d_src = reinterpret_cast<int64_t*>(d);
  int64_t* d_off = d_src + batch;
  int32_t* d_deg = reinterpret_cast<int32_t*>(d_off + batch + 1);
  GLX_HIP(hipMemcpyAsync(d_src, src, (size_t)batch * 8, hipMemcpyHostToDevice, s));
  rc = full_sizes_device(g, d_src, batch, max_limit, d_deg, d_off, s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipMemcpyAsync(degrees_out, d_deg, (size_t)batch * 4, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipMemcpyAsync(offsets_out, d_off, ((size_t)batch + 1) * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

extern "C" int glx_sample_full(const glx_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                               const int64_t* offsets, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                               void* stream) {
  (void)max_limit;  // already folded into `offsets` by glx_sample_full_sizes
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(batch >= 0, "negative batch");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  if (batch == 0) return GLX_OK;
  GLX_REQUIRE(src && offsets, "NULL data pointer");
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  const unsigned blocks = (unsigned)(((int64_t)batch * 64 + 255) / 256);
  if (ptr_kind == GLX_PTR_DEVICE) {
    if (!nbr_out || !eid_out) {
      // an EMPTY response has no buffer to point at (an empty tensor's data pointer is NULL): fine when the offsets
      // agree -- they live on the device, so this one case costs a host wait
      int64_t total = -1;
      GLX_HIP(hipMemcpyAsync(&total, offsets + batch, 8, hipMemcpyDeviceToHost, s));
      GLX_HIP(hipStreamSynchronize(s));
      GLX_REQUIRE(total == 0, "NULL output pointer for a response of %lld values", (long long)total);
      return GLX_OK;
    }
    glx_full_copy_kernel<<<blocks, 256, 0, s>>>(g->map(), g->row_ptr, g->adj, src, batch, offsets, nbr_out,
                                                eid_out);
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  const int64_t total = offsets[batch];
  GLX_REQUIRE(total >= 0 && total <= INT32_MAX, "response exceeds int32 values (tensor.h:47)");
  if (total == 0) return GLX_OK;
  GLX_REQUIRE(nbr_out && eid_out, "NULL output pointer");
  int64_t* d = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), ((size_t)batch * 2 + 1 + (size_t)total * 2) * 8, s, 0);
  if (rc != GLX_OK) return rc;
  int64_t* d_src = d;
  int64_t* d_off = d + batch;
  int64_t* d_nbr = d_off + batch + 1;
  int64_t* d_eid = d_nbr + total;
  GLX_HIP(hipMemcpyAsync(d_src, src, (size_t)batch * 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipMemcpyAsync(d_off, offsets, ((size_t)batch + 1) * 8, hipMemcpyHostToDevice, s));
  glx_full_copy_kernel<<<blocks, 256, 0, s>>>(g->map(), g->row_ptr, g->adj, d_src, batch, d_off, d_nbr, d_eid);
  GLX_HIP(hipMemcpyAsync(nbr_out, d_nbr, (size_t)total * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipMemcpyAsync(eid_out, d_eid, (size_t)total * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}
