// glx induced sub-graph: replaces SubGraphSampler::InduceSubGraph
// (graphlearn/src/core/operator/subgraph/subgraph_sampler.cc:34-95).
//
// The reference takes the node list of a sub-graph request (seeds, then the sorted set of their sampled
// neighbours), asks FullSampler for every node's first DefaultFullNbrNum neighbours, and for every node i
//   * builds a map  neighbour id -> edge id  over row i (a later slot with the same neighbour id overwrites
//     an earlier one, :60-64), then
//   * walks ALL nodes j in list order and, where nodes[j] is a key of that map, appends the edges
//     (i, j, eid) and (j, i, eid) (:66-70).
// Here the rows are FullSampler's device-resident response (offsets + nbr + eid: glx_sample_full or, across
// shards, glx_dist_sample_full), and the two loops become
//   1. per-row open-addressing tables in one arena (row i gets next_pow2(2 deg_i) slots): key = neighbour id,
//      value = the LARGEST slot index holding it (atomicMax = "the later slot overwrites");
//   2. one wave per row i probing its table with nodes[j], 64 j's at a time; a ballot prefix keeps the
//      reference's j order, so the output is entry-for-entry the reference's.
// N^2 probes of one or two 16-byte reads each: integer work bound by the table reads (L2-resident at the
// reference's sizes: its response tensors are N^2 entries themselves).
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "glx_common.h"

namespace {

struct SubTab {
  int64_t key;
  int64_t slot;  // row-local index of the last entry with this key
};

__device__ __forceinline__ int64_t pow2_slots(int64_t deg) {
  int64_t c = 2;
  while (c < 2 * deg) c <<= 1;
  return c;
}

__global__ void glx_sub_caps_kernel(const int64_t* __restrict__ offsets, int32_t n, int64_t* __restrict__ caps) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  caps[i] = pow2_slots(offsets[i + 1] - offsets[i]);
}

__global__ void glx_sub_clear_kernel(SubTab* __restrict__ tab, int64_t total) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += step) tab[t] = SubTab{GLX_EMPTY_KEY, -1};
}

// One wave per row: lanes insert the row's entries.
__global__ __launch_bounds__(256) void glx_sub_build_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ nbr,
                                                            int32_t n, const int64_t* __restrict__ tab_off,
                                                            SubTab* __restrict__ tab) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t o0 = offsets[i], deg = offsets[i + 1] - o0;
  SubTab* t = tab + tab_off[i];
  const uint64_t mask = (uint64_t)pow2_slots(deg) - 1;
  for (int64_t k = lane; k < deg; k += 64) {
    const int64_t v = nbr[o0 + k];
    if (v == GLX_EMPTY_KEY) continue;
    uint64_t h = glx_mix64((uint64_t)v) & mask;
    while (true) {
      const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&t[h].key),
                                                (unsigned long long)GLX_EMPTY_KEY, (unsigned long long)v);
      if ((int64_t)prev == GLX_EMPTY_KEY || (int64_t)prev == v) {
        atomicMax(reinterpret_cast<long long*>(&t[h].slot), (long long)k);
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

__device__ __forceinline__ int64_t sub_probe(const SubTab* t, uint64_t mask, int64_t v) {
  if (v == GLX_EMPTY_KEY) return -1;
  uint64_t h = glx_mix64((uint64_t)v) & mask;
  while (true) {
    const SubTab e = t[h];
    if (e.key == v) return e.slot;
    if (e.key == GLX_EMPTY_KEY) return -1;
    h = (h + 1) & mask;
  }
}

// FILL = false: counts[i] = matches of row i.  FILL = true: writes row i's edges at out_off[i] (2 entries per match).
template <bool FILL>
__global__ __launch_bounds__(256) void glx_sub_match_kernel(const int64_t* __restrict__ nodes, int32_t n,
                                                            const int64_t* __restrict__ offsets, const int64_t* __restrict__ eid,
                                                            const int64_t* __restrict__ tab_off, const SubTab* __restrict__ tab,
                                                            int64_t* __restrict__ counts, const int64_t* __restrict__ out_off,
                                                            int32_t* __restrict__ row_out, int32_t* __restrict__ col_out,
                                                            int64_t* __restrict__ eid_out, int64_t capacity) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t o0 = offsets[i], deg = offsets[i + 1] - o0;
  const SubTab* t = tab + tab_off[i];
  const uint64_t mask = (uint64_t)pow2_slots(deg) - 1;
  int64_t done = 0;
  const int64_t base = FILL ? out_off[i] : 0;
  for (int32_t j0 = 0; j0 < n; j0 += 64) {
    const int32_t j = j0 + lane;
    int64_t slot = -1;
    if (j < n && deg > 0) slot = sub_probe(t, mask, nodes[j]);
    const uint64_t hit = __ballot(slot >= 0);
    if (FILL && slot >= 0) {
      const int64_t pos = base + 2 * (done + __popcll(hit & ((1ull << lane) - 1ull)));
      if (pos + 1 < capacity) {
        const int64_t e = eid[o0 + slot];
        row_out[pos] = (int32_t)i;  // AppendEdge(i, j, eid)
        col_out[pos] = j;
        eid_out[pos] = e;
        row_out[pos + 1] = j;       // AppendEdge(j, i, eid)
        col_out[pos + 1] = (int32_t)i;
        eid_out[pos + 1] = e;
      }
    }
    done += __popcll(hit);
  }
  if (!FILL && lane == 0) counts[i] = 2 * done;
}

#define GLX_ROCPRIM_SUB(call)                              \
  do {                                                     \
    size_t bytes__ = 0;                                    \
    GLX_HIP(call(nullptr, bytes__));                       \
    GlxTemp tmp__;                                         \
    GLX_HIP(hipMalloc(&tmp__.p, bytes__ ? bytes__ : 16));  \
    GLX_HIP(call(tmp__.p, bytes__));                       \
    GLX_HIP(hipStreamSynchronize(s));                      \
  } while (0)

int induce_device(const int64_t* d_nodes, int32_t n, const int64_t* d_off, const int64_t* d_nbr, const int64_t* d_eid,
                  int32_t* d_row, int32_t* d_col, int64_t* d_eout, int64_t capacity, int64_t* count_out, hipStream_t s) {
  GlxTemp caps, tab_off, counts, out_off, tab;
  GLX_HIP(hipMalloc(&caps.p, ((size_t)n + 1) * 8));
  GLX_HIP(hipMalloc(&tab_off.p, ((size_t)n + 1) * 8));
  GLX_HIP(hipMalloc(&counts.p, ((size_t)n + 1) * 8));
  GLX_HIP(hipMalloc(&out_off.p, ((size_t)n + 1) * 8));
  GLX_HIP(hipMemsetAsync(caps.p, 0, ((size_t)n + 1) * 8, s));
  GLX_HIP(hipMemsetAsync(counts.p, 0, ((size_t)n + 1) * 8, s));
  const unsigned g1 = (unsigned)((n + 255) / 256);
  const unsigned gw = (unsigned)(((int64_t)n * 64 + 255) / 256);
  glx_sub_caps_kernel<<<g1, 256, 0, s>>>(d_off, n, caps.as<int64_t>());
#define SCAN_C(tmp, bytes) rocprim::exclusive_scan(tmp, bytes, caps.as<int64_t>(), tab_off.as<int64_t>(), (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), s)
  GLX_ROCPRIM_SUB(SCAN_C);
#undef SCAN_C
  int64_t total_slots = 0;
  GLX_HIP(hipMemcpyAsync(&total_slots, tab_off.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  GLX_HIP(hipMalloc(&tab.p, (size_t)(total_slots > 0 ? total_slots : 1) * sizeof(SubTab)));
  int64_t cb = (total_slots + 255) / 256;
  glx_sub_clear_kernel<<<(unsigned)(cb < 1 ? 1 : (cb > 8192 ? 8192 : cb)), 256, 0, s>>>(tab.as<SubTab>(), total_slots);
  glx_sub_build_kernel<<<gw, 256, 0, s>>>(d_off, d_nbr, n, tab_off.as<int64_t>(), tab.as<SubTab>());
  glx_sub_match_kernel<false><<<gw, 256, 0, s>>>(d_nodes, n, d_off, d_eid, tab_off.as<int64_t>(), tab.as<SubTab>(),
                                                 counts.as<int64_t>(), nullptr, nullptr, nullptr, nullptr, 0);
#define SCAN_O(tmp, bytes) rocprim::exclusive_scan(tmp, bytes, counts.as<int64_t>(), out_off.as<int64_t>(), (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), s)
  GLX_ROCPRIM_SUB(SCAN_O);
#undef SCAN_O
  int64_t total = 0;
  GLX_HIP(hipMemcpyAsync(&total, out_off.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  *count_out = total;
  if (capacity > 0 && total > 0) {
    glx_sub_match_kernel<true><<<gw, 256, 0, s>>>(d_nodes, n, d_off, d_eid, tab_off.as<int64_t>(), tab.as<SubTab>(), nullptr,
                                                  out_off.as<int64_t>(), d_row, d_col, d_eout, capacity);
  }
  GLX_HIP(hipGetLastError());
  GLX_HIP(hipStreamSynchronize(s));  // the temporaries above are freed on return
  return GLX_OK;
}

}  // namespace

extern "C" int glx_subgraph_induce(int device, const int64_t* nodes, int32_t n, const int64_t* offsets, const int64_t* nbr,
                                   const int64_t* eid, int32_t* row_out, int32_t* col_out, int64_t* eid_out, int64_t capacity,
                                   int64_t* count_out, int ptr_kind, void* stream) {
  GLX_REQUIRE(count_out != nullptr, "count_out is NULL");
  *count_out = 0;
  GLX_REQUIRE(n >= 0 && capacity >= 0, "negative sizes");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  if (n == 0) return GLX_OK;
  GLX_REQUIRE(nodes && offsets, "NULL data pointer");
  GLX_REQUIRE(capacity == 0 || (row_out && col_out && eid_out), "NULL output with capacity > 0");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, device) : glx_stream(stream);
  if (ptr_kind == GLX_PTR_DEVICE) {
    return induce_device(nodes, n, offsets, nbr, eid, row_out, col_out, eid_out, capacity, count_out, s);
  }
  const int64_t total_in = offsets[n];
  GLX_REQUIRE(total_in >= 0 && (total_in == 0 || (nbr && eid)), "bad offsets / NULL rows");
  GlxTemp d_nodes, d_off, d_nbr, d_eid, d_row, d_col, d_eout;
  GLX_HIP(hipMalloc(&d_nodes.p, (size_t)n * 8));
  GLX_HIP(hipMalloc(&d_off.p, ((size_t)n + 1) * 8));
  GLX_HIP(hipMalloc(&d_nbr.p, (size_t)(total_in > 0 ? total_in : 1) * 8));
  GLX_HIP(hipMalloc(&d_eid.p, (size_t)(total_in > 0 ? total_in : 1) * 8));
  GLX_HIP(hipMemcpyAsync(d_nodes.p, nodes, (size_t)n * 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipMemcpyAsync(d_off.p, offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s));
  if (total_in > 0) {
    GLX_HIP(hipMemcpyAsync(d_nbr.p, nbr, (size_t)total_in * 8, hipMemcpyHostToDevice, s));
    GLX_HIP(hipMemcpyAsync(d_eid.p, eid, (size_t)total_in * 8, hipMemcpyHostToDevice, s));
  }
  if (capacity > 0) {
    GLX_HIP(hipMalloc(&d_row.p, (size_t)capacity * 4));
    GLX_HIP(hipMalloc(&d_col.p, (size_t)capacity * 4));
    GLX_HIP(hipMalloc(&d_eout.p, (size_t)capacity * 8));
  }
  rc = induce_device(d_nodes.as<int64_t>(), n, d_off.as<int64_t>(), d_nbr.as<int64_t>(), d_eid.as<int64_t>(), d_row.as<int32_t>(),
                     d_col.as<int32_t>(), d_eout.as<int64_t>(), capacity, count_out, s);
  if (rc != GLX_OK) return rc;
  const int64_t got = *count_out < capacity ? *count_out : capacity;
  if (got > 0) {
    GLX_HIP(hipMemcpyAsync(row_out, d_row.p, (size_t)got * 4, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(col_out, d_col.p, (size_t)got * 4, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(eid_out, d_eout.p, (size_t)got * 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  return GLX_OK;
}
