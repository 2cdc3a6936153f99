// glx: neighbour sampling with a Filter.
// Replaces op::Filter (core/operator/sampler/filter.h:30-125, filter.cc:69-229) as the
// samplers call it: RandomSampler's HitAll / Hit rejection loop (random_sampler.cc:52-71)
// and ActOn in front of Topk / RandomWithoutReplacement / EdgeWeight / InDegree / Full
// (topk_sampler.cc:52-61, random_without_replacement_sampler.cc:56-68,
// edge_weight_sampler.cc:55-66,94-112, in_degree_sampler.cc:54-65,94-113,
// full_sampler.cc:66-84).
//
// A filtered request is served in stages that all stay on the device:
//   rows      request row -> (first slot, degree); exclusive scan of the degrees gives every
//             row a private span of a scratch array
//   reserve   one wave per row writes ActOn's reserved positions, in the reference's order,
//             into the row's span (closed form of the in-place partition, see the kernel)
//   draw      per strategy: nothing (Topk/Full), a Fisher-Yates over the span (RWoR), or a
//             per-row alias build over the reserved weights (EdgeWeight/InDegree)
//   pad       slots are filled through the span with the padders' rules
// RandomSampler never materialises the reserved set: a count kernel answers HitAll and the
// slot kernel redraws hits from the row's random stream.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <rocprim/rocprim.hpp>

#include "glx_common.h"

namespace {

struct FilterDev {
  int32_t type, field;
  const int64_t* values;
  const int64_t* ts;  // per slot, or nullptr
  int64_t default_ts;
  int32_t retry;
};

// Filter::GetFieldFunc, filter.cc:122-150
__device__ __forceinline__ int64_t field_of(const FilterDev& f, const GlxAdj* __restrict__ adj, int64_t slot) {
  if (f.field == GLX_FILTER_FIELD_ID) return adj[slot].nbr;
  if (f.field == GLX_FILTER_FIELD_TIMESTAMP) return f.ts ? f.ts[slot] : f.default_ts;
  return -1;
}

// Filter::GetFilterFunc, filter.cc:152-193
__device__ __forceinline__ bool hit_of(const FilterDev& f, int64_t v, int64_t value) {
  if (f.type == GLX_FILTER_EQUAL) return v == value;
  if (f.type == GLX_FILTER_LARGER_THAN) return v > value;
  return false;
}

struct RowArgs {
  GlxIdMap map;
  const int64_t* row_ptr;
  const GlxAdj* adj;
  const int64_t* src;
  const int64_t* rng_rows;
  int32_t batch;
};

__global__ void glx_filter_rows_kernel(RowArgs a, int64_t* __restrict__ start, int32_t* __restrict__ deg,
                                       int64_t* __restrict__ deg64) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.batch) return;
  const int64_t row = glx_row_of(a.map, a.src[i]);
  int64_t s = 0, d = 0;
  if (row >= 0) {
    s = a.row_ptr[row];
    d = a.row_ptr[row + 1] - s;
  }
  start[i] = s;
  deg[i] = (int32_t)d;
  if (deg64) deg64[i] = d;
}

__global__ void glx_filter_zero_kernel(int64_t* p) { *p = 0; }

// Filter::HitAll (filter.cc:109-120): nonhit[i] > 0 as soon as one neighbour of request row i
// does not hit -- the scan stops at the first 64-slot chunk that holds a survivor, so it only
// walks a whole row when (nearly) everything in it hits.
__global__ __launch_bounds__(64) void glx_filter_count_kernel(FilterDev f, const GlxAdj* __restrict__ adj,
                                                              const int64_t* __restrict__ start,
                                                              const int32_t* __restrict__ deg,
                                                              int32_t* __restrict__ nonhit) {
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t n = deg[i];
  const int64_t s = start[i];
  const int64_t val = f.values[i];
  int32_t cnt = 0;
  for (int32_t base = 0; base < n; base += 64) {
    const int32_t p = base + lane;
    const bool nh = p < n && !hit_of(f, field_of(f, adj, s + p), val);
    cnt += __popcll(__ballot(nh));
    if (cnt > 0) break;
  }
  if (lane == 0) nonhit[i] = cnt;
}

// Timestamp + LARGER_THAN (filter.cc:74-82, FindkthLargest :196-229): binary search for
// values[0] -- request row 0's value, whatever the row -- over the timestamp-ascending row.  The
// reserved set is the prefix below the found position, listed descending; a row with a single
// neighbour keeps nothing.  Returns the prefix length.
__device__ __forceinline__ int32_t ts_prefix_of(const FilterDev& f, const GlxAdj* __restrict__ adj, int64_t s,
                                                int32_t n) {
  const int64_t filter = f.values[0];
  int32_t lo = 0, hi = n - 1, mid = 0;
  if (hi <= 0) return 0;  // n == 1: FindkthLargest returns -1, ActOn clamps to 0
  while (hi >= lo) {
    mid = lo + (hi - lo) / 2;
    const int64_t v = field_of(f, adj, s + mid);
    if (v == filter) return mid;
    if (v > filter) hi = mid - 1;
    else lo = mid + 1;
  }
  if (field_of(f, adj, s + mid) < filter) mid += 1;
  return mid;
}

// One thread per request row: prefix[i] for the samplers that can work on a prefix directly.
__global__ void glx_filter_tsprefix_kernel(RowArgs a, FilterDev f, int32_t* __restrict__ prefix) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.batch) return;
  const int64_t row = glx_row_of(a.map, a.src[i]);
  int32_t m = 0;
  if (row >= 0) {
    const int64_t s = a.row_ptr[row];
    m = ts_prefix_of(f, a.adj, s, (int32_t)(a.row_ptr[row + 1] - s));
  }
  prefix[i] = m;
}

// Filter::ActOn for one request row per wave.
//
// General path (filter.cc:83-94): the reference walks l upwards and, while indices[l] hits,
// swaps it with indices[r] and shrinks r.  Net effect on the kept prefix [0, m), m = number
// of survivors: a surviving position keeps its place; the holes (hit positions below m), taken
// in ascending order, receive the survivors at positions >= m in DESCENDING order.  So three
// passes with ballots: count m; write survivors below m in place and list the ones at or above m
// (ascending) in the unused tail [m, n) of the row's span; fill hole h from tail entry
// F - 1 - h (F = number of holes = number of tail survivors).
//
// Timestamp + LARGER_THAN: the descending prefix of ts_prefix_of.
//
// Rows [row0, row0 + gridDim.x) of the request are served per launch; their spans start at
// soff[i] - base of the scratch array (the request is cut into chunks of bounded total degree).
__global__ __launch_bounds__(64) void glx_filter_reserve_kernel(FilterDev f, const GlxAdj* __restrict__ adj,
                                                                const int64_t* __restrict__ start,
                                                                const int32_t* __restrict__ deg,
                                                                const int64_t* __restrict__ soff, int32_t row0,
                                                                int64_t base, int32_t* __restrict__ res,
                                                                int32_t* __restrict__ res_cnt) {
  const int64_t i = row0 + (int64_t)blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t n = deg[i];
  if (n == 0) {
    if (lane == 0) res_cnt[i] = 0;
    return;
  }
  const int64_t s = start[i];
  int32_t* R = res + (soff[i] - base);
  if (f.field == GLX_FILTER_FIELD_TIMESTAMP && f.type == GLX_FILTER_LARGER_THAN) {
    int32_t k = 0;
    if (lane == 0) k = ts_prefix_of(f, adj, s, n);
    k = __shfl(k, 0);
    for (int32_t t = lane; t < k; t += 64) R[t] = k - 1 - t;
    if (lane == 0) res_cnt[i] = k;
    return;
  }
  const int64_t val = f.values[i];
  const uint64_t lt = (1ull << lane) - 1ull;
  int32_t m = 0;
  for (int32_t base = 0; base < n; base += 64) {
    const int32_t p = base + lane;
    const bool nh = p < n && !hit_of(f, field_of(f, adj, s + p), val);
    m += __popcll(__ballot(nh));
  }
  int32_t tail = 0;  // survivors at positions >= m seen so far
  for (int32_t base = 0; base < n; base += 64) {
    const int32_t p = base + lane;
    const bool nh = p < n && !hit_of(f, field_of(f, adj, s + p), val);
    const bool up = nh && p >= m;
    const uint64_t b = __ballot(up);
    if (nh && p < m) R[p] = p;
    if (up) R[m + tail + __popcll(b & lt)] = p;
    tail += __popcll(b);
  }
  __syncthreads();  // one wave per block: orders the tail list before the hole fill
  int32_t holes = 0;
  for (int32_t base = 0; base < m; base += 64) {
    const int32_t p = base + lane;
    const bool h = p < m && hit_of(f, field_of(f, adj, s + p), val);
    const uint64_t b = __ballot(h);
    if (h) {
      const int32_t rank = holes + __popcll(b & lt);
      R[p] = R[m + tail - 1 - rank];
    }
    holes += __popcll(b);
  }
  if (lane == 0) res_cnt[i] = m;
}

struct DrawArgs {
  const GlxAdj* adj;
  const int64_t* start;
  const int32_t* deg;
  const int64_t* soff;
  const int32_t* res;
  const int32_t* res_cnt;
  const int64_t* rng_rows;
  uint64_t seed, cc;
  int64_t default_nbr;
  int32_t batch, k;
  int32_t row0, nrows;  // the rows of the request this launch serves
  int64_t base;         // soff[row0]: spans are relative to it
};

// RandomWithoutReplacement: std::shuffle(reserved) under the contract's forward Fisher-Yates
// (step j swaps entry j with entry j + bounded(draw_j, m - j)); only the first min(k, m) steps
// can reach the padded output.  One thread per request row, in place on the row's span.
__global__ void glx_filter_shuffle_kernel(DrawArgs a, int32_t* __restrict__ res) {
  const int32_t li = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t i = a.row0 + li;
  if (li >= a.nrows || a.deg[i] == 0) return;
  const int32_t m = a.res_cnt[i];
  int32_t* R = res + (a.soff[i] - a.base);
  const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
  const int32_t steps = m < a.k ? m : a.k;
  for (int32_t j = 0; j < steps; ++j) {
    const int32_t r = j + (int32_t)glx_bounded(glx_draw64(a.seed, a.cc, rr, (uint32_t)j), (uint64_t)(m - j));
    const int32_t t = R[j];
    R[j] = R[r];
    R[r] = t;
  }
}

// SampleFromIndices (edge_weight_sampler.cc:94-112, in_degree_sampler.cc:94-113): the alias
// table of the reserved neighbours' weights, rebuilt per request row like the reference does.
// One thread per row; dist / tab / stk are spans parallel to the reserved list.
__global__ void glx_filter_alias_build_kernel(DrawArgs a, const float* __restrict__ weight, GlxIdMap dst_map,
                                              const int64_t* __restrict__ dst_count, float* __restrict__ dist,
                                              GlxAlias* __restrict__ tab, GlxAlias* __restrict__ stk) {
  const int32_t li = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t i = a.row0 + li;
  if (li >= a.nrows || a.deg[i] == 0) return;
  const int32_t m = a.res_cnt[i];
  if (m == 0 || m > kAliasLaneRowMax) return;  // longer reserved lists: glx_filter_alias_build_wave_kernel
  const int64_t off = a.soff[i] - a.base, s = a.start[i];
  for (int32_t t = 0; t < m; ++t) {
    const int64_t slot = s + a.res[off + t];
    float w;
    if (weight) {
      w = weight[slot];
    } else {
      const int64_t r = glx_row_of(dst_map, a.adj[slot].nbr);
      w = r < 0 ? 0.0f : (float)(int32_t)dst_count[r];  // static_cast<float>(GetInDegree(id))
    }
    dist[off + t] = w;
  }
  glx_alias_build_row_dev(dist + off, m, tab + off, stk + off);
}

// The same for reserved lists longer than kAliasLaneRowMax: one wave per request row gathers the weights
// (coalesced over the reserved list) and runs glx_alias_build_row_wave.
__global__ __launch_bounds__(256) void glx_filter_alias_build_wave_kernel(DrawArgs a, const float* __restrict__ weight,
                                                                          GlxIdMap dst_map,
                                                                          const int64_t* __restrict__ dst_count,
                                                                          float* __restrict__ dist, GlxAlias* __restrict__ tab,
                                                                          GlxAlias* __restrict__ stk) {
  const int32_t li = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int32_t i = a.row0 + li;
  if (li >= a.nrows || a.deg[i] == 0) return;
  const int32_t m = a.res_cnt[i];
  if (m <= kAliasLaneRowMax) return;
  const int64_t off = a.soff[i] - a.base, s = a.start[i];
  for (int32_t t = lane; t < m; t += 64) {
    const int64_t slot = s + a.res[off + t];
    float w;
    if (weight) {
      w = weight[slot];
    } else {
      const int64_t r = glx_row_of(dst_map, a.adj[slot].nbr);
      w = r < 0 ? 0.0f : (float)(int32_t)dst_count[r];
    }
    dist[off + t] = w;
  }
  __threadfence_block();
  glx_alias_build_row_wave(dist + off, m, tab + off, stk + off);
}

// EdgeWeight / InDegree slots under circular padding: k alias draws mapped back through the
// reserved list (indices.size() == k, so the padder is the identity over them).
__global__ __launch_bounds__(256) void glx_filter_alias_slots_kernel(DrawArgs a, const GlxAlias* __restrict__ tab,
                                                                     int64_t* __restrict__ nbr_out,
                                                                     int64_t* __restrict__ eid_out) {
  const int64_t lt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (lt >= (int64_t)a.nrows * a.k) return;
  const int32_t i = a.row0 + (int32_t)(lt / a.k);
  const int32_t j = (int32_t)(lt % a.k);
  const int64_t t = (int64_t)i * a.k + j;
  GlxAdj rec = GlxAdj{a.default_nbr, -1};
  const int32_t m = a.deg[i] > 0 ? a.res_cnt[i] : 0;
  if (m > 0) {
    const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
    const int64_t off = a.soff[i] - a.base;
    const int32_t pick = glx_alias_pick(glx_draw64(a.seed, a.cc, rr, (uint32_t)j), m, tab + off);
    rec = a.adj[a.start[i] + a.res[off + pick]];
  }
  nbr_out[t] = rec.nbr;
  eid_out[t] = rec.eid;
}

// The padders over a reserved list (circular_padder.h:36-66, replicate_padder.h:37-56); one
// wave per request row, target = k (dense) or the row's segment (FullSampler).
//   kPadCircular        m == 0: default ids; else slot j = reserved[j % m]
//   kPadReplicate       the first min(target, m) neighbours IN ROW ORDER (the padder ignores the
//                       index values), then default ids
//   kPadReplicateDrawn  EdgeWeight / InDegree: the index list has k entries unless nothing was
//                       reserved, so min(target, deg) row-order neighbours (clamped to the row
//                       instead of the reference's out-of-bounds read), or all default
enum { kPadCircular = 0, kPadReplicate = 1, kPadReplicateDrawn = 2 };
__global__ __launch_bounds__(256) void glx_filter_pad_kernel(DrawArgs a, const int64_t* __restrict__ offsets, int mode,
                                                             int64_t* __restrict__ nbr_out,
                                                             int64_t* __restrict__ eid_out) {
  const int64_t li = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (li >= a.nrows) return;
  const int64_t i = a.row0 + li;
  const int32_t n = a.deg[i];
  const int32_t m = n > 0 ? a.res_cnt[i] : 0;
  const int64_t s = a.start[i];
  const int64_t o0 = offsets ? offsets[i] : i * (int64_t)a.k;
  const int64_t target = offsets ? offsets[i + 1] - o0 : a.k;
  // res == nullptr: the reserved list is the descending prefix (timestamp > value), not stored
  const int32_t* R = a.res ? a.res + (a.soff[i] - a.base) : nullptr;
  for (int64_t j = lane; j < target; j += 64) {
    GlxAdj rec = GlxAdj{a.default_nbr, -1};
    if (mode == kPadCircular) {
      if (m > 0) rec = a.adj[s + (R ? R[j % m] : m - 1 - (int32_t)(j % m))];
    } else if (mode == kPadReplicate) {
      if (j < m) rec = a.adj[s + j];
    } else {
      if (m > 0 && j < n) rec = a.adj[s + j];
    }
    nbr_out[o0 + j] = rec.nbr;
    eid_out[o0 + j] = rec.eid;
  }
}

// RandomSampler (random_sampler.cc:58-71): HitAll rows are default-filled; every other slot
// redraws a hit `retry` times and keeps the next draw whatever it is.
__global__ __launch_bounds__(256) void glx_filter_random_kernel(DrawArgs a, FilterDev f,
                                                                const int32_t* __restrict__ nonhit,
                                                                int64_t* __restrict__ nbr_out,
                                                                int64_t* __restrict__ eid_out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)a.batch * a.k) return;
  const int32_t i = (int32_t)(t / a.k);
  const int32_t j = (int32_t)(t - (int64_t)i * a.k);
  GlxAdj rec = GlxAdj{a.default_nbr, -1};
  const int32_t n = a.deg[i];
  if (n > 0 && nonhit[i] > 0) {
    const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
    const int64_t s = a.start[i];
    const int64_t val = f.values[i];
    for (int32_t at = 0;; ++at) {
      const uint32_t draw = (uint32_t)j + (uint32_t)at * (uint32_t)a.k;
      const int64_t d = (int64_t)glx_bounded(glx_draw64(a.seed, a.cc, rr, draw), (uint64_t)n);
      if (at >= f.retry || !hit_of(f, field_of(f, a.adj, s + d), val)) {
        rec = a.adj[s + d];
        break;
      }
    }
  }
  nbr_out[t] = rec.nbr;
  eid_out[t] = rec.eid;
}

// id == value filters (GSL's .filter()): the closed form of Filter::ActOn (filter.cc:83-94, the one
// glx_filter_reserve_kernel materialises) evaluated lazily.  With H hit positions in a row of n neighbours and
// m = n - H survivors, the reserved list is
//   R(p) = p                                                 when position p is not a hit,
//   R(h) = the r-th survivor of [m, n), counted downwards    when h < m is the r-th hit (ascending) -- a hole.
// Whether p is a hit is one look at adj[p]; r = the number of hits below h; the r-th survivor from the top is the
// fixed point of q = n - 1 - r - #(hits >= q).  Both counts run over the row's hit list in ANY order, so the list
// is never sorted or copied: with the id-sorted row index it is the run slot_sorted[lo, lo + H) the binary
// search lands on (any H); without the index one ballot scan lists up to kMaxHits positions and rows with more
// take the general path.
constexpr int kMaxHits = 8;
constexpr int kFastMaxK = 32;

struct HitArgs {
  const GlxAdj* adj;
  const int64_t* start;
  const int32_t* deg;
  const int64_t* values;
  const int64_t* nbr_sorted;  // per row ascending ids + the CSR slots they came from, or nullptr
  const uint32_t* slot_sorted;
  int32_t* nhits;             // [batch]
  int32_t* hit_lo;            // [batch] with the index: where the run of hits starts in the row's sorted ids
  int32_t* hits;              // [batch * kMaxHits] without the index: the first hit positions
  int32_t batch;
};

// With the index: one thread per row, a binary search and a count.
__global__ void glx_filter_idhits_index_kernel(HitArgs a) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.batch) return;
  const int32_t n = a.deg[i];
  const int64_t s = a.start[i], val = a.values[i];
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (a.nbr_sorted[s + mid] < val) lo = mid + 1; else hi = mid;
  }
  int32_t H = 0;
  if (lo < n && a.nbr_sorted[s + lo] == val) {  // the run's end: a second search (a hub can hold thousands of hits)
    int32_t l2 = lo + 1, h2 = n;
    while (l2 < h2) {
      const int32_t mid = l2 + ((h2 - l2) >> 1);
      if (a.nbr_sorted[s + mid] <= val) l2 = mid + 1; else h2 = mid;
    }
    H = l2 - lo;
  }
  a.nhits[i] = H;
  a.hit_lo[i] = lo;
}

// Without it: one wave per row, one pass.
__global__ __launch_bounds__(64) void glx_filter_idhits_scan_kernel(HitArgs a) {
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t n = a.deg[i];
  const int64_t s = a.start[i], val = a.values[i];
  const uint64_t lt = (1ull << lane) - 1ull;
  int32_t H = 0;
  for (int32_t base = 0; base < n; base += 64) {
    const int32_t p = base + lane;
    const bool h = p < n && a.adj[s + p].nbr == val;
    const uint64_t b = __ballot(h);
    if (h) {
      const int32_t at = H + __popcll(b & lt);
      if (at < kMaxHits) a.hits[i * kMaxHits + at] = p;
    }
    H += __popcll(b);
  }
  if (lane == 0) a.nhits[i] = H;
}

// One row's hit positions, in no particular order.
struct HitList {
  const uint32_t* run;  // slot_sorted + start + lo (CSR slots), or nullptr
  const int32_t* arr;   // row-local positions
  int64_t s;
  int32_t H;
  __device__ __forceinline__ int32_t pos(int32_t t) const { return run ? (int32_t)((int64_t)run[t] - s) : arr[t]; }
};

struct FastArgs {
  DrawArgs d;
  const int32_t* nhits;
  const int32_t* hit_lo;
  const int32_t* hits;
  const uint32_t* slot_sorted;  // nullptr: the lists are in `hits`
  const int64_t* values;
  int32_t* general;   // [batch] 1 = this row takes the general path
  int32_t sampler;
  int32_t circular;
};

__device__ __forceinline__ HitList hit_list(const FastArgs& a, int32_t i, int64_t s, int32_t H) {
  if (a.slot_sorted) return HitList{a.slot_sorted + s + a.hit_lo[i], nullptr, s, H};
  return HitList{nullptr, a.hits + (int64_t)i * kMaxHits, s, H};
}

// R(p) for p in [0, m), m = n - H > 0.
__device__ __forceinline__ int32_t reserved_at(int32_t p, const HitList& h, int32_t n, const GlxAdj* __restrict__ row,
                                               int64_t val) {
  if (h.H == 0 || row[p].nbr != val) return p;
  int32_t r = 0;
  for (int32_t t = 0; t < h.H; ++t) r += h.pos(t) < p ? 1 : 0;
  int32_t q = n - 1 - r;
  while (true) {
    int32_t c = 0;
    for (int32_t t = 0; t < h.H; ++t) c += h.pos(t) >= q ? 1 : 0;
    const int32_t q2 = n - 1 - r - c;
    if (q2 == q) return q;
    q = q2;
  }
}

// Which rows the closed form does not serve: more hits than the scan lists (no index); without-replacement
// beyond the register budget; and the alias samplers as soon as a single neighbour is filtered out (their table
// must be rebuilt over the reserved weights -- rows WITHOUT a hit keep the table built at load, which is what
// the reference's per-request build over the unchanged row yields).
__global__ void glx_filter_classify_kernel(FastArgs a) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.d.batch) return;
  const int32_t H = a.nhits[i];
  const bool alias = a.sampler == GLX_SAMPLER_EDGE_WEIGHT || a.sampler == GLX_SAMPLER_IN_DEGREE;
  bool gen = a.slot_sorted == nullptr && H > kMaxHits;
  if (alias) gen = H > 0;
  if (a.sampler == GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT && a.circular && a.d.k > kFastMaxK) gen = H > 0;
  a.general[i] = gen ? 1 : 0;
}

// Topk (and every sampler's replicate padding): one thread per output slot.
__global__ __launch_bounds__(256) void glx_filter_fast_topk_kernel(FastArgs a, int64_t* __restrict__ nbr_out,
                                                                   int64_t* __restrict__ eid_out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)a.d.batch * a.d.k) return;
  const int32_t i = (int32_t)(t / a.d.k);
  const int32_t j = (int32_t)(t - (int64_t)i * a.d.k);
  if (a.general[i]) return;
  const int32_t n = a.d.deg[i];
  const int32_t H = a.nhits[i];
  const int32_t m = n - H;
  GlxAdj rec = GlxAdj{a.d.default_nbr, -1};
  const int64_t s = a.d.start[i];
  if (a.circular) {
    if (m > 0) {
      const int32_t p = j % m;
      rec = a.d.adj[s + p];
      if (H > 0 && rec.nbr == a.values[i]) rec = a.d.adj[s + reserved_at(p, hit_list(a, i, s, H), n, a.d.adj + s, a.values[i])];
    }
  } else if (j < m) {
    rec = a.d.adj[s + j];  // ReplicatePadder ignores the index values
  }
  nbr_out[t] = rec.nbr;
  eid_out[t] = rec.eid;
}

// RandomWithoutReplacement, circular padding, k <= kFastMaxK: one thread per row runs the contract's
// forward Fisher-Yates over the implicit reserved list with a sparse record of the swapped entries.
__global__ void glx_filter_fast_rwor_kernel(FastArgs a, int64_t* __restrict__ nbr_out, int64_t* __restrict__ eid_out) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.d.batch || a.general[i]) return;
  const int32_t n = a.d.deg[i];
  const int32_t H = a.nhits[i];
  const int32_t m = n - H;
  const int32_t k = a.d.k;
  const int64_t o = (int64_t)i * k;
  if (m <= 0) {
    for (int32_t j = 0; j < k; ++j) {
      nbr_out[o + j] = a.d.default_nbr;
      eid_out[o + j] = -1;
    }
    return;
  }
  const int64_t s = a.d.start[i];
  const HitList hl = hit_list(a, i, s, H);
  const int64_t val = a.values[i];
  const GlxAdj* __restrict__ row = a.d.adj + s;
  const uint32_t rr = a.d.rng_rows ? (uint32_t)a.d.rng_rows[i] : (uint32_t)i;
  const int32_t steps = m < k ? m : k;
  int32_t okey[kFastMaxK], oval[kFastMaxK], outv[kFastMaxK];
  int32_t no = 0;
  for (int32_t j = 0; j < steps; ++j) {
    const int32_t r = j + (int32_t)glx_bounded(glx_draw64(a.d.seed, a.d.cc, rr, (uint32_t)j), (uint64_t)(m - j));
    int32_t aj = -1, ar = -1, at_r = -1;
    for (int32_t t = 0; t < no; ++t) {
      if (okey[t] == j) aj = oval[t];
      if (okey[t] == r) {
        ar = oval[t];
        at_r = t;
      }
    }
    if (aj < 0) aj = reserved_at(j, hl, n, row, val);
    if (ar < 0) ar = r == j ? aj : reserved_at(r, hl, n, row, val);
    outv[j] = ar;  // entry j is final after this step
    if (r != j) {
      if (at_r >= 0) oval[at_r] = aj;
      else {
        okey[no] = r;
        oval[no] = aj;
        ++no;
      }
    }
  }
  for (int32_t j = 0; j < k; ++j) {
    const GlxAdj rec = row[outv[j % m]];  // j % m < steps
    nbr_out[o + j] = rec.nbr;
    eid_out[o + j] = rec.eid;
  }
}

// The rows left to the general path, packed into a request of their own (src, random stream row,
// filter value) + where their answers go.
__global__ void glx_filter_pack_general_kernel(const int32_t* __restrict__ general, const int64_t* __restrict__ src,
                                               const int64_t* __restrict__ rng, const int64_t* __restrict__ values,
                                               int32_t batch, int32_t* __restrict__ count, int32_t* __restrict__ gidx,
                                               int64_t* __restrict__ sub_src, int64_t* __restrict__ sub_rng,
                                               int64_t* __restrict__ sub_val) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool take = i < batch && general[i];
  // one atomic per wave, not per row: a request can leave hundreds of thousands of rows to the general path
  const uint64_t b = __ballot(take);
  if (!b) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)b) - 1;
  int32_t start = 0;
  if (lane == leader) start = atomicAdd(count, __popcll(b));
  start = __shfl(start, leader);
  if (!take) return;
  const int32_t at = start + __popcll(b & ((1ull << lane) - 1ull));
  gidx[at] = i;
  sub_src[at] = src[i];
  sub_rng[at] = rng ? rng[i] : (int64_t)i;
  sub_val[at] = values[i];
}

__global__ void glx_filter_unpack_general_kernel(const int32_t* __restrict__ gidx, int32_t G, int32_t k,
                                                 const int64_t* __restrict__ sub_nbr, const int64_t* __restrict__ sub_eid,
                                                 int64_t* __restrict__ nbr_out, int64_t* __restrict__ eid_out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)G * k) return;
  const int32_t r = (int32_t)(t / k);
  const int32_t j = (int32_t)(t - (int64_t)r * k);
  const int64_t o = (int64_t)gidx[r] * k + j;
  nbr_out[o] = sub_nbr[t];
  eid_out[o] = sub_eid[t];
}

constexpr int kFullSampler = -1;
// Upper bound on the reserved positions held in scratch at once.  A request whose rows'
// degrees add up to more is served in chunks of consecutive rows (a single larger row is a
// chunk of its own), one after the other.  1 Gi positions keep the workspace at 4 GiB (24 GiB
// with alias tables), small enough to stay cached per (thread, stream): allocating and
// freeing tens of GB per call costs seconds, far more than the kernels.  When HBM is short the
// bound shrinks to what half of the free memory holds; a workspace beyond 32 GiB (one huge
// row) is returned to the device after the call.
constexpr int64_t kSpanCap = (int64_t)1 << 30;
int64_t span_cap(size_t bytes_per_position) {
  const int64_t v = glx_side_knobs().filter_span_cap.load(std::memory_order_relaxed);  // test knob: force many small chunks
  if (v > 0) return v;
  size_t free_b = 0, total_b = 0;
  int64_t cap = kSpanCap;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    const int64_t fit = (int64_t)(free_b / 2 / bytes_per_position);
    if (fit < cap) cap = fit;
  }
  return cap > ((int64_t)1 << 20) ? cap : ((int64_t)1 << 20);
}

// ---- one alias table per DISTINCT (vertex, filter value) pair of a request -----------------------------------
// The reference rebuilds the table for every request row (edge_weight_sampler.cc:94-112).  The table is a function
// of the row's reserved list and weights only, i.e. of (vertex, filter value) -- and timestamp > value filters use
// request row 0's value for every row (filter.cc:74-82), so there the vertex alone decides.  Hop-2 frontiers of a
// power-law graph name the same hubs thousands of times: the request rows are sorted by that pair, every distinct
// pair gets ONE span / reserved list / table, and each request row draws from its pair's table with its own
// random stream.  Bit-identical to the per-row build (same list, same weights, same serial algorithm).
__global__ void glx_filter_dedup_keys_kernel(const int64_t* __restrict__ start, const int32_t* __restrict__ deg,
                                             const int64_t* __restrict__ values, int same_value, int32_t batch,
                                             int64_t* __restrict__ key1, int64_t* __restrict__ key2,
                                             int32_t* __restrict__ iota) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  key1[i] = deg[i] > 0 ? start[i] : -1;
  key2[i] = same_value ? 0 : values[i];
  iota[i] = i;
}

__global__ void glx_filter_dedup_gather_kernel(const int64_t* __restrict__ key, const int32_t* __restrict__ perm,
                                               int32_t batch, int64_t* __restrict__ out) {
  const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < batch) out[t] = key[perm[t]];
}

__global__ void glx_filter_dedup_flag_kernel(const int64_t* __restrict__ k1, const int64_t* __restrict__ k2,
                                             int32_t batch, int32_t* __restrict__ flag) {
  const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch) return;
  flag[t] = (t == 0 || k1[t] != k1[t - 1] || k2[t] != k2[t - 1]) ? 1 : 0;
}

// uid1[t] = 1 + the pair index of sorted position t (inclusive scan of the head flags).
__global__ void glx_filter_dedup_sub_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ uid1,
                                            const int32_t* __restrict__ perm, const int64_t* __restrict__ start,
                                            const int32_t* __restrict__ deg, const int64_t* __restrict__ values,
                                            int same_value, int32_t batch, int64_t* __restrict__ sub_start,
                                            int32_t* __restrict__ sub_deg, int64_t* __restrict__ sub_deg64,
                                            int64_t* __restrict__ sub_val, int32_t* __restrict__ run_begin) {
  const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch) return;
  if (t == batch - 1) run_begin[uid1[t]] = batch;
  if (!flag[t]) return;
  const int32_t u = uid1[t] - 1, i = perm[t];
  sub_start[u] = start[i];
  sub_deg[u] = deg[i];
  sub_deg64[u] = deg[i];
  sub_val[u] = same_value ? values[0] : values[i];
  run_begin[u] = t;
}

// k alias draws per request row from its pair's table; sorted positions [t0, t1) of the request.
__global__ __launch_bounds__(256) void glx_filter_alias_slots_dedup_kernel(DrawArgs a, const GlxAlias* __restrict__ tab,
                                                                           const int32_t* __restrict__ perm,
                                                                           const int32_t* __restrict__ uid1, int32_t t0,
                                                                           int32_t t1, const int64_t* __restrict__ rng_rows,
                                                                           int64_t* __restrict__ nbr_out,
                                                                           int64_t* __restrict__ eid_out) {
  const int64_t lt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (lt >= (int64_t)(t1 - t0) * a.k) return;
  const int32_t t = t0 + (int32_t)(lt / a.k);
  const int32_t j = (int32_t)(lt % a.k);
  const int32_t i = perm[t], u = uid1[t] - 1;
  GlxAdj rec = GlxAdj{a.default_nbr, -1};
  const int32_t m = a.deg[u] > 0 ? a.res_cnt[u] : 0;
  if (m > 0) {
    const uint32_t rr = rng_rows ? (uint32_t)rng_rows[i] : (uint32_t)i;
    const int64_t off = a.soff[u] - a.base;
    const int32_t pick = glx_alias_pick(glx_draw64(a.seed, a.cc, rr, (uint32_t)j), m, tab + off);
    rec = a.adj[a.start[u] + a.res[off + pick]];
  }
  const int64_t o = (int64_t)i * a.k + j;
  nbr_out[o] = rec.nbr;
  eid_out[o] = rec.eid;
}

int dedup_min_rows() {
  const int64_t v = glx_side_knobs().filter_dedup_min_rows.load(std::memory_order_relaxed);  // test knob; 0 disables
  return v < 0 ? 1024 : (int)v;
}

#define GLX_FILTER_PRIM(call_with)                                             \
  do {                                                                         \
    size_t bytes__ = 0;                                                        \
    void* tmp__ = nullptr;                                                     \
    GLX_HIP(call_with(tmp__, bytes__));                                        \
    int rc__ = glx_scratch_alloc(&tmp__, bytes__ ? bytes__ : 8, s, 3);         \
    if (rc__ != GLX_OK) return rc__;                                           \
    GLX_HIP(call_with(tmp__, bytes__));                                        \
  } while (0)

// EdgeWeight / InDegree under circular padding, any filter: start / deg are the request rows' (filled by the caller).
int filtered_alias_dedup(const glx_graph* g, int sampler, const int64_t* d_rng, int32_t batch, int32_t k,
                         int64_t default_nbr, uint64_t seed, uint64_t cc, FilterDev f, const int64_t* start,
                         const int32_t* deg, int64_t* d_nbr, int64_t* d_eid, hipStream_t s) {
  const size_t nb = (size_t)batch;
  const bool same_value = f.field == GLX_FILTER_FIELD_TIMESTAMP && f.type == GLX_FILTER_LARGER_THAN;
  // i64: key1 key2 k1s k2s sub_start sub_val soff_u[nb + 1] ; i32: iota permA perm flag uid1 sub_deg cnt_u run_begin[nb + 1]
  char* buf = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&buf), (nb * 7 + 1) * 8 + (nb * 8 + 1) * 4 + 16, s, 6);
  if (rc != GLX_OK) return rc;
  int64_t* key1 = reinterpret_cast<int64_t*>(buf);
  int64_t* key2 = key1 + nb;
  int64_t* k1s = key2 + nb;
  int64_t* k2s = k1s + nb;
  int64_t* sub_start = k2s + nb;
  int64_t* sub_val = sub_start + nb;
  int64_t* soff = sub_val + nb;
  int32_t* iota = reinterpret_cast<int32_t*>(soff + nb + 1);
  int32_t* permA = iota + nb;
  int32_t* perm = permA + nb;
  int32_t* flag = perm + nb;
  int32_t* uid1 = flag + nb;
  int32_t* sub_deg = uid1 + nb;
  int32_t* cnt = sub_deg + nb;
  int32_t* run_begin = cnt + nb;
  const unsigned row_blocks = (unsigned)((nb + 255) / 256);
  glx_filter_dedup_keys_kernel<<<row_blocks, 256, 0, s>>>(start, deg, f.values, same_value ? 1 : 0, batch, key1, key2, iota);
  if (same_value) {
#define SORT1(tmp, bytes) rocprim::radix_sort_pairs(tmp, bytes, key1, k1s, iota, perm, nb, 0, 64, s)
    GLX_FILTER_PRIM(SORT1);
#undef SORT1
    GLX_HIP(hipMemsetAsync(k2s, 0, nb * 8, s));
  } else {
    // by value first, then (stable) by vertex: rows of one (vertex, value) pair end up adjacent
#define SORTA(tmp, bytes) rocprim::radix_sort_pairs(tmp, bytes, key2, k2s, iota, permA, nb, 0, 64, s)
    GLX_FILTER_PRIM(SORTA);
#undef SORTA
    glx_filter_dedup_gather_kernel<<<row_blocks, 256, 0, s>>>(key1, permA, batch, k2s);  // k2s reused: key1 in A order
#define SORTB(tmp, bytes) rocprim::radix_sort_pairs(tmp, bytes, k2s, k1s, permA, perm, nb, 0, 64, s)
    GLX_FILTER_PRIM(SORTB);
#undef SORTB
    glx_filter_dedup_gather_kernel<<<row_blocks, 256, 0, s>>>(key2, perm, batch, k2s);
  }
  glx_filter_dedup_flag_kernel<<<row_blocks, 256, 0, s>>>(k1s, k2s, batch, flag);
#define SCANF(tmp, bytes) rocprim::inclusive_scan(tmp, bytes, flag, uid1, nb, rocprim::plus<int32_t>(), s)
  GLX_FILTER_PRIM(SCANF);
#undef SCANF
  glx_filter_zero_kernel<<<1, 1, 0, s>>>(soff);
  glx_filter_dedup_sub_kernel<<<row_blocks, 256, 0, s>>>(flag, uid1, perm, start, deg, f.values, same_value ? 1 : 0, batch,
                                                         sub_start, sub_deg, soff + 1, sub_val, run_begin);
  int32_t U = 0;
  GLX_HIP(hipMemcpyAsync(&U, uid1 + nb - 1, 4, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  const size_t nu = (size_t)U;
#define SCANS(tmp, bytes) rocprim::inclusive_scan(tmp, bytes, soff + 1, soff + 1, nu, rocprim::plus<int64_t>(), s)
  GLX_FILTER_PRIM(SCANS);
#undef SCANS
  std::vector<int64_t> h_soff(nu + 1);
  std::vector<int32_t> h_run(nu + 1);
  GLX_HIP(hipMemcpyAsync(h_soff.data(), soff, (nu + 1) * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipMemcpyAsync(h_run.data(), run_begin, (nu + 1) * 4, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  const int64_t cap = span_cap(24);
  std::vector<int32_t> cuts{0};
  int64_t widest = 0;
  for (int32_t a = 0; a < U;) {
    int32_t b = a + 1;
    while (b < U && h_soff[b + 1] - h_soff[a] <= cap) ++b;
    if (h_soff[b] - h_soff[a] > widest) widest = h_soff[b] - h_soff[a];
    cuts.push_back(b);
    a = b;
  }
  const size_t span = ((size_t)widest + 1) & ~(size_t)1;
  char* work = nullptr;
  rc = glx_scratch_alloc(reinterpret_cast<void**>(&work), span * 24 + 16, s, 2);
  if (rc != GLX_OK) return rc;
  GlxAlias* tab = reinterpret_cast<GlxAlias*>(work);
  GlxAlias* stk = reinterpret_cast<GlxAlias*>(work + span * 8);
  int32_t* res = reinterpret_cast<int32_t*>(work + span * 16);
  float* dist = reinterpret_cast<float*>(work + span * 20);
  FilterDev fu = f;
  fu.values = sub_val;
  DrawArgs da{g->adj, sub_start, sub_deg, soff, res, cnt, nullptr, seed, cc, default_nbr, U, k, 0, U, 0};
  const GlxIdMap dm = GlxIdMap{g->dst_map.keys, g->dst_map.vals, g->dst_map.cap - 1, g->num_dst};
  const bool by_weight = sampler == GLX_SAMPLER_EDGE_WEIGHT;
  for (size_t c = 0; c + 1 < cuts.size(); ++c) {
    da.row0 = cuts[c];
    da.nrows = cuts[c + 1] - cuts[c];
    da.base = h_soff[da.row0];
    const size_t nr = (size_t)da.nrows;
    const unsigned rb = (unsigned)((nr + 255) / 256), wb = (unsigned)((nr * 64 + 255) / 256);
    glx_filter_reserve_kernel<<<(unsigned)nr, 64, 0, s>>>(fu, g->adj, sub_start, sub_deg, soff, da.row0, da.base, res, cnt);
    glx_filter_alias_build_kernel<<<rb, 256, 0, s>>>(da, by_weight ? g->weight : nullptr, dm, g->dst_count, dist, tab, stk);
    glx_filter_alias_build_wave_kernel<<<wb, 256, 0, s>>>(da, by_weight ? g->weight : nullptr, dm, g->dst_count, dist, tab,
                                                         stk);
    const int32_t t0 = h_run[cuts[c]], t1 = h_run[cuts[c + 1]];
    const int64_t total = (int64_t)(t1 - t0) * k;
    if (total > 0) {
      glx_filter_alias_slots_dedup_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(da, tab, perm, uid1, t0, t1, d_rng,
                                                                                         d_nbr, d_eid);
    }
  }
  GLX_HIP(hipGetLastError());
  glx_scratch_trim(s, 2, (size_t)32 << 30);
  return GLX_OK;
}

// All pointers are device pointers.  `sampler` is a GLX_SAMPLER_* id or kFullSampler (then
// d_offsets[batch + 1] gives the segments and k is unused).
int filtered_general(const glx_graph* g, int sampler, const int64_t* d_src, const int64_t* d_rng, int32_t batch,
                     int32_t k, const int64_t* d_offsets, int padding_mode, int64_t default_nbr, uint64_t seed,
                     uint64_t cc, FilterDev f, int64_t* d_nbr, int64_t* d_eid, hipStream_t s) {
  const bool circular = padding_mode == GLX_PAD_CIRCULAR;
  const size_t nb = (size_t)batch;
  // row info: start[batch] i64 | soff[batch + 1] i64 | deg[batch] i32 | cnt[batch] i32
  char* info = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&info), (nb * 2 + 1) * 8 + nb * 2 * 4, s, 1);
  if (rc != GLX_OK) return rc;
  int64_t* start = reinterpret_cast<int64_t*>(info);
  int64_t* soff = start + nb;
  int32_t* deg = reinterpret_cast<int32_t*>(soff + nb + 1);
  int32_t* cnt = deg + nb;
  RowArgs ra{g->map(), g->row_ptr, g->adj, d_src, d_rng, batch};
  DrawArgs da{g->adj, start, deg, soff, nullptr, cnt, d_rng, seed, cc, default_nbr, batch, k, 0, batch, 0};
  const unsigned row_blocks = (unsigned)((nb + 255) / 256);
  const unsigned wave_blocks = (unsigned)((nb * 64 + 255) / 256);
  GlxKernelTimer timer(GLX_KERNEL_SAMPLE, s);
  if (sampler == GLX_SAMPLER_RANDOM) {
    glx_filter_rows_kernel<<<row_blocks, 256, 0, s>>>(ra, start, deg, nullptr);
    glx_filter_count_kernel<<<(unsigned)batch, 64, 0, s>>>(f, g->adj, start, deg, cnt);
    const int64_t total = (int64_t)batch * k;
    glx_filter_random_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(da, f, cnt, d_nbr, d_eid);
    timer.stop();
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  const bool ts_prefix = f.field == GLX_FILTER_FIELD_TIMESTAMP && f.type == GLX_FILTER_LARGER_THAN;
  if (ts_prefix && (sampler == GLX_SAMPLER_TOPK || sampler == GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT)) {
    // the reserved set is a row prefix: the plain kernels sample from it directly
    glx_filter_tsprefix_kernel<<<row_blocks, 256, 0, s>>>(ra, f, cnt);
    rc = glx_sample_prefix_device(g, sampler, d_src, d_rng, cnt, batch, k, padding_mode, default_nbr, seed, cc, d_nbr,
                                  d_eid, s);
    timer.stop();
    return rc;
  }
  if (ts_prefix && sampler == kFullSampler) {
    glx_filter_rows_kernel<<<row_blocks, 256, 0, s>>>(ra, start, deg, nullptr);
    glx_filter_tsprefix_kernel<<<row_blocks, 256, 0, s>>>(ra, f, cnt);
    glx_filter_pad_kernel<<<wave_blocks, 256, 0, s>>>(da, d_offsets, circular ? kPadCircular : kPadReplicate, d_nbr, d_eid);
    timer.stop();
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  if (circular && (sampler == GLX_SAMPLER_EDGE_WEIGHT || sampler == GLX_SAMPLER_IN_DEGREE) && dedup_min_rows() > 0 &&
      batch >= dedup_min_rows()) {
    glx_filter_rows_kernel<<<row_blocks, 256, 0, s>>>(ra, start, deg, nullptr);
    rc = filtered_alias_dedup(g, sampler, d_rng, batch, k, default_nbr, seed, cc, f, start, deg, d_nbr, d_eid, s);
    timer.stop();
    return rc;
  }
  glx_filter_rows_kernel<<<row_blocks, 256, 0, s>>>(ra, start, deg, soff + 1);
  glx_filter_zero_kernel<<<1, 1, 0, s>>>(soff);
  {
    size_t bytes = 0;
    GLX_HIP(rocprim::inclusive_scan(nullptr, bytes, soff + 1, soff + 1, nb, rocprim::plus<int64_t>(), s));
    void* tmp = nullptr;
    rc = glx_scratch_alloc(&tmp, bytes, s, 3);
    if (rc != GLX_OK) return rc;
    GLX_HIP(rocprim::inclusive_scan(tmp, bytes, soff + 1, soff + 1, nb, rocprim::plus<int64_t>(), s));
  }
  std::vector<int64_t> h_soff(nb + 1);
  GLX_HIP(hipMemcpyAsync(h_soff.data(), soff, (nb + 1) * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  // chunks of consecutive rows with at most kSpanCap reserved positions
  const bool alias_draw = circular && (sampler == GLX_SAMPLER_EDGE_WEIGHT || sampler == GLX_SAMPLER_IN_DEGREE);
  const int64_t cap = span_cap(alias_draw ? 24 : 4);
  std::vector<int32_t> cuts{0};
  int64_t widest = 0;
  for (int32_t a = 0; a < batch;) {
    int32_t b = a + 1;
    while (b < batch && h_soff[b + 1] - h_soff[a] <= cap) ++b;
    if (h_soff[b] - h_soff[a] > widest) widest = h_soff[b] - h_soff[a];
    cuts.push_back(b);
    a = b;
  }
  // [alias table 8 B | stack pairs 8 B |] reserved positions i32 [| weights f32], each `span` long
  const size_t span = ((size_t)widest + 1) & ~(size_t)1;  // keeps the 8-byte table aligned
  char* work = nullptr;
  rc = glx_scratch_alloc(reinterpret_cast<void**>(&work), span * (alias_draw ? 24 : 4) + 16, s, 2);
  if (rc != GLX_OK) return rc;
  GlxAlias* tab = reinterpret_cast<GlxAlias*>(work);
  GlxAlias* stk = reinterpret_cast<GlxAlias*>(work + span * 8);
  int32_t* res = alias_draw ? reinterpret_cast<int32_t*>(work + span * 16) : reinterpret_cast<int32_t*>(work);
  float* dist = reinterpret_cast<float*>(work + span * 20);
  da.res = res;
  const GlxIdMap dm = GlxIdMap{g->dst_map.keys, g->dst_map.vals, g->dst_map.cap - 1, g->num_dst};
  for (size_t c = 0; c + 1 < cuts.size(); ++c) {
    da.row0 = cuts[c];
    da.nrows = cuts[c + 1] - cuts[c];
    da.base = h_soff[da.row0];
    const size_t nr = (size_t)da.nrows;
    const unsigned rb = (unsigned)((nr + 255) / 256), wb = (unsigned)((nr * 64 + 255) / 256);
    glx_filter_reserve_kernel<<<(unsigned)nr, 64, 0, s>>>(f, g->adj, start, deg, soff, da.row0, da.base, res, cnt);
    if (sampler == kFullSampler) {
      glx_filter_pad_kernel<<<wb, 256, 0, s>>>(da, d_offsets, circular ? kPadCircular : kPadReplicate, d_nbr, d_eid);
    } else if (sampler == GLX_SAMPLER_TOPK) {
      glx_filter_pad_kernel<<<wb, 256, 0, s>>>(da, nullptr, circular ? kPadCircular : kPadReplicate, d_nbr, d_eid);
    } else if (sampler == GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT) {
      if (circular) glx_filter_shuffle_kernel<<<rb, 256, 0, s>>>(da, res);
      glx_filter_pad_kernel<<<wb, 256, 0, s>>>(da, nullptr, circular ? kPadCircular : kPadReplicate, d_nbr, d_eid);
    } else if (!circular) {  // EdgeWeight / InDegree, replicate padding
      glx_filter_pad_kernel<<<wb, 256, 0, s>>>(da, nullptr, kPadReplicateDrawn, d_nbr, d_eid);
    } else {
      const bool by_weight = sampler == GLX_SAMPLER_EDGE_WEIGHT;
      glx_filter_alias_build_kernel<<<rb, 256, 0, s>>>(da, by_weight ? g->weight : nullptr, dm, g->dst_count, dist, tab,
                                                      stk);
      glx_filter_alias_build_wave_kernel<<<wb, 256, 0, s>>>(da, by_weight ? g->weight : nullptr, dm, g->dst_count, dist,
                                                           tab, stk);
      const int64_t total = (int64_t)nr * k;
      glx_filter_alias_slots_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(da, tab, d_nbr, d_eid);
    }
  }
  timer.stop();
  GLX_HIP(hipGetLastError());
  glx_scratch_trim(s, 2, (size_t)32 << 30);
  return GLX_OK;
}

// Entry of every filtered request.  id == value filters in front of Topk / RandomWithoutReplacement /
// EdgeWeight / InDegree take the closed-form path above; the few rows it does not serve (and every other
// filter) go through filtered_general.
int filtered_device(const glx_graph* g, int sampler, const int64_t* d_src, const int64_t* d_rng, int32_t batch,
                    int32_t k, const int64_t* d_offsets, int padding_mode, int64_t default_nbr, uint64_t seed,
                    uint64_t cc, FilterDev f, int64_t* d_nbr, int64_t* d_eid, hipStream_t s) {
  const bool id_equal = f.type == GLX_FILTER_EQUAL && f.field == GLX_FILTER_FIELD_ID;
  const bool served = sampler == GLX_SAMPLER_TOPK || sampler == GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT ||
                      sampler == GLX_SAMPLER_EDGE_WEIGHT || sampler == GLX_SAMPLER_IN_DEGREE;
  static const bool disabled = getenv("GLX_FILTER_NO_FAST_PATH") != nullptr;  // A/B and test knob
  if (!id_equal || !served || batch == 0 || k == 0 || disabled) {
    return filtered_general(g, sampler, d_src, d_rng, batch, k, d_offsets, padding_mode, default_nbr, seed, cc, f, d_nbr,
                            d_eid, s);
  }
  const bool circular = padding_mode == GLX_PAD_CIRCULAR;
  const size_t nb = (size_t)batch;
  // start | sub_src | sub_rng | sub_val (i64) ; deg | nhits | general | gidx | hit_lo (i32) ; hits (i32 x kMaxHits,
  // only without the index) ; count
  const bool indexed = g->nbr_sorted && g->slot_sorted;
  char* buf = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&buf),
                             nb * 4 * 8 + nb * 5 * 4 + (indexed ? 0 : nb * kMaxHits * 4) + 64, s, 4);
  if (rc != GLX_OK) return rc;
  int64_t* start = reinterpret_cast<int64_t*>(buf);
  int64_t* sub_src = start + nb;
  int64_t* sub_rng = sub_src + nb;
  int64_t* sub_val = sub_rng + nb;
  int32_t* deg = reinterpret_cast<int32_t*>(sub_val + nb);
  int32_t* nhits = deg + nb;
  int32_t* general = nhits + nb;
  int32_t* gidx = general + nb;
  int32_t* hit_lo = gidx + nb;
  int32_t* hits = hit_lo + nb;
  int32_t* count = hits + (indexed ? 0 : nb * kMaxHits);
  const unsigned row_blocks = (unsigned)((nb + 255) / 256);
  RowArgs ra{g->map(), g->row_ptr, g->adj, d_src, d_rng, batch};
  glx_filter_rows_kernel<<<row_blocks, 256, 0, s>>>(ra, start, deg, nullptr);
  HitArgs ha{g->adj, start, deg, f.values, g->nbr_sorted, g->slot_sorted, nhits, hit_lo, hits, batch};
  if (indexed) glx_filter_idhits_index_kernel<<<row_blocks, 256, 0, s>>>(ha);
  else glx_filter_idhits_scan_kernel<<<(unsigned)batch, 64, 0, s>>>(ha);
  FastArgs fa;
  fa.d = DrawArgs{g->adj, start, deg, nullptr, nullptr, nullptr, d_rng, seed, cc, default_nbr, batch, k, 0, batch, 0};
  fa.nhits = nhits;
  fa.hit_lo = hit_lo;
  fa.hits = hits;
  fa.slot_sorted = indexed ? g->slot_sorted : nullptr;
  fa.values = f.values;
  fa.general = general;
  fa.sampler = sampler;
  fa.circular = circular ? 1 : 0;
  glx_filter_classify_kernel<<<row_blocks, 256, 0, s>>>(fa);
  GLX_HIP(hipMemsetAsync(count, 0, 4, s));
  glx_filter_pack_general_kernel<<<row_blocks, 256, 0, s>>>(general, d_src, d_rng, f.values, batch, count, gidx, sub_src,
                                                            sub_rng, sub_val);
  int32_t G = 0;
  GLX_HIP(hipMemcpyAsync(&G, count, 4, hipMemcpyDeviceToHost, s));
  const bool alias = sampler == GLX_SAMPLER_EDGE_WEIGHT || sampler == GLX_SAMPLER_IN_DEGREE;
  const bool wide_rwor = sampler == GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT && circular && k > kFastMaxK;
  if (alias || wide_rwor) {
    // rows without a hit: the unfiltered sampler IS the answer (same reserved list, same weights, hence the
    // table built at load); rows with a hit are overwritten by the general path below
    rc = glx_sample_ex(g, sampler, d_src, d_rng, batch, k, padding_mode, default_nbr, seed, cc, d_nbr, d_eid,
                       GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
  } else if (sampler == GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT && circular) {
    glx_filter_fast_rwor_kernel<<<row_blocks, 256, 0, s>>>(fa, d_nbr, d_eid);
  } else {
    const int64_t total = (int64_t)batch * k;
    glx_filter_fast_topk_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(fa, d_nbr, d_eid);
  }
  GLX_HIP(hipGetLastError());
  GLX_HIP(hipStreamSynchronize(s));
  if (G > 0) {
    int64_t* sub_out = nullptr;
    rc = glx_scratch_alloc(reinterpret_cast<void**>(&sub_out), (size_t)G * k * 2 * 8, s, 5);
    if (rc != GLX_OK) return rc;
    FilterDev fs = f;
    fs.values = sub_val;
    rc = filtered_general(g, sampler, sub_src, sub_rng, G, k, nullptr, padding_mode, default_nbr, seed, cc, fs, sub_out,
                          sub_out + (size_t)G * k, s);
    if (rc != GLX_OK) return rc;
    const int64_t total = (int64_t)G * k;
    glx_filter_unpack_general_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(gidx, G, k, sub_out,
                                                                                    sub_out + (size_t)G * k, d_nbr, d_eid);
    GLX_HIP(hipGetLastError());
  }
  return GLX_OK;
}

int check_filter(const glx_graph* g, const glx_filter* filter) {
  GLX_REQUIRE(filter->type == GLX_FILTER_EQUAL || filter->type == GLX_FILTER_LARGER_THAN, "unknown filter type %d",
              filter->type);
  GLX_REQUIRE(filter->field >= GLX_FILTER_FIELD_NONE && filter->field <= GLX_FILTER_FIELD_TIMESTAMP,
              "unknown filter field %d", filter->field);
  GLX_REQUIRE(filter->values != nullptr, "filter values are NULL");
  (void)g;
  return GLX_OK;
}

// Stages host arrays next to each other in one slot-0 workspace; entries with a NULL source
// are outputs (or absent).
struct Staged {
  int64_t* d = nullptr;
  size_t used = 0;
  int64_t* take(size_t n) {
    int64_t* p = d + used;
    used += n;
    return p;
  }
};

}  // namespace

extern "C" int glx_graph_set_timestamps(glx_graph* g, const int64_t* ts_slot, int ptr_kind, void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE(ts_slot != nullptr || g->num_edges == 0, "timestamps are NULL");
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = glx_stream(stream);
  if (!g->ts) GLX_HIP(hipMalloc(&g->ts, (size_t)(g->num_edges > 0 ? g->num_edges : 1) * 8));
  if (g->num_edges > 0) {
    GLX_HIP(hipMemcpyAsync(g->ts, ts_slot, (size_t)g->num_edges * 8,
                           ptr_kind == GLX_PTR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

extern "C" int glx_sample_filtered(const glx_graph* g, int sampler, const int64_t* src, const int64_t* rng_rows,
                                   int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id,
                                   uint64_t seed, uint64_t call_counter, const glx_filter* filter,
                                   int64_t* nbr_out, int64_t* eid_out, int ptr_kind, void* stream) {
  if (filter == nullptr || filter->type == GLX_FILTER_NONE) {
    return glx_sample_ex(g, sampler, src, rng_rows, batch, k, padding_mode, default_neighbor_id, seed, call_counter,
                         nbr_out, eid_out, ptr_kind, stream);
  }
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(batch >= 0 && k >= 0, "negative batch / neighbor_count");
  GLX_REQUIRE(padding_mode == GLX_PAD_CIRCULAR || padding_mode == GLX_PAD_REPLICATE, "bad padding_mode %d",
              padding_mode);
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE(sampler >= GLX_SAMPLER_RANDOM && sampler <= GLX_SAMPLER_IN_DEGREE, "unknown sampler id %d", sampler);
  GLX_REQUIRE((int64_t)batch * k <= INT32_MAX, "batch * neighbor_count exceeds int32 (tensor.h:47)");
  if (batch == 0 || k == 0) return GLX_OK;
  GLX_REQUIRE(src && nbr_out && eid_out, "NULL data pointer");
  int rc = check_filter(g, filter);
  if (rc != GLX_OK) return rc;
  if (g->num_edges == 0) {
    // no row has a neighbour, filtered or not: the plain samplers write the default ids (and an empty shard of a
    // weighted type has no weight array to be recognised by)
    return glx_sample_ex(g, sampler, src, rng_rows, batch, k, padding_mode, default_neighbor_id, seed, call_counter, nbr_out,
                         eid_out, ptr_kind, stream);
  }
  GLX_REQUIRE(sampler != GLX_SAMPLER_EDGE_WEIGHT || g->weight != nullptr, "EdgeWeightSampler needs a weighted graph");
  GLX_REQUIRE(sampler != GLX_SAMPLER_IN_DEGREE || g->alias_indeg != nullptr,
              "InDegreeSampler needs glx_graph_enable_in_degree()");
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  FilterDev f{filter->type, filter->field, filter->values, g->ts, filter->default_timestamp, filter->retry_times};
  if (ptr_kind == GLX_PTR_DEVICE) {
    return filtered_device(g, sampler, src, rng_rows, batch, k, nullptr, padding_mode, default_neighbor_id, seed,
                           call_counter, f, nbr_out, eid_out, s);
  }
  GlxHostCallSlot admitted(g->device);
  const size_t nb = (size_t)batch, n_out = nb * (size_t)k;
  int64_t* m_nbr = static_cast<int64_t*>(glx_mapped_ptr(nbr_out, n_out * 8));  // pinned caller buffers are written directly
  int64_t* m_eid = static_cast<int64_t*>(glx_mapped_ptr(eid_out, n_out * 8));
  const bool direct = m_nbr != nullptr && m_eid != nullptr;
  Staged st;
  rc = glx_scratch_alloc(reinterpret_cast<void**>(&st.d), (nb * 3 + (direct ? 0 : n_out * 2)) * 8, s, 0);
  if (rc != GLX_OK) return rc;
  int64_t* d_src = st.take(nb);
  int64_t* d_val = st.take(nb);
  int64_t* d_rng = rng_rows ? st.take(nb) : nullptr;
  int64_t* d_nbr = direct ? m_nbr : st.take(n_out);
  int64_t* d_eid = direct ? m_eid : st.take(n_out);
  GLX_HIP(hipMemcpyAsync(d_src, src, nb * 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipMemcpyAsync(d_val, filter->values, nb * 8, hipMemcpyHostToDevice, s));
  if (rng_rows) GLX_HIP(hipMemcpyAsync(d_rng, rng_rows, nb * 8, hipMemcpyHostToDevice, s));
  f.values = d_val;
  rc = filtered_device(g, sampler, d_src, d_rng, batch, k, nullptr, padding_mode, default_neighbor_id, seed,
                       call_counter, f, d_nbr, d_eid, s);
  if (rc != GLX_OK) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  if (!direct) {
    GLX_HIP(hipMemcpyAsync(nbr_out, d_nbr, n_out * 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(eid_out, d_eid, n_out * 8, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

extern "C" int glx_sample_full_filtered(const glx_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                                        const int64_t* offsets, int padding_mode, int64_t default_neighbor_id,
                                        const glx_filter* filter, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                                        void* stream) {
  if (filter == nullptr || filter->type == GLX_FILTER_NONE) {
    return glx_sample_full(g, src, batch, max_limit, offsets, nbr_out, eid_out, ptr_kind, stream);
  }
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(batch >= 0, "negative batch");
  GLX_REQUIRE(padding_mode == GLX_PAD_CIRCULAR || padding_mode == GLX_PAD_REPLICATE, "bad padding_mode %d",
              padding_mode);
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  if (batch == 0) return GLX_OK;
  GLX_REQUIRE(src && offsets, "NULL data pointer");
  int rc = check_filter(g, filter);
  if (rc != GLX_OK) return rc;
  (void)max_limit;  // the segments in `offsets` already carry the truncation
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  FilterDev f{filter->type, filter->field, filter->values, g->ts, filter->default_timestamp, filter->retry_times};
  if (ptr_kind == GLX_PTR_DEVICE) {
    if (!nbr_out || !eid_out) {  // an empty response has no buffer to point at: fine when the (device) offsets agree
      int64_t total = -1;
      GLX_HIP(hipMemcpyAsync(&total, offsets + batch, 8, hipMemcpyDeviceToHost, s));
      GLX_HIP(hipStreamSynchronize(s));
      GLX_REQUIRE(total == 0, "NULL output pointer for a response of %lld values", (long long)total);
      return GLX_OK;
    }
    return filtered_device(g, kFullSampler, src, nullptr, batch, 0, offsets, padding_mode, default_neighbor_id, 0, 0,
                           f, nbr_out, eid_out, s);
  }
  const size_t nb = (size_t)batch;
  const int64_t total = offsets[batch];
  GLX_REQUIRE(total >= 0, "bad offsets");
  if (total == 0) return GLX_OK;
  GLX_REQUIRE(nbr_out && eid_out, "NULL output pointer");
  Staged st;
  rc = glx_scratch_alloc(reinterpret_cast<void**>(&st.d), (nb * 3 + 1 + (size_t)total * 2) * 8, s, 0);
  if (rc != GLX_OK) return rc;
  int64_t* d_src = st.take(nb);
  int64_t* d_val = st.take(nb);
  int64_t* d_off = st.take(nb + 1);
  int64_t* d_nbr = st.take((size_t)total);
  int64_t* d_eid = st.take((size_t)total);
  GLX_HIP(hipMemcpyAsync(d_src, src, nb * 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipMemcpyAsync(d_val, filter->values, nb * 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipMemcpyAsync(d_off, offsets, (nb + 1) * 8, hipMemcpyHostToDevice, s));
  f.values = d_val;
  rc = filtered_device(g, kFullSampler, d_src, nullptr, batch, 0, d_off, padding_mode, default_neighbor_id, 0, 0, f,
                       d_nbr, d_eid, s);
  if (rc != GLX_OK) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  GLX_HIP(hipMemcpyAsync(nbr_out, d_nbr, (size_t)total * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipMemcpyAsync(eid_out, d_eid, (size_t)total * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}
