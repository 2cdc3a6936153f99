// glx node features + segmented aggregation (Sum/Mean/Max/Min/Prod) + lookup.
// Replaces graphlearn/src/core/operator/aggregator/{aggregator.cc:25-86,
// sum_aggregator.cc:25-33, mean_aggregator.cc:26-61, max_aggregator.cc:26-40,
// min_aggregator.cc, prod_aggregator.cc} and the GetAttribute()->GetFloats()
// reads behind them (memory_node_storage.cc:127-138).
//
// SpMM-shaped but HBM-bound (0.25 flop/byte): the design goal is to keep many
// independent 16-byte row loads in flight per lane and to touch every feature
// byte once.  A group of G lanes owns one segment and walks its rows in the
// reference's order; lane c of the group owns columns [4c, 4c+4) so each output
// element is accumulated left-to-right exactly like the reference's serial loop
// (bit-identical results, no cross-lane reduction at all).
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>

#include "glx_common.h"

namespace {

// ---- segment bookkeeping ---------------------------------------------------
// AggregatingRequest's cursor (aggregating_request.cc:86-105) consumes ids in
// order and hands id c to segment segment_ids[c] only while the sequence stays
// non-decreasing and inside [0, num_segments); the first violation stalls the
// cursor for good.  valid_len = index of that first violation.
//
// One streaming pass over segment_ids does all of it (round 5; rounds 1-4 ran a validity pass and then one binary
// search per segment -- 24 dependent loads each -- which cost 0.1 ms on the headline's 16.4 M ids, 5 % of the step):
//   valid_len
//   level = 0  when the request is what a dense sampler response implies -- segment s = ids [s f, (s + 1) f),
//              f = n / num_segments -- and the reduce may use arithmetic segment bounds (no seg_start read);
//           1  when the reduce must read seg_start;
//           2  when the scan left seg_start incomplete and glx_seg_fixup_kernel fills it (the reduce reads it too)
//   seg_start[s] = lower_bound(seg[0..valid_len), s), s in [0, num_segments]: position i writes it for every s in
//                  (seg[i - 1], seg[i]] (the last position for the segments after seg[n - 1]).
// A violation, or a run of more than kSegRunMax empty segments (one thread would write them all), leaves the rest
// to the fix-up kernel: the old binary search, which exits at once otherwise.
// valid_len and level live in two 64-bit words of a small per-stream buffer, each tagged with the call's epoch in its
// high half and raised with atomicMax: a word whose tag is not this call's reads as "nothing raised", so no launch is
// spent on resetting them (a launch costs as much as the whole scan of a small request).
constexpr int32_t kSegRunMax = 64;

struct SegState {
  const unsigned long long* words;  // [0]: epoch << 32 | n - valid_len   [1]: epoch << 32 | level
  uint32_t epoch;
  int32_t floor_level;  // level the host already knows: 1 = the ids do not divide evenly, 2 = no ids at all
};

__device__ __forceinline__ int32_t seg_level(const SegState& st) {
  const unsigned long long w = st.words[1];
  const int32_t raised = (uint32_t)(w >> 32) == st.epoch ? (int32_t)(uint32_t)w : 0;
  return raised > st.floor_level ? raised : st.floor_level;
}
__device__ __forceinline__ int32_t seg_valid_len(const SegState& st, int32_t n) {
  const unsigned long long w = st.words[0];
  return (uint32_t)(w >> 32) == st.epoch ? n - (int32_t)(uint32_t)w : n;
}

__global__ void glx_seg_reset_kernel(unsigned long long* words) {  // captured launches only (see prepare_segments)
  words[0] = 0;
  words[1] = 0;
}

__global__ __launch_bounds__(256) void glx_seg_scan_kernel(const int32_t* __restrict__ seg, int32_t n, int32_t num_segments,
                                                           int32_t f, unsigned long long* __restrict__ words, uint32_t epoch,
                                                           int32_t* __restrict__ seg_start) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  int32_t v[4];
  const int32_t m = (n - i0) < 4 ? (int32_t)(n - i0) : 4;
  if (m == 4 && (reinterpret_cast<uintptr_t>(seg) & 15) == 0) {
    const int4 q = *reinterpret_cast<const int4*>(seg + i0);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
    for (int j = 0; j < 4; ++j) v[j] = j < m ? seg[i0 + j] : 0;
  }
  int32_t prev = i0 > 0 ? seg[i0 - 1] : -1;
  // the id before this thread's is itself out of range: a violation at or before it (its own thread reports it), so
  // nothing from here on belongs to the request -- and `prev` is no segment to count on from
  if (i0 > 0 && (prev < 0 || prev >= num_segments)) return;
  int32_t quo = f > 0 ? (int32_t)(i0 / f) : 0, rem = f > 0 ? (int32_t)(i0 % f) : 0;  // i / f and i % f, carried along
  bool nonuniform = false, incomplete = false;
  int32_t bad_at = n;
  for (int j = 0; j < m; ++j) {
    const int32_t i = (int32_t)i0 + j, cur = v[j];
    if (cur < 0 || cur >= num_segments || cur < prev) {  // prev == -1 before the first id
      bad_at = i;
      break;
    }
    if (cur != prev) {
      if (cur - prev > kSegRunMax) incomplete = true;
      else for (int32_t s = prev + 1; s <= cur; ++s) seg_start[s] = i;
    }
    if (f > 0) {
      nonuniform |= cur != quo;
      if (++rem == f) { rem = 0; ++quo; }
    }
    prev = cur;
  }
  if (bad_at == n && i0 + m == n) {  // the last ids: the segments after the last one used start (and end) at n
    if (num_segments - prev > kSegRunMax) incomplete = true;
    else for (int32_t s = prev + 1; s <= num_segments; ++s) seg_start[s] = n;
  }
  const unsigned long long tag = (unsigned long long)epoch << 32;
  if (bad_at < n) atomicMax(&words[0], tag | (uint32_t)(n - bad_at));  // the largest n - i = the first violation
  const int32_t level = (incomplete || bad_at < n) ? 2 : (nonuniform ? 1 : 0);
  // One atomic per WAVE, and none once the word already says as much (round 6): a ragged request whose length divides
  // evenly raises level 1 from nearly every thread -- 4 M atomics on one address for the headline's 16.4 M ids, which
  // queue at the memory side: +0.78 ms on a 3.6 ms aggregate (profiles/r06/seg_ragged_probe.txt).  The word only grows,
  // so a (coherent) read that already shows this wave's level or more makes the atomic redundant.
  // (ballots see the lanes still here: threads past the end of the request, or behind an out-of-range id, have left)
  const uint64_t here = __ballot(true), at2 = __ballot(level == 2), at1 = __ballot(level >= 1);
  const int32_t wave_level = at2 ? 2 : (at1 ? 1 : 0);
  if (wave_level && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)here) - 1)) {
    const unsigned long long mine = tag | (uint32_t)wave_level;
    if (__hip_atomic_load(&words[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(&words[1], mine);
  }
}

// seg_start[s] = lower_bound(seg[0..valid_len), s) for every s -- only when the scan left the table incomplete.
__global__ void glx_seg_fixup_kernel(const int32_t* __restrict__ seg, SegState st, int32_t n, int32_t num_segments,
                                     int32_t* __restrict__ seg_start) {
  if (seg_level(st) != 2) return;
  int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > num_segments) return;
  int32_t lo = 0, hi = seg_valid_len(st, n);
  while (lo < hi) {
    int32_t mid = lo + ((hi - lo) >> 1);
    if (seg[mid] < s) lo = mid + 1; else hi = mid;
  }
  seg_start[s] = lo;
}

// raw id -> feature row (int32, -1 = unknown id) for the hashed id map.
__global__ void glx_rows_kernel(GlxIdMap map, const int64_t* __restrict__ ids, int64_t n,
                                int32_t* __restrict__ rows) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  rows[i] = (int32_t)glx_row_of(map, ids[i]);
}

template <int OP>
__device__ __forceinline__ float agg_init() {
  if (OP == GLX_AGG_MAX) return (float)FLT_MIN_10_EXP;  // max_aggregator.cc:28 (-37, sic)
  if (OP == GLX_AGG_MIN) return FLT_MAX;                // min_aggregator.cc:28
  if (OP == GLX_AGG_PROD) return 1.0f;                  // prod_aggregator.cc
  return 0.0f;                                          // aggregator.cc:61-65
}

template <int OP>
__device__ __forceinline__ float agg_combine(float l, float r) {
  if (OP == GLX_AGG_MAX) return (l < r) ? r : l;  // std::max(l, r)
  if (OP == GLX_AGG_MIN) return (r < l) ? r : l;  // std::min(l, r)
  if (OP == GLX_AGG_PROD) return l * r;
  return l + r;  // sum, mean
}

struct AggArgs {
  const float* X;
  const int64_t* node_ids;   // raw ids (dense map) ...
  const int32_t* rows;       // ... or pre-translated rows (hashed map / multi-source); one is null
  const int32_t* seg_start;  // [num_segments + 1], or null: uniform segments of `fanout` ids
  SegState seg_state;        // with seg_start: level 0 -> the segments ARE uniform (`fanout` ids each), seg_start is not read
  float* emb_out;
  int32_t* cnt_out;
  int64_t num_rows;
  int64_t stride;
  int64_t swizzle_rows;
  int32_t dim;
  int32_t num_segments;
  int32_t fanout;
  float default_attr;
  int32_t col0, ncols;       // the columns [col0, col0 + ncols) this launch reduces (a column slice, or all)
  int32_t store_mode;        // glx_aggregate_grp_kernel: 0 non-temporal output stores (default), 1 plain stores (ablation)
  int32_t segs_per_group;    // glx_aggregate_grp_kernel: consecutive segments one lane group reduces
  int32_t xcd_slices;        // glx_aggregate_grp_kernel: > 1 = workgroup b reduces column slice b % xcd_slices (ncols each)
  // further row sources of the distributed store (glx_dist.hip): virtual row r lives in
  // source 0 when r < base1, in source 1 (the hot-row replica) when r < base2, else in
  // source 2 (the halo rows of this request, plain row-major, no swizzle).
  const float* X1;
  const float* X2;
  int64_t stride1, swizzle1, stride2;
  int32_t base1, base2;
};

// Feature rows are < 2^31 (checked at creation), so a row index fits an int32 --
// half the registers of the raw int64 id.
__device__ __forceinline__ int32_t agg_row_at(const AggArgs& a, int32_t pos) {
  if (a.rows) return a.rows[pos];
  const int64_t id = a.node_ids[pos];
  return (id >= 0 && id < a.num_rows) ? (int32_t)id : -1;
}

template <int NSRC>
__device__ __forceinline__ const float* agg_row_ptr(const AggArgs& a, int32_t row) {
  if (NSRC == 1 || row < a.base1) return a.X + glx_swizzle_row(row, a.swizzle_rows) * a.stride;
  if (row < a.base2) return a.X1 + glx_swizzle_row(row - a.base1, a.swizzle1) * a.stride1;
  return a.X2 + (int64_t)(row - a.base2) * a.stride2;
}

// G lanes per segment, VEC floats per lane per pass (VEC = 4: one dwordx4 per
// row per lane; VEC = 1 for dims that are not a multiple of 4).  U rows are
// issued back-to-back before the first is consumed.  NSRC = 3: rows come from the
// three sources of a distributed store.
template <int OP, int G, int VEC, int U, int NSRC>
__global__ __launch_bounds__(256) void glx_aggregate_kernel(AggArgs a) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int c = threadIdx.x & (G - 1);
  if (gid >= a.num_segments) return;
  int32_t s0, s1;
  if (a.seg_start && seg_level(a.seg_state) != 0) {
    s0 = a.seg_start[gid];
    s1 = a.seg_start[gid + 1];
  } else {  // a dense sampler response: segment gid = ids [gid * fanout, (gid + 1) * fanout)
    s0 = (int32_t)gid * a.fanout;
    s1 = s0 + a.fanout;
  }
  const int32_t n = s1 - s0;
  if (c == 0 && a.col0 == 0) a.cnt_out[gid] = n;
  float* out = a.emb_out + gid * (int64_t)a.dim;
  const int32_t col_end = a.col0 + a.ncols;
  for (int32_t col = a.col0 + c * VEC; col < col_end; col += G * VEC) {
    vec_t acc;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = agg_init<OP>();
    for (int32_t base = s0; base < s1; base += U) {
      int32_t row[U];
#pragma unroll
      for (int u = 0; u < U; ++u) row[u] = (base + u < s1) ? agg_row_at(a, base + u) : -2;
      vec_t val[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (row[u] >= 0) {
          val[u] = *reinterpret_cast<const vec_t*>(agg_row_ptr<NSRC>(a, row[u]) + col);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) val[u][v] = a.default_attr;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (row[u] != -2) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = agg_combine<OP>(acc[v], val[u][v]);
        }
      }
    }
    // FinalFunc: aggregator.cc:74-86 (empty -> default), mean_aggregator.cc:45-61.
    if (n == 0) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = a.default_attr;
    } else if (OP == GLX_AGG_MEAN) {
      const float fn = (float)n;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = acc[v] / fn;
    }
    *reinterpret_cast<vec_t*>(out + col) = acc;
  }
}

// ---- the wide shapes (16-byte loads, 8..64 lanes per segment): ids fetched ONCE, coalesced ------------------
// glx_aggregate_kernel above asks for every id with a load of its own in which all lanes of a group read the same
// address, each followed by a wait: a fanout-10 segment is ten dependent round trips before the first row load
// issues.  Here a group of G lanes keeps a CHUNK of G * IDR consecutive positions of its id
// range in registers -- lane c holds positions chunk_base + k G + c, one coalesced load per register -- already
// translated to feature rows, and hands row `pos` to all its lanes with a cross-lane read: v_readlane for a whole
// wave (the row index, the swizzle, the multiply by the pitch and the bounds test then run on the scalar unit and
// the row load takes its base address from SGPRs), ds_bpermute for the narrower groups.  A group reduces S
// consecutive segments one after the other so a chunk serves several of them (S f <= chunk for the dense sampler
// responses); up to U row loads are issued back to back before the first is consumed.  Accumulation order is
// unchanged: lane c owns columns [4c, 4c + 4) of its group's segment and folds the rows left to right
// (aggregator.cc:45-56), so results stay bit-identical to the reference's serial loop.
template <int G, int IDR>
__device__ __forceinline__ void agg_chunk_load(const AggArgs& a, int32_t chunk_base, int32_t pos_end, int c,
                                               int32_t (&myrow)[IDR]) {
#pragma unroll
  for (int k = 0; k < IDR; ++k) {
    const int32_t pos = chunk_base + k * G + c;
    myrow[k] = pos < pos_end ? agg_row_at(a, pos) : -1;
  }
}

// row held for chunk slot idx in [0, G * IDR) (group-uniform idx)
template <int G, int IDR>
__device__ __forceinline__ int32_t agg_chunk_get(const int32_t (&myrow)[IDR], int32_t idx) {
  if (G == 64) {
    int32_t r = __builtin_amdgcn_readlane(myrow[0], idx & 63);
#pragma unroll
    for (int k = 1; k < IDR; ++k) {
      const int32_t t = __builtin_amdgcn_readlane(myrow[k], idx & 63);
      r = (idx >> 6) == k ? t : r;
    }
    return r;
  }
  int32_t r = __shfl(myrow[0], idx & (G - 1), G);
#pragma unroll
  for (int k = 1; k < IDR; ++k) {
    const int32_t t = __shfl(myrow[k], idx & (G - 1), G);
    r = (idx / G) == k ? t : r;
  }
  return r;
}

// Row pointer with 32-bit row arithmetic (rows < 2^31, pitch < 2^31 floats): for a wave-uniform row this is a
// handful of SALU instructions.
__device__ __forceinline__ uint32_t agg_swizzle32(uint32_t r, uint32_t swizzle_rows) {
  const uint32_t m = ((r >> GLX_SWIZZLE_BITS) * 0x9E3779B1u) >> (32 - GLX_SWIZZLE_BITS);
  return r < swizzle_rows ? r ^ m : r;
}

template <int NSRC>
__device__ __forceinline__ const float* agg_row_ptr32(const AggArgs& a, int32_t row) {
  if (NSRC == 1 || row < a.base1) {
    return a.X + (uint64_t)agg_swizzle32((uint32_t)row, (uint32_t)a.swizzle_rows) * (uint32_t)a.stride;
  }
  if (row < a.base2) {
    return a.X1 + (uint64_t)agg_swizzle32((uint32_t)(row - a.base1), (uint32_t)a.swizzle1) * (uint32_t)a.stride1;
  }
  return a.X2 + (uint64_t)(uint32_t)(row - a.base2) * (uint32_t)a.stride2;
}

// One batch: rows of chunk slots [slot0, slot0 + count), count <= U (kFull: count == U, no tests at all), all
// loads issued before the first is folded into acc, in slot order.
template <int OP, int G, int VEC, int U, int NSRC, int IDR, bool kFull>
__device__ __forceinline__ void agg_grp_batch(const AggArgs& a, const int32_t (&myrow)[IDR], int32_t slot0, int32_t count,
                                              uint32_t col_ld, float __attribute__((ext_vector_type(VEC)))& acc) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  int32_t row[U];
#pragma unroll
  for (int u = 0; u < U; ++u) row[u] = agg_chunk_get<G, IDR>(myrow, slot0 + u);
  vec_t val[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (kFull || u < count) {
      // an unknown id (row -1) reads row 0 and is replaced below: no divergent branch around the load
      const int32_t r = row[u] >= 0 ? row[u] : 0;
      val[u] = *reinterpret_cast<const vec_t*>(agg_row_ptr32<NSRC>(a, r) + col_ld);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (kFull || u < count) {
      vec_t x = val[u];
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[v] = row[u] < 0 ? a.default_attr : x[v];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = agg_combine<OP>(acc[v], x[v]);
    }
  }
}

template <int OP, int G, int VEC, int U, int NSRC, int IDR>
__global__ __launch_bounds__(256) void glx_aggregate_grp_kernel(AggArgs a) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  constexpr int kChunk = G * IDR;
  static_assert(U <= kChunk, "a batch must fit the chunk");
  constexpr int kGroupsPerBlock = 256 / G;
  // XCD-affine column slices (a.xcd_slices > 1): workgroup b runs on XCD b % 8 (observed placement, used for speed
  // only), so slice = b % xcd_slices keeps one column slice of EVERY row in one XCD's L2
  const int32_t slice = a.xcd_slices > 1 ? (int32_t)(blockIdx.x % (unsigned)a.xcd_slices) : 0;
  const int64_t blk = a.xcd_slices > 1 ? blockIdx.x / (unsigned)a.xcd_slices : blockIdx.x;
  int64_t grp = blk * kGroupsPerBlock + threadIdx.x / G;
  const int c = threadIdx.x & (G - 1);
  const int32_t S = a.segs_per_group;
  int64_t seg64 = grp * S;
  if (seg64 >= a.num_segments) return;
  int32_t seg_first = (int32_t)seg64;
  if (G == 64) seg_first = __builtin_amdgcn_readfirstlane(seg_first);
  const int32_t seg_last = (a.num_segments - seg_first) < S ? a.num_segments : seg_first + S;
  // segment boundaries: arithmetic for a dense sampler response, else lane j of the group holds start[seg_first + j]
  // (an explicit segment_ids tensor that spells out the dense response -- flags[1] == 0 -- takes the arithmetic too)
  const bool use_starts = a.seg_start != nullptr && seg_level(a.seg_state) != 0;
  int32_t mystart = 0;
  if (use_starts) mystart = a.seg_start[seg_first + (c <= seg_last - seg_first ? c : 0)];
  const int32_t starts[1] = {mystart};
  const int32_t pos_first = use_starts ? agg_chunk_get<G, 1>(starts, 0) : seg_first * a.fanout;
  const int32_t pos_end = use_starts ? agg_chunk_get<G, 1>(starts, seg_last - seg_first) : seg_last * a.fanout;
  const int32_t col_lo = a.col0 + slice * a.ncols;
  const int32_t col_end = col_lo + a.ncols;
  for (int32_t col_pass = col_lo; col_pass < col_end; col_pass += G * VEC) {
    const int32_t col = col_pass + c * VEC;
    const bool col_ok = col < col_end;
    const uint32_t col_ld = col_ok ? (uint32_t)col : (uint32_t)col_lo;  // lanes past the end re-read lane 0's columns, unused
    int32_t chunk_base = pos_first;
    int32_t myrow[IDR];
    agg_chunk_load<G, IDR>(a, chunk_base, pos_end, c, myrow);
    for (int32_t sg = seg_first; sg < seg_last; ++sg) {
      int32_t s0, s1;
      if (use_starts) {
        s0 = agg_chunk_get<G, 1>(starts, sg - seg_first);
        s1 = agg_chunk_get<G, 1>(starts, sg - seg_first + 1);
      } else {
        s0 = sg * a.fanout;
        s1 = s0 + a.fanout;
      }
      vec_t acc;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = agg_init<OP>();
      for (int32_t base = s0; base < s1; base += U) {
        const int32_t stop = (s1 - base) < U ? s1 : base + U;
        if (stop > chunk_base + kChunk) {  // the batch runs past the chunk: next chunk starts at this batch
          chunk_base = base;
          agg_chunk_load<G, IDR>(a, chunk_base, pos_end, c, myrow);
        }
        if (stop - base == U) agg_grp_batch<OP, G, VEC, U, NSRC, IDR, true>(a, myrow, base - chunk_base, U, col_ld, acc);
        else agg_grp_batch<OP, G, VEC, U, NSRC, IDR, false>(a, myrow, base - chunk_base, stop - base, col_ld, acc);
      }
      // FinalFunc: aggregator.cc:74-86 (empty -> default), mean_aggregator.cc:45-61.
      const int32_t n = s1 - s0;
      if (n == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = a.default_attr;
      } else if (OP == GLX_AGG_MEAN) {
        const float fn = (float)n;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = acc[v] / fn;
      }
      if (col_ok) {
        vec_t* dst = reinterpret_cast<vec_t*>(a.emb_out + sg * (int64_t)a.dim + col);
        // the outputs are written once and never read by this launch: the non-temporal hint (round 4, A/B in
        // profiles/r04/agg_probe_run8_nt_stores.txt: uniform rows -2.4 .. -5 %, the power-law requests -1 %, L2-resident -12 .. -20 %;
        // `sc1` write-through stores: no gain).  GLX_AGG_STORE=1 restores plain stores.
        if (a.store_mode == 1) *dst = acc;
        else __builtin_nontemporal_store(acc, dst);
      }
      if (c == 0 && col_pass == 0) a.cnt_out[sg] = n;
    }
  }
}

// ---- north_star's "dense feature tile staged in LDS, MFMA on the tile" formulation of Sum / Mean, kept as a
// measured ABLATION (GLX_AGG_MFMA=1; profiles/r03/mfma_ablation.txt) ----------------------------
// A segmented sum is the SpMM  Out[Sg, D] = S[Sg, N] * Xg[N, D]  with S the 0/1 segment-indicator matrix.  For a
// dense sampler response (uniform fanout f) a workgroup owns 16 consecutive segments = 16 f consecutive ids:
//   * the 4 waves gather KC rows at a time into LDS with 16-byte loads (row pitch D + 16 floats: the four rows
//     of one MFMA step then sit in four disjoint groups of 16 banks);
//   * wave w owns the columns [w D/4, (w+1) D/4) as NB = D/64 blocks of 16 and issues, per 4 rows,
//     v_mfma_f32_16x16x4_f32 with A[m][k] = (row k belongs to segment m) and B[k][n] = the staged row;
//   * f32 MFMA is an exact fmaf chain over k in order and the indicator is 0 or 1, so every output element
//     is still accumulated in the reference's left-to-right order (sum_aggregator.cc:25-33): bit-identical.
// 15/16 of the multiply-adds multiply by zero, f32 MFMA runs at the vector rate (157 TF), and the reduce is
// 0.25 flop/byte: the matrix core cannot make it faster than the HBM stream -- the ablation measures what the
// LDS round trip and the workgroup barriers cost instead.
template <int OP, int NB>
__global__ __launch_bounds__(256) void glx_aggregate_mfma_kernel(AggArgs a) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int KC = 32;  // rows staged per round
  extern __shared__ float tile[];  // [KC][pitch]
  constexpr int32_t D = 64 * NB;
  constexpr int32_t pitch = D + 16;
  const int32_t f = a.fanout;
  const int64_t seg0 = (int64_t)blockIdx.x * 16;
  const int32_t nseg = (int32_t)((a.num_segments - seg0) < 16 ? (a.num_segments - seg0) : 16);
  const int64_t id0 = seg0 * f;
  const int32_t nrows = nseg * f;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m = lane & 15;   // A: segment row of this lane / B, C: column inside the block
  const int kq = lane >> 4;  // A, B: which of the step's 4 rows / C: row group
  f4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f4{0.f, 0.f, 0.f, 0.f};
  constexpr int32_t row_f4 = D / 4;
  constexpr int32_t PT = KC * row_f4 / 256;  // 16-byte pieces per thread per round (2 NB)
  for (int32_t base = 0; base < nrows; base += KC) {
    // stage rows [base, base + KC): thread t moves pieces t, t + 256, ...; all loads issued before the first store
    f4 v[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int32_t p = threadIdx.x + i * 256;
      const int32_t r = p / row_f4;
      const int32_t c4 = p - r * row_f4;
      v[i] = f4{0.f, 0.f, 0.f, 0.f};
      if (base + r < nrows) {
        const int32_t row = agg_row_at(a, (int32_t)(id0 + base + r));
        if (row >= 0) v[i] = *reinterpret_cast<const f4*>(agg_row_ptr<1>(a, row) + c4 * 4);
        else v[i] = f4{a.default_attr, a.default_attr, a.default_attr, a.default_attr};
      }
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int32_t p = threadIdx.x + i * 256;
      const int32_t r = p / row_f4;
      const int32_t c4 = p - r * row_f4;
      *reinterpret_cast<f4*>(tile + r * pitch + c4 * 4) = v[i];
    }
    __syncthreads();
#pragma unroll 2
    for (int32_t k4 = 0; k4 < KC; k4 += 4) {
      const int32_t krow = base + k4 + kq;
      const float ind = (krow < nrows && krow / f == m) ? 1.0f : 0.0f;
      const float* brow = tile + (k4 + kq) * pitch + wave * (D / 4) + m;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind, brow[b * 16], acc[b], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int32_t srow = kq * 4 + reg;
    if (srow < nseg) {
      float* out = a.emb_out + (seg0 + srow) * (int64_t)D + wave * (D / 4) + m;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float v = acc[b][reg];
        if (f == 0) v = a.default_attr;
        else if (OP == GLX_AGG_MEAN) v = v / (float)f;
        out[b * 16] = v;
      }
    }
  }
  if (threadIdx.x < nseg) a.cnt_out[seg0 + threadIdx.x] = f;
}

// ---- experiment knobs ------------------------------------------------------------------------------------
// Read from the environment ONCE (first launch in the process; Process() runs on up to 32 pool threads and
// getenv is not something to call per launch), and settable at run time through glx_tune() so a probe can A/B
// kernels inside one process.  Every knob is an ablation / measurement aid: 0 = the product's default.
struct AggKnobs {
  std::atomic<int> mfma{0};     // GLX_AGG_MFMA=1: the LDS-staged MFMA formulation of Sum / Mean (ablation)
  std::atomic<int> unroll{0};   // GLX_AGG_UNROLL: rows in flight per lane (0 = chosen from the fanout)
  std::atomic<int> slices{0};   // GLX_AGG_SLICES=2|4|8: column slices as consecutive launches (ablation)
  std::atomic<int> legacy{0};   // GLX_AGG_LEGACY=1: the round-1..3 kernel (one id load per row per lane)
  std::atomic<int> segs{0};     // GLX_AGG_SEGS: segments per lane group (0 = chosen from fanout and grid size)
  std::atomic<int> xcd{0};      // GLX_AGG_XCD_SLICES=1|2|4|8: column slice = workgroup % n (XCD-affine); 0 = 2 for big requests
  std::atomic<int> occ{0};      // GLX_AGG_OCCUPANCY=3..7: workgroups per CU, capped with an unused LDS allocation
  std::atomic<int> store{0};    // GLX_AGG_STORE=1: plain output stores instead of non-temporal ones (ablation)
};

AggKnobs& agg_knobs() {
  static AggKnobs k;
  static std::once_flag once;
  std::call_once(once, [] {
    auto env = [](const char* name) {
      const char* e = getenv(name);
      return e ? atoi(e) : 0;
    };
    k.mfma = env("GLX_AGG_MFMA");
    k.unroll = env("GLX_AGG_UNROLL");
    k.slices = env("GLX_AGG_SLICES");
    k.legacy = env("GLX_AGG_LEGACY");
    k.segs = env("GLX_AGG_SEGS");
    k.xcd = env("GLX_AGG_XCD_SLICES");
    k.occ = env("GLX_AGG_OCCUPANCY");
    k.store = env("GLX_AGG_STORE");
  });
  return k;
}

bool agg_use_mfma(const AggArgs& a, int op) {
  if (agg_knobs().mfma.load(std::memory_order_relaxed) == 0) return false;
  return (op == GLX_AGG_SUM || op == GLX_AGG_MEAN) && a.seg_start == nullptr && a.fanout > 0 && a.X1 == nullptr &&
         a.X2 == nullptr && (a.dim == 64 || a.dim == 128 || a.dim == 256) && (a.stride % 4) == 0 &&
         (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
}

template <int OP>
void launch_agg_mfma(const AggArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.num_segments + 15) / 16);
  const size_t lds = (size_t)32 * (a.dim + 16) * sizeof(float);
  if (a.dim == 64) glx_aggregate_mfma_kernel<OP, 1><<<grid, 256, lds, s>>>(a);
  else if (a.dim == 128) glx_aggregate_mfma_kernel<OP, 2><<<grid, 256, lds, s>>>(a);
  else glx_aggregate_mfma_kernel<OP, 4><<<grid, 256, lds, s>>>(a);
}

// ---- legacy kernel launch (narrow / unaligned shapes, and GLX_AGG_LEGACY=1) --------------------------------
// Rows in flight per lane of glx_aggregate_kernel: A/B on the C3 hop-2 request (profiles/r02/agg_unroll_probe.txt)
// 3 / 4 / 5 / 6 / 8 / 10 / 12 rows -> 2.26 / 2.19 / 2.11 / 2.10 / 2.21 / 2.49 / 2.48 ms; wide float4 shapes 6, narrow 8.
template <int OP, int G, int VEC, int NSRC>
void launch_agg_g(const AggArgs& a, hipStream_t s) {
  const int64_t threads = (int64_t)a.num_segments * G;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  constexpr bool kWide = VEC == 4 && G >= 32;
  glx_aggregate_kernel<OP, G, VEC, kWide ? 6 : 8, NSRC><<<grid, 256, 0, s>>>(a);
}

// ---- grouped kernel launch ---------------------------------------------------------------------------------
// Rows in flight per lane (U): the fewest batches per segment wins, then the smaller batch (registers):
// fanout 10 -> 10, 15 -> 15, 25 -> 15 (15 + 10), 20 -> 10, 5 -> 5 of 6.  GLX_AGG_UNROLL overrides.
int agg_grp_unroll(int32_t fanout, int32_t avg_len) {
  static const int kU[] = {6, 8, 10, 12, 15};
  const int want = agg_knobs().unroll.load(std::memory_order_relaxed);
  for (int u : kU) {
    if (u == want) return u;
  }
  const int32_t f = fanout > 0 ? fanout : (avg_len > 0 ? avg_len : 10);
  int best = 10, best_batches = INT32_MAX;
  for (int u : kU) {
    const int batches = (f + u - 1) / u;
    if (batches < best_batches) {
      best = u;
      best_batches = batches;
    }
  }
  return best;
}

template <int OP, int G, int VEC, int NSRC, int IDR>
void launch_agg_grp(AggArgs a, int32_t num_ids, hipStream_t s) {
  const int32_t avg = a.num_segments > 0 ? (int32_t)(num_ids / a.num_segments) : 0;
  // segments per group: one.  More (a chunk of ids then serves several segments: one id load per G * IDR / fanout
  // segments) lowers the L2-resident floor but loses on every real request (profiles/r04/agg_probe_*): more, shorter
  // waves keep more independent row loads in flight.  GLX_AGG_SEGS overrides.
  int32_t S = 1;
  const int want = agg_knobs().segs.load(std::memory_order_relaxed);
  // ... except over a table small enough that most of a big request is served on-die (<= 1 GiB = four Infinity Caches;
  // round 6): such a launch is not waiting for HBM but for its own vector-memory pipeline -- C5's item -> shop reduce
  // (6.55 M ids over the 1 GB shop table: three quarters of its row reads hit L1, the per-CU TA/TCP/TD path busy ~95 % of the
  // launch, profiles/r06/c5_is_reduce.md) -- and there three segments per group (one id chunk serves all three, a
  // third of the waves to launch and drain) took 0.49 -> 0.43 ms, while the same setting LOSES on the tables HBM
  // serves: C2 (1.2 GB) 0.55 -> 0.63, C4 0.98 -> 1.08, C3 (10 GB) 1.84 -> 1.84 (profiles/r06/agg_probe_*_segs.txt).
  // Small requests stay at one (a group per segment fills the chip sooner).
  if (NSRC == 1 && (int64_t)a.num_rows * a.stride * 4 <= ((int64_t)1 << 30) && a.num_segments >= 3 * 32768) S = 3;
  if (want > 0) S = want;
  if (S > G - 1) S = G - 1;  // lane j of the group holds the start of its j-th segment (and lane S the end)
  if (S < 1) S = 1;
  a.segs_per_group = S;
  a.store_mode = agg_knobs().store.load(std::memory_order_relaxed);
  const int64_t groups = ((int64_t)a.num_segments + S - 1) / S;
  const int64_t blocks = (groups + (256 / G) - 1) / (256 / G) * (a.xcd_slices > 1 ? a.xcd_slices : 1);
  const unsigned grid = (unsigned)blocks;
  // occupancy cap (experiment): k workgroups per CU by declaring 160 KiB / k of LDS nobody touches
  const int occ = agg_knobs().occ.load(std::memory_order_relaxed);
  const size_t lds = (occ >= 3 && occ <= 7) ? (size_t)(160 * 1024 / occ) & ~(size_t)255 : 0;
  switch (agg_grp_unroll(a.seg_start ? 0 : a.fanout, avg)) {
    case 6: glx_aggregate_grp_kernel<OP, G, VEC, 6, NSRC, IDR><<<grid, 256, lds, s>>>(a); break;
    case 8: glx_aggregate_grp_kernel<OP, G, VEC, 8, NSRC, IDR><<<grid, 256, lds, s>>>(a); break;
    case 12: glx_aggregate_grp_kernel<OP, G, VEC, 12, NSRC, IDR><<<grid, 256, lds, s>>>(a); break;
    case 15: glx_aggregate_grp_kernel<OP, G, VEC, 15, NSRC, IDR><<<grid, 256, lds, s>>>(a); break;
    default: glx_aggregate_grp_kernel<OP, G, VEC, 10, NSRC, IDR><<<grid, 256, lds, s>>>(a); break;
  }
}

template <int OP, int NSRC>
void launch_agg_cols(const AggArgs& a0, int32_t num_ids, int want_xcd, hipStream_t s) {
  AggArgs a = a0;
  bool vec4 = a.dim % 4 == 0 && a.ncols % 4 == 0 && a.col0 % 4 == 0 && (reinterpret_cast<uintptr_t>(a.emb_out) & 15) == 0 &&
              (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 && (a.stride % 4) == 0;
  if (NSRC > 1) {
    vec4 = vec4 && (reinterpret_cast<uintptr_t>(a.X1) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.X2) & 15) == 0 &&
           (a.stride1 % 4) == 0 && (a.stride2 % 4) == 0;
  }
  // the grouped kernel reads row 0 in place of an unknown id's row: some row must exist
  const bool has_rows = NSRC > 1 ? (a.X2 != nullptr || a.base2 > 0) : a.num_rows > 0;
  if (vec4 && has_rows && a.ncols >= 32 && agg_knobs().legacy.load(std::memory_order_relaxed) == 0) {
    if (want_xcd > 1 && a.col0 == 0 && a.ncols == a.dim && a.dim % (4 * want_xcd) == 0 && a.dim / want_xcd >= 32) {
      a.xcd_slices = want_xcd;  // only the grouped kernel knows about slices
      a.ncols = a.dim / want_xcd;
    }
    // lanes per segment, 16-byte loads each.  (8-byte loads over twice the lanes -- a whole wave per 128-column slice
    // keeps the scalar row path -- were measured and dropped: equal on uniform rows, 13 % slower on the power-law
    // request; profiles/r04/agg_probe_*_run4.txt.  The kernel keeps its VEC parameter.)
    const int lanes = a.ncols / 4;
    if (lanes >= 64) launch_agg_grp<OP, 64, 4, NSRC, 1>(a, num_ids, s);
    else if (lanes >= 32) launch_agg_grp<OP, 32, 4, NSRC, 1>(a, num_ids, s);
    else if (lanes >= 16) launch_agg_grp<OP, 16, 4, NSRC, 1>(a, num_ids, s);
    else launch_agg_grp<OP, 8, 4, NSRC, 2>(a, num_ids, s);
    return;
  }
  if (vec4) {
    const int lanes = a.ncols / 4;
    if (lanes >= 64) launch_agg_g<OP, 64, 4, NSRC>(a, s);
    else if (lanes >= 32) launch_agg_g<OP, 32, 4, NSRC>(a, s);
    else if (lanes >= 16) launch_agg_g<OP, 16, 4, NSRC>(a, s);
    else if (lanes >= 8) launch_agg_g<OP, 8, 4, NSRC>(a, s);
    else if (lanes >= 4) launch_agg_g<OP, 4, 4, NSRC>(a, s);
    else if (lanes >= 2) launch_agg_g<OP, 2, 4, NSRC>(a, s);
    else launch_agg_g<OP, 1, 4, NSRC>(a, s);
  } else {
    const int lanes = a.ncols;
    if (lanes >= 64) launch_agg_g<OP, 64, 1, NSRC>(a, s);
    else if (lanes >= 16) launch_agg_g<OP, 16, 1, NSRC>(a, s);
    else if (lanes >= 4) launch_agg_g<OP, 4, 1, NSRC>(a, s);
    else launch_agg_g<OP, 1, 1, NSRC>(a, s);
  }
}

// Column slices.
//  * XCD-affine slices (the default for big requests): ONE launch, workgroup b reduces column slice b % n.  With the
//    observed b % 8 workgroup -> XCD placement each XCD's 4 MB L2 then holds 1 / n of every hub row instead of whole
//    rows of 1 / 8 of the segments: n times as many hub rows stay L2-resident per XCD, at the price of reading the ids
//    n times and of 1 / n-row pieces.  Measured (profiles/r04/agg_probe_*_run4.txt, same process, ms): C3 hop-2
//    (D = 256, 16.4 M ids) n = 1 / 2 / 4: 1.96 / 1.85 / 1.82 on the power-law request, 3.48 / 3.52 / 3.72 on uniformly
//    random rows; C2 (D = 128) 0.633 / 0.561 / 0.596 and 0.853 / 0.881 / 0.929; C4's shape 1.03 / 0.96 / 1.07 and
//    1.86 / 1.90 / 1.98.  n = 2 is the default: -6 .. -11 % where rows are re-used, +1 .. 3 % where none is; small
//    requests (C3 hop 1: 0.098 -> 0.120 ms) stay whole.  GLX_AGG_XCD_SLICES = 1 (off) | 2 | 4 | 8 overrides.
//  * GLX_AGG_SLICES = n: the columns in n slices, one LAUNCH after the other (ablation; profiles/r02: no gain).
constexpr int32_t kXcdSliceMinIds = 4 << 20;

template <int OP, int NSRC>
void launch_agg_n(const AggArgs& a0, int32_t num_ids, hipStream_t s) {
  AggArgs a = a0;
  a.col0 = 0;
  a.ncols = a.dim;
  a.xcd_slices = 0;
  int xcd = agg_knobs().xcd.load(std::memory_order_relaxed);
  if (xcd == 0) xcd = (num_ids >= kXcdSliceMinIds && a.dim % 8 == 0 && a.dim >= 128) ? 2 : 1;
  int slices = agg_knobs().slices.load(std::memory_order_relaxed);
  if ((slices != 2 && slices != 4 && slices != 8) || a.dim % (4 * slices) != 0) slices = 1;
  if (xcd != 2 && xcd != 4 && xcd != 8) xcd = 1;
  for (int c = 0; c < slices; ++c) {
    a.ncols = a.dim / slices;
    a.col0 = c * a.ncols;
    launch_agg_cols<OP, NSRC>(a, num_ids, slices == 1 ? xcd : 1, s);
  }
}

template <int OP>
void launch_agg(const AggArgs& a, int32_t num_ids, hipStream_t s) {
  if (a.X1 || a.X2) launch_agg_n<OP, 3>(a, num_ids, s);
  else launch_agg_n<OP, 1>(a, num_ids, s);
}

// Upload: row r of the caller's dense [V, D] matrix -> its (swizzled, pitched) slot.
__global__ void glx_place_rows_kernel(const float* __restrict__ src, int64_t num_rows, int32_t dim,
                                      int64_t stride, int64_t swizzle_rows, float* __restrict__ dst) {
  const int64_t total = num_rows * (int64_t)dim;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += step) {
    const int64_t r = t / dim;
    const int32_t c = (int32_t)(t - r * dim);
    dst[glx_swizzle_row(r, swizzle_rows) * stride + c] = src[t];
  }
}

// Feature gather (LookupNodes float attributes): G lanes per output row.
__global__ __launch_bounds__(256) void glx_lookup_kernel(GlxIdMap map, const float* __restrict__ X,
                                                         int64_t stride, int64_t swizzle_rows, int32_t dim, const int64_t* __restrict__ ids,
                                                         int64_t n, float default_attr,
                                                         float* __restrict__ out, int G) {
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int c = threadIdx.x % G;
  if (gid >= n) return;
  int64_t row = glx_row_of(map, ids[gid]);
  if (row >= 0) row = glx_swizzle_row(row, swizzle_rows);
  float* o = out + gid * (int64_t)dim;
  if ((dim & 3) == 0) {
    for (int32_t col = c * 4; col < dim; col += G * 4) {
      float4 v = make_float4(default_attr, default_attr, default_attr, default_attr);
      if (row >= 0) v = *reinterpret_cast<const float4*>(X + row * stride + col);
      *reinterpret_cast<float4*>(o + col) = v;
    }
  } else {
    for (int32_t col = c; col < dim; col += G) o[col] = row >= 0 ? X[row * stride + col] : default_attr;
  }
}

int run_aggregate(AggArgs& a, int op, const int32_t* d_seg, int32_t num_ids, hipStream_t s) {
  GlxKernelTimer timer(GLX_KERNEL_AGGREGATE, s);
  if (agg_use_mfma(a, op)) {  // ablation knob (GLX_AGG_MFMA=1): the LDS-staged MFMA formulation of Sum / Mean
    if (op == GLX_AGG_SUM) launch_agg_mfma<GLX_AGG_SUM>(a, s);
    else launch_agg_mfma<GLX_AGG_MEAN>(a, s);
    timer.stop();
    return GLX_OK;
  }
  switch (op) {
    case GLX_AGG_SUM: launch_agg<GLX_AGG_SUM>(a, num_ids, s); break;
    case GLX_AGG_MEAN: launch_agg<GLX_AGG_MEAN>(a, num_ids, s); break;
    case GLX_AGG_MAX: launch_agg<GLX_AGG_MAX>(a, num_ids, s); break;
    case GLX_AGG_MIN: launch_agg<GLX_AGG_MIN>(a, num_ids, s); break;
    case GLX_AGG_PROD: launch_agg<GLX_AGG_PROD>(a, num_ids, s); break;
    default: break;
  }
  timer.stop();
  return GLX_OK;
}

// Segment bookkeeping shared by the single-table and the multi-source entry: fills
// a.seg_start (or a.fanout for uniform segments) from scratch laid out by the caller.
// scratch = seg_start (Sg + 1) + kSegScratchExtra int32 (a captured call's two words).
namespace {
// Two 64-bit words per (calling thread, device, stream) -- the same scope as the scratch workspace, so that host
// threads that share a stream (the default one, say) do not raise each other's words -- zeroed when made, freed with the
// thread.  The epoch that tags them is the BUFFER's own counter (round 6; a process-wide counter wrapped after 2^32
// explicit-segment_ids calls of all threads together, after which stale words carried larger tags than new calls and
// atomicMax never landed: ADVICE r05): every call on a buffer is ordered on its stream, so when the counter wraps the
// words are cleared by a memset on that stream and counting restarts at 1 -- no tag of an earlier lap can be seen again.
// A recycled stream handle finds words of an older epoch of the same buffer, which read as "nothing raised".
struct SegBuf {
  int dev;
  hipStream_t stream;
  unsigned long long* words;
  uint32_t epoch;  // the last epoch handed out (0 = none yet: 0 tags the zeroed words)
};
constexpr size_t kSegBufsMax = 64;  // streams a thread keeps words for; the oldest goes when a 65th shows up
struct SegWords {
  std::vector<SegBuf> bufs;
  ~SegWords() {
    for (auto& e : bufs) {  // best effort: at process exit the runtime may already be gone
      if (hipSetDevice(e.dev) == hipSuccess) (void)hipFree(e.words);
    }
    (void)hipGetLastError();
  }
};
thread_local SegWords g_seg_words;

// *out = the words of (this thread, the current device, s) and the next epoch on them.  Never called while s is being
// captured (prepare_segments keeps a captured call's words in the call's own scratch): it may allocate.
int seg_words_for(hipStream_t s, unsigned long long** out, uint32_t* epoch) {
  int dev = 0;
  GLX_HIP(hipGetDevice(&dev));
  SegBuf* buf = nullptr;
  for (auto& e : g_seg_words.bufs) {
    if (e.dev == dev && e.stream == s) {
      buf = &e;
      break;
    }
  }
  if (buf == nullptr) {
    if (g_seg_words.bufs.size() >= kSegBufsMax) {  // streams come and go: drop the oldest (hipFree waits for its readers)
      const SegBuf old = g_seg_words.bufs.front();
      g_seg_words.bufs.erase(g_seg_words.bufs.begin());
      GlxDeviceGuard guard(old.dev);
      (void)hipFree(old.words);
    }
    unsigned long long* p = nullptr;
    GLX_HIP(hipMalloc(&p, 256));
    GLX_HIP(hipMemsetAsync(p, 0, 256, s));  // ordered before the first scan on this stream
    g_seg_words.bufs.push_back(SegBuf{dev, s, p, 0u});
    buf = &g_seg_words.bufs.back();
  }
  if (++buf->epoch == 0) {  // wrapped: clear the words behind every earlier call of this stream, restart at 1
    GLX_HIP(hipMemsetAsync(buf->words, 0, 16, s));
    buf->epoch = 1;
  }
  *out = buf->words;
  *epoch = buf->epoch;
  return GLX_OK;
}
}  // namespace

// int32 words a caller adds to its scratch allocation behind seg_start[num_segments + 1] for prepare_segments
constexpr size_t kSegScratchExtra = 6;  // two 8-byte aligned 64-bit words

int prepare_segments(AggArgs& a, const int32_t* d_seg, int32_t num_ids, int32_t num_segments, int32_t* scratch,
                     hipStream_t s) {
  a.fanout = num_segments > 0 ? num_ids / num_segments : 0;
  memset(&a.seg_state, 0, sizeof(a.seg_state));
  if (d_seg == nullptr) {  // uniform segments: nothing to derive, nothing to read
    a.seg_start = nullptr;
    return GLX_OK;
  }
  int32_t* seg_start = scratch;
  unsigned long long* words = nullptr;
  SegState st;
  // A captured launch is replayed with the SAME epoch: the words a replay raised would still carry it in the next
  // replay -- and the per-stream buffer may not even exist yet (no allocation inside a capture), nor is the capture
  // stream the one a replay runs on.  A captured call therefore keeps its two words in its OWN scratch (the plan's
  // replayable arena), cleared by one more node of the graph.
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
    const uintptr_t at = (reinterpret_cast<uintptr_t>(scratch + (size_t)num_segments + 1) + 7) & ~(uintptr_t)7;
    words = reinterpret_cast<unsigned long long*>(at);
    st.epoch = 1;
    glx_seg_reset_kernel<<<1, 1, 0, s>>>(words);
  } else {
    int rc = seg_words_for(s, &words, &st.epoch);
    if (rc != GLX_OK) return rc;
  }
  // uniform segments are possible when the ids divide evenly; the scan then checks seg[i] == i / fanout
  const int32_t f = (num_segments > 0 && num_ids > 0 && num_ids % num_segments == 0) ? num_ids / num_segments : 0;
  st.words = words;
  st.floor_level = num_ids == 0 ? 2 : (f > 0 ? 0 : 1);
  if (num_ids > 0) {
    glx_seg_scan_kernel<<<(unsigned)(((int64_t)num_ids + 1023) / 1024), 256, 0, s>>>(d_seg, num_ids, num_segments, f, words,
                                                                                   st.epoch, seg_start);
  }
  glx_seg_fixup_kernel<<<(unsigned)((num_segments + 1 + 255) / 256), 256, 0, s>>>(d_seg, st, num_ids, num_segments, seg_start);
  a.seg_start = seg_start;
  a.seg_state = st;
  return GLX_OK;
}

int aggregate_device(const glx_features* f, int op, const int64_t* d_ids, const int32_t* d_seg,
                     int32_t num_ids, int32_t num_segments, float default_attr, float* d_emb,
                     int32_t* d_cnt, hipStream_t s) {
  // scratch: seg_start (Sg+1) [+ rows (N) for hashed ids]
  const bool hashed = f->idmap.any();  // raw ids are not rows: translate first (a table lookup, or arithmetic)
  const size_t n_i32 = (size_t)num_segments + 1 + kSegScratchExtra + (hashed ? (size_t)num_ids : 0);
  int32_t* scratch = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&scratch), n_i32 * sizeof(int32_t), s, 1);
  if (rc != GLX_OK) return rc;
  int32_t* rows = hashed ? scratch + num_segments + 1 + kSegScratchExtra : nullptr;
  AggArgs a;
  memset(&a, 0, sizeof(a));
  rc = prepare_segments(a, d_seg, num_ids, num_segments, scratch, s);
  if (rc != GLX_OK) return rc;
  if (hashed && num_ids > 0) {
    glx_rows_kernel<<<(unsigned)((num_ids + 255) / 256), 256, 0, s>>>(f->map(), d_ids, num_ids, rows);
  }
  a.X = f->X;
  a.node_ids = hashed ? nullptr : d_ids;
  a.rows = rows;
  a.emb_out = d_emb;
  a.cnt_out = d_cnt;
  a.num_rows = f->num_rows;
  a.stride = f->stride;
  a.swizzle_rows = f->swizzle_rows;
  a.dim = f->dim;
  a.num_segments = num_segments;
  a.default_attr = default_attr;
  run_aggregate(a, op, d_seg, num_ids, s);
  hipError_t le = hipGetLastError();
  glx_scratch_free(scratch, s);
  GLX_HIP(le);
  return GLX_OK;
}

// ---- AggregatingResponse::Stitch (aggregating_request.cc:172-213) on the device:
// fold P partial results [P][Sg][dim] in shard order, starting from InitFunc's
// value.  Pure streaming (reads P*Sg*dim*4 B once, coalesced; writes Sg*dim*4 B):
// one thread per VEC output columns of one segment.
template <int OP, int VEC>
__global__ __launch_bounds__(256) void glx_agg_stitch_kernel(const float* __restrict__ parts,
                                                             const int32_t* __restrict__ cnts, int32_t P,
                                                             int64_t num_segments, int32_t dim,
                                                             float default_attr, float* __restrict__ out,
                                                             int32_t* __restrict__ cnt_out) {
  const int32_t cols = dim / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= num_segments * cols) return;
  const int64_t sg = t / cols;
  const int32_t c0 = (int32_t)(t - sg * cols) * VEC;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = agg_init<OP>();
  int32_t total = 0;
  for (int32_t p = 0; p < P; ++p) {
    const int32_t c = cnts[(int64_t)p * num_segments + sg];
    if (c == 0) continue;  // this shard saw none of the segment's ids
    const float* src = parts + ((int64_t)p * num_segments + sg) * dim + c0;
    float r[VEC];
    if (VEC == 4) {
      const float4 q = *reinterpret_cast<const float4*>(src);
      r[0] = q.x; r[1 % VEC] = q.y; r[2 % VEC] = q.z; r[3 % VEC] = q.w;
    } else {
      r[0] = src[0];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      // Mean: left += right * segments[i] (mean_aggregator.cc:31-37)
      const float x = OP == GLX_AGG_MEAN ? r[v] * (float)c : r[v];
      acc[v] = agg_combine<OP>(acc[v], x);
    }
    total += c;
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    if (total == 0) acc[v] = default_attr;                       // aggregator.cc:74-86
    else if (OP == GLX_AGG_MEAN) acc[v] = acc[v] / (float)total;  // mean_aggregator.cc:45-61
  }
  float* dst = out + sg * dim + c0;
  if (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
  else dst[0] = acc[0];
  if (c0 == 0) cnt_out[sg] = total;
}

template <int OP>
void launch_agg_stitch(bool vec4, const float* parts, const int32_t* cnts, int32_t P, int64_t sg, int32_t dim,
                       float default_attr, float* out, int32_t* cnt_out, hipStream_t s) {
  const int64_t threads = sg * (vec4 ? dim / 4 : dim);
  const unsigned grid = (unsigned)((threads + 255) / 256);
  if (vec4) glx_agg_stitch_kernel<OP, 4><<<grid, 256, 0, s>>>(parts, cnts, P, sg, dim, default_attr, out, cnt_out);
  else glx_agg_stitch_kernel<OP, 1><<<grid, 256, 0, s>>>(parts, cnts, P, sg, dim, default_attr, out, cnt_out);
}

}  // namespace

// glx_dist.hip: segmented reduce over pre-resolved virtual rows of up to three sources
// (own shard, hot-row replica, halo rows); vrows[i] = -1 is the default row.
int glx_aggregate_vrows_device(const GlxRowSource* src, int nsrc, int32_t dim, int op, const int32_t* vrows,
                               const int32_t* d_seg, int32_t num_ids, int32_t num_segments, float default_attr,
                               float* d_emb, int32_t* d_cnt, hipStream_t s) {
  int32_t* scratch = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&scratch), ((size_t)num_segments + 1 + kSegScratchExtra) * sizeof(int32_t), s, 1);
  if (rc != GLX_OK) return rc;
  AggArgs a;
  memset(&a, 0, sizeof(a));
  rc = prepare_segments(a, d_seg, num_ids, num_segments, scratch, s);
  if (rc != GLX_OK) return rc;
  const GlxRowSource none{nullptr, dim, 0, 0};
  const GlxRowSource& s0 = nsrc > 0 ? src[0] : none;
  const GlxRowSource& s1 = nsrc > 1 ? src[1] : none;
  const GlxRowSource& s2 = nsrc > 2 ? src[2] : none;
  a.X = s0.X;
  a.stride = s0.stride;
  a.swizzle_rows = s0.swizzle_rows;
  a.num_rows = s0.rows;
  a.X1 = s1.X;
  a.stride1 = s1.stride;
  a.swizzle1 = s1.swizzle_rows;
  a.X2 = s2.X;
  a.stride2 = s2.stride;
  a.base1 = (int32_t)s0.rows;
  a.base2 = (int32_t)(s0.rows + s1.rows);
  // the multi-source kernel is selected by a non-null X1 / X2; with neither (no replica, no
  // halo) the single-table kernel reads the same rows
  a.node_ids = nullptr;
  a.rows = vrows;
  a.emb_out = d_emb;
  a.cnt_out = d_cnt;
  a.dim = dim;
  a.num_segments = num_segments;
  a.default_attr = default_attr;
  run_aggregate(a, op, d_seg, num_ids, s);
  hipError_t le = hipGetLastError();
  glx_scratch_free(scratch, s);
  GLX_HIP(le);
  return GLX_OK;
}

extern "C" int glx_tune(const char* name, int32_t value) {
  GLX_REQUIRE(name != nullptr, "name is NULL");
  AggKnobs& k = agg_knobs();
  std::atomic<int>* slot = nullptr;
  if (strcmp(name, "agg_mfma") == 0) slot = &k.mfma;
  else if (strcmp(name, "agg_unroll") == 0) slot = &k.unroll;
  else if (strcmp(name, "agg_slices") == 0) slot = &k.slices;
  else if (strcmp(name, "agg_legacy") == 0) slot = &k.legacy;
  else if (strcmp(name, "agg_segs") == 0) slot = &k.segs;
  else if (strcmp(name, "agg_xcd_slices") == 0) slot = &k.xcd;
  else if (strcmp(name, "agg_occupancy") == 0) slot = &k.occ;
  else if (strcmp(name, "agg_store") == 0) slot = &k.store;
  else if (strcmp(name, "seg_epochs_before_wrap") == 0) {
    // test aid: the calling thread's segment-word buffers hand out `value` more epochs before their counter wraps
    GLX_REQUIRE(value >= 0, "seg_epochs_before_wrap must be >= 0");
    for (auto& e : g_seg_words.bufs) e.epoch = 0xffffffffu - (uint32_t)value;
    return GLX_OK;
  }
  if (slot == nullptr) {  // the side paths' knobs (glx_common.h GlxSideKnobs): -1 restores the default
    GlxSideKnobs& sk = glx_side_knobs();
    std::atomic<int64_t>* side = nullptr;
    if (strcmp(name, "cond_sequential") == 0) side = &sk.cond_sequential;
    else if (strcmp(name, "dist_no_bitmap") == 0) side = &sk.dist_no_bitmap;
    else if (strcmp(name, "filter_span_cap") == 0) side = &sk.filter_span_cap;
    else if (strcmp(name, "filter_dedup_min_rows") == 0) side = &sk.filter_dedup_min_rows;
    else if (strcmp(name, "idmap_hash_only") == 0) side = &sk.idmap_hash_only;
    else if (strcmp(name, "resolve_ids") == 0) side = &sk.resolve_ids;
    else if (strcmp(name, "resolve_blocks") == 0) side = &sk.resolve_blocks;
    else if (strcmp(name, "resolve_set_share") == 0) side = &sk.resolve_set_share;
    else if (strcmp(name, "resolve_peek") == 0) side = &sk.resolve_peek;
    else if (strcmp(name, "resolve_own_first") == 0) side = &sk.resolve_own_first;
    GLX_REQUIRE(side != nullptr, "unknown knob '%s'", name);
    side->store(value, std::memory_order_relaxed);
    return GLX_OK;
  }
  slot->store(value, std::memory_order_relaxed);
  return GLX_OK;
}

extern "C" int glx_features_create(int device, int64_t num_rows, int32_t dim, const float* X,
                                   const int64_t* ids, int ptr_kind, void* stream,
                                   glx_features** out) {
  return glx_features_create_impl(device, num_rows, dim, X, ids, ptr_kind, stream, true, out);
}

int glx_features_create_impl(int device, int64_t num_rows, int32_t dim, const float* X, const int64_t* ids, int ptr_kind,
                             void* stream, bool allow_arithmetic_ids, glx_features** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(num_rows >= 0 && dim > 0, "bad shape [%lld, %d]", (long long)num_rows, dim);
  GLX_REQUIRE(num_rows < INT32_MAX, "num_rows must be < 2^31");
  GLX_REQUIRE(num_rows == 0 || X != nullptr, "X is NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  hipStream_t s = glx_stream(stream);
  glx_features* f = new (std::nothrow) glx_features();
  GLX_REQUIRE(f != nullptr, "out of host memory");
  memset(static_cast<void*>(f), 0, sizeof(*f));
  f->device = device;
  f->num_rows = num_rows;
  f->dim = dim;
  f->owns_x = true;
  // Row pitch.  When a row is a multiple of 256 bytes, rows whose ids share low
  // zero bits (RMAT-style / power-of-two-structured ids are exactly the hub ids)
  // alias onto the same cache sets and HBM channels; one extra 64-byte sector per
  // row breaks the power-of-two stride.  GLX_FEATURE_ROW_PAD overrides (floats).
  int64_t pad = 0;
  if (const char* env = getenv("GLX_FEATURE_ROW_PAD")) pad = atoll(env);  // experiment knob (floats)
  if (pad < 0 || (pad % 4) != 0) pad = 0;
  f->stride = dim + pad;
  f->swizzle_rows = (num_rows >> GLX_SWIZZLE_BITS) << GLX_SWIZZLE_BITS;
  if (const char* env = getenv("GLX_FEATURE_SWIZZLE")) {
    if (atoi(env) == 0) f->swizzle_rows = 0;
  }
  const size_t bytes = (size_t)(num_rows > 0 ? num_rows : 1) * f->stride * sizeof(float);
  hipError_t e = hipMalloc(&f->X, bytes);
  GlxTemp staged;
  if (e == hipSuccess && num_rows > 0) {
    const float* d_src = X;
    if (ptr_kind == GLX_PTR_HOST) {
      e = hipMalloc(&staged.p, (size_t)num_rows * dim * sizeof(float));
      if (e == hipSuccess) {
        e = hipMemcpyAsync(staged.p, X, (size_t)num_rows * dim * sizeof(float), hipMemcpyHostToDevice, s);
      }
      d_src = staged.as<float>();
    }
    if (e == hipSuccess) {
      const int64_t total = num_rows * (int64_t)dim;
      int64_t blocks = (total + 255) / 256;
      if (blocks > 65536) blocks = 65536;
      glx_place_rows_kernel<<<(unsigned)blocks, 256, 0, s>>>(d_src, num_rows, dim, f->stride,
                                                            f->swizzle_rows, f->X);
    }
  }
  int64_t* tmp_ids = nullptr;
  if (e == hipSuccess && ids) {
    const int64_t* d_ids = ids;
    if (ptr_kind == GLX_PTR_HOST) {
      e = hipMalloc(&tmp_ids, (size_t)(num_rows > 0 ? num_rows : 1) * sizeof(int64_t));
      if (e == hipSuccess) e = hipMemcpyAsync(tmp_ids, ids, (size_t)num_rows * sizeof(int64_t), hipMemcpyHostToDevice, s);
      d_ids = tmp_ids;
    }
    if (e == hipSuccess) {
      rc = allow_arithmetic_ids ? glx_idmap_build_auto(d_ids, num_rows, &f->idmap, s) : glx_idmap_build(d_ids, num_rows, &f->idmap, s);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (tmp_ids) (void)hipFree(tmp_ids);
  if (e != hipSuccess || rc != GLX_OK) {
    if (e != hipSuccess) glx_set_error("feature upload failed: %s", hipGetErrorString(e));
    glx_features_destroy(f);
    return e == hipErrorOutOfMemory ? GLX_RESOURCE_EXHAUSTED : (rc != GLX_OK ? rc : GLX_INTERNAL);
  }
  *out = f;
  return GLX_OK;
}

extern "C" int glx_features_view(int device, int64_t num_rows, int32_t dim, const float* X_device,
                                 glx_features** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(num_rows >= 0 && dim > 0, "bad shape [%lld, %d]", (long long)num_rows, dim);
  GLX_REQUIRE(num_rows < INT32_MAX, "num_rows must be < 2^31");
  GLX_REQUIRE(num_rows == 0 || X_device != nullptr, "X is NULL");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  glx_features* f = new (std::nothrow) glx_features();
  GLX_REQUIRE(f != nullptr, "out of host memory");
  memset(static_cast<void*>(f), 0, sizeof(*f));
  f->device = device;
  f->num_rows = num_rows;
  f->dim = dim;
  f->stride = dim;
  f->swizzle_rows = 0;  // a view addresses the caller's rows as they are
  f->X = const_cast<float*>(X_device);
  f->owns_x = false;
  *out = f;
  return GLX_OK;
}

extern "C" void glx_features_destroy(glx_features* f) {
  if (!f) return;
  GlxDeviceGuard guard(f->device);
  if (f->X && f->owns_x) (void)hipFree(f->X);
  glx_idmap_free(&f->idmap);
  delete f;
}

extern "C" int glx_features_info(const glx_features* f, int64_t* num_rows, int32_t* dim,
                                 int* has_id_map, int* device) {
  GLX_REQUIRE(f != nullptr, "features is NULL");
  if (num_rows) *num_rows = f->num_rows;
  if (dim) *dim = f->dim;
  if (has_id_map) *has_id_map = f->idmap.any();
  if (device) *device = f->device;
  return GLX_OK;
}

extern "C" int glx_aggregate(const glx_features* f, int op, const int64_t* node_ids,
                             const int32_t* segment_ids, int32_t num_ids, int32_t num_segments,
                             float default_attr, float* emb_out, int32_t* cnt_out, int ptr_kind,
                             void* stream) {
  GLX_REQUIRE(f != nullptr, "features is NULL");
  GLX_REQUIRE(op >= GLX_AGG_SUM && op <= GLX_AGG_PROD, "unknown aggregator id %d", op);
  GLX_REQUIRE(num_ids >= 0 && num_segments >= 0, "negative sizes");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE((int64_t)num_segments * f->dim <= INT32_MAX,
              "num_segments * dim exceeds int32 (tensor.h:47)");
  if (num_segments == 0) return GLX_OK;
  GLX_REQUIRE(emb_out && cnt_out && (num_ids == 0 || node_ids), "NULL data pointer");
  GLX_REQUIRE(segment_ids != nullptr || num_ids % num_segments == 0,
              "segment_ids == NULL means equal segments: num_ids must be a multiple of num_segments");
  GlxDeviceGuard guard(f->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", f->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, f->device) : glx_stream(stream);
  if (ptr_kind == GLX_PTR_DEVICE) {
    return aggregate_device(f, op, node_ids, segment_ids, num_ids, num_segments, default_attr,
                            emb_out, cnt_out, s);
  }
  GlxHostCallSlot admitted(f->device);
  const size_t emb_n = (size_t)num_segments * f->dim;
  // outputs go straight into the caller's buffers when those are pinned (glx_mapped_ptr): the embeddings are
  // by far the largest part of a response (4 * dim bytes per segment)
  float* m_emb = static_cast<float*>(glx_mapped_ptr(emb_out, emb_n * 4));
  int32_t* m_cnt = static_cast<int32_t*>(glx_mapped_ptr(cnt_out, (size_t)num_segments * 4));
  const bool direct = m_emb != nullptr && m_cnt != nullptr;
  const size_t out_b = direct ? 0 : ((emb_n * 4 + 15) & ~(size_t)15) + (size_t)num_segments * 4;
  const size_t bytes = (size_t)num_ids * 8 + (size_t)num_ids * 4 + out_b + 64;
  char* d = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), bytes, s, 0);
  if (rc != GLX_OK) return rc;
  int64_t* d_ids = reinterpret_cast<int64_t*>(d);
  int32_t* d_seg = reinterpret_cast<int32_t*>(d + (size_t)num_ids * 8);
  char* d_out = d + (((size_t)num_ids * 12 + 15) & ~(size_t)15);
  float* d_emb = direct ? m_emb : reinterpret_cast<float*>(d_out);
  int32_t* d_cnt = direct ? m_cnt : reinterpret_cast<int32_t*>(d_out + ((emb_n * 4 + 15) & ~(size_t)15));
  hipError_t e = hipSuccess;
  if (num_ids > 0) {
    e = hipMemcpyAsync(d_ids, node_ids, (size_t)num_ids * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && segment_ids) e = hipMemcpyAsync(d_seg, segment_ids, (size_t)num_ids * 4, hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess) {
    rc = aggregate_device(f, op, d_ids, segment_ids ? d_seg : nullptr, num_ids, num_segments, default_attr, d_emb, d_cnt, s);
    if (rc == GLX_OK && !direct) {
      e = hipMemcpyAsync(emb_out, d_emb, emb_n * 4, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipMemcpyAsync(cnt_out, d_cnt, (size_t)num_segments * 4, hipMemcpyDeviceToHost, s);
    }
  }
  hipError_t e2 = hipStreamSynchronize(s);
  glx_scratch_free(d, s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(e);
  GLX_HIP(e2);
  return GLX_OK;
}

extern "C" int glx_lookup(const glx_features* f, const int64_t* node_ids, int64_t n,
                          float default_attr, float* out, int ptr_kind, void* stream) {
  GLX_REQUIRE(f != nullptr, "features is NULL");
  GLX_REQUIRE(n >= 0, "negative n");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  if (n == 0) return GLX_OK;
  GLX_REQUIRE(node_ids && out, "NULL data pointer");
  GlxDeviceGuard guard(f->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", f->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, f->device) : glx_stream(stream);
  int G = 1;
  const int want = (f->dim & 3) == 0 ? f->dim / 4 : f->dim;
  while (G < 64 && G < want) G <<= 1;
  const int64_t threads = n * G;
  if (ptr_kind == GLX_PTR_DEVICE) {
    GlxKernelTimer timer(GLX_KERNEL_LOOKUP, s);
    glx_lookup_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(f->map(), f->X, f->stride, f->swizzle_rows, f->dim, node_ids,
                                                                       n, default_attr, out, G);
    timer.stop();
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  }
  GlxHostCallSlot admitted(f->device);
  const size_t out_bytes = (size_t)n * f->dim * 4;
  float* m_out = static_cast<float*>(glx_mapped_ptr(out, out_bytes));  // pinned caller buffer: the kernel writes it directly
  const size_t ids_b = ((size_t)n * 8 + 15) & ~(size_t)15;  // ids first, 16-byte aligned rows after them
  char* d = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), ids_b + (m_out ? 0 : out_bytes), s, 0);
  if (rc != GLX_OK) return rc;
  int64_t* d_ids = reinterpret_cast<int64_t*>(d);
  float* d_out = m_out ? m_out : reinterpret_cast<float*>(d + ids_b);
  hipError_t e = hipMemcpyAsync(d_ids, node_ids, (size_t)n * 8, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    glx_lookup_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(f->map(), f->X, f->stride, f->swizzle_rows, f->dim, d_ids, n,
                                                                       default_attr, d_out, G);
    if (!m_out) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s);
  }
  hipError_t e2 = hipStreamSynchronize(s);
  glx_scratch_free(d, s);
  GLX_HIP(e);
  GLX_HIP(e2);
  return GLX_OK;
}

extern "C" int glx_aggregate_stitch(int device, int op, int32_t num_parts, const float* parts,
                                    const int32_t* cnts, int32_t num_segments, int32_t dim,
                                    float default_attr, float* emb_out, int32_t* cnt_out, void* stream) {
  GLX_REQUIRE(op >= GLX_AGG_SUM && op <= GLX_AGG_PROD, "unknown aggregator %d", op);
  GLX_REQUIRE(num_parts >= 1 && num_segments >= 0 && dim >= 1, "bad sizes");
  if (num_segments == 0) return GLX_OK;
  GLX_REQUIRE(parts && cnts && emb_out && cnt_out, "NULL data pointer");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  hipStream_t s = glx_stream(stream);
  const bool vec4 = dim % 4 == 0 && ((uintptr_t)parts % 16 == 0) && ((uintptr_t)emb_out % 16 == 0);
  switch (op) {
    case GLX_AGG_SUM: launch_agg_stitch<GLX_AGG_SUM>(vec4, parts, cnts, num_parts, num_segments, dim, default_attr, emb_out, cnt_out, s); break;
    case GLX_AGG_MEAN: launch_agg_stitch<GLX_AGG_MEAN>(vec4, parts, cnts, num_parts, num_segments, dim, default_attr, emb_out, cnt_out, s); break;
    case GLX_AGG_MAX: launch_agg_stitch<GLX_AGG_MAX>(vec4, parts, cnts, num_parts, num_segments, dim, default_attr, emb_out, cnt_out, s); break;
    case GLX_AGG_MIN: launch_agg_stitch<GLX_AGG_MIN>(vec4, parts, cnts, num_parts, num_segments, dim, default_attr, emb_out, cnt_out, s); break;
    default: launch_agg_stitch<GLX_AGG_PROD>(vec4, parts, cnts, num_parts, num_segments, dim, default_attr, emb_out, cnt_out, s); break;
  }
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}
