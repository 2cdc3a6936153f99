// glx internal header: handle layouts, error plumbing, the contract RNG and the
// id->row translation shared by every kernel.  gfx950 (CDNA4) only; wave = 64.
#ifndef GLX_COMMON_H_
#define GLX_COMMON_H_
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "glx.h"

#define GLX_WAVE 64

// ---------------------------------------------------------------- errors ----
void glx_set_error(const char* fmt, ...);

#define GLX_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      glx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,   \
                    __LINE__);                                                          \
      return e__ == hipErrorOutOfMemory ? GLX_RESOURCE_EXHAUSTED : GLX_INTERNAL;        \
    }                                                                                   \
  } while (0)

#define GLX_REQUIRE(cond, ...)         \
  do {                                 \
    if (!(cond)) {                     \
      glx_set_error(__VA_ARGS__);      \
      return GLX_INVALID_ARGUMENT;     \
    }                                  \
  } while (0)

// RAII device guard: every entry point runs on the handle's device and restores
// the caller's current device (one process may drive several GPUs).
struct GlxDeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit GlxDeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) return;
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    if (prev == dev) prev = -1;
  }
  ~GlxDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Cached per-(thread, device, stream, slot) workspaces (glx_graph.hip).  slot 0 =
// host-pointer staging of an entry point, slot 1 = kernel-internal scratch.
int glx_init_device(int device);
int glx_scratch_alloc(void** p, size_t bytes, hipStream_t s, int slot);
void glx_scratch_free(void* p, hipStream_t s);
void glx_scratch_trim(hipStream_t s, int slot, size_t keep_bytes);

// Request plans (glx_plan.hip) capture a sequence of entry points into a hipGraph.  While a plan is
// being built on this thread: (1) workspaces must not be allocated inside the capture, and must belong to
// the plan, not to the per-thread cache -- glx_scratch_alloc first RECORDS the sizes of a dry run, then
// REPLAYS them out of the plan's arena; (2) the samplers read the run's call counter from device memory.
enum { GLX_SCRATCH_NORMAL = 0, GLX_SCRATCH_RECORD = 1, GLX_SCRATCH_REPLAY = 2 };
void glx_scratch_mode(int mode, char* arena, size_t arena_bytes);  // RECORD clears the recorded sizes
size_t glx_scratch_recorded_bytes();
void glx_capture_set_cc_dev(const uint64_t* p);
const uint64_t* glx_capture_cc_dev();
bool glx_profile_suspend(bool suspend);  // returns the previous "enabled" state when suspending

// Temporary device allocation released on every exit path.
struct GlxTemp {
  void* p = nullptr;
  ~GlxTemp() {
    if (p) (void)hipFree(p);
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
  GlxTemp() = default;
  GlxTemp(const GlxTemp&) = delete;
  GlxTemp& operator=(const GlxTemp&) = delete;
};

// Storage slot of feature row r.  Row r of an owned feature table lives at slot
// r' = r with its low 12 bits XOR-ed by a hash of the high bits (a bijection inside
// every aligned block of 4096 rows).  Power-of-two-structured ids -- RMAT hubs are
// exactly the ids with many low zero bits -- would otherwise put the hottest rows
// at addresses that share their low bits, i.e. on the same cache sets and HBM
// channels; the swizzle decorrelates address bits from id bits at zero memory cost.
#define GLX_SWIZZLE_BITS 12
__host__ __device__ __forceinline__ int64_t glx_swizzle_row(int64_t r, int64_t swizzle_rows) {
  if (r >= swizzle_rows) return r;
  const uint32_t hi = (uint32_t)(r >> GLX_SWIZZLE_BITS);
  const uint32_t m = (hi * 0x9E3779B1u) >> (32 - GLX_SWIZZLE_BITS);
  return r ^ (int64_t)m;
}

// Kernel timing hook (glx_profile_enable): record an event on `s` when enabled.
struct GlxKernelTimer {
  int slot = -1;
  hipStream_t s;
  GlxKernelTimer(int kind, hipStream_t stream);  // records the start event
  void stop();                                   // records the stop event
};

// ---------------------------------------------------------- id -> row map ---
// Open-addressing table over raw int64 ids (AutoIndex::Get, auto_indexing.cc:26-33).
// keys == nullptr means the dense identity map (raw id v is row v).
#define GLX_EMPTY_KEY INT64_MIN
struct GlxIdMap {
  const int64_t* keys;  // [cap], GLX_EMPTY_KEY = free slot
  const int32_t* vals;  // [cap]
  uint64_t mask;        // cap - 1 (cap is a power of two)
  int64_t num_rows;
  // step > 0 (keys == nullptr): the ids are the arithmetic progression base, base + step, ... -- what the reference's
  // ownership rule llabs(id) % P leaves on shard r of a dense id space (r, r + P, r + 2P, ...).  The row is then
  // arithmetic: no table, no random access per lookup (round 5: the partitioned aggregation resolves 18 M ids per
  // step against the own shard's map; hashed, that was a quarter of the step's memory traffic at world size 1).
  int64_t base, step;
};

__host__ __device__ __forceinline__ uint64_t glx_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ int64_t glx_row_of(const GlxIdMap& m, int64_t id) {
  if (m.keys == nullptr) {
    if (m.step > 0) {
      if (id < m.base) return -1;
      const uint64_t d = (uint64_t)id - (uint64_t)m.base;  // exact: id >= base, so the true difference fits 64 unsigned bits
      // (a 64-bit division is ~100 instructions on this ISA; distances and steps nearly always fit 32 bits)
      const uint64_t q = ((d | (uint64_t)m.step) <= 0xffffffffull) ? (uint64_t)((uint32_t)d / (uint32_t)m.step)
                                                                  : d / (uint64_t)m.step;
      return (q * (uint64_t)m.step == d && q < (uint64_t)m.num_rows) ? (int64_t)q : -1;
    }
    return (id >= 0 && id < m.num_rows) ? id : -1;
  }
  if (id == GLX_EMPTY_KEY) return -1;
  uint64_t h = glx_mix64((uint64_t)id) & m.mask;
  while (true) {
    int64_t k = m.keys[h];
    if (k == id) return m.vals[h];
    if (k == GLX_EMPTY_KEY) return -1;
    h = (h + 1) & m.mask;
  }
}

struct GlxIdMapStorage {
  int64_t* keys = nullptr;
  int32_t* vals = nullptr;
  uint64_t cap = 0;
  int64_t base = 0, step = 0;  // step > 0: arithmetic ids, no table (GlxIdMap)
  bool any() const { return keys != nullptr || step > 0; }  // false: raw id v IS row v
  GlxIdMap view(int64_t num_rows) const { return GlxIdMap{keys, vals, cap - 1, num_rows, base, step}; }
};
// Builds the table for ids[num_rows] (device pointer) on `s`.
int glx_idmap_build(const int64_t* d_ids, int64_t num_rows, GlxIdMapStorage* out, hipStream_t s);
// The same, but ids that form an arithmetic progression with a positive step get no table (synchronises `s`).
int glx_idmap_build_auto(const int64_t* d_ids, int64_t num_rows, GlxIdMapStorage* out, hipStream_t s);
// glx_sample_ex on device pointers where request row i draws from stream d_rows[i] AND answers into output row d_rows[i].
int glx_sample_scatter_device(const glx_graph* g, int sampler, const int64_t* d_src, const int64_t* d_rows,
                              int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                              uint64_t call_counter, int64_t* d_nbr, int64_t* d_eid, hipStream_t s);
// Membership of non-negative ids as one bit each (bit id % 64 of bits[id / 64], ids in [0, max]); bits == nullptr: none.
struct GlxMember {
  const uint64_t* bits;
  int64_t max;
};
// glx_partition with one more bucket (the last) for the ids `divert` knows; counts has num_shards + 1 entries.
// `member` (optional): the same set as a bitmap -- a few MB that stay in L2 where the hash probe is a random DRAM access.
int glx_partition_divert(int device, const int64_t* ids, int64_t n, int32_t num_shards, GlxIdMap divert, GlxMember member,
                         int64_t* bucketed, int64_t* order, int64_t* counts, hipStream_t s);
// Values the kernel that writes the bucket sizes also writes beside them (counts[at + j] = v[j], j < n; bucket sizes
// from index `at` on are then NOT written -- the distributed store ships the P owners' sizes + the parameters and
// keeps the diverted bucket's size to itself): a partitioned request's scalar parameters travel with its counts, and
// a launch of their own to put them there costs as much as partitioning a small request (glx_dist.hip).
struct GlxPartitionTail {
  int64_t v[12];
  int32_t n;
  int32_t at;
};
// glx_partition (divert.keys == nullptr && divert.step == 0: num_buckets = num_shards) or glx_partition_divert
// (num_buckets = num_shards + 1) with a tail; device already selected.  An empty request is ONE small launch.
int glx_partition_tail(int device, const int64_t* ids, int64_t n, int32_t num_shards, GlxIdMap divert, GlxMember member,
                       int64_t* bucketed, int64_t* order, int64_t* counts, const GlxPartitionTail& tail, hipStream_t s);
void glx_idmap_free(GlxIdMapStorage* m);
struct glx_features;
// glx_features_create; allow_arithmetic_ids = false keeps a hash table whatever the ids look like (the hot-row replica
// of a distributed store packs that table into its own slots).
int glx_features_create_impl(int device, int64_t num_rows, int32_t dim, const float* X, const int64_t* ids, int ptr_kind,
                             void* stream, bool allow_arithmetic_ids, glx_features** out);

// ---------------------------------------------------------------- handles ---
struct GlxAdj {  // one CSR slot: a single 16-byte gather per draw
  int64_t nbr;
  int64_t eid;
};
struct GlxAlias {  // AliasMethod probs_/alias_ of the slot's row (row-local index)
  float prob;
  int32_t alias;
};

// EdgeWeightSampler fast path: everything one alias draw needs in ONE 32-byte record
// (half a 64-byte sector): the slot's probability, and the (nbr, eid) of both possible
// outcomes -- the slot itself and its alias.  Halves the random sectors per draw of
// the two-gather formulation (alias[idx] then adj[final]).  Built when every edge id
// fits an int32 (edge ids are insertion indices, so E < 2^31 suffices).
struct GlxEwRec {
  float prob;
  int32_t eid_self;
  int32_t eid_alias;
  int32_t pad_;
  int64_t nbr_self;
  int64_t nbr_alias;
};

// Test / A-B knobs of the paths beside the hot one, read from the environment ONCE (first use: operators run on up to
// 32 pool threads and getenv is not something to call per request) and settable at run time through glx_tune().
// -1 = unset (the product's default).
struct GlxSideKnobs {
  std::atomic<int64_t> cond_sequential{-1};       // GLX_COND_SEQUENTIAL (set = 1): the one-wave walk for every conditional-negative request
  std::atomic<int64_t> dist_no_bitmap{-1};        // GLX_DIST_NO_BITMAP (set = 1): the hot-row replica's membership test as a hash map
  std::atomic<int64_t> filter_span_cap{-1};       // GLX_FILTER_SPAN_CAP: total degree per chunk of a filtered request
  std::atomic<int64_t> filter_dedup_min_rows{-1}; // GLX_FILTER_DEDUP_MIN_ROWS: rows from which (vertex, value) pairs share a table; 0 disables
  std::atomic<int64_t> resolve_ids{-1};           // GLX_RESOLVE_IDS=4|8: ids per thread per pass of the partitioned aggregation's resolve kernel (default 2)
  std::atomic<int64_t> resolve_blocks{-1};        // GLX_RESOLVE_BLOCKS=n: workgroups of that kernel (default 1024)
  std::atomic<int64_t> resolve_set_share{-1};     // GLX_RESOLVE_SET_SHARE=n: the halo id set holds at least n / 1024 of a request's ids (A/B; default 16 once a share is known)
  std::atomic<int64_t> resolve_own_first{-1};     // GLX_RESOLVE_OWN_FIRST=1|0: ids this rank owns skip / take the replica lookup (default: skip at world size 1 only)
  std::atomic<int64_t> resolve_peek{-1};          // GLX_RESOLVE_PEEK=0: no plain load of a set slot before the compare-and-swap (A/B)
  std::atomic<int64_t> idmap_hash_only{-1};       // GLX_IDMAP_HASH_ONLY (set = 1): feature tables keep a hash table even for arithmetic ids (A/B)
};
GlxSideKnobs& glx_side_knobs();  // glx_graph.hip

struct glx_graph {
  int device;
  int64_t num_rows, num_edges;
  int64_t* row_ptr;  // [V+1]
  GlxAdj* adj;       // [E] 16-byte aligned records
  float* weight;     // [E] or nullptr
  GlxAlias* alias;   // [E] or nullptr
  GlxAlias* alias_indeg;  // [E] alias tables over the neighbours' in-degrees, or nullptr
  int64_t* nbr_sorted;    // [E] every row's neighbour ids ascending (strict negative sampling, id filters), or nullptr
  uint32_t* slot_sorted;  // [E] the CSR slot each entry of nbr_sorted came from (built together with it)
  bool indeg_global;      // alias_indeg / dst_count hold in-degrees summed over all shards (glx_dist_enable_in_degree)
  GlxIdMapStorage dst_map;  // destination id -> index into dst_count (with alias_indeg), for in-degree lookups
  int64_t* dst_count;     // [num_dst] in-degree of every distinct destination id
  int64_t num_dst;
  GlxEwRec* ew;           // [E] packed EdgeWeight records, or nullptr (edge ids beyond int32)
  int64_t* ts;            // [E] GetEdgeTimestamp of every slot (timestamp filters), or nullptr
  GlxIdMapStorage idmap;
  GlxIdMap map() const { return idmap.view(num_rows); }
};

struct glx_features {
  int device;
  int64_t num_rows;
  int32_t dim;
  int64_t stride;  // floats between consecutive rows (>= dim; see glx_features_create)
  int64_t swizzle_rows;  // rows [0, swizzle_rows) are stored at glx_swizzle_row(r); 0 = off
  float* X;  // [V, stride] row-major, base 256-byte aligned
  bool owns_x;
  GlxIdMapStorage idmap;
  GlxIdMap map() const { return idmap.view(num_rows); }
};

// AliasMethod::Build (alias_method.cc:57-107) for ONE distribution of `count` weights,
// bit-identical to the serial reference: LIFO low/high stacks (`low` grows up from its
// base, `high` grows down from its base; |low| + |high| <= count always, so both may
// share one array of `count` ints from opposite ends), sum accumulated in double then
// narrowed to float (:73), float arithmetic without contraction (-ffp-contract=off).
// Runs per row on the device (one lane per row) and on the host for the one global
// table of a negative sampler.
__host__ __device__ inline void glx_alias_build_row(const float* dist, int32_t count, GlxAlias* tab,
                                                    int32_t* low, int32_t* high) {
  const float avg_prob = (float)(1.0 / (double)count);
  double acc = 0.0;
  for (int32_t i = 0; i < count; ++i) acc += (double)dist[i];
  const float sum = (float)acc;
  int32_t low_num = 0, high_num = 0;
  for (int32_t i = 0; i < count; ++i) {
    float prob = dist[i] / sum;
    tab[i] = GlxAlias{prob * (float)count, i};
    if (prob < avg_prob) {
      low[low_num++] = i;
    } else if (prob > avg_prob) {
      high[-(high_num++)] = i;
    }
  }
  while (low_num > 0 && high_num > 0) {
    int32_t low_idx = low[--low_num];
    int32_t high_idx = high[-(--high_num)];
    float p = tab[high_idx].prob - 1.0f + tab[low_idx].prob;
    tab[high_idx].prob = p;
    tab[low_idx].alias = high_idx;
    if (p < 1.0f) {
      low[low_num++] = high_idx;
    } else if (p > 1.0f) {
      high[-(high_num++)] = high_idx;
    }
  }
  while (low_num > 0) tab[low[--low_num]].prob = 1.0f;
  while (high_num > 0) tab[high[-(--high_num)]].prob = 1.0f;
}

// The same build for the device's one-lane-per-row kernels, restructured around memory
// latency; every float operation and every stack decision is the one above, so the tables are
// bit-identical.  The stacks hold {prob, index} pairs (one 8-byte load per pop instead of an
// index load and a dependent table load), and the tops of both stacks live in registers: the
// high entry that was just reduced is the next one popped (LIFO), and a high entry that drops
// below 1 is pushed onto -- hence immediately popped from -- the low stack.  `stk` holds
// `count` pairs: low grows up from stk, high grows down from stk + count - 1.
__device__ inline void glx_alias_build_row_dev(const float* __restrict__ dist, int32_t count,
                                               GlxAlias* __restrict__ tab, GlxAlias* __restrict__ stk) {
  const float avg_prob = (float)(1.0 / (double)count);
  double acc = 0.0;
  for (int32_t i = 0; i < count; ++i) acc += (double)dist[i];
  const float sum = (float)acc;
  GlxAlias* low = stk;
  GlxAlias* high = stk + count - 1;
  int32_t low_num = 0, high_num = 0;
  for (int32_t i = 0; i < count; ++i) {
    const float prob = dist[i] / sum;
    const GlxAlias e = GlxAlias{prob * (float)count, i};
    tab[i] = e;
    if (prob < avg_prob) {
      low[low_num++] = e;
    } else if (prob > avg_prob) {
      high[-(high_num++)] = e;
    }
  }
  bool have_lo = false, have_hi = false;
  GlxAlias lo = GlxAlias{0.0f, 0}, hi = GlxAlias{0.0f, 0};  // .alias carries the entry's own index here
  while (low_num + (have_lo ? 1 : 0) > 0 && high_num + (have_hi ? 1 : 0) > 0) {
    if (!have_lo) lo = low[--low_num];
    if (!have_hi) hi = high[-(--high_num)];
    have_lo = false;
    const float p = hi.prob - 1.0f + lo.prob;
    tab[lo.alias].alias = hi.alias;
    hi.prob = p;
    if (p < 1.0f) {
      tab[hi.alias].prob = p;
      lo = hi;
      have_lo = true;
      have_hi = false;
    } else if (p > 1.0f) {
      have_hi = true;  // back on top of the high stack; its final prob is written when it leaves
    } else {
      tab[hi.alias].prob = p;
      have_hi = false;
    }
  }
  if (have_lo) tab[lo.alias].prob = 1.0f;
  while (low_num > 0) tab[low[--low_num].alias].prob = 1.0f;
  if (have_hi) tab[hi.alias].prob = 1.0f;
  while (high_num > 0) tab[high[-(--high_num)].alias].prob = 1.0f;
}

// scripts/probes/alias_row_probe.hip: cycle stamps of the phases of one row
#ifdef GLX_ALIAS_PROFILE
__device__ unsigned long long glx_alias_prof[8];
#define GLX_ALIAS_STAMP(i) do { if ((threadIdx.x & 63) == 0) glx_alias_prof[i] = __builtin_readcyclecounter(); } while (0)
#else
#define GLX_ALIAS_STAMP(i) do { } while (0)
#endif
// The pairing loop of the wave build (see glx_alias_build_row_wave).  stk holds the low stack in [0, low_num)
// (top = highest address) and the high stack in [count - high_num, count) (top = lowest address).
// kNan: some probability is not finite -- compare the way IEEE does (every comparison with a NaN is false).
template <bool kNan>
__device__ inline void glx_alias_pair_wave(GlxAlias* __restrict__ tab, const GlxAlias* __restrict__ stk, int32_t count_v,
                                           int32_t low_num_v, int32_t high_num_v) {
  const int lane = threadIdx.x & 63;
  const int32_t count = __builtin_amdgcn_readfirstlane(count_v);
  const uint64_t tab_bits = reinterpret_cast<uint64_t>(tab);  // the same in every lane: make that known
  GlxAlias* const tab_u = reinterpret_cast<GlxAlias*>(
      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(tab_bits >> 32)) << 32) |
      (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)tab_bits));
  int32_t lows = __builtin_amdgcn_readfirstlane(low_num_v);    // entries of the low stack not yet in the current window
  int32_t highs = __builtin_amdgcn_readfirstlane(high_num_v);
  // windows: lane j holds the j-th next pop; `n*` = the prefetched following window
  int32_t wl_p = 0, wl_i = 0, wh_p = 0, wh_i = 0, nl_p = 0, nl_i = 0, nh_p = 0, nh_i = 0;
  int32_t l_n = 0, la = 0, h_n = 0, ha = 0;
  auto fetch_low = [&](int32_t& pp, int32_t& ii) {  // the next 64 pops of what is left in memory
    const int32_t j = lows - 1 - lane;
    GlxAlias e = GlxAlias{0.0f, 0};
    if (j >= 0) e = stk[j];
    pp = __float_as_int(e.prob);
    ii = e.alias;
  };
  auto fetch_high = [&](int32_t& pp, int32_t& ii) {
    const int32_t j = highs - 1 - lane;
    GlxAlias e = GlxAlias{0.0f, 0};
    if (j >= 0) e = stk[count - 1 - j];
    pp = __float_as_int(e.prob);
    ii = e.alias;
  };
  fetch_low(nl_p, nl_i);
  fetch_high(nh_p, nh_i);
  int32_t c_p = 0, c_i = 0;  // the entry the last step left on top of a stack
  int state = 0;             // 0: nothing carried, 1: a high is carried, 2: a low is carried
  while (true) {
    // ---- outer: a window ran out
    if (state != 2 && la == l_n) {
      if (lows == 0) break;
      wl_p = nl_p;
      wl_i = nl_i;
      l_n = lows < 64 ? lows : 64;
      lows -= l_n;
      la = 0;
      fetch_low(nl_p, nl_i);
    }
    if (state != 1 && ha == h_n) {
      if (highs == 0) break;
      wh_p = nh_p;
      wh_i = nh_i;
      h_n = highs < 64 ? highs : 64;
      highs -= h_n;
      ha = 0;
      fetch_high(nh_p, nh_i);
    }
    // ---- inner: steps until a window runs out; touches no prefetch register, waits for no memory.
    // A step: p = high - 1 + low; low's alias = high; p > 1: `high` stays on top of the high stack (carried);
    // else it leaves with probability p and, below 1, is the next low (carried).
    int32_t xa = __builtin_amdgcn_readfirstlane(la), ya = __builtin_amdgcn_readfirstlane(ha);
    const int32_t xn = __builtin_amdgcn_readfirstlane(l_n), yn = __builtin_amdgcn_readfirstlane(h_n);
    int st = __builtin_amdgcn_readfirstlane(state);
    int32_t kp = __builtin_amdgcn_readfirstlane(c_p), ki = __builtin_amdgcn_readfirstlane(c_i);
    if constexpr (!kNan) {
      // Hand-scheduled: the compiler turns the three-state machine into mask-valued booleans and dispatch codes
      // (45-55 instructions per step).  Written out: the entry a step leaves on top of a stack is carried in two
      // VGPRs (probability, index; every lane holds the same value), a pop is two v_readlane, the decision is a
      // v_cmp + s_cbranch_vccnz -- 12 instructions while a high stays carried, ~16 otherwise.
      // gfx950 wait states respected by construction: an SGPR written by v_readlane is read by a VALU
      // instruction no sooner than the third instruction after it; lane selects are written by SALU.
      int32_t s_p, s_i, v_t, v_off;
      int32_t v_kp = kp, v_ki = ki;
      asm volatile(
          "s_mov_b64 exec, 1\n\t"  // lane 0 alone: a 64-lane store to one address is 64 conflicting writes
          "s_cmp_eq_u32 %[st], 1\n\t"
          "s_cbranch_scc1 .Lglx_high%=\n\t"
          "s_cmp_eq_u32 %[st], 2\n\t"
          "s_cbranch_scc1 .Lglx_low%=\n"
          ".Lglx_none%=:\n\t"  // nothing carried: pop a low and a high
          "s_mov_b32 %[st], 0\n\t"
          "s_cmp_ge_i32 %[xa], %[xn]\n\t"
          "s_cbranch_scc1 .Lglx_out%=\n\t"
          "s_cmp_ge_i32 %[ya], %[yn]\n\t"
          "s_cbranch_scc1 .Lglx_out%=\n\t"
          "v_readlane_b32 %[sp], %[whp], %[ya]\n\t"
          "v_readlane_b32 %[si], %[whi], %[ya]\n\t"
          "s_add_i32 %[ya], %[ya], 1\n\t"
          "s_nop 0\n\t"
          "v_mov_b32 %[vkp], %[sp]\n\t"
          "v_mov_b32 %[vki], %[si]\n"
          ".Lglx_high%=:\n\t"  // (vkp, vki) is the top of the high stack: pop a low
          "s_mov_b32 %[st], 1\n\t"
          "s_cmp_ge_i32 %[xa], %[xn]\n\t"
          "s_cbranch_scc1 .Lglx_out%=\n"
          ".Lglx_high_go%=:\n\t"
          "v_readlane_b32 %[sp], %[wlp], %[xa]\n\t"
          "v_readlane_b32 %[si], %[wli], %[xa]\n\t"
          "v_add_f32 %[vkp], -1.0, %[vkp]\n\t"
          "s_add_i32 %[xa], %[xa], 1\n\t"
          "v_lshlrev_b32 %[voff], 3, %[si]\n\t"
          "v_add_f32 %[vkp], %[sp], %[vkp]\n\t"
          "global_store_dword %[voff], %[vki], %[tab] offset:4\n\t"
          "v_cmp_lt_f32 vcc, 1.0, %[vkp]\n\t"
          "s_cbranch_vccz .Lglx_leave%=\n\t"
          "s_cmp_lt_i32 %[xa], %[xn]\n\t"  // still above 1: the same high takes the next low
          "s_cbranch_scc1 .Lglx_high_go%=\n\t"
          "s_branch .Lglx_out%=\n"
          ".Lglx_leave%=:\n\t"  // (vkp, vki) leaves the high stack with probability vkp <= 1
          "v_lshlrev_b32 %[voff], 3, %[vki]\n\t"
          "v_cmp_eq_f32 vcc, 1.0, %[vkp]\n\t"
          "global_store_dword %[voff], %[vkp], %[tab]\n\t"
          "s_cbranch_vccnz .Lglx_none%=\n"
          ".Lglx_low%=:\n\t"  // (vkp, vki) is the top of the low stack: pop a high
          "s_mov_b32 %[st], 2\n\t"
          "s_cmp_ge_i32 %[ya], %[yn]\n\t"
          "s_cbranch_scc1 .Lglx_out%=\n\t"
          "v_readlane_b32 %[sp], %[whp], %[ya]\n\t"
          "v_readlane_b32 %[si], %[whi], %[ya]\n\t"
          "v_lshlrev_b32 %[voff], 3, %[vki]\n\t"
          "s_add_i32 %[ya], %[ya], 1\n\t"
          "v_add_f32 %[vt], -1.0, %[sp]\n\t"
          "v_mov_b32 %[vki], %[si]\n\t"
          "v_add_f32 %[vkp], %[vkp], %[vt]\n\t"
          "global_store_dword %[voff], %[vki], %[tab] offset:4\n\t"
          "v_cmp_lt_f32 vcc, 1.0, %[vkp]\n\t"
          "s_cbranch_vccz .Lglx_leave%=\n\t"
          "s_branch .Lglx_high%=\n"
          ".Lglx_out%=:\n\t"
          "s_mov_b64 exec, -1\n\t"
          : [st] "+s"(st), [xa] "+s"(xa), [ya] "+s"(ya), [vkp] "+v"(v_kp), [vki] "+v"(v_ki), [sp] "=&s"(s_p),
            [si] "=&s"(s_i), [vt] "=&v"(v_t), [voff] "=&v"(v_off)
          : [xn] "s"(xn), [yn] "s"(yn), [wlp] "v"(wl_p), [wli] "v"(wl_i), [whp] "v"(wh_p), [whi] "v"(wh_i),
            [tab] "s"(tab_u)
          : "memory", "scc", "vcc");
      kp = __builtin_amdgcn_readfirstlane(v_kp);
      ki = __builtin_amdgcn_readfirstlane(v_ki);
    } else {
      char* const tab_b = reinterpret_cast<char*>(tab);
      while (true) {
        if (st == 1) {
          // a high is carried: it takes low after low while it stays above 1
          bool out = false;
          float p;
          do {
            if (xa >= xn) {
              out = true;
              break;
            }
            const int32_t lp = __builtin_amdgcn_readlane(wl_p, xa), li = __builtin_amdgcn_readlane(wl_i, xa);
            ++xa;
            p = __int_as_float(kp) - 1.0f + __int_as_float(lp);
            kp = __builtin_amdgcn_readfirstlane(__float_as_int(p));
            *reinterpret_cast<int32_t*>(tab_b + ((uint32_t)li << 3) + 4) = ki;
          } while (kNan ? ((kp & 0x7fffffff) <= 0x7f800000 && kp > 0x3f800000) : kp > 0x3f800000);
          if (out) break;
          *reinterpret_cast<float*>(tab_b + ((uint32_t)ki << 3)) = p;
          st = (kNan && (kp & 0x7fffffff) > 0x7f800000) ? 0 : (kp < 0x3f800000 ? 2 : 0);
          continue;
        }
        int32_t lp, li;
        if (st == 2) {
          if (ya >= yn) break;
          lp = kp;
          li = ki;
        } else {
          if (xa >= xn || ya >= yn) break;
          lp = __builtin_amdgcn_readlane(wl_p, xa);
          li = __builtin_amdgcn_readlane(wl_i, xa);
          ++xa;
        }
        const int32_t hp = __builtin_amdgcn_readlane(wh_p, ya);
        ki = __builtin_amdgcn_readlane(wh_i, ya);
        ++ya;
        const float p = __int_as_float(hp) - 1.0f + __int_as_float(lp);
        kp = __builtin_amdgcn_readfirstlane(__float_as_int(p));
        *reinterpret_cast<int32_t*>(tab_b + ((uint32_t)li << 3) + 4) = ki;
        if (!(kNan && (kp & 0x7fffffff) > 0x7f800000) && kp > 0x3f800000) {
          st = 1;
          continue;
        }
        *reinterpret_cast<float*>(tab_b + ((uint32_t)ki << 3)) = p;
        st = (kNan && (kp & 0x7fffffff) > 0x7f800000) ? 0 : (kp < 0x3f800000 ? 2 : 0);
      }
    }
    la = xa;
    ha = ya;
    state = st;
    c_p = kp;
    c_i = ki;
  }
  GLX_ALIAS_STAMP(3);
  // ---- whatever is left on either stack has probability 1
  if (state != 0 && lane == 0) tab[c_i].prob = 1.0f;
  if (lane >= la && lane < l_n) tab[wl_i].prob = 1.0f;
  if (lane >= ha && lane < h_n) tab[wh_i].prob = 1.0f;
  for (int32_t j = lane; j < lows; j += 64) tab[stk[j].alias].prob = 1.0f;
  for (int32_t j = lane; j < highs; j += 64) tab[stk[count - 1 - j].alias].prob = 1.0f;
}

// The same build once more, for a whole WAVE per distribution (long rows: the per-lane version walks
// a 100 K-entry hub row alone while 63 lanes of its wave wait).  Bit-identical again:
//   sum       every lane adds a strided share in double and the wave combines them.  That is only
//             allowed when no addition rounds: the addends are floats, i.e. integers times 2^elow; with
//             emax the largest exponent every partial sum is an integer multiple of 2^elow_min below
//             count * 2^(emax+1), so it is exact in double -- in ANY order -- as long as
//             emax + 1 + ceil(log2 count) - elow_min <= 52.  (Weights in [0.01, 1] span 30 bits, in-degrees
//             are integers: both qualify up to 2^20-entry rows.)  Otherwise lane 0 adds sequentially.
//   classify  prob / table entry per position in parallel; the low and high stacks are filled by ballot
//             compaction, which keeps the push order of the serial loop (ascending position).
//   pairing   inherently serial (each step's float result decides the next), so the cost is instructions per step
//             of ONE wave (a wave issues one instruction every few cycles whatever its width).  The machine of
//             glx_alias_build_row_dev runs on wave-uniform state: the next 64 pops of each stack sit one per
//             lane in a register (window), a pop is a v_readlane with a scalar lane index, the entry a step
//             leaves on top of a stack is carried in scalar registers, the step's float result goes to a scalar
//             register and the three-way decision (p < 1 / == 1 / > 1) is an integer compare and a scalar
//             branch between three straight-line states -- no exec masking, no LDS, no selects.  The following
//             window of each stack is prefetched (coalesced) while the current one is consumed.
// All 64 lanes of the wave call it with the same arguments.
constexpr int kAliasLaneRowMax = 96;  // rows up to this length are built by one lane (64 rows per wave)
__device__ inline void glx_alias_build_row_wave(const float* __restrict__ dist, int32_t count,
                                                GlxAlias* __restrict__ tab, GlxAlias* __restrict__ stk) {
  const int lane = threadIdx.x & 63;
  GLX_ALIAS_STAMP(0);
  const uint64_t lt = (1ull << lane) - 1ull;
  // ---- sum
  double acc = 0.0;
  int32_t emax = -1000, elow = 1000;
  bool odd = false;  // inf / nan / denormal: take the sequential sum
  for (int32_t i = lane; i < count; i += 64) {
    const float x = dist[i];
    acc += (double)x;
    const uint32_t b = __float_as_uint(x);
    const int32_t ef = (int32_t)((b >> 23) & 0xff);
    const uint32_t man = b & 0x7fffffu;
    if (ef == 0xff || (ef == 0 && man != 0)) odd = true;
    if (ef != 0 && ef != 0xff) {
      const int32_t e = ef - 127;
      const uint32_t m24 = man | 0x800000u;
      const int32_t lo = e - 23 + (__ffs((int)m24) - 1);
      emax = e > emax ? e : emax;
      elow = lo < elow ? lo : elow;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_xor(acc, off);
    const int32_t em = __shfl_xor(emax, off), el = __shfl_xor(elow, off);
    emax = em > emax ? em : emax;
    elow = el < elow ? el : elow;
  }
  int32_t log2n = 0;
  while ((1 << log2n) < count) ++log2n;
  const bool exact = !__any(odd) && (emax < -999 || emax + 1 + log2n - elow <= 52);
  if (!exact) {
    acc = 0.0;
    if (lane == 0) {
      for (int32_t i = 0; i < count; ++i) acc += (double)dist[i];
    }
    acc = __shfl(acc, 0);
  }
  const float sum = (float)acc;
  const float avg_prob = (float)(1.0 / (double)count);
  GLX_ALIAS_STAMP(1);
  // ---- classify: low stack grows up from stk[0], high stack down from stk[count - 1]
  int32_t low_num = 0, high_num = 0;
  bool bad = false;  // a probability that is not finite: the pairing loop must compare like IEEE (NaN is unordered)
  for (int32_t base = 0; base < count; base += 64) {
    const int32_t i = base + lane;
    bool is_low = false, is_high = false;
    GlxAlias e = GlxAlias{0.0f, 0};
    if (i < count) {
      const float prob = dist[i] / sum;
      e = GlxAlias{prob * (float)count, i};
      bad = bad || (__float_as_uint(e.prob) & 0x7f800000u) == 0x7f800000u;
      tab[i] = e;
      is_low = prob < avg_prob;
      is_high = prob > avg_prob;
    }
    const uint64_t bl = __ballot(is_low), bh = __ballot(is_high);
    if (is_low) stk[low_num + __popcll(bl & lt)] = e;
    if (is_high) stk[count - 1 - (high_num + __popcll(bh & lt))] = e;
    low_num += __popcll(bl);
    high_num += __popcll(bh);
  }
  __threadfence_block();
  GLX_ALIAS_STAMP(2);
  if (__any(bad)) {
    glx_alias_pair_wave<true>(tab, stk, count, low_num, high_num);
  } else {
    glx_alias_pair_wave<false>(tab, stk, count, low_num, high_num);
  }
  GLX_ALIAS_STAMP(4);
}

// ------------------------------------------------------------- contract RNG -
// Philox4x32-10; key = (seed lo, seed hi); counter = (j >> 1, row, cc lo, cc hi).
// Draw j of a row is words {2(j&1), 2(j&1)+1} of block j >> 1.
struct GlxPhilox {
  uint32_t w[4];
};

__device__ __forceinline__ GlxPhilox glx_philox_block(uint32_t blk, uint32_t row, uint64_t seed,
                                                      uint64_t cc) {
  uint32_t c0 = blk, c1 = row, c2 = (uint32_t)cc, c3 = (uint32_t)(cc >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return GlxPhilox{{c0, c1, c2, c3}};
}

__device__ __forceinline__ uint64_t glx_draw_of(const GlxPhilox& b, uint32_t j) {
  return (j & 1u) ? (((uint64_t)b.w[3] << 32) | b.w[2]) : (((uint64_t)b.w[1] << 32) | b.w[0]);
}

__device__ __forceinline__ uint64_t glx_draw64(uint64_t seed, uint64_t cc, uint32_t row,
                                               uint32_t j) {
  return glx_draw_of(glx_philox_block(j >> 1, row, seed, cc), j);
}

// [0, n): high 64 bits of u * n.
__device__ __forceinline__ uint64_t glx_bounded(uint64_t u, uint64_t n) { return __umul64hi(u, n); }

// alias_method.cc:117-121 under the contract: rand = float(U53 * (deg - 1)).
__device__ __forceinline__ int32_t glx_alias_pick(uint64_t u, int64_t deg,
                                                  const GlxAlias* __restrict__ row_alias) {
  double rd = ((double)(u >> 11) * 0x1.0p-53) * (double)(deg - 1);
  float rnd = (float)rd;
  int32_t ix = (int32_t)rnd;
  GlxAlias a = row_alias[ix];
  return (a.prob <= (rnd - (float)ix)) ? a.alias : ix;
}

// glx_graph.hip: shared tail of glx_graph_create / glx_graph_build.  Expects
// g->row_ptr, g->adj and (for weighted graphs) g->weight filled on `s`; builds the
// alias tables and, when d_ids != nullptr, the id map; synchronises `s`.
int glx_graph_finalize(glx_graph* g, const int64_t* d_ids, hipStream_t s);
// Launches AliasMethod::Build for every row of `row_ptr` over per-slot weights.
int glx_alias_build_launch(const int64_t* row_ptr, const float* weight, int64_t V, int64_t E,
                           GlxAlias* out, hipStream_t s);
void glx_graph_free(glx_graph* g);

// glx_full.hip: the in-degree machinery shared with the distributed store (glx_dist_enable_in_degree).
int glx_graph_dst_counts(const glx_graph* g, GlxTemp* uniq, GlxTemp* counts, int64_t* U, hipStream_t s);
int glx_graph_install_in_degree(glx_graph* g, const int64_t* d_uniq, const int64_t* d_counts, int64_t U,
                                hipStream_t s);

// glx_sample.hip: TopkSampler / RandomWithoutReplacementSampler restricted to the first
// d_prefix[i] slots of request row i (listed descending under circular padding).
int glx_sample_prefix_device(const glx_graph* g, int sampler, const int64_t* d_src, const int64_t* d_rng,
                             const int32_t* d_prefix, int32_t batch, int32_t k, int padding_mode,
                             int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter, int64_t* d_nbr,
                             int64_t* d_eid, hipStream_t s);

// glx_aggregate.hip: one row source of the multi-source segmented reduce (glx_dist.hip).
struct GlxRowSource {
  const float* X;
  int64_t stride;        // floats between rows
  int64_t swizzle_rows;  // glx_swizzle_row bound (0 = rows stored in order)
  int64_t rows;
};
// glx_negative.hip, for glx_dist.hip: InDegreeNegativeSampler on the rows an owner received
int glx_negative_sample_rows_device(const glx_negative* t, const glx_graph* g, const int64_t* src, const int64_t* rng_rows,
                                    int32_t batch, int32_t count, int64_t default_neighbor_id, uint64_t seed,
                                    uint64_t call_counter, int64_t* out, hipStream_t s);
int glx_aggregate_vrows_device(const GlxRowSource* src, int nsrc, int32_t dim, int op, const int32_t* vrows,
                               const int32_t* d_seg, int32_t num_ids, int32_t num_segments, float default_attr,
                               float* d_emb, int32_t* d_cnt, hipStream_t s);

// Admission control for host-pointer calls.  The reference's servers run up to 32 pool threads on one
// operator (in_memory_service.cc:64-71); on one GPU more than a dozen host-pointer calls in flight only contend
// (runtime locks, copy queues: 32 unlimited threads measured 2.2e8 edges/s, 3.7e8 with 12 admitted), so the
// surplus waits here.  GLX_HOST_CALL_CONCURRENCY overrides the limit (0 = unlimited).
struct GlxHostCallSlot {
  int device;
  explicit GlxHostCallSlot(int device);
  ~GlxHostCallSlot();
  GlxHostCallSlot(const GlxHostCallSlot&) = delete;
  GlxHostCallSlot& operator=(const GlxHostCallSlot&) = delete;
};

// Device-visible alias of a host buffer the caller pinned with glx_host_register, nullptr for anything else
// (pageable memory, and memory pinned by other means: glx trusts its own registry only).  Host-pointer calls let
// their kernels write results
// straight into such a buffer -- the response crosses PCIe once, as the kernel's own coalesced stores, with no
// staging copy and no copy engine in between; pageable buffers are served through a device workspace + copy.
// The WHOLE of [host_ptr, host_ptr + bytes) must lie inside one registered range (round 6: only the first byte used to be
// looked up -- a buffer that began inside a registered range and ran past its end would have been written directly,
// beyond the mapping; now it takes the staged path, where the runtime refuses the copy across the boundary by name).
void* glx_mapped_ptr(const void* host_ptr, size_t bytes);

static inline hipStream_t glx_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Host-pointer calls are synchronous.  When the caller passes no stream they run on
// a per-(host thread, device) non-blocking stream instead of the null stream, so the
// reference's pool threads (up to 32 concurrent Process() calls on one operator,
// in_memory_service.cc:64-71) overlap their copies and kernels instead of
// serialising on -- and synchronising with -- each other.
hipStream_t glx_thread_stream(int device);
static inline hipStream_t glx_host_call_stream(void* s, int device) {
  return s ? reinterpret_cast<hipStream_t>(s) : glx_thread_stream(device);
}

#endif  // GLX_COMMON_H_
