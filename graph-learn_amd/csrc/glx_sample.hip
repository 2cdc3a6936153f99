// glx neighbour samplers: RandomSampler, RandomWithoutReplacementSampler,
// EdgeWeightSampler (alias), TopkSampler + circular / replicate padding.
// Replaces graphlearn/src/core/operator/sampler/{random_sampler.cc:33-76,
// random_without_replacement_sampler.cc:31-75, edge_weight_sampler.cc:31-92,
// alias_method.cc:109-124, topk_sampler.cc:29-68, padder/*.h}.
//
// HBM-bound integer work: the kernels are organised around the memory system,
// not around arithmetic.  Output slots are the unit of parallelism (a request
// has batch*k of them, >> 256 CUs x 32 waves), every draw costs exactly one
// 16-byte {nbr, eid} gather (+ one 8-byte alias gather for EdgeWeight), and the
// [batch, k] outputs are written fully coalesced.
#include "glx_common.h"

namespace {

struct SampleArgs {
  GlxIdMap map;
  const int64_t* row_ptr;
  const GlxAdj* adj;
  const GlxAlias* alias;
  const GlxEwRec* ew;
  const int64_t* src;
  const int64_t* rng_rows;  // nullptr: request row i uses stream i
  const int64_t* out_rows;  // nullptr: request row i answers into output row i (else into output row out_rows[i])
  // Filtered requests whose reserved set is a row PREFIX (timestamp > value, filter.cc:74-82):
  // row i samples from its first prefix[i] slots only, listed in descending order
  // (reserved[x] = prefix[i] - 1 - x) when `reversed`.  nullptr: the whole row, as is.
  const int32_t* prefix;
  int32_t reversed;
  int64_t* nbr_out;
  int64_t* eid_out;
  int64_t default_nbr;
  uint64_t seed;
  uint64_t cc;
  // Inside a captured request plan (glx_plan.hip) the call counter of a run lives in device memory:
  // the effective counter is cc + *cc_dev.  nullptr everywhere else.
  const uint64_t* cc_dev;
  int32_t batch;
  int32_t k;
};

__device__ __forceinline__ uint64_t sample_cc(const SampleArgs& a) { return a.cc_dev ? a.cc + *a.cc_dev : a.cc; }

enum SlotOp { kSlotRandom = 0, kSlotEdgeWeight = 1, kSlotCircular = 2, kSlotReplicate = 3, kSlotEdgeWeightPacked = 4 };

// One thread = two consecutive output slots (2q, 2q+1) of one request row, i.e.
// exactly one Philox block for the randomised ops.  Consecutive lanes own
// consecutive slot pairs, so the nbr/eid stores of a wave cover one contiguous
// 1 KiB span each.
template <int OP>
__global__ __launch_bounds__(256) void glx_sample_slots_kernel(SampleArgs a, int32_t kpairs) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.batch * kpairs;
  if (t >= total) return;
  const int32_t i = (int32_t)(t / kpairs);
  const int32_t q = (int32_t)(t - (int64_t)i * kpairs);
  const int64_t row = glx_row_of(a.map, a.src[i]);
  int64_t start = 0, deg = 0;
  if (row >= 0) {
    start = a.row_ptr[row];
    deg = a.row_ptr[row + 1] - start;
  }
  if (a.prefix && row >= 0) deg = a.prefix[i];
  const int64_t obase = (a.out_rows ? a.out_rows[i] : (int64_t)i) * a.k;
  GlxPhilox blk;
  if (OP == kSlotRandom || OP == kSlotEdgeWeight || OP == kSlotEdgeWeightPacked) {
    if (deg > 0) {
      const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
      blk = glx_philox_block((uint32_t)q, rr, a.seed, sample_cc(a));
    }
  }
  GlxAdj got[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int32_t j = 2 * q + h;
    if (j >= a.k) break;
    int64_t pick = -1;
    if (deg > 0) {
      if (OP == kSlotRandom) {
        // random_sampler.cc:61-63: uniform index in [0, deg).
        pick = (int64_t)glx_bounded(glx_draw_of(blk, (uint32_t)j), (uint64_t)deg);
      } else if (OP == kSlotEdgeWeight) {
        // alias_method.cc:117-121 then CircularPadder with indices.size()==k.
        pick = glx_alias_pick(glx_draw_of(blk, (uint32_t)j), deg, a.alias + start);
      } else if (OP == kSlotCircular) {
        // circular_padder.h:46-63 over iota(deg) (TopkSampler).
        pick = j % deg;
        if (a.reversed) pick = deg - 1 - pick;
      } else {
        // replicate_padder.h:37-56: first min(k, deg) slots, then default fill.
        pick = j < deg ? j : -1;
      }
    }
    GlxAdj r = GlxAdj{a.default_nbr, -1};
    if (OP == kSlotEdgeWeightPacked) {
      if (deg > 0) {
        // alias_method.cc:117-121 on one packed record: a single 32-byte gather per draw
        const uint64_t u = glx_draw_of(blk, (uint32_t)j);
        const double rd = ((double)(u >> 11) * 0x1.0p-53) * (double)(deg - 1);
        const float rnd = (float)rd;
        const int32_t ix = (int32_t)rnd;
        const GlxEwRec rec = a.ew[start + ix];
        const bool take_alias = rec.prob <= (rnd - (float)ix);
        r = take_alias ? GlxAdj{rec.nbr_alias, (int64_t)rec.eid_alias} : GlxAdj{rec.nbr_self, (int64_t)rec.eid_self};
      }
    } else if (pick >= 0) {
      r = a.adj[start + pick];
    }
    got[h] = r;
  }
  // Both slots of the pair in ONE 16-byte store per output array whenever the pair is
  // complete and 16-byte aligned (always for even k): a wave then writes whole lines; two
  // 8-byte stores with a 16-byte lane stride showed up as 1.8x WRITE_SIZE in the PMC pass.
  int64_t* pn = a.nbr_out + obase + 2 * q;
  int64_t* pe = a.eid_out + obase + 2 * q;
  if (2 * q + 1 < a.k && ((reinterpret_cast<uintptr_t>(pn) | reinterpret_cast<uintptr_t>(pe)) & 15) == 0) {
    *reinterpret_cast<longlong2*>(pn) = longlong2{got[0].nbr, got[1].nbr};
    *reinterpret_cast<longlong2*>(pe) = longlong2{got[0].eid, got[1].eid};
  } else {
    pn[0] = got[0].nbr;
    pe[0] = got[0].eid;
    if (2 * q + 1 < a.k) {
      pn[1] = got[1].nbr;
      pe[1] = got[1].eid;
    }
  }
}

// RandomWithoutReplacement, circular padding, k <= W <= 64.
// A sub-group of W lanes owns one request row.  The contract permutation is a
// forward Fisher-Yates over the virtual array A[p] = p: step j swaps A[j] with
// A[r_j], r_j = j + bounded(draw_j, deg - j).  Only min(k, deg) steps matter,
// and only the <= k touched entries of A are ever materialised: lane t keeps
// r_t and w_t = A[t] at time t (the value step t moved into A[r_t]).  Looking
// A[p] up at step j is "the latest t < j with r_t == p" -- one ballot + one
// highest-set-bit per lookup -- so a row costs k cheap wave steps and touches
// HBM only for the k selected slots (no O(deg) scan of hub rows).
template <int W>
__global__ __launch_bounds__(256) void glx_rwor_kernel(SampleArgs a) {
  const int lane = threadIdx.x & 63;
  const int l = lane & (W - 1);
  const int base = lane - l;
  const uint64_t wmask = (W == 64) ? ~0ull : ((1ull << W) - 1ull);
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / W;
  const bool active = i < a.batch;
  int64_t start = 0, deg = 0;
  if (active) {
    const int64_t row = glx_row_of(a.map, a.src[i]);
    if (row >= 0) {
      start = a.row_ptr[row];
      deg = a.row_ptr[row + 1] - start;
    }
  }
  if (a.prefix && active) deg = deg > 0 ? a.prefix[i] : 0;
  const int32_t m = (int32_t)(deg < a.k ? deg : a.k);
  int32_t r = -1;
  if (l < m) {
    const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
    r = l + (int32_t)glx_bounded(glx_draw64(a.seed, sample_cc(a), rr, (uint32_t)l), (uint64_t)(deg - l));
  }
  int32_t w = 0, perm = 0;
  for (int32_t j = 0; j < a.k; ++j) {
    const bool in = j < m;
    if (!__any(in)) break;
    const int32_t rj = __shfl(r, base + j);
    const bool earlier = in && l < j;
    const uint64_t bw = (__ballot(earlier && r == j) >> base) & wmask;
    const uint64_t bp = (__ballot(earlier && r == rj) >> base) & wmask;
    const int tw = bw ? 63 - __clzll(bw) : -1;
    const int tp = bp ? 63 - __clzll(bp) : -1;
    const int32_t wsrc = __shfl(w, base + (tw < 0 ? 0 : tw));
    const int32_t psrc = __shfl(w, base + (tp < 0 ? 0 : tp));
    if (in && l == j) {
      w = tw < 0 ? j : wsrc;
      perm = tp < 0 ? rj : psrc;
    }
  }
  // Slot l of the row: circular_padder.h:46-63 with indices_ = the permutation.
  const bool has_slot = l < a.k;
  const int32_t c = (has_slot && m > 0) ? l % m : 0;
  const int32_t pc = __shfl(perm, base + c);
  if (active && has_slot) {
    GlxAdj rec = GlxAdj{a.default_nbr, -1};
    if (m > 0) rec = a.adj[start + (a.reversed ? (int32_t)deg - 1 - pc : pc)];
    const int64_t o = (a.out_rows ? a.out_rows[i] : (int64_t)i) * a.k + l;
    a.nbr_out[o] = rec.nbr;
    a.eid_out[o] = rec.eid;
  }
}

// k <= 16: the same permutation without the k dependent cross-lane steps.  Every lane of the
// W-lane group fetches all W draws (W independent shuffles), replays the sparse Fisher-Yates
// bookkeeping in registers -- w_t = "value found at position t when step t ran" -- with fully
// unrolled compares, and reads off its own entry: perm_l = value at position r_l before
// step l.  ~W^2 VALU compares instead of 3k serialised ds_bpermute round trips.
template <int W, int K>
__global__ __launch_bounds__(256) void glx_rwor_small_kernel(SampleArgs a) {
  // W lanes per request row, 64 / W rows per wavefront.  W need not be a power of two (round 4): k = 10 takes W = 10 and
  // six rows per wave instead of W = 16 and four with six idle lanes each -- this kernel is VALU-bound, so lanes are time.
  // Lanes past the last whole group of a wave idle; every cross-lane read below names an absolute lane of the own group.
  constexpr int kRowsPerWave = 64 / W;
  const int lane = threadIdx.x & 63;
  const int grp = lane / W;
  const int l = lane - grp * W;
  const int base = lane - l;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t i = wave * kRowsPerWave + grp;
  const bool active = grp < kRowsPerWave && i < a.batch;
  int64_t start = 0, deg = 0;
  if (active) {
    const int64_t row = glx_row_of(a.map, a.src[i]);
    if (row >= 0) {
      start = a.row_ptr[row];
      deg = a.row_ptr[row + 1] - start;
    }
  }
  if (a.prefix && active) deg = deg > 0 ? a.prefix[i] : 0;
  const int32_t m = (int32_t)(deg < a.k ? deg : a.k);
  int32_t r = l;
  if (l < m) {
    const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
    r = l + (int32_t)glx_bounded(glx_draw64(a.seed, sample_cc(a), rr, (uint32_t)l), (uint64_t)(deg - l));
  }
  // Only the first k steps exist (lanes and entries k .. W-1 are never read below): K = k is a template parameter and
  // the unrolled loops stop there instead of at W -- for k = 10 in the W = 16 shape that is 45 + 10 compare/select pairs and 10
  // cross-lane reads instead of 120 + 16 and 16, and this kernel is VALU-bound (C2 hop 2: ~500 lane-instructions per
  // slot group before, 0.234 ms; see profiles/r03/SUMMARY.md).
  int32_t rs[K], ws[K];
#pragma unroll
  for (int t = 0; t < K; ++t) rs[t] = __shfl(r, base + t);
#pragma unroll
  for (int t = 0; t < K; ++t) {
    int32_t v = t;
#pragma unroll
    for (int u = 0; u < t; ++u) v = (rs[u] == t) ? ws[u] : v;  // ascending u: the latest step wins
    ws[t] = v;
  }
  int32_t perm = r;
#pragma unroll
  for (int u = 0; u < K; ++u) perm = (u < l && rs[u] == r) ? ws[u] : perm;
  // Slot l of the row: circular_padder.h:46-63 with indices_ = the permutation.
  const bool has_slot = l < a.k;
  const int32_t c = (has_slot && m > 0) ? l % m : 0;
  const int32_t pc = __shfl(perm, base + c);
  if (active && has_slot) {
    GlxAdj rec = GlxAdj{a.default_nbr, -1};
    if (m > 0) rec = a.adj[start + (a.reversed ? (int32_t)deg - 1 - pc : pc)];
    const int64_t o = (a.out_rows ? a.out_rows[i] : (int64_t)i) * a.k + l;
    a.nbr_out[o] = rec.nbr;
    a.eid_out[o] = rec.eid;
  }
}

// Same algorithm for 64 < k <= 8192: one wave per row, r/w/perm in LDS, the
// "latest t < j" lookup is a strided scan + wave max.
__device__ __forceinline__ int32_t glx_wave_max_i32(int32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int32_t o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(64) void glx_rwor_lds_kernel(SampleArgs a) {
  extern __shared__ int32_t lds[];
  const int lane = threadIdx.x;
  const int32_t k = a.k;
  int32_t* r = lds;
  int32_t* w = lds + k;
  int32_t* perm = lds + 2 * k;
  const int64_t i = blockIdx.x;
  const int64_t row = glx_row_of(a.map, a.src[i]);
  int64_t start = 0, deg = 0;
  if (row >= 0) {
    start = a.row_ptr[row];
    deg = a.row_ptr[row + 1] - start;
  }
  if (a.prefix) deg = deg > 0 ? a.prefix[i] : 0;
  const int32_t m = (int32_t)(deg < k ? deg : k);
  const uint32_t rr = a.rng_rows ? (uint32_t)a.rng_rows[i] : (uint32_t)i;
  for (int32_t t = lane; t < m; t += 64) {
    r[t] = t + (int32_t)glx_bounded(glx_draw64(a.seed, sample_cc(a), rr, (uint32_t)t), (uint64_t)(deg - t));
  }
  __syncthreads();
  for (int32_t j = 0; j < m; ++j) {
    const int32_t rj = r[j];
    int32_t tw = -1, tp = -1;
    for (int32_t t = lane; t < j; t += 64) {
      const int32_t rt = r[t];
      if (rt == j) tw = t;
      if (rt == rj) tp = t;
    }
    tw = glx_wave_max_i32(tw);
    tp = glx_wave_max_i32(tp);
    if (lane == 0) {
      w[j] = tw < 0 ? j : w[tw];
      perm[j] = tp < 0 ? rj : w[tp];
    }
    __syncthreads();
  }
  for (int32_t j = lane; j < k; j += 64) {
    GlxAdj rec = GlxAdj{a.default_nbr, -1};
    if (m > 0) rec = a.adj[start + (a.reversed ? (int32_t)deg - 1 - perm[j % m] : perm[j % m])];
    const int64_t o = (a.out_rows ? a.out_rows[i] : i) * k + j;
    a.nbr_out[o] = rec.nbr;
    a.eid_out[o] = rec.eid;
  }
}

template <int OP>
void launch_slots(const SampleArgs& a, hipStream_t s) {
  const int32_t kpairs = (a.k + 1) / 2;
  const int64_t total = (int64_t)a.batch * kpairs;
  glx_sample_slots_kernel<OP><<<(unsigned)((total + 255) / 256), 256, 0, s>>>(a, kpairs);
}

template <int W>
void launch_rwor(const SampleArgs& a, hipStream_t s) {
  const int64_t threads = (int64_t)a.batch * W;
  glx_rwor_kernel<W><<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(a);
}

// k <= 16: one instantiation per k (the unrolled bookkeeping is k^2 / 2 compares: stopping at k instead of at the lane
// group's width is the difference between VALU-bound and not)
template <int K>
void launch_rwor_small(const SampleArgs& a, hipStream_t s) {
  // lanes per row: k itself when that packs more rows into a wavefront than the next power of two does (k = 3, 5, 6, 7,
  // 9 .. 12), else the power of two (cheaper lane arithmetic, same packing)
  constexpr int kPow2 = K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : 16;
  constexpr int W = (64 / K > 64 / kPow2) ? K : kPow2;
  constexpr int kRowsPerWave = 64 / W;
  const int64_t waves = ((int64_t)a.batch + kRowsPerWave - 1) / kRowsPerWave;
  glx_rwor_small_kernel<W, K><<<(unsigned)((waves + 3) / 4), 256, 0, s>>>(a);
}

void launch_rwor_small_k(const SampleArgs& a, hipStream_t s) {
  switch (a.k) {
    case 1: launch_rwor_small<1>(a, s); break;
    case 2: launch_rwor_small<2>(a, s); break;
    case 3: launch_rwor_small<3>(a, s); break;
    case 4: launch_rwor_small<4>(a, s); break;
    case 5: launch_rwor_small<5>(a, s); break;
    case 6: launch_rwor_small<6>(a, s); break;
    case 7: launch_rwor_small<7>(a, s); break;
    case 8: launch_rwor_small<8>(a, s); break;
    case 9: launch_rwor_small<9>(a, s); break;
    case 10: launch_rwor_small<10>(a, s); break;
    case 11: launch_rwor_small<11>(a, s); break;
    case 12: launch_rwor_small<12>(a, s); break;
    case 13: launch_rwor_small<13>(a, s); break;
    case 14: launch_rwor_small<14>(a, s); break;
    case 15: launch_rwor_small<15>(a, s); break;
    default: launch_rwor_small<16>(a, s); break;
  }
}

int sample_device(const glx_graph* g, int sampler, const SampleArgs& a, int padding_mode,
                  hipStream_t s) {
  const bool circular = padding_mode == GLX_PAD_CIRCULAR;
  GlxKernelTimer timer(GLX_KERNEL_SAMPLE, s);
  switch (sampler) {
    case GLX_SAMPLER_RANDOM:
      launch_slots<kSlotRandom>(a, s);
      break;
    case GLX_SAMPLER_EDGE_WEIGHT:
      if (g->alias == nullptr && g->num_edges == 0) {
        // a shard that holds no edge of a weighted type has no weight array to tell it is weighted: every row is
        // empty, every slot is the default neighbour -- whichever kernel writes it
        launch_slots<kSlotRandom>(a, s);
        break;
      }
      GLX_REQUIRE(g->alias != nullptr, "EdgeWeightSampler needs a weighted graph");
      // Replicate mode: ReplicatePadder ignores the drawn indices and (in the
      // reference) reads neighbors_[0..k) -- out of bounds when deg < k.  glx
      // returns the first min(k, deg) slots then default-fills.
      if (!circular) launch_slots<kSlotReplicate>(a, s);
      else if (a.ew) launch_slots<kSlotEdgeWeightPacked>(a, s);
      else launch_slots<kSlotEdgeWeight>(a, s);
      break;
    case GLX_SAMPLER_IN_DEGREE: {
      // in_degree_sampler.cc:79-92: the alias draw of EdgeWeightSampler over the
      // neighbours' in-degrees (tables built once by glx_graph_enable_in_degree).
      if (g->alias_indeg == nullptr && g->num_edges == 0) {
        launch_slots<kSlotRandom>(a, s);  // an empty shard: default ids only
        break;
      }
      GLX_REQUIRE(g->alias_indeg != nullptr, "InDegreeSampler needs glx_graph_enable_in_degree()");
      SampleArgs b = a;
      b.alias = g->alias_indeg;
      b.ew = nullptr;
      if (circular) launch_slots<kSlotEdgeWeight>(b, s);
      else launch_slots<kSlotReplicate>(b, s);
      break;
    }
    case GLX_SAMPLER_TOPK:
      if (circular) launch_slots<kSlotCircular>(a, s);
      else launch_slots<kSlotReplicate>(a, s);
      break;
    case GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT:
      if (!circular) {
        // ReplicatePadder discards the shuffle: first min(k, deg) in row order.
        launch_slots<kSlotReplicate>(a, s);
      } else if (a.k <= 16) {
        launch_rwor_small_k(a, s);
      } else if (a.k <= 32) {
        launch_rwor<32>(a, s);
      } else if (a.k <= 64) {
        launch_rwor<64>(a, s);
      } else {
        GLX_REQUIRE(a.k <= 8192, "RandomWithoutReplacementSampler supports neighbor_count <= 8192");
        glx_rwor_lds_kernel<<<(unsigned)a.batch, 64, (size_t)a.k * 3 * sizeof(int32_t), s>>>(a);
      }
      break;
    default:
      glx_set_error("unknown sampler id %d", sampler);
      return GLX_INVALID_ARGUMENT;
  }
  timer.stop();
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

}  // namespace

// glx_filter.hip: Topk / RandomWithoutReplacement over a per-row prefix (device pointers).
int glx_sample_prefix_device(const glx_graph* g, int sampler, const int64_t* d_src, const int64_t* d_rng,
                             const int32_t* d_prefix, int32_t batch, int32_t k, int padding_mode,
                             int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter, int64_t* d_nbr,
                             int64_t* d_eid, hipStream_t s) {
  SampleArgs a;
  a.map = g->map();
  a.row_ptr = g->row_ptr;
  a.adj = g->adj;
  a.alias = nullptr;
  a.ew = nullptr;
  a.src = d_src;
  a.rng_rows = d_rng;
  a.out_rows = nullptr;
  a.prefix = d_prefix;
  a.reversed = padding_mode == GLX_PAD_CIRCULAR ? 1 : 0;  // ReplicatePadder ignores the index values
  a.nbr_out = d_nbr;
  a.eid_out = d_eid;
  a.default_nbr = default_neighbor_id;
  a.seed = seed;
  a.cc = call_counter;
  a.cc_dev = glx_capture_cc_dev();
  a.batch = batch;
  a.k = k;
  return sample_device(g, sampler, a, padding_mode, s);
}

// glx_dist.hip: request rows served by a graph replica answer straight into their rows of the caller's response
// (output row = random-stream row = the row's index in the original request).  Device pointers, device selected.
int glx_sample_scatter_device(const glx_graph* g, int sampler, const int64_t* d_src, const int64_t* d_rows,
                              int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                              uint64_t call_counter, int64_t* d_nbr, int64_t* d_eid, hipStream_t s) {
  if (batch == 0 || k == 0) return GLX_OK;
  SampleArgs a;
  a.map = g->map();
  a.row_ptr = g->row_ptr;
  a.adj = g->adj;
  a.alias = g->alias;
  a.ew = g->ew;
  a.src = d_src;
  a.rng_rows = d_rows;
  a.out_rows = d_rows;
  a.prefix = nullptr;
  a.reversed = 0;
  a.nbr_out = d_nbr;
  a.eid_out = d_eid;
  a.default_nbr = default_neighbor_id;
  a.seed = seed;
  a.cc = call_counter;
  a.cc_dev = glx_capture_cc_dev();
  a.batch = batch;
  a.k = k;
  return sample_device(g, sampler, a, padding_mode, s);
}

extern "C" int glx_sample_ex(const glx_graph* g, int sampler, const int64_t* src,
                             const int64_t* rng_rows, int32_t batch, int32_t k, int padding_mode,
                             int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                             int64_t* nbr_out, int64_t* eid_out, int ptr_kind, void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(batch >= 0 && k >= 0, "negative batch / neighbor_count");
  GLX_REQUIRE(padding_mode == GLX_PAD_CIRCULAR || padding_mode == GLX_PAD_REPLICATE,
              "bad padding_mode %d", padding_mode);
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  // Tensor sizes are int32 in the reference (tensor.h:47): batch*k must fit.
  GLX_REQUIRE((int64_t)batch * k <= INT32_MAX, "batch * neighbor_count exceeds int32 (tensor.h:47)");
  if (batch == 0 || k == 0) return GLX_OK;
  GLX_REQUIRE(src && nbr_out && eid_out, "NULL data pointer");
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);

  SampleArgs a;
  a.map = g->map();
  a.row_ptr = g->row_ptr;
  a.adj = g->adj;
  a.alias = g->alias;
  a.ew = g->ew;
  a.default_nbr = default_neighbor_id;
  a.seed = seed;
  a.cc = call_counter;
  a.cc_dev = glx_capture_cc_dev();
  a.batch = batch;
  a.k = k;
  a.prefix = nullptr;
  a.reversed = 0;
  a.out_rows = nullptr;
  if (ptr_kind == GLX_PTR_DEVICE) {
    a.src = src;
    a.rng_rows = rng_rows;
    a.nbr_out = nbr_out;
    a.eid_out = eid_out;
    return sample_device(g, sampler, a, padding_mode, s);
  }
  // Host pointers: inputs are staged through a device workspace; outputs are written by the kernel straight
  // into the caller's buffers when those are pinned (glx_mapped_ptr), else staged and copied.  Synchronous.
  GlxHostCallSlot admitted(g->device);
  const size_t n_out = (size_t)batch * k;
  int64_t* m_nbr = static_cast<int64_t*>(glx_mapped_ptr(nbr_out, n_out * 8));
  int64_t* m_eid = static_cast<int64_t*>(glx_mapped_ptr(eid_out, n_out * 8));
  const bool direct = m_nbr != nullptr && m_eid != nullptr;
  int64_t* d = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), ((size_t)batch * 2 + (direct ? 0 : 2 * n_out)) * 8, s, 0);
  if (rc != GLX_OK) return rc;
  a.src = d;
  a.rng_rows = rng_rows ? d + batch : nullptr;
  a.nbr_out = direct ? m_nbr : d + 2 * (size_t)batch;
  a.eid_out = direct ? m_eid : d + 2 * (size_t)batch + n_out;
  hipError_t e = hipMemcpyAsync(d, src, (size_t)batch * 8, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && rng_rows) e = hipMemcpyAsync(d + batch, rng_rows, (size_t)batch * 8, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    rc = sample_device(g, sampler, a, padding_mode, s);
    if (rc == GLX_OK && !direct) {
      e = hipMemcpyAsync(nbr_out, a.nbr_out, n_out * 8, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipMemcpyAsync(eid_out, a.eid_out, n_out * 8, hipMemcpyDeviceToHost, s);
    }
  }
  hipError_t e2 = hipStreamSynchronize(s);
  glx_scratch_free(d, s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(e);
  GLX_HIP(e2);
  return GLX_OK;
}

extern "C" int glx_sample(const glx_graph* g, int sampler, const int64_t* src, int32_t batch,
                          int32_t k, int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                          uint64_t call_counter, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                          void* stream) {
  return glx_sample_ex(g, sampler, src, nullptr, batch, k, padding_mode, default_neighbor_id, seed,
                       call_counter, nbr_out, eid_out, ptr_kind, stream);
}

extern "C" int glx_sample_hops(const glx_graph* const* graphs, int32_t num_hops, int sampler,
                               const int64_t* seeds, int32_t batch, const int32_t* fanouts,
                               int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                               uint64_t call_counter, int64_t* const* nbr_out,
                               int64_t* const* eid_out, int ptr_kind, void* stream) {
  GLX_REQUIRE(graphs && fanouts && nbr_out, "NULL argument");
  GLX_REQUIRE(num_hops >= 1 && num_hops <= 16, "num_hops must be in [1, 16]");
  GLX_REQUIRE(batch >= 0, "negative batch");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  int64_t rows = batch;
  for (int32_t h = 0; h < num_hops; ++h) {
    GLX_REQUIRE(graphs[h] != nullptr && nbr_out[h] != nullptr, "hop %d: NULL graph / output", h);
    GLX_REQUIRE(graphs[h]->device == graphs[0]->device, "all hops must live on one device");
    GLX_REQUIRE(fanouts[h] >= 0, "negative fanout");
    GLX_REQUIRE(rows <= INT32_MAX && rows * fanouts[h] <= INT32_MAX,
                "hop %d exceeds int32 slots (tensor.h:47)", h);
    rows *= fanouts[h];
  }
  if (ptr_kind == GLX_PTR_DEVICE) {
    const int64_t* frontier = seeds;
    int64_t n = batch;
    for (int32_t h = 0; h < num_hops; ++h) {
      int64_t* e = eid_out ? eid_out[h] : nullptr;
      GLX_REQUIRE(e != nullptr || n * fanouts[h] == 0, "device mode needs eid_out[%d]", h);
      int rc = glx_sample(graphs[h], sampler, frontier, (int32_t)n, fanouts[h], padding_mode,
                          default_neighbor_id, seed, call_counter + (uint64_t)h, nbr_out[h], e,
                          GLX_PTR_DEVICE, stream);
      if (rc != GLX_OK) return rc;
      frontier = nbr_out[h];
      n *= fanouts[h];
    }
    return GLX_OK;
  }
  // Host pointers: one upload of the seeds, every hop sampled from the previous
  // hop's device buffer, one download per hop output.
  GlxDeviceGuard guard(graphs[0]->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", graphs[0]->device);
  hipStream_t s = glx_host_call_stream(stream, graphs[0]->device);
  size_t total = (size_t)batch;
  {
    int64_t n = batch;
    for (int32_t h = 0; h < num_hops; ++h) {
      n *= fanouts[h];
      total += 2 * (size_t)n;
    }
  }
  int64_t* d = nullptr;
  int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), total * 8, s, 0);
  if (rc != GLX_OK) return rc;
  // Errors are collected, not returned on the spot: copies already queued into the caller's
  // buffers must drain (and the shared staging buffer must be idle) before this call returns.
  hipError_t e = hipMemcpyAsync(d, seeds, (size_t)batch * 8, hipMemcpyHostToDevice, s);
  const int64_t* frontier = d;
  int64_t* cursor = d + batch;
  int64_t n = batch;
  for (int32_t h = 0; h < num_hops && e == hipSuccess && rc == GLX_OK; ++h) {
    const int64_t slots = n * fanouts[h];
    int64_t* dn = cursor;
    int64_t* de = cursor + slots;
    cursor += 2 * slots;
    if (slots > 0) {
      rc = glx_sample(graphs[h], sampler, frontier, (int32_t)n, fanouts[h], padding_mode,
                      default_neighbor_id, seed, call_counter + (uint64_t)h, dn, de, GLX_PTR_DEVICE, s);
      if (rc != GLX_OK) break;
      e = hipMemcpyAsync(nbr_out[h], dn, (size_t)slots * 8, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess && eid_out && eid_out[h]) {
        e = hipMemcpyAsync(eid_out[h], de, (size_t)slots * 8, hipMemcpyDeviceToHost, s);
      }
    }
    frontier = dn;
    n = slots;
  }
  hipError_t e2 = hipStreamSynchronize(s);
  glx_scratch_free(d, s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(e);
  GLX_HIP(e2);
  return GLX_OK;
}
