// glx distributed store: the edge-cut multi-GPU path of the hot operators.
// Replaces DistributeRunner<Req, Res>::Run (graphlearn/src/core/runner/op_runner.h:60-152):
//   Partition (core/partition/hash_partitioner.h:33-92: shard = llabs(id) % P, stable)
//   -> one sub-request per shard -> Process on the owner -> Stitch
//   (core/partition/stitcher.h:67-107; service/request/aggregating_request.cc:117-213)
// with device kernels on both sides of two transport exchanges (glx_comm.hip: RCCL
// send/recv groups over xGMI).  Requests and responses never leave HBM.
//
// Sampling: request rows are bucketed by owner (glx_partition), each row travels with its
// index in the original request (the Sticker value) so the owner draws from THAT row's
// random stream, results travel back and are scattered by the Sticker: bit-identical to the
// unpartitioned sampler for every shard count.
//
// Aggregation (design H of SURVEY.md 8(e), north_star's halo-vertex feature exchange): the
// reduce runs on the requester in request order over rows that come from three places --
// its own shard, the replica of hot rows every GPU keeps (power-law graphs: a few percent of
// the rows take most accesses), and the halo: the remaining remote ids, DEDUPLICATED on the
// device (open-addressing set + per-owner compaction, no sort), fetched from their owners.
// Only the cold tail crosses the links, once per distinct id.  One host synchronisation per
// call (the P x P count matrix).
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include <chrono>

#include <rocprim/rocprim.hpp>

#include "glx_comm.h"

namespace {

constexpr int kMaxWorld = 64;
constexpr int kMaxProbe = 128;

// ---------------------------------------------------------------- memory -----
// Grow-only device arena; growth synchronises the device (hipFree), which also makes it safe
// to drop a buffer earlier work of this store may still read.
struct Arena {
  char* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return GLX_OK;
    if (p) GLX_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4;
    if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
    want = (want + 255) & ~(size_t)255;
    GLX_HIP(hipMalloc(reinterpret_cast<void**>(&p), want));
    cap = want;
    return GLX_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Carves 256-byte aligned pieces out of one arena allocation.
struct Carver {
  size_t at = 0;
  size_t take(size_t bytes) {
    const size_t off = at;
    at += (bytes + 255) & ~(size_t)255;
    return off;
  }
};

__device__ __forceinline__ int32_t dist_owner(int64_t id, int32_t P) {
  const uint64_t a = id < 0 ? (uint64_t)0 - (uint64_t)id : (uint64_t)id;  // llabs (hash_partitioner.h:90-92)
  if ((P & (P - 1)) == 0) return (int32_t)(a & (uint64_t)(P - 1));  // 1, 2, 4, 8 ranks: a mask (P is kernel-uniform)
  // a 64-bit remainder is ~100 instructions on this ISA, a 32-bit one a quarter of that -- and ids nearly always fit
  if (a <= 0xffffffffull) return (int32_t)((uint32_t)a % (uint32_t)P);
  return (int32_t)(a % (uint64_t)P);
}

// ------------------------------------------------------------- sampling ------
__global__ void glx_dist_gather_i64_kernel(const int64_t* __restrict__ in, const int64_t* __restrict__ order,
                                           int64_t n, int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[order[i]];
}

// A request's scalar parameters travel with it: the owner serves requester q's rows with q's
// seed / call counter / flags, not its own (every rank drives its own request).
struct ReqParams {
  int64_t v[12];
};
__global__ void glx_dist_set_params_kernel(int64_t* dst, ReqParams p, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}

// Stitcher::DoStitch for both response tensors at once: row i of the bucketed answer goes to
// request row order[i].
__global__ __launch_bounds__(256) void glx_dist_stitch2_kernel(const int64_t* __restrict__ nbr_in,
                                                               const int64_t* __restrict__ eid_in,
                                                               const int64_t* __restrict__ order, int64_t n, int32_t k,
                                                               int64_t* __restrict__ nbr_out,
                                                               int64_t* __restrict__ eid_out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * k) return;
  const int64_t i = t / k;
  const int32_t c = (int32_t)(t - i * k);
  const int64_t o = order[i] * k + c;
  nbr_out[o] = nbr_in[t];
  eid_out[o] = eid_in[t];
}

// ----------------------------------------------------------- aggregation -----
// Counter block (int32, zeroed before every resolve):
//   [0, P)     distinct halo ids per owner         [P, 2P)   compaction cursors
//   [2P]       overflow flag (set table too small)  [2P+1..3] ids served by replica / own shard /
//   [2P+4 ..]  exclusive offsets per owner (P + 1)            remote (with repeats)
//   [3P+5]     distinct halo ids inserted so far (all owners)
//   [3P+6]     workgroups of the resolve kernel that have finished (the last one turns the counts into offsets)
// The replica's id map with key and row in ONE 16-byte slot: a probe is a single load (the generic GlxIdMap
// keeps keys and rows in two arrays = two dependent random loads per hit, and every id of a request probes it).
struct PackedSlot {
  int64_t key;
  int32_t row;
  int32_t pad_;
};
struct PackedMap {
  const PackedSlot* slots;
  uint64_t mask;
};
__global__ void glx_dist_pack_map_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                         uint64_t cap, PackedSlot* __restrict__ out) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) out[i] = PackedSlot{keys[i], vals[i], 0};
}

struct RankWord {
  uint64_t bits;
  uint32_t rank;  // set bits in all words before this one
  uint32_t pad_;
};

struct ResolveArgs {
  PackedMap cache_map;
  GlxIdMap own_map;
  const int64_t* ids;
  int64_t n;
  int32_t* loc;
  int64_t* tkeys;
  uint64_t tmask;
  int32_t* ctr;
  int32_t P, me;
  int32_t cache_base;
  int32_t has_cache;
  // replica membership by rank-select when the hot ids are small non-negative numbers (else nullptr: the packed hash
  // map): bit `id % 64` of bm_member[id / 64].bits = id is replicated, its row = .rank + the set bits below it
  // (the replica's rows are in ascending id order); bm_valid clears the ids their owner does not know.
  const RankWord* bm_member;  // one 16-byte record per 64 ids: ONE sector per lookup
  const uint64_t* bm_valid;   // nullptr when every hot id is known to its owner (the usual case)
  int64_t bm_max;
  int32_t insert_limit;  // distinct ids the set takes before it counts as too small (60 % of its slots)
  int32_t max_probe;     // probes before a lookup gives up on a set that is too small; unbounded at the safe size
  // kQueue: where the remote cold ids sit in the request, so that pass 4 rewrites those entries only (round 6; it used
  // to stream all of loc for the few percent that were pending).  Workgroup b owns cold_idx[b * region, (b + 1) * region)
  // -- region = the ids one workgroup resolves, so its list cannot overflow and no global counter is shared -- and
  // leaves its length in cold_cnt[b].
  int32_t* cold_idx;
  int32_t* cold_cnt;
  int64_t region;
  // pass 2, run by the LAST workgroup to finish (a launch of its own for one thread's work was the price of a small
  // request's whole resolve): the values every rank shares, and the requester's default_attr behind them
  int64_t* vals;
  float default_attr;
  // One row of 3 + P int32 per workgroup: its replica / own / remote id counts and its new distinct ids per owner.
  // The LAST workgroup adds the rows up.  (Rounds 1-5, and this round's first version, raised shared counters with
  // 5 + P same-address atomics per workgroup; the workgroups all finish together, the atomics queue at the memory side
  // one behind the other, and the kernel ended with ~0.1 us per WORKGROUP of nothing else: 235 us at 1024 workgroups,
  // 380 at 2048, 1147 at 8192 for the same 18 M ids -- profiles/r06/resolve_set_probe.txt.)
  int32_t* part;
  int32_t peek;  // 1: load a set slot before trying to claim it (resolve_cold)
  int32_t own_first;  // 1: ids this rank owns skip the replica lookup (arithmetic own-shard ids)
};

// Pass 2 (one thread): per-owner offsets, and the values every rank shares:
// vals[0..P) = ids requested from each owner, [P] overflow, [P+1] distinct total,
// [P+2..4] replica / own / remote id counts, [P+5] the requester's default_attr (float bits).
// c[] = the counter block as read (LDS or global); writes the offsets / cleared cursors to ctr, the values to vals.
__device__ __forceinline__ void dist_offsets_body(const int32_t* c, int32_t* ctr, int32_t P, int64_t* vals, float default_attr) {
  int32_t* off = ctr + 2 * P + 4;
  int32_t total = 0;
  for (int32_t p = 0; p < P; ++p) {
    off[p] = total;
    vals[p] = c[p];
    total += c[p];
    ctr[P + p] = 0;
  }
  off[P] = total;
  vals[P] = c[2 * P];
  vals[P + 1] = total;
  vals[P + 2] = c[2 * P + 1];
  vals[P + 3] = c[2 * P + 2];
  vals[P + 4] = c[2 * P + 3];
  vals[P + 5] = (int64_t)(uint32_t)__float_as_uint(default_attr);
}

// ... by the last workgroup of the resolve kernel (all 256 threads call it): adds up the workgroups' rows -- written
// with agent-scope atomic stores and acknowledged before their owners took their tickets, read here with agent-scope
// atomic loads (a plain load could be served by this CU's L1 or this XCD's L2 from a line fetched earlier) -- into the
// counter block, decides whether the set was too small, and publishes offsets and shared values.  The counts check
// themselves (replica + own + remote ids = n): a row that were still on its way is waited for, not published.
__device__ __forceinline__ void dist_offsets_last_block(const int32_t* part, int32_t nblocks, int32_t* ctr, int32_t P,
                                                        int64_t* vals, float default_attr, int64_t n, int32_t insert_limit) {
  __shared__ int32_t s_c[3 * kMaxWorld + 8];
  __shared__ int32_t s_ok;
  const int32_t S = 3 + P;
  for (int spin = 0; spin < 1024; ++spin) {
    if ((int32_t)threadIdx.x < 3 * P + 8) s_c[threadIdx.x] = 0;
    __syncthreads();
    for (int32_t c = 0; c < S; ++c) {
      int32_t sum = 0;
      for (int32_t b = threadIdx.x; b < nblocks; b += 256) {
        sum += __hip_atomic_load(&part[(int64_t)b * S + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      // stats live at [2P + 1, 2P + 4), per-owner counts at [0, P)
      if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&s_c[c < 3 ? 2 * P + 1 + c : c - 3], sum);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int64_t ids = (int64_t)s_c[2 * P + 1] + s_c[2 * P + 2] + s_c[2 * P + 3];
      s_ok = ids == n;
    }
    __syncthreads();
    if (s_ok) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) {
    int32_t distinct = 0;
    for (int32_t p = 0; p < P; ++p) distinct += s_c[p];
    // too small: more distinct ids than the set takes, or a probe sequence that gave up (its flag is an atomic in ctr)
    s_c[2 * P] = (distinct > insert_limit || atomicAdd(&ctr[2 * P], 0) != 0) ? 1 : 0;
    for (int32_t p = 0; p < P; ++p) ctr[p] = s_c[p];  // the compaction pass reads the per-owner counts through `off`
    ctr[2 * P] = s_c[2 * P];
    ctr[3 * P + 5] = distinct;
    dist_offsets_body(s_c, ctr, P, vals, default_attr);
  }
}

// Pass 1: every id -> a virtual row (own shard / replica), -1 (default row), or -(h + 2) when
// it is remote and cold: h = its slot in the open-addressing set of distinct halo ids.
// New distinct ids are counted per owner in LDS and flushed to the global counters every few
// iterations: per-wave atomics on P global addresses (a quarter of a million waves on eight
// counters) serialised the whole kernel -- 10 ms for a 16 M-id request.
constexpr int kFlushIds = 2048;  // ids a block resolves between two flushes of its counters
// kQueue (a replica is attached, so ids it does not hold are a small share of the request): those ids are queued per
// wave in LDS and resolved 64 at a time with every lane busy.  Their path -- owner, own-shard probe or insert into the
// halo set: three or four dependent memory round trips -- otherwise runs with one or two lanes of a wave, and every
// wave-iteration waits for it: 0.67 ms instead of 0.1 for an 18 M-id request at P = 8 with 4 % of the ids off the replica.
template <int kIds, bool kQueue>
__global__ __launch_bounds__(256) void glx_dist_resolve_kernel(ResolveArgs a) {
  __shared__ int32_t s_stat[3];
  __shared__ int32_t s_cnt[kMaxWorld];
  __shared__ int32_t s_sum;
  __shared__ int32_t s_void;  // the overflow flag as of the last flush (reading the global flag per id made one
                              // L2 address the hot spot of the kernel)
  __shared__ int32_t s_list_n;  // kQueue: entries of this workgroup's cold list
  __shared__ int32_t s_last;
  if (threadIdx.x == 0) {
    s_void = 0;
    s_sum = 0;
    s_list_n = 0;
  }
  if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
  if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int32_t n_hit = 0, n_own = 0, n_cold = 0;
  int it = 0;
  __shared__ int64_t s_qid[kQueue ? 4 : 1][kQueue ? 128 : 1];
  __shared__ int32_t s_qi[kQueue ? 4 : 1][kQueue ? 128 : 1];
  int64_t* qid = s_qid[kQueue ? (threadIdx.x >> 6) : 0];
  int32_t* qi = s_qi[kQueue ? (threadIdx.x >> 6) : 0];
  int qn = 0;  // queued ids of this wave (wave-uniform)
  const bool own_inline = a.own_map.keys == nullptr;  // arithmetic (or identity) own-shard ids

  // One id off the replica: own shard -> its row; remote -> its slot in the halo set (winner: the first to claim it).
  auto resolve_cold = [&](int64_t id, int32_t& owner, bool& winner, int32_t voided) -> int32_t {
    int32_t out = -1;
    owner = dist_owner(id, a.P);
    if (owner == a.me || id == GLX_EMPTY_KEY) {
      const int64_t rr = glx_row_of(a.own_map, id);
      out = rr >= 0 ? (int32_t)rr : -1;
      ++n_own;
    } else if (voided != 0) {
      ++n_cold;  // the set already overflowed: this pass is void, the host retries with a larger one
    } else {
      ++n_cold;
      uint64_t h = glx_mix64((uint64_t)id) & a.tmask;
      int probes = 0;
      while (true) {
        // A slot's key never changes once set, so a plain load that returns a key is final; one that returns "empty"
        // may be stale (this XCD's L2 is not coherent with the others') and only then is the compare-and-swap paid.
        // Most cold ids of a request repeat (0.63 M occurrences of 0.25 M distinct ids in the headline's 18 M-id
        // request at P = 8): their later occurrences find the key with a load.
        int64_t cur = a.peek ? *reinterpret_cast<const volatile int64_t*>(&a.tkeys[h]) : GLX_EMPTY_KEY;
        if (cur == GLX_EMPTY_KEY) {
          cur = (int64_t)atomicCAS(reinterpret_cast<unsigned long long*>(&a.tkeys[h]), (unsigned long long)GLX_EMPTY_KEY,
                                   (unsigned long long)id);
          if (cur == GLX_EMPTY_KEY) {
            winner = true;  // the first to claim the slot
            out = -(int32_t)h - 2;
            break;
          }
        }
        if (cur == id) {
          out = -(int32_t)h - 2;
          break;
        }
        h = (h + 1) & a.tmask;
        if (++probes > a.max_probe) {  // the set is too small: the host retries with a larger one
          atomicExch(&a.ctr[2 * a.P], 1);
          break;
        }
      }
    }
    return out;
  };
  // new distinct ids: one LDS atomic per (wave, owner)
  auto count_winners = [&](bool winner, int32_t owner) {
    uint64_t pending = __ballot(winner);
    while (pending) {
      const int leader = __ffsll((long long)pending) - 1;
      const int32_t o = __shfl(owner, leader);
      const uint64_t same = __ballot(winner && owner == o);
      if (lane == leader) atomicAdd(&s_cnt[o], __popcll(same));
      pending &= ~same;
    }
  };
  auto serve_queue = [&](int count) {  // lanes [0, count) resolve the head of the wave's queue
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // entries other lanes queued
    __builtin_amdgcn_wave_barrier();
    bool winner = false;
    int32_t owner = 0;
    // the overflow flag once per 64 ids (per id it made one L2 address the hot spot of the kernel); no block-wide
    // flush in this mode: its barriers made every wave wait for whichever wave was serving its queue (0.51 -> 0.19 ms)
    const int32_t voided = __shfl(lane == 0 ? __atomic_load_n(&a.ctr[2 * a.P], __ATOMIC_RELAXED) : 0, 0);
    int32_t out = 0;
    if (lane < count) {
      out = resolve_cold(qid[lane], owner, winner, voided);
      a.loc[qi[lane]] = out;
    }
    count_winners(winner, owner);
    // the entries pass 4 must rewrite (halo set slots): remembered per workgroup, one LDS atomic per batch
    const uint64_t pend = __ballot(lane < count && out <= -2);
    if (pend && a.cold_idx) {
      int32_t at = 0;
      if (lane == 0) at = atomicAdd(&s_list_n, __popcll(pend));
      at = __shfl(at, 0);
      if (lane < count && out <= -2) {
        a.cold_idx[(int64_t)blockIdx.x * a.region + at + __popcll(pend & ((1ull << lane) - 1ull))] = qi[lane];
      }
    }
  };
  // kIds ids per thread per iteration: the loads of the replica lookup (the common case: a hop-2 request finds
  // ~96 % of its ids there) are independent and issued back to back; the rare rest is handled id by id.
  for (int64_t base = blockIdx.x * (256ll * kIds); base < a.n; base += gridDim.x * (256ll * kIds), ++it) {
    int64_t id[kIds];
    int64_t r[kIds];
    bool mine[kIds];
#pragma unroll
    for (int j = 0; j < kIds; ++j) {
      const int64_t i = base + j * 256 + threadIdx.x;
      id[j] = i < a.n ? a.ids[i] : GLX_EMPTY_KEY;
      r[j] = -1;
      mine[j] = false;
    }
    if (a.bm_member) {
      RankWord w[kIds];
#pragma unroll
      for (int j = 0; j < kIds; ++j) {
        // own_first (world size 1; see the launch site): an id this rank owns is read from its own shard whether or
        // not the replica holds a copy (same row, same bytes) -- with arithmetic own-shard ids a remainder and a
        // division instead of the divergent 16-byte lookup.  Its lane reads record 0 (one shared line).
        mine[j] = kQueue && own_inline && a.own_first && id[j] != GLX_EMPTY_KEY && dist_owner(id[j], a.P) == a.me;
        const bool in = id[j] >= 0 && id[j] <= a.bm_max && !mine[j];
        w[j] = a.bm_member[in ? (id[j] >> 6) : 0];
        if (!in) w[j].bits = 0;
      }
#pragma unroll
      for (int j = 0; j < kIds; ++j) {
        const uint64_t bit = 1ull << (id[j] & 63);
        if (w[j].bits & bit) {
          if (a.bm_valid == nullptr || (a.bm_valid[id[j] >> 6] & bit)) r[j] = (int64_t)w[j].rank + __popcll(w[j].bits & (bit - 1));
        }
      }
    } else if (a.has_cache) {
      PackedSlot first[kIds];
      uint64_t h[kIds];
#pragma unroll
      for (int j = 0; j < kIds; ++j) {
        h[j] = glx_mix64((uint64_t)id[j]) & a.cache_map.mask;
        first[j] = a.cache_map.slots[h[j]];
      }
#pragma unroll
      for (int j = 0; j < kIds; ++j) {
        if (id[j] == GLX_EMPTY_KEY) continue;
        PackedSlot sl = first[j];
        uint64_t hh = h[j];
        while (sl.key != id[j] && sl.key != GLX_EMPTY_KEY) {
          hh = (hh + 1) & a.cache_map.mask;
          sl = a.cache_map.slots[hh];
        }
        if (sl.key == id[j]) r[j] = sl.row;
      }
    }
#pragma unroll
    for (int j = 0; j < kIds; ++j) {
      const int64_t i = base + j * 256 + threadIdx.x;
      bool winner = false, cold = false;
      int32_t owner = 0;
      if (i < a.n) {
        if (r[j] >= 0) {
          a.loc[i] = a.cache_base + (int32_t)r[j];
          ++n_hit;
        } else if (kQueue && own_inline && (mine[j] || id[j] == GLX_EMPTY_KEY || dist_owner(id[j], a.P) == a.me)) {
          // off the replica but this rank's own, and the own shard's ids are arithmetic: the row is a division, no
          // memory round trip to wait for -- nothing the queue could hide (every id off the replica of a world-1
          // request, an eighth of them at P = 8, used to queue for it: 0.185 -> 0.135 ms for the 16.4 M ids of the
          // headline's hop-2 request at world size 1, profiles/r05/world1_partition_resolve.txt)
          const int64_t rr = glx_row_of(a.own_map, id[j]);
          a.loc[i] = rr >= 0 ? (int32_t)rr : -1;
          ++n_own;
        } else if (kQueue) {
          cold = true;
        } else {
          a.loc[i] = resolve_cold(id[j], owner, winner, s_void);
        }
      }
      if (kQueue) {
        const uint64_t m = __ballot(cold);
        if (cold) {
          const int at = qn + __popcll(m & ((1ull << lane) - 1ull));
          qid[at] = id[j];
          qi[at] = (int32_t)i;
        }
        qn += __popcll(m);
        if (qn >= 64) {
          serve_queue(64);
          qn -= 64;
          const int64_t mv_id = lane < qn ? qid[64 + lane] : 0;
          const int32_t mv_i = lane < qn ? qi[64 + lane] : 0;
          __builtin_amdgcn_wave_barrier();
          if (lane < qn) {
            qid[lane] = mv_id;
            qi[lane] = mv_i;
          }
        }
      } else {
        count_winners(winner, owner);
      }
    }
    if (!kQueue && (it % (kFlushIds / (256 * kIds))) == kFlushIds / (256 * kIds) - 1) {
      // flush: P global atomics per block, and the running total decides early whether the set is too small
      // ONE global atomic per flush (the running total decides early whether the set is too small); the per-owner
      // counts stay in LDS until the block ends -- flushing them too put 8 same-address atomics per flush and block
      // on the path of a request with remote ids: 0.67 ms instead of 0.2 for 18 M ids at P = 8
      __syncthreads();
      if (threadIdx.x == 0) {
        int32_t sum = 0;
        for (int32_t p = 0; p < a.P; ++p) sum += s_cnt[p];
        const int32_t fresh = sum - s_sum;
        s_sum = sum;
        if (fresh) {
          const int32_t before = atomicAdd(&a.ctr[3 * a.P + 5], fresh);
          if (before + fresh > a.insert_limit) atomicExch(&a.ctr[2 * a.P], 1);
        }
        s_void = __atomic_load_n(&a.ctr[2 * a.P], __ATOMIC_RELAXED);
      }
      __syncthreads();
    }
  }
  if (kQueue && qn > 0) serve_queue(qn);
  atomicAdd(&s_stat[0], n_hit);
  atomicAdd(&s_stat[1], n_own);
  atomicAdd(&s_stat[2], n_cold);
  __syncthreads();
  // This workgroup's row: no address is shared with another workgroup.
  {
    int32_t* row = a.part + (int64_t)blockIdx.x * (3 + a.P);
    if (threadIdx.x < 3) __hip_atomic_store(&row[threadIdx.x], s_stat[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < a.P) __hip_atomic_store(&row[3 + threadIdx.x], s_cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    if (!kQueue) {  // (this variant keeps its running total for the early "too small" test of its flushes)
      int32_t sum = 0;
      for (int32_t p = 0; p < a.P; ++p) sum += s_cnt[p];
      const int32_t fresh = sum - s_sum;
      if (fresh) {
        const int32_t before = atomicAdd(&a.ctr[3 * a.P + 5], fresh);
        if (before + fresh > a.insert_limit) atomicExch(&a.ctr[2 * a.P], 1);
      }
    }
    if (kQueue && a.cold_cnt) a.cold_cnt[blockIdx.x] = s_list_n;
  }
  // Pass 2 in the same launch: the last workgroup to take a ticket has every other workgroup's row before it.  No
  // cache-flushing fence is involved (an agent-scope fence writes this XCD's L2 back: two per workgroup cost the kernel
  // 50 us): the rows are written with agent-scope atomic stores, a workgroup waits for their acknowledgement
  // (s_waitcnt) before it takes its ticket -- the ONE same-address atomic it issues -- and the last workgroup reads the
  // rows with agent-scope atomic loads.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&a.ctr[3 * a.P + 6], 1) == (int32_t)gridDim.x - 1;
  __syncthreads();
  if (s_last) dist_offsets_last_block(a.part, (int32_t)gridDim.x, a.ctr, a.P, a.vals, a.default_attr, a.n, a.insert_limit);
}

// Pass 2 for a request without ids (no resolve launch): the shared values of an empty request.  The counter block need
// not be cleared first: kZero treats it as zeros and leaves it so.
__global__ void glx_dist_offsets_kernel(int32_t* ctr, int32_t P, int64_t* vals, float default_attr) {
  if (threadIdx.x != 0) return;
  for (int32_t i = 0; i < 3 * P + 8; ++i) ctr[i] = 0;
  dist_offsets_body(ctr, ctr, P, vals, default_attr);
}

// Pass 4 over the cold lists (kQueue resolve): workgroup b rewrites the entries workgroup b of the resolve listed.
__global__ __launch_bounds__(256) void glx_dist_finalize_list_kernel(int32_t* __restrict__ loc,
                                                                     const int32_t* __restrict__ cold_idx,
                                                                     const int32_t* __restrict__ cold_cnt, int64_t region,
                                                                     const int32_t* __restrict__ tvals, int32_t halo_base) {
  const int32_t m = cold_cnt[blockIdx.x];
  const int32_t* mine = cold_idx + (int64_t)blockIdx.x * region;
  for (int32_t j = threadIdx.x; j < m; j += 256) {
    const int32_t i = mine[j];
    const int32_t v = loc[i];
    if (v <= -2) loc[i] = halo_base + tvals[-(v + 2)];
  }
}

// Pass 3: compact the set into per-owner buckets (the ids each owner is asked for) and give
// every member its halo row = position in that concatenation.  A block takes a tile of kAssignTile
// slots at a time -- kAssignPer per thread, read ONCE and kept in registers: count its members per owner in LDS,
// reserve the block's ranges with P global atomics, then place the members (order inside a bucket is arbitrary;
// results do not depend on it).  (Rounds 2-5 read every tile twice, 4096 slots per tile: 0.10 ms for the 8 M slots
// behind the headline's 18 M-id request, a few hundred workgroups of sixteen dependent passes each.)
constexpr int kAssignPer = 4;
constexpr int kAssignTile = 256 * kAssignPer;
__global__ __launch_bounds__(256) void glx_dist_assign_kernel(const int64_t* __restrict__ tkeys, uint64_t tcap,
                                                              int32_t* __restrict__ tvals, int32_t* ctr, int32_t P,
                                                              int64_t* __restrict__ cold_ids) {
  __shared__ int32_t s_cnt[kMaxWorld];
  __shared__ int32_t s_base[kMaxWorld];
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (1ull << lane) - 1ull;
  int32_t* cursor = ctr + P;
  const int32_t* off = ctr + 2 * P + 4;
  if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t tile = blockIdx.x * (uint64_t)kAssignTile; tile < tcap; tile += gridDim.x * (uint64_t)kAssignTile) {
    int64_t key[kAssignPer];
    int32_t owner[kAssignPer];
    bool any = false;
#pragma unroll
    for (int j = 0; j < kAssignPer; ++j) {
      const uint64_t h = tile + j * 256 + threadIdx.x;
      key[j] = h < tcap ? tkeys[h] : GLX_EMPTY_KEY;
      any = any || key[j] != GLX_EMPTY_KEY;
    }
    if (!__syncthreads_or(any)) continue;  // an empty tile (the set is sized for a multiple of what it holds)
    // count
#pragma unroll
    for (int j = 0; j < kAssignPer; ++j) {
      const bool live = key[j] != GLX_EMPTY_KEY;
      owner[j] = live ? dist_owner(key[j], P) : 0;
      uint64_t pending = __ballot(live);
      while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int32_t o = __shfl(owner[j], leader);
        const uint64_t same = __ballot(live && owner[j] == o);
        if (lane == leader) atomicAdd(&s_cnt[o], __popcll(same));
        pending &= ~same;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < P) {
      const int32_t c = s_cnt[threadIdx.x];
      s_base[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0;
      s_cnt[threadIdx.x] = 0;  // becomes the running offset inside the block's range
    }
    __syncthreads();
    // place
#pragma unroll
    for (int j = 0; j < kAssignPer; ++j) {
      const uint64_t h = tile + j * 256 + threadIdx.x;
      const bool live = key[j] != GLX_EMPTY_KEY;
      uint64_t pending = __ballot(live);
      while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int32_t o = __shfl(owner[j], leader);
        const uint64_t same = __ballot(live && owner[j] == o);
        int32_t start = 0;
        if (lane == leader) start = atomicAdd(&s_cnt[o], __popcll(same));
        start = __shfl(start, leader);
        if (live && owner[j] == o) {
          const int32_t pos = off[o] + s_base[o] + start + __popcll(same & lt);
          tvals[h] = pos;
          cold_ids[pos] = key[j];
        }
        pending &= ~same;
      }
    }
    __syncthreads();
    if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
}

// Pass 4: pending entries -> halo virtual rows.
__global__ void glx_dist_finalize_kernel(int32_t* __restrict__ loc, int64_t n, const int32_t* __restrict__ tvals,
                                         int32_t halo_base) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = loc[i];
  if (v <= -2) loc[i] = halo_base + tvals[-(v + 2)];
}

// ... and the counter block of the resolve passes, cleared by the same launch (it was a memset of its own)
__global__ void glx_dist_fill_keys_kernel(int64_t* keys, uint64_t cap, int32_t* ctr, int32_t nctr) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && (int32_t)threadIdx.x < nctr) ctr[threadIdx.x] = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) keys[i] = GLX_EMPTY_KEY;
}

// LookupNodes over resolved virtual rows: G lanes per output row.
struct GatherArgs {
  GlxRowSource src[3];
  int32_t base1, base2;
  const int32_t* vrows;
  int64_t n;
  int32_t dim;
  float default_attr;
  float* out;
};
__global__ __launch_bounds__(256) void glx_dist_gather_rows_kernel(GatherArgs a, int G) {
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int c = threadIdx.x % G;
  if (gid >= a.n) return;
  const int32_t row = a.vrows[gid];
  const float* src = nullptr;
  if (row >= 0) {
    if (row < a.base1) src = a.src[0].X + glx_swizzle_row(row, a.src[0].swizzle_rows) * a.src[0].stride;
    else if (row < a.base2) src = a.src[1].X + glx_swizzle_row(row - a.base1, a.src[1].swizzle_rows) * a.src[1].stride;
    else src = a.src[2].X + (int64_t)(row - a.base2) * a.src[2].stride;
  }
  float* o = a.out + gid * (int64_t)a.dim;
  for (int32_t col = c; col < a.dim; col += G) o[col] = src ? src[col] : a.default_attr;
}

// Replica build: an id its owner does not know must stay unknown (a request's own
// default_attr applies to it), so it is withheld from the replica's id map.
__global__ void glx_dist_known_kernel(GlxIdMap map, const int64_t* __restrict__ ids, int64_t n,
                                      int64_t* __restrict__ known) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) known[i] = glx_row_of(map, ids[i]) >= 0 ? 1 : 0;
}
__global__ void glx_dist_mask_ids_kernel(int64_t* __restrict__ ids, const int64_t* __restrict__ known, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && !known[i]) ids[i] = GLX_EMPTY_KEY;
}

// sorted[n] ascending (masked[i] = sorted[i], or GLX_EMPTY_KEY when the owner does not know the id):
// member / valid bits; *dup is set when two equal ids are adjacent (the rank of an id would not be its row).
__global__ void glx_dist_bitmap_set_kernel(const int64_t* __restrict__ sorted, const int64_t* __restrict__ masked,
                                           int64_t n, unsigned long long* __restrict__ member,
                                           unsigned long long* __restrict__ valid, int* __restrict__ flags) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = sorted[i];
  if (i > 0 && sorted[i - 1] == id) flags[0] = 1;  // listed twice
  const unsigned long long bit = 1ull << (id & 63);
  atomicOr(&member[id >> 6], bit);
  if (masked[i] != GLX_EMPTY_KEY) atomicOr(&valid[id >> 6], bit);
  else flags[1] = 1;  // an id its owner does not know
}
__global__ void glx_dist_bitmap_popc_kernel(const uint64_t* __restrict__ member, int64_t words, uint32_t* __restrict__ cnt) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < words) cnt[i] = (uint32_t)__popcll(member[i]);
}
__global__ void glx_dist_bitmap_pack_kernel(const uint64_t* __restrict__ member, const uint32_t* __restrict__ rank,
                                            int64_t words, RankWord* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < words) out[i] = RankWord{member[i], rank[i], 0};
}

// ---- graph replica built from the shards (glx_dist_build_graph_replica) ----
// degree of each of this owner's hot vertices (0 when the shard has no such row)
__global__ void glx_dist_rep_deg_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ ids,
                                        int64_t n, int64_t* __restrict__ deg) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = glx_row_of(map, ids[i]);
  deg[i] = row >= 0 ? row_ptr[row + 1] - row_ptr[row] : 0;
}
// one wave per hot vertex: its slots, in storage order, into the piece this owner ships
__global__ __launch_bounds__(256) void glx_dist_rep_rows_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr,
                                                                const GlxAdj* __restrict__ adj,
                                                                const float* __restrict__ weight,
                                                                const int64_t* __restrict__ ids, int64_t n,
                                                                const int64_t* __restrict__ off, int64_t* __restrict__ col,
                                                                int64_t* __restrict__ eid, float* __restrict__ w) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t row = glx_row_of(map, ids[i]);
  if (row < 0) return;
  const int64_t s = row_ptr[row], d = row_ptr[row + 1] - s, o = off[i];
  for (int64_t j = lane; j < d; j += 64) {
    const GlxAdj a = adj[s + j];
    col[o + j] = a.nbr;
    eid[o + j] = a.eid;
    if (w) w[o + j] = weight[s + j];
  }
}

// ---- partitioned FullSampler (glx_dist_sample_full_*) ----
// row `order[pos]` of the request takes the `deg[pos]` values that arrived at `src_off[pos]` (rows arrive in bucketed
// order, owner by owner) to its place `dst_off[order[pos]]`; one wave per row
__global__ __launch_bounds__(256) void glx_dist_ragged_stitch_kernel(const int64_t* __restrict__ order,
                                                                     const int64_t* __restrict__ src_off,
                                                                     const int64_t* __restrict__ deg_b,
                                                                     const int64_t* __restrict__ dst_off, int64_t n,
                                                                     const int64_t* __restrict__ nbr_in,
                                                                     const int64_t* __restrict__ eid_in,
                                                                     int64_t* __restrict__ nbr_out,
                                                                     int64_t* __restrict__ eid_out) {
  const int64_t pos = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (pos >= n) return;
  const int64_t s = src_off[pos], d = deg_b[pos], o = dst_off[order[pos]];
  for (int64_t j = lane; j < d; j += 64) {
    nbr_out[o + j] = nbr_in[s + j];
    eid_out[o + j] = eid_in[s + j];
  }
}
__global__ void glx_dist_widen_i32_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void glx_dist_stitch_deg_kernel(const int64_t* __restrict__ deg_b, const int64_t* __restrict__ order, int64_t n,
                                           int32_t* __restrict__ deg_out, int64_t* __restrict__ deg64_out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  deg_out[order[i]] = (int32_t)deg_b[i];
  deg64_out[order[i]] = deg_b[i];
}

__global__ void glx_dist_walk_column_kernel(const int64_t* __restrict__ step, int64_t n, int32_t walk_len, int32_t t,
                                            int64_t* __restrict__ walks) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) walks[i * walk_len + t] = step[i];
}

// node2vec across shards (glx_dist_random_walk): the owner's half -- slot x of request row r's answer carries, in place of
// its edge id, the weight of that edge (float bits; default_weight on an unweighted edge type).  One wave per row.
__global__ __launch_bounds__(256) void glx_dist_slot_weights_kernel(GlxIdMap map, const int64_t* __restrict__ row_ptr,
                                                                    const float* __restrict__ weight,
                                                                    const int64_t* __restrict__ ids,
                                                                    const int64_t* __restrict__ off, int64_t m,
                                                                    float default_weight, int64_t* __restrict__ out) {
  const int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= m) return;
  const int64_t o = off[r], d = off[r + 1] - o;
  if (d == 0) return;
  const int64_t row = glx_row_of(map, ids[r]);
  const int64_t s = row >= 0 ? row_ptr[row] : 0;
  for (int64_t x = lane; x < d; x += 64) {
    const float w = (weight && row >= 0) ? weight[s + x] : default_weight;
    out[o + x] = (int64_t)__float_as_uint(w);
  }
}

// ... and the requester's half: WeightedRandomWalkKernel (random_walk.cc:192-272) on the lists the owners sent -- the
// current vertex's first min(deg, F) neighbours with their weights, the parent's first min(deg, F) neighbours (the
// previous step's list) -- same arithmetic, same draw (stream (seed, cc + t, walker), draw 0) as glx_node2vec_step_kernel.
// One wave per walker; LDS: weights [F] f32 | table [F] 8 B | stack [F] 8 B.
__global__ __launch_bounds__(64) void glx_dist_node2vec_step_kernel(
    const int64_t* __restrict__ parent, int32_t t, const int32_t* __restrict__ deg_c, const int64_t* __restrict__ off_c,
    const int64_t* __restrict__ nbr_c, const int64_t* __restrict__ wbits_c, const int32_t* __restrict__ deg_p,
    const int64_t* __restrict__ off_p, const int64_t* __restrict__ nbr_p, float p, float q, int32_t F, uint64_t seed,
    uint64_t cc, int64_t default_nbr, int64_t* __restrict__ next) {
  // off_p: where the reference's cursor stands in the concatenated parent lists when it reaches walker i -- it is not
  // advanced for walkers whose current vertex has no out-edges (random_walk.cc:214-226), so behind a stuck walker the
  // windows start too early; glx_dist_walk_cursor_kernel computes it (== the true offsets while nobody is stuck)
  extern __shared__ int64_t lds64[];
  GlxAlias* tab = reinterpret_cast<GlxAlias*>(lds64);
  GlxAlias* stk = tab + F;
  float* dist = reinterpret_cast<float*>(stk + F);
  const int lane = threadIdx.x;
  const int64_t i = blockIdx.x;
  const int32_t n = deg_c[i];
  if (n == 0) {
    if (lane == 0) next[i] = default_nbr;
    return;
  }
  const int64_t oc = off_c[i];
  const int64_t par = parent[i];
  const int32_t pn = (t > 0 && deg_p) ? deg_p[i] : 0;
  const int64_t op = pn > 0 ? off_p[i] : 0;
  for (int32_t x = lane; x < n; x += 64) {
    const int64_t nbr = nbr_c[oc + x];
    const float w = __uint_as_float((uint32_t)wbits_c[oc + x]);
    float biased;
    if (nbr == par) {
      biased = (float)((double)w * 1.0 / ((double)p + 1e-6));
    } else {
      bool shared = false;
      for (int32_t y = 0; y < pn; ++y) shared |= nbr_p[op + y] == nbr;
      biased = shared ? w : (float)((double)w * 1.0 / ((double)q + 1e-6));
    }
    dist[x] = biased;
  }
  __syncthreads();
  if (lane == 0) {
    glx_alias_build_row_dev(dist, n, tab, stk);
    const int32_t pick = glx_alias_pick(glx_draw64(seed, cc + (uint64_t)t, (uint32_t)i, 0u), n, tab);
    next[i] = nbr_c[oc + pick];
  }
}

// off_used[i] = sum over j < i of (deg_c[j] > 0 ? deg_p[j] : 0): the reference's cursor (see the step kernel).  One
// workgroup: walk batches are small.
__global__ __launch_bounds__(1024) void glx_dist_walk_cursor_kernel(const int32_t* __restrict__ deg_c,
                                                                   const int32_t* __restrict__ deg_p, int64_t n,
                                                                   int64_t* __restrict__ off_used) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024, lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
  int64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += deg_c[i] > 0 ? deg_p[i] : 0;
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int64_t run = 0;
    for (int k = 0; k < 1024; ++k) {
      const int64_t v = part[k];
      part[k] = run;
      run += v;
    }
  }
  __syncthreads();
  sum = part[t];
  for (int64_t i = lo; i < hi; ++i) {
    off_used[i] = sum;
    sum += deg_c[i] > 0 ? deg_p[i] : 0;
  }
}

inline unsigned grid_for(int64_t n, int64_t cap = 4096) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b < cap ? b : cap);
}

inline uint64_t pow2_at_least(uint64_t x) {
  uint64_t c = 1;
  while (c < x) c <<= 1;
  return c;
}

#define GLX_ROCPRIM(call)                                             \
  do {                                                                \
    size_t bytes__ = 0;                                               \
    GLX_HIP(call(nullptr, bytes__));                                  \
    GlxTemp tmp__;                                                    \
    GLX_HIP(hipMalloc(&tmp__.p, bytes__ ? bytes__ : 16));             \
    GLX_HIP(call(tmp__.p, bytes__));                                  \
    GLX_HIP(hipStreamSynchronize(s)); /* tmp__ is freed right after */ \
  } while (0)

}  // namespace

// ---- speculation ledger (glx.h, ABI 4) ----
constexpr int kLedgerClasses = 8;
constexpr int kLedgerTail = 3 + kLedgerClasses;  // words appended to every count exchange of an attached store
// device words: [0] epoch + 1 of the newest call one of whose buckets did not fit its message (an abort starts a new epoch:
// calls of an aborted epoch that are still in flight raise flags nobody heeds)  [1 .. 8] per request shape: the largest
// per-owner share of a request seen since the last exchange, as (rows << 20) / request length, rounded up
struct glx_dist_ledger {
  int device = 0;
  int64_t* d_words = nullptr;  // [1 + kLedgerClasses]
  int64_t* d_stage = nullptr;  // [kMaxWorld + 32 + kLedgerTail]: a count exchange's values + the tail
  // Speculation is keyed by POSITION: the i-th glx_dist_sample call since the last confirmation point (a count exchange
  // that is not a sampling call's own: the aggregation's, glx_dist_confirm, ...).  Whether call i speculates depends
  // only on (i, what the ranks learned together at position i, hold) -- state every rank holds identically as long as
  // the ranks issue the same SEQUENCE of calls -- never on this rank's request length: a rank with a short tail batch,
  // or an idle rank with an empty request, still enters the same collectives as its peers (a decision
  // taken from the local length let one rank take the fixed-capacity exchange while another took the count exchange).
  struct Shape {
    int64_t rows = 0;  // the largest bucket any rank sent to any owner at this position so far; 0 = not learned
    int64_t n = 0;     // the longest request seen at this position (informational: largest_share)
  };
  Shape shapes[kLedgerClasses];
  int pos = 0;  // index of the next sampling call in the current window
  bool hold = false;
  int64_t epoch = 0;
  // speculated calls since the last exchange: how many, and a digest of their parameters (compared across ranks)
  int64_t pending = 0;
  uint64_t digest = 0;
  double slack = 1.25;
  int64_t pad_rows = 1024;
  glx_dist_ledger_stats stats;
  std::vector<int64_t> h_tmp;
  bool learned(int c) const { return c >= 0 && c < kLedgerClasses && shapes[c].rows > 0; }
  int64_t capacity(int c) const {
    int64_t cap = (int64_t)((double)shapes[c].rows * slack) + pad_rows;
    cap = (cap + 63) & ~(int64_t)63;
    return cap < 64 ? 64 : cap;
  }
  void note_share(int c) {
    if (shapes[c].n > 0) {
      const double sh = (double)shapes[c].rows / (double)shapes[c].n;
      if (sh > stats.largest_share) stats.largest_share = sh < 1.0 ? sh : 1.0;
    }
  }
};

namespace {

// vals[nvals] + the ledger's words (taken: a flag raised after this kernel ran travels with the next exchange).
__global__ void glx_dist_ledger_pack_kernel(const int64_t* __restrict__ vals, int nvals, int64_t* __restrict__ words,
                                            int64_t pending, int64_t digest, int64_t* __restrict__ out) {
  const int t = threadIdx.x;
  for (int i = t; i < nvals; i += blockDim.x) out[i] = vals[i];
  if (t == 0) {
    out[nvals + 1] = pending;
    out[nvals + 2] = digest;
  }
  if (t <= kLedgerClasses) {
    const int64_t w = (int64_t)atomicExch(reinterpret_cast<unsigned long long*>(words + t), 0ull);
    out[nvals + (t == 0 ? 0 : 2 + t)] = w;
  }
}

// Fixed-capacity messages of a speculated request: bucket p's first cap rows, padded with an id no shard knows.
// blockIdx.y = owner.  cnt_off[0 .. P) <- the bucket sizes, [P .. 2P) <- their offsets in the bucketed request (the
// stitch reads them after the store's shared counters have moved on).
__global__ __launch_bounds__(256) void glx_dist_spec_pack_kernel(const int64_t* __restrict__ bucketed,
                                                                 const int64_t* __restrict__ order,
                                                                 const int64_t* __restrict__ counts, int32_t P,
                                                                 int64_t cap, int64_t n, int64_t* __restrict__ ids_out,
                                                                 int64_t* __restrict__ rows_out,
                                                                 int64_t* __restrict__ cnt_off,
                                                                 int64_t* __restrict__ words, int32_t shape,
                                                                 int64_t tag) {
  const int32_t p = blockIdx.y;
  int64_t off = 0;
  for (int32_t q = 0; q < p; ++q) off += counts[q];
  const int64_t c = counts[p];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cnt_off[p] = c;
    cnt_off[P + p] = off;
    if (c > cap) atomicMax(reinterpret_cast<unsigned long long*>(words), (unsigned long long)tag);
    atomicMax(reinterpret_cast<unsigned long long*>(words + 1 + shape), (unsigned long long)c);  // rows this bucket needed
  }
  const int64_t take = c < cap ? c : cap;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    ids_out[p * cap + i] = i < take ? bucketed[off + i] : GLX_EMPTY_KEY;
    rows_out[p * cap + i] = i < take ? order[off + i] : 0;
  }
}

// Stitcher::DoStitch over the fixed-capacity answers: slot i of owner p's answer is request row order[off_p + i].
__global__ __launch_bounds__(256) void glx_dist_spec_stitch_kernel(const int64_t* __restrict__ nbr_in,
                                                                   const int64_t* __restrict__ eid_in,
                                                                   const int64_t* __restrict__ order,
                                                                   const int64_t* __restrict__ cnt_off, int32_t P,
                                                                   int64_t cap, int32_t k, int64_t* __restrict__ nbr_out,
                                                                   int64_t* __restrict__ eid_out) {
  const int32_t p = blockIdx.y;
  const int64_t c = cnt_off[p], off = cnt_off[P + p];
  const int64_t total = (c < cap ? c : cap) * k;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / k;
    const int32_t col = (int32_t)(t - i * k);
    const int64_t o = order[off + i] * k + col;
    nbr_out[o] = nbr_in[p * cap * k + t];
    eid_out[o] = eid_in[p * cap * k + t];
  }
}

// The live keys of an id hash table: their minimum / maximum, and one bit per key (glx_dist_store_set_graph_replica).
__global__ __launch_bounds__(256) void glx_dist_keys_minmax_kernel(const int64_t* __restrict__ keys, int64_t cap,
                                                                   long long* __restrict__ mm) {
  long long lo = INT64_MAX, hi = INT64_MIN;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = keys[i];
    if (k != GLX_EMPTY_KEY) {
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const long long l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0 && lo <= hi) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

__global__ __launch_bounds__(256) void glx_dist_keys_setbits_kernel(const int64_t* __restrict__ keys, int64_t cap,
                                                                    unsigned long long* __restrict__ bits) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = keys[i];
    if (k != GLX_EMPTY_KEY) atomicOr(&bits[k >> 6], 1ull << (k & 63));
  }
}

}  // namespace

struct glx_dist_store {
  glx_dist_ledger* ledger = nullptr;
  glx_comm* comm = nullptr;
  const glx_graph* graph = nullptr;
  const glx_graph* graph_replica = nullptr;  // complete rows of the hot vertices, borrowed (glx_dist_store_set_graph_replica)
  uint64_t* rg_bits = nullptr;  // the replica's vertex ids as a bitmap over [0, rg_bits_max] (owned), or nullptr: the
  int64_t rg_bits_max = -1;     // request partition tests membership there instead of probing the replica's hash map
  int64_t sample_rows = 0, sample_rows_replica = 0, sample_rows_remote = 0;  // the last glx_dist_sample
  const glx_features* feats = nullptr;
  glx_features* cache = nullptr;
  PackedSlot* cache_slots = nullptr;  // the replica's id map, packed (glx_dist_pack_map_kernel)
  RankWord* bm_member = nullptr;      // ... or, for small non-negative ids, a bitmap with per-word ranks
  uint64_t* bm_valid = nullptr;       // (same allocation) the known ids, only when some hot id is unknown to its owner
  int64_t bm_max = -1;
  int device = 0, rank = 0, world = 1;
  // the destination ids this rank OWNS (llabs(id) % P == rank), ascending, with their in-degree summed over all shards:
  // built collectively on first use (glx_dist_in_degrees, glx_dist_negative_create); -1 = not built
  int64_t* own_id = nullptr;
  int64_t* own_cnt = nullptr;
  int64_t own_M = -1;
  bool shortcut = true;  // world == 1: call the local operator directly
  Arena req, recv;  // sampling: request-sized and receive-sized buffers
  Arena r_req, r_recv, r_back;  // design R (glx_dist_aggregate_partial): request-, receive- and partial-sized buffers
  // The last kernels of a sampling call (the stitch into the caller's response) still READ these arenas when the call
  // returns; the next call re-carves them at once.  On the same stream that is ordered; a caller that alternates
  // streams is ordered through this event (recorded when a call has enqueued its last kernel, waited on by the next
  // call's stream before it touches the arenas).
  hipEvent_t arena_free = nullptr;
  hipStream_t arena_stream = nullptr;
  bool arena_used = false;
  // aggregation / lookup: one buffer set per request in flight (glx_dist_aggregate_begin .. _end)
  struct Slot {
    Arena req, recv, tab, halo;
    const int32_t* loc = nullptr;
    GlxRowSource src[3];
    int32_t num_ids = -1;  // -1: nothing begun
    float default_attr = 0.0f;
    glx_dist_stats stats;
  };
  Slot slots[GLX_DIST_SLOTS];
  int last_slot = 0;
  int64_t* d_vals = nullptr;  // [world + 8] values shared by the count exchange
  int32_t* d_ctr = nullptr;   // [3 * world + 8] counter block of the resolve passes
  double halo_share = 0.0;  // largest (distinct halo ids / request ids) seen so far
  bool halo_share_known = false;  // ... by at least one request with ids
  // every count exchange blocks the calling host thread until the slowest rank's counts have arrived: how often, and
  // for how long, since the store was created (glx_dist_stats.host_syncs / host_stall_us)
  int64_t host_syncs = 0, host_stall_us = 0;
  glx_dist_stats stats;
  std::vector<int64_t> h_mat;
};

namespace {

struct Routing {
  std::vector<int64_t> send_counts, send_offs, recv_counts, recv_offs;
  int64_t n_send = 0, n_recv = 0;
};

// h_mat[q * nvals + p] (p < P) = elements rank q sends to rank p.
void routing_from_matrix(const glx_dist_store* st, int nvals, Routing* r) {
  const int P = st->world, me = st->rank;
  r->send_counts.assign(P, 0);
  r->recv_counts.assign(P, 0);
  r->send_offs.assign(P + 1, 0);
  r->recv_offs.assign(P + 1, 0);
  for (int p = 0; p < P; ++p) {
    r->send_counts[p] = st->h_mat[(size_t)me * nvals + p];
    r->recv_counts[p] = st->h_mat[(size_t)p * nvals + me];
    r->send_offs[p + 1] = r->send_offs[p] + r->send_counts[p];
    r->recv_offs[p + 1] = r->recv_offs[p] + r->recv_counts[p];
  }
  r->n_send = r->send_offs[P];
  r->n_recv = r->recv_offs[P];
}

// The count exchange of a partitioned request: the one place its host thread waits for the other ranks.  With a
// ledger the exchange is also the confirmation point of the calls that skipped theirs: GLX_ABORTED on every rank when
// any rank's speculated message overflowed (or the ranks speculated on different requests).
int exchange_counts(glx_dist_store* st, const int64_t* d_vals, int nvals, int64_t* h_out, hipStream_t s,
                    bool window_end = true) {
  glx_dist_ledger* lg = st->ledger;
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  if (lg == nullptr) {
    rc = st->comm->allgather_i64(d_vals, nvals, h_out, s);
  } else {
    const int P = st->world, wide = nvals + kLedgerTail;
    GLX_REQUIRE(nvals <= kMaxWorld + 32, "count exchange of %d values with a ledger", nvals);
    glx_dist_ledger_pack_kernel<<<1, 64, 0, s>>>(d_vals, nvals, lg->d_words, lg->pending, (int64_t)lg->digest, lg->d_stage);
    lg->h_tmp.resize((size_t)P * wide);
    rc = st->comm->allgather_i64(lg->d_stage, wide, lg->h_tmp.data(), s);
    if (rc == GLX_OK) {
      bool overflow = false, disagree = false;
      int64_t need[kLedgerClasses] = {0};
      for (int q = 0; q < P; ++q) {
        const int64_t* row = &lg->h_tmp[(size_t)q * wide];
        memcpy(h_out + (size_t)q * nvals, row, (size_t)nvals * 8);
        overflow = overflow || row[nvals] == lg->epoch + 1;
        disagree = disagree || row[nvals + 1] != lg->pending || row[nvals + 2] != (int64_t)lg->digest;
        for (int c = 0; c < kLedgerClasses; ++c) need[c] = need[c] > row[nvals + 3 + c] ? need[c] : row[nvals + 3 + c];
      }
      for (int c = 0; c < kLedgerClasses; ++c) {
        if (lg->shapes[c].rows > 0 && need[c] > lg->shapes[c].rows) lg->shapes[c].rows = need[c];  // what would have fitted
        lg->note_share(c);
      }
      lg->pending = 0;
      lg->digest = 0;
      // a confirmation point that is not a sampling call's own exchange starts the next window of positions; so does
      // every abort (the caller repeats the window's calls)
      if (window_end || overflow || disagree) lg->pos = 0;
      if (disagree) {
        lg->hold = true;
        lg->stats.holding = 1;
      }
      if (overflow || disagree) {
        ++lg->stats.aborted;
        ++lg->epoch;
        glx_set_error(disagree ? "speculated requests differ between the ranks (neighbor_count, sampler, padding, "
                                 "seed or call_counter): their results are void, and this ledger no longer speculates"
                               : "a speculated request did not fit its fixed-capacity messages: the results of the "
                                 "glx_dist_sample calls since the last count exchange are void -- repeat them (the "
                                 "capacities have been raised)");
        rc = GLX_ABORTED;
      }
    }
  }
  st->host_stall_us += (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  ++st->host_syncs;
  return rc;
}

struct ArenaOrder {
  glx_dist_store* st;
  hipStream_t s;
  ArenaOrder(glx_dist_store* st_, hipStream_t s_) : st(st_), s(s_) {
    if (st->arena_used && st->arena_free && st->arena_stream != s) (void)hipStreamWaitEvent(s, st->arena_free, 0);
  }
  ~ArenaOrder() {
    if (!st->arena_free && hipEventCreateWithFlags(&st->arena_free, hipEventDisableTiming) != hipSuccess) {
      st->arena_free = nullptr;
      (void)hipStreamSynchronize(s);  // no event to order the next call by: drain instead
      return;
    }
    (void)hipEventRecord(st->arena_free, s);
    st->arena_stream = s;
    st->arena_used = true;
  }
};

// Result of routing a request's ids to their rows: loc[n] virtual rows + the row sources.
struct Resolved {
  const int32_t* loc = nullptr;
  GlxRowSource src[3];
};

// Shared front half of glx_dist_aggregate / glx_dist_lookup: resolve ids, dedup the cold
// remote ones, fetch their rows from the owners.  d_ids is a device pointer.
int resolve_and_fetch(glx_dist_store* st, int slot, const int64_t* d_ids, int64_t n, float default_attr, hipStream_t s,
                      Resolved* out) {
  glx_dist_store::Slot& sl = st->slots[slot];
  st->last_slot = slot;
  const glx_features* f = st->feats;
  const int P = st->world, me = st->rank;
  const int32_t dim = f->dim;
  const int64_t n_own = f->num_rows, n_cache = st->cache ? st->cache->num_rows : 0;
  GLX_REQUIRE(n_own + n_cache + n < (int64_t)INT32_MAX, "virtual row space exceeds int32");
  GLX_REQUIRE(n < ((int64_t)1 << 29), "a partitioned request is limited to 2^29 ids");
  glx_dist_stats& stat = sl.stats;
  memset(&stat, 0, sizeof(stat));
  stat.ids = n;

  const int nvals = P + 6;  // + the requester's default_attr (what its unknown ids look like)
  // The resolve launch.  Few, long-lived workgroups -- 1024, four per CU: every wave ends with a partial batch of its
  // queue (a full chain of dependent round trips for a few ids), and more resident waves contend for the same gather
  // path (P = 8, 18 M ids, round 6's kernel: 192 us at 768 workgroups x 4 ids, 199 at 1024 x 2, 205 at 1280, 236 at
  // 1536, 290 at 2048: profiles/r06/resolve_probe_history.txt D).  Two ids per thread-iteration with the rank
  // records, one with the hash map (scripts/resolve_probe.py).  With a replica the ids it does not hold are a small
  // share of the request: queued per wave and resolved 64 at a time (kQueue); without one every id takes that path and
  // a queue would only add work.
  static const bool kNoQueue = getenv("GLX_RESOLVE_NO_QUEUE") != nullptr;  // (ablation)
  const bool has_cache = st->cache != nullptr;
  const bool ranked = has_cache && st->bm_member != nullptr;
  const bool queue = has_cache && !kNoQueue;
  int per = ranked ? 2 : 1;  // ids per thread per pass
  int64_t nb = 1024;
  if (queue) {
    const int64_t kper = glx_side_knobs().resolve_ids.load(std::memory_order_relaxed);  // (A/B)
    const int64_t kb = glx_side_knobs().resolve_blocks.load(std::memory_order_relaxed);
    if (ranked && (kper == 8 || kper == 4)) per = (int)kper;
    if (kb > 0) nb = kb;
  }
  const unsigned rgrid = grid_for((n + per - 1) / per, nb);
  // ids one workgroup resolves at most = its cold list's capacity (kQueue)
  const int64_t tile = 256ll * per;
  const int64_t region = ((n + (int64_t)rgrid * tile - 1) / ((int64_t)rgrid * tile)) * tile;
  // request-sized buffers: loc[n] (int32) + cold_ids[n] (int64; at most n distinct) + the cold lists
  Carver cv;
  const size_t o_loc = cv.take((size_t)(n > 0 ? n : 1) * 4);
  const size_t o_cold = cv.take((size_t)(n > 0 ? n : 1) * 8);
  const size_t o_list = cv.take(queue && P > 1 ? (size_t)rgrid * (size_t)region * 4 : 0);
  const size_t o_lcnt = cv.take(queue && P > 1 ? (size_t)rgrid * 4 : 0);
  const size_t o_part = cv.take((size_t)rgrid * (3 + P) * 4);
  int rc = sl.req.ensure(cv.at);
  if (rc != GLX_OK) return rc;
  int32_t* loc = reinterpret_cast<int32_t*>(sl.req.p + o_loc);
  int64_t* cold_ids = reinterpret_cast<int64_t*>(sl.req.p + o_cold);
  int32_t* cold_idx = queue && P > 1 ? reinterpret_cast<int32_t*>(sl.req.p + o_list) : nullptr;
  int32_t* cold_cnt = queue && P > 1 ? reinterpret_cast<int32_t*>(sl.req.p + o_lcnt) : nullptr;

  // Set of distinct halo ids.  Sized for the request at hand: 2.5x the largest share of distinct halo ids this store
  // has seen (a hop-2 and a hop-1 request alternate: sizing from the PREVIOUS request's absolute count made every
  // other call overflow) -- a quarter of the request's ids while nothing has been seen yet (round 6: that quarter
  // used to be the floor for good: 8 M slots to clear and to compact for the 0.25 M distinct halo ids of the
  // headline's 18 M-id request, 0.15 ms of a rank-step) -- and regrown to the safe bound 2n -- on the ranks that
  // overflowed, in lockstep with the others -- when that is not enough.
  const uint64_t safe_cap = pow2_at_least((uint64_t)(n > 0 ? n : 1) * 2);
  double share = st->halo_share_known ? 2.5 * st->halo_share : 0.25;
  const int64_t floor_1024 = glx_side_knobs().resolve_set_share.load(std::memory_order_relaxed);
  const double floor_share = floor_1024 > 0 ? (double)floor_1024 / 1024.0 : 1.0 / 64;
  if (share < floor_share) share = floor_share;
  uint64_t tcap = P == 1 ? 64 : pow2_at_least((uint64_t)((double)(n > 0 ? n : 1) * share) + 65536);
  if (tcap > safe_cap) tcap = safe_cap;
  bool first = true, mine_overflow = false;
  int64_t* tkeys = nullptr;
  int32_t* tvals = nullptr;
  while (true) {
    if (first || mine_overflow) {
      if (mine_overflow) tcap = safe_cap;
      rc = sl.tab.ensure((size_t)tcap * 12 + 256);
      if (rc != GLX_OK) return rc;
      tkeys = reinterpret_cast<int64_t*>(sl.tab.p);
      tvals = reinterpret_cast<int32_t*>(sl.tab.p + (((size_t)tcap * 8 + 255) & ~(size_t)255));
      if (n > 0) {
        // one launch clears the set and the counter block; the resolve's last workgroup turns the counts into offsets
        // and the shared values (rounds 1-5: memset + fill + resolve + offsets + parameter kernel)
        glx_dist_fill_keys_kernel<<<grid_for((int64_t)tcap), 256, 0, s>>>(tkeys, tcap, st->d_ctr, 3 * P + 8);
        ResolveArgs a{};
        a.cache_map = PackedMap{st->cache_slots, st->cache ? st->cache->idmap.cap - 1 : 0};
        a.own_map = f->map();
        a.ids = d_ids;
        a.n = n;
        a.loc = loc;
        a.tkeys = tkeys;
        a.tmask = tcap - 1;
        a.ctr = st->d_ctr;
        a.P = P;
        a.me = me;
        a.cache_base = (int32_t)n_own;
        a.has_cache = has_cache;
        a.bm_member = has_cache ? st->bm_member : nullptr;
        a.bm_valid = st->bm_valid;
        a.bm_max = st->bm_max;
        a.insert_limit = tcap >= safe_cap ? INT32_MAX : (int32_t)(tcap / 10 * 6);
        // at the safe size (>= 2 slots per id of the request) the load stays <= 50 %: every probe sequence ends, and a
        // capped one could only fail this rank AFTER its peers passed the count exchange -- leaving them in the
        // row exchange waiting for it
        a.max_probe = tcap >= safe_cap ? INT32_MAX : kMaxProbe;
        a.cold_idx = cold_idx;
        a.cold_cnt = cold_cnt;
        a.region = region;
        a.vals = st->d_vals;
        a.default_attr = default_attr;
        a.part = reinterpret_cast<int32_t*>(sl.req.p + o_part);
        a.peek = glx_side_knobs().resolve_peek.load(std::memory_order_relaxed) != 0;
        // own ids before the replica: only where it pays -- at world size 1 every lookup goes away (0.145 -> 0.116 ms
        // for the headline's 18 M ids); with more ranks the test costs every id a remainder and saves 1 / P of the
        // lookups: level at P = 2, +4 % at P = 4 and 8 (same-process A/B, profiles/r06/resolve_set_probe.txt).
        // GLX_RESOLVE_OWN_FIRST = 1 / 0 forces it on / off.
        const int64_t own_knob = glx_side_knobs().resolve_own_first.load(std::memory_order_relaxed);
        a.own_first = own_knob < 0 ? P == 1 : own_knob != 0;
        if (queue) {
          if (ranked && per == 8) glx_dist_resolve_kernel<8, true><<<rgrid, 256, 0, s>>>(a);
          else if (ranked && per == 4) glx_dist_resolve_kernel<4, true><<<rgrid, 256, 0, s>>>(a);
          else if (ranked) glx_dist_resolve_kernel<2, true><<<rgrid, 256, 0, s>>>(a);
          else glx_dist_resolve_kernel<1, true><<<rgrid, 256, 0, s>>>(a);
        } else if (ranked) {
          glx_dist_resolve_kernel<2, false><<<rgrid, 256, 0, s>>>(a);
        } else {
          glx_dist_resolve_kernel<1, false><<<rgrid, 256, 0, s>>>(a);
        }
        if (P > 1) {
          glx_dist_assign_kernel<<<grid_for((int64_t)(tcap / kAssignPer + 1), 2048), 256, 0, s>>>(tkeys, tcap, tvals, st->d_ctr, P, cold_ids);
          if (cold_idx) {
            glx_dist_finalize_list_kernel<<<rgrid, 256, 0, s>>>(loc, cold_idx, cold_cnt, region, tvals,
                                                                (int32_t)(n_own + n_cache));
          } else {
            glx_dist_finalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(loc, n, tvals, (int32_t)(n_own + n_cache));
          }
        }
      } else {
        glx_dist_offsets_kernel<<<1, 64, 0, s>>>(st->d_ctr, P, st->d_vals, default_attr);  // an empty request: one launch
      }
      GLX_HIP(hipGetLastError());
    }
    first = false;
    st->h_mat.resize((size_t)P * nvals);
    rc = exchange_counts(st, st->d_vals, nvals, st->h_mat.data(), s);  // the one host sync
    if (rc != GLX_OK) return rc;
    bool any = false;
    for (int q = 0; q < P; ++q) any = any || st->h_mat[(size_t)q * nvals + P] != 0;
    mine_overflow = st->h_mat[(size_t)me * nvals + P] != 0;
    if (!any) break;
    GLX_REQUIRE(!(mine_overflow && tcap >= safe_cap), "halo id set overflowed at its safe size (internal error)");
  }
  const int64_t* mine = &st->h_mat[(size_t)me * nvals];
  const int64_t U = mine[P + 1];
  if (n > 0 && (double)U / (double)n > st->halo_share) st->halo_share = (double)U / (double)n;
  if (n > 0) st->halo_share_known = true;
  stat.remote_distinct = U;
  stat.from_replica = mine[P + 2];
  stat.from_own_shard = mine[P + 3];
  stat.remote = mine[P + 4];

  Routing rt;
  routing_from_matrix(st, nvals, &rt);
  const int64_t m = rt.n_recv;  // rows this rank serves to its peers
  stat.served_rows = m;
  float* halo = nullptr;
  if (P > 1) {
    // receive-sized: ids_in[m] + rows_loc[m, dim]; halo[U, dim] lives in its own arena
    Carver cr;
    const size_t o_ids = cr.take((size_t)(m > 0 ? m : 1) * 8);
    const size_t o_rows = cr.take((size_t)(m > 0 ? m : 1) * dim * 4);
    rc = sl.recv.ensure(cr.at);
    if (rc == GLX_OK) rc = sl.halo.ensure((size_t)(U > 0 ? U : 1) * dim * 4);
    if (rc != GLX_OK) return rc;
    int64_t* ids_in = reinterpret_cast<int64_t*>(sl.recv.p + o_ids);
    float* rows_loc = reinterpret_cast<float*>(sl.recv.p + o_rows);
    halo = reinterpret_cast<float*>(sl.halo.p);
    GlxSeg seg_ids{cold_ids, ids_in, 8};
    rc = st->comm->alltoallv(&seg_ids, 1, rt.send_counts.data(), rt.send_offs.data(), rt.recv_counts.data(),
                             rt.recv_offs.data(), s);
    if (rc != GLX_OK) return rc;
    // gather for every requester with ITS default row for unknown ids
    bool same_default = true;
    for (int q = 0; q < P; ++q) same_default = same_default && st->h_mat[(size_t)q * nvals + P + 5] == mine[P + 5];
    for (int q = 0; q < (same_default ? 1 : P); ++q) {
      const int64_t begin = same_default ? 0 : rt.recv_offs[q];
      const int64_t cnt = same_default ? m : rt.recv_counts[q];
      float dq = default_attr;
      if (!same_default) memcpy(&dq, &st->h_mat[(size_t)q * nvals + P + 5], sizeof(float));
      if (cnt > 0) {
        rc = glx_lookup(f, ids_in + begin, cnt, dq, rows_loc + begin * dim, GLX_PTR_DEVICE, s);
        if (rc != GLX_OK) return rc;
      }
    }
    GlxSeg seg_rows{rows_loc, halo, (size_t)dim * 4};
    rc = st->comm->alltoallv(&seg_rows, 1, rt.recv_counts.data(), rt.recv_offs.data(), rt.send_counts.data(),
                             rt.send_offs.data(), s);
    if (rc != GLX_OK) return rc;
    stat.exchange_rounds = st->comm->last_rounds;
    const int64_t self_ids = rt.send_counts[me];  // always 0: own ids never enter the halo set
    stat.bytes_sent = (rt.n_send - self_ids) * 8 + (m - rt.recv_counts[me]) * (int64_t)dim * 4;
    stat.bytes_received = (m - rt.recv_counts[me]) * 8 + (rt.n_send - self_ids) * (int64_t)dim * 4;
  }
  out->loc = loc;
  out->src[0] = GlxRowSource{f->X, f->stride, f->swizzle_rows, n_own};
  out->src[1] = st->cache ? GlxRowSource{st->cache->X, st->cache->stride, st->cache->swizzle_rows, n_cache}
                          : GlxRowSource{nullptr, dim, 0, 0};
  out->src[2] = GlxRowSource{U > 0 ? halo : nullptr, dim, 0, U};
  return GLX_OK;
}

// A request whose shape the ledger knows: fixed-capacity messages, no count leaves the device (glx.h, ABI 4).
// bucketed / order / counts: the partitioned request, all in the store's request arena -- NOT in st->d_vals: this call
// returns without a host wait, and a call on another stream (an aggregation's resolve on a store that holds both a graph
// and features) rewrites st->d_vals at once; the arena is ordered between sampling calls by ArenaOrder.
int dist_sample_speculated(glx_dist_store* st, glx_dist_ledger* lg, int shape, int sampler, int64_t n, int32_t k,
                           int padding_mode, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                           const glx_graph* rg, const int64_t* bucketed, const int64_t* order, const int64_t* counts,
                           int64_t* nbr_out, int64_t* eid_out, hipStream_t s) {
  const int P = st->world;
  const int64_t cap = lg->capacity(shape), m = cap * P;
  GLX_REQUIRE(m * k <= (int64_t)INT32_MAX * 8, "speculated request too large");
  Carver cr;
  const size_t o_sid = cr.take((size_t)m * 8);
  const size_t o_srow = cr.take((size_t)m * 8);
  const size_t o_ids = cr.take((size_t)m * 8);
  const size_t o_rows = cr.take((size_t)m * 8);
  const size_t o_nl = cr.take((size_t)m * k * 8);
  const size_t o_el = cr.take((size_t)m * k * 8);
  const size_t o_nb = cr.take((size_t)m * k * 8);
  const size_t o_eb = cr.take((size_t)m * k * 8);
  const size_t o_co = cr.take((size_t)2 * P * 8);
  int rc = st->recv.ensure(cr.at);
  if (rc != GLX_OK) return rc;
  char* base = st->recv.p;
  int64_t* send_ids = reinterpret_cast<int64_t*>(base + o_sid);
  int64_t* send_rows = reinterpret_cast<int64_t*>(base + o_srow);
  int64_t* ids_in = reinterpret_cast<int64_t*>(base + o_ids);
  int64_t* rows_in = reinterpret_cast<int64_t*>(base + o_rows);
  int64_t* nbr_loc = reinterpret_cast<int64_t*>(base + o_nl);
  int64_t* eid_loc = reinterpret_cast<int64_t*>(base + o_el);
  int64_t* nbr_back = reinterpret_cast<int64_t*>(base + o_nb);
  int64_t* eid_back = reinterpret_cast<int64_t*>(base + o_eb);
  int64_t* cnt_off = reinterpret_cast<int64_t*>(base + o_co);

  const unsigned gx = (unsigned)((cap + 255) / 256 < 1024 ? (cap + 255) / 256 : 1024);
  glx_dist_spec_pack_kernel<<<dim3(gx > 0 ? gx : 1, (unsigned)P), 256, 0, s>>>(bucketed, order, counts, P, cap, n, send_ids,
                                                                              send_rows, cnt_off, lg->d_words, shape,
                                                                              lg->epoch + 1);
  GLX_HIP(hipGetLastError());
  if (rg != nullptr) {
    // the replica serves its bucket straight into the caller's response.  Its size stays on the device, so the launch
    // covers the whole request: rows of the other buckets are ids the replica does not know -- it writes their
    // default answer, and the stitch below overwrites it with the owner's
    const int64_t chunk_r = (int64_t)INT32_MAX / k;
    for (int64_t lo = 0; lo < n; lo += chunk_r) {
      const int64_t cnt = n - lo < chunk_r ? n - lo : chunk_r;
      rc = glx_sample_scatter_device(rg, sampler, bucketed + lo, order + lo, (int32_t)cnt, k, padding_mode,
                                     default_neighbor_id, seed, call_counter, nbr_out, eid_out, s);
      if (rc != GLX_OK) return rc;
    }
  }
  std::vector<int64_t> cnts((size_t)P, cap), offs((size_t)P + 1, 0);
  for (int p = 0; p < P; ++p) offs[p + 1] = offs[p] + cap;
  GlxSeg out_segs[2] = {{send_ids, ids_in, 8}, {send_rows, rows_in, 8}};
  rc = st->comm->alltoallv(out_segs, 2, cnts.data(), offs.data(), cnts.data(), offs.data(), s);
  if (rc != GLX_OK) return rc;
  const int64_t chunk = (int64_t)INT32_MAX / k;
  for (int64_t lo = 0; lo < m; lo += chunk) {
    const int64_t cnt = m - lo < chunk ? m - lo : chunk;
    rc = glx_sample_filtered(st->graph, sampler, ids_in + lo, rows_in + lo, (int32_t)cnt, k, padding_mode,
                             default_neighbor_id, seed, call_counter, nullptr, nbr_loc + lo * k, eid_loc + lo * k,
                             GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
  }
  GlxSeg back_segs[2] = {{nbr_loc, nbr_back, (size_t)k * 8}, {eid_loc, eid_back, (size_t)k * 8}};
  rc = st->comm->alltoallv(back_segs, 2, cnts.data(), offs.data(), cnts.data(), offs.data(), s);
  if (rc != GLX_OK) return rc;
  const int64_t per = cap * k;
  const unsigned sx = (unsigned)((per + 255) / 256 < 4096 ? (per + 255) / 256 : 4096);
  glx_dist_spec_stitch_kernel<<<dim3(sx > 0 ? sx : 1, (unsigned)P), 256, 0, s>>>(nbr_back, eid_back, order, cnt_off, P, cap, k,
                                                                                nbr_out, eid_out);
  GLX_HIP(hipGetLastError());
  // what the ranks must have agreed on for the owners' answers to be the requesters' (compared at the confirmation)
  const uint64_t words[6] = {seed, call_counter, (uint64_t)k, (uint64_t)sampler, (uint64_t)padding_mode,
                             (uint64_t)default_neighbor_id};  // not the length: ranks may differ in it (tail batches)
  uint64_t d = lg->digest;
  for (uint64_t w : words) d = glx_mix64(d ^ (w + 0x9e3779b97f4a7c15ull));
  lg->digest = d;
  ++lg->pending;
  ++lg->pos;
  ++lg->stats.speculated;
  st->sample_rows = n;
  st->sample_rows_replica = -1;  // sizes of a speculated request stay on the device
  st->sample_rows_remote = -1;
  return GLX_OK;
}

int dist_sample_device(glx_dist_store* st, int sampler, const int64_t* src, int32_t batch, int32_t k,
                       int padding_mode, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                       const glx_filter* filter, int64_t* nbr_out, int64_t* eid_out, hipStream_t s) {
  const int P = st->world;
  const bool filtered = filter != nullptr && filter->type != GLX_FILTER_NONE;
  if (P == 1 && st->shortcut) {
    return glx_sample_filtered(st->graph, sampler, src, nullptr, batch, k, padding_mode, default_neighbor_id, seed,
                               call_counter, filter, nbr_out, eid_out, GLX_PTR_DEVICE, s);
  }
  const int64_t n = batch;
  ArenaOrder arena_order(st, s);
  Carver cv;
  const size_t o_buck = cv.take((size_t)(n > 0 ? n : 1) * 8);
  const size_t o_ord = cv.take((size_t)(n > 0 ? n : 1) * 8);
  const size_t o_val = cv.take(filtered ? (size_t)(n > 0 ? n : 1) * 8 : 0);
  const size_t o_nb = cv.take((size_t)(n > 0 ? n : 1) * k * 8);
  const size_t o_eb = cv.take((size_t)(n > 0 ? n : 1) * k * 8);
  const size_t o_cnt = cv.take((size_t)(P + 16) * 8);  // bucket sizes + request parameters: this call's own (see above)
  int rc = st->req.ensure(cv.at);
  if (rc != GLX_OK) return rc;
  int64_t* bucketed = reinterpret_cast<int64_t*>(st->req.p + o_buck);
  int64_t* order = reinterpret_cast<int64_t*>(st->req.p + o_ord);
  int64_t* vals_b = filtered ? reinterpret_cast<int64_t*>(st->req.p + o_val) : nullptr;
  int64_t* nbr_back = reinterpret_cast<int64_t*>(st->req.p + o_nb);
  int64_t* eid_back = reinterpret_cast<int64_t*>(st->req.p + o_eb);
  int64_t* d_cnt = reinterpret_cast<int64_t*>(st->req.p + o_cnt);

  // Rows of vertices the graph replica holds are served here, from the same adjacency (and alias tables) their
  // owner holds and with the same random stream (their index in the request): they form one more bucket, last.
  const glx_graph* rg = st->graph_replica;
  const bool divert = rg != nullptr && !filtered && sampler != GLX_SAMPLER_IN_DEGREE &&
                      (sampler != GLX_SAMPLER_EDGE_WEIGHT || rg->weight != nullptr);
  constexpr int kParams = 11;
  glx_dist_ledger* lg = st->ledger;
  // position-keyed: the same answer on every rank that issued the same sequence of calls (see glx_dist_ledger)
  const int shape = lg && !lg->hold && !filtered && k > 0 && lg->learned(lg->pos) ? lg->pos : -1;
  // The request's scalar parameters ride behind its bucket sizes, written by the partition's own kernel (round 6: a
  // launch of their own cost as much as partitioning a hop-1 request; a speculated request ships none)
  GlxPartitionTail mine;
  memset(&mine, 0, sizeof(mine));
  if (shape < 0) {
    mine.n = kParams;
    mine.at = P;  // behind the P owners' sizes (the graph replica's own bucket size never leaves the rank)
    mine.v[0] = (int64_t)seed;
    mine.v[1] = (int64_t)call_counter;
    mine.v[2] = k;
    mine.v[3] = sampler;
    mine.v[4] = padding_mode;
    mine.v[5] = default_neighbor_id;
    mine.v[6] = filtered ? filter->type : GLX_FILTER_NONE;
    mine.v[7] = filtered ? filter->field : GLX_FILTER_FIELD_NONE;
    mine.v[8] = filtered ? filter->retry_times : 0;
    mine.v[9] = filtered ? filter->default_timestamp : 0;
    mine.v[10] = n;
  }
  rc = glx_partition_tail(st->device, src, n, P, divert ? rg->map() : GlxIdMap{nullptr, nullptr, 0, 0},
                          divert ? GlxMember{st->rg_bits, st->rg_bits_max} : GlxMember{nullptr, -1}, bucketed, order, d_cnt,
                          mine, s);
  if (rc != GLX_OK) return rc;
  if (filtered && n > 0) {
    // every row's filter value travels with its id (HashPartitioner copies every tensor of a
    // request: hash_partitioner.h:69-74)
    glx_dist_gather_i64_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(filter->values, order, n, vals_b);
  }
  if (shape >= 0) {
    return dist_sample_speculated(st, lg, shape, sampler, n, k, padding_mode, default_neighbor_id, seed, call_counter,
                                  divert ? rg : nullptr, bucketed, order, d_cnt, nbr_out, eid_out, s);
  }
  const int nvals = P + kParams;
  st->h_mat.resize((size_t)P * nvals);
  const int my_pos = lg ? lg->pos : 0;
  rc = exchange_counts(st, d_cnt, nvals, st->h_mat.data(), s, /*window_end=*/false);
  if (rc != GLX_OK) return rc;
  if (lg) ++lg->pos;  // this call's place in the window, speculated or not
  Routing rt;
  routing_from_matrix(st, nvals, &rt);
  const int64_t m = rt.n_recv;
  GLX_REQUIRE(m <= INT32_MAX, "more than 2^31 request rows arrived at one shard");
  bool uniform = true;
  int64_t longest = 0;
  for (int q = 0; q < P; ++q) {
    const int64_t* pq = &st->h_mat[(size_t)q * nvals + P];
    // the response width and the set of tensors that travel are part of the exchange's shape
    GLX_REQUIRE(pq[2] == k, "rank %d asks for neighbor_count %lld, this rank for %d: one collective, one count", q,
                (long long)pq[2], k);
    GLX_REQUIRE((pq[6] != GLX_FILTER_NONE) == filtered, "rank %d and this rank disagree on whether the request has a filter",
                q);
    for (int j = 0; j < 10; ++j) uniform = uniform && pq[j] == mine.v[j];
    longest = longest > pq[10] ? longest : pq[10];
  }
  if (lg && !lg->hold && !filtered && uniform && k > 0 && longest > 0 && my_pos < kLedgerClasses) {
    // a request every rank issued with the same parameters: later calls at this position of the window may skip the
    // count exchange.  Its shape = the largest bucket any rank sent to any owner (the same number on every rank: the
    // whole matrix is here); lengths may differ between the ranks.
    int64_t most = 0;
    for (int q = 0; q < P; ++q)
      for (int p2 = 0; p2 < P; ++p2) most = most > st->h_mat[(size_t)q * nvals + p2] ? most : st->h_mat[(size_t)q * nvals + p2];
    glx_dist_ledger::Shape& sh = lg->shapes[my_pos];
    if (most > sh.rows) sh.rows = most;
    if (sh.rows < 1) sh.rows = 1;  // learned, even if every request at this position was served by a replica
    if (longest > sh.n) sh.n = longest;
    lg->note_share(my_pos);
    ++lg->stats.learned;
  }

  Carver cr;
  const size_t o_ids = cr.take((size_t)(m > 0 ? m : 1) * 8);
  const size_t o_rows = cr.take((size_t)(m > 0 ? m : 1) * 8);
  const size_t o_vin = cr.take(filtered ? (size_t)(m > 0 ? m : 1) * 8 : 0);
  const size_t o_nl = cr.take((size_t)(m > 0 ? m : 1) * k * 8);
  const size_t o_el = cr.take((size_t)(m > 0 ? m : 1) * k * 8);
  rc = st->recv.ensure(cr.at);
  if (rc != GLX_OK) return rc;
  int64_t* ids_in = reinterpret_cast<int64_t*>(st->recv.p + o_ids);
  int64_t* rows_in = reinterpret_cast<int64_t*>(st->recv.p + o_rows);
  int64_t* vals_in = filtered ? reinterpret_cast<int64_t*>(st->recv.p + o_vin) : nullptr;
  int64_t* nbr_loc = reinterpret_cast<int64_t*>(st->recv.p + o_nl);
  int64_t* eid_loc = reinterpret_cast<int64_t*>(st->recv.p + o_el);

  GlxSeg out_segs[3] = {{bucketed, ids_in, 8}, {order, rows_in, 8}, {vals_b, vals_in, 8}};
  rc = st->comm->alltoallv(out_segs, filtered ? 3 : 2, rt.send_counts.data(), rt.send_offs.data(),
                           rt.recv_counts.data(), rt.recv_offs.data(), s);
  if (rc != GLX_OK) return rc;
  int64_t sent = 0;
  for (int q = 0; q < P; ++q) sent += rt.send_counts[q];
  st->sample_rows = n;
  st->sample_rows_replica = n - sent;
  st->sample_rows_remote = sent - rt.send_counts[st->rank];
  if (divert && n > sent) {
    // the replica's bucket: every row answers straight into ITS row of the caller's response (its index in the
    // request is both its random stream and its output row), so these rows need no stitch
    const int64_t chunk_r = k > 0 ? (int64_t)INT32_MAX / k : n;
    for (int64_t lo = sent; lo < n; lo += chunk_r) {
      const int64_t cnt = n - lo < chunk_r ? n - lo : chunk_r;
      rc = glx_sample_scatter_device(rg, sampler, bucketed + lo, order + lo, (int32_t)cnt, k, padding_mode,
                                     default_neighbor_id, seed, call_counter, nbr_out, eid_out, s);
      if (rc != GLX_OK) return rc;
    }
  }

  // Process on the owner: rows draw from the random stream of their ORIGINAL index, with their
  // requester's seed / call counter / flags.  One launch when every rank sent the same
  // parameters (SPMD lockstep), else one per requester.
  const int64_t chunk = k > 0 ? (int64_t)INT32_MAX / k : (m > 0 ? m : 1);
  for (int q = 0; q < (uniform ? 1 : P); ++q) {
    const int64_t begin = uniform ? 0 : rt.recv_offs[q];
    const int64_t end = uniform ? m : rt.recv_offs[q + 1];
    const int64_t* pq = &st->h_mat[(size_t)(uniform ? st->rank : q) * nvals + P];
    for (int64_t lo = begin; lo < end; lo += chunk) {
      const int64_t cnt = end - lo < chunk ? end - lo : chunk;
      glx_filter part;
      const glx_filter* fp = nullptr;
      if (filtered) {
        part.type = (int32_t)pq[6];
        part.field = (int32_t)pq[7];
        part.retry_times = (int32_t)pq[8];
        part.default_timestamp = pq[9];
        part.values = vals_in + lo;
        fp = &part;
      }
      rc = glx_sample_filtered(st->graph, (int)pq[3], ids_in + lo, rows_in + lo, (int32_t)cnt, k, (int)pq[4], pq[5],
                               (uint64_t)pq[0], (uint64_t)pq[1], fp, nbr_loc + lo * k, eid_loc + lo * k,
                               GLX_PTR_DEVICE, s);
      if (rc != GLX_OK) return rc;
    }
  }
  GlxSeg back_segs[2] = {{nbr_loc, nbr_back, (size_t)k * 8}, {eid_loc, eid_back, (size_t)k * 8}};
  rc = st->comm->alltoallv(back_segs, 2, rt.recv_counts.data(), rt.recv_offs.data(), rt.send_counts.data(),
                           rt.send_offs.data(), s);
  if (rc != GLX_OK) return rc;
  if (sent > 0 && k > 0) {  // the rows that travelled (or were served by this rank's own shard)
    const int64_t total = sent * k;
    glx_dist_stitch2_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(nbr_back, eid_back, order, sent, k,
                                                                            nbr_out, eid_out);
    GLX_HIP(hipGetLastError());
  }
  return GLX_OK;
}

// Partitioned FullSampler.  Phase 1 (always): request rows to their owners, the owners' row sizes back, offsets.
// Phase 2 (fill): the owners' values back, every row to its place.  Device pointers.
// filter (device values, one per request row) may be NULL / GLX_FILTER_NONE.  With a filter every row's value travels
// with its id, like every tensor of a partitioned request (hash_partitioner.h:69-74), and the owners answer with
// glx_sample_full_filtered: the row sizes stay the unfiltered ones (full_sampler.cc:55-62), so the sizes half is the same.
int dist_sample_full_device(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit, int32_t* deg_out,
                            int64_t* offsets_out, int64_t* nbr_out, int64_t* eid_out, int64_t capacity, bool fill,
                            int64_t* total_out, hipStream_t s, const glx_filter* filter = nullptr,
                            int padding_mode = GLX_PAD_CIRCULAR, int64_t default_neighbor_id = 0,
                            bool weights_in_eid = false, float default_weight = 0.0f) {
  const int P = st->world;
  const int64_t n = batch, n1 = n > 0 ? n : 1;
  const bool filtered = fill && filter != nullptr && filter->type != GLX_FILTER_NONE;
  ArenaOrder arena_order(st, s);
  Carver cv;
  const size_t o_buck = cv.take((size_t)n1 * 8);
  const size_t o_ord = cv.take((size_t)n1 * 8);
  const size_t o_val = cv.take(filtered ? (size_t)n1 * 8 : 0);
  const size_t o_degb = cv.take((size_t)n1 * 8);
  const size_t o_soff = cv.take((size_t)(n1 + 1) * 8);
  const size_t o_d64 = cv.take((size_t)(n1 + 1) * 8);
  int rc = st->req.ensure(cv.at);
  if (rc != GLX_OK) return rc;
  int64_t* bucketed = reinterpret_cast<int64_t*>(st->req.p + o_buck);
  int64_t* order = reinterpret_cast<int64_t*>(st->req.p + o_ord);
  int64_t* deg_b = reinterpret_cast<int64_t*>(st->req.p + o_degb);
  int64_t* src_off = reinterpret_cast<int64_t*>(st->req.p + o_soff);
  int64_t* deg64 = reinterpret_cast<int64_t*>(st->req.p + o_d64);
  int64_t* vals_b = filtered ? reinterpret_cast<int64_t*>(st->req.p + o_val) : nullptr;
  rc = glx_partition(st->device, src, n, P, bucketed, order, st->d_vals, s);
  if (rc != GLX_OK) return rc;
  if (filtered && n > 0) {
    glx_dist_gather_i64_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(filter->values, order, n, vals_b);
  }
  // the filter's kind rides with the counts: the owners serve every requester with ITS filter, and a rank without a
  // filter among ranks with one is a mismatched request, not a silent difference
  constexpr int kFullParams = 6;
  ReqParams mine;
  memset(&mine, 0, sizeof(mine));
  mine.v[0] = filtered ? filter->type : GLX_FILTER_NONE;
  mine.v[1] = filtered ? filter->field : GLX_FILTER_FIELD_NONE;
  mine.v[2] = filtered ? filter->default_timestamp : 0;
  mine.v[3] = padding_mode;
  mine.v[4] = default_neighbor_id;
  mine.v[5] = max_limit;
  glx_dist_set_params_kernel<<<1, 64, 0, s>>>(st->d_vals + P, mine, kFullParams);
  const int nvals = P + kFullParams;
  st->h_mat.resize((size_t)P * nvals);
  rc = exchange_counts(st, st->d_vals, nvals, st->h_mat.data(), s);
  if (rc != GLX_OK) return rc;
  Routing rt;
  routing_from_matrix(st, nvals, &rt);
  for (int q = 0; q < P; ++q) {
    const int64_t* pq = &st->h_mat[(size_t)q * nvals + P];
    GLX_REQUIRE((pq[0] != GLX_FILTER_NONE) == filtered, "rank %d and this rank disagree on whether the FullSampler request has a filter", q);
    GLX_REQUIRE(pq[5] == max_limit, "rank %d asks for at most %lld neighbours per row, this rank for %d", q, (long long)pq[5], max_limit);
  }
  const int64_t m = rt.n_recv, m1 = m > 0 ? m : 1;
  GLX_REQUIRE(m <= INT32_MAX, "more than 2^31 request rows arrived at one shard");
  GlxTemp ids_in, deg_loc, off_loc, deg_loc64, vals_in;
  GLX_HIP(hipMalloc(&ids_in.p, (size_t)m1 * 8));
  if (filtered) GLX_HIP(hipMalloc(&vals_in.p, (size_t)m1 * 8));
  GLX_HIP(hipMalloc(&deg_loc.p, (size_t)m1 * 4));
  GLX_HIP(hipMalloc(&off_loc.p, (size_t)(m1 + 1) * 8));
  GLX_HIP(hipMalloc(&deg_loc64.p, (size_t)m1 * 8));
  GlxSeg seg_ids[2] = {{bucketed, ids_in.p, 8}, {vals_b, vals_in.p, 8}};
  rc = st->comm->alltoallv(seg_ids, filtered ? 2 : 1, rt.send_counts.data(), rt.send_offs.data(), rt.recv_counts.data(),
                           rt.recv_offs.data(), s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipMemsetAsync(off_loc.p, 0, (size_t)(m1 + 1) * 8, s));
  if (m > 0) {
    rc = glx_sample_full_sizes(st->graph, ids_in.as<int64_t>(), (int32_t)m, max_limit, deg_loc.as<int32_t>(),
                               off_loc.as<int64_t>(), GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
    glx_dist_widen_i32_kernel<<<(unsigned)((m + 255) / 256), 256, 0, s>>>(deg_loc.as<int32_t>(), m, deg_loc64.as<int64_t>());
  }
  GlxSeg seg_deg{deg_loc64.p, deg_b, 8};
  rc = st->comm->alltoallv(&seg_deg, 1, rt.recv_counts.data(), rt.recv_offs.data(), rt.send_counts.data(),
                           rt.send_offs.data(), s);
  if (rc != GLX_OK) return rc;
  int64_t total = 0;
  if (n > 0) {
    glx_dist_stitch_deg_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(deg_b, order, n, deg_out, deg64);
#define SCAN_O(tmp, bytes) rocprim::exclusive_scan(tmp, bytes, deg64, offsets_out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), s)
    GLX_ROCPRIM(SCAN_O);
#undef SCAN_O
#define SCAN_S(tmp, bytes) rocprim::exclusive_scan(tmp, bytes, deg_b, src_off, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), s)
    GLX_ROCPRIM(SCAN_S);
#undef SCAN_S
    int64_t last[2] = {0, 0};
    GLX_HIP(hipMemcpyAsync(&last[0], offsets_out + (n - 1), 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(&last[1], deg64 + (n - 1), 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    total = last[0] + last[1];
  }
  GLX_HIP(hipMemcpyAsync(offsets_out + n, &total, 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipStreamSynchronize(s));  // `total` lives on this frame
  if (total_out) *total_out = total;
  if (!fill) return GLX_OK;
  GLX_REQUIRE(total <= capacity, "the response holds %lld values, the buffers %lld", (long long)total, (long long)capacity);
  // the owners' values: how many each requester gets = the span of its rows in off_loc
  std::vector<int64_t> h_cut((size_t)P + 1, 0);
  for (int q = 0; q <= P; ++q) {
    GLX_HIP(hipMemcpyAsync(&h_cut[(size_t)q], off_loc.as<int64_t>() + rt.recv_offs[q], 8, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  std::vector<int64_t> send_vals((size_t)P), send_voffs((size_t)P);
  for (int q = 0; q < P; ++q) {
    send_vals[(size_t)q] = h_cut[(size_t)q + 1] - h_cut[(size_t)q];
    send_voffs[(size_t)q] = h_cut[(size_t)q];
  }
  GlxTemp d_cnt, nbr_loc, eid_loc, nbr_in, eid_in;
  GLX_HIP(hipMalloc(&d_cnt.p, (size_t)P * 8));
  GLX_HIP(hipMemcpyAsync(d_cnt.p, send_vals.data(), (size_t)P * 8, hipMemcpyHostToDevice, s));
  std::vector<int64_t> mat((size_t)P * P);
  rc = exchange_counts(st, d_cnt.as<int64_t>(), P, mat.data(), s);
  if (rc != GLX_OK) return rc;
  std::vector<int64_t> recv_vals((size_t)P), recv_voffs((size_t)P + 1, 0);
  for (int q = 0; q < P; ++q) {
    recv_vals[(size_t)q] = mat[(size_t)q * P + st->rank];  // what owner q sends to me
    recv_voffs[(size_t)q + 1] = recv_voffs[(size_t)q] + recv_vals[(size_t)q];
  }
  GLX_REQUIRE(recv_voffs[(size_t)P] == total, "the owners announce %lld values, the row sizes add up to %lld",
              (long long)recv_voffs[(size_t)P], (long long)total);
  const int64_t loc_total = h_cut[(size_t)P];
  GLX_HIP(hipMalloc(&nbr_loc.p, (size_t)(loc_total + 1) * 8));
  GLX_HIP(hipMalloc(&eid_loc.p, (size_t)(loc_total + 1) * 8));
  GLX_HIP(hipMalloc(&nbr_in.p, (size_t)(total + 1) * 8));
  GLX_HIP(hipMalloc(&eid_in.p, (size_t)(total + 1) * 8));
  if (m > 0 && loc_total > 0 && !filtered) {
    rc = glx_sample_full(st->graph, ids_in.as<int64_t>(), (int32_t)m, max_limit, off_loc.as<int64_t>(),
                         nbr_loc.as<int64_t>(), eid_loc.as<int64_t>(), GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
    if (weights_in_eid) {  // node2vec: the answer carries edge weights where the edge ids would be
      glx_dist_slot_weights_kernel<<<(unsigned)((m * 64 + 255) / 256), 256, 0, s>>>(
          st->graph->map(), st->graph->row_ptr, st->graph->weight, ids_in.as<int64_t>(), off_loc.as<int64_t>(), m,
          default_weight, eid_loc.as<int64_t>());
      GLX_HIP(hipGetLastError());
    }
  } else if (m > 0 && loc_total > 0) {
    // one launch per requester: its rows, its values, its filter kind / padding / default id.  (A timestamp > value
    // filter uses the first value of the rows a launch serves for all of them, as the reference's servers do with the
    // first value of the part they receive: filter.h:107-111, DESIGN quirk 11.)
    for (int q = 0; q < P; ++q) {
      const int64_t begin = rt.recv_offs[q], cnt = rt.recv_offs[q + 1] - begin;
      if (cnt == 0) continue;
      const int64_t* pq = &st->h_mat[(size_t)q * nvals + P];
      glx_filter part;
      part.type = (int32_t)pq[0];
      part.field = (int32_t)pq[1];
      part.values = vals_in.as<int64_t>() + begin;
      part.retry_times = 0;
      part.default_timestamp = pq[2];
      rc = glx_sample_full_filtered(st->graph, ids_in.as<int64_t>() + begin, (int32_t)cnt, max_limit,
                                    off_loc.as<int64_t>() + begin, (int)pq[3], pq[4], &part, nbr_loc.as<int64_t>(),
                                    eid_loc.as<int64_t>(), GLX_PTR_DEVICE, s);
      if (rc != GLX_OK) return rc;
    }
  }
  GlxSeg seg_vals[2] = {{nbr_loc.p, nbr_in.p, 8}, {eid_loc.p, eid_in.p, 8}};
  rc = st->comm->alltoallv(seg_vals, 2, send_vals.data(), send_voffs.data(), recv_vals.data(), recv_voffs.data(), s);
  if (rc != GLX_OK) return rc;
  if (n > 0 && total > 0) {
    glx_dist_ragged_stitch_kernel<<<(unsigned)((n * 64 + 255) / 256), 256, 0, s>>>(
        order, src_off, deg_b, offsets_out, n, nbr_in.as<int64_t>(), eid_in.as<int64_t>(), nbr_out, eid_out);
    GLX_HIP(hipGetLastError());
  }
  GLX_HIP(hipStreamSynchronize(s));  // the temporaries above are released on return
  return GLX_OK;
}

int check_store(const glx_dist_store* st, int ptr_kind) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  return GLX_OK;
}

}  // namespace

extern "C" int glx_dist_store_create(glx_comm* comm, const glx_graph* graph, const glx_features* features,
                                     glx_dist_store** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(comm != nullptr, "comm is NULL");
  GLX_REQUIRE(graph != nullptr || features != nullptr, "a store needs a graph shard, a feature shard or both");
  GLX_REQUIRE(!graph || graph->device == comm->device, "the graph shard lives on device %d, the communicator on %d",
              graph ? graph->device : -1, comm->device);
  GLX_REQUIRE(!features || features->device == comm->device,
              "the feature shard lives on device %d, the communicator on %d", features ? features->device : -1,
              comm->device);
  GLX_REQUIRE(comm->world <= kMaxWorld, "world size above %d", kMaxWorld);
  GlxDeviceGuard guard(comm->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", comm->device);
  glx_dist_store* st = new (std::nothrow) glx_dist_store();
  GLX_REQUIRE(st != nullptr, "out of host memory");
  st->comm = comm;
  st->graph = graph;
  st->feats = features;
  st->device = comm->device;
  st->rank = comm->rank;
  st->world = comm->world;
  if (const char* e = getenv("GLX_DIST_NO_SHORTCUT")) st->shortcut = atoi(e) == 0;
  for (auto& sl : st->slots) memset(&sl.stats, 0, sizeof(sl.stats));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&st->d_vals), (size_t)(st->world + 16) * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->d_ctr), (size_t)(3 * st->world + 8) * 4);
  if (e != hipSuccess) {
    glx_dist_store_destroy(st);
    GLX_HIP(e);
  }
  *out = st;
  return GLX_OK;
}

extern "C" int glx_dist_build_graph_replica(glx_dist_store* st, const int64_t* hot_ids, int64_t n, int ptr_kind,
                                            void* stream, glx_graph** out) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(n > 0 && n < INT32_MAX, "the hot list must hold between 1 and 2^31 - 1 ids");
  GLX_REQUIRE(hot_ids != nullptr, "NULL data pointer");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  GLX_HIP(hipStreamSynchronize(s));
  const glx_graph* g = st->graph;
  const int P = st->world, me = st->rank;
  bool weighted = g->weight != nullptr;  // agreed on below: a shard without edges cannot tell
  const int64_t n1 = n > 0 ? n : 1;
  GlxTemp ids_d, sorted, bucketed, order, deg_mine, off_mine;
  const int64_t* d_hot = hot_ids;
  if (ptr_kind == GLX_PTR_HOST && n > 0) {
    GLX_HIP(hipMalloc(&ids_d.p, (size_t)n * 8));
    GLX_HIP(hipMemcpyAsync(ids_d.p, hot_ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    d_hot = ids_d.as<int64_t>();
  }
  // ascending ids: every rank then lists every owner's rows in the same order
  GLX_HIP(hipMalloc(&sorted.p, (size_t)n1 * 8));
  if (n > 0) {
#define SORTK(tmp, bytes) rocprim::radix_sort_keys(tmp, bytes, d_hot, sorted.as<int64_t>(), (size_t)n, 0, 64, s)
    GLX_ROCPRIM(SORTK);
#undef SORTK
  }
  GLX_HIP(hipMalloc(&bucketed.p, (size_t)n1 * 8));
  GLX_HIP(hipMalloc(&order.p, (size_t)n1 * 8));
  rc = glx_partition(st->device, sorted.as<int64_t>(), n, P, bucketed.as<int64_t>(), order.as<int64_t>(), st->d_vals, s);
  if (rc != GLX_OK) return rc;
  std::vector<int64_t> counts((size_t)P);
  GLX_HIP(hipMemcpyAsync(counts.data(), st->d_vals, (size_t)P * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  std::vector<int64_t> offs((size_t)P + 1, 0);
  for (int p = 0; p < P; ++p) offs[p + 1] = offs[p] + counts[p];
  const int64_t c_me = counts[me];
  const int64_t* my_ids = bucketed.as<int64_t>() + offs[me];
  // this owner's piece: degrees, then the slots
  GLX_HIP(hipMalloc(&deg_mine.p, (size_t)(c_me + 1) * 8));
  GLX_HIP(hipMalloc(&off_mine.p, (size_t)(c_me + 1) * 8));
  int64_t e_me = 0;
  if (c_me > 0) {
    glx_dist_rep_deg_kernel<<<(unsigned)((c_me + 255) / 256), 256, 0, s>>>(g->map(), g->row_ptr, my_ids, c_me,
                                                                          deg_mine.as<int64_t>());
#define SCANO(tmp, bytes)                                                                                      \
  rocprim::exclusive_scan(tmp, bytes, deg_mine.as<int64_t>(), off_mine.as<int64_t>(), (int64_t)0, (size_t)c_me, \
                          rocprim::plus<int64_t>(), s)
    GLX_ROCPRIM(SCANO);
#undef SCANO
    int64_t last[2] = {0, 0};
    GLX_HIP(hipMemcpyAsync(&last[0], off_mine.as<int64_t>() + (c_me - 1), 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(&last[1], deg_mine.as<int64_t>() + (c_me - 1), 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    e_me = last[0] + last[1];
  }
  GlxTemp col_mine, eid_mine, w_mine;
  GLX_HIP(hipMalloc(&col_mine.p, (size_t)(e_me + 1) * 8));
  GLX_HIP(hipMalloc(&eid_mine.p, (size_t)(e_me + 1) * 8));
  GLX_HIP(hipMalloc(&w_mine.p, (size_t)(e_me + 1) * 4));
  if (c_me > 0) {
    glx_dist_rep_rows_kernel<<<(unsigned)((c_me * 64 + 255) / 256), 256, 0, s>>>(
        g->map(), g->row_ptr, g->adj, weighted ? g->weight : nullptr, my_ids, c_me, off_mine.as<int64_t>(),
        col_mine.as<int64_t>(), eid_mine.as<int64_t>(), weighted ? w_mine.as<float>() : nullptr);
    GLX_HIP(hipGetLastError());
  }
  // every owner's edge total, and whether the edge type is weighted (a shard without edges holds no weight array)
  GlxTemp d_e;
  GLX_HIP(hipMalloc(&d_e.p, 16));
  const int64_t mine2[2] = {e_me, weighted ? 1 : 0};
  GLX_HIP(hipMemcpyAsync(d_e.p, mine2, 16, hipMemcpyHostToDevice, s));
  std::vector<int64_t> all2((size_t)P * 2), e_all((size_t)P);
  rc = exchange_counts(st, d_e.as<int64_t>(), 2, all2.data(), s);
  if (rc != GLX_OK) return rc;
  std::vector<int64_t> eoffs((size_t)P + 1, 0);
  for (int p = 0; p < P; ++p) {
    e_all[p] = all2[(size_t)p * 2];
    weighted = weighted || all2[(size_t)p * 2 + 1] != 0;
    eoffs[p + 1] = eoffs[p] + e_all[p];
  }
  const int64_t E = eoffs[P];
  GLX_REQUIRE(E < ((int64_t)1 << 40), "replica too large");
  // all-gather(v) of the pieces (an all-to-all whose every outgoing message is the same buffer), owner-major
  GlxTemp deg_all, col_all, eid_all, w_all, row_ptr;
  GLX_HIP(hipMalloc(&deg_all.p, (size_t)n1 * 8));
  GLX_HIP(hipMalloc(&col_all.p, (size_t)(E + 1) * 8));
  GLX_HIP(hipMalloc(&eid_all.p, (size_t)(E + 1) * 8));
  GLX_HIP(hipMalloc(&w_all.p, (size_t)(E + 1) * 4));
  std::vector<int64_t> same_rows((size_t)P, c_me), same_edges((size_t)P, e_me), zero((size_t)P, 0);
  GlxSeg seg_rows{deg_mine.p, deg_all.p, 8};
  rc = st->comm->alltoallv(&seg_rows, 1, same_rows.data(), zero.data(), counts.data(), offs.data(), s);
  if (rc != GLX_OK) return rc;
  GlxSeg seg_edges[3] = {{col_mine.p, col_all.p, 8}, {eid_mine.p, eid_all.p, 8}, {w_mine.p, w_all.p, 4}};
  rc = st->comm->alltoallv(seg_edges, weighted ? 3 : 2, same_edges.data(), zero.data(), e_all.data(), eoffs.data(), s);
  if (rc != GLX_OK) return rc;
  // rows in owner-major order = `bucketed`; row_ptr = exclusive scan of the degrees (+ the total)
  GLX_HIP(hipMalloc(&row_ptr.p, (size_t)(n + 1) * 8));
  if (n > 0) {
#define SCANR(tmp, bytes)                                                                                    \
  rocprim::exclusive_scan(tmp, bytes, deg_all.as<int64_t>(), row_ptr.as<int64_t>(), (int64_t)0, (size_t)n,   \
                          rocprim::plus<int64_t>(), s)
    GLX_ROCPRIM(SCANR);
#undef SCANR
  }
  GLX_HIP(hipMemcpyAsync(row_ptr.as<int64_t>() + n, &E, 8, hipMemcpyHostToDevice, s));
  GLX_HIP(hipStreamSynchronize(s));
  return glx_graph_create(st->device, n, E, row_ptr.as<int64_t>(), col_all.as<int64_t>(), eid_all.as<int64_t>(),
                          weighted ? w_all.as<float>() : nullptr, bucketed.as<int64_t>(), GLX_PTR_DEVICE, s, out);
}

extern "C" int glx_dist_store_set_graph_replica(glx_dist_store* st, const glx_graph* replica) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  if (replica != nullptr) {
    GLX_REQUIRE(replica->device == st->device, "the replica lives on device %d, the store on %d", replica->device,
                st->device);
    GLX_REQUIRE(replica->idmap.any(), "a graph replica needs its vertex ids (glx_graph_build with ids)");
    // the request partition gets one more bucket for the replica (glx_partition_divert: at most 64 buckets); refuse
    // here, where it is a configuration error, not inside a collective sample call
    GLX_REQUIRE(st->world + 1 <= 64, "a graph replica needs world size <= 63 (the request partition has 64 buckets), got %d",
                st->world);
    GLX_REQUIRE(st->graph->num_edges == 0 || replica->num_edges == 0 ||
                    (replica->weight != nullptr) == (st->graph->weight != nullptr),
                "the replica and the shard must both be weighted or both unweighted");
  }
  st->graph_replica = nullptr;
  if (st->rg_bits) (void)hipFree(st->rg_bits);
  st->rg_bits = nullptr;
  st->rg_bits_max = -1;
  if (replica != nullptr && replica->idmap.keys != nullptr && replica->num_rows > 0 &&
      glx_side_knobs().dist_no_bitmap.load(std::memory_order_relaxed) <= 0) {
    // Hashed vertex ids: every id of every sampling request probes that table once to pick its bucket (a random DRAM
    // access per id: 34 us for the 1.6 M ids of a hop-2 request).  When the ids are small non-negative numbers, one
    // bit per id of the range answers the same question from a few MB that stay in L2.  Same bound on the range as
    // the feature replica's rank records: at most 4 words per listed id (+ a floor), else the hash map stays.
    GlxDeviceGuard guard(st->device);
    GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
    hipStream_t s = nullptr;
    GlxTemp d_mm;
    GLX_HIP(hipMalloc(&d_mm.p, 16));
    const int64_t init[2] = {INT64_MAX, INT64_MIN};
    GLX_HIP(hipMemcpyAsync(d_mm.p, init, 16, hipMemcpyHostToDevice, s));
    const int64_t cap = (int64_t)replica->idmap.cap;
    glx_dist_keys_minmax_kernel<<<grid_for(cap, 1024), 256, 0, s>>>(replica->idmap.keys, cap, d_mm.as<long long>());
    int64_t mm[2] = {0, -1};
    GLX_HIP(hipMemcpyAsync(mm, d_mm.p, 16, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    const int64_t words = mm[1] >= 0 ? (mm[1] >> 6) + 1 : 0;
    if (mm[0] >= 0 && mm[1] < ((int64_t)1 << 33) && words <= 4 * replica->num_rows + 4096) {
      uint64_t* bits = nullptr;
      GLX_HIP(hipMalloc(reinterpret_cast<void**>(&bits), (size_t)words * 8));
      hipError_t e = hipMemsetAsync(bits, 0, (size_t)words * 8, s);
      if (e == hipSuccess) {
        glx_dist_keys_setbits_kernel<<<grid_for(cap, 1024), 256, 0, s>>>(replica->idmap.keys, cap,
                                                                         reinterpret_cast<unsigned long long*>(bits));
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        (void)hipFree(bits);
        GLX_HIP(e);
      }
      st->rg_bits = bits;
      st->rg_bits_max = mm[1];
    }
  }
  st->graph_replica = replica;
  return GLX_OK;
}

extern "C" int glx_dist_last_sample_rows(const glx_dist_store* st, int64_t* rows, int64_t* from_replica,
                                         int64_t* remote) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  if (rows) *rows = st->sample_rows;
  if (from_replica) *from_replica = st->sample_rows_replica;
  if (remote) *remote = st->sample_rows_remote;
  return GLX_OK;
}

extern "C" void glx_dist_store_destroy(glx_dist_store* st) {
  if (!st) return;
  GlxDeviceGuard guard(st->device);
  (void)hipDeviceSynchronize();
  if (st->arena_free) (void)hipEventDestroy(st->arena_free);
  st->req.release();
  st->recv.release();
  st->r_req.release();
  st->r_recv.release();
  st->r_back.release();
  for (auto& sl : st->slots) {
    sl.req.release();
    sl.recv.release();
    sl.tab.release();
    sl.halo.release();
  }
  if (st->d_vals) (void)hipFree(st->d_vals);
  if (st->d_ctr) (void)hipFree(st->d_ctr);
  if (st->cache) glx_features_destroy(st->cache);
  if (st->cache_slots) (void)hipFree(st->cache_slots);
  if (st->bm_member) (void)hipFree(st->bm_member);
  if (st->rg_bits) (void)hipFree(st->rg_bits);
  if (st->own_id) (void)hipFree(st->own_id);
  if (st->own_cnt) (void)hipFree(st->own_cnt);
  delete st;
}

extern "C" int glx_dist_last_stats(const glx_dist_store* st, glx_dist_stats* out) {
  GLX_REQUIRE(st != nullptr && out != nullptr, "NULL argument");
  *out = st->slots[st->last_slot].stats;
  out->host_syncs = st->host_syncs;
  out->host_stall_us = st->host_stall_us;
  return GLX_OK;
}

extern "C" int glx_dist_ledger_create(int device, glx_dist_ledger** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  glx_dist_ledger* l = new (std::nothrow) glx_dist_ledger();
  GLX_REQUIRE(l != nullptr, "out of host memory");
  l->device = device;
  memset(&l->stats, 0, sizeof(l->stats));
  const size_t words = 1 + kLedgerClasses, stage = kMaxWorld + 32 + kLedgerTail;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&l->d_words), (words + stage) * 8);
  if (e == hipSuccess) e = hipMemset(l->d_words, 0, (words + stage) * 8);
  if (e != hipSuccess) {
    glx_dist_ledger_destroy(l);
    GLX_HIP(e);
  }
  l->d_stage = l->d_words + words;
  *out = l;
  return GLX_OK;
}

extern "C" void glx_dist_ledger_destroy(glx_dist_ledger* l) {
  if (!l) return;
  GlxDeviceGuard guard(l->device);
  (void)hipDeviceSynchronize();
  if (l->d_words) (void)hipFree(l->d_words);
  delete l;
}

extern "C" int glx_dist_store_set_ledger(glx_dist_store* st, glx_dist_ledger* l) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GLX_REQUIRE(l == nullptr || l->device == st->device, "the ledger lives on device %d, the store on %d", l ? l->device : -1,
              st->device);
  st->ledger = l;
  return GLX_OK;
}

extern "C" int glx_dist_confirm(glx_dist_store* st, void* stream) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  if (st->ledger == nullptr) return GLX_OK;  // nothing speculates: every call confirmed itself
  std::vector<int64_t> sink((size_t)st->world);
  // one value so the exchange has a body; the ledger's words ride behind it
  return exchange_counts(st, st->ledger->d_stage + kMaxWorld + 31, 1, sink.data(), glx_stream(stream));
}

extern "C" int glx_dist_ledger_get_stats(const glx_dist_ledger* l, glx_dist_ledger_stats* out) {
  GLX_REQUIRE(l != nullptr && out != nullptr, "NULL argument");
  *out = l->stats;
  return GLX_OK;
}

extern "C" int glx_dist_ledger_set_slack(glx_dist_ledger* l, double slack, int64_t pad_rows) {
  GLX_REQUIRE(l != nullptr, "ledger is NULL");
  GLX_REQUIRE(slack > 0.0 && pad_rows >= 0, "slack must be positive, pad_rows non-negative");
  l->slack = slack;
  l->pad_rows = pad_rows;
  return GLX_OK;
}

extern "C" int glx_dist_sample(glx_dist_store* st, int sampler, const int64_t* src, int32_t batch, int32_t k,
                               int padding_mode, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                               const glx_filter* filter, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                               void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(batch >= 0 && k >= 0, "negative batch / neighbor_count");
  GLX_REQUIRE(padding_mode == GLX_PAD_CIRCULAR || padding_mode == GLX_PAD_REPLICATE, "bad padding_mode %d",
              padding_mode);
  GLX_REQUIRE((int64_t)batch * k <= INT32_MAX, "batch * neighbor_count exceeds int32 (tensor.h:47)");
  GLX_REQUIRE(batch == 0 || k == 0 || (src && nbr_out && eid_out), "NULL data pointer");
  const bool filtered = filter != nullptr && filter->type != GLX_FILTER_NONE;
  GLX_REQUIRE(!filtered || batch == 0 || filter->values != nullptr, "filter without values");
  // InDegreeSampler weighs a neighbour by its in-degree over ALL shards; a shard's tables
  // (glx_graph_enable_in_degree) only count the edges it owns.
  GLX_REQUIRE(sampler != GLX_SAMPLER_IN_DEGREE || st->world == 1 || st->graph->indeg_global,
              "InDegreeSampler on a partitioned store needs glx_dist_enable_in_degree(): a shard's own in-degree "
              "tables cover its own edges only");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  if (ptr_kind == GLX_PTR_DEVICE) {
    return dist_sample_device(st, sampler, src, batch, k, padding_mode, default_neighbor_id, seed, call_counter,
                              filter, nbr_out, eid_out, glx_stream(stream));
  }
  hipStream_t s = glx_host_call_stream(stream, st->device);
  const size_t n_out = (size_t)batch * k;
  GlxTemp d;
  GLX_HIP(hipMalloc(&d.p, ((size_t)batch * 2 + 2 * n_out + 4) * 8));
  int64_t* d_src = d.as<int64_t>();
  int64_t* d_val = d_src + batch;
  int64_t* d_nbr = d_val + batch;
  int64_t* d_eid = d_nbr + n_out;
  if (batch) GLX_HIP(hipMemcpyAsync(d_src, src, (size_t)batch * 8, hipMemcpyHostToDevice, s));
  glx_filter dev_filter;
  const glx_filter* fp = nullptr;
  if (filtered) {
    if (batch) GLX_HIP(hipMemcpyAsync(d_val, filter->values, (size_t)batch * 8, hipMemcpyHostToDevice, s));
    dev_filter = *filter;
    dev_filter.values = d_val;
    fp = &dev_filter;
  }
  rc = dist_sample_device(st, sampler, d_src, batch, k, padding_mode, default_neighbor_id, seed, call_counter, fp,
                          d_nbr, d_eid, s);
  hipError_t e = hipSuccess;
  if (rc == GLX_OK && n_out) {
    e = hipMemcpyAsync(nbr_out, d_nbr, n_out * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(eid_out, d_eid, n_out * 8, hipMemcpyDeviceToHost, s);
  }
  hipError_t e2 = hipStreamSynchronize(s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(e);
  GLX_HIP(e2);
  return GLX_OK;
}

namespace {
// Both entry points: host pointers are staged (ids in; sizes, offsets and values out), device pointers are used as given.
int dist_sample_full_any(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit, int32_t* degrees_out,
                         int64_t* offsets_out, int64_t* nbr_out, int64_t* eid_out, int64_t capacity, bool fill, int ptr_kind,
                         void* stream, const glx_filter* filter = nullptr, int padding_mode = GLX_PAD_CIRCULAR,
                         int64_t default_neighbor_id = 0) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  const bool filtered = fill && filter != nullptr && filter->type != GLX_FILTER_NONE;
  GLX_REQUIRE(!filtered || batch == 0 || filter->values != nullptr, "the filter has no values");
  GLX_REQUIRE(padding_mode == GLX_PAD_CIRCULAR || padding_mode == GLX_PAD_REPLICATE, "bad padding_mode %d", padding_mode);
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(batch >= 0 && offsets_out != nullptr && (batch == 0 || (src && degrees_out)), "bad request");
  GLX_REQUIRE(!fill || (capacity >= 0 && (capacity == 0 || (nbr_out && eid_out))), "bad response buffers");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  const size_t nb = (size_t)batch;
  GlxTemp stage;
  const int64_t* d_src = src;
  int32_t* d_deg = degrees_out;
  int64_t* d_off = offsets_out;
  int64_t* d_nbr = nbr_out;
  int64_t* d_eid = eid_out;
  glx_filter dev_filter;
  if (filtered) dev_filter = *filter;
  if (ptr_kind == GLX_PTR_HOST) {
    const size_t cap = fill ? (size_t)capacity : 0;
    GLX_HIP(hipMalloc(&stage.p, (nb + (nb + 1) + 2 * cap + 2 + nb) * 8 + (nb + 2) * 4));
    int64_t* b = stage.as<int64_t>();
    if (nb) GLX_HIP(hipMemcpyAsync(b, src, nb * 8, hipMemcpyHostToDevice, s));
    d_src = b;
    d_off = b + nb;
    d_nbr = d_off + nb + 1;
    d_eid = d_nbr + cap;
    int64_t* d_val = d_eid + cap;
    d_deg = reinterpret_cast<int32_t*>(d_val + nb);
    if (filtered && nb) {
      GLX_HIP(hipMemcpyAsync(d_val, filter->values, nb * 8, hipMemcpyHostToDevice, s));
      dev_filter.values = d_val;
    }
  }
  int64_t total = 0;
  if (st->world == 1 && st->shortcut) {
    rc = glx_sample_full_sizes(st->graph, d_src, batch, max_limit, d_deg, d_off, GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemcpyAsync(&total, d_off + batch, 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    if (fill) {
      GLX_REQUIRE(total <= capacity, "the response holds %lld values, the buffers %lld", (long long)total, (long long)capacity);
      rc = glx_sample_full_filtered(st->graph, d_src, batch, max_limit, d_off, padding_mode, default_neighbor_id,
                                    filtered ? &dev_filter : nullptr, d_nbr, d_eid, GLX_PTR_DEVICE, s);
      if (rc != GLX_OK) return rc;
    }
  } else {
    rc = dist_sample_full_device(st, d_src, batch, max_limit, d_deg, d_off, d_nbr, d_eid, capacity, fill, &total, s,
                                 filtered ? &dev_filter : nullptr, padding_mode, default_neighbor_id);
    if (rc != GLX_OK) return rc;
  }
  if (ptr_kind == GLX_PTR_HOST) {
    if (nb) GLX_HIP(hipMemcpyAsync(degrees_out, d_deg, nb * 4, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(offsets_out, d_off, (nb + 1) * 8, hipMemcpyDeviceToHost, s));
    if (fill && total > 0) {
      GLX_HIP(hipMemcpyAsync(nbr_out, d_nbr, (size_t)total * 8, hipMemcpyDeviceToHost, s));
      GLX_HIP(hipMemcpyAsync(eid_out, d_eid, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    }
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}
}  // namespace

extern "C" int glx_dist_sample_full_sizes(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit,
                                          int32_t* degrees_out, int64_t* offsets_out, int ptr_kind, void* stream) {
  return dist_sample_full_any(st, src, batch, max_limit, degrees_out, offsets_out, nullptr, nullptr, 0, false, ptr_kind, stream);
}

extern "C" int glx_dist_sample_full(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit,
                                    int32_t* degrees_out, int64_t* offsets_out, int64_t* nbr_out, int64_t* eid_out,
                                    int64_t capacity, int ptr_kind, void* stream) {
  return dist_sample_full_any(st, src, batch, max_limit, degrees_out, offsets_out, nbr_out, eid_out, capacity, true, ptr_kind,
                              stream);
}

extern "C" int glx_dist_sample_full_filtered(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit,
                                             int32_t* degrees_out, int64_t* offsets_out, int padding_mode,
                                             int64_t default_neighbor_id, const glx_filter* filter, int64_t* nbr_out,
                                             int64_t* eid_out, int64_t capacity, int ptr_kind, void* stream) {
  return dist_sample_full_any(st, src, batch, max_limit, degrees_out, offsets_out, nbr_out, eid_out, capacity, true, ptr_kind,
                              stream, filter, padding_mode, default_neighbor_id);
}

namespace {
int dist_random_walk(glx_dist_store* st, const int64_t* seeds, int32_t batch, int32_t walk_len, float p, float q,
                     int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id, uint64_t seed,
                     uint64_t call_counter, int64_t* walks_out, int ptr_kind, void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(batch >= 0 && walk_len >= 0, "negative batch / walk_len");
  GLX_REQUIRE((int64_t)batch * walk_len <= INT32_MAX, "batch * walk_len exceeds int32 (tensor.h:47)");
  GLX_REQUIRE(batch == 0 || walk_len == 0 || (seeds && walks_out), "NULL data pointer");
  const bool deep = fabsf(p - 1.0f) < 32 * 1.1920929e-07f && fabsf(q - 1.0f) < 32 * 1.1920929e-07f;
  GLX_REQUIRE(deep || (full_nbr_num >= 1 && full_nbr_num <= 2048), "DefaultFullNbrNum must be in [1, 2048], got %d",
              full_nbr_num);
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  if (!deep && st->world == 1 && st->shortcut) {
    return glx_random_walk(st->graph, seeds, batch, walk_len, p, q, full_nbr_num, default_weight, default_neighbor_id, seed,
                           call_counter, walks_out, ptr_kind, stream);
  }
  const size_t nb = (size_t)(batch > 0 ? batch : 1), total = (size_t)batch * (size_t)walk_len;
  GlxTemp buf;
  GLX_HIP(hipMalloc(&buf.p, (nb * 4 + (ptr_kind == GLX_PTR_HOST ? total : 0) + 2) * 8));
  int64_t* cur = buf.as<int64_t>();
  int64_t* nxt = cur + nb;
  int64_t* eid = nxt + nb;
  int64_t* par = eid + nb;  // node2vec: the vertex each walker came from
  int64_t* d_walks = ptr_kind == GLX_PTR_HOST ? par + nb : walks_out;
  if (batch > 0) {
    GLX_HIP(hipMemcpyAsync(cur, seeds, (size_t)batch * 8, ptr_kind == GLX_PTR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
  }
  if (deep) {
    // DeepWalk (random_walk.cc:168-190): step t of walker i = RandomSampler's draw 0 of the stream (seed, call_counter + t,
    // i) on the vertex it stands on -- one partitioned request with neighbor_count 1 per step, every walker a row; a
    // walker on a vertex without out-edges continues from the default id, like the single-store walk
    // Walks ALWAYS take the count exchange (glx.h): a step feeds the next one, so a step speculated on capacities
    // learned from some other request's shape could lose rows and the walk would carry the loss on unnoticed until a
    // later confirmation.  The ledger is detached for the steps (their positions are not recorded either).
    struct LedgerOff {
      glx_dist_store* st;
      glx_dist_ledger* saved;
      explicit LedgerOff(glx_dist_store* s) : st(s), saved(s->ledger) { st->ledger = nullptr; }
      ~LedgerOff() { st->ledger = saved; }
    } ledger_off(st);
    for (int32_t t = 0; t < walk_len; ++t) {
      rc = dist_sample_device(st, GLX_SAMPLER_RANDOM, cur, batch, 1, GLX_PAD_CIRCULAR, default_neighbor_id, seed,
                              call_counter + (uint64_t)t, nullptr, nxt, eid, s);
      if (rc != GLX_OK) return rc;
      if (batch > 0) {
        glx_dist_walk_column_kernel<<<(unsigned)((batch + 255) / 256), 256, 0, s>>>(nxt, batch, walk_len, t, d_walks);
      }
      int64_t* tmp = cur;
      cur = nxt;
      nxt = tmp;
    }
  } else {
    // node2vec (WeightedRandomWalk, random_walk.cc:192-272): a step needs the first F neighbours (+ weights) of the
    // vertex the walker stands on and the first F neighbours of the vertex it came from.  Per step ONE partitioned
    // FullSampler request (limit F) brings the current vertices' lists to the requester, weights in place of the edge ids;
    // the parents' lists are the previous step's; the step itself then runs here, with the single store's arithmetic
    // and draws -- walks are identical to glx_random_walk on the unpartitioned graph.
    const int32_t F = full_nbr_num;
    const size_t cap = nb * (size_t)F;
    GlxTemp lists;
    GLX_HIP(hipMalloc(&lists.p, 2 * ((nb + 1) * 4 + (nb + 2) * 8 + 2 * cap * 8) + 64 + (nb + 1) * 8));
    char* base = lists.as<char>();
    struct Lists { int32_t* deg; int64_t* off; int64_t* nbr; int64_t* w; } L[2];
    for (int k = 0; k < 2; ++k) {
      L[k].off = reinterpret_cast<int64_t*>(base);
      base += (nb + 2) * 8;
      L[k].nbr = reinterpret_cast<int64_t*>(base);
      base += cap * 8;
      L[k].w = reinterpret_cast<int64_t*>(base);
      base += cap * 8;
      L[k].deg = reinterpret_cast<int32_t*>(base);
      base += ((nb + 1) * 4 + 7) & ~(size_t)7;
    }
    int64_t* cursor = reinterpret_cast<int64_t*>(base);  // [nb]: the reference's cursor into the parents' lists
    if (batch > 0) GLX_HIP(hipMemcpyAsync(par, cur, (size_t)batch * 8, hipMemcpyDeviceToDevice, s));  // step 0: the seed itself
    for (int32_t t = 0; t < walk_len; ++t) {
      Lists& c = L[t & 1];
      Lists& pv = L[(t & 1) ^ 1];
      int64_t tot = 0;
      rc = dist_sample_full_device(st, cur, batch, F, c.deg, c.off, c.nbr, c.w, (int64_t)cap, true, &tot, s, nullptr,
                                   GLX_PAD_CIRCULAR, 0, /*weights_in_eid=*/true, default_weight);
      if (rc != GLX_OK) return rc;
      if (batch > 0) {
        if (t > 0) glx_dist_walk_cursor_kernel<<<1, 1024, 0, s>>>(c.deg, pv.deg, (int64_t)batch, cursor);
        glx_dist_node2vec_step_kernel<<<(unsigned)batch, 64, (size_t)F * 20, s>>>(
            par, t, c.deg, c.off, c.nbr, c.w, t > 0 ? pv.deg : nullptr, cursor, pv.nbr, p, q, F, seed, call_counter,
            default_neighbor_id, nxt);
        glx_dist_walk_column_kernel<<<(unsigned)((batch + 255) / 256), 256, 0, s>>>(nxt, batch, walk_len, t, d_walks);
        // the next step's parent is this step's vertex -- except after step 0, whose parent stays the seed
        // (random_walk_request.cc:120-131: t <= 1 -> the seed): the seed IS this step's vertex then
        GLX_HIP(hipMemcpyAsync(par, cur, (size_t)batch * 8, hipMemcpyDeviceToDevice, s));
      }
      int64_t* tmp = cur;
      cur = nxt;
      nxt = tmp;
    }
    GLX_HIP(hipStreamSynchronize(s));  // `lists` is released on return
  }
  GLX_HIP(hipGetLastError());
  if (ptr_kind == GLX_PTR_HOST && total > 0) {
    GLX_HIP(hipMemcpyAsync(walks_out, d_walks, total * 8, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}
}  // namespace

extern "C" int glx_dist_random_walk(glx_dist_store* st, const int64_t* seeds, int32_t batch, int32_t walk_len, float p,
                                    float q, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                                    int64_t* walks_out, int ptr_kind, void* stream) {
  // node2vec with the reference's defaults: GLOBAL_FLAG(DefaultFullNbrNum) = 100, GLOBAL_FLAG(DefaultWeight) = 0
  // (config.cc:102,111); glx_dist_random_walk_ex takes them explicitly
  return dist_random_walk(st, seeds, batch, walk_len, p, q, 100, 0.0f, default_neighbor_id, seed, call_counter, walks_out,
                          ptr_kind, stream);
}

extern "C" int glx_dist_random_walk_ex(glx_dist_store* st, const int64_t* seeds, int32_t batch, int32_t walk_len, float p,
                                       float q, int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id,
                                       uint64_t seed, uint64_t call_counter, int64_t* walks_out, int ptr_kind,
                                       void* stream) {
  return dist_random_walk(st, seeds, batch, walk_len, p, q, full_nbr_num, default_weight, default_neighbor_id, seed,
                          call_counter, walks_out, ptr_kind, stream);
}

extern "C" int glx_dist_aggregate(glx_dist_store* st, int op, const int64_t* node_ids, const int32_t* segment_ids,
                                  int32_t num_ids, int32_t num_segments, float default_attr, float* emb_out,
                                  int32_t* cnt_out, int ptr_kind, void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->feats != nullptr, "this store has no feature shard");
  GLX_REQUIRE(op >= GLX_AGG_SUM && op <= GLX_AGG_PROD, "unknown aggregator id %d", op);
  GLX_REQUIRE(num_ids >= 0 && num_segments >= 0, "negative sizes");
  GLX_REQUIRE((int64_t)num_segments * st->feats->dim <= INT32_MAX, "num_segments * dim exceeds int32 (tensor.h:47)");
  GLX_REQUIRE(num_segments == 0 || (emb_out && cnt_out), "NULL output pointer");
  GLX_REQUIRE(num_ids == 0 || node_ids, "NULL data pointer");
  GLX_REQUIRE(segment_ids != nullptr || num_segments == 0 || num_ids % num_segments == 0,
              "segment_ids == NULL means equal segments: num_ids must be a multiple of num_segments");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  const glx_features* f = st->feats;
  if (st->world == 1 && st->shortcut && st->cache == nullptr) {
    return glx_aggregate(f, op, node_ids, segment_ids, num_ids, num_segments, default_attr, emb_out, cnt_out,
                         ptr_kind, stream);
  }
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  const int64_t* d_ids = node_ids;
  const int32_t* d_seg = segment_ids;
  float* d_emb = emb_out;
  int32_t* d_cnt = cnt_out;
  GlxTemp stage;
  const size_t emb_n = (size_t)num_segments * f->dim;
  if (ptr_kind == GLX_PTR_HOST) {
    const size_t emb_b = (emb_n * 4 + 255) & ~(size_t)255;
    const size_t ids_b = ((size_t)num_ids * 8 + 255) & ~(size_t)255;
    const size_t seg_b = ((size_t)num_ids * 4 + 255) & ~(size_t)255;
    GLX_HIP(hipMalloc(&stage.p, emb_b + ids_b + seg_b + (size_t)num_segments * 4 + 256));
    char* b = stage.as<char>();
    d_emb = reinterpret_cast<float*>(b);
    int64_t* ids_w = reinterpret_cast<int64_t*>(b + emb_b);
    int32_t* seg_w = reinterpret_cast<int32_t*>(b + emb_b + ids_b);
    d_cnt = reinterpret_cast<int32_t*>(b + emb_b + ids_b + seg_b);
    if (num_ids) GLX_HIP(hipMemcpyAsync(ids_w, node_ids, (size_t)num_ids * 8, hipMemcpyHostToDevice, s));
    if (num_ids && segment_ids) {
      GLX_HIP(hipMemcpyAsync(seg_w, segment_ids, (size_t)num_ids * 4, hipMemcpyHostToDevice, s));
    }
    d_ids = ids_w;
    d_seg = segment_ids ? seg_w : nullptr;
  }
  Resolved rs;
  rc = resolve_and_fetch(st, 0, d_ids, num_ids, default_attr, s, &rs);
  st->slots[0].num_ids = -1;  // a whole call: nothing left pending in the slot
  if (rc == GLX_OK && num_segments > 0) {
    rc = glx_aggregate_vrows_device(rs.src, 3, f->dim, op, rs.loc, d_seg, num_ids, num_segments, default_attr, d_emb,
                                    d_cnt, s);
  }
  if (ptr_kind == GLX_PTR_HOST) {
    hipError_t e = hipSuccess;
    if (rc == GLX_OK && num_segments > 0) {
      e = hipMemcpyAsync(emb_out, d_emb, emb_n * 4, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipMemcpyAsync(cnt_out, d_cnt, (size_t)num_segments * 4, hipMemcpyDeviceToHost, s);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    if (rc != GLX_OK) return rc;
    GLX_HIP(e);
    GLX_HIP(e2);
  }
  return rc;
}

namespace {

__global__ void glx_dist_seg_of_kernel(const int32_t* __restrict__ seg, const int64_t* __restrict__ order, int64_t n,
                                       int32_t fanout, int32_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t at = order[i];
  out[i] = seg ? seg[at] : (int32_t)(at / fanout);  // a dense sampler response: segment = request row
}

// Design R: the reference's own distributed aggregation -- AggregatingRequest::Partition sends every
// (id, segment id) to the id's owner, the owner reduces what it received per segment, AggregatingResponse::Stitch
// folds the P partial [Sg, D] results on the requester (aggregating_request.cc:117-213).  Device pointers.
int dist_aggregate_partial_device(glx_dist_store* st, int op, const int64_t* d_ids, const int32_t* d_seg, int32_t num_ids,
                                  int32_t num_segments, float default_attr, float* d_emb, int32_t* d_cnt, hipStream_t s) {
  const int P = st->world, me = st->rank;
  const glx_features* f = st->feats;
  const int32_t D = f->dim;
  const int64_t n = num_ids, n1 = n > 0 ? n : 1;
  const int32_t fanout = (d_seg == nullptr && num_segments > 0) ? num_ids / num_segments : 1;
  // grow-only arenas of the store (a hipMalloc / hipFree pair per temporary cost more than the reduce itself:
  // 8.3 ms per step at world size 1); ordered against other streams like the sampling arenas
  ArenaOrder arena_order(st, s);
  Carver cv;
  const size_t o_buck = cv.take((size_t)n1 * 8);
  const size_t o_ord = cv.take((size_t)n1 * 8);
  const size_t o_segb = cv.take((size_t)n1 * 4);
  int rc = st->r_req.ensure(cv.at);
  if (rc != GLX_OK) return rc;
  int64_t* buck = reinterpret_cast<int64_t*>(st->r_req.p + o_buck);
  int64_t* ord = reinterpret_cast<int64_t*>(st->r_req.p + o_ord);
  int32_t* seg_b = reinterpret_cast<int32_t*>(st->r_req.p + o_segb);
  rc = glx_partition(st->device, d_ids, n, P, buck, ord, st->d_vals, s);
  if (rc != GLX_OK) return rc;
  if (n > 0) {
    // stable inside a shard: a requester's segment ids stay non-decreasing on their way to every owner
    glx_dist_seg_of_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_seg, ord, n, fanout > 0 ? fanout : 1, seg_b);
  }
  // counts + this requester's segment count and default value (requests differ per rank)
  ReqParams mine;
  memset(&mine, 0, sizeof(mine));
  mine.v[0] = num_segments;
  mine.v[1] = op;
  memcpy(&mine.v[2], &default_attr, sizeof(float));
  constexpr int kParams = 3;
  glx_dist_set_params_kernel<<<1, 64, 0, s>>>(st->d_vals + P, mine, kParams);
  const int nvals = P + kParams;
  st->h_mat.resize((size_t)P * nvals);
  rc = exchange_counts(st, st->d_vals, nvals, st->h_mat.data(), s);
  if (rc != GLX_OK) return rc;
  Routing rt;
  routing_from_matrix(st, nvals, &rt);
  const int64_t m = rt.n_recv;
  GLX_REQUIRE(m <= INT32_MAX, "more than 2^31 ids arrived at one shard");
  std::vector<int64_t> sg_of((size_t)P), sg_off((size_t)P + 1, 0), back_counts((size_t)P), back_offs((size_t)P + 1, 0);
  for (int q = 0; q < P; ++q) {
    const int64_t* pq = &st->h_mat[(size_t)q * nvals + P];
    GLX_REQUIRE(pq[1] == op, "rank %d asks for aggregator %lld, this rank for %d: one collective, one operator", q,
                (long long)pq[1], op);
    sg_of[(size_t)q] = pq[0];
    sg_off[(size_t)q + 1] = sg_off[(size_t)q] + pq[0];
    back_counts[(size_t)q] = num_segments;  // every owner answers with one partial row per segment of MINE
    back_offs[(size_t)q + 1] = back_offs[(size_t)q] + num_segments;
  }
  const int64_t served = sg_off[(size_t)P];  // partial rows this owner produces, requester-major
  Carver cr, cb;
  const size_t o_ids = cr.take((size_t)(m > 0 ? m : 1) * 8);
  const size_t o_seg = cr.take((size_t)(m > 0 ? m : 1) * 4);
  const size_t o_part = cr.take((size_t)(served > 0 ? served : 1) * D * 4);
  const size_t o_cntp = cr.take((size_t)(served > 0 ? served : 1) * 4);
  const size_t o_pin = cb.take((size_t)P * (size_t)(num_segments > 0 ? num_segments : 1) * D * 4);
  const size_t o_cin = cb.take((size_t)P * (size_t)(num_segments > 0 ? num_segments : 1) * 4);
  rc = st->r_recv.ensure(cr.at);
  if (rc == GLX_OK) rc = st->r_back.ensure(cb.at);
  if (rc != GLX_OK) return rc;
  int64_t* ids_in = reinterpret_cast<int64_t*>(st->r_recv.p + o_ids);
  int32_t* seg_in = reinterpret_cast<int32_t*>(st->r_recv.p + o_seg);
  float* part_out = reinterpret_cast<float*>(st->r_recv.p + o_part);
  int32_t* cnt_part = reinterpret_cast<int32_t*>(st->r_recv.p + o_cntp);
  float* part_in = reinterpret_cast<float*>(st->r_back.p + o_pin);
  int32_t* cnt_in = reinterpret_cast<int32_t*>(st->r_back.p + o_cin);
  GlxSeg out_segs[2] = {{buck, ids_in, 8}, {seg_b, seg_in, 4}};
  rc = st->comm->alltoallv(out_segs, 2, rt.send_counts.data(), rt.send_offs.data(), rt.recv_counts.data(),
                           rt.recv_offs.data(), s);
  if (rc != GLX_OK) return rc;
  // Process on the owner: one local segmented reduce per requester (Aggregator::Aggregate on the ids it owns)
  for (int q = 0; q < P; ++q) {
    const int64_t* pq = &st->h_mat[(size_t)q * nvals + P];
    float dq;
    memcpy(&dq, &pq[2], sizeof(float));
    if (sg_of[(size_t)q] == 0) continue;
    rc = glx_aggregate(f, op, ids_in + rt.recv_offs[(size_t)q], seg_in + rt.recv_offs[(size_t)q],
                       (int32_t)rt.recv_counts[(size_t)q], (int32_t)sg_of[(size_t)q], dq, part_out + sg_off[(size_t)q] * D,
                       cnt_part + sg_off[(size_t)q], GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
  }
  // partial rows back: to requester q its sg_of[q] rows, from every owner my num_segments rows (shard-major)
  GlxSeg back_segs[2] = {{part_out, part_in, (size_t)D * 4}, {cnt_part, cnt_in, 4}};
  rc = st->comm->alltoallv(back_segs, 2, sg_of.data(), sg_off.data(), back_counts.data(), back_offs.data(), s);
  if (rc != GLX_OK) return rc;
  st->last_slot = 0;
  glx_dist_stats& stat = st->slots[0].stats;
  stat = glx_dist_stats{};
  stat.ids = n;
  stat.from_own_shard = rt.send_counts[(size_t)me];
  stat.remote = n - rt.send_counts[(size_t)me];
  stat.served_rows = served - sg_of[(size_t)me];  // partial rows reduced here for the other requesters
  stat.bytes_sent = (n - rt.send_counts[(size_t)me]) * 12 + (served - sg_of[(size_t)me]) * ((int64_t)D * 4 + 4);
  stat.bytes_received = (m - rt.recv_counts[(size_t)me]) * 12 + (int64_t)(P - 1) * num_segments * ((int64_t)D * 4 + 4);
  stat.exchange_rounds = st->comm->last_rounds;
  if (num_segments > 0) {
    rc = glx_aggregate_stitch(st->device, op, P, part_in, cnt_in, num_segments, D, default_attr, d_emb, d_cnt, s);
    if (rc != GLX_OK) return rc;
  }
  return GLX_OK;
}

}  // namespace

extern "C" int glx_dist_aggregate_partial(glx_dist_store* st, int op, const int64_t* node_ids, const int32_t* segment_ids,
                                          int32_t num_ids, int32_t num_segments, float default_attr, float* emb_out,
                                          int32_t* cnt_out, int ptr_kind, void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->feats != nullptr, "this store has no feature shard");
  GLX_REQUIRE(op >= GLX_AGG_SUM && op <= GLX_AGG_PROD, "unknown aggregator id %d", op);
  GLX_REQUIRE(num_ids >= 0 && num_segments >= 0, "negative sizes");
  GLX_REQUIRE((int64_t)num_segments * st->feats->dim <= INT32_MAX, "num_segments * dim exceeds int32 (tensor.h:47)");
  GLX_REQUIRE(num_segments == 0 || (emb_out && cnt_out), "NULL output pointer");
  GLX_REQUIRE(num_ids == 0 || node_ids, "NULL data pointer");
  GLX_REQUIRE(segment_ids != nullptr || num_segments == 0 || num_ids % num_segments == 0,
              "segment_ids == NULL means equal segments: num_ids must be a multiple of num_segments");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  if (ptr_kind == GLX_PTR_DEVICE) {
    return dist_aggregate_partial_device(st, op, node_ids, segment_ids, num_ids, num_segments, default_attr, emb_out,
                                         cnt_out, s);
  }
  const glx_features* f = st->feats;
  const size_t emb_n = (size_t)num_segments * f->dim;
  GlxTemp d_ids, d_seg, d_emb, d_cnt;
  GLX_HIP(hipMalloc(&d_ids.p, (size_t)(num_ids > 0 ? num_ids : 1) * 8));
  GLX_HIP(hipMalloc(&d_seg.p, (size_t)(num_ids > 0 ? num_ids : 1) * 4));
  GLX_HIP(hipMalloc(&d_emb.p, (emb_n > 0 ? emb_n : 1) * 4));
  GLX_HIP(hipMalloc(&d_cnt.p, (size_t)(num_segments > 0 ? num_segments : 1) * 4));
  if (num_ids) GLX_HIP(hipMemcpyAsync(d_ids.p, node_ids, (size_t)num_ids * 8, hipMemcpyHostToDevice, s));
  if (num_ids && segment_ids) GLX_HIP(hipMemcpyAsync(d_seg.p, segment_ids, (size_t)num_ids * 4, hipMemcpyHostToDevice, s));
  rc = dist_aggregate_partial_device(st, op, d_ids.as<int64_t>(), segment_ids ? d_seg.as<int32_t>() : nullptr, num_ids,
                                     num_segments, default_attr, d_emb.as<float>(), d_cnt.as<int32_t>(), s);
  if (rc != GLX_OK) return rc;
  if (num_segments > 0) {
    GLX_HIP(hipMemcpyAsync(emb_out, d_emb.p, emb_n * 4, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipMemcpyAsync(cnt_out, d_cnt.p, (size_t)num_segments * 4, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

// The two halves of glx_dist_aggregate for software pipelining (device pointers only).
extern "C" int glx_dist_aggregate_begin(glx_dist_store* st, int32_t slot, const int64_t* node_ids, int32_t num_ids,
                                        float default_attr, void* stream) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GLX_REQUIRE(st->feats != nullptr, "this store has no feature shard");
  GLX_REQUIRE(slot >= 0 && slot < GLX_DIST_SLOTS, "slot %d outside [0, %d)", slot, GLX_DIST_SLOTS);
  GLX_REQUIRE(num_ids >= 0 && (num_ids == 0 || node_ids), "bad request");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  glx_dist_store::Slot& sl = st->slots[slot];
  Resolved rs;
  int rc = resolve_and_fetch(st, slot, node_ids, num_ids, default_attr, glx_stream(stream), &rs);
  if (rc != GLX_OK) {
    sl.num_ids = -1;
    return rc;
  }
  sl.loc = rs.loc;
  for (int j = 0; j < 3; ++j) sl.src[j] = rs.src[j];
  sl.num_ids = num_ids;
  sl.default_attr = default_attr;
  return GLX_OK;
}

extern "C" int glx_dist_aggregate_end_range(glx_dist_store* st, int32_t slot, int32_t first_id, int32_t num_ids, int release,
                                            int op, const int32_t* segment_ids, int32_t num_segments, float* emb_out,
                                            int32_t* cnt_out, void* stream) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GLX_REQUIRE(slot >= 0 && slot < GLX_DIST_SLOTS, "slot %d outside [0, %d)", slot, GLX_DIST_SLOTS);
  GLX_REQUIRE(op >= GLX_AGG_SUM && op <= GLX_AGG_PROD, "unknown aggregator id %d", op);
  glx_dist_store::Slot& sl = st->slots[slot];
  GLX_REQUIRE(sl.num_ids >= 0, "slot %d holds no begun request", slot);
  GLX_REQUIRE(first_id >= 0 && num_ids >= 0 && (int64_t)first_id + num_ids <= sl.num_ids,
              "ids [%d, %d + %d) outside the begun request of %d ids", first_id, first_id, num_ids, sl.num_ids);
  GLX_REQUIRE(num_segments >= 0 && (int64_t)num_segments * st->feats->dim <= INT32_MAX, "bad num_segments");
  GLX_REQUIRE(num_segments == 0 || (emb_out && cnt_out), "NULL output pointer");
  GLX_REQUIRE(segment_ids != nullptr || num_segments == 0 || num_ids % num_segments == 0,
              "segment_ids == NULL means equal segments: num_ids must be a multiple of num_segments");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  if (release) sl.num_ids = -1;
  if (num_segments == 0) return GLX_OK;
  return glx_aggregate_vrows_device(sl.src, 3, st->feats->dim, op, sl.loc + first_id, segment_ids, num_ids, num_segments,
                                    sl.default_attr, emb_out, cnt_out, glx_stream(stream));
}

extern "C" int glx_dist_aggregate_end(glx_dist_store* st, int32_t slot, int op, const int32_t* segment_ids,
                                      int32_t num_segments, float* emb_out, int32_t* cnt_out, void* stream) {
  GLX_REQUIRE(st != nullptr, "store is NULL");
  GLX_REQUIRE(slot >= 0 && slot < GLX_DIST_SLOTS, "slot %d outside [0, %d)", slot, GLX_DIST_SLOTS);
  GLX_REQUIRE(st->slots[slot].num_ids >= 0, "slot %d holds no begun request", slot);
  return glx_dist_aggregate_end_range(st, slot, 0, st->slots[slot].num_ids, 1, op, segment_ids, num_segments, emb_out, cnt_out,
                                      stream);
}

extern "C" int glx_dist_lookup(glx_dist_store* st, const int64_t* node_ids, int64_t n, float default_attr,
                               float* out, int ptr_kind, void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->feats != nullptr, "this store has no feature shard");
  GLX_REQUIRE(n >= 0 && n < INT32_MAX, "bad n");
  GLX_REQUIRE(n == 0 || (node_ids && out), "NULL data pointer");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  const glx_features* f = st->feats;
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  const int64_t* d_ids = node_ids;
  float* d_out = out;
  GlxTemp stage;
  const size_t out_b = (size_t)n * f->dim * 4;
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMalloc(&stage.p, ((out_b + 255) & ~(size_t)255) + (size_t)n * 8 + 256));
    d_out = stage.as<float>();
    int64_t* ids_w = reinterpret_cast<int64_t*>(stage.as<char>() + ((out_b + 255) & ~(size_t)255));
    if (n) GLX_HIP(hipMemcpyAsync(ids_w, node_ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    d_ids = ids_w;
  }
  Resolved rs;
  rc = resolve_and_fetch(st, 0, d_ids, n, default_attr, s, &rs);
  st->slots[0].num_ids = -1;
  if (rc == GLX_OK && n > 0) {
    GatherArgs g;
    for (int j = 0; j < 3; ++j) g.src[j] = rs.src[j];
    g.base1 = (int32_t)rs.src[0].rows;
    g.base2 = (int32_t)(rs.src[0].rows + rs.src[1].rows);
    g.vrows = rs.loc;
    g.n = n;
    g.dim = f->dim;
    g.default_attr = default_attr;
    g.out = d_out;
    int G = 1;
    while (G < 64 && G < f->dim) G <<= 1;
    const int64_t threads = n * G;
    glx_dist_gather_rows_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(g, G);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
      glx_set_error("gather launch failed: %s", hipGetErrorString(le));
      rc = GLX_INTERNAL;
    }
  }
  if (ptr_kind == GLX_PTR_HOST) {
    hipError_t e = hipSuccess;
    if (rc == GLX_OK && n > 0) e = hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    if (rc != GLX_OK) return rc;
    GLX_HIP(e);
    GLX_HIP(e2);
  }
  return rc;
}

// Collective: every rank holds the same hot id list; each fetches its owned rows once and
// all ranks end up with the same [n, dim] replica (+ id map), in owner-major order.
extern "C" int glx_dist_store_set_cache(glx_dist_store* st, const int64_t* hot_ids, int64_t n, float default_attr,
                                        int ptr_kind, void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->feats != nullptr, "this store has no feature shard");
  GLX_REQUIRE(n >= 0 && n < INT32_MAX, "bad n");
  GLX_REQUIRE(n == 0 || hot_ids, "NULL data pointer");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  GLX_HIP(hipStreamSynchronize(s));
  if (st->cache) {
    glx_features_destroy(st->cache);
    st->cache = nullptr;
  }
  if (st->cache_slots) {
    GLX_HIP(hipFree(st->cache_slots));
    st->cache_slots = nullptr;
  }
  if (st->bm_member) {
    GLX_HIP(hipFree(st->bm_member));
    st->bm_member = nullptr;
    st->bm_valid = nullptr;
    st->bm_max = -1;
  }
  if (n == 0) return GLX_OK;
  const glx_features* f = st->feats;
  const int P = st->world, me = st->rank;
  const int32_t dim = f->dim;
  GlxTemp ids_d, bucketed, order, rows_mine, table, known_mine, known_all;
  const int64_t* d_hot = hot_ids;
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMalloc(&ids_d.p, (size_t)n * 8));
    GLX_HIP(hipMemcpyAsync(ids_d.p, hot_ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    d_hot = ids_d.as<int64_t>();
  }
  // the replica's rows are kept in ascending id order (an id's row is then its rank among the hot ids)
  GlxTemp sorted, sorted_masked, table_sorted;
  GLX_HIP(hipMalloc(&sorted.p, (size_t)n * 8));
#define SORTK(tmp, bytes) rocprim::radix_sort_keys(tmp, bytes, d_hot, sorted.as<int64_t>(), (size_t)n, 0, 64, s)
  GLX_ROCPRIM(SORTK);
#undef SORTK
  d_hot = sorted.as<int64_t>();
  GLX_HIP(hipMalloc(&bucketed.p, (size_t)n * 8));
  GLX_HIP(hipMalloc(&order.p, (size_t)n * 8));
  rc = glx_partition(st->device, d_hot, n, P, bucketed.as<int64_t>(), order.as<int64_t>(), st->d_vals, s);
  if (rc != GLX_OK) return rc;
  std::vector<int64_t> counts((size_t)P);
  GLX_HIP(hipMemcpyAsync(counts.data(), st->d_vals, (size_t)P * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  std::vector<int64_t> offs((size_t)P + 1, 0), same((size_t)P), zero((size_t)P, 0);
  for (int p = 0; p < P; ++p) offs[p + 1] = offs[p] + counts[p];
  for (int p = 0; p < P; ++p) same[p] = counts[me];
  const int64_t c_me = counts[me];
  GLX_HIP(hipMalloc(&rows_mine.p, (size_t)(c_me > 0 ? c_me : 1) * dim * 4));
  GLX_HIP(hipMalloc(&table.p, (size_t)n * dim * 4));
  GLX_HIP(hipMalloc(&known_mine.p, (size_t)(c_me > 0 ? c_me : 1) * 8));
  GLX_HIP(hipMalloc(&known_all.p, (size_t)n * 8));
  if (c_me > 0) {
    rc = glx_lookup(f, bucketed.as<int64_t>() + offs[me], c_me, default_attr, rows_mine.as<float>(), GLX_PTR_DEVICE,
                    s);
    if (rc != GLX_OK) return rc;
    glx_dist_known_kernel<<<(unsigned)((c_me + 255) / 256), 256, 0, s>>>(f->map(), bucketed.as<int64_t>() + offs[me],
                                                                         c_me, known_mine.as<int64_t>());
  }
  // all-gather(v) as an all-to-all whose every outgoing message is the same buffer
  GlxSeg segs[2] = {{rows_mine.p, table.p, (size_t)dim * 4}, {known_mine.p, known_all.p, 8}};
  rc = st->comm->alltoallv(segs, 2, same.data(), zero.data(), counts.data(), offs.data(), s);
  if (rc != GLX_OK) return rc;
  glx_dist_mask_ids_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(bucketed.as<int64_t>(), known_all.as<int64_t>(),
                                                                       n);
  GLX_HIP(hipGetLastError());
  // owner-major (the all-gather's layout) -> ascending ids: bucketed position pos came from sorted position order[pos]
  GLX_HIP(hipMalloc(&table_sorted.p, (size_t)n * dim * 4));
  GLX_HIP(hipMalloc(&sorted_masked.p, (size_t)n * 8));
  rc = glx_stitch_f32(st->device, table.as<float>(), order.as<int64_t>(), n, dim, table_sorted.as<float>(), s);
  if (rc == GLX_OK) {
    rc = glx_stitch_i64(st->device, bucketed.as<int64_t>(), order.as<int64_t>(), n, 1, sorted_masked.as<int64_t>(), s);
  }
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipStreamSynchronize(s));
  (void)hipFree(table.p);
  table.p = nullptr;
  rc = glx_features_create_impl(st->device, n, dim, table_sorted.as<float>(), sorted_masked.as<int64_t>(), GLX_PTR_DEVICE, s,
                                false, &st->cache);
  if (rc != GLX_OK) return rc;
  // from here on a failure must not leave a half-installed replica behind: st->cache set while cache_slots (or the
  // bitmap) is missing would have the next resolve kernel dereference null slots
  struct Rollback {
    glx_dist_store* st;
    bool armed = true;
    ~Rollback() {
      if (!armed) return;
      if (st->cache) glx_features_destroy(st->cache);
      st->cache = nullptr;
      if (st->bm_member) (void)hipFree(st->bm_member);
      st->bm_member = nullptr;
      st->bm_valid = nullptr;
      st->bm_max = -1;
      if (st->cache_slots) (void)hipFree(st->cache_slots);
      st->cache_slots = nullptr;
    }
  } rollback{st};
  // rank-select membership when the ids allow it
  st->bm_max = -1;
  int64_t lo_hi[2] = {0, 0};
  GLX_HIP(hipMemcpyAsync(&lo_hi[0], sorted.as<int64_t>(), 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipMemcpyAsync(&lo_hi[1], sorted.as<int64_t>() + (n - 1), 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  const bool no_bitmap = glx_side_knobs().dist_no_bitmap.load(std::memory_order_relaxed) > 0;  // A/B and test knob
  // ... and are dense enough: the bitmap costs 24 bytes per 64 ids of the RANGE, persistent, plus scratch -- one stray
  // id near 2^31 in a short hot list would pin ~800 MB per GPU.  Within 4 words per listed id (+ a floor) it never
  // exceeds ~100 bytes per hot row, a tenth of the row itself at dim 256; sparser lists keep the hash map.
  const int64_t words_needed = lo_hi[1] >= 0 ? (lo_hi[1] >> 6) + 1 : 0;
  if (!no_bitmap && lo_hi[0] >= 0 && lo_hi[1] < ((int64_t)1 << 31) && words_needed <= 4 * n + 4096) {
    const int64_t words = words_needed;
    const size_t wb = ((size_t)words * 8 + 255) & ~(size_t)255;
    // kept: RankWord[words] | valid[words]; scratch: member[words] | rank[words] | flags
    GlxTemp bm_owner;  // released on every early return; handed to the store only when complete
    GlxTemp tmp_bm;
    GLX_HIP(hipMalloc(&bm_owner.p, (size_t)words * sizeof(RankWord) + wb + 256));
    char* bm = bm_owner.as<char>();
    GLX_HIP(hipMalloc(&tmp_bm.p, wb + (size_t)words * 4 + 256));
    GLX_HIP(hipMemsetAsync(bm, 0, (size_t)words * sizeof(RankWord) + wb + 256, s));
    GLX_HIP(hipMemsetAsync(tmp_bm.p, 0, wb + (size_t)words * 4 + 256, s));
    RankWord* packed = reinterpret_cast<RankWord*>(bm);
    uint64_t* valid = reinterpret_cast<uint64_t*>(bm + (size_t)words * sizeof(RankWord));
    uint64_t* member = reinterpret_cast<uint64_t*>(tmp_bm.as<char>());
    uint32_t* rank = reinterpret_cast<uint32_t*>(tmp_bm.as<char>() + wb);
    int* flags = reinterpret_cast<int*>(tmp_bm.as<char>() + wb + (size_t)words * 4);
    glx_dist_bitmap_set_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(
        sorted.as<int64_t>(), sorted_masked.as<int64_t>(), n, reinterpret_cast<unsigned long long*>(member),
        reinterpret_cast<unsigned long long*>(valid), flags);
    glx_dist_bitmap_popc_kernel<<<(unsigned)((words + 255) / 256), 256, 0, s>>>(member, words, rank);
#define SCANP(tmp, bytes) rocprim::exclusive_scan(tmp, bytes, rank, rank, 0u, (size_t)words, rocprim::plus<uint32_t>(), s)
    GLX_ROCPRIM(SCANP);
#undef SCANP
    glx_dist_bitmap_pack_kernel<<<(unsigned)((words + 255) / 256), 256, 0, s>>>(member, rank, words, packed);
    int h_flags[2] = {0, 0};
    GLX_HIP(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
    if (h_flags[0]) {
      // an id listed twice: ranks are not rows, keep the hash map (bm_owner frees the bitmap)
    } else {
      bm_owner.p = nullptr;
      st->bm_member = packed;
      st->bm_valid = h_flags[1] ? valid : nullptr;
      st->bm_max = lo_hi[1];
    }
  }
  const uint64_t cap = st->cache->idmap.cap;
  GLX_HIP(hipMalloc(reinterpret_cast<void**>(&st->cache_slots), (size_t)cap * sizeof(PackedSlot)));
  glx_dist_pack_map_kernel<<<grid_for((int64_t)cap), 256, 0, s>>>(st->cache->idmap.keys, st->cache->idmap.vals, cap,
                                                                 st->cache_slots);
  GLX_HIP(hipGetLastError());
  GLX_HIP(hipStreamSynchronize(s));
  rollback.armed = false;
  return GLX_OK;
}

namespace {

// Every destination id of the edge type with its in-degree summed over ALL shards, at the id's owner
// (llabs(id) % P): each shard run-length-encodes its own destinations, the (id, count) pairs travel to the
// owners, the owners reduce by key.  Collective.
struct DstTotals {
  GlxTemp uniq, cnt;       // this shard's distinct destination ids (ascending) + its own counts
  int64_t Ul = 0;
  GlxTemp buck, ord;       // the same ids bucketed by owner + their index in `uniq`
  Routing rt;              // uniq ids out / in
  GlxTemp ids_in;          // ids the other shards asked about (requester-major), rt.n_recv of them
  GlxTemp own_id, own_cnt; // owned ids ascending + global in-degree
  int64_t M = 0;
};

int dst_totals(glx_dist_store* st, hipStream_t s, DstTotals* t) {
  const glx_graph* g = st->graph;
  const int P = st->world;
  int rc = glx_graph_dst_counts(g, &t->uniq, &t->cnt, &t->Ul, s);
  if (rc != GLX_OK) return rc;
  const int64_t Ul = t->Ul;
  GlxTemp cnt_b;
  GLX_HIP(hipMalloc(&t->buck.p, (size_t)(Ul ? Ul : 1) * 8));
  GLX_HIP(hipMalloc(&t->ord.p, (size_t)(Ul ? Ul : 1) * 8));
  GLX_HIP(hipMalloc(&cnt_b.p, (size_t)(Ul ? Ul : 1) * 8));
  rc = glx_partition(st->device, t->uniq.as<int64_t>(), Ul, P, t->buck.as<int64_t>(), t->ord.as<int64_t>(), st->d_vals, s);
  if (rc != GLX_OK) return rc;
  if (Ul > 0) {
    glx_dist_gather_i64_kernel<<<(unsigned)((Ul + 255) / 256), 256, 0, s>>>(t->cnt.as<int64_t>(), t->ord.as<int64_t>(), Ul,
                                                                            cnt_b.as<int64_t>());
  }
  st->h_mat.resize((size_t)P * P);
  rc = exchange_counts(st, st->d_vals, P, st->h_mat.data(), s);
  if (rc != GLX_OK) return rc;
  routing_from_matrix(st, P, &t->rt);
  const size_t m = (size_t)t->rt.n_recv;
  GlxTemp cnt_in, ids_s, cnt_s, nown;
  GLX_HIP(hipMalloc(&t->ids_in.p, (m ? m : 1) * 8));
  GLX_HIP(hipMalloc(&cnt_in.p, (m ? m : 1) * 8));
  GlxSeg segs[2] = {{t->buck.p, t->ids_in.p, 8}, {cnt_b.p, cnt_in.p, 8}};
  rc = st->comm->alltoallv(segs, 2, t->rt.send_counts.data(), t->rt.send_offs.data(), t->rt.recv_counts.data(),
                           t->rt.recv_offs.data(), s);
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipMalloc(&ids_s.p, (m ? m : 1) * 8));
  GLX_HIP(hipMalloc(&cnt_s.p, (m ? m : 1) * 8));
  GLX_HIP(hipMalloc(&t->own_id.p, (m ? m : 1) * 8));
  GLX_HIP(hipMalloc(&t->own_cnt.p, (m ? m : 1) * 8));
  GLX_HIP(hipMalloc(&nown.p, 8));
  t->M = 0;
  if (m > 0) {
#define SORTP(tmp, bytes)                                                                                       \
  rocprim::radix_sort_pairs(tmp, bytes, t->ids_in.as<int64_t>(), ids_s.as<int64_t>(), cnt_in.as<int64_t>(), \
                            cnt_s.as<int64_t>(), m, 0, 64, s)
    GLX_ROCPRIM(SORTP);
#undef SORTP
#define REDUCE(tmp, bytes)                                                                                        \
  rocprim::reduce_by_key(tmp, bytes, ids_s.as<int64_t>(), cnt_s.as<int64_t>(), m, t->own_id.as<int64_t>(),    \
                         t->own_cnt.as<int64_t>(), nown.as<int64_t>(), rocprim::plus<int64_t>(),                \
                         rocprim::equal_to<int64_t>(), s)
    GLX_ROCPRIM(REDUCE);
#undef REDUCE
    GLX_HIP(hipMemcpyAsync(&t->M, nown.p, 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  return GLX_OK;
}

// out[i] = total of ids[i] in the owner's ascending (own_id, own_cnt) table; every id asked about is there.
__global__ void glx_dist_total_of_kernel(const int64_t* __restrict__ own_id, const int64_t* __restrict__ own_cnt,
                                         int64_t M, const int64_t* __restrict__ ids, int64_t n,
                                         int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  int64_t lo = 0, hi = M;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (own_id[mid] < id) lo = mid + 1; else hi = mid;
  }
  out[i] = (lo < M && own_id[lo] == id) ? own_cnt[lo] : 0;
}

__global__ void glx_dist_scatter_i64_kernel(const int64_t* __restrict__ in, const int64_t* __restrict__ order, int64_t n,
                                            int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[order[i]] = in[i];
}

}  // namespace

// Collective: global in-degree top-`want` of the store's edge type.
extern "C" int glx_dist_hot_ids(glx_dist_store* st, int64_t want, int64_t* ids_out, int64_t* n_out, void* stream) {
  GLX_REQUIRE(st != nullptr && ids_out != nullptr && n_out != nullptr, "NULL argument");
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(want >= 0 && want < INT32_MAX, "bad want");
  *n_out = 0;
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = glx_stream(stream);
  const int P = st->world;
  DstTotals t;
  int rc = dst_totals(st, s, &t);
  if (rc != GLX_OK) return rc;
  const int64_t M = t.M;
  // this owner's best `want` (count descending; ids ascending among equals: the input is id-sorted and the
  // radix sort is stable)
  GlxTemp top_cnt, top_id;
  GLX_HIP(hipMalloc(&top_cnt.p, (size_t)(M ? M : 1) * 8));
  GLX_HIP(hipMalloc(&top_id.p, (size_t)(M ? M : 1) * 8));
  if (M > 0) {
    const size_t Ms = (size_t)M;
#define SORTD(tmp, bytes)                                                                                             \
  rocprim::radix_sort_pairs_desc(tmp, bytes, t.own_cnt.as<int64_t>(), top_cnt.as<int64_t>(), t.own_id.as<int64_t>(), \
                                 top_id.as<int64_t>(), Ms, 0, 64, s)
    GLX_ROCPRIM(SORTD);
#undef SORTD
  }
  const int64_t c_me = M < want ? M : want;
  // share the candidates: everyone gets every owner's list (rank-major) and merges the same way
  int64_t c_me_copy = c_me;
  GLX_HIP(hipMemcpyAsync(st->d_vals, &c_me_copy, 8, hipMemcpyHostToDevice, s));
  std::vector<int64_t> cand((size_t)P);
  rc = exchange_counts(st, st->d_vals, 1, cand.data(), s);
  if (rc != GLX_OK) return rc;
  std::vector<int64_t> offs((size_t)P + 1, 0), same((size_t)P, c_me), zero((size_t)P, 0);
  for (int p = 0; p < P; ++p) offs[p + 1] = offs[p] + cand[p];
  const size_t T = (size_t)offs[P];
  GlxTemp all_cnt, all_id, by_id_cnt, by_id_id, fin_cnt, fin_id;
  GLX_HIP(hipMalloc(&all_cnt.p, (T ? T : 1) * 8));
  GLX_HIP(hipMalloc(&all_id.p, (T ? T : 1) * 8));
  GlxSeg csegs[2] = {{top_cnt.p, all_cnt.p, 8}, {top_id.p, all_id.p, 8}};
  rc = st->comm->alltoallv(csegs, 2, same.data(), zero.data(), cand.data(), offs.data(), s);
  if (rc != GLX_OK) return rc;
  if (T == 0) return GLX_OK;
  GLX_HIP(hipMalloc(&by_id_cnt.p, T * 8));
  GLX_HIP(hipMalloc(&by_id_id.p, T * 8));
  GLX_HIP(hipMalloc(&fin_cnt.p, T * 8));
  GLX_HIP(hipMalloc(&fin_id.p, T * 8));
#define SORT1(tmp, bytes)                                                                                     \
  rocprim::radix_sort_pairs(tmp, bytes, all_id.as<int64_t>(), by_id_id.as<int64_t>(), all_cnt.as<int64_t>(), \
                            by_id_cnt.as<int64_t>(), T, 0, 64, s)
  GLX_ROCPRIM(SORT1);
#undef SORT1
#define SORT2(tmp, bytes)                                                                                            \
  rocprim::radix_sort_pairs_desc(tmp, bytes, by_id_cnt.as<int64_t>(), fin_cnt.as<int64_t>(), by_id_id.as<int64_t>(), \
                                 fin_id.as<int64_t>(), T, 0, 64, s)
  GLX_ROCPRIM(SORT2);
#undef SORT2
  const int64_t take = (int64_t)T < want ? (int64_t)T : want;
  GLX_HIP(hipMemcpyAsync(ids_out, fin_id.p, (size_t)take * 8, hipMemcpyDeviceToHost, s));
  GLX_HIP(hipStreamSynchronize(s));
  *n_out = take;
  return GLX_OK;
}

namespace {
// Collective on first use: every rank ends up holding the (id ascending, global in-degree) table of the destination
// ids it owns -- the per-owner half of the unpartitioned storage's GetAllDstIds() / GetAllInDegrees()
// (topo_statics.cc:32-69).
int ensure_owner_table(glx_dist_store* st, hipStream_t s) {
  if (st->own_M >= 0) return GLX_OK;
  DstTotals t;
  int rc = dst_totals(st, s, &t);
  if (rc != GLX_OK) return rc;
  st->own_id = t.own_id.as<int64_t>();
  st->own_cnt = t.own_cnt.as<int64_t>();
  t.own_id.p = nullptr;  // ownership moves to the store
  t.own_cnt.p = nullptr;
  st->own_M = t.M;
  return GLX_OK;
}

__global__ void glx_dist_stitch_narrow_kernel(const int64_t* __restrict__ in, const int64_t* __restrict__ order, int64_t n,
                                              int32_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[order[i]] = (int32_t)in[i];
}

__global__ void glx_dist_i64_to_f32_kernel(const int64_t* __restrict__ in, int64_t n, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
}  // namespace

// Collective.  GetDegree for DESTINATION ids across the shards (degree_getter.cc with node_from = kEdgeDst;
// GraphStorage::GetInDegree, topo_statics.cc:62-69): the in-degree of an id is the sum over ALL shards of the edges
// that point to it, held by the id's owner (llabs(id) % P); ids nobody points to answer 0.
extern "C" int glx_dist_in_degrees(glx_dist_store* st, const int64_t* ids, int32_t n, int32_t* degrees_out, int ptr_kind,
                                   void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(n >= 0 && (n == 0 || (ids && degrees_out)), "bad request");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  rc = ensure_owner_table(st, s);
  if (rc != GLX_OK) return rc;
  const int P = st->world;
  const int64_t n1 = n > 0 ? n : 1;
  GlxTemp d_ids, buck, ord, ids_in, cnt_in, cnt_b, d_out;
  const int64_t* p_ids = ids;
  int32_t* p_out = degrees_out;
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMalloc(&d_ids.p, (size_t)n1 * 8));
    GLX_HIP(hipMalloc(&d_out.p, (size_t)n1 * 4));
    if (n > 0) GLX_HIP(hipMemcpyAsync(d_ids.p, ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    p_ids = d_ids.as<int64_t>();
    p_out = d_out.as<int32_t>();
  }
  GLX_HIP(hipMalloc(&buck.p, (size_t)n1 * 8));
  GLX_HIP(hipMalloc(&ord.p, (size_t)n1 * 8));
  GLX_HIP(hipMalloc(&cnt_b.p, (size_t)n1 * 8));
  rc = glx_partition(st->device, p_ids, n, P, buck.as<int64_t>(), ord.as<int64_t>(), st->d_vals, s);
  if (rc != GLX_OK) return rc;
  st->h_mat.resize((size_t)P * P);
  rc = exchange_counts(st, st->d_vals, P, st->h_mat.data(), s);
  if (rc != GLX_OK) return rc;
  Routing rt;
  routing_from_matrix(st, P, &rt);
  const int64_t m = rt.n_recv, m1 = m > 0 ? m : 1;
  GLX_HIP(hipMalloc(&ids_in.p, (size_t)m1 * 8));
  GLX_HIP(hipMalloc(&cnt_in.p, (size_t)m1 * 8));
  GlxSeg out_seg{buck.p, ids_in.p, 8};
  rc = st->comm->alltoallv(&out_seg, 1, rt.send_counts.data(), rt.send_offs.data(), rt.recv_counts.data(),
                           rt.recv_offs.data(), s);
  if (rc != GLX_OK) return rc;
  if (m > 0) {
    glx_dist_total_of_kernel<<<(unsigned)((m + 255) / 256), 256, 0, s>>>(st->own_id, st->own_cnt, st->own_M,
                                                                         ids_in.as<int64_t>(), m, cnt_in.as<int64_t>());
  }
  GlxSeg back_seg{cnt_in.p, cnt_b.p, 8};
  rc = st->comm->alltoallv(&back_seg, 1, rt.recv_counts.data(), rt.recv_offs.data(), rt.send_counts.data(),
                           rt.send_offs.data(), s);
  if (rc != GLX_OK) return rc;
  if (n > 0) {
    glx_dist_stitch_narrow_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(cnt_b.as<int64_t>(), ord.as<int64_t>(), n, p_out);
    GLX_HIP(hipGetLastError());
    if (ptr_kind == GLX_PTR_HOST) GLX_HIP(hipMemcpyAsync(degrees_out, p_out, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

// Collective.  The candidate list of the negative samplers over the WHOLE edge type, on every rank
// (random_negative_sampler.cc:30-63 / in_degree_negative_sampler.cc:29-135 draw from the storage's GetAllDstIds() /
// GetAllInDegrees(); a shard's own storage only knows the destinations of its own edges): the owners' (id, global
// in-degree) tables are gathered and merged, ids ascending -- the one order every shard count agrees on (a single
// store's first-appearance order depends on edge insertion order, which shards do not share) -- and one alias table is
// built over it.  The same table on every rank: glx_negative_sample on it draws exactly what an unpartitioned store
// would draw from glx_negative_create(ids ascending, in-degrees).
extern "C" int glx_dist_negative_create(glx_dist_store* st, int by_in_degree, void* stream, glx_negative** out) {
  GLX_REQUIRE(st != nullptr && out != nullptr, "NULL argument");
  *out = nullptr;
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = glx_stream(stream);
  int rc = ensure_owner_table(st, s);
  if (rc != GLX_OK) return rc;
  const int P = st->world;
  const int64_t M = st->own_M;
  GLX_HIP(hipMemcpyAsync(st->d_vals, &M, 8, hipMemcpyHostToDevice, s));
  std::vector<int64_t> sizes((size_t)P);
  rc = exchange_counts(st, st->d_vals, 1, sizes.data(), s);
  if (rc != GLX_OK) return rc;
  std::vector<int64_t> offs((size_t)P + 1, 0), same((size_t)P, M), zero((size_t)P, 0);
  for (int q = 0; q < P; ++q) offs[(size_t)q + 1] = offs[(size_t)q] + sizes[(size_t)q];
  const size_t T = (size_t)offs[(size_t)P];
  GLX_REQUIRE(T < (size_t)INT32_MAX, "more than 2^31 candidates");
  GlxTemp all_id, all_cnt, s_id, s_cnt, w;
  GLX_HIP(hipMalloc(&all_id.p, (T ? T : 1) * 8));
  GLX_HIP(hipMalloc(&all_cnt.p, (T ? T : 1) * 8));
  GlxSeg segs[2] = {{st->own_id, all_id.p, 8}, {st->own_cnt, all_cnt.p, 8}};
  rc = st->comm->alltoallv(segs, 2, same.data(), zero.data(), sizes.data(), offs.data(), s);  // everyone gets every table
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipMalloc(&s_id.p, (T ? T : 1) * 8));
  GLX_HIP(hipMalloc(&s_cnt.p, (T ? T : 1) * 8));
  GLX_HIP(hipMalloc(&w.p, (T ? T : 1) * 4));
  if (T > 0) {
    // ascending as signed ids (rocPRIM orders signed keys as such)
#define SORTN(tmp, bytes)                                                                                       \
  rocprim::radix_sort_pairs(tmp, bytes, all_id.as<int64_t>(), s_id.as<int64_t>(), all_cnt.as<int64_t>(),        \
                            s_cnt.as<int64_t>(), T, 0, 64, s)
    GLX_ROCPRIM(SORTN);
#undef SORTN
    glx_dist_i64_to_f32_kernel<<<(unsigned)((T + 255) / 256), 256, 0, s>>>(s_cnt.as<int64_t>(), (int64_t)T, w.as<float>());
    GLX_HIP(hipGetLastError());
  }
  rc = glx_negative_create(st->device, (int64_t)T, s_id.as<int64_t>(), by_in_degree ? w.as<float>() : nullptr, GLX_PTR_DEVICE,
                           s, out);
  if (rc != GLX_OK) return rc;
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

// Collective for GLX_NEG_EXCLUDE_NEIGHBORS (InDegreeNegativeSampler, in_degree_negative_sampler.cc:29-135 behind
// DistributeRunner): the exclusion set of a row is its source vertex's adjacency, which lives with the row's owner -- the
// rows travel there (with their index in the request, the random stream they draw from), the owner samples from the
// SAME candidate table every rank holds (glx_dist_negative_create) and the answers travel back.  The other modes need
// nothing from another shard and run locally: what the caller gets is what an unpartitioned store with the same table
// answers, for every shard count.  The store's graph shard needs glx_graph_enable_negative().
extern "C" int glx_dist_negative_sample(glx_dist_store* st, const glx_negative* table, int exclude, const int64_t* src,
                                        int32_t batch, int32_t count, int64_t default_neighbor_id, uint64_t seed,
                                        uint64_t call_counter, int64_t* out, int ptr_kind, void* stream) {
  int rc = check_store(st, ptr_kind);
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(table != nullptr, "table is NULL");
  GLX_REQUIRE(exclude >= GLX_NEG_EXCLUDE_NONE && exclude <= GLX_NEG_EXCLUDE_BATCH, "unknown exclusion mode %d", exclude);
  GLX_REQUIRE(batch >= 0 && count >= 0 && (int64_t)batch * count <= INT32_MAX, "bad batch / count");
  if (exclude != GLX_NEG_EXCLUDE_NEIGHBORS || (st->world == 1 && st->shortcut)) {
    return glx_negative_sample(table, exclude, st->graph, src, batch, count, default_neighbor_id, seed, call_counter, out,
                               ptr_kind, stream);
  }
  GLX_REQUIRE(st->graph != nullptr, "this store has no graph shard");
  GLX_REQUIRE(batch == 0 || count == 0 || (src && out), "NULL data pointer");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, st->device) : glx_stream(stream);
  const int P = st->world;
  const int64_t n = batch, n1 = n > 0 ? n : 1, k = count, k1 = k > 0 ? k : 1;
  GlxTemp d_src, d_out, buck, ord, back, ids_in, rows_in, loc;
  const int64_t* p_src = src;
  int64_t* p_out = out;
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMalloc(&d_src.p, (size_t)n1 * 8));
    GLX_HIP(hipMalloc(&d_out.p, (size_t)n1 * k1 * 8));
    if (n > 0) GLX_HIP(hipMemcpyAsync(d_src.p, src, (size_t)n * 8, hipMemcpyHostToDevice, s));
    p_src = d_src.as<int64_t>();
    p_out = d_out.as<int64_t>();
  }
  GLX_HIP(hipMalloc(&buck.p, (size_t)n1 * 8));
  GLX_HIP(hipMalloc(&ord.p, (size_t)n1 * 8));
  GLX_HIP(hipMalloc(&back.p, (size_t)n1 * k1 * 8));
  rc = glx_partition(st->device, p_src, n, P, buck.as<int64_t>(), ord.as<int64_t>(), st->d_vals, s);
  if (rc != GLX_OK) return rc;
  constexpr int kNegParams = 4;
  ReqParams mine;
  memset(&mine, 0, sizeof(mine));
  mine.v[0] = (int64_t)seed;
  mine.v[1] = (int64_t)call_counter;
  mine.v[2] = count;
  mine.v[3] = default_neighbor_id;
  glx_dist_set_params_kernel<<<1, 64, 0, s>>>(st->d_vals + P, mine, kNegParams);
  const int nvals = P + kNegParams;
  st->h_mat.resize((size_t)P * nvals);
  rc = exchange_counts(st, st->d_vals, nvals, st->h_mat.data(), s);
  if (rc != GLX_OK) return rc;
  Routing rt;
  routing_from_matrix(st, nvals, &rt);
  for (int q = 0; q < P; ++q) {
    GLX_REQUIRE(st->h_mat[(size_t)q * nvals + P + 2] == count, "rank %d asks for %lld negatives per row, this rank for %d", q,
                (long long)st->h_mat[(size_t)q * nvals + P + 2], count);
  }
  const int64_t m = rt.n_recv, m1 = m > 0 ? m : 1;
  GLX_REQUIRE(m * k <= INT32_MAX, "more than 2^31 answers at one shard");
  GLX_HIP(hipMalloc(&ids_in.p, (size_t)m1 * 8));
  GLX_HIP(hipMalloc(&rows_in.p, (size_t)m1 * 8));
  GLX_HIP(hipMalloc(&loc.p, (size_t)m1 * k1 * 8));
  GlxSeg out_segs[2] = {{buck.p, ids_in.p, 8}, {ord.p, rows_in.p, 8}};
  rc = st->comm->alltoallv(out_segs, 2, rt.send_counts.data(), rt.send_offs.data(), rt.recv_counts.data(),
                           rt.recv_offs.data(), s);
  if (rc != GLX_OK) return rc;
  for (int q = 0; q < P && k > 0; ++q) {  // every requester's rows with ITS seed / call counter / default id
    const int64_t begin = rt.recv_offs[q], cnt = rt.recv_offs[q + 1] - begin;
    if (cnt == 0) continue;
    const int64_t* pq = &st->h_mat[(size_t)q * nvals + P];
    rc = glx_negative_sample_rows_device(table, st->graph, ids_in.as<int64_t>() + begin, rows_in.as<int64_t>() + begin,
                                         (int32_t)cnt, count, pq[3], (uint64_t)pq[0], (uint64_t)pq[1],
                                         loc.as<int64_t>() + begin * k, s);
    if (rc != GLX_OK) return rc;
  }
  GlxSeg back_seg{loc.p, back.p, (size_t)k1 * 8};
  if (k > 0) {
    rc = st->comm->alltoallv(&back_seg, 1, rt.recv_counts.data(), rt.recv_offs.data(), rt.send_counts.data(),
                             rt.send_offs.data(), s);
    if (rc != GLX_OK) return rc;
  }
  if (n > 0 && k > 0) {
    const int64_t total = n * k;
    glx_dist_stitch2_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(back.as<int64_t>(), back.as<int64_t>(),
                                                                            ord.as<int64_t>(), n, count, p_out, p_out);
    GLX_HIP(hipGetLastError());
    if (ptr_kind == GLX_PTR_HOST) GLX_HIP(hipMemcpyAsync(out, p_out, (size_t)total * 8, hipMemcpyDeviceToHost, s));
  }
  GLX_HIP(hipStreamSynchronize(s));
  return GLX_OK;
}

// Collective: InDegreeSampler on a partitioned edge type.  A neighbour's weight is its in-degree over ALL
// shards (GraphStorage::GetInDegree of the unpartitioned storage, topo_statics.cc:62-69; the sampler:
// in_degree_sampler.cc:33-114), so every shard asks the owners for the totals of the destinations it holds
// and builds its per-row alias tables from those.
extern "C" int glx_dist_enable_in_degree(glx_dist_store* st, glx_graph* shard, void* stream) {
  GLX_REQUIRE(st != nullptr && shard != nullptr, "NULL argument");
  GLX_REQUIRE(st->graph == shard, "`shard` must be the graph this store was created with");
  GlxDeviceGuard guard(st->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", st->device);
  hipStream_t s = glx_stream(stream);
  DstTotals t;
  int rc = dst_totals(st, s, &t);
  if (rc != GLX_OK) return rc;
  const int64_t m = t.rt.n_recv, Ul = t.Ul;
  GlxTemp tot_in, tot_b, total;
  GLX_HIP(hipMalloc(&tot_in.p, (size_t)(m ? m : 1) * 8));
  GLX_HIP(hipMalloc(&tot_b.p, (size_t)(Ul ? Ul : 1) * 8));
  GLX_HIP(hipMalloc(&total.p, (size_t)(Ul ? Ul : 1) * 8));
  if (m > 0) {
    glx_dist_total_of_kernel<<<(unsigned)((m + 255) / 256), 256, 0, s>>>(t.own_id.as<int64_t>(), t.own_cnt.as<int64_t>(),
                                                                         t.M, t.ids_in.as<int64_t>(), m,
                                                                         tot_in.as<int64_t>());
  }
  GlxSeg seg{tot_in.p, tot_b.p, 8};
  rc = st->comm->alltoallv(&seg, 1, t.rt.recv_counts.data(), t.rt.recv_offs.data(), t.rt.send_counts.data(),
                           t.rt.send_offs.data(), s);
  if (rc != GLX_OK) return rc;
  if (Ul > 0) {
    glx_dist_scatter_i64_kernel<<<(unsigned)((Ul + 255) / 256), 256, 0, s>>>(tot_b.as<int64_t>(), t.ord.as<int64_t>(), Ul,
                                                                             total.as<int64_t>());
  }
  GLX_HIP(hipGetLastError());
  rc = glx_graph_install_in_degree(shard, t.uniq.as<int64_t>(), total.as<int64_t>(), Ul, s);
  if (rc != GLX_OK) return rc;
  shard->indeg_global = true;
  return GLX_OK;
}
