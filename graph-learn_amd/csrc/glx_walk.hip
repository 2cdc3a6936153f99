// glx: the "RandomWalk" operator (core/operator/random_walk/random_walk.cc:30-276).
//   p = q = 1 (RandomWalkRequest::IsDeepWalk, random_walk_request.cc:152-160): every step is one
//     uniform neighbour draw over the whole row (DeepWalk, :168-190) -- the RandomSampler kernel
//     with neighbor_count 1.
//   otherwise node2vec (WeightedRandomWalk / WeightedRandomWalkKernel, :192-272): a step looks at
//     the first min(deg, DefaultFullNbrNum) neighbours of the current vertex, weighs each edge
//     weight by 1/(p + 1e-6) when it leads back to the parent, by 1 when it leads to one of the
//     parent's first DefaultFullNbrNum neighbours, by 1/(q + 1e-6) otherwise, builds the alias
//     table of those weights (AliasMethod, alias_method.cc:57-107) and draws once (:109-124).
// All walk_len steps run inside one call; the walks never leave HBM between steps.
#include <math.h>

#include "glx_common.h"

namespace {

struct WalkArgs {
  GlxIdMap map;
  const int64_t* row_ptr;
  const GlxAdj* adj;
  const float* weight;  // per slot, or nullptr: default_weight
  const int64_t* seeds;
  int64_t* walks;  // [batch, walk_len]
  int64_t default_nbr;
  uint64_t seed, cc;
  float p, q, default_weight;
  int32_t batch, walk_len, step, full_nbr_num;
  // node2vec, steps >= 1 -- the reference hands a step the parents' neighbour lists as ONE concatenated array and walks
  // it with a cursor that it only advances for walkers whose current vertex has out-edges (random_walk.cc:214-226):
  // behind a stuck walker every later walker reads a window that starts too early by that walker's list length.
  // seg[i] = length of walker i's parent list, off_true[i] = where it really starts, off_used[i] = where the reference's
  // cursor stands when it reaches walker i.  Equal offsets: the walker's own parent list (the common case).
  const int32_t* seg;
  const int64_t* off_true;
  const int64_t* off_used;
};

__device__ __forceinline__ int64_t walk_parent_of(const WalkArgs& a, int64_t i) {
  // the first step's parent is the seed itself, without neighbours (random_walk_request.cc:120-131)
  return a.step <= 1 ? a.seeds[i] : a.walks[i * a.walk_len + a.step - 2];
}

// seg / seg-if-not-stuck of every walker for step a.step >= 1 (one thread per walker)
__global__ void glx_walk_segments_kernel(WalkArgs a, int32_t* __restrict__ seg, int32_t* __restrict__ seg_live) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= a.batch) return;
  const int64_t cur = a.walks[i * a.walk_len + a.step - 1];
  const int64_t crow = glx_row_of(a.map, cur);
  const bool live = crow >= 0 && a.row_ptr[crow + 1] > a.row_ptr[crow];
  const int64_t prow = glx_row_of(a.map, walk_parent_of(a, i));
  int32_t n = 0;
  if (prow >= 0) {
    const int64_t d = a.row_ptr[prow + 1] - a.row_ptr[prow];
    n = (int32_t)(d < a.full_nbr_num ? d : a.full_nbr_num);
  }
  seg[i] = n;
  seg_live[i] = live ? n : 0;
}

// exclusive prefix sums of two int32 arrays into int64 ones; one workgroup (walk batches are small next to a graph)
__global__ __launch_bounds__(1024) void glx_walk_scan2_kernel(const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
                                                             int64_t n, int64_t* __restrict__ o0, int64_t* __restrict__ o1) {
  __shared__ int64_t part[2][1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024, lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
  int64_t s0 = 0, s1 = 0;
  for (int64_t i = lo; i < hi; ++i) { s0 += a0[i]; s1 += a1[i]; }
  part[0][t] = s0;
  part[1][t] = s1;
  __syncthreads();
  if (t == 0) {
    int64_t r0 = 0, r1 = 0;
    for (int k = 0; k < 1024; ++k) {
      const int64_t v0 = part[0][k], v1 = part[1][k];
      part[0][k] = r0; part[1][k] = r1;
      r0 += v0; r1 += v1;
    }
  }
  __syncthreads();
  s0 = part[0][t];
  s1 = part[1][t];
  for (int64_t i = lo; i < hi; ++i) {
    o0[i] = s0; o1[i] = s1;
    s0 += a0[i]; s1 += a1[i];
  }
}

// One wave per walker.  LDS: parent's neighbour ids [F] i64 | weights [F] f32 | table [F] 8 B |
// stack pairs [F] 8 B = 28 F bytes (F <= 2048 keeps it under the 64 KiB a launch may ask for).
__global__ __launch_bounds__(64) void glx_node2vec_step_kernel(WalkArgs a) {
  extern __shared__ int64_t lds64[];
  const int32_t F = a.full_nbr_num;
  int64_t* pnbr = lds64;
  GlxAlias* tab = reinterpret_cast<GlxAlias*>(pnbr + F);
  GlxAlias* stk = tab + F;
  float* dist = reinterpret_cast<float*>(stk + F);
  const int lane = threadIdx.x;
  const int64_t i = blockIdx.x;
  int64_t* walk = a.walks + i * a.walk_len;
  const int32_t t = a.step;
  const int64_t cur = t == 0 ? a.seeds[i] : walk[t - 1];
  const int64_t parent = walk_parent_of(a, i);
  const int64_t row = glx_row_of(a.map, cur);
  int64_t s = 0;
  int32_t n = 0;
  if (row >= 0) {
    s = a.row_ptr[row];
    const int64_t d = a.row_ptr[row + 1] - s;
    n = (int32_t)(d < F ? d : F);
  }
  if (n == 0) {
    if (lane == 0) walk[t] = a.default_nbr;
    return;
  }
  int32_t pn = 0;
  if (t > 0) {
    pn = a.seg[i];
    const int64_t used = a.off_used[i];
    if (used == a.off_true[i]) {  // no stuck walker with a parent list before this one: its own parent's list
      const int64_t prow = glx_row_of(a.map, parent);
      const int64_t ps = prow >= 0 ? a.row_ptr[prow] : 0;
      for (int32_t x = lane; x < pn; x += 64) pnbr[x] = a.adj[ps + x].nbr;
    } else {
      // the reference's window [used, used + pn) of the concatenated lists: element x belongs to the walker j with
      // off_true[j] <= x < off_true[j] + seg[j] (the last j whose list starts at or before x)
      for (int32_t x = lane; x < pn; x += 64) {
        const int64_t flat = used + x;
        int64_t lo = 0, hi = a.batch;  // largest j with off_true[j] <= flat
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if (a.off_true[mid] <= flat) lo = mid;
          else hi = mid;
        }
        const int64_t prow = glx_row_of(a.map, walk_parent_of(a, lo));
        pnbr[x] = a.adj[a.row_ptr[prow] + (flat - a.off_true[lo])].nbr;
      }
    }
  }
  __syncthreads();
  for (int32_t x = lane; x < n; x += 64) {
    const int64_t nbr = a.adj[s + x].nbr;
    const float w = a.weight ? a.weight[s + x] : a.default_weight;
    float biased;
    if (nbr == parent) {
      biased = (float)((double)w * 1.0 / ((double)a.p + 1e-6));  // :247-248 (double arithmetic, float store)
    } else {
      bool shared = false;
      for (int32_t y = 0; y < pn; ++y) shared |= pnbr[y] == nbr;
      biased = shared ? w : (float)((double)w * 1.0 / ((double)a.q + 1e-6));  // :250-260
    }
    dist[x] = biased;
  }
  __syncthreads();
  if (lane == 0) {
    glx_alias_build_row_dev(dist, n, tab, stk);
    const int32_t pick = glx_alias_pick(glx_draw64(a.seed, a.cc + (uint64_t)t, (uint32_t)i, 0u), n, tab);
    walk[t] = a.adj[s + pick].nbr;
  }
}

// DeepWalk step: uniform over the whole row; the draw of RandomSampler with neighbor_count 1.
__global__ void glx_deepwalk_step_kernel(WalkArgs a) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= a.batch) return;
  int64_t* walk = a.walks + i * a.walk_len;
  const int32_t t = a.step;
  const int64_t cur = t == 0 ? a.seeds[i] : walk[t - 1];
  const int64_t row = glx_row_of(a.map, cur);
  int64_t out = a.default_nbr;
  if (row >= 0) {
    const int64_t s = a.row_ptr[row];
    const int64_t d = a.row_ptr[row + 1] - s;
    if (d > 0) out = a.adj[s + (int64_t)glx_bounded(glx_draw64(a.seed, a.cc + (uint64_t)t, (uint32_t)i, 0u), (uint64_t)d)].nbr;
  }
  walk[t] = out;
}

}  // namespace

extern "C" int glx_random_walk(const glx_graph* g, const int64_t* seeds, int32_t batch, int32_t walk_len, float p,
                               float q, int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id,
                               uint64_t seed, uint64_t call_counter, int64_t* walks_out, int ptr_kind,
                               void* stream) {
  GLX_REQUIRE(g != nullptr, "graph is NULL");
  GLX_REQUIRE(batch >= 0 && walk_len >= 0, "negative batch / walk_len");
  GLX_REQUIRE(ptr_kind == GLX_PTR_HOST || ptr_kind == GLX_PTR_DEVICE, "bad ptr_kind");
  GLX_REQUIRE((int64_t)batch * walk_len <= INT32_MAX, "batch * walk_len exceeds int32 (tensor.h:47)");
  if (batch == 0 || walk_len == 0) return GLX_OK;
  GLX_REQUIRE(seeds && walks_out, "NULL data pointer");
  // RandomWalkRequest::IsDeepWalk, random_walk_request.cc:152-160
  const bool deep = fabsf(p - 1.0f) < 32 * 1.1920929e-07f && fabsf(q - 1.0f) < 32 * 1.1920929e-07f;
  GLX_REQUIRE(deep || (full_nbr_num >= 1 && full_nbr_num <= 2048), "DefaultFullNbrNum must be in [1, 2048], got %d",
              full_nbr_num);
  GlxDeviceGuard guard(g->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", g->device);
  hipStream_t s = ptr_kind == GLX_PTR_HOST ? glx_host_call_stream(stream, g->device) : glx_stream(stream);
  WalkArgs a;
  a.map = g->map();
  a.row_ptr = g->row_ptr;
  a.adj = g->adj;
  a.weight = g->weight;
  a.default_nbr = default_neighbor_id;
  a.seed = seed;
  a.cc = call_counter;
  a.p = p;
  a.q = q;
  a.default_weight = default_weight;
  a.batch = batch;
  a.walk_len = walk_len;
  a.full_nbr_num = full_nbr_num;
  const size_t nb = (size_t)batch, n_out = nb * (size_t)walk_len;
  int64_t* d = nullptr;
  if (ptr_kind == GLX_PTR_HOST) {
    int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d), (nb + n_out) * 8, s, 0);
    if (rc != GLX_OK) return rc;
    GLX_HIP(hipMemcpyAsync(d, seeds, nb * 8, hipMemcpyHostToDevice, s));
    a.seeds = d;
    a.walks = d + nb;
  } else {
    a.seeds = seeds;
    a.walks = walks_out;
  }
  a.seg = nullptr;
  a.off_true = a.off_used = nullptr;
  int32_t* d_seg = nullptr;
  if (!deep && walk_len > 1) {  // [seg i32 | seg_live i32 | off_true i64 | off_used i64] per walker
    int rc = glx_scratch_alloc(reinterpret_cast<void**>(&d_seg), nb * 24 + 16, s, 1);
    if (rc != GLX_OK) {
      if (d) glx_scratch_free(d, s);
      return rc;
    }
  }
  int32_t* d_live = d_seg ? d_seg + nb : nullptr;
  int64_t* d_true = d_seg ? reinterpret_cast<int64_t*>(d_seg + 2 * nb)  /* 8 nb bytes in: 8-byte aligned */ : nullptr;
  int64_t* d_used = d_seg ? d_true + nb : nullptr;
  GlxKernelTimer timer(GLX_KERNEL_SAMPLE, s);
  for (int32_t t = 0; t < walk_len; ++t) {
    a.step = t;
    if (deep) {
      glx_deepwalk_step_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(a);
    } else {
      if (t > 0) {
        glx_walk_segments_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(a, d_seg, d_live);
        glx_walk_scan2_kernel<<<1, 1024, 0, s>>>(d_seg, d_live, (int64_t)batch, d_true, d_used);
        a.seg = d_seg;
        a.off_true = d_true;
        a.off_used = d_used;
      }
      glx_node2vec_step_kernel<<<(unsigned)batch, 64, (size_t)full_nbr_num * 28, s>>>(a);
    }
  }
  timer.stop();
  if (d_seg) glx_scratch_free(d_seg, s);
  GLX_HIP(hipGetLastError());
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMemcpyAsync(walks_out, a.walks, n_out * 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  return GLX_OK;
}
