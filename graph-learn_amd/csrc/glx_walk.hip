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
};

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
  // the first step's parent is the seed itself, without neighbours (random_walk_request.cc:120-131)
  const int64_t parent = t <= 1 ? a.seeds[i] : walk[t - 2];
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
    const int64_t prow = glx_row_of(a.map, parent);
    if (prow >= 0) {
      const int64_t ps = a.row_ptr[prow];
      const int64_t pd = a.row_ptr[prow + 1] - ps;
      pn = (int32_t)(pd < F ? pd : F);
      for (int32_t x = lane; x < pn; x += 64) pnbr[x] = a.adj[ps + x].nbr;
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
  GlxKernelTimer timer(GLX_KERNEL_SAMPLE, s);
  for (int32_t t = 0; t < walk_len; ++t) {
    a.step = t;
    if (deep) {
      glx_deepwalk_step_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(a);
    } else {
      glx_node2vec_step_kernel<<<(unsigned)batch, 64, (size_t)full_nbr_num * 28, s>>>(a);
    }
  }
  timer.stop();
  GLX_HIP(hipGetLastError());
  if (ptr_kind == GLX_PTR_HOST) {
    GLX_HIP(hipMemcpyAsync(walks_out, a.walks, n_out * 8, hipMemcpyDeviceToHost, s));
    GLX_HIP(hipStreamSynchronize(s));
  }
  return GLX_OK;
}
