// glx request plans: a multi-hop sample (+ aggregate) request captured once into a hipGraph and
// replayed per batch.  Replaces the per-batch walk over a chain of DAG nodes
// (graphlearn/src/core/runner/dag_node_runner.cc:32-109: every node builds a request, runs its
// operator, hands its output tensors to the next node; core/dag/dag.h) and the Python hop loop of
// NeighborSampler.get (python/sampler/neighbor_sampler.py:93-127).
//
// Why: at small batches (B0 <= 8192) a step is a handful of kernels of a few microseconds each, and
// the time goes to launching them one by one.  A plan issues them as ONE graph launch.  What changes
// from run to run -- the seed ids and the call counter of the random streams -- enters through a
// single "stage" node (copies the seeds into the plan's input buffer, writes the counter to device
// memory) whose arguments are updated before each launch; everything downstream reads device memory,
// so the captured kernels never change.  Workspaces belong to the plan (glx_scratch_mode), outputs are
// plan-owned device buffers the caller reads after the run.
#include <string.h>

#include <new>
#include <vector>

#include "glx_common.h"

namespace {

__global__ void glx_plan_stage_kernel(int64_t* __restrict__ seeds_dst, const int64_t* __restrict__ seeds_src,
                                      int32_t batch, uint64_t* cc_dst, uint64_t cc) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) seeds_dst[i] = seeds_src[i];
  if (i == 0) *cc_dst = cc;
}

}  // namespace

struct glx_plan {
  int device = 0;
  int32_t num_hops = 0, batch = 0;
  std::vector<int32_t> fanouts;
  std::vector<int64_t> rows;  // request rows of hop h
  int32_t dim = 0;
  bool aggregates = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipGraphNode_t stage_node = nullptr;
  hipStream_t cap_stream = nullptr;
  int64_t* d_seeds = nullptr;
  uint64_t* d_cc = nullptr;
  char* arena = nullptr;
  std::vector<int64_t*> nbr, eid;
  std::vector<float*> emb;
  std::vector<int32_t*> cnt;
  // arguments of the stage node (kernelParams points here)
  int64_t* a_dst = nullptr;
  const int64_t* a_src = nullptr;
  int32_t a_batch = 0;
  uint64_t* a_cc_dst = nullptr;
  uint64_t a_cc = 0;
  void* params[5];
};

extern "C" void glx_plan_destroy(glx_plan* p) {
  if (!p) return;
  GlxDeviceGuard guard(p->device);
  (void)hipDeviceSynchronize();
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  if (p->d_seeds) (void)hipFree(p->d_seeds);
  if (p->d_cc) (void)hipFree(p->d_cc);
  if (p->arena) (void)hipFree(p->arena);
  for (auto* q : p->nbr) if (q) (void)hipFree(q);
  for (auto* q : p->eid) if (q) (void)hipFree(q);
  for (auto* q : p->emb) if (q) (void)hipFree(q);
  for (auto* q : p->cnt) if (q) (void)hipFree(q);
  delete p;
}

namespace {

// The captured sequence: hop h samples fanouts[h] neighbours of every vertex of hop h - 1 with the
// random stream (seed, call counter + h); then, deepest hop first, hop h's neighbours are reduced
// into hop h - 1's rows.
int issue(glx_plan* p, const glx_graph* const* graphs, int sampler, int padding_mode, int64_t default_nbr, uint64_t seed,
          const glx_features* const* feats, int agg_op, float default_attr, hipStream_t s) {
  const int64_t* frontier = p->d_seeds;
  for (int32_t h = 0; h < p->num_hops; ++h) {
    int rc = glx_sample(graphs[h], sampler, frontier, (int32_t)p->rows[h], p->fanouts[h], padding_mode, default_nbr, seed,
                        (uint64_t)h, p->nbr[h], p->eid[h], GLX_PTR_DEVICE, s);
    if (rc != GLX_OK) return rc;
    frontier = p->nbr[h];
  }
  if (p->aggregates) {
    for (int32_t h = p->num_hops - 1; h >= 0; --h) {
      const int64_t n_ids = p->rows[h] * p->fanouts[h];
      int rc = glx_aggregate(feats[h], agg_op, p->nbr[h], nullptr, (int32_t)n_ids, (int32_t)p->rows[h], default_attr,
                             p->emb[h], p->cnt[h], GLX_PTR_DEVICE, s);
      if (rc != GLX_OK) return rc;
    }
  }
  return GLX_OK;
}

}  // namespace

extern "C" int glx_plan_create(const glx_graph* const* graphs, int32_t num_hops, int sampler, const int32_t* fanouts,
                               int32_t batch, int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                               const glx_features* const* features, int agg_op, float default_attr,
                               glx_plan** out) {
  GLX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  GLX_REQUIRE(graphs && fanouts, "NULL argument");
  GLX_REQUIRE(num_hops >= 1 && num_hops <= 16, "num_hops must be in [1, 16]");
  GLX_REQUIRE(batch >= 1, "batch must be positive");
  GLX_REQUIRE(sampler >= GLX_SAMPLER_RANDOM && sampler <= GLX_SAMPLER_IN_DEGREE, "unknown sampler id %d", sampler);
  GLX_REQUIRE(padding_mode == GLX_PAD_CIRCULAR || padding_mode == GLX_PAD_REPLICATE, "bad padding_mode %d", padding_mode);
  GLX_REQUIRE(!features || (agg_op >= GLX_AGG_SUM && agg_op <= GLX_AGG_PROD), "unknown aggregator id %d", agg_op);
  int64_t rows = batch;
  for (int32_t h = 0; h < num_hops; ++h) {
    GLX_REQUIRE(graphs[h] != nullptr, "hop %d: NULL graph", h);
    GLX_REQUIRE(graphs[h]->device == graphs[0]->device, "all hops must live on one device");
    GLX_REQUIRE(fanouts[h] >= 1, "fanouts must be positive");
    GLX_REQUIRE(rows * fanouts[h] <= INT32_MAX, "hop %d exceeds int32 slots (tensor.h:47)", h);
    GLX_REQUIRE(!features || (features[h] != nullptr && features[h]->device == graphs[0]->device),
                "hop %d: NULL feature table / other device", h);
    GLX_REQUIRE(!features || rows * features[h]->dim <= INT32_MAX, "hop %d: segments * dim exceeds int32", h);
    rows *= fanouts[h];
  }
  const int device = graphs[0]->device;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  glx_plan* p = new (std::nothrow) glx_plan();
  GLX_REQUIRE(p != nullptr, "out of host memory");
  p->device = device;
  p->num_hops = num_hops;
  p->batch = batch;
  p->aggregates = features != nullptr;
  p->nbr.assign(num_hops, nullptr);
  p->eid.assign(num_hops, nullptr);
  p->emb.assign(num_hops, nullptr);
  p->cnt.assign(num_hops, nullptr);
  hipError_t e = hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_seeds), (size_t)batch * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_cc), 8);
  if (e == hipSuccess) e = hipMemset(p->d_seeds, 0, (size_t)batch * 8);
  if (e == hipSuccess) e = hipMemset(p->d_cc, 0, 8);
  rows = batch;
  for (int32_t h = 0; h < num_hops && e == hipSuccess; ++h) {
    p->fanouts.push_back(fanouts[h]);
    p->rows.push_back(rows);
    const size_t slots = (size_t)rows * fanouts[h];
    e = hipMalloc(reinterpret_cast<void**>(&p->nbr[h]), slots * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->eid[h]), slots * 8);
    if (e == hipSuccess && features) {
      e = hipMalloc(reinterpret_cast<void**>(&p->emb[h]), (size_t)rows * features[h]->dim * 4);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->cnt[h]), (size_t)rows * 4);
    }
    rows *= fanouts[h];
  }
  if (e != hipSuccess) {
    glx_set_error("request plan: allocation failed: %s", hipGetErrorString(e));
    glx_plan_destroy(p);
    return e == hipErrorOutOfMemory ? GLX_RESOURCE_EXHAUSTED : GLX_INTERNAL;
  }
  hipStream_t s = p->cap_stream;
  const bool was_profiling = glx_profile_suspend(true);
  (void)was_profiling;
  glx_capture_set_cc_dev(p->d_cc);
  // 1. dry run: executes once for real (valid inputs: seeds 0, counter 0) and records the workspaces
  glx_scratch_mode(GLX_SCRATCH_RECORD, nullptr, 0);
  int rc = issue(p, graphs, sampler, padding_mode, default_neighbor_id, seed, features, agg_op, default_attr, s);
  const size_t arena_bytes = glx_scratch_recorded_bytes();
  if (rc == GLX_OK && hipStreamSynchronize(s) != hipSuccess) rc = GLX_INTERNAL;
  if (rc == GLX_OK && arena_bytes > 0 && hipMalloc(reinterpret_cast<void**>(&p->arena), arena_bytes) != hipSuccess) {
    glx_set_error("request plan: workspace allocation failed");
    rc = GLX_RESOURCE_EXHAUSTED;
  }
  // 2. capture: the stage node, then the same sequence with the workspaces replayed from the arena
  if (rc == GLX_OK) {
    e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      glx_scratch_mode(GLX_SCRATCH_REPLAY, p->arena, arena_bytes);
      p->a_dst = p->d_seeds;
      p->a_src = p->d_seeds;
      p->a_batch = batch;
      p->a_cc_dst = p->d_cc;
      p->a_cc = 0;
      glx_plan_stage_kernel<<<(unsigned)((batch + 255) / 256), 256, 0, s>>>(p->a_dst, p->a_src, p->a_batch, p->a_cc_dst,
                                                                            p->a_cc);
      rc = issue(p, graphs, sampler, padding_mode, default_neighbor_id, seed, features, agg_op, default_attr, s);
      hipError_t e2 = hipStreamEndCapture(s, &p->graph);
      if (rc == GLX_OK && e2 != hipSuccess) e = e2;
    }
    if (rc == GLX_OK && e == hipSuccess) e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    if (rc == GLX_OK && e == hipSuccess) {
      size_t n = 0;
      e = hipGraphGetNodes(p->graph, nullptr, &n);
      std::vector<hipGraphNode_t> nodes(n);
      if (e == hipSuccess && n > 0) e = hipGraphGetNodes(p->graph, nodes.data(), &n);
      for (size_t i = 0; i < n && e == hipSuccess && !p->stage_node; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess || t != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess) continue;
        if (kp.func == reinterpret_cast<void*>(glx_plan_stage_kernel)) p->stage_node = nodes[i];
      }
      if (e == hipSuccess && !p->stage_node) {
        // the stage kernel is the first captured operation, hence the graph's only root
        size_t nr = 0;
        if (hipGraphGetRootNodes(p->graph, nullptr, &nr) == hipSuccess && nr == 1) {
          (void)hipGraphGetRootNodes(p->graph, &p->stage_node, &nr);
        }
      }
      if (e == hipSuccess && !p->stage_node) {
        glx_set_error("request plan: the stage node was not found in the captured graph");
        rc = GLX_INTERNAL;
      }
    }
    if (rc == GLX_OK && e != hipSuccess) {
      glx_set_error("request plan: graph capture failed: %s", hipGetErrorString(e));
      rc = GLX_INTERNAL;
    }
  }
  glx_scratch_mode(GLX_SCRATCH_NORMAL, nullptr, 0);
  glx_capture_set_cc_dev(nullptr);
  glx_profile_suspend(false);
  (void)hipGetLastError();
  if (rc != GLX_OK) {
    glx_plan_destroy(p);
    return rc;
  }
  if (features) p->dim = features[num_hops - 1]->dim;
  *out = p;
  return GLX_OK;
}

extern "C" int glx_plan_run(glx_plan* p, const int64_t* seeds, uint64_t call_counter, void* stream) {
  GLX_REQUIRE(p != nullptr && seeds != nullptr, "NULL argument");
  GlxDeviceGuard guard(p->device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", p->device);
  p->a_src = seeds;
  p->a_cc = call_counter;
  p->params[0] = &p->a_dst;
  p->params[1] = &p->a_src;
  p->params[2] = &p->a_batch;
  p->params[3] = &p->a_cc_dst;
  p->params[4] = &p->a_cc;
  hipKernelNodeParams kp;
  memset(&kp, 0, sizeof(kp));
  kp.func = reinterpret_cast<void*>(glx_plan_stage_kernel);
  kp.gridDim = dim3((unsigned)((p->batch + 255) / 256), 1, 1);
  kp.blockDim = dim3(256, 1, 1);
  kp.sharedMemBytes = 0;
  kp.kernelParams = p->params;
  kp.extra = nullptr;
  GLX_HIP(hipGraphExecKernelNodeSetParams(p->exec, p->stage_node, &kp));
  GLX_HIP(hipGraphLaunch(p->exec, glx_stream(stream)));
  return GLX_OK;
}

extern "C" int glx_plan_output(const glx_plan* p, int32_t hop, int64_t** nbr, int64_t** eid, float** emb,
                               int32_t** cnt, int64_t* rows, int32_t* fanout) {
  GLX_REQUIRE(p != nullptr, "plan is NULL");
  GLX_REQUIRE(hop >= 0 && hop < p->num_hops, "hop %d outside [0, %d)", hop, p->num_hops);
  if (nbr) *nbr = p->nbr[hop];
  if (eid) *eid = p->eid[hop];
  if (emb) *emb = p->emb[hop];
  if (cnt) *cnt = p->cnt[hop];
  if (rows) *rows = p->rows[hop];
  if (fanout) *fanout = p->fanouts[hop];
  return GLX_OK;
}
