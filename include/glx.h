/* glx -- MI355X-native drop-in for the graph-learn sampling / aggregation hot path.
 *
 * This is the C-ABI boundary (plain pointers and sizes; no C++/torch types).
 * The C++ operators that mirror the reference's registry (graph-learn_amd/host)
 * call it from their Process() bodies; a maintainer of the reference would call
 * it the same way from graphlearn::op::Sampler / Aggregator subclasses (see
 * INTEGRATION.md).  Every entry point names the reference interface it replaces.
 *
 * Conventions
 *   - return value: graphlearn::error::Code (graphlearn/src/include/status.h:29-49),
 *     0 = OK; a message for the last failure on this thread: glx_last_error().
 *   - `ptr_kind` says whether the data pointers of THAT call are host
 *     (GLX_PTR_HOST: the call copies in/out and returns when results are in
 *     host memory) or device pointers on the handle's GPU (GLX_PTR_DEVICE: the
 *     call only enqueues work on `stream` and returns; the caller orders later
 *     use on that stream).
 *   - `stream` is a hipStream_t passed as void*.  NULL means the null stream for
 *     device-pointer calls; host-pointer calls with NULL run on a private
 *     per-(host thread, device) stream so that concurrent callers overlap.
 *   - handles are immutable after creation and may be used concurrently from
 *     any number of host threads (the reference calls Process() from up to 32
 *     pool threads on one operator instance: in_memory_service.cc:64-71).
 *   - no entry point throws or takes ownership of caller memory.
 */
#ifndef GLX_H_
#define GLX_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: graph / features / samplers / aggregators / partition helpers.  2 (additions only): host registration, shard
 * communicator, distributed store (+ replicas), request plans.  3 (additions only): memory-system probes,
 * induced sub-graph, conditional negative sampling.  4 (additions only): the speculation ledger of the distributed store
 * (glx_dist_ledger_*, glx_dist_confirm, GLX_ABORTED), glx_tune.  5 (additions only): glx_dist_sample_full_filtered,
 * glx_dist_in_degrees, glx_dist_negative_create, glx_dist_negative_sample, glx_dist_random_walk_ex (all added during
 * round 4 under 4), glx_graph_enable_default_weight. */
#define GLX_ABI_VERSION 5

/* Exported symbols: libglx.so is built with -fvisibility=hidden. */
#if defined(__GNUC__)
#define GLX_API __attribute__((visibility("default")))
#else
#define GLX_API
#endif

/* graphlearn::error::Code values used by this library (status.h:29-49). */
#define GLX_OK 0
#define GLX_INVALID_ARGUMENT 3
#define GLX_RESOURCE_EXHAUSTED 8
#define GLX_ABORTED 10 /* ABI 4: a speculated exchange did not fit (glx_dist_ledger); redo the unconfirmed calls */
#define GLX_OUT_OF_RANGE 11
#define GLX_UNIMPLEMENTED 12
#define GLX_INTERNAL 13
#define GLX_UNAVAILABLE 14

#define GLX_PTR_HOST 0
#define GLX_PTR_DEVICE 1

/* Sampler ids; the registry names they serve are in the comments
 * (REGISTER_OPERATOR in random_sampler.cc:80, random_without_replacement_sampler.cc:79,
 * edge_weight_sampler.cc:128, topk_sampler.cc:72). */
#define GLX_SAMPLER_RANDOM 0                     /* "RandomSampler" */
#define GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT 1 /* "RandomWithoutReplacementSampler" */
#define GLX_SAMPLER_EDGE_WEIGHT 2                /* "EdgeWeightSampler" */
#define GLX_SAMPLER_TOPK 3                       /* "TopkSampler" */
#define GLX_SAMPLER_IN_DEGREE 4                  /* "InDegreeSampler" (in_degree_sampler.cc:117); needs
                                                    glx_graph_enable_in_degree() */

/* Aggregator ids ("SumAggregator" ... sum_aggregator.cc:36, mean_aggregator.cc:64,
 * max_aggregator.cc:43, min_aggregator.cc:43, prod_aggregator.cc:43). */
#define GLX_AGG_SUM 0
#define GLX_AGG_MEAN 1
#define GLX_AGG_MAX 2
#define GLX_AGG_MIN 3
#define GLX_AGG_PROD 4

/* GLOBAL_FLAG(PaddingMode) values (constants.h:119-122, config.cc:94). */
#define GLX_PAD_REPLICATE 0
#define GLX_PAD_CIRCULAR 1

typedef struct glx_graph glx_graph;       /* one edge type: device CSR + alias table + id map */
typedef struct glx_features glx_features; /* one node type: device [V, D] float matrix + id map */

/* ---- library ------------------------------------------------------------ */
GLX_API int glx_abi_version(void);
/* Number of visible GPUs; GLX_UNAVAILABLE (and *count = 0) when there is no
 * usable HIP device.  Nothing in this library falls back to the CPU. */
GLX_API int glx_device_count(int* count);
GLX_API const char* glx_last_error(void);

/* Pins (page-locks and maps) a caller-owned host buffer so that the copies of host-pointer calls into / out
 * of it are single DMA transfers instead of bouncing through the runtime's pageable staging -- what the
 * response tensors of the C++ layer are backed by (the role of the reference's RepeatedField storage,
 * service/tensor_impl.h:72-75,196-200, on this path).  Both return GLX_UNAVAILABLE without a GPU runtime;
 * an unregistered buffer works everywhere, only slower.  A result is written directly only when the WHOLE output extent
 * lies inside one range registered HERE; memory pinned by other means is staged.  A buffer that straddles the end of a
 * registered range is the caller's error (the runtime refuses a copy across that boundary: GLX_INTERNAL, "invalid argument").
 * WHERE the memory comes from matters (ROCm 7.0, measured in round 6): register private anonymous mappings of their own
 * -- mmap(MAP_PRIVATE | MAP_ANONYMOUS), 2 MiB aligned, whole 2 MiB granules -- NOT ranges of the malloc heap.  A process
 * that has registered parts of its heap and also holds heap memory marked MADV_HUGEPAGE (numpy marks every array of
 * 4 MiB or more) gets "an illegal memory access" from later PAGEABLE host-to-device copies, anywhere in the process:
 * scripts/r06/repro/hostreg_pageable.hip shows it with the runtime alone (registered ranges cut from the heap: a fault
 * within ~50 rounds of copies; cut from mappings of their own: none in 600), unregistered or not.  The C++ layer's
 * response pool (host/src/base.cc) follows the rule; long-lived pools that never unregister are the intended use.
 * A range inside the process's brk heap is refused with GLX_INVALID_ARGUMENT (GLX_HOST_REGISTER_HEAP=1 lifts the check). */
GLX_API int glx_host_register(void* p, uint64_t bytes);
GLX_API int glx_host_unregister(void* p);

/* ---- graph storage: replaces GraphStorage::GetNeighbors / GetOutEdges /
 * GetEdgeWeight (graph_storage.h:40-57) + CompressedMemoryAdjMatrix
 * (memory_adj_matrix.cc:159-225) + AutoIndex (auto_indexing.cc:21-33). -----
 *
 * Input is the adjacency exactly as the reference exposes it after Build():
 * row r holds neighbours col[row_ptr[r] .. row_ptr[r+1]) and their edge ids
 * eid[..] IN THE REFERENCE'S ROW ORDER (weight-descending for weighted edge
 * types, memory_adj_matrix.cc:105-125).  `weight` is per CSR slot (the
 * reference stores it per edge id: memory_edge_storage.cc:97-103) or NULL for
 * an unweighted type.  `ids` maps row -> raw vertex id, or NULL when raw id v
 * is row v.  row_ptr is 64-bit (the reference's int32 offsets overflow at
 * E >= 2^31: types.h:28).  When weights are given the per-row alias tables of
 * AliasMethod::Build (alias_method.cc:57-107) are built once, on the device.
 */
GLX_API int glx_graph_create(int device, int64_t num_rows, int64_t num_edges, const int64_t* row_ptr,
                     const int64_t* col, const int64_t* eid, const float* weight,
                     const int64_t* ids, int ptr_kind, void* stream, glx_graph** out);
/* Device-side storage build from a raw edge list (the loader's output): replaces
 * LocalGraph::UpdateEdges -> GraphStorage::Add (local_graph.cc:50-64,
 * memory_graph_storage.cc:49-54), MemoryAdjMatrix::Build / Sort
 * (memory_adj_matrix.cc:60-66,105-125) and CompressedMemoryAdjMatrix::Build
 * (:169-189).  src/dst/weight[num_edges] are in insertion order, so edge id i is
 * edge i (memory_edge_storage.cc:53-57); weight may be NULL.  Rows are the distinct
 * source ids (a device hash map replaces AutoIndex).  edge_ids (may be NULL) overrides
 * the ids -- a shard that holds a subset of the edges passes their GLOBAL ids; the
 * array order is still the insertion order.  sort_by_weight != 0 orders
 * every row by weight descending like Build() with IndexOption "sort" does; ties
 * keep insertion order (the reference's std::sort leaves them unspecified).
 * The whole build (two radix sorts, run-length encode, scan, gather, alias
 * tables, id map) runs on the GPU. */
GLX_API int glx_graph_build(int device, int64_t num_edges, const int64_t* src, const int64_t* dst,
                    const float* weight, const int64_t* edge_ids, int sort_by_weight, int ptr_kind,
                    void* stream, glx_graph** out);
/* Same, with the row order spelled out.  GLX_ORDER_TIMESTAMP_ASC orders every row by the
 * edges' timestamps ascending, which is what Build() does for timestamped edge types -- it
 * takes precedence over the weight order (memory_adj_matrix.cc:60-66,129-148); ties keep
 * insertion order.  timestamp (may be NULL for the other orders) is also kept per slot
 * for timestamp filters (glx_sample_filtered). */
#define GLX_ORDER_INSERTION 0
#define GLX_ORDER_WEIGHT_DESC 1
#define GLX_ORDER_TIMESTAMP_ASC 2
GLX_API int glx_graph_build_ordered(int device, int64_t num_edges, const int64_t* src, const int64_t* dst,
                            const float* weight, const int64_t* edge_ids, const int64_t* timestamp, int order,
                            int ptr_kind, void* stream, glx_graph** out);
GLX_API void glx_graph_destroy(glx_graph* g);
GLX_API int glx_graph_info(const glx_graph* g, int64_t* num_rows, int64_t* num_edges, int* weighted,
                   int* has_id_map, int* device);
/* EdgeWeightSampler on an UNWEIGHTED edge type: the reference reads GLOBAL_FLAG(DefaultWeight) for every edge
 * (MemoryEdgeStorage::GetWeight, memory_edge_storage.cc:97-103: an id beyond the empty weight array) and builds its
 * alias row from that (edge_weight_sampler.cc:78-92) -- with the default 0.0 every prob is 0/0 = NaN, no slot takes its
 * alias and a draw is idx = (int)(float)uniform[0, deg - 1) (alias_method.cc:117-121).  This gives the graph that
 * constant weight per slot and the alias tables AliasMethod::Build makes of it (NaN rows included, bit for bit); it
 * MUTATES the handle like glx_graph_enable_in_degree: call it once, before sampling from other threads.  A graph that
 * has weights is left alone.  (ABI 5) */
GLX_API int glx_graph_enable_default_weight(glx_graph* g, float default_weight, void* stream);
/* Copy the device alias table out (parity checks against alias_method.cc:57-107). */
GLX_API int glx_graph_export_alias(const glx_graph* g, float* prob, int32_t* alias, int ptr_kind,
                           void* stream);
/* Degrees of a batch of raw ids (0 for unknown ids), as GetNeighbors().Size(). */
GLX_API int glx_graph_degrees(const glx_graph* g, const int64_t* src, int64_t n, int64_t* deg_out,
                      int ptr_kind, void* stream);
/* In-degrees of a batch of raw destination ids (0 for ids no edge points to), as
 * GraphStorage::GetInDegree (topo_statics.cc:62-69).  Needs glx_graph_enable_in_degree(g). */
GLX_API int glx_graph_in_degrees(const glx_graph* g, const int64_t* ids, int64_t n, int64_t* deg_out,
                         int ptr_kind, void* stream);

/* ---- neighbour sampling: replaces Sampler::Sample of the four samplers
 * (random_sampler.cc:33-76, random_without_replacement_sampler.cc:31-75,
 * edge_weight_sampler.cc:31-92, topk_sampler.cc:29-68) incl. the padders
 * (padder/circular_padder.h:36-66, padder/replicate_padder.h:37-56) and
 * SamplingResponse::FillWith for unknown / empty rows. --------------------
 *
 * src[batch] raw ids -> nbr_out[batch*k], eid_out[batch*k], row-major dense,
 * the layout of SamplingResponse kNodeIds / kEdgeIds (sampling_request.cc:226-251).
 * `padding_mode` and `default_neighbor_id` are the reference's global flags,
 * passed explicitly.  (`seed`, `call_counter`) select the random stream (the
 * seeding contract in DESIGN.md): the same pair gives the same output,
 * bit-for-bit, on every run and every GPU count.  Requests with a Filter go through
 * glx_sample_filtered.
 */
GLX_API int glx_sample(const glx_graph* g, int sampler, const int64_t* src, int32_t batch, int32_t k,
               int padding_mode, int64_t default_neighbor_id, uint64_t seed,
               uint64_t call_counter, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
               void* stream);

/* ---- further samplers of the registry -------------
 * InDegreeSampler (in_degree_sampler.cc:33-114) is EdgeWeightSampler with the
 * neighbour's in-degree inside this edge type as the weight (GraphStorage::
 * GetInDegree, topo_statics.cc:33-69), rebuilt per row per request in the
 * reference.  glx_graph_enable_in_degree counts the in-degrees and builds that
 * second set of alias tables once, on the device; it MUTATES the handle: call it
 * right after creation, before the handle is shared between threads. */
GLX_API int glx_graph_enable_in_degree(glx_graph* g, void* stream);

/* FullSampler (full_sampler.cc:28-97): every row returns its first
 * min(max_limit, deg) neighbours in storage order (max_limit <= 0: all) as a
 * sparse response -- SamplingResponse's segments + values (sampling_request.cc:
 * 198-224).  Two calls, like the reference's two passes:
 *   glx_sample_full_sizes  -> degrees_out[batch] (the segments) and
 *                             offsets_out[batch+1] (exclusive prefix sum; the last
 *                             entry is the total number of values);
 *   glx_sample_full        -> nbr_out / eid_out[offsets[batch]], row i at offsets[i]
 *                             (both may be NULL for an empty response, offsets[batch] == 0;
 *                             glx_sample_full_filtered likewise). */
GLX_API int glx_sample_full_sizes(const glx_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                          int32_t* degrees_out, int64_t* offsets_out, int ptr_kind, void* stream);
GLX_API int glx_sample_full(const glx_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                    const int64_t* offsets, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                    void* stream);

/* Same as glx_sample, but request row i draws from the random stream of row
 * rng_rows[i] instead of row i (rng_rows == NULL: identical to glx_sample).  A
 * shard that serves a slice of a partitioned request (DistributeRunner::Run,
 * op_runner.h:60-84) passes the rows' indices in the ORIGINAL request -- the
 * Sticker values of hash_partitioner.h:69 -- so the stitched result is
 * bit-identical to the unpartitioned one for every shard count. */
GLX_API int glx_sample_ex(const glx_graph* g, int sampler, const int64_t* src, const int64_t* rng_rows,
                  int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id,
                  uint64_t seed, uint64_t call_counter, int64_t* nbr_out, int64_t* eid_out,
                  int ptr_kind, void* stream);

/* Multi-hop driver: replaces the host-side hop loop of NeighborSampler.get
 * (graphlearn/python/sampler/neighbor_sampler.py:93-127) / a chain of sampling
 * DAG nodes (core/runner/dag_node_runner.cc:32-109).  Hop h samples fanouts[h]
 * neighbours of every vertex of hop h-1 (hop 0 = `seeds`) from graphs[h] -- one
 * edge type per hop, i.e. the meta-path -- with random stream (seed,
 * call_counter + h).  Frontiers stay on the device between hops; hop h's
 * batch * fanouts[0] * ... * fanouts[h] slots are written to nbr_out[h] /
 * eid_out[h] (eid_out, or single entries of it, may be NULL).  Equivalent to
 * num_hops glx_sample calls feeding each other. */
GLX_API int glx_sample_hops(const glx_graph* const* graphs, int32_t num_hops, int sampler,
                    const int64_t* seeds, int32_t batch, const int32_t* fanouts, int padding_mode,
                    int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                    int64_t* const* nbr_out, int64_t* const* eid_out, int ptr_kind, void* stream);

/* ---- sampling filters: replaces op::Filter (core/operator/sampler/filter.h:30-125,
 * filter.cc:69-229) as the samplers use it (random_sampler.cc:52-71, topk_sampler.cc:52-61,
 * random_without_replacement_sampler.cc:56-68, edge_weight_sampler.cc:55-66,94-112,
 * in_degree_sampler.cc:54-65,94-113, full_sampler.cc:66-84). ----------------------------
 * type / field carry the reference's enum values (include/constants.h:135-145).  values[batch]
 * is the filter tensor AFTER Filter::FillValues: one value per request row, in the same
 * memory kind as src.  A neighbour "hits" when its field (ID = the neighbour id, TIMESTAMP =
 * the edge's timestamp, default_timestamp when the graph has none) == / > the row's value,
 * and hits are what sampling avoids:
 *   RandomSampler   a row whose neighbours all hit is default-filled; otherwise every slot
 *                   redraws a hit up to retry_times times and then keeps it (attempt a of
 *                   slot j is draw j + a * k of the row's stream);
 *   the others      sample from the reserved positions Filter::ActOn leaves, in ITS order:
 *                   survivors keep their place, holes left by hits are refilled from the
 *                   right end, rightmost survivor first (filter.cc:83-94); TIMESTAMP +
 *                   LARGER_THAN instead keeps the prefix below values[0] -- the value of
 *                   request row 0, for every row (filter.h:107-111) -- in descending order,
 *                   and keeps nothing of a single-neighbour row (filter.cc:208-210).
 *                   EdgeWeight / InDegree rebuild the alias table over the reserved
 *                   neighbours' weights per row, like the reference; padding as before.
 * glx_sample_full_filtered: the row sizes stay those of glx_sample_full_sizes (the
 *   UNFILTERED min(limit, deg), full_sampler.cc:55-62) and the reserved neighbours are padded
 *   up to that size; a row whose neighbours all hit yields default ids (the reference's
 *   FillWith(dim2) breaks the ragged layout there).
 * filter == NULL or type == GLX_FILTER_NONE: identical to the unfiltered entry points. */
#define GLX_FILTER_NONE 0
#define GLX_FILTER_EQUAL 1
#define GLX_FILTER_LARGER_THAN 2
#define GLX_FILTER_FIELD_NONE 0
#define GLX_FILTER_FIELD_ID 1
#define GLX_FILTER_FIELD_TIMESTAMP 2
typedef struct glx_filter {
  int32_t type;
  int32_t field;
  const int64_t* values;
  int32_t retry_times;       /* GLOBAL_FLAG(SamplingRetryTimes), config.cc:108; RandomSampler only */
  int64_t default_timestamp; /* GLOBAL_FLAG(DefaultTimestamp), memory_edge_storage.cc:113-119 */
} glx_filter;
/* Per-slot timestamps for a handle made by glx_graph_create: ts_slot[num_edges] in the
 * order of `col`.  MUTATES the handle, like glx_graph_enable_in_degree. */
GLX_API int glx_graph_set_timestamps(glx_graph* g, const int64_t* ts_slot, int ptr_kind, void* stream);
GLX_API int glx_sample_filtered(const glx_graph* g, int sampler, const int64_t* src, const int64_t* rng_rows,
                                int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id,
                                uint64_t seed, uint64_t call_counter, const glx_filter* filter,
                                int64_t* nbr_out, int64_t* eid_out, int ptr_kind, void* stream);
GLX_API int glx_sample_full_filtered(const glx_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                                     const int64_t* offsets, int padding_mode, int64_t default_neighbor_id,
                                     const glx_filter* filter, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                                     void* stream);

/* ---- random walks: replaces the "RandomWalk" operator (core/operator/random_walk/
 * random_walk.cc:30-276) -- DeepWalk (:168-190) when p = q = 1 within 32 FLT_EPSILON
 * (RandomWalkRequest::IsDeepWalk, random_walk_request.cc:152-160), node2vec otherwise
 * (WeightedRandomWalk / WeightedRandomWalkKernel, :192-272). ----------------------------
 * walks_out[batch * walk_len], row i = the walk_len vertices visited from seeds[i] (the
 * response layout, :121-127).  A vertex without out-edges yields default_neighbor_id, and
 * the walk goes on from THAT id, like the reference.  node2vec looks at the first
 * min(deg, full_nbr_num) neighbours only (GLOBAL_FLAG(DefaultFullNbrNum), <= 2048 here),
 * weighs an edge by 1/(p + 1e-6) when it returns to the parent, by 1 when it leads to one of
 * the parent's first full_nbr_num neighbours, by 1/(q + 1e-6) otherwise, and draws once from
 * the alias table of those weights; the first step's parent is the seed itself, without
 * neighbours (random_walk_request.cc:120-131).  default_weight stands in for the edge
 * weights of an unweighted graph (GLOBAL_FLAG(DefaultWeight)).  Step t of walker i uses draw 0
 * of the stream (seed, call_counter + t, i).  The reference does not advance its cursor
 * into the parents' neighbour lists past a walker that is stuck (:214-226), which shifts the
 * windows of all later walkers of the batch: reproduced (round 4) -- walks equal the reference's
 * also on graphs with dead ends. */
GLX_API int glx_random_walk(const glx_graph* g, const int64_t* seeds, int32_t batch, int32_t walk_len, float p,
                            float q, int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id,
                            uint64_t seed, uint64_t call_counter, int64_t* walks_out, int ptr_kind, void* stream);

/* ---- node features: replaces NodeStorage::GetAttribute()->GetFloats()
 * (node_storage.h:51-54, compressed_memory_node_storage.cc:149-176). -------
 * X is [num_rows, dim] row-major float32 (SideInfo.f_num == dim). */
GLX_API int glx_features_create(int device, int64_t num_rows, int32_t dim, const float* X,
                        const int64_t* ids, int ptr_kind, void* stream, glx_features** out);
/* Non-owning view of a device-resident [num_rows, dim] matrix with the dense id
 * map (id v is row v): lets glx_aggregate reduce rows that arrived from other
 * shards (the halo exchange) without a copy.  X must outlive the handle. */
GLX_API int glx_features_view(int device, int64_t num_rows, int32_t dim, const float* X_device,
                      glx_features** out);
GLX_API void glx_features_destroy(glx_features* f);
GLX_API int glx_features_info(const glx_features* f, int64_t* num_rows, int32_t* dim, int* has_id_map,
                      int* device);

/* ---- aggregation: replaces Aggregator::Aggregate (aggregator.cc:25-59) with
 * Sum/Mean/Max/Min/Prod Init/Agg/Final (sum_aggregator.cc:25-33,
 * mean_aggregator.cc:26-61, max_aggregator.cc:26-40, min_aggregator.cc,
 * prod_aggregator.cc). ---------------------------------------------------
 * node_ids[num_ids], segment_ids[num_ids] (consumed with the reference's
 * cursor rule, aggregating_request.cc:86-105) -> emb_out[num_segments*dim],
 * cnt_out[num_segments].  Unknown ids contribute a row of `default_attr`;
 * empty segments are `default_attr` (GLOBAL_FLAG(DefaultFloatAttribute)).
 * Each output element is accumulated in the reference's left-to-right order,
 * so results are bit-identical to the reference for every op.  segment_ids == NULL
 * means num_segments equal segments of num_ids / num_segments consecutive ids -- the
 * layout of a dense sampler response (segment i = the neighbours of request row i):
 * the segment bookkeeping kernels and the read of the segment ids are skipped.  Device-pointer
 * callers get the float4 path when dim % 4 == 0 and emb_out (and a view's X) are
 * 16-byte aligned; any other alignment silently takes the scalar path. */
GLX_API int glx_aggregate(const glx_features* f, int op, const int64_t* node_ids,
                  const int32_t* segment_ids, int32_t num_ids, int32_t num_segments,
                  float default_attr, float* emb_out, int32_t* cnt_out, int ptr_kind,
                  void* stream);

/* ---- negative sampling: replaces RandomNegativeSampler (random_negative_sampler.cc:30-63),
 * InDegreeNegativeSampler / SoftInDegreeNegativeSampler (in_degree_negative_sampler.cc:29-135)
 * and NodeWeightNegativeSampler (node_weight_negative_sampler.cc:29-110). ---------------
 * A glx_negative is the candidate list all rows of a request draw from, in HBM:
 *   glx_negative_from_graph: the edge type's distinct destination ids in first-appearance
 *     order (TopoStatics::Add, topo_statics.cc:32-55 = GetAllDstIds()), uniform
 *     (by_in_degree = 0) or weighted by in-degree (GetAllInDegrees()) through ONE alias
 *     table over the whole list (AliasMethodFactory::LookupOrCreate), bit-identical to
 *     AliasMethod::Build (alias_method.cc:57-107);
 *   glx_negative_create: explicit ids (+ weights or NULL = uniform): a node type's ids and
 *     node weights for NodeWeightNegativeSampler.
 * glx_negative_sample: out[batch * count] candidates (no edge ids, as in the reference).
 *   exclude = GLX_NEG_EXCLUDE_NONE      every draw is taken (Random, SoftInDegree);
 *             GLX_NEG_EXCLUDE_NEIGHBORS a candidate that is an out-neighbour of src[i] in `g` is
 *                                       skipped (InDegree; needs glx_graph_enable_negative(g));
 *             GLX_NEG_EXCLUDE_BATCH     a candidate equal to ANY src id of the request is skipped
 *                                       (NodeWeight).
 *   The strict modes follow the reference's retry loop: blocks of `count` draws, accepted
 *   candidates keep their order, the exclusion set is dropped from the 4th block on
 *   (kRetryTimes = 3).  Row i draws from the random stream (seed, call_counter, i); draw
 *   b * count + j is candidate j of block b.  Weighted draws use AliasMethod::Sample's formula
 *   (alias_method.cc:117-121), uniform ones floor(u * num_ids / 2^64).  An empty candidate
 *   list yields `default_neighbor_id` everywhere. */
#define GLX_NEG_EXCLUDE_NONE 0
#define GLX_NEG_EXCLUDE_NEIGHBORS 1
#define GLX_NEG_EXCLUDE_BATCH 2
typedef struct glx_negative glx_negative;
GLX_API int glx_negative_create(int device, int64_t num_ids, const int64_t* ids, const float* weights, int ptr_kind,
                                void* stream, glx_negative** out);
GLX_API int glx_negative_from_graph(const glx_graph* g, int by_in_degree, void* stream, glx_negative** out);
GLX_API void glx_negative_destroy(glx_negative* t);
GLX_API int glx_negative_info(const glx_negative* t, int64_t* num_ids, int* weighted);
/* Copies the candidate ids and the alias table to HOST arrays (parity checks). */
GLX_API int glx_negative_export(const glx_negative* t, int64_t* ids, float* prob, int32_t* alias, void* stream);
/* Builds every row's neighbour ids in ascending order, each with the CSR slot it came from: the exclusion
 * test of strict negative sampling is a binary search in it, and an id == value filter
 * (glx_sample_filtered) finds its hits the same way instead of scanning the row.  MUTATES the handle. */
GLX_API int glx_graph_enable_negative(glx_graph* g, void* stream);
GLX_API int glx_negative_sample(const glx_negative* t, int exclude, const glx_graph* g, const int64_t* src,
                                int32_t batch, int32_t count, int64_t default_neighbor_id, uint64_t seed,
                                uint64_t call_counter, int64_t* out, int ptr_kind, void* stream);

/* ---- feature lookup: replaces LookupNodes' float-attribute gather
 * (node_lookuper.cc:24-52); out[n*dim], unknown ids -> default_attr. */
GLX_API int glx_lookup(const glx_features* f, const int64_t* node_ids, int64_t n, float default_attr,
               float* out, int ptr_kind, void* stream);

/* ---- shard exchange helpers: replace HashPartitioner::Partition
 * (hash_partitioner.h:33-92; shard = llabs(id) % P, stable inside a shard) and
 * Stitcher::DoStitch (stitcher.h:67-107).  These take device pointers only
 * (they sit between two RCCL all-to-alls). ---------------------------------
 * glx_partition: ids[n] -> bucketed[n] (ids grouped by shard, original order
 *   kept inside a shard), order[n] (the concatenated Sticker lists: order[i] is
 *   the original index of bucketed[i]), counts[num_shards] (int64, device).
 * glx_stitch_i64 / _f32: out[order[i]*width + c] = in[i*width + c]. */
GLX_API int glx_partition(int device, const int64_t* ids, int64_t n, int32_t num_shards,
                  int64_t* bucketed, int64_t* order, int64_t* counts, void* stream);
GLX_API int glx_stitch_i64(int device, const int64_t* in, const int64_t* order, int64_t n, int32_t width,
                   int64_t* out, void* stream);
GLX_API int glx_stitch_f32(int device, const float* in, const int64_t* order, int64_t n, int32_t width,
                   float* out, void* stream);

/* glx_aggregate_stitch: replaces AggregatingResponse::Stitch
 *   (aggregating_request.cc:172-213 + the aggregators' AggFunc/FinalFunc): combines the
 *   partial results of `num_parts` shards, parts[num_parts][num_segments*dim] and
 *   cnts[num_parts][num_segments] (the receive buffers of one all-to-all), folding in
 *   shard order from InitFunc's value; Mean re-weights partial means by their counts.
 *   A shard with count 0 for a segment is skipped (the reference folds its
 *   DefaultFloatAttribute row in, which corrupts Max/Min/Prod: SURVEY 8(a) quirk 8);
 *   when no partial is empty the result is bit-identical to the reference's Stitch. */
GLX_API int glx_aggregate_stitch(int device, int op, int32_t num_parts, const float* parts,
                                 const int32_t* cnts, int32_t num_segments, int32_t dim, float default_attr,
                                 float* emb_out, int32_t* cnt_out, void* stream);

/* ---- shard communicator: replaces the RPC layer under DistributeRunner
 * (core/runner/op_runner.h:86-152 RunInParallel: one gRPC call per remote shard,
 * service/client_impl.cc + rpc/) with RCCL point-to-point groups over xGMI. -----------
 * One glx_comm per process (= per GPU) and per concurrently used stream; `rank` plays the
 * role of the reference's server id and `world` of its server count (hash_partitioner.h:33-92
 * routes id v to shard llabs(v) % world).  Three transports, one interface:
 *   glx_comm_init_rccl       ncclCommInitRank from a unique id that rank 0 made with
 *                            glx_comm_unique_id and handed to the others out of band (the
 *                            role of the reference's coordinator / naming engine);
 *                            exchanges are ncclGroupStart + ncclSend/ncclRecv + ncclGroupEnd
 *                            on the caller's stream (librccl is loaded with dlopen, so the
 *                            library loads without it);
 *   glx_comm_init_local      `world` ranks that are threads of ONE process (each may use its
 *                            own GPU or all the same one): peers copy device-to-device out of
 *                            each other's send buffers between two host barriers.  Ranks that
 *                            pass the same `fabric_key` form one communicator.  This is the
 *                            one-GPU test rig and the single-process multi-GPU mode;
 *   glx_comm_init_callbacks  host-staged: buffers are copied to pinned host memory and handed
 *                            to the caller's all-to-all / all-gather (e.g. torch.distributed
 *                            gloo, MPI) -- for ranks that cannot run RCCL together. */
#define GLX_UNIQUE_ID_BYTES 128
#define GLX_COMM_RCCL 0
#define GLX_COMM_LOCAL 1
#define GLX_COMM_CALLBACKS 2
typedef struct glx_comm glx_comm;
/* send/recv are packed host buffers (peer-major); counts are in elements of elem_bytes. */
typedef int (*glx_host_alltoallv_fn)(void* user, const void* send, const int64_t* send_counts, void* recv,
                                     const int64_t* recv_counts, int64_t elem_bytes);
typedef int (*glx_host_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes_per_rank);
GLX_API int glx_comm_unique_id(void* id_out /* GLX_UNIQUE_ID_BYTES */);
GLX_API int glx_comm_init_rccl(int device, int rank, int world, const void* unique_id, glx_comm** out);
GLX_API int glx_comm_init_local(int64_t fabric_key, int device, int rank, int world, glx_comm** out);
GLX_API int glx_comm_init_callbacks(int device, int rank, int world, glx_host_alltoallv_fn alltoallv,
                                    glx_host_allgather_fn allgather, void* user, glx_comm** out);
GLX_API void glx_comm_destroy(glx_comm* c);
GLX_API int glx_comm_info(const glx_comm* c, int* rank, int* world, int* device, int* transport);
/* No peer message of one exchange round exceeds this many bytes (default 512 MiB; RCCL 2.26
 * was seen to deliver only half of an all-to-all message above 1 GiB); larger exchanges are
 * cut into rounds of pointer offsets -- no staging copies.  Returns the previous value. */
GLX_API int64_t glx_comm_set_max_message_bytes(glx_comm* c, int64_t bytes);
/* all-to-all(v): send_counts[p] consecutive elements of `send` (peer-major, packed) go to rank
 * p; recv_counts[q] elements arrive from rank q into `recv` (packed in rank order).  Counts are
 * HOST arrays of `world` entries; data pointers are host or device per ptr_kind.  Collective:
 * every rank of the communicator must call it.  Device pointers: enqueued on `stream`. */
GLX_API int glx_exchange_v(glx_comm* c, const void* send, const int64_t* send_counts, void* recv,
                           const int64_t* recv_counts, int64_t elem_bytes, int ptr_kind, void* stream);
/* vals[nvals] int64 of every rank -> out[world * nvals] on every rank (rank-major), host or
 * device pointers.  The call returns when `out` is valid (it synchronises `stream`). */
GLX_API int glx_comm_allgather_i64(glx_comm* c, const int64_t* vals, int32_t nvals, int64_t* out, int ptr_kind,
                                   void* stream);
GLX_API int glx_comm_barrier(glx_comm* c, void* stream);

/* ---- distributed store: replaces DistributeRunner<Req, Res>::Run (op_runner.h:60-84) for
 * the sampling (dense samplers, FullSampler, DeepWalk), aggregating and node-lookup requests:
 * Partition (hash_partitioner.h:33-92) -> ship the
 * parts -> Process on the owning shard -> ship the results back -> Stitch
 * (stitcher.h:67-107; aggregating_request.cc:117-213) -- all on the device, between the
 * caller's request and response buffers.  SPMD: every rank calls the same entry point at the
 * same time with ITS OWN request (a rank with nothing to ask passes batch / num_ids = 0).
 * `graph` is this rank's shard of one edge type (out-edges of the vertices it owns, GLOBAL
 * destination ids, global edge ids); `features` its shard of one node type's rows, with the
 * raw ids as id map.  Either may be NULL when only the other operator family is used.
 * Results are bit-identical to the unpartitioned operator for every world size: a sampled
 * row draws from the random stream of its index in the ORIGINAL request (glx_sample_ex), and
 * aggregation reduces on the requester, in request order, over rows fetched from
 *   (1) this rank's own shard,
 *   (2) the HOT-ROW REPLICA: a copy, on every GPU, of the rows of a caller-chosen id set
 *       (glx_dist_store_set_cache; glx_dist_hot_ids picks the top-K vertices by global
 *       in-degree) -- in power-law graphs a few percent of the rows take most accesses, and
 *   (3) the HALO: the remaining remote ids, deduplicated on the device, requested from their
 *       owners (ids out, rows back: the halo-vertex feature exchange).
 * One store must not be used from two host threads at once; use one store (and one
 * communicator) per concurrently used stream. */
typedef struct glx_dist_store glx_dist_store;
GLX_API int glx_dist_store_create(glx_comm* comm, const glx_graph* graph, const glx_features* features,
                                  glx_dist_store** out);
GLX_API void glx_dist_store_destroy(glx_dist_store* st);
/* Collective.  hot_ids[n] (identical on every rank; host or device) -> every rank fetches the
 * rows from their owners once and keeps them.  n = 0 drops the replica. */
GLX_API int glx_dist_store_set_cache(glx_dist_store* st, const int64_t* hot_ids, int64_t n, float default_attr,
                                     int ptr_kind, void* stream);
/* Graph replica: complete adjacency rows of a set of (hot) vertices on this GPU, as a glx_graph the caller built with
 * explicit vertex ids -- every row with exactly the edges, weights and edge ids its owner holds (hub vertices of a
 * power-law graph are both the rows requests ask for most and cheap to hold everywhere: the hybrid-cut idea on top of
 * the hash partition of graphlearn/src/core/partition/hash_partitioner.h:33-92).  glx_dist_sample then serves the
 * request rows the replica knows locally -- same draws, the random stream is the row's index in the request -- and
 * only the rest travels to its owner.  Requests with a filter and InDegreeSampler keep the full exchange.  The store
 * borrows `replica` (NULL detaches it); the caller keeps it alive and destroys it.  Not collective. */
GLX_API int glx_dist_store_set_graph_replica(glx_dist_store* st, const glx_graph* replica);
/* Collective.  Builds such a replica from the shards themselves: every owner cuts the rows of its vertices in
 * hot_ids[n] (the same list on every rank; host or device) out of its shard -- neighbours, ITS edge ids and weights, in
 * storage order -- and the pieces are all-gathered; *out is a new glx_graph over the hot vertices (ids unknown to their
 * owner become empty rows), with its own alias tables, identical on every rank.  The caller owns it
 * (glx_graph_destroy) and attaches it with glx_dist_store_set_graph_replica.  The load-time counterpart of the
 * per-request exchange, as glx_dist_store_set_cache is for features. */
GLX_API int glx_dist_build_graph_replica(glx_dist_store* st, const int64_t* hot_ids, int64_t n, int ptr_kind,
                                         void* stream, glx_graph** out);
/* Rows of the last glx_dist_sample on this rank: all, served by the graph replica, sent to another rank. */
GLX_API int glx_dist_last_sample_rows(const glx_dist_store* st, int64_t* rows, int64_t* from_replica, int64_t* remote);
/* Collective.  The `want` destination ids with the largest in-degree summed over all shards
 * of the store's graph (ties: smaller id first), the same list on every rank, in descending
 * order of in-degree; *n_out <= want.  ids_out is a HOST array of `want` entries. */
GLX_API int glx_dist_hot_ids(glx_dist_store* st, int64_t want, int64_t* ids_out, int64_t* n_out, void* stream);
/* Collective.  InDegreeSampler (in_degree_sampler.cc:33-114) weighs a neighbour by its in-degree in the WHOLE
 * edge type (GraphStorage::GetInDegree, topo_statics.cc:62-69); a shard only sees the edges it owns.  This
 * sums the per-destination counts over all shards (pairs routed to the destination's owner, reduced there,
 * answers routed back) and builds `shard`'s per-row alias tables -- and glx_graph_in_degrees -- from the
 * totals.  `shard` must be the graph the store was created with; MUTATES it like glx_graph_enable_in_degree.
 * Until it has run, glx_dist_sample refuses GLX_SAMPLER_IN_DEGREE on a store with more than one shard. */
GLX_API int glx_dist_enable_in_degree(glx_dist_store* st, glx_graph* shard, void* stream);
/* Collective.  DistributeRunner<SamplingRequest, SamplingResponse>::Run: glx_sample_filtered's
 * arguments (filter may be NULL), request rows routed to their owners and back. */
GLX_API int glx_dist_sample(glx_dist_store* st, int sampler, const int64_t* src, int32_t batch, int32_t k,
                            int padding_mode, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                            const glx_filter* filter, int64_t* nbr_out, int64_t* eid_out, int ptr_kind,
                            void* stream);
/* Collective.  The same runner for FullSampler's sparse response (full_sampler.cc:28-97; the segments + values of
 * sampling_request.cc:198-224, stitched row by row as SamplingResponse::Stitch does for the sparse case): request rows
 * to their owners, the owners' row sizes back -> degrees_out[batch], offsets_out[batch + 1] (last entry = the total);
 * glx_dist_sample_full additionally brings the values back, row i at offsets_out[i] of nbr_out / eid_out, which hold
 * `capacity` entries (InvalidArgument when the response is larger: ask _sizes first).  Host or device pointers; both
 * return when their outputs are valid. */
GLX_API int glx_dist_sample_full_sizes(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit,
                                       int32_t* degrees_out, int64_t* offsets_out, int ptr_kind, void* stream);
GLX_API int glx_dist_sample_full(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit,
                                 int32_t* degrees_out, int64_t* offsets_out, int64_t* nbr_out, int64_t* eid_out,
                                 int64_t capacity, int ptr_kind, void* stream);
/* glx_dist_sample_full with a filter (full_sampler.cc:66-84 behind DistributeRunner): every row's filter value travels
 * with its id, the owners answer with glx_sample_full_filtered -- same sizes as without a filter, reserved neighbours
 * padded up to them, rows whose neighbours all hit default-filled.  Every rank passes a filter of the same kind (or none). */
GLX_API int glx_dist_sample_full_filtered(glx_dist_store* st, const int64_t* src, int32_t batch, int32_t max_limit,
                                          int32_t* degrees_out, int64_t* offsets_out, int padding_mode,
                                          int64_t default_neighbor_id, const glx_filter* filter, int64_t* nbr_out,
                                          int64_t* eid_out, int64_t capacity, int ptr_kind, void* stream);
/* Collective.  GetDegree for DESTINATION ids over the shards (GraphStorage::GetInDegree, topo_statics.cc:62-69;
 * degree_getter.cc): degrees_out[n] = edges pointing to ids[i] summed over ALL shards (held by the id's owner,
 * llabs(id) % P, in a table built collectively on first use); ids nobody points to: 0. */
GLX_API int glx_dist_in_degrees(glx_dist_store* st, const int64_t* ids, int32_t n, int32_t* degrees_out, int ptr_kind,
                                void* stream);
/* Collective.  The negative samplers' candidate list over the WHOLE edge type on every rank
 * (random_negative_sampler.cc:30-63, in_degree_negative_sampler.cc:29-135: GetAllDstIds() / GetAllInDegrees() of the
 * unpartitioned storage): every destination id of any shard, ids ascending, uniform (by_in_degree = 0) or weighted by
 * the in-degree summed over all shards.  The result is an ordinary glx_negative, identical on every rank and equal to
 * glx_negative_create(ids ascending, global in-degrees) on an unpartitioned store; destroy with glx_negative_destroy. */
GLX_API int glx_dist_negative_create(glx_dist_store* st, int by_in_degree, void* stream, glx_negative** out);
/* glx_negative_sample across the shards.  GLX_NEG_EXCLUDE_NEIGHBORS is collective: the rows travel to the owners of their
 * source ids (whose adjacency is the exclusion set; the shard needs glx_graph_enable_negative) and draw there from their
 * ORIGINAL row's random stream; the other modes run locally.  Every rank passes the same `table` contents and `count`;
 * answers equal the unpartitioned glx_negative_sample on that table for every shard count. */
GLX_API int glx_dist_negative_sample(glx_dist_store* st, const glx_negative* table, int exclude, const int64_t* src,
                                     int32_t batch, int32_t count, int64_t default_neighbor_id, uint64_t seed,
                                     uint64_t call_counter, int64_t* out, int ptr_kind, void* stream);
/* Collective.  The RandomWalk operator (random_walk.cc:30-276) across the shards, glx_random_walk's layout and draws.
 * DeepWalk (p = q = 1 as RandomWalkRequest::IsDeepWalk decides, :168-190): step t is one partitioned RandomSampler request
 * with neighbor_count 1 and call counter call_counter + t, walker i drawing from stream i.  node2vec (:192-272): per step
 * ONE partitioned FullSampler request (limit full_nbr_num) brings the first full_nbr_num neighbours of the vertices the
 * walkers stand on -- with their edge weights -- to the requesters; the parents' lists are the previous step's; the step
 * (bias by 1/p, 1, 1/q, alias table, one draw) then runs on the requester with the single store's arithmetic: walks equal
 * glx_random_walk on the unpartitioned graph for every shard count.  glx_dist_random_walk uses the reference's flag
 * defaults (DefaultFullNbrNum 100, DefaultWeight 0: config.cc:102,111), _ex takes them. */
GLX_API int glx_dist_random_walk(glx_dist_store* st, const int64_t* seeds, int32_t batch, int32_t walk_len, float p, float q,
                                 int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter, int64_t* walks_out,
                                 int ptr_kind, void* stream);
GLX_API int glx_dist_random_walk_ex(glx_dist_store* st, const int64_t* seeds, int32_t batch, int32_t walk_len, float p,
                                    float q, int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id,
                                    uint64_t seed, uint64_t call_counter, int64_t* walks_out, int ptr_kind, void* stream);
/* Collective.  DistributeRunner<AggregatingRequest, AggregatingResponse>::Run with
 * glx_aggregate's arguments; segment_ids == NULL means num_segments equal segments of
 * num_ids / num_segments consecutive ids (a dense sampler response). */
GLX_API int glx_dist_aggregate(glx_dist_store* st, int op, const int64_t* node_ids, const int32_t* segment_ids,
                               int32_t num_ids, int32_t num_segments, float default_attr, float* emb_out,
                               int32_t* cnt_out, int ptr_kind, void* stream);
/* Collective.  The same request served the way the reference's servers serve it (SURVEY 8(e) design R, kept as the
 * ablation of the halo exchange above): AggregatingRequest::Partition routes every (id, segment id) to the id's owner
 * (aggregating_request.cc:117-170), each owner runs Aggregator::Aggregate over what it received, and
 * AggregatingResponse::Stitch folds the world-size partial [num_segments, dim] results and counts on the requester
 * (:172-213; glx_aggregate_stitch).  Per request it moves 12 B per remote id out and world * num_segments * (4 dim + 4) B
 * back instead of 4 dim B per distinct remote id: cheaper when segments are long (fan-out >> world).  Max / Min / counts
 * equal the single-store operator exactly; Sum / Mean / Prod fold per-shard partial results, so they differ from it by
 * the floating-point reassociation the reference's distributed mode has as well (<= 1e-5 relative at the fan-outs of
 * BASELINE's configs).  A segment none of whose ids a shard owns contributes nothing from that shard (SURVEY 8(a)
 * quirk 8: the reference folds a default row in there).  Replicas (hot rows, graph) are not consulted. */
GLX_API int glx_dist_aggregate_partial(glx_dist_store* st, int op, const int64_t* node_ids, const int32_t* segment_ids,
                                       int32_t num_ids, int32_t num_segments, float default_attr, float* emb_out,
                                       int32_t* cnt_out, int ptr_kind, void* stream);
/* The same in two halves, for software pipelining (device pointers only).  _begin is the collective part:
 * it resolves the ids and fetches the halo rows into buffer set `slot` (0 <= slot < GLX_DIST_SLOTS); it may run
 * on the stream that PRODUCED the ids, one or more requests ahead of the reduce.  _end is local: the segmented
 * reduce over the slot's rows, on any stream ordered after _begin's.  The caller keeps a slot untouched between
 * its _begin and the completion of its _end.  glx_dist_aggregate == _begin + _end on slot 0.  Why: with the
 * halo exchange of request i+1 running beside the HBM-bound reduce of request i, link time and the probe passes
 * leave the critical path. */
#define GLX_DIST_SLOTS 8
GLX_API int glx_dist_aggregate_begin(glx_dist_store* st, int32_t slot, const int64_t* node_ids, int32_t num_ids,
                                     float default_attr, void* stream);
GLX_API int glx_dist_aggregate_end(glx_dist_store* st, int32_t slot, int op, const int32_t* segment_ids,
                                   int32_t num_segments, float* emb_out, int32_t* cnt_out, void* stream);
/* ABI 3.  The reduce over ids [first_id, first_id + num_ids) of the request begun in `slot`: ONE _begin (one count
 * exchange, one deduplicated halo fetch) can serve several aggregating requests whose ids were handed over back to back
 * -- the hop-2 and the hop-1 neighbours of one sampling step.  release != 0 ends the slot's request (as _end does). */
GLX_API int glx_dist_aggregate_end_range(glx_dist_store* st, int32_t slot, int32_t first_id, int32_t num_ids, int release,
                                         int op, const int32_t* segment_ids, int32_t num_segments, float* emb_out,
                                         int32_t* cnt_out, void* stream);
/* Collective.  LookupNodes in distributed mode (node_lookuper.cc:24-52 behind
 * DistributeRunner): out[n * dim] rows of ids owned by any shard. */
GLX_API int glx_dist_lookup(glx_dist_store* st, const int64_t* node_ids, int64_t n, float default_attr,
                            float* out, int ptr_kind, void* stream);
/* Where the ids of this store's LAST glx_dist_aggregate / glx_dist_lookup came from, and what
 * crossed the links because of it (this rank's view).  The call synchronises nothing: read it
 * after the stream has drained. */
typedef struct glx_dist_stats {
  int64_t ids;             /* request size */
  int64_t from_replica;    /* served by the hot-row replica */
  int64_t from_own_shard;  /* owned by this rank (or unknown everywhere: default row) */
  int64_t remote;          /* remote and not replicated (with repeats) */
  int64_t remote_distinct; /* ... distinct: the ids sent, = halo rows received */
  int64_t served_rows;     /* halo rows this rank gathered for its peers */
  int64_t bytes_sent;      /* ids out + rows out over the transport, self-copies excluded */
  int64_t bytes_received;
  int64_t exchange_rounds; /* transport rounds of the last row exchange (message-size limit) */
  /* ABI 3.  Cumulative since the store was created: every partitioned request exchanges its per-shard counts once and the
   * calling host thread waits there for the slowest rank (the reference's RunInParallel waits for every remote shard's RPC
   * the same way, op_runner.h:86-117) -- how many such waits, and how long they took in total. */
  int64_t host_syncs;
  int64_t host_stall_us;
} glx_dist_stats;
GLX_API int glx_dist_last_stats(const glx_dist_store* st, glx_dist_stats* out);

/* ---- ABI 4: partitioned sampling WITHOUT a count exchange.  DistributeRunner waits for every shard's reply before it
 * goes on (RunInParallel, core/runner/op_runner.h:86-117); glx_dist_sample does the same once per request, at the count
 * exchange that tells the ranks how many rows each message carries -- a blocking host wait per hop.  A ledger removes it
 * for calls whose place in the step repeats: the i-th glx_dist_sample call after a confirmation point goes through the
 * count exchange once and records the largest bucket any rank sent to any owner; later calls at position i send
 * FIXED-capacity messages (that bucket x 1.25 + 1024 rows, padded with ids no shard knows), the owners answer every slot,
 * and each requester keeps the answers of its real rows -- no count leaves the device, the host never waits.  Whether a
 * call speculates depends only on its position and on what the ranks learned together, never on this rank's request
 * length: a rank with a shorter tail batch (or an empty request) enters the same collectives as its peers.  Whether every bucket fitted its
 * message is recorded in a device word and travels with the NEXT count exchange of any store attached to the same
 * ledger (the aggregation's, glx_dist_aggregate_begin; or glx_dist_confirm): that call returns GLX_ABORTED on every rank
 * when any bucket of any rank did not fit, and the results of all speculated calls since the previous successful
 * exchange are void -- redo them (same call counters: same answers); the capacities have been raised to what was
 * needed, so the repeat fits (voided calls still in flight on other streams need no draining: an abort starts a new
 * epoch and their flags are not heeded).  Results of confirmed calls are bit-identical to the count-exchange path.
 * Contract (SPMD lockstep): every rank issues the same SEQUENCE of glx_dist_sample calls and confirmation points, with
 * the same neighbor_count, sampler, padding, seed and call_counter -- the owners serve all requesters with their OWN
 * parameters; request lengths may differ between the ranks.  (Ranks that issue different sequences of calls enter
 * different collectives: that is a hang on RCCL, with or without a ledger.)
 * A digest of those parameters travels with the confirmation; ranks that disagree get GLX_ABORTED and the ledger stops
 * speculating for good.  Filtered requests, glx_dist_sample_full and random walks always take the count exchange.
 * One ledger per rank, shared by that rank's stores; calls that use it come from one host thread. */
typedef struct glx_dist_ledger glx_dist_ledger;
GLX_API int glx_dist_ledger_create(int device, glx_dist_ledger** out);
GLX_API void glx_dist_ledger_destroy(glx_dist_ledger* l);
/* NULL detaches.  The ledger must outlive the stores attached to it. */
GLX_API int glx_dist_store_set_ledger(glx_dist_store* st, glx_dist_ledger* l);
/* A confirmation point of its own (collective; one count exchange on st's communicator): GLX_OK = every speculated
 * call enqueued before it, on streams ordered before `stream`, is valid. */
GLX_API int glx_dist_confirm(glx_dist_store* st, void* stream);
typedef struct glx_dist_ledger_stats {
  int64_t speculated;  /* glx_dist_sample calls that skipped their count exchange */
  int64_t learned;     /* ... that took it and recorded a request shape */
  int64_t aborted;     /* confirmations that failed */
  int64_t holding;     /* 1: the ranks' requests disagreed once; no more speculation */
  double largest_share; /* largest per-owner share of a request recorded so far */
} glx_dist_ledger_stats;
GLX_API int glx_dist_ledger_get_stats(const glx_dist_ledger* l, glx_dist_ledger_stats* out);
/* Test knob: capacity = share x slack + pad rows (defaults 1.25, 1024).  slack < 1 forces the abort path. */
GLX_API int glx_dist_ledger_set_slack(glx_dist_ledger* l, double slack, int64_t pad_rows);

/* ---- request plans: a multi-hop sample (+ aggregate) request as ONE graph launch.  Replaces the
 * per-batch walk over a chain of DAG nodes (core/runner/dag_node_runner.cc:32-109: each node builds a
 * request, runs its operator, feeds the next) and the hop loop of NeighborSampler.get
 * (python/sampler/neighbor_sampler.py:93-127) for fixed-shape requests. -------------------------
 * glx_plan_create captures, for `batch` seeds: hop h = glx_sample(graphs[h], sampler, hop h-1's
 * neighbours, fanouts[h], ...) with random stream (seed, call_counter + h) -- exactly glx_sample_hops --
 * and, when `features` is given, for every hop h (deepest first) glx_aggregate(features[h], agg_op) of hop
 * h's neighbours into the fanouts[h]-sized segments of hop h-1's rows (features[h] = the node type hop h
 * reaches).  The kernels are recorded once into a hipGraph; glx_plan_run(seeds, call_counter) replays it on
 * `stream`: one launch instead of 2 * num_hops, which is what a small batch (B0 <= 8192) needs -- its kernels
 * take microseconds, launching them one by one takes longer.  Outputs live in plan-owned device buffers
 * (glx_plan_output) valid until the next run; results are bit-identical to the separate calls.  A plan is
 * bound to one device and must not run concurrently with itself. */
typedef struct glx_plan glx_plan;
GLX_API int glx_plan_create(const glx_graph* const* graphs, int32_t num_hops, int sampler, const int32_t* fanouts,
                            int32_t batch, int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                            const glx_features* const* features, int agg_op, float default_attr, glx_plan** out);
/* seeds[batch]: DEVICE pointer, read when the graph executes. */
GLX_API int glx_plan_run(glx_plan* p, const int64_t* seeds, uint64_t call_counter, void* stream);
/* Device buffers of hop `hop`: nbr / eid [rows * fanout], emb [rows * dim] and cnt [rows] (NULL without
 * aggregation); rows = request rows of the hop.  Any out pointer may be NULL. */
GLX_API int glx_plan_output(const glx_plan* p, int32_t hop, int64_t** nbr, int64_t** eid, float** emb, int32_t** cnt,
                            int64_t* rows, int32_t* fanout);
GLX_API void glx_plan_destroy(glx_plan* p);

/* ---- kernel timing: the device-side counterpart of the reference's
 * PROFILING(key) scope timers (common/base/profiling.h:24-71). ------------
 * While enabled (per host thread), every glx_sample / glx_aggregate /
 * glx_lookup call brackets its dominant kernel -- and only that kernel -- with
 * a pair of HIP events on the call's stream.  glx_profile_collect synchronises
 * those events, writes the launch durations of `kind` (oldest first, at most
 * `cap`) to ms_out, stores their number in *count and forgets them. */
#define GLX_KERNEL_SAMPLE 0
#define GLX_KERNEL_AGGREGATE 1
#define GLX_KERNEL_LOOKUP 2
GLX_API int glx_profile_enable(int on);
GLX_API int glx_profile_collect(int kind, float* ms_out, int32_t cap, int32_t* count);

/* ---- conditional negative sampling: replaces ConditionalNegativeSampler (core/operator/sampler/
 * conditional_negative_sampler.cc:37-161) with its ConditionTable (condition_table.cc:65-148) and AttributeNodesMap
 * (attribute_nodes_map.h:74-127). ----------------------------------------------------------------------------
 * glx_cond_table_create: the candidates ids[num_ids] (the edge type's destination ids in first-appearance order, or a
 *   node type's ids: StorageWrapper::GetIds) with weights[num_ids] (in-degrees / node weights; NULL = 1 each, the
 *   "random" strategy) and, per selected attribute column c, the candidates' attribute values as int64 keys
 *   cand_keys[c * num_ids + u] (an int attribute itself, a float's bit pattern with -0 folded onto +0, a string's
 *   dictionary index: equal keys <=> equal attribute values).  Groups candidates by key per column and builds one alias
 *   table per group plus the default table over all candidates -- the reference's lazily built, per-type cached
 *   ConditionTable + AliasMethod pair -- on the device.
 * glx_cond_negative_sample: out[batch * count].  Row i is sampled AFTER rows 0..i-1: the exclusion set grows through the
 *   request (all dst ids up front when batch_share; else the neighbours of src i in `g` -- may be NULL -- and dst i are
 *   added before row i samples; `unique` also adds every accepted id) exactly as the reference's nbr_set does.  Per
 *   column c the row takes (int32)(count * props[c]) ids from the group whose key is dst_keys[i * num_cols + c]
 *   (INT64_MIN: no such group) with AttributeNodesMap::Sample's retry schedule (retry_times = GLOBAL_FLAG(
 *   SamplingRetryTimes)); the rest of the row comes from the default table, then default_neighbor_id.  props is a HOST
 *   array of num_cols floats; the other pointers follow ptr_kind.  Draws: stream (seed, call_counter, i); the layout is
 *   spelled out in DESIGN.md section 5.  The call returns when `out` is valid. */
typedef struct glx_cond_table glx_cond_table;
GLX_API int glx_cond_table_create(int device, int64_t num_ids, const int64_t* ids, const float* weights, int32_t num_cols,
                                  const int64_t* cand_keys, int ptr_kind, void* stream, glx_cond_table** out);
GLX_API void glx_cond_table_destroy(glx_cond_table* t);
GLX_API int glx_cond_negative_sample(const glx_cond_table* t, const glx_graph* g, const int64_t* src, const int64_t* dst,
                                     const int64_t* dst_keys, const float* props, int32_t batch, int32_t count,
                                     int batch_share, int unique, int32_t retry_times, int64_t default_neighbor_id,
                                     uint64_t seed, uint64_t call_counter, int64_t* out, int ptr_kind, void* stream);

/* ---- induced sub-graph: replaces SubGraphSampler::InduceSubGraph (core/operator/subgraph/subgraph_sampler.cc:34-95).
 * nodes[n] is the sub-graph's node list (SubGraphSampler::Process, subgraph_sampler.h:36-78: the seeds, then the sorted set
 * of every neighbour the hop-wise FullSampler calls returned -- duplicates between the two parts are kept, as in the
 * reference); offsets[n + 1] / nbr / eid are FullSampler's response for `nodes` with limit GLOBAL_FLAG(DefaultFullNbrNum)
 * (glx_sample_full, or glx_dist_sample_full across shards).  For every node i, in order, and every j in list order whose
 * id is among row i's neighbours, the entries (i, j, e) and (j, i, e) are appended, e = the edge id of the LAST slot of row
 * i holding that neighbour (node2edge[nbrs[k]] = edge_ids[k], :60-64).  *count_out (host) = the number of entries, 2 per
 * match; at most `capacity` are written (capacity 0: count only).  The call returns when the outputs are valid. */
GLX_API int glx_subgraph_induce(int device, const int64_t* nodes, int32_t n, const int64_t* offsets, const int64_t* nbr,
                                const int64_t* eid, int32_t* row_out, int32_t* col_out, int64_t* eid_out, int64_t capacity,
                                int64_t* count_out, int ptr_kind, void* stream);

/* ---- memory-system probes: the measured ceilings the kernels above are priced against -- an addition of this
 * engine with no counterpart in the reference (its PROFILING timers, common/base/profiling.h:24-71, time scopes; they
 * do not measure the machine).  Each probe allocates its own buffers on `device`, runs 2 untimed + `reps` timed
 * launches between two HIP events on `stream` and reports the bytes one launch moves and its average duration.
 *   GLX_PROBE_STREAM_READ  reads a `bytes` buffer once (16 B per lane, grid-strided)          moved = bytes
 *   GLX_PROBE_COPY         b = a                                                             moved = 2 * bytes
 *   GLX_PROBE_TRIAD        a = b + s * c  (STREAM triad)                                     moved = 3 * bytes
 *   GLX_PROBE_GATHER32     `units` aligned 32-byte records from uniformly random positions of a `bytes` table, 16 B
 *                          written per record: EdgeWeightSampler's access per output slot (alias_method.cc:117-121 on
 *                          one packed record)                                                moved = 48 * units
 *   GLX_PROBE_GATHER_ROWS  `units` rows of `unit_bytes` from uniformly random rows of a `bytes` table, reduced ten to
 *                          one in registers: the gather of Aggregator::Aggregate (aggregator.cc:25-59) without cache
 *                          reuse                                                             moved = 1.1 * units * unit_bytes */
#define GLX_PROBE_STREAM_READ 0
#define GLX_PROBE_COPY 1
#define GLX_PROBE_TRIAD 2
#define GLX_PROBE_GATHER32 3
#define GLX_PROBE_GATHER_ROWS 4
GLX_API int glx_probe_bandwidth(int device, int kind, int64_t bytes, int64_t units, int32_t unit_bytes, int32_t reps,
                                double* moved_bytes_out, double* avg_ms_out, void* stream);

/* Measurement knobs of the aggregation launch (Aggregator::Aggregate, aggregator.cc:25-59), for A/B probes inside one
 * process.  Each is read from the environment once (first use) and may be set here at any time; 0 restores the
 * product's default.  Names: "agg_mfma" (GLX_AGG_MFMA), "agg_unroll" (GLX_AGG_UNROLL), "agg_slices" (GLX_AGG_SLICES),
 * "agg_legacy" (GLX_AGG_LEGACY), "agg_segs" (GLX_AGG_SEGS), "agg_xcd_slices" (GLX_AGG_XCD_SLICES), "agg_occupancy"
 * (GLX_AGG_OCCUPANCY), "agg_store" (GLX_AGG_STORE).  Results are
 * bit-identical under every setting.  Unknown name: GLX_INVALID_ARGUMENT.
 * Test knobs of the side paths, same rules, -1 restores the default: "cond_sequential" (GLX_COND_SEQUENTIAL),
 * "dist_no_bitmap" (GLX_DIST_NO_BITMAP), "filter_span_cap" (GLX_FILTER_SPAN_CAP), "filter_dedup_min_rows"
 * (GLX_FILTER_DEDUP_MIN_ROWS), "idmap_hash_only" (GLX_IDMAP_HASH_ONLY: feature tables created afterwards keep a hash table
 * even when their ids are an arithmetic progression); of the partitioned aggregation's resolve pass: "resolve_ids"
 * (GLX_RESOLVE_IDS: ids per thread per pass), "resolve_blocks" (GLX_RESOLVE_BLOCKS: workgroups), "resolve_set_share"
 * (GLX_RESOLVE_SET_SHARE: the halo id set holds at least n / 1024 of a request's ids), "resolve_peek" (GLX_RESOLVE_PEEK = 0:
 * no plain load of a set slot before the compare-and-swap), "resolve_own_first" (GLX_RESOLVE_OWN_FIRST = 1 | 0: ids this
 * rank owns skip / take the replica lookup; default: skip at world size 1 only).  Test aid: "seg_epochs_before_wrap" (the
 * calling thread's segment-word buffers hand out `value` more epochs before their counter wraps). */
GLX_API int glx_tune(const char* name, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* GLX_H_ */
