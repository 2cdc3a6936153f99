/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the graph-learn sampling/aggregation hot path under the
 * glx seeding contract ("Seeding contract").  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product library (graph-learn_amd/csrc) never does.
 *
 * Parity pinning: the deterministic functions here (topk, padding,
 * default-fill, alias build, aggregators, partition/stitch) are checked
 * bit-exactly against the real reference (oracle/_ref, built from the
 * reference's own sources) and against the reference's known-answer tests
 * (tests/golden/); the random samplers are checked distributionally against
 * the reference because the reference itself is unseeded
 * (random_sampler.cc:46-47) -- and, since round 4, DRAW FOR DRAW: with
 * glxo_set_reference_entropy() the same row code consumes the reference's own
 * variates (sequential MT19937 + libstdc++'s distributions) and its outputs
 * equal oracle/_ref's bit for bit (tests/test_oracle_refseq.py); the filtered
 * samplers and the RandomWalk operator are checked distributionally
 * (tests/golden/filtered.npz, walk.npz: the reference's own filter.cc /
 * random_walk.cc built into oracle/_ref).
 */
#ifndef GLX_ORACLE_H_
#define GLX_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GLXO_RANDOM = 0, GLXO_RANDOM_WITHOUT_REPLACEMENT = 1, GLXO_EDGE_WEIGHT = 2, GLXO_TOPK = 3,
       GLXO_IN_DEGREE = 4 };
enum { GLXO_SUM = 0, GLXO_MEAN = 1, GLXO_MAX = 2, GLXO_MIN = 3, GLXO_PROD = 4 };
enum { GLXO_PAD_REPLICATE = 0, GLXO_PAD_CIRCULAR = 1 };

/* Philox4x32-10 block: ctr[4], key[2] -> out[4]. */
void glxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* The 64-bit draw j of row `row` in call (seed, call_counter). */
uint64_t glxo_draw64(uint64_t seed, uint64_t call_counter, uint32_t row, uint32_t j);

/* Host CSR in the layout the reference exposes after Build():
 *   row r = neighbours col[row_ptr[r] .. row_ptr[r+1]) with edge ids eid[..].
 *   ids == NULL: raw vertex id v IS row v (0 <= v < V); otherwise ids[r] is the
 *   raw id of row r (the reference's AutoIndex, auto_indexing.cc:21-33). */
typedef struct {
  int64_t V, E;
  const int64_t* row_ptr;
  const int64_t* col;
  const int64_t* eid;
  const float* weight;     /* per CSR slot; may be NULL */
  const float* alias_prob; /* per CSR slot; from glxo_alias_build; may be NULL */
  const int32_t* alias_idx;
  const int64_t* ids;      /* may be NULL */
  /* alias tables over float(in-degree of the neighbour) for InDegreeSampler
   * (in_degree_sampler.cc:79-92); from glxo_in_degree_weights + glxo_alias_build */
  const float* indeg_prob;
  const int32_t* indeg_alias;
} glxo_graph;

/* Restates AliasMethod::Build (alias_method.cc:57-107) per CSR row. */
void glxo_alias_build(const int64_t* row_ptr, const float* weight, int64_t V, float* prob_out,
                      int32_t* alias_out);

/* In-degree of every slot's neighbour id inside this edge type, as float, i.e. the
 * weights InDegreeSampler::SampleFrom gathers (in_degree_sampler.cc:79-92 via
 * GraphStorage::GetInDegree, topo_statics.cc:33-69). */
void glxo_in_degree_weights(const int64_t* col, int64_t E, float* w_out);

/* Restates FullSampler::Sample (full_sampler.cc:28-97): row i yields its first
 * min(limit, deg) neighbours in storage order (limit <= 0: all); degrees_out[batch];
 * values appended to nbr_out / eid_out (capacity cap).  Returns the total count. */
int64_t glxo_sample_full(const glxo_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                         int32_t* degrees_out, int64_t* nbr_out, int64_t* eid_out, int64_t cap);

/* ---- sampling filters (core/operator/sampler/filter.{h,cc}) -----------------------
 * type / field use the reference's enum values (include/constants.h:135-145).  values[]
 * is the request's filter tensor AFTER Filter::FillValues (filter.cc:53-67): one value per
 * request row.  ts_slot = GetEdgeTimestamp of every CSR slot (NULL: default_timestamp
 * everywhere); indeg_weight = glxo_in_degree_weights (InDegreeSampler only). */
enum { GLXO_FILTER_NONE = 0, GLXO_FILTER_EQUAL = 1, GLXO_FILTER_LARGER_THAN = 2 };
enum { GLXO_FIELD_NONE = 0, GLXO_FIELD_ID = 1, GLXO_FIELD_TIMESTAMP = 2 };
typedef struct {
  int type, field;
  const int64_t* values;
  int32_t retry_times;       /* GLOBAL_FLAG(SamplingRetryTimes), RandomSampler only */
  const int64_t* ts_slot;
  int64_t default_timestamp;
  const float* indeg_weight;
} glxo_filter;

/* Restates Filter::ActOn (filter.cc:69-96) for request row `batch_idx` over a row of n
 * neighbours: indices[] (capacity n) receives the reserved positions in the reference's
 * order; returns their number.  Timestamp + LARGER_THAN takes the binary-search path
 * (FindkthLargest, filter.cc:196-229), which reads values[0] for EVERY row (the default
 * batch_share_idx = 0, filter.h:107-111) and returns the positions descending. */
int32_t glxo_filter_act_on(const glxo_filter* f, int32_t batch_idx, const int64_t* row_nbr,
                           const int64_t* row_ts, int32_t n, int32_t* indices);

/* The samplers with a filter set (random_sampler.cc:52-71, topk_sampler.cc:52-61,
 * random_without_replacement_sampler.cc:56-68, edge_weight_sampler.cc:55-66,94-112,
 * in_degree_sampler.cc:54-65,94-113).  Same seeding contract as glxo_sample; RandomSampler's
 * retry a of slot j uses draw j + a * k.  EdgeWeight / InDegree build the alias table of
 * the reserved neighbours' weights per row, as the reference does. */
int glxo_sample_filtered(const glxo_graph* g, int op, const int64_t* src, const int64_t* rng_rows,
                         int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id,
                         uint64_t seed, uint64_t call_counter, const glxo_filter* f, int64_t* nbr_out,
                         int64_t* eid_out);
/* FullSampler with a filter (full_sampler.cc:66-84): the row sizes stay the UNFILTERED
 * min(limit, deg) and the reserved neighbours are padded up to that size.  A row whose
 * neighbours are all filtered out yields that many default ids (the reference's
 * FillWith(dim2) breaks the ragged layout there: SURVEY 8(a) quirk 12). */
int64_t glxo_sample_full_filtered(const glxo_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                                  int padding_mode, int64_t default_neighbor_id, const glxo_filter* f,
                                  int32_t* degrees_out, int64_t* nbr_out, int64_t* eid_out, int64_t cap);

/* Restates the "RandomWalk" operator (core/operator/random_walk/random_walk.cc:30-276):
 * DeepWalk (:168-190) when p = q = 1 within 32 FLT_EPSILON, node2vec otherwise
 * (WeightedRandomWalkKernel :228-272: biased weights over the first min(deg, full_nbr_num)
 * neighbours, AliasMethod over them, one draw).  walks_out[batch * walk_len] row-major.
 * Step t of walker i uses draw 0 of stream (seed, call_counter + t, i).  The parents' neighbour
 * lists are walked with the reference's cursor (:214-226: not advanced for a walker whose current
 * vertex has no out-edges, so the walkers behind it read a window that starts too early). */
int glxo_random_walk(const glxo_graph* g, const int64_t* seeds, int32_t batch, int32_t walk_len, float p, float q,
                     int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id, uint64_t seed,
                     uint64_t call_counter, int64_t* walks_out);
/* The biased weights of one node2vec step (exposed for the distribution tests). */
void glxo_node2vec_weights(const glxo_graph* g, int64_t cur, int64_t parent, int has_parent_nbrs, float p, float q,
                           int32_t full_nbr_num, float default_weight, float* w_out, int32_t* n_out);

/* Restates MemoryAdjMatrix::Sort (memory_adj_matrix.cc:105-125): each row by
 * weight descending.  Ties keep insertion order (the reference's std::sort
 * leaves tie order unspecified).  Sorts col/eid/weight in place. */
void glxo_sort_rows_by_weight_desc(const int64_t* row_ptr, int64_t V, int64_t* col, int64_t* eid,
                                   float* weight);

/* Restates MemoryAdjMatrix::SortByTimestamp (memory_adj_matrix.cc:129-148): each row by the
 * edges' timestamps ascending (ts_slot[] is per CSR slot); ties keep insertion order.  Sorts
 * col/eid/ts_slot (and weight when not NULL) in place. */
void glxo_sort_rows_by_timestamp_asc(const int64_t* row_ptr, int64_t V, int64_t* col, int64_t* eid,
                                     int64_t* ts_slot, float* weight);

/* The four samplers (random_sampler.cc:33-76,
 * random_without_replacement_sampler.cc:31-75, edge_weight_sampler.cc:31-92 +
 * alias_method.cc:109-124, topk_sampler.cc:29-68) with the padders
 * (padder/circular_padder.h:36-66, padder/replicate_padder.h:37-56).
 * Returns 0, or 3 (InvalidArgument) for a bad op / missing alias table. */
/* rng_rows (may be NULL): request row i draws from the stream of row rng_rows[i]
 * (used when a shard serves a slice of a partitioned request). */
int glxo_sample(const glxo_graph* g, int op, const int64_t* src, const int64_t* rng_rows,
                int32_t batch, int32_t k,
                int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                uint64_t call_counter, int64_t* nbr_out, int64_t* eid_out);

/* Timing aid for bench.py's cpu_baseline of kind "port" (used only when the
 * reference build oracle/_ref is not available): when on, EdgeWeight rebuilds the
 * row's alias table from the weights on EVERY request row, which is the cost
 * structure of the reference (edge_weight_sampler.cc:78-92 calls AliasMethod's
 * constructor per row per request).  Results are unchanged. */
void glxo_set_reference_cost_model(int on);

/* L1 entropy (tests/test_oracle_refseq.py): when on, glxo_sample draws the way the reference does -- one sequential
 * MT19937 per sampler source file, seeded with `seed` at its first use after this call, through libstdc++ 11's
 * uniform_int_distribution / std::shuffle / uniform_real_distribution -- instead of the contract's per-row Philox
 * streams (seed / call_counter / rng_rows are then ignored).  Outputs equal oracle/_ref's draw for draw when its
 * random_device is pinned to the same seed and the requests are issued in the same order in one thread.  Not
 * thread-safe (process-wide engines, like the mode switch).  See glx_oracle.c for the restated algorithms. */
void glxo_set_reference_entropy(int on, uint32_t seed);

/* Restates Aggregator::Aggregate (aggregator.cc:25-59) with the Sum/Mean/Max/
 * Min/Prod Init/Agg/Final functions.  feats is [V, dim] row-major; ids as in
 * glxo_graph.ids (NULL = dense). Unknown id -> a row of default_attr
 * (memory_node_storage.cc:127-138, element_value.cc:26-50). */
int glxo_aggregate(const float* feats, int64_t V, int32_t dim, const int64_t* ids, int op,
                   const int64_t* node_ids, const int32_t* segment_ids, int32_t num_ids,
                   int32_t num_segments, float default_attr, float* emb_out, int32_t* cnt_out);

/* Restates TopoStatics::Add (topo_statics.cc:32-55) as seen through
 * GetAllDstIds()/GetAllInDegrees(): distinct destination ids in first-appearance order
 * (edges taken in edge-id = insertion order) and their in-degrees.  col/eid: the E CSR
 * slots.  Returns the number of distinct ids; fills ids_out / in_degrees_out (capacity E). */
int64_t glxo_dst_statics(const int64_t* col, const int64_t* eid, int64_t E, int64_t* ids_out,
                         int32_t* in_degrees_out);

/* Restates the negative samplers' sampling loops under the seeding contract:
 * RandomNegativeSampler (random_negative_sampler.cc:55-61; exclude 0, alias NULL),
 * SoftInDegreeNegativeSampler (in_degree_negative_sampler.cc:107-121; exclude 0),
 * InDegreeNegativeSampler (in_degree_negative_sampler.cc:57-92; exclude 1: src's neighbours)
 * and NodeWeightNegativeSampler (node_weight_negative_sampler.cc:58-92; exclude 2: the
 * request's own ids).  ids[U] candidates; prob/alias: the global alias table or NULL
 * (uniform).  Draw `blk * count + j` of row i's stream is candidate j of retry block blk. */
int glxo_negative_sample(const int64_t* ids, int64_t U, const float* prob, const int32_t* alias, int exclude,
                         const glxo_graph* g, const int64_t* src, int32_t batch, int32_t count,
                         int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter, int64_t* out);

/* Restates HashPartitioner::Partition (hash_partitioner.h:33-92): shard =
 * llabs(id) % P, stable within a shard.  order_out[n]: indices grouped by
 * shard (the concatenated Sticker lists); counts_out[P]. */
void glxo_partition(const int64_t* ids, int64_t n, int32_t P, int64_t* order_out,
                    int64_t* counts_out);
/* Restates AggregatingResponse::Stitch (aggregating_request.cc:172-213) with the
 * aggregators' AggFunc/FinalFunc (sum_/mean_/max_/min_/prod_aggregator.cc):
 * parts[P][num_segments*dim] + cnts[P][num_segments] partial results -> one result.
 * Fold order = shard order, starting from InitFunc's value; Mean re-weights each
 * partial mean by its count and divides by the total (mean_aggregator.cc:31-37,45-61).
 * ONE deliberate difference: a shard that saw none of a
 * segment's ids (count 0) is skipped instead of folding its DefaultFloatAttribute
 * row in, so Max/Min/Prod equal the single-shard answer.  `reference_fold` != 0
 * restores the reference's behaviour (used to pin this function against it). */
int glxo_aggregate_stitch(int op, int32_t P, const float* parts, const int32_t* cnts,
                          int32_t num_segments, int32_t dim, float default_attr, int reference_fold,
                          float* emb_out, int32_t* cnt_out);
/* Restates Stitcher::DoStitch (stitcher.h:67-107) for dense row payloads:
 * out[order[i]] = shard_major[i], each row `width` int64s. */
void glxo_stitch_i64(const int64_t* shard_major, const int64_t* order, int64_t n, int32_t width,
                     int64_t* out);

/* Restates ConditionalNegativeSampler (conditional_negative_sampler.cc:37-161, condition_table.cc:65-148,
 * attribute_nodes_map.h:74-127) under the seeding contract; see the comment above its definition for the layout of the
 * keys, the draw numbering and the one deliberate divergence (the reference's dead fill loop runs here).  out[batch * count]. */
#define GLXO_NO_KEY INT64_MIN
int glxo_cond_negative_sample(const int64_t* ids, const float* weights, int64_t U, int32_t ncols, const int64_t* cand_keys,
                              const float* props, const glxo_graph* g, const int64_t* src, const int64_t* dst,
                              const int64_t* dst_keys, int32_t batch, int32_t count, int batch_share, int unique,
                              int32_t retry, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                              int64_t* out, int32_t* filled_by_columns_out /* [batch] or NULL: ids the condition columns gave */);

/* Restates SubGraphSampler::InduceSubGraph (subgraph/subgraph_sampler.cc:34-95) on FullSampler's response rows and
 * its need_dist labelling (:71-93, subgraph_utils.cc:36-57). */
int64_t glxo_subgraph_induce(const int64_t* nodes, int32_t n, const int64_t* offsets, const int64_t* nbr, const int64_t* eid,
                             int32_t* row_out, int32_t* col_out, int64_t* eid_out, int64_t cap);
void glxo_subgraph_dist(int32_t n, const int32_t* row, const int32_t* col, int64_t m, int32_t* dist_to_src,
                        int32_t* dist_to_dst);

#ifdef __cplusplus
}
#endif
#endif /* GLX_ORACLE_H_ */
