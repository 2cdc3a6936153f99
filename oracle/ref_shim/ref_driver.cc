// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// C driver over the verbatim reference hot path (see oracle/Makefile): feeds
// graphs straight into the reference storages the way LocalGraph::UpdateEdges
// does (graphlearn/src/core/graph/local_graph.cc:50-64: SetSideInfo, Add, then
// Build), looks operators up through the reference's own OpFactory
// (graphlearn/src/core/operator/op_factory.cc:45-61) and calls
// Operator::Process (graphlearn/src/core/operator/operator.h:36-37) on the
// reference's own SamplingRequest / AggregatingRequest objects.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

// Test-only access to the reference AliasMethod's private tables (probs_/alias_)
// so that golden vectors can pin glx's device alias build bit-for-bit.
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include "common/threading/sync/lock.h"
#include "core/graph/storage/types.h"
#define private public
#include "core/operator/sampler/alias_method.h"
#undef private
#include "common/base/hash.h"
#include "core/graph/graph_store.h"
#include "core/io/parser.h"
#include "core/io/element_value.h"
#include "core/operator/op_factory.h"
#include "include/aggregating_request.h"
#include "include/config.h"
#include "include/index_option.h"
#include "include/random_walk_request.h"
#include "include/subgraph_request.h"
#include "include/sampling_request.h"

namespace std {
struct glx_fixed_random_device;  // from fixed_rd.h (force-included into sampler TUs)
}
// The seed hook lives in a function-local static of a header-only struct; we
// re-declare the accessor through the same header so both sides agree.
#include "fixed_rd.h"

using namespace graphlearn;  // NOLINT

namespace {
struct Ref {
  GraphStore* store;
};

template <class F>
void RunMaybeFresh(int fresh_thread, F&& f) {
  // thread_local mt19937 engines in the reference are seeded on first use in a
  // thread; a fresh thread therefore restarts the stream from the pinned seed.
  if (fresh_thread) {
    std::thread t(f);
    t.join();
  } else {
    f();
  }
}
}  // namespace

extern "C" {

// storage_mode: reference flag bits (graphlearn/src/core/graph/storage/storage_mode.cc:22-38)
//   2 = default (vector-of-vectors + stats), 3 = CSR ("compressed") + stats.
void* glref_create(int storage_mode, int padding_mode, int64_t default_neighbor_id,
                   float default_float_attr) {
  SetGlobalFlagStorageMode(storage_mode);
  SetGlobalFlagPaddingMode(padding_mode);
  SetGlobalFlagDefaultNeighborId(default_neighbor_id);
  SetGlobalFlagDefaultFloatAttribute(default_float_attr);
  Ref* r = new Ref;
  r->store = new GraphStore(nullptr);
  op::OpFactory::GetInstance()->Set(r->store);
  return r;
}

void glref_destroy(void* h) {
  Ref* r = static_cast<Ref*>(h);
  op::OpFactory::GetInstance()->Set(nullptr);
  delete r->store;
  delete r;
}

void glref_set_flags(int padding_mode, int64_t default_neighbor_id, float default_float_attr) {
  SetGlobalFlagPaddingMode(padding_mode);
  SetGlobalFlagDefaultNeighborId(default_neighbor_id);
  SetGlobalFlagDefaultFloatAttribute(default_float_attr);
}

void glref_set_seed(unsigned int seed) { std::glx_fixed_random_device::seed() = seed; }

// weights may be NULL (unweighted edge type).  Edge ids are insertion indices,
// exactly as MemoryEdgeStorage::Add assigns them.
int glref_add_edges(void* h, const char* edge_type, const int64_t* src, const int64_t* dst,
                    const float* weights, int64_t n) {
  Ref* r = static_cast<Ref*>(h);
  io::GraphStorage* st = r->store->GetGraph(edge_type)->GetLocalStorage();
  io::SideInfo info;
  info.format = weights ? io::kWeighted : io::kDefault;
  info.type = edge_type;
  st->SetSideInfo(&info);
  io::EdgeValue v;
  for (int64_t i = 0; i < n; ++i) {
    v.src_id = src[i];
    v.dst_id = dst[i];
    v.weight = weights ? weights[i] : 0.0f;
    st->Add(&v);
  }
  return 0;
}

// Timestamped edge type: Build() then orders every row by timestamp
// (memory_adj_matrix.cc:60-66,129-148).  weights may be NULL.
int glref_add_edges_ts(void* h, const char* edge_type, const int64_t* src, const int64_t* dst,
                       const float* weights, const int64_t* timestamps, int64_t n) {
  Ref* r = static_cast<Ref*>(h);
  io::GraphStorage* st = r->store->GetGraph(edge_type)->GetLocalStorage();
  io::SideInfo info;
  info.format = io::kTimestamped | (weights ? io::kWeighted : 0);
  info.type = edge_type;
  st->SetSideInfo(&info);
  io::EdgeValue v;
  for (int64_t i = 0; i < n; ++i) {
    v.src_id = src[i];
    v.dst_id = dst[i];
    v.weight = weights ? weights[i] : 0.0f;
    v.timestamp = timestamps[i];
    st->Add(&v);
  }
  return 0;
}

int glref_build_graph(void* h, const char* edge_type) {
  Ref* r = static_cast<Ref*>(h);
  IndexOption opt;
  opt.name = "sort";
  return r->store->GetGraph(edge_type)->Build(opt).ok() ? 0 : 1;
}

int glref_add_nodes(void* h, const char* node_type, const int64_t* ids, const float* feats,
                    int64_t n, int32_t dim) {
  Ref* r = static_cast<Ref*>(h);
  io::NodeStorage* st = r->store->GetNoder(node_type)->GetLocalStorage();
  io::SideInfo info;
  info.format = io::kAttributed;
  info.f_num = dim;
  info.type = node_type;
  st->SetSideInfo(&info);
  io::NodeValue v;
  for (int64_t i = 0; i < n; ++i) {
    v.id = ids[i];
    v.attrs->Clear();
    v.attrs->Add(feats + i * dim, dim);
    st->Add(&v);
  }
  return 0;
}

int glref_build_nodes(void* h, const char* node_type) {
  Ref* r = static_cast<Ref*>(h);
  IndexOption opt;
  opt.name = "sort";
  return r->store->GetNoder(node_type)->Build(opt).ok() ? 0 : 1;
}

// Post-Build adjacency export (the device CSR must be built from exactly this
// order: SURVEY.md 8(a) quirk 10).  Returns the degree; copies min(deg, cap).
int64_t glref_get_row(void* h, const char* edge_type, int64_t src, int64_t* nbr, int64_t* eid,
                      int64_t cap) {
  Ref* r = static_cast<Ref*>(h);
  io::GraphStorage* st = r->store->GetGraph(edge_type)->GetLocalStorage();
  auto nb = st->GetNeighbors(src);
  auto ed = st->GetOutEdges(src);
  if (!nb) return 0;
  int64_t deg = nb.Size();
  for (int64_t i = 0; i < deg && i < cap; ++i) {
    nbr[i] = nb[i];
    eid[i] = ed[i];
  }
  return deg;
}

float glref_edge_weight(void* h, const char* edge_type, int64_t edge_id) {
  Ref* r = static_cast<Ref*>(h);
  return r->store->GetGraph(edge_type)->GetLocalStorage()->GetEdgeWeight(edge_id);
}

// Runs the reference sampler named `strategy` ("RandomSampler", ...).
// Returns 0 on OK, else the reference error code (or -1: unknown op).
int glref_sample(void* h, const char* edge_type, const char* strategy, const int64_t* src,
                 int32_t batch, int32_t k, int64_t* nbr_out, int64_t* eid_out, int fresh_thread) {
  (void)h;
  int rc = 0;
  RunMaybeFresh(fresh_thread, [&]() {
    SamplingRequest req(edge_type, strategy, k);
    SamplingResponse res;
    req.Set(src, batch);
    op::Operator* op = op::OpFactory::GetInstance()->Create(req.Name());
    if (!op) { rc = -1; return; }
    Status s = op->Process(&req, &res);
    if (!s.ok()) { rc = static_cast<int>(s.code()); return; }
    size_t n = static_cast<size_t>(batch) * k;
    if (nbr_out) memcpy(nbr_out, res.GetNeighborIds(), n * sizeof(int64_t));
    if (eid_out) memcpy(eid_out, res.GetEdgeIds(), n * sizeof(int64_t));
  });
  return rc;
}

// `calls` consecutive requests of the same (strategy, src, k) in ONE fresh thread: the reference's thread_local
// engines carry their state from one request to the next, which is what the outputs [calls][batch][k] show.
// strategies = calls names separated by ',' (e.g. "EdgeWeightSampler,InDegreeSampler": the two share
// AliasMethod::Sample's engine, alias_method.cc:114-115).
int glref_sample_sequence(void* h, const char* edge_type, const char* strategies, const int64_t* src, int32_t batch,
                          int32_t k, int32_t calls, int64_t* nbr_out, int64_t* eid_out) {
  (void)h;
  int rc = 0;
  std::vector<std::string> names;
  {
    std::string all(strategies), cur;
    for (char c : all) {
      if (c == ',') { names.push_back(cur); cur.clear(); } else { cur.push_back(c); }
    }
    names.push_back(cur);
  }
  if (static_cast<int32_t>(names.size()) != calls) return -2;
  RunMaybeFresh(1, [&]() {
    for (int32_t c = 0; c < calls; ++c) {
      SamplingRequest req(edge_type, names[c], k);
      SamplingResponse res;
      req.Set(src, batch);
      op::Operator* op = op::OpFactory::GetInstance()->Create(req.Name());
      if (!op) { rc = -1; return; }
      Status s = op->Process(&req, &res);
      if (!s.ok()) { rc = static_cast<int>(s.code()); return; }
      size_t n = static_cast<size_t>(batch) * k;
      memcpy(nbr_out + c * n, res.GetNeighborIds(), n * sizeof(int64_t));
      memcpy(eid_out + c * n, res.GetEdgeIds(), n * sizeof(int64_t));
    }
  });
  return rc;
}

int glref_aggregate(void* h, const char* node_type, const char* strategy, const int64_t* ids,
                    const int32_t* segs, int32_t num_ids, int32_t num_segments, float* emb_out,
                    int32_t* cnt_out, int32_t* dim_out) {
  (void)h;
  AggregatingRequest req(node_type, strategy);
  AggregatingResponse res;
  req.Set(ids, segs, num_ids, num_segments);
  op::Operator* op = op::OpFactory::GetInstance()->Create(req.Name());
  if (!op) return -1;
  Status s = op->Process(&req, &res);
  if (!s.ok()) return static_cast<int>(s.code());
  int32_t dim = res.EmbeddingDim();
  if (dim_out) *dim_out = dim;
  if (emb_out) memcpy(emb_out, res.Embeddings(), sizeof(float) * dim * num_segments);
  if (cnt_out) memcpy(cnt_out, res.Segments(), sizeof(int32_t) * num_segments);
  return 0;
}

// The reference's own cross-shard combine, AggregatingResponse::Stitch
// (aggregating_request.cc:172-213), fed with P hand-built shard responses.
int glref_aggregate_stitch(const char* strategy, int32_t P, const float* parts, const int32_t* cnts,
                           int32_t num_segments, int32_t dim, float* emb_out, int32_t* cnt_out) {
  ShardsPtr<OpResponse> shards(new Shards<OpResponse>(P));
  for (int32_t p = 0; p < P; ++p) {
    AggregatingResponse* r = new AggregatingResponse;
    r->SetName(strategy);
    r->SetEmbeddingDim(dim);
    r->SetNumSegments(num_segments);
    for (int32_t s = 0; s < num_segments; ++s) {
      r->AppendEmbedding(parts + ((int64_t)p * num_segments + s) * dim);
      r->AppendSegment(cnts[(int64_t)p * num_segments + s]);
    }
    shards->Add(p, r, true);
  }
  AggregatingResponse out;
  out.Stitch(shards);
  memcpy(emb_out, out.Embeddings(), sizeof(float) * (size_t)dim * num_segments);
  memcpy(cnt_out, out.Segments(), sizeof(int32_t) * num_segments);
  return 0;
}

// The loader's primitives: Hash64 (common/base/hash.cc:144-150) and ParseAttribute
// (core/io/parser.cc:39-104) on one packed attribute string.  types[i]: io::DataType;
// returns the status code; the three outputs are filled in attribute order per kind.
uint64_t glref_hash64(const char* data, int64_t n) { return ::graphlearn::Hash64(data, (size_t)n); }

int glref_parse_attribute(const char* input, int64_t len, const char* delimiter, const int32_t* types,
                          const int64_t* hash_buckets, int32_t num_types, int32_t with_buckets, int64_t* ints_out,
                          int32_t* num_ints, float* floats_out, int32_t* num_floats, char* strings_out,
                          int64_t strings_cap, int32_t* num_strings) {
  io::AttributeInfo info;
  info.delimiter = delimiter;
  for (int32_t i = 0; i < num_types; ++i) {
    info.AppendType(static_cast<DataType>(types[i]));
    if (with_buckets) info.AppendHashBucket(hash_buckets[i]);
  }
  io::AttributeValue* value = io::NewDataHeldAttributeValue();
  LiteString s(input, (size_t)len);
  Status st = io::ParseAttribute(s, info, value);
  int ni = 0, nf = 0, ns = 0;
  const int64_t* iv = value->GetInts(&ni);
  const float* fv = value->GetFloats(&nf);
  const std::string* sv = value->GetStrings(&ns);
  for (int i = 0; i < ni; ++i) ints_out[i] = iv[i];
  for (int i = 0; i < nf; ++i) floats_out[i] = fv[i];
  int64_t at = 0;  // strings come back '\n'-separated
  for (int i = 0; i < ns; ++i) {
    for (char c : sv[i]) {
      if (at < strings_cap - 1) strings_out[at++] = c;
    }
    if (at < strings_cap - 1) strings_out[at++] = '\n';
  }
  strings_out[at] = 0;
  *num_ints = ni;
  *num_floats = nf;
  *num_strings = ns;
  delete value;
  return static_cast<int>(st.code());
}

// Weighted node table for NodeWeightNegativeSampler (node_weight_negative_sampler.cc:29-110).
int glref_add_weighted_nodes(void* h, const char* node_type, const int64_t* ids, const float* weights, int64_t n) {
  Ref* r = static_cast<Ref*>(h);
  io::NodeStorage* st = r->store->GetNoder(node_type)->GetLocalStorage();
  io::SideInfo info;
  info.format = io::kWeighted;
  info.type = node_type;
  st->SetSideInfo(&info);
  io::NodeValue v;
  for (int64_t i = 0; i < n; ++i) {
    v.id = ids[i];
    v.weight = weights[i];
    st->Add(&v);
  }
  return 0;
}

// GetAllDstIds / GetAllInDegrees (memory_topo_storage.cc:119-141, topo_statics.cc:32-55): the
// candidate list of the negative samplers.  Returns the count; copies min(count, cap).
int64_t glref_dst_statics(void* h, const char* edge_type, int64_t* ids_out, int32_t* in_degrees_out, int64_t cap) {
  Ref* r = static_cast<Ref*>(h);
  io::GraphStorage* st = r->store->GetGraph(edge_type)->GetLocalStorage();
  auto ids = st->GetAllDstIds();
  auto deg = st->GetAllInDegrees();
  const int64_t n = ids.Size();
  for (int64_t i = 0; i < n && i < cap; ++i) {
    ids_out[i] = ids[i];
    in_degrees_out[i] = deg[i];
  }
  return n;
}

// FullSampler (full_sampler.cc:28-97) answers with a sparse response: per-row
// neighbour counts (Shape::segments) + concatenated values.  degrees_out[batch];
// nbr_out / eid_out need capacity `cap`; returns the total or -(error code) - 1.
int64_t glref_sample_full(void* h, const char* edge_type, const int64_t* src, int32_t batch,
                          int32_t max_limit, int32_t* degrees_out, int64_t* nbr_out, int64_t* eid_out,
                          int64_t cap) {
  (void)h;
  SamplingRequest req(edge_type, "FullSampler", max_limit);
  SamplingResponse res;
  req.Set(src, batch);
  op::Operator* op = op::OpFactory::GetInstance()->Create(req.Name());
  if (!op) return -2;
  Status s = op->Process(&req, &res);
  if (!s.ok()) return -static_cast<int64_t>(s.code()) - 1;
  const Shape shape = res.GetShape();
  int64_t total = 0;
  for (int32_t i = 0; i < batch; ++i) {
    degrees_out[i] = shape.segments[i];
    total += shape.segments[i];
  }
  for (int64_t i = 0; i < total && i < cap; ++i) {
    nbr_out[i] = res.GetNeighborIds()[i];
    eid_out[i] = res.GetEdgeIds()[i];
  }
  return total;
}

// The samplers with a Filter (core/operator/sampler/filter.{h,cc}): the filter values reach
// the request the way the DAG runner hands them over -- SamplingRequest::Set(tensors), which
// runs Filter::FillValues (sampling_request.cc:127-136).  strategy "FullSampler" answers
// with segments (degrees_out != NULL, capacity `cap` values); the others with [batch, k].
// Returns the number of values, or -(error code) - 1.
int64_t glref_sample_filtered(void* h, const char* edge_type, const char* strategy, const int64_t* src,
                              int32_t batch, int32_t k, int filter_type, int filter_field,
                              const int64_t* values, int32_t num_values, int32_t retry_times,
                              int32_t* degrees_out, int64_t* nbr_out, int64_t* eid_out, int64_t cap,
                              int fresh_thread) {
  (void)h;
  int64_t rc = 0;
  SetGlobalFlagSamplingRetryTimes(retry_times);
  RunMaybeFresh(fresh_thread, [&]() {
    SamplingRequest req(edge_type, strategy, k, static_cast<FilterType>(filter_type),
                        static_cast<FilterField>(filter_field));
    SamplingResponse res;
    Tensor::Map tensors;
    ADD_TENSOR(tensors, kSrcIds, kInt64, batch);
    tensors[kSrcIds].AddInt64(src, src + batch);
    ADD_TENSOR(tensors, kFilterValues, kInt64, num_values);
    tensors[kFilterValues].AddInt64(values, values + num_values);
    req.Set(tensors);
    op::Operator* op = op::OpFactory::GetInstance()->Create(req.Name());
    if (!op) { rc = -2; return; }
    Status s = op->Process(&req, &res);
    if (!s.ok()) { rc = -static_cast<int64_t>(s.code()) - 1; return; }
    int64_t total = static_cast<int64_t>(batch) * k;
    if (degrees_out) {
      const Shape shape = res.GetShape();
      total = 0;
      for (int32_t i = 0; i < batch; ++i) {
        degrees_out[i] = shape.segments[i];
        total += shape.segments[i];
      }
    }
    for (int64_t i = 0; i < total && i < cap; ++i) {
      nbr_out[i] = res.GetNeighborIds()[i];
      eid_out[i] = res.GetEdgeIds()[i];
    }
    rc = total;
  });
  return rc;
}

// The reference's RandomWalk operator (core/operator/random_walk/random_walk.cc) on a request
// built the way the DAG runner builds it -- RandomWalkRequest::Set(tensors), which makes every
// src id its own parent without neighbours (random_walk_request.cc:120-131).  core/runner's
// OpRunner is stubbed to Operator::Process (ref_shim/stubs/core/runner/op_runner.h).
// walks_out[batch * walk_len]; returns 0 or the reference's error code (-1: unknown op).
int glref_random_walk(void* h, const char* edge_type, const int64_t* src, int32_t batch, int32_t walk_len,
                      float p, float q, int32_t full_nbr_num, int64_t* walks_out, int fresh_thread) {
  (void)h;
  int rc = 0;
  SetGlobalFlagDefaultFullNbrNum(full_nbr_num);
  RunMaybeFresh(fresh_thread, [&]() {
    RandomWalkRequest req(edge_type, p, q, walk_len);
    RandomWalkResponse res;
    Tensor::Map tensors;
    ADD_TENSOR(tensors, kSrcIds, kInt64, batch);
    tensors[kSrcIds].AddInt64(src, src + batch);
    req.Set(tensors);
    op::Operator* op = op::OpFactory::GetInstance()->Create("RandomWalk");
    if (!op) { rc = -1; return; }
    Status s = op->Process(&req, &res);
    if (!s.ok()) { rc = static_cast<int>(s.code()); return; }
    memcpy(walks_out, res.GetWalks(), sizeof(int64_t) * static_cast<size_t>(batch) * walk_len);
  });
  return rc;
}

// Nodes with a weight and int / float / string attributes (the node type ConditionalNegativeSampler looks its conditions
// up in).  int_attrs[n * I], float_attrs[n * F]; the strings of all nodes back to back in `strs` with lengths str_len[n * S].
int glref_add_attr_nodes(void* h, const char* node_type, const int64_t* ids, const float* weights, const int64_t* int_attrs,
                         int32_t I, const float* float_attrs, int32_t F, const char* strs, const int32_t* str_len, int32_t S,
                         int64_t n) {
  Ref* r = static_cast<Ref*>(h);
  io::NodeStorage* st = r->store->GetNoder(node_type)->GetLocalStorage();
  io::SideInfo info;
  info.format = io::kAttributed | (weights ? io::kWeighted : 0);
  info.i_num = I;
  info.f_num = F;
  info.s_num = S;
  info.type = node_type;
  st->SetSideInfo(&info);
  io::NodeValue v;
  const char* cur = strs;
  for (int64_t i = 0; i < n; ++i) {
    v.id = ids[i];
    v.weight = weights ? weights[i] : 0.0f;
    v.attrs->Clear();
    if (I > 0) v.attrs->Add(int_attrs + i * I, I);
    if (F > 0) v.attrs->Add(float_attrs + i * F, F);
    for (int32_t k = 0; k < S; ++k) {
      v.attrs->Add(cur, str_len[i * S + k]);
      cur += str_len[i * S + k];
    }
    st->Add(&v);
  }
  return 0;
}

// The reference's ConditionalNegativeSampler (conditional_negative_sampler.cc) on a ConditionalSamplingRequest built with
// its own constructor + SetIds + SetSelectedCols.  The condition tables and default alias tables are cached per `type`
// by the reference's factories: use a fresh type name per configuration.  out[cap]; *n_out = ids in the response (the
// reference's fill loop never runs, so this may be less than batch * count).  Returns 0 or the error code.
int glref_cond_neg_sample(void* h, const char* type, const char* strategy, const char* dst_node_type, const int64_t* src,
                          const int64_t* dst, int32_t batch, int32_t count, int batch_share, int unique,
                          const int32_t* int_cols, const float* int_props, int32_t n_int, const int32_t* float_cols,
                          const float* float_props, int32_t n_float, const int32_t* str_cols, const float* str_props,
                          int32_t n_str, int32_t retry_times, int64_t* out, int64_t cap, int64_t* n_out, int fresh_thread) {
  (void)h;
  int rc = 0;
  SetGlobalFlagSamplingRetryTimes(retry_times);
  RunMaybeFresh(fresh_thread, [&]() {
    ConditionalSamplingRequest req(type, strategy, count, dst_node_type, batch_share != 0, unique != 0);
    req.SetIds(src, dst, batch);
    req.SetSelectedCols(std::vector<int32_t>(int_cols, int_cols + n_int), std::vector<float>(int_props, int_props + n_int),
                        std::vector<int32_t>(float_cols, float_cols + n_float),
                        std::vector<float>(float_props, float_props + n_float),
                        std::vector<int32_t>(str_cols, str_cols + n_str), std::vector<float>(str_props, str_props + n_str));
    SamplingResponse res;
    op::Operator* op = op::OpFactory::GetInstance()->Create("ConditionalNegativeSampler");
    if (!op) { rc = -1; return; }
    Status s = op->Process(&req, &res);
    if (!s.ok()) { rc = static_cast<int>(s.code()); return; }
    const int64_t n = res.tensors_[kNodeIds].Size();
    *n_out = n;
    for (int64_t i = 0; i < n && i < cap; ++i) out[i] = res.GetNeighborIds()[i];
  });
  return rc;
}

// The reference's SubGraphSampler (core/operator/subgraph/subgraph_sampler.{h,cc}): seeds -> per hop FullSampler with
// limit num_nbrs[h] -> nodes = seeds + sorted set of all sampled neighbours -> InduceSubGraph with limit
// DefaultFullNbrNum.  Outputs: nodes_out[cap_nodes], row/col/eid_out[cap_edges], dist_*_out[cap_nodes] (need_dist).
// sizes_out = {node count, edge count}; returns 0, -1 (unknown op) or the reference's error code.
int glref_subgraph(void* h, const char* nbr_type, const int64_t* src, int32_t batch, const int32_t* num_nbrs, int32_t hops,
                   int need_dist, int32_t full_nbr_num, int64_t* nodes_out, int64_t cap_nodes, int32_t* row_out,
                   int32_t* col_out, int64_t* eid_out, int64_t cap_edges, int32_t* dist_src_out, int32_t* dist_dst_out,
                   int64_t* sizes_out) {
  (void)h;
  SetGlobalFlagDefaultFullNbrNum(full_nbr_num);
  SubGraphRequest req(nbr_type, std::vector<int32_t>(num_nbrs, num_nbrs + hops), need_dist != 0);
  req.Set(src, batch);
  SubGraphResponse res;
  op::Operator* op = op::OpFactory::GetInstance()->Create("SubGraphSampler");
  if (!op) return -1;
  Status s = op->Process(&req, &res);
  if (!s.ok()) return static_cast<int>(s.code());
  sizes_out[0] = res.NodeCount();
  sizes_out[1] = res.EdgeCount();
  for (int64_t i = 0; i < res.NodeCount() && i < cap_nodes; ++i) {
    nodes_out[i] = res.NodeIds()[i];
    if (need_dist) {
      dist_src_out[i] = res.DistToSrc()[i];
      dist_dst_out[i] = res.DistToDst()[i];
    }
  }
  for (int64_t i = 0; i < res.EdgeCount() && i < cap_edges; ++i) {
    row_out[i] = res.RowIndices()[i];
    col_out[i] = res.ColIndices()[i];
    eid_out[i] = res.EdgeIds()[i];
  }
  return 0;
}

// GraphStorage::GetInDegree (memory_topo_storage.cc:103-109, topo_statics.cc:62-69).
int32_t glref_in_degree(void* h, const char* edge_type, int64_t dst_id) {
  Ref* r = static_cast<Ref*>(h);
  return r->store->GetGraph(edge_type)->GetLocalStorage()->GetInDegree(dst_id);
}

// The reference's own AliasMethod::Build (alias_method.cc:57-107) on one
// weight vector; copies its private tables out.
void glref_alias_build(const float* w, int32_t n, float* probs_out, int32_t* alias_out) {
  std::vector<float> dist(w, w + n);
  op::AliasMethod am(&dist);
  for (int32_t i = 0; i < n; ++i) {
    probs_out[i] = am.probs_[i];
    alias_out[i] = am.alias_[i];
  }
}

// ---- CPU-baseline timing legs (reference's concurrency model: one request per
// pool thread, no intra-request parallelism:
// graphlearn/src/service/local/in_memory_service.cc:64-71).
// Each of `threads` workers issues `reps` independent 2-hop requests over its
// own slice of `seeds` (per-request cost includes request/response
// construction, as in the reference).  Returns wall seconds; *edges_out gets
// the number of response slots produced.
double glref_time_sample_2hop(void* h, const char* edge_type, const char* strategy,
                              const int64_t* seeds, int32_t seeds_per_req, int32_t k1, int32_t k2,
                              int32_t reps, int32_t threads, int64_t* edges_out) {
  (void)h;
  std::vector<std::thread> pool;
  std::vector<int64_t> produced(threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      op::Operator* op = op::OpFactory::GetInstance()->Create(strategy);
      for (int r = 0; r < reps; ++r) {
        const int64_t* s = seeds + (static_cast<int64_t>(t) * reps + r) * seeds_per_req;
        SamplingRequest q1(edge_type, strategy, k1);
        SamplingResponse r1;
        q1.Set(s, seeds_per_req);
        op->Process(&q1, &r1);
        int32_t n1 = seeds_per_req * k1;
        produced[t] += n1;
        if (k2 > 0) {
          SamplingRequest q2(edge_type, strategy, k2);
          SamplingResponse r2;
          q2.Set(r1.GetNeighborIds(), n1);
          op->Process(&q2, &r2);
          produced[t] += static_cast<int64_t>(n1) * k2;
        }
      }
    });
  }
  for (auto& th : pool) th.join();
  auto t1 = std::chrono::steady_clock::now();
  int64_t tot = 0;
  for (auto p : produced) tot += p;
  if (edges_out) *edges_out = tot;
  return std::chrono::duration<double>(t1 - t0).count();
}

// Each worker aggregates `reps` requests of `num_ids` ids in fixed-size
// segments of `seg_len` (ids drawn by the caller).  Returns wall seconds.
double glref_time_aggregate(void* h, const char* node_type, const char* strategy,
                            const int64_t* ids, int32_t num_ids, int32_t seg_len, int32_t reps,
                            int32_t threads, int64_t* vertices_out) {
  (void)h;
  int32_t num_segments = num_ids / seg_len;
  std::vector<int32_t> segs(num_ids);
  for (int32_t i = 0; i < num_ids; ++i) segs[i] = i / seg_len;
  std::vector<std::thread> pool;
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      op::Operator* op = op::OpFactory::GetInstance()->Create(strategy);
      for (int r = 0; r < reps; ++r) {
        const int64_t* p = ids + (static_cast<int64_t>(t) * reps + r) * num_ids;
        AggregatingRequest req(node_type, strategy);
        AggregatingResponse res;
        req.Set(p, segs.data(), num_ids, num_segments);
        op->Process(&req, &res);
      }
    });
  }
  for (auto& th : pool) th.join();
  auto t1 = std::chrono::steady_clock::now();
  if (vertices_out) *vertices_out = static_cast<int64_t>(threads) * reps * num_ids;
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
