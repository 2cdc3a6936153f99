// The four neighbour samplers behind the reference's registry names.  Each
// Process() is what a maintainer of the reference would put in
// Sampler::Sample (core/operator/sampler/sampler.h:36-57): size the dense
// response, then one C-ABI call that runs the HIP kernels and writes the
// response tensors in place.  No CPU sampling code exists in this layer: if the
// GPU library fails, the Status says so.
#include <atomic>
#include <vector>

#include "glx.h"
#include "graphlearn/config.h"
#include "graphlearn/dag.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"
#include "graphlearn/sampling_request.h"

namespace graphlearn {
namespace op {

namespace {
// The request's op::Filter as the C-ABI wants it (sampler/filter.h:30-125); the enum
// values are shared with include/glx.h.
glx_filter FilterOf(const SamplingRequest* req) {
  glx_filter f;
  f.type = req->HasFilter() ? (int32_t)req->GetFilterType() : GLX_FILTER_NONE;
  f.field = (int32_t)req->GetFilterField();
  f.values = req->GetFilterValues();
  f.retry_times = GLOBAL_FLAG(SamplingRetryTimes);
  f.default_timestamp = GLOBAL_FLAG(DefaultTimestamp);
  if (f.values == nullptr) f.type = GLX_FILTER_NONE;  // empty batch
  return f;
}
}  // namespace

class Sampler : public Operator, public HopFusable {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    return Sample(static_cast<const SamplingRequest*>(req), static_cast<SamplingResponse*>(res));
  }

  // A chain of hops of a query (dag.h) as one glx_sample_hops call: hop h samples requests[h]'s neighbour count
  // from requests[h]'s edge type for every neighbour hop h - 1 returned; only hop 0's source ids come from the
  // host and only the responses travel back.  Draw for draw what Process would return hop after hop: hop h uses
  // this operator's call counter + h.
  Status ProcessHops(const std::vector<const OpRequest*>& requests, const std::vector<OpResponse*>& responses) override {
    const size_t hops = requests.size();
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    std::vector<const SamplingRequest*> reqs(hops);
    std::vector<SamplingResponse*> ress(hops);
    std::vector<const glx_graph*> graphs(hops);
    std::vector<int32_t> fanouts(hops);
    bool all_loaded = true;
    for (size_t h = 0; h < hops; ++h) {
      reqs[h] = static_cast<const SamplingRequest*>(requests[h]);
      ress[h] = static_cast<SamplingResponse*>(responses[h]);
      Graph* graph = graph_store_->GetGraph(reqs[h]->Type());
      graphs[h] = graph->Device();
      fanouts[h] = reqs[h]->NeighborCount();
      // ... and a request that pins its call counter or its rows' random streams (a part of a partitioned request) is
      // served by Sample(), which honours both (the fused call drew from the operator's own counter)
      if (!graphs[h] || reqs[h]->HasFilter() || reqs[h]->HasCallCounter() || reqs[h]->GetRngRows() != nullptr) all_loaded = false;
      if (graphs[h] && SamplerId() == GLX_SAMPLER_IN_DEGREE) {
        Status s = graph->EnsureInDegree();
        if (!s.ok()) return s;
      }
      if (graphs[h] && SamplerId() == GLX_SAMPLER_EDGE_WEIGHT) {
        Status s = graph->EnsureDefaultWeights();
        if (!s.ok()) return s;
      }
    }
    if (!all_loaded) {  // an edge type nobody loaded default-fills (Sample), a filter / pinned counter needs Sample: hop by hop, each fed by the one before
      for (size_t h = 0; h < hops; ++h) {
        SamplingRequest next(reqs[h]->Type(), reqs[h]->Strategy(), fanouts[h]);
        if (h > 0) next.Set(ress[h - 1]->GetNeighborIds(), (int32_t)ress[h - 1]->GetShape().size);
        Status s = Sample(h == 0 ? reqs[0] : &next, ress[h]);
        if (!s.ok()) return s;
      }
      return Status::OK();
    }
    std::vector<int64_t*> nbr_out(hops), eid_out(hops);
    size_t rows = (size_t)reqs[0]->BatchSize();
    for (size_t h = 0; h < hops; ++h) {
      ress[h]->SetShape(rows, fanouts[h]);
      ress[h]->InitNeighborIds();
      ress[h]->InitEdgeIds();
      ress[h]->ResizeDense();
      nbr_out[h] = ress[h]->GetNeighborIds();
      eid_out[h] = ress[h]->GetEdgeIds();
      rows *= (size_t)fanouts[h];
    }
    const uint64_t cc = call_counter_.fetch_add(hops, std::memory_order_relaxed);
    int rc = glx_sample_hops(graphs.data(), (int32_t)hops, SamplerId(), reqs[0]->GetSrcIds(), reqs[0]->BatchSize(),
                             fanouts.data(), GLOBAL_FLAG(PaddingMode), GLOBAL_FLAG(DefaultNeighborId),
                             (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, nbr_out.data(), eid_out.data(), GLX_PTR_HOST,
                             nullptr);
    return error::FromGlx(rc);
  }

protected:
  virtual int SamplerId() const = 0;

  Status Sample(const SamplingRequest* req, SamplingResponse* res) {
    const int32_t count = req->NeighborCount();
    const int32_t batch_size = req->BatchSize();
    res->SetShape(batch_size, count);
    res->InitNeighborIds();
    res->InitEdgeIds();
    if (req->HasFilter() && batch_size > 0 && !req->GetFilterValues()) {
      return error::InvalidArgument("the request has a filter but not one filter value per src id");
    }
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    Graph* graph = graph_store_->GetGraph(req->Type());
    const glx_graph* g = graph->Device();
    if (g && SamplerId() == GLX_SAMPLER_IN_DEGREE) {
      Status s = graph->EnsureInDegree();
      if (!s.ok()) return s;
    }
    if (g && SamplerId() == GLX_SAMPLER_EDGE_WEIGHT) {
      Status s = graph->EnsureDefaultWeights();
      if (!s.ok()) return s;
    }
    if (g && req->HasFilter() && (int32_t)req->GetFilterType() == GLX_FILTER_EQUAL &&
        (int32_t)req->GetFilterField() == GLX_FILTER_FIELD_ID && SamplerId() != GLX_SAMPLER_RANDOM) {
      Status s = graph->EnsureIdIndex();  // hits by binary search instead of a row scan
      if (!s.ok()) return s;
    }
    res->ResizeDense();
    if (!g) {
      // An edge type nobody loaded behaves like a storage without any row:
      // every src id is unknown -> default fill (random_sampler.cc:58-59).
      int64_t* n = res->GetNeighborIds();
      int64_t* e = res->GetEdgeIds();
      for (int64_t i = 0; i < (int64_t)batch_size * count; ++i) {
        n[i] = GLOBAL_FLAG(DefaultNeighborId);
        e[i] = -1;
      }
      return Status::OK();
    }
    // The reference's RNG state advances from call to call (thread_local
    // mt19937); the contract's equivalent is a per-operator call counter.
    const uint64_t cc = req->HasCallCounter() ? (uint64_t)req->CallCounter()
                                               : call_counter_.fetch_add(1, std::memory_order_relaxed);
    // A part of a partitioned request draws from its rows' ORIGINAL random streams.
    const glx_filter filter = FilterOf(req);
    int rc = glx_sample_filtered(g, SamplerId(), req->GetSrcIds(), req->GetRngRows(), batch_size, count,
                                 GLOBAL_FLAG(PaddingMode), GLOBAL_FLAG(DefaultNeighborId),
                                 (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, &filter, res->GetNeighborIds(),
                                 res->GetEdgeIds(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }

private:
  std::atomic<uint64_t> call_counter_{0};
};

class RandomSampler : public Sampler {
  int SamplerId() const override { return GLX_SAMPLER_RANDOM; }
};
class RandomWithoutReplacementSampler : public Sampler {
  int SamplerId() const override { return GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT; }
};
class EdgeWeightSampler : public Sampler {
  int SamplerId() const override { return GLX_SAMPLER_EDGE_WEIGHT; }
};
class TopkSampler : public Sampler {
  int SamplerId() const override { return GLX_SAMPLER_TOPK; }
};
class InDegreeSampler : public Sampler {  // in_degree_sampler.cc:33-114
  int SamplerId() const override { return GLX_SAMPLER_IN_DEGREE; }
};

// FullSampler (full_sampler.cc:28-97): sparse response = per-row counts + values.
class FullSampler : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const SamplingRequest* request = static_cast<const SamplingRequest*>(req);
    SamplingResponse* response = static_cast<SamplingResponse*>(res);
    const int32_t batch_size = request->BatchSize();
    const int32_t max_limit = request->NeighborCount();
    if (request->HasFilter() && batch_size > 0 && !request->GetFilterValues()) {
      return error::InvalidArgument("the request has a filter but not one filter value per src id");
    }
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    const glx_graph* g = graph_store_->GetGraph(request->Type())->Device();
    std::vector<int32_t> degrees(batch_size, 0);
    std::vector<int64_t> offsets(batch_size + 1, 0);
    if (g) {
      int rc = glx_sample_full_sizes(g, request->GetSrcIds(), batch_size, max_limit, degrees.data(),
                                     offsets.data(), GLX_PTR_HOST, nullptr);
      if (rc != GLX_OK) return error::FromGlx(rc);
    }
    response->SetShape(batch_size, max_limit, degrees);
    response->InitNeighborIds();
    response->InitEdgeIds();
    response->ResizeDense();  // sizes both tensors to shape.size (the sum of the counts)
    if (g && offsets[batch_size] > 0) {
      const glx_filter filter = FilterOf(request);
      int rc = glx_sample_full_filtered(g, request->GetSrcIds(), batch_size, max_limit, offsets.data(),
                                        GLOBAL_FLAG(PaddingMode), GLOBAL_FLAG(DefaultNeighborId), &filter,
                                        response->GetNeighborIds(), response->GetEdgeIds(), GLX_PTR_HOST, nullptr);
      if (rc != GLX_OK) return error::FromGlx(rc);
    }
    return Status::OK();
  }
};

REGISTER_OPERATOR("RandomSampler", RandomSampler)
REGISTER_OPERATOR("RandomWithoutReplacementSampler", RandomWithoutReplacementSampler)
REGISTER_OPERATOR("EdgeWeightSampler", EdgeWeightSampler)
REGISTER_OPERATOR("TopkSampler", TopkSampler)
REGISTER_OPERATOR("InDegreeSampler", InDegreeSampler)
REGISTER_OPERATOR("FullSampler", FullSampler)

// Negative samplers (random_negative_sampler.cc:30-63, in_degree_negative_sampler.cc:29-135,
// node_weight_negative_sampler.cc:29-110): [batch, count] candidate ids, no edge ids.
class NegativeSampler : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const SamplingRequest* request = static_cast<const SamplingRequest*>(req);
    SamplingResponse* response = static_cast<SamplingResponse*>(res);
    const int32_t count = request->NeighborCount();
    const int32_t batch_size = request->BatchSize();
    response->SetShape(batch_size, count);
    response->InitEdgeIds();
    response->InitNeighborIds();
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    const glx_negative* table = nullptr;
    const glx_graph* graph = nullptr;
    Status s = Candidates(request->Type(), &table, &graph);
    if (!s.ok()) return s;
    response->ResizeNeighborIds();
    const uint64_t cc = request->HasCallCounter() ? (uint64_t)request->CallCounter()
                                                   : call_counter_.fetch_add(1, std::memory_order_relaxed);
    int rc = glx_negative_sample(table, Exclude(), graph, request->GetSrcIds(), batch_size, count,
                                 GLOBAL_FLAG(DefaultNeighborId), (uint64_t)GLOBAL_FLAG(SamplingSeed), cc,
                                 response->GetNeighborIds(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }

protected:
  virtual int Exclude() const = 0;
  virtual Status Candidates(const std::string& type, const glx_negative** table, const glx_graph** graph) = 0;

private:
  std::atomic<uint64_t> call_counter_{0};
};

class RandomNegativeSampler : public NegativeSampler {
  int Exclude() const override { return GLX_NEG_EXCLUDE_NONE; }
  Status Candidates(const std::string& type, const glx_negative** table, const glx_graph**) override {
    return graph_store_->GetGraph(type)->Negative(false, false, table);
  }
};
class SoftInDegreeNegativeSampler : public NegativeSampler {
  int Exclude() const override { return GLX_NEG_EXCLUDE_NONE; }
  Status Candidates(const std::string& type, const glx_negative** table, const glx_graph**) override {
    return graph_store_->GetGraph(type)->Negative(true, false, table);
  }
};
class InDegreeNegativeSampler : public NegativeSampler {
  int Exclude() const override { return GLX_NEG_EXCLUDE_NEIGHBORS; }
  Status Candidates(const std::string& type, const glx_negative** table, const glx_graph** graph) override {
    Graph* g = graph_store_->GetGraph(type);
    Status s = g->Negative(true, true, table);
    *graph = g->Device();
    return s;
  }
};
class NodeWeightNegativeSampler : public NegativeSampler {
  int Exclude() const override { return GLX_NEG_EXCLUDE_BATCH; }
  Status Candidates(const std::string& type, const glx_negative** table, const glx_graph**) override {
    return graph_store_->GetNoder(type)->Negative(table);  // Type() is a NODE type here
  }
};

REGISTER_OPERATOR("RandomNegativeSampler", RandomNegativeSampler)
REGISTER_OPERATOR("SoftInDegreeNegativeSampler", SoftInDegreeNegativeSampler)
REGISTER_OPERATOR("InDegreeNegativeSampler", InDegreeNegativeSampler)
REGISTER_OPERATOR("NodeWeightNegativeSampler", NodeWeightNegativeSampler)

}  // namespace op
}  // namespace graphlearn
