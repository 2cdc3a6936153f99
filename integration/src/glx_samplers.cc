// INTEGRATION.md 1.2 -- the four dense neighbour samplers of graphlearn::op with glx bodies.
//
// Replaces, in the reference tree, core/operator/sampler/{random_sampler.cc:25-78,
// random_without_replacement_sampler.cc:25-77, edge_weight_sampler.cc:27-128, topk_sampler.cc:25-72}: same class
// names, same base class (sampler.h:31-63), same REGISTER_OPERATOR lines, same request / response objects -- only the
// bodies change, to ONE C-ABI call.  Compiled against the reference's headers (integration/Makefile).
#include <atomic>
#include <vector>

#include "core/operator/sampler/sampler.h"
#include "core/operator/sampler/filter.h"
#include "glx_mirror.h"
#include "include/config.h"

namespace graphlearn {
namespace op {
namespace {
// The reference seeds thread_local mt19937 engines from std::random_device and so has no reproducible stream
// (SURVEY.md 8(c)); the device draws are counter-based: (seed, call counter, row, draw).  A maintainer would make the
// seed a GLOBAL_FLAG; the call counter is the one piece of mutable operator state.
std::atomic<uint64_t> g_seed{0};
std::atomic<uint64_t> g_calls{0};
}  // namespace

extern "C" void glx_integration_set_stream(uint64_t seed, uint64_t next_call_counter) {
  g_seed = seed;
  g_calls = next_call_counter;
}

// [glx-dense-sampler]
class GlxDenseSampler : public Sampler {
public:
  virtual ~GlxDenseSampler() {}

  Status Sample(const SamplingRequest* req, SamplingResponse* res) override {
    const int32_t count = req->NeighborCount();
    const int32_t batch_size = req->BatchSize();
    res->SetShape(batch_size, count);
    res->InitNeighborIds();
    res->InitEdgeIds();

    const glx_graph* graph = nullptr;
    int rc = GlxGraphOf(graph_store_, req->Type(), &graph);
    if (rc != GLX_OK) return GlxStatus(rc);
    std::vector<int64_t> nbr(static_cast<size_t>(batch_size) * count), eid(nbr.size());
    const uint64_t call = g_calls.fetch_add(1);
    const Filter* flt = req->GetFilter();
    if (*flt) {  // op::Filter maps onto glx_filter one to one
      glx_filter f = {static_cast<int32_t>(flt->GetType()), static_cast<int32_t>(flt->GetField()),
                      flt->GetValue()->GetInt64(), GLOBAL_FLAG(SamplingRetryTimes), GLOBAL_FLAG(DefaultTimestamp)};
      rc = glx_sample_filtered(graph, SamplerId(), req->GetSrcIds(), /*rng_rows=*/nullptr, batch_size, count,
                               GLOBAL_FLAG(PaddingMode), GLOBAL_FLAG(DefaultNeighborId), g_seed, call, &f, nbr.data(),
                               eid.data(), GLX_PTR_HOST, /*stream=*/nullptr);
    } else {
      rc = glx_sample(graph, SamplerId(), req->GetSrcIds(), batch_size, count, GLOBAL_FLAG(PaddingMode),
                      GLOBAL_FLAG(DefaultNeighborId), g_seed, call, nbr.data(), eid.data(), GLX_PTR_HOST,
                      /*stream=*/nullptr);
    }
    if (rc != GLX_OK) return GlxStatus(rc);
    // the response tensors are append-only (tensor_impl.h:72-75): one bulk append each
    res->tensors_[kNodeIds].AddInt64(nbr.data(), nbr.data() + nbr.size());
    res->tensors_[kEdgeIds].AddInt64(eid.data(), eid.data() + eid.size());
    return Status::OK();
  }

protected:
  virtual int SamplerId() const = 0;
};

class RandomSampler : public GlxDenseSampler {
  int SamplerId() const override { return GLX_SAMPLER_RANDOM; }
};
class RandomWithoutReplacementSampler : public GlxDenseSampler {
  int SamplerId() const override { return GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT; }
};
class EdgeWeightSampler : public GlxDenseSampler {
  int SamplerId() const override { return GLX_SAMPLER_EDGE_WEIGHT; }
};
class TopkSampler : public GlxDenseSampler {
  int SamplerId() const override { return GLX_SAMPLER_TOPK; }
};

REGISTER_OPERATOR("RandomSampler", RandomSampler);
REGISTER_OPERATOR("RandomWithoutReplacementSampler", RandomWithoutReplacementSampler);
REGISTER_OPERATOR("EdgeWeightSampler", EdgeWeightSampler);
REGISTER_OPERATOR("TopkSampler", TopkSampler);
// [/glx-dense-sampler]

}  // namespace op
}  // namespace graphlearn
