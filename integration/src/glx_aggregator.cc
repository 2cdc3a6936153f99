// INTEGRATION.md 1.2 -- Aggregator::Aggregate with a glx body.
//
// Replaces core/operator/aggregator/aggregator.cc of the reference tree: the five operators (sum_, mean_, max_, min_,
// prod_aggregator.cc) stay as they are -- they only override InitFunc / AggFunc / FinalFunc, which
// AggregatingResponse::Stitch still calls on the host (aggregating_request.cc:172-213) -- and inherit Aggregate from
// here.  Compiled against the reference's headers (integration/Makefile).
#include <algorithm>
#include <vector>

#include "core/operator/aggregator/aggregator.h"
#include "glx_mirror.h"
#include "include/config.h"
#include "common/base/errors.h"
#include "include/constants.h"

namespace graphlearn {
namespace op {
namespace {
int GlxAggregatorId(const std::string& name) {  // the operator is looked up by the request's name (op_factory.cc:45-61)
  if (name == "SumAggregator") return GLX_AGG_SUM;
  if (name == "MeanAggregator") return GLX_AGG_MEAN;
  if (name == "MaxAggregator") return GLX_AGG_MAX;
  if (name == "MinAggregator") return GLX_AGG_MIN;
  if (name == "ProdAggregator") return GLX_AGG_PROD;
  return -1;
}
}  // namespace

// [glx-aggregate]
Status Aggregator::Aggregate(const AggregatingRequest* req, AggregatingResponse* res) {
  const glx_features* feats = nullptr;
  int rc = GlxFeaturesOf(graph_store_, req->Type(), &feats);
  if (rc != GLX_OK) return GlxStatus(rc);
  int32_t dim = 0;
  glx_features_info(feats, nullptr, &dim, nullptr, nullptr);  // SideInfo.f_num
  const int32_t num_segments = req->NumSegments();
  res->SetEmbeddingDim(dim);
  res->SetNumSegments(num_segments);
  res->SetName(req->Name());
  const int op = GlxAggregatorId(req->Name());
  if (op < 0) return error::InvalidArgument("not an aggregator: " + req->Name());

  std::vector<float> emb(static_cast<size_t>(num_segments) * dim);
  std::vector<int32_t> cnt(num_segments);
  const int64_t* ids = req->tensors_.at(kNodeIds).GetInt64();
  const int32_t* segs = req->tensors_.at(kSegmentIds).GetInt32();
  rc = glx_aggregate(feats, op, ids, segs, req->NumIds(), num_segments, GLOBAL_FLAG(DefaultFloatAttribute),
                         emb.data(), cnt.data(), GLX_PTR_HOST, /*stream=*/nullptr);
  if (rc != GLX_OK) return GlxStatus(rc);
  for (int32_t s = 0; s < num_segments; ++s) {
    res->AppendEmbedding(emb.data() + static_cast<size_t>(s) * dim);
    res->AppendSegment(cnt[s]);
  }
  return Status::OK();
}
// [/glx-aggregate]

// The host-side fold of AggregatingResponse::Stitch keeps calling these on the five operators, which override what
// they need; the base starts at zero, folds nothing and replaces empty segments by the default attribute.
void Aggregator::InitFunc(float* value, int32_t dim) {
  std::fill(value, value + dim, 0.0f);
}

void Aggregator::AggFunc(float* /*left*/, const float* /*right*/, int32_t /*size*/, const int32_t* /*segments*/,
                         int32_t /*num_segments*/) {}

void Aggregator::FinalFunc(float* values, int32_t size, const int32_t* segments, int32_t num_segments) {
  const int32_t dim = num_segments > 0 ? size / num_segments : 0;
  for (int32_t s = 0; s < num_segments; ++s) {
    if (segments[s] == 0) std::fill(values + s * dim, values + (s + 1) * dim, GLOBAL_FLAG(DefaultFloatAttribute));
  }
}

}  // namespace op
}  // namespace graphlearn
