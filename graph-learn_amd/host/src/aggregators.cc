// Sum / Mean / Max / Min / Prod aggregators behind the reference's registry
// names: Aggregator::Aggregate (core/operator/aggregator/aggregator.cc:25-59)
// with the response filled by one C-ABI call.
#include "glx.h"
#include "graphlearn/aggregating_request.h"
#include "graphlearn/config.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"

namespace graphlearn {
namespace op {

class Aggregator : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    return Aggregate(static_cast<const AggregatingRequest*>(req), static_cast<AggregatingResponse*>(res));
  }

protected:
  virtual int AggId() const = 0;

  Status Aggregate(const AggregatingRequest* req, AggregatingResponse* res) {
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    Noder* noder = graph_store_->GetNoder(req->Type());
    const int32_t dim = noder->GetSideInfo()->f_num;
    res->SetEmbeddingDim(dim);
    res->SetNumSegments(req->NumSegments());
    res->SetName(req->Name());
    const glx_features* f = noder->Device();
    if (!f) return error::InvalidArgument("node type '" + req->Type() + "' has no float attributes on the device");
    int rc = glx_aggregate(f, AggId(), req->NodeIds(), req->SegmentIds(), req->NumIds(), req->NumSegments(),
                           GLOBAL_FLAG(DefaultFloatAttribute), res->MutableEmbeddings(),
                           res->MutableSegments(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }
};

#define DEFINE_AGGREGATOR(Name, Id)            \
  class Name : public Aggregator {             \
    int AggId() const override { return Id; }  \
  };                                           \
  REGISTER_OPERATOR(#Name, Name)

DEFINE_AGGREGATOR(SumAggregator, GLX_AGG_SUM)
DEFINE_AGGREGATOR(MeanAggregator, GLX_AGG_MEAN)
DEFINE_AGGREGATOR(MaxAggregator, GLX_AGG_MAX)
DEFINE_AGGREGATOR(MinAggregator, GLX_AGG_MIN)
DEFINE_AGGREGATOR(ProdAggregator, GLX_AGG_PROD)

}  // namespace op
}  // namespace graphlearn
