// "LookupNodes" operator: float attributes of a batch of node ids, gathered on
// the device (glx_lookup).  Unknown ids yield rows of DefaultFloatAttribute like
// AttributeValue::Default (core/io/element_value.cc:26-50).
#include "glx.h"
#include "graphlearn/config.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"

namespace graphlearn {

LookupNodesRequest::LookupNodesRequest() : OpRequest(kNodeIds), cursor_(0) {}

LookupNodesRequest::LookupNodesRequest(const std::string& node_type) : OpRequest(kNodeIds), cursor_(0) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("LookupNodes");
  ADD_TENSOR(params_, kNodeType, kString, 1);
  params_[kNodeType].AddString(node_type);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, 64);
}

OpRequest* LookupNodesRequest::Clone() const { return new LookupNodesRequest(NodeType()); }

void LookupNodesRequest::Set(const int64_t* node_ids, int32_t batch_size) {
  tensors_[kNodeIds].AddInt64(node_ids, node_ids + batch_size);
}

const std::string& LookupNodesRequest::NodeType() const { return params_.at(kNodeType).GetString(0); }
int32_t LookupNodesRequest::Size() const { return tensors_.at(kNodeIds).Size(); }
const int64_t* LookupNodesRequest::NodeIds() const { return tensors_.at(kNodeIds).GetInt64(); }

bool LookupNodesRequest::Next(int64_t* node_id) const {
  if (cursor_ >= Size()) return false;
  *node_id = tensors_.at(kNodeIds).GetInt64(cursor_++);
  return true;
}

LookupNodesResponse::LookupNodesResponse() : OpResponse(), f_num_(0) {}

void LookupNodesResponse::SetShape(int32_t batch_size, int32_t float_attr_num) {
  batch_size_ = batch_size;
  f_num_ = float_attr_num;
  ADD_TENSOR(tensors_, kFloatAttrKey, kFloat, batch_size * float_attr_num);
  tensors_[kFloatAttrKey].Resize(batch_size * float_attr_num);
}

const float* LookupNodesResponse::FloatAttrs() const { return tensors_.at(kFloatAttrKey).GetFloat(); }
float* LookupNodesResponse::MutableFloatAttrs() { return tensors_[kFloatAttrKey].MutableFloat(); }

REGISTER_REQUEST(LookupNodes, LookupNodesRequest, LookupNodesResponse)

namespace op {

class NodeLookuper : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const LookupNodesRequest* request = static_cast<const LookupNodesRequest*>(req);
    LookupNodesResponse* response = static_cast<LookupNodesResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    Noder* noder = graph_store_->GetNoder(request->NodeType());
    const int32_t dim = noder->GetSideInfo()->f_num;
    response->SetShape(request->Size(), dim);
    const glx_features* f = noder->Device();
    if (!f) return error::InvalidArgument("node type '" + request->NodeType() + "' has no float attributes on the device");
    int rc = glx_lookup(f, request->NodeIds(), request->Size(), GLOBAL_FLAG(DefaultFloatAttribute),
                        response->MutableFloatAttrs(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }
};

REGISTER_OPERATOR("LookupNodes", NodeLookuper)

}  // namespace op
}  // namespace graphlearn
