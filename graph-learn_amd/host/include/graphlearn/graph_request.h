// LookupNodesRequest / LookupNodesResponse -- the float-attribute part of the
// reference's node lookup (graphlearn/src/include/graph_request.h:240-322,
// core/graph/local_noder.cc:85-97, core/operator/graph/node_lookuper.cc:24-52):
// the step that follows sampling in every pipeline (NeighborSampler.get ->
// graph.get_nodes, python/sampler/neighbor_sampler.py:105-109).  SURVEY 8(f) rank 1.
// Weights / labels / int / string attributes are not mirrored on the device.
#ifndef GLX_HOST_GRAPH_REQUEST_H_
#define GLX_HOST_GRAPH_REQUEST_H_
#include <string>

#include "graphlearn/op_request.h"

namespace graphlearn {

class LookupNodesRequest : public OpRequest {
public:
  LookupNodesRequest();
  explicit LookupNodesRequest(const std::string& node_type);
  OpRequest* Clone() const override;
  void Set(const int64_t* node_ids, int32_t batch_size);
  const std::string& NodeType() const;
  int32_t Size() const;
  bool Next(int64_t* node_id) const;  // cursor interface of the reference
  const int64_t* NodeIds() const;

private:
  mutable int32_t cursor_;
};

class LookupNodesResponse : public OpResponse {
public:
  LookupNodesResponse();
  OpResponse* New() const override { return new LookupNodesResponse; }
  void SetShape(int32_t batch_size, int32_t float_attr_num);
  int32_t Size() const { return batch_size_; }
  int32_t FloatAttrNum() const { return f_num_; }
  const float* FloatAttrs() const;  // [Size() * FloatAttrNum()], row-major
  float* MutableFloatAttrs();

private:
  int32_t f_num_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_GRAPH_REQUEST_H_
