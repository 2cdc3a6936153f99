// LookupNodes / LookupEdges requests and their response -- the property lookup
// that follows sampling in every pipeline (NeighborSampler.get -> graph.get_nodes /
// get_edges, python/sampler/neighbor_sampler.py:105-121).  Mirrors
// graphlearn/src/include/graph_request.h:160-322 (LookupEdgesRequest,
// LookupNodesRequest, LookupResponse) and the operators
// core/operator/graph/{node,edge}_lookuper.cc over LocalNoder::LookupNodes /
// LocalGraph::LookupEdges (core/graph/local_noder.cc:85-97, local_graph.cc:72-85).
// Float attributes of nodes are gathered on the device (glx_lookup); weights,
// labels, timestamps, int and string attributes are host-resident columns.
// SURVEY 8(f) rank 1.
#ifndef GLX_HOST_GRAPH_REQUEST_H_
#define GLX_HOST_GRAPH_REQUEST_H_
#include <string>
#include <vector>

#include "graphlearn/config.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/op_request.h"

namespace graphlearn {

class LookupNodesRequest : public OpRequest {
public:
  LookupNodesRequest();
  explicit LookupNodesRequest(const std::string& node_type);
  OpRequest* Clone() const override;
  using OpRequest::Set;
  void Init(const Tensor::Map& params) override;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  void Set(const int64_t* node_ids, int32_t batch_size);
  const std::string& NodeType() const;
  int32_t Size() const;
  bool Next(int64_t* node_id) const;  // cursor interface of the reference
  const int64_t* NodeIds() const;

private:
  mutable int32_t cursor_;
};

class LookupEdgesRequest : public OpRequest {
public:
  LookupEdgesRequest();
  explicit LookupEdgesRequest(const std::string& edge_type);
  OpRequest* Clone() const override;
  using OpRequest::Set;
  void Init(const Tensor::Map& params) override;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  // src_ids route the request in distributed mode (graph_request.h:171-173); the
  // lookup itself is by edge id.
  void Set(const int64_t* edge_ids, const int64_t* src_ids, int32_t batch_size);
  const std::string& EdgeType() const;
  int32_t Size() const;
  bool Next(int64_t* edge_id, int64_t* src_id) const;
  const int64_t* EdgeIds() const;
  const int64_t* SrcIds() const;

private:
  mutable int32_t cursor_;
};

// One response type for both lookups (graph_request.h:200-260): per element a
// weight / label / timestamp (if the type has them) and i_num + f_num + s_num
// attribute values, each kind in its own row-major array.
class LookupResponse : public OpResponse {
public:
  LookupResponse();
  OpResponse* New() const override { return new LookupResponse; }
  void SetSideInfo(const io::SideInfo* info, int32_t batch_size);
  int32_t Size() const { return batch_size_; }
  int32_t Format() const { return info_.format; }
  int32_t IntAttrNum() const { return info_.i_num; }
  int32_t FloatAttrNum() const { return info_.f_num; }
  int32_t StringAttrNum() const { return info_.s_num; }
  void AppendWeight(float weight);
  void AppendLabel(int32_t label);
  void AppendTimestamp(int64_t timestamp);
  // nullptr = AttributeValue::Default(side_info) (element_value.cc:26-50)
  void AppendAttribute(const int64_t* ints, const float* floats, const std::string* strings);
  const float* Weights() const;
  const int32_t* Labels() const;
  const int64_t* Timestamps() const;
  const int64_t* IntAttrs() const;
  const float* FloatAttrs() const;
  const std::vector<std::string>& StringAttrs() const { return strings_; }
  // device path: the float attributes arrive as one block
  void SetShape(int32_t batch_size, int32_t float_attr_num);
  float* MutableFloatAttrs();
  // bulk path of the host-resident columns: each sized for the whole batch at once (after SetSideInfo), then written
  // through the pointer -- one pass per column instead of one tensor-map lookup per element and column
  float* ResizeWeights();
  int32_t* ResizeLabels();
  int64_t* ResizeTimestamps();
  int64_t* ResizeIntAttrs();
  std::vector<std::string>* MutableStringAttrs() { return &strings_; }

private:
  io::SideInfo info_;
  std::vector<std::string> strings_;
};

typedef LookupResponse LookupNodesResponse;
typedef LookupResponse LookupEdgesResponse;

// GetDegreeRequest / GetDegreeResponse (graph_request.h:324-370): out-degrees of a
// batch of vertices for one edge type, read from the device CSR (glx_graph_degrees).
class GetDegreeRequest : public OpRequest {
public:
  GetDegreeRequest();
  // node_from: kEdgeSrc = out-degrees of source ids, kEdgeDst = in-degrees of destination ids
  explicit GetDegreeRequest(const std::string& edge_type, NodeFrom node_from = kEdgeSrc);
  NodeFrom GetNodeFrom() const;
  OpRequest* Clone() const override;
  using OpRequest::Set;
  void Init(const Tensor::Map& params) override;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  void Set(const int64_t* node_ids, int32_t batch_size);
  const std::string& EdgeType() const;
  int32_t Size() const;
  const int64_t* NodeIds() const;
};

class GetDegreeResponse : public OpResponse {
public:
  GetDegreeResponse();
  OpResponse* New() const override { return new GetDegreeResponse; }
  void InitDegrees(int32_t batch_size);
  const int32_t* GetDegrees() const;
  int32_t* MutableDegrees();
};

// GetCountRequest / GetCountResponse (graph_request.h:324-349; operator core/operator/graph/
// local_count_getter.cc:25-49): the number of edges / nodes THIS server holds, one int32 per declared
// edge type then per declared node type, each group in type-name order (GraphStore::GetLocalCount).
class GetCountRequest : public OpRequest {
public:
  GetCountRequest();
};

class GetCountResponse : public OpResponse {
public:
  GetCountResponse();
  OpResponse* New() const override { return new GetCountResponse; }
  void Init(int32_t type_num);
  void Append(int32_t count);
  const int32_t* Count() const;
  int32_t Size() const;
};

// GetStatsRequest / GetStatsResponse (graph_request.h:401-417; operator core/operator/graph/
// stats_getter.cc:25-48): per type, the counts of every server -- one int32 tensor per type name
// (GetStatsResponse::SetCounts, graph_lookup_request.cc:742-749).
class GetStatsRequest : public OpRequest {
public:
  GetStatsRequest();
};

class GetStatsResponse : public OpResponse {
public:
  GetStatsResponse();
  OpResponse* New() const override { return new GetStatsResponse; }
  void SetCounts(const Counts& counts);
  Counts GetCounts() const;  // the tensors read back as a map (what the Python client does, python/client.py get_stats)
};

// RandomWalk (include/random_walk_request.h, service/request/random_walk_request.cc;
// operator core/operator/random_walk/random_walk.cc): walk_len steps from every src id over
// one edge type.  p = q = 1 is DeepWalk, anything else node2vec.  The reference's operator
// feeds itself one sub-request per step with the parents and their neighbour lists
// (random_walk.cc:95-130); here all steps run in one device call (glx_random_walk).
class RandomWalkRequest : public OpRequest {
public:
  RandomWalkRequest();
  RandomWalkRequest(const std::string& type, float p, float q, int32_t walk_len = 1);
  OpRequest* Clone() const override;
  using OpRequest::Set;
  void Init(const Tensor::Map& params) override;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  void Set(const int64_t* src_ids, int32_t batch_size);
  const std::string& Type() const;
  float P() const;
  float Q() const;
  int32_t WalkLen() const;
  bool IsDeepWalk() const;  // random_walk_request.cc:152-160
  int32_t BatchSize() const;
  const int64_t* GetSrcIds() const;
  void SetCallCounter(int64_t call_counter);  // pins the random stream, as on SamplingRequest
  bool HasCallCounter() const;
  int64_t CallCounter() const;
};

class RandomWalkResponse : public OpResponse {
public:
  RandomWalkResponse();
  OpResponse* New() const override { return new RandomWalkResponse; }
  void InitWalks(int32_t batch_size, int32_t walk_len);
  const int64_t* GetWalks() const;  // [batch, walk_len] row-major
  int64_t* MutableWalks();
  int32_t WalkLen() const { return walk_len_; }

private:
  int32_t walk_len_ = 0;
};

// GetNodes / GetEdges (graph_request.h:60-260; operators core/operator/graph/node_getter.cc,
// edge_getter.cc over node_generator.h / edge_generator.h): batch traversal of a type's
// ids.  strategy "by_order" | "shuffle" | "random".  The cursor lives with the operator,
// one per (type, node_from): a batch may be short at the end of an epoch, the call after it
// returns OUT_OF_RANGE and starts the next epoch; a request whose `epoch` is behind the
// generator's is OUT_OF_RANGE too (its caller has not noticed the epoch change yet).
class GetNodesRequest : public OpRequest {
public:
  GetNodesRequest();
  GetNodesRequest(const std::string& type, const std::string& strategy, NodeFrom node_from, int32_t batch_size,
                  int32_t epoch = 0);
  OpRequest* Clone() const override;
  void Init(const Tensor::Map& params) override;
  const std::string& Type() const;
  const std::string& Strategy() const;
  NodeFrom GetNodeFrom() const;
  int32_t BatchSize() const;
  int32_t Epoch() const;
};

class GetNodesResponse : public OpResponse {
public:
  GetNodesResponse();
  OpResponse* New() const override { return new GetNodesResponse; }
  void Init(int32_t batch_size);
  void Append(int64_t node_id);
  int32_t Size() const { return batch_size_; }
  const int64_t* NodeIds() const;
};

class GetEdgesRequest : public OpRequest {
public:
  GetEdgesRequest();
  GetEdgesRequest(const std::string& edge_type, const std::string& strategy, int32_t batch_size, int32_t epoch = 0);
  OpRequest* Clone() const override;
  void Init(const Tensor::Map& params) override;
  const std::string& EdgeType() const;
  const std::string& Strategy() const;
  int32_t BatchSize() const;
  int32_t Epoch() const;
};

class GetEdgesResponse : public OpResponse {
public:
  GetEdgesResponse();
  OpResponse* New() const override { return new GetEdgesResponse; }
  void Init(int32_t batch_size);
  void Append(int64_t src_id, int64_t dst_id, int64_t edge_id);
  int32_t Size() const { return batch_size_; }
  const int64_t* SrcIds() const;
  const int64_t* DstIds() const;
  const int64_t* EdgeIds() const;
};

}  // namespace graphlearn
#endif  // GLX_HOST_GRAPH_REQUEST_H_
