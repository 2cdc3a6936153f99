// SubGraphRequest / SubGraphResponse (graphlearn/src/include/subgraph_request.h:25-92,
// service/request/subgraph_request.cc) for the "SubGraphSampler" operator
// (core/operator/subgraph/subgraph_sampler.{h,cc}): the seeds' sampled neighbourhood and the
// edges it induces, as COO (row / column positions in the node list + edge ids), optionally
// with the SEAL-style distances of every node to the first two nodes (need_dist).
#ifndef GLX_HOST_SUBGRAPH_REQUEST_H_
#define GLX_HOST_SUBGRAPH_REQUEST_H_
#include <string>
#include <vector>

#include "graphlearn/op_request.h"

namespace graphlearn {

class SubGraphRequest : public OpRequest {
public:
  // what a request says -----------------------------------------------------------------
  const std::string& NbrType() const;        // the edge type the sub-graph is induced over
  std::vector<int32_t> GetNumNbrs() const;   // neighbours sampled per hop around the seeds (0: the seeds alone)
  bool NeedDist() const;                     // also return every node's distance to the first two nodes
  int32_t BatchSize() const;
  const int64_t* GetSrcIds() const;

  // how it is made ----------------------------------------------------------------------
  SubGraphRequest();
  SubGraphRequest(const std::string& nbr_type, const std::vector<int32_t>& num_nbrs = std::vector<int32_t>(1),
                  bool need_dist = false);
  void Set(const int64_t* seeds, int32_t count);
  // link sub-graphs: the batch is the source ids followed by the destination ids (subgraph_request.cc:82-87)
  void Set(const int64_t* sources, const int64_t* destinations, int32_t pairs);
  using OpRequest::Set;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  void Init(const Tensor::Map& params) override;
  OpRequest* Clone() const override;
};

class SubGraphResponse : public OpResponse {
public:
  SubGraphResponse();
  OpResponse* New() const override { return new SubGraphResponse; }

  // the sub-graph: nodes, then its edges as COO over positions in the node list --------------
  int32_t NodeCount() const { return batch_size_; }
  const int64_t* NodeIds() const;
  int32_t EdgeCount() const;
  const int32_t* RowIndices() const;
  const int32_t* ColIndices() const;
  const int64_t* EdgeIds() const;
  const int32_t* DistToSrc() const;  // need_dist only
  const int32_t* DistToDst() const;

  // filled by the operator --------------------------------------------------------------
  void Init(int32_t node_count);
  void SetNodeIds(const int64_t* ids, int32_t count);
  void SetDistToSrc(const int32_t* hops, int32_t count);
  void SetDistToDst(const int32_t* hops, int32_t count);
  void AppendEdge(int32_t row, int32_t col, int64_t edge_id);
  // device path: sizes the three COO tensors for `count` entries; the kernel's output is copied straight in
  void ResizeEdges(int32_t count);
  int32_t* MutableRowIndices();
  int32_t* MutableColIndices();
  int64_t* MutableEdgeIds();
};

}  // namespace graphlearn
#endif  // GLX_HOST_SUBGRAPH_REQUEST_H_
