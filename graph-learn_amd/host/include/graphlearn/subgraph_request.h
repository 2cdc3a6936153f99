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
  SubGraphRequest();
  SubGraphRequest(const std::string& nbr_type, const std::vector<int32_t>& num_nbrs = std::vector<int32_t>(1),
                  bool need_dist = false);
  OpRequest* Clone() const override;
  void Init(const Tensor::Map& params) override;
  void Set(const Tensor::Map& tensors) override;
  void Set(const int64_t* src_id, int32_t batch_size);
  // link sub-graphs: the batch is the src ids followed by the dst ids (subgraph_request.cc:82-87)
  void Set(const int64_t* src_id, const int64_t* dst_id, int32_t batch_size);

  const std::string& NbrType() const;
  std::vector<int32_t> GetNumNbrs() const;
  bool NeedDist() const;
  const int64_t* GetSrcIds() const;
  int32_t BatchSize() const;
};

class SubGraphResponse : public OpResponse {
public:
  SubGraphResponse();
  OpResponse* New() const override { return new SubGraphResponse; }
  void Init(int32_t batch_size);
  void SetNodeIds(const int64_t* begin, int32_t size);
  void AppendEdge(int32_t row_idx, int32_t col_idx, int64_t e_id);
  void SetDistToSrc(const int32_t* begin, int32_t size);
  void SetDistToDst(const int32_t* begin, int32_t size);
  // Device path: sizes the three COO tensors for `count` entries; the kernel's output is copied straight in.
  void ResizeEdges(int32_t count);
  int32_t* MutableRowIndices();
  int32_t* MutableColIndices();
  int64_t* MutableEdgeIds();

  int32_t NodeCount() const { return batch_size_; }
  int32_t EdgeCount() const;
  const int64_t* NodeIds() const;
  const int32_t* RowIndices() const;
  const int32_t* ColIndices() const;
  const int64_t* EdgeIds() const;
  const int32_t* DistToSrc() const;
  const int32_t* DistToDst() const;
};

}  // namespace graphlearn
#endif  // GLX_HOST_SUBGRAPH_REQUEST_H_
