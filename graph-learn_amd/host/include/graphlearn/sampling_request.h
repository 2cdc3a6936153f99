// SamplingRequest / SamplingResponse / Shape with the reference's surface
// (graphlearn/src/include/sampling_request.h:31-149).
#ifndef GLX_HOST_SAMPLING_REQUEST_H_
#define GLX_HOST_SAMPLING_REQUEST_H_
#include <string>
#include <vector>

#include "graphlearn/op_request.h"
#include "graphlearn/partition.h"

namespace graphlearn {

enum FilterType { kOperatorUnspecified = 0, kEqual = 1, kLargerThan = 2 };  // include/constants.h:135-139
enum FilterField { kFieldUnspecified = 0, kId = 1, kTimestamp = 2 };

struct Shape {
  size_t dim1;  // batch size
  size_t dim2;  // neighbor count
  size_t size;  // dim1 * dim2 (dense)
  std::vector<int32_t> segments;
  bool sparse;
  Shape() : dim1(0), dim2(0), size(0), sparse(false) {}
  Shape(size_t x, size_t y) : dim1(x), dim2(y), size(x * y), segments(x, (int32_t)y), sparse(false) {}
  // sparse response (FullSampler): per-row counts (sampling_request.h:46-51)
  Shape(size_t x, size_t y, const std::vector<int32_t>& inds)
      : dim1(x), dim2(y), size(0), segments(inds), sparse(true) {
    for (int32_t v : inds) size += (size_t)v;
  }
};

class SamplingRequest : public OpRequest {
public:
  SamplingRequest();
  SamplingRequest(const std::string& type, const std::string& strategy, int32_t neighbor_count,
                  FilterType filter_type = kOperatorUnspecified,
                  FilterField filter_field = kFieldUnspecified);
  OpRequest* Clone() const override;
  void Init(const Tensor::Map& params) override;
  using OpRequest::Set;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  void Set(const int64_t* src_ids, int32_t batch_size);

  const std::string& Type() const;
  const std::string& Strategy() const;
  int32_t BatchSize() const;
  int32_t NeighborCount() const { return neighbor_count_; }
  const int64_t* GetSrcIds() const;
  // Original row indices of a part request (set by HashPartitioner), or nullptr.
  const int64_t* GetRngRows() const;
  // Pin the random stream of this request.  Unset (the default),
  // the operator uses its own per-call counter; a caller that partitions a request
  // sets one value on it so that every part draws from the same streams.
  void SetCallCounter(int64_t call_counter);
  bool HasCallCounter() const;
  int64_t CallCounter() const;
  // true when a filter was requested: Filter::operator bool, sampler/filter.h:73-75
  bool HasFilter() const { return filter_type_ != kOperatorUnspecified; }
  FilterType GetFilterType() const { return filter_type_; }
  FilterField GetFilterField() const { return filter_field_; }
  // The filter values, one per src id.  Set(tensors) fills them from tensors[kFilterValues]
  // the way Filter::FillValues does (filter.cc:53-67: each value covers
  // batch / values.Size() consecutive src ids); SetFilterValues appends values as they are
  // (a direct caller's shortcut past the tensor map).  nullptr unless there is exactly one
  // value per src id.
  void SetFilterValues(const int64_t* values, int32_t count);
  const int64_t* GetFilterValues() const;

private:
  void InitParams(const std::string& type, const std::string& strategy);
  int32_t neighbor_count_;
  FilterType filter_type_;
  FilterField filter_field_;
};

// ConditionalSamplingRequest (include/sampling_request.h:152-202, service/request/conditional_sampling_request.cc) for
// "ConditionalNegativeSampler": negatives for (src, dst) pairs that share selected attributes with the dst
// (conditional_negative_sampler.cc:37-161).  `type` is an edge type (strategies "random", "in_degree") or a node type
// ("node_weight"); dst_node_type is the node type the condition attributes are looked up in.
class ConditionalSamplingRequest : public SamplingRequest {
public:
  ConditionalSamplingRequest();
  ConditionalSamplingRequest(const std::string& type, const std::string& strategy, int32_t neighbor_count,
                             const std::string& dst_node_type, bool batch_share, bool unique);
  OpRequest* Clone() const override;
  using SamplingRequest::Set;
  void Init(const Tensor::Map& params) override;
  void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) override;
  void SetIds(const int64_t* src_ids, const int64_t* dst_ids, int32_t batch_size);
  void SetSelectedCols(const std::vector<int32_t>& int_cols, const std::vector<float>& int_props,
                       const std::vector<int32_t>& float_cols, const std::vector<float>& float_props,
                       const std::vector<int32_t>& str_cols, const std::vector<float>& str_props);
  // The sampling strategy of the DEFAULT table ("random" / "in_degree" / "node_weight"); the operator's registry
  // name ("ConditionalNegativeSampler") is what OpRequest::Name() returns.
  const std::string& Strategy() const;
  const std::string& DstNodeType() const;
  bool BatchShare() const;
  bool Unique() const;
  const int64_t* GetDstIds() const;
  std::vector<int32_t> IntCols() const;
  std::vector<float> IntProps() const;
  std::vector<int32_t> FloatCols() const;
  std::vector<float> FloatProps() const;
  std::vector<int32_t> StrCols() const;
  std::vector<float> StrProps() const;
};

class SamplingResponse : public OpResponse {
public:
  SamplingResponse();
  OpResponse* New() const override { return new SamplingResponse; }
  void Swap(OpResponse& right) override;

  void SetShape(size_t dim1, size_t dim2);
  void SetShape(size_t dim1, size_t dim2, const std::vector<int32_t>& segments);  // sparse
  void InitNeighborIds();
  void InitEdgeIds();
  void AppendNeighborId(int64_t id);
  void AppendEdgeId(int64_t id);
  void FillWith(int64_t neighbor_id, int64_t edge_id = -1);

  const Shape GetShape() const { return shape_; }
  int64_t* GetNeighborIds();
  int64_t* GetEdgeIds();
  const int64_t* GetNeighborIds() const;
  const int64_t* GetEdgeIds() const;

  // Device-path addition: size both tensors to dim1*dim2 for one bulk write.
  void ResizeDense();
  // Negative samplers answer with neighbour ids only (random_negative_sampler.cc:55-61).
  void ResizeNeighborIds();
  // Scatter the shards' rows back to their request positions (stitcher.h:67-107).
  void Stitch(ShardsPtr<OpResponse> shards);

private:
  Shape shape_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_SAMPLING_REQUEST_H_
