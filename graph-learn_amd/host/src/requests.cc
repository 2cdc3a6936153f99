// OpRequest / SamplingRequest / AggregatingRequest and their responses.
// Behaviour follows graphlearn/src/service/request/{op_request.cc,
// sampling_request.cc, aggregating_request.cc}; see the headers for line refs.
#include "graphlearn/aggregating_request.h"
#include "graphlearn/sampling_request.h"

namespace graphlearn {

// ------------------------------------------------------------ base classes --
OpRequest::OpRequest(const std::string& shard_key) : shard_key_(shard_key), shardable_(true) {}

std::string OpRequest::Name() const {
  auto it = params_.find(kOpName);
  return it == params_.end() ? std::string() : it->second.GetString(0);
}

OpRequest* OpRequest::Clone() const { return new OpRequest(shard_key_); }

OpResponse::OpResponse() : batch_size_(0) {}

void OpResponse::Swap(OpResponse& right) {
  std::swap(batch_size_, right.batch_size_);
  params_.swap(right.params_);
  tensors_.swap(right.tensors_);
}

RequestFactory* RequestFactory::GetInstance() {
  static RequestFactory factory;
  return &factory;
}

void RequestFactory::Register(const std::string& name, RequestCreator req, ResponseCreator res) {
  std::lock_guard<std::mutex> g(mtx_);
  req_[name] = req;
  res_[name] = res;
}

OpRequest* RequestFactory::NewRequest(const std::string& name) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = req_.find(name);
  return it == req_.end() ? nullptr : it->second();
}

OpResponse* RequestFactory::NewResponse(const std::string& name) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = res_.find(name);
  return it == res_.end() ? nullptr : it->second();
}

// ---------------------------------------------------------------- sampling --
SamplingRequest::SamplingRequest()
    : OpRequest(kSrcIds), neighbor_count_(0), filter_type_(kOperatorUnspecified),
      filter_field_(kFieldUnspecified) {}

SamplingRequest::SamplingRequest(const std::string& type, const std::string& strategy,
                                 int32_t neighbor_count, FilterType filter_type,
                                 FilterField filter_field)
    : OpRequest(kSrcIds), neighbor_count_(neighbor_count), filter_type_(filter_type),
      filter_field_(filter_field) {
  InitParams(type, strategy);
}

void SamplingRequest::InitParams(const std::string& type, const std::string& strategy) {
  ADD_TENSOR(params_, kType, kString, 1);
  params_[kType].AddString(type);
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString(strategy);
  ADD_TENSOR(params_, kNeighborCount, kInt32, 1);
  params_[kNeighborCount].AddInt32(neighbor_count_);
  ADD_TENSOR(params_, kFilterType, kInt32, 1);
  params_[kFilterType].AddInt32(filter_type_);
  ADD_TENSOR(params_, kFilterField, kInt32, 1);
  params_[kFilterField].AddInt32(filter_field_);
  ADD_TENSOR(tensors_, kSrcIds, kInt64, 64);
  if (HasFilter()) ADD_TENSOR(tensors_, kFilterValues, kInt64, 64);
}

OpRequest* SamplingRequest::Clone() const {
  SamplingRequest* r = new SamplingRequest(Type(), Strategy(), neighbor_count_, filter_type_, filter_field_);
  if (HasCallCounter()) r->SetCallCounter(CallCounter());
  return r;
}

void SamplingRequest::SetCallCounter(int64_t call_counter) {
  params_.erase(kCallCounter);
  ADD_TENSOR(params_, kCallCounter, kInt64, 1);
  params_[kCallCounter].AddInt64(call_counter);
}
bool SamplingRequest::HasCallCounter() const { return params_.count(kCallCounter) != 0; }
int64_t SamplingRequest::CallCounter() const { return params_.at(kCallCounter).GetInt64(0); }

// DagNodeRunner-style construction (sampling_request.cc:87-136).
void SamplingRequest::Init(const Tensor::Map& params) {
  neighbor_count_ = params.at(kNeighborCount).GetInt32(0);
  auto ft = params.find(kFilterType);
  auto ff = params.find(kFilterField);
  filter_type_ = ft == params.end() ? kOperatorUnspecified : static_cast<FilterType>(ft->second.GetInt32(0));
  filter_field_ = ff == params.end() ? kFieldUnspecified : static_cast<FilterField>(ff->second.GetInt32(0));
  InitParams(params.at(kEdgeType).GetString(0), params.at(kStrategy).GetString(0));
}

void SamplingRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) {
  // a ragged upstream (FullSampler) feeds its values: one request row per neighbour (TensorMap::Find, tensor_map.cc:57-71)
  auto dense = tensors.find(kSrcIds);
  const Tensor& ids = dense != tensors.end() ? dense->second : sparse_tensors.at(kSrcIds).Values();
  Set(ids.GetInt64(), ids.Size());
  if (HasFilter()) {
    // Filter::FillValues, filter.cc:53-67
    const Tensor& values = tensors.at(kFilterValues);
    const int32_t filter_size = values.Size();
    if (filter_size != 0) {
      const int32_t fanout = ids.Size() / filter_size;
      Tensor& mine = tensors_[kFilterValues];
      for (int32_t i = 0; i < filter_size; ++i) {
        for (int32_t j = 0; j < fanout; ++j) mine.AddInt64(values.GetInt64(i));
      }
    }
  }
}

void SamplingRequest::SetFilterValues(const int64_t* values, int32_t count) {
  if (HasFilter()) tensors_[kFilterValues].AddInt64(values, values + count);
}

const int64_t* SamplingRequest::GetFilterValues() const {
  auto it = tensors_.find(kFilterValues);
  return (it == tensors_.end() || it->second.Size() != BatchSize()) ? nullptr : it->second.GetInt64();
}

void SamplingRequest::Set(const int64_t* src_ids, int32_t batch_size) {
  tensors_[kSrcIds].AddInt64(src_ids, src_ids + batch_size);
}

const std::string& SamplingRequest::Type() const { return params_.at(kType).GetString(0); }
const std::string& SamplingRequest::Strategy() const { return params_.at(kOpName).GetString(0); }
int32_t SamplingRequest::BatchSize() const { return tensors_.at(kSrcIds).Size(); }
const int64_t* SamplingRequest::GetSrcIds() const { return tensors_.at(kSrcIds).GetInt64(); }
const int64_t* SamplingRequest::GetRngRows() const {
  auto it = tensors_.find(kRngRows);
  return (it == tensors_.end() || it->second.Size() != BatchSize()) ? nullptr : it->second.GetInt64();
}

SamplingResponse::SamplingResponse() : OpResponse() {}

void SamplingResponse::Swap(OpResponse& right) {
  OpResponse::Swap(right);
  std::swap(shape_, static_cast<SamplingResponse&>(right).shape_);
}

void SamplingResponse::SetShape(size_t dim1, size_t dim2) {
  batch_size_ = (int32_t)dim1;
  ADD_TENSOR(params_, kNeighborCount, kInt32, 1);
  params_[kNeighborCount].AddInt32((int32_t)dim2);
  shape_ = Shape(dim1, dim2);
}

void SamplingResponse::SetShape(size_t dim1, size_t dim2, const std::vector<int32_t>& segments) {
  batch_size_ = (int32_t)dim1;
  ADD_TENSOR(params_, kNeighborCount, kInt32, 1);
  params_[kNeighborCount].AddInt32((int32_t)dim2);
  shape_ = Shape(dim1, dim2, segments);
}

void SamplingResponse::InitNeighborIds() { ADD_TENSOR(tensors_, kNodeIds, kInt64, (int32_t)shape_.size); }
void SamplingResponse::InitEdgeIds() { ADD_TENSOR(tensors_, kEdgeIds, kInt64, (int32_t)shape_.size); }
void SamplingResponse::AppendNeighborId(int64_t id) { tensors_[kNodeIds].AddInt64(id); }
void SamplingResponse::AppendEdgeId(int64_t id) { tensors_[kEdgeIds].AddInt64(id); }

void SamplingResponse::FillWith(int64_t neighbor_id, int64_t edge_id) {
  for (size_t i = 0; i < shape_.dim2; ++i) tensors_[kNodeIds].AddInt64(neighbor_id);
  if (tensors_.count(kEdgeIds)) {
    for (size_t i = 0; i < shape_.dim2; ++i) tensors_[kEdgeIds].AddInt64(edge_id);
  }
}

void SamplingResponse::ResizeDense() {
  tensors_[kNodeIds].Resize((int32_t)shape_.size);
  tensors_[kEdgeIds].Resize((int32_t)shape_.size);
}

void SamplingResponse::ResizeNeighborIds() { tensors_[kNodeIds].Resize((int32_t)shape_.size); }

int64_t* SamplingResponse::GetNeighborIds() {
  auto it = tensors_.find(kNodeIds);
  return it == tensors_.end() ? nullptr : it->second.MutableInt64();
}
int64_t* SamplingResponse::GetEdgeIds() {
  auto it = tensors_.find(kEdgeIds);
  return it == tensors_.end() ? nullptr : it->second.MutableInt64();
}
const int64_t* SamplingResponse::GetNeighborIds() const {
  auto it = tensors_.find(kNodeIds);
  return it == tensors_.end() ? nullptr : it->second.GetInt64();
}
const int64_t* SamplingResponse::GetEdgeIds() const {
  auto it = tensors_.find(kEdgeIds);
  return it == tensors_.end() ? nullptr : it->second.GetInt64();
}

#define REGISTER_SAMPLING_REQUEST(Type) REGISTER_REQUEST(Type##Sampler, SamplingRequest, SamplingResponse)
REGISTER_SAMPLING_REQUEST(Random)
REGISTER_SAMPLING_REQUEST(RandomWithoutReplacement)
REGISTER_SAMPLING_REQUEST(Topk)
REGISTER_SAMPLING_REQUEST(EdgeWeight)
REGISTER_SAMPLING_REQUEST(InDegree)
REGISTER_SAMPLING_REQUEST(Full)
REGISTER_SAMPLING_REQUEST(RandomNegative)
REGISTER_SAMPLING_REQUEST(InDegreeNegative)
REGISTER_SAMPLING_REQUEST(SoftInDegreeNegative)
REGISTER_SAMPLING_REQUEST(NodeWeightNegative)
#undef REGISTER_SAMPLING_REQUEST

// ------------------------------------------------------------- aggregating --
AggregatingRequest::AggregatingRequest() : OpRequest(kNodeIds), cursor_(0), num_segments_(0) {}

AggregatingRequest::AggregatingRequest(const std::string& type, const std::string& strategy)
    : OpRequest(kNodeIds), cursor_(0), num_segments_(0) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString(strategy);
  ADD_TENSOR(params_, kNodeType, kString, 1);
  params_[kNodeType].AddString(type);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, 64);
  ADD_TENSOR(tensors_, kSegmentIds, kInt32, 64);
}

OpRequest* AggregatingRequest::Clone() const {
  AggregatingRequest* req = new AggregatingRequest(Type(), Strategy());
  req->num_segments_ = num_segments_;
  return req;
}

void AggregatingRequest::Set(const int64_t* node_ids, const int32_t* segment_ids, int32_t num_ids,
                             int32_t num_segments) {
  tensors_[kNodeIds].AddInt64(node_ids, node_ids + num_ids);
  tensors_[kSegmentIds].AddInt32(segment_ids, segment_ids + num_ids);
  num_segments_ = num_segments;
}

const std::string& AggregatingRequest::Type() const { return params_.at(kNodeType).GetString(0); }
const std::string& AggregatingRequest::Strategy() const { return params_.at(kOpName).GetString(0); }
int32_t AggregatingRequest::NumIds() const { return tensors_.at(kNodeIds).Size(); }
const int64_t* AggregatingRequest::NodeIds() const { return tensors_.at(kNodeIds).GetInt64(); }
const int32_t* AggregatingRequest::SegmentIds() const { return tensors_.at(kSegmentIds).GetInt32(); }

bool AggregatingRequest::Next(int64_t* node_id, int32_t* segment_id) {
  if (cursor_ >= NumIds()) return false;
  *node_id = tensors_.at(kNodeIds).GetInt64(cursor_);
  *segment_id = tensors_.at(kSegmentIds).GetInt32(cursor_);
  ++cursor_;
  return true;
}

bool AggregatingRequest::SegmentEnd(int32_t segment_id) const {
  if (cursor_ >= NumIds()) return true;
  return tensors_.at(kSegmentIds).GetInt32(cursor_) != segment_id;
}

AggregatingResponse::AggregatingResponse() : OpResponse(), emb_dim_(0) {}

void AggregatingResponse::Swap(OpResponse& right) {
  OpResponse::Swap(right);
  AggregatingResponse& r = static_cast<AggregatingResponse&>(right);
  std::swap(name_, r.name_);
  std::swap(emb_dim_, r.emb_dim_);
}

void AggregatingResponse::SetName(const std::string& name) {
  name_ = name;
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString(name_);
  ADD_TENSOR(tensors_, kFloatAttrKey, kFloat, 64);
  ADD_TENSOR(tensors_, kSegments, kInt32, 64);
}

void AggregatingResponse::SetEmbeddingDim(int32_t dim) {
  emb_dim_ = dim;
  ADD_TENSOR(params_, kSideInfo, kInt32, 1);
  params_[kSideInfo].AddInt32(dim);
}

void AggregatingResponse::SetNumSegments(int32_t num_segments) { batch_size_ = num_segments; }

void AggregatingResponse::AppendEmbedding(const float* value) {
  tensors_[kFloatAttrKey].AddFloat(value, value + emb_dim_);
}
void AggregatingResponse::AppendSegment(int32_t size) { tensors_[kSegments].AddInt32(size); }
const float* AggregatingResponse::Embeddings() const { return tensors_.at(kFloatAttrKey).GetFloat(); }
const int32_t* AggregatingResponse::Segments() const { return tensors_.at(kSegments).GetInt32(); }

float* AggregatingResponse::MutableEmbeddings() {
  Tensor& t = tensors_[kFloatAttrKey];
  t.Resize(batch_size_ * emb_dim_);
  return t.MutableFloat();
}
int32_t* AggregatingResponse::MutableSegments() {
  Tensor& t = tensors_[kSegments];
  t.Resize(batch_size_);
  return t.MutableInt32();
}

REGISTER_REQUEST(SumAggregator, AggregatingRequest, AggregatingResponse)
REGISTER_REQUEST(MeanAggregator, AggregatingRequest, AggregatingResponse)
REGISTER_REQUEST(MaxAggregator, AggregatingRequest, AggregatingResponse)
REGISTER_REQUEST(MinAggregator, AggregatingRequest, AggregatingResponse)
REGISTER_REQUEST(ProdAggregator, AggregatingRequest, AggregatingResponse)

}  // namespace graphlearn
