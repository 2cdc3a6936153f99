// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Link-time shim that lets the *verbatim* reference hot-path translation units
// (compiled in place from /root/reference by oracle/Makefile) link without
// protobuf / glog / gRPC.  It supplies, written from scratch:
//   * a std::vector-backed Tensor / SparseTensor (the reference backs them by
//     protobuf RepeatedField: graphlearn/src/service/tensor_impl.h:196-200),
//   * the OpRequest / OpResponse bases without (de)serialisation
//     (reference: graphlearn/src/service/request/op_request.cc),
//   * RequestFactory (graphlearn/src/include/op_request.h:111-136),
//   * a minimal GraphStore that hands out one Graph/Noder per type
//     (reference: graphlearn/src/core/graph/graph_store.cc:319-325),
//   * stubs for the RPC client, vineyard storages and logging.
// Nothing here restates sampler/aggregator arithmetic: that all comes from the
// reference's own .cc files.
#include <cstring>
#include <string>
#include <vector>

#include "common/base/errors.h"
#include "common/base/log.h"
#include "core/graph/graph_store.h"
#include "core/graph/storage/graph_storage.h"
#include "core/graph/storage/node_storage.h"
#include "core/graph/storage/storage_mode.h"
#include "include/client.h"
#include "include/config.h"
#include "include/op_request.h"
#include "include/sparse_tensor.h"
#include "include/tensor.h"

namespace graphlearn {

// ----------------------------------------------------------------- Tensor --
class TensorImpl {
public:
  explicit TensorImpl(DataType t) : type(t) {}
  DataType type;
  std::vector<int32_t> i32;
  std::vector<int64_t> i64;
  std::vector<float> f32;
  std::vector<double> f64;
  std::vector<std::string*> str;
  ~TensorImpl() { for (auto* s : str) delete s; }
  TensorImpl(const TensorImpl& o)
      : type(o.type), i32(o.i32), i64(o.i64), f32(o.f32), f64(o.f64) {
    for (auto* s : o.str) str.push_back(new std::string(*s));
  }
  int32_t Size() const {
    switch (type) {
      case kInt32: return i32.size();
      case kInt64: return i64.size();
      case kFloat: return f32.size();
      case kDouble: return f64.size();
      case kString: return str.size();
      default: return 0;
    }
  }
  void Reserve(int32_t n) {
    if (n <= 0) return;
    switch (type) {
      case kInt32: i32.reserve(n); break;
      case kInt64: i64.reserve(n); break;
      case kFloat: f32.reserve(n); break;
      case kDouble: f64.reserve(n); break;
      case kString: str.reserve(n); break;
      default: break;
    }
  }
  void Resize(int32_t n) {
    switch (type) {
      case kInt32: i32.resize(n); break;
      case kInt64: i64.resize(n); break;
      case kFloat: f32.resize(n); break;
      case kDouble: f64.resize(n); break;
      case kString: {
        size_t old = str.size();
        for (size_t i = n; i < old; ++i) delete str[i];
        str.resize(n);
        for (size_t i = old; i < static_cast<size_t>(n); ++i) str[i] = new std::string();
        break;
      }
      default: break;
    }
  }
};

Tensor::Tensor() : impl_(nullptr) {}
Tensor::Tensor(DataType dtype) : impl_(new TensorImpl(dtype)) {}
Tensor::Tensor(DataType dtype, int32_t capacity) : impl_(new TensorImpl(dtype)) {
  impl_->Reserve(capacity);
}
Tensor::Tensor(const Tensor& t) : impl_(t.impl_) {}
Tensor::Tensor(Tensor&& t) : impl_(std::move(t.impl_)) {}
Tensor& Tensor::operator=(const Tensor& t) { impl_ = t.impl_; return *this; }
Tensor& Tensor::operator=(Tensor&& t) { impl_ = std::move(t.impl_); return *this; }
Tensor::~Tensor() {}
DataType Tensor::DType() const { return impl_->type; }
int32_t Tensor::Size() const { return impl_ ? impl_->Size() : 0; }
void Tensor::Resize(int32_t size) { impl_->Resize(size); }
void Tensor::AddInt32(int32_t v) { impl_->i32.push_back(v); }
void Tensor::AddInt64(int64_t v) { impl_->i64.push_back(v); }
void Tensor::AddFloat(float v) { impl_->f32.push_back(v); }
void Tensor::AddDouble(double v) { impl_->f64.push_back(v); }
void Tensor::AddString(const std::string& v) { impl_->str.push_back(new std::string(v)); }
void Tensor::AddInt32(const int32_t* b, const int32_t* e) { impl_->i32.insert(impl_->i32.end(), b, e); }
void Tensor::AddInt64(const int64_t* b, const int64_t* e) { impl_->i64.insert(impl_->i64.end(), b, e); }
void Tensor::AddFloat(const float* b, const float* e) { impl_->f32.insert(impl_->f32.end(), b, e); }
void Tensor::AddDouble(const double* b, const double* e) { impl_->f64.insert(impl_->f64.end(), b, e); }
void Tensor::SetInt32(int32_t i, int32_t v) { impl_->i32[i] = v; }
void Tensor::SetInt64(int32_t i, int64_t v) { impl_->i64[i] = v; }
void Tensor::SetFloat(int32_t i, float v) { impl_->f32[i] = v; }
void Tensor::SetDouble(int32_t i, double v) { impl_->f64[i] = v; }
void Tensor::SetString(int32_t i, const std::string& v) { *impl_->str[i] = v; }
int32_t Tensor::GetInt32(int32_t i) const { return impl_->i32[i]; }
int64_t Tensor::GetInt64(int32_t i) const { return impl_->i64[i]; }
float Tensor::GetFloat(int32_t i) const { return impl_->f32[i]; }
double Tensor::GetDouble(int32_t i) const { return impl_->f64[i]; }
const std::string& Tensor::GetString(int32_t i) const { return *impl_->str[i]; }
const int32_t* Tensor::GetInt32() const { return impl_->i32.data(); }
const int64_t* Tensor::GetInt64() const { return impl_->i64.data(); }
const float* Tensor::GetFloat() const { return impl_->f32.data(); }
const double* Tensor::GetDouble() const { return impl_->f64.data(); }
const std::string* const* Tensor::GetString() const { return impl_->str.data(); }
void Tensor::Swap(Tensor& right) { std::swap(impl_, right.impl_); }
void Tensor::SwapWithProto(TensorValue*) {}

// ----------------------------------------------------------- SparseTensor --
SparseTensor::SparseTensor() : segments_(kInt32), values_(kInt64) {}
SparseTensor::SparseTensor(const Tensor& s, const Tensor& v) : segments_(s), values_(v) {}
SparseTensor::SparseTensor(Tensor&& s, Tensor&& v) : segments_(std::move(s)), values_(std::move(v)) {}
SparseTensor::SparseTensor(const SparseTensor& o) noexcept : segments_(o.segments_), values_(o.values_) {}
SparseTensor::SparseTensor(SparseTensor&& o) noexcept
    : segments_(std::move(o.segments_)), values_(std::move(o.values_)) {}
SparseTensor& SparseTensor::operator=(const SparseTensor& o) noexcept {
  segments_ = o.segments_; values_ = o.values_; return *this;
}
SparseTensor& SparseTensor::operator=(SparseTensor&& o) noexcept {
  segments_ = std::move(o.segments_); values_ = std::move(o.values_); return *this;
}
SparseTensor::~SparseTensor() {}
const Tensor& SparseTensor::Segments() const { return segments_; }
const Tensor& SparseTensor::Values() const { return values_; }
Tensor* SparseTensor::MutableSegments() { return &segments_; }
Tensor* SparseTensor::MutableValues() { return &values_; }
void SparseTensor::Swap(SparseTensor& r) { segments_.Swap(r.segments_); values_.Swap(r.values_); }
void SparseTensor::SwapWithProto(SparseTensorValue*) {}

// ------------------------------------------------- OpRequest / OpResponse --
OpRequest::OpRequest(const std::string& shard_key)
    : ShardableRequest<OpRequest>(shard_key), is_parse_from_(false) {}
std::string OpRequest::Name() const {
  auto it = params_.find(kOpName);
  return it == params_.end() ? std::string() : it->second.GetString(0);
}
OpRequest* OpRequest::Clone() const { return new OpRequest(shard_key_); }
void OpRequest::SerializeTo(void*) {}
bool OpRequest::ParseFrom(const void*) { return false; }
ShardsPtr<OpRequest> OpRequest::Partition() const {
  // Single-shard oracle: the whole request is shard 0.
  ShardsPtr<OpRequest> ret(new Shards<OpRequest>(1));
  ret->Add(0, const_cast<OpRequest*>(this), false);
  return ret;
}

OpResponse::OpResponse() : batch_size_(0), is_parse_from_(false) {}
void OpResponse::SerializeTo(void*) {}
bool OpResponse::ParseFrom(const void*) { return false; }
void OpResponse::Stitch(ShardsPtr<OpResponse>) {}
void OpResponse::Swap(OpResponse& right) {
  std::swap(batch_size_, right.batch_size_);
  params_.swap(right.params_);
  tensors_.swap(right.tensors_);
  sparse_tensors_.swap(right.sparse_tensors_);
}

void RequestFactory::Register(const std::string& name, RequestCreator req, ResponseCreator res) {
  std::lock_guard<std::mutex> g(mtx_);
  req_[name] = req;
  res_[name] = res;
}
OpRequest* RequestFactory::NewRequest(const std::string& name) {
  auto it = req_.find(name);
  return it == req_.end() ? nullptr : it->second();
}
OpResponse* RequestFactory::NewResponse(const std::string& name) {
  auto it = res_.find(name);
  return it == res_.end() ? nullptr : it->second();
}

// ------------------------------------------------------- Graph / Noder ----
namespace {

class ShimGraph : public Graph {
public:
  ShimGraph() {
    storage_ = io::IsCompressedStorageEnabled() ? io::NewCompressedMemoryGraphStorage()
                                                : io::NewMemoryGraphStorage();
  }
  ~ShimGraph() override { delete storage_; }
  Status Build(const IndexOption&) override { storage_->Build(); return Status::OK(); }
  io::GraphStorage* GetLocalStorage() override { return storage_; }
#define UNSUPPORTED(Name)                                                          \
  Status Name(const Name##Request*, Name##Response*) override {                    \
    return error::Unimplemented("oracle shim");                                    \
  }                                                                                \
  Status Name(int32_t, const Name##Request*, Name##Response*) override {           \
    return error::Unimplemented("oracle shim");                                    \
  }
  UNSUPPORTED(UpdateEdges)
  UNSUPPORTED(LookupEdges)
#undef UNSUPPORTED
private:
  io::GraphStorage* storage_;
};

class ShimNoder : public Noder {
public:
  ShimNoder() {
    storage_ = io::IsCompressedStorageEnabled() ? io::NewCompressedMemoryNodeStorage()
                                                : io::NewMemoryNodeStorage();
  }
  ~ShimNoder() override { delete storage_; }
  Status Build(const IndexOption&) override { storage_->Build(); return Status::OK(); }
  io::NodeStorage* GetLocalStorage() override { return storage_; }
#define UNSUPPORTED(Name)                                                          \
  Status Name(const Name##Request*, Name##Response*) override {                    \
    return error::Unimplemented("oracle shim");                                    \
  }                                                                                \
  Status Name(int32_t, const Name##Request*, Name##Response*) override {           \
    return error::Unimplemented("oracle shim");                                    \
  }
  UNSUPPORTED(UpdateNodes)
#undef UNSUPPORTED
  // LocalNoder::LookupNodes (core/graph/local_noder.cc:85-97), which ConditionalNegativeSampler reaches through
  // GetNodeAttributesWrapper -> "LookupNodes" (get_node_attributes_wrapper.cc:45-63).
  Status LookupNodes(const LookupNodesRequest* req, LookupNodesResponse* res) override {
    int64_t node_id = 0;
    LookupNodesRequest* request = const_cast<LookupNodesRequest*>(req);
    res->SetSideInfo(storage_->GetSideInfo(), req->Size());
    while (request->Next(&node_id)) {
      res->AppendWeight(storage_->GetWeight(node_id));
      res->AppendLabel(storage_->GetLabel(node_id));
      res->AppendTimestamp(storage_->GetTimestamp(node_id));
      res->AppendAttribute(storage_->GetAttribute(node_id).get());
    }
    return Status::OK();
  }
  Status LookupNodes(int32_t, const LookupNodesRequest*, LookupNodesResponse*) override {
    return error::Unimplemented("oracle shim");
  }
private:
  io::NodeStorage* storage_;
};

Graph* MakeShimGraph(const std::string&, const std::string&, const std::string&) {
  return new ShimGraph();
}
Noder* MakeShimNoder(const std::string&, const std::string&, const std::string&) {
  return new ShimNoder();
}

}  // namespace

GraphStore::GraphStore(Env* env)
    : env_(env),
      graphs_(new HeterDispatcher<Graph>(MakeShimGraph)),
      noders_(new HeterDispatcher<Noder>(MakeShimNoder)) {}
GraphStore::~GraphStore() { delete graphs_; delete noders_; }
Graph* GraphStore::GetGraph(const std::string& edge_type) { return graphs_->LookupOrCreate(edge_type); }
Noder* GraphStore::GetNoder(const std::string& node_type) { return noders_->LookupOrCreate(node_type); }

// ------------------------------------------------------------------ stubs --
Client::~Client() {}
Client* NewRpcClient(int32_t, bool) { return nullptr; }
Status Client::Sampling(const SamplingRequest*, SamplingResponse*) {
  return error::Unimplemented("oracle shim: no RPC");
}
Status Client::Aggregating(const AggregatingRequest*, AggregatingResponse*) {
  return error::Unimplemented("oracle shim: no RPC");
}
Status Client::RandomWalk(const RandomWalkRequest*, RandomWalkResponse*) {
  return error::Unimplemented("oracle shim: no RPC");
}
void Log(const char*) {}
void Log(const std::string&) {}

namespace io {
GraphStorage* NewVineyardGraphStorage(const std::string&, const std::string&, const std::string&) {
  return nullptr;
}
NodeStorage* NewVineyardNodeStorage(const std::string&, const std::string&, const std::string&) {
  return nullptr;
}
}  // namespace io

}  // namespace graphlearn
