// "GetNodes" / "GetEdges" operators (see graph_request.h): host-side traversal of the
// store's id lists -- seed selection for the device samplers.
#include <algorithm>
#include <map>
#include <mutex>
#include <numeric>
#include <random>

#include "graphlearn/config.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"

namespace graphlearn {

namespace {

void InitTraversal(Tensor::Map* params, const char* op, const char* type_key, const std::string& type,
                   const std::string& strategy, int32_t batch_size, int32_t epoch) {
  ADD_TENSOR((*params), kOpName, kString, 1);
  (*params)[kOpName].AddString(op);
  ADD_TENSOR((*params), type_key, kString, 1);
  (*params)[type_key].AddString(type);
  ADD_TENSOR((*params), kStrategy, kString, 1);
  (*params)[kStrategy].AddString(strategy);
  ADD_TENSOR((*params), kBatchSize, kInt32, 1);
  (*params)[kBatchSize].AddInt32(batch_size);
  ADD_TENSOR((*params), kEpoch, kInt32, 1);
  (*params)[kEpoch].AddInt32(epoch);
}
}  // namespace

GetNodesRequest::GetNodesRequest() : OpRequest(kNodeIds) {}
GetNodesRequest::GetNodesRequest(const std::string& type, const std::string& strategy, NodeFrom node_from,
                                 int32_t batch_size, int32_t epoch)
    : OpRequest(kNodeIds) {
  InitTraversal(&params_, "GetNodes", kType, type, strategy, batch_size, epoch);
  ADD_TENSOR(params_, kNodeFrom, kInt32, 1);
  params_[kNodeFrom].AddInt32((int32_t)node_from);
}
// DagNodeRunner-style construction (graph_lookup_request.cc:146-157): the root of a `g.V(t).batch(n)` query.
void GetNodesRequest::Init(const Tensor::Map& params) {
  InitTraversal(&params_, "GetNodes", kType, params.at(kNodeType).GetString(0), params.at(kStrategy).GetString(0),
                params.at(kBatchSize).GetInt32(0), params.at(kEpoch).GetInt32(0));
  ADD_TENSOR(params_, kNodeFrom, kInt32, 1);
  params_[kNodeFrom].AddInt32(params.at(kNodeFrom).GetInt32(0));
}
OpRequest* GetNodesRequest::Clone() const {
  return new GetNodesRequest(Type(), Strategy(), GetNodeFrom(), BatchSize(), Epoch());
}
const std::string& GetNodesRequest::Type() const { return params_.at(kType).GetString(0); }
const std::string& GetNodesRequest::Strategy() const { return params_.at(kStrategy).GetString(0); }
NodeFrom GetNodesRequest::GetNodeFrom() const { return (NodeFrom)params_.at(kNodeFrom).GetInt32(0); }
int32_t GetNodesRequest::BatchSize() const { return params_.at(kBatchSize).GetInt32(0); }
int32_t GetNodesRequest::Epoch() const { return params_.at(kEpoch).GetInt32(0); }

GetNodesResponse::GetNodesResponse() : OpResponse() {}
void GetNodesResponse::Init(int32_t batch_size) {
  batch_size_ = 0;
  tensors_.erase(kNodeIds);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, batch_size);
}
void GetNodesResponse::Append(int64_t node_id) {
  tensors_[kNodeIds].AddInt64(node_id);
  ++batch_size_;
}
const int64_t* GetNodesResponse::NodeIds() const { return tensors_.at(kNodeIds).GetInt64(); }

GetEdgesRequest::GetEdgesRequest() : OpRequest(kEdgeIds) {}
GetEdgesRequest::GetEdgesRequest(const std::string& edge_type, const std::string& strategy, int32_t batch_size,
                                 int32_t epoch)
    : OpRequest(kEdgeIds) {
  InitTraversal(&params_, "GetEdges", kEdgeType, edge_type, strategy, batch_size, epoch);
}
// graph_lookup_request.cc:50-60: the root of a `g.E(t).batch(n)` query.
void GetEdgesRequest::Init(const Tensor::Map& params) {
  InitTraversal(&params_, "GetEdges", kEdgeType, params.at(kEdgeType).GetString(0), params.at(kStrategy).GetString(0),
                params.at(kBatchSize).GetInt32(0), params.at(kEpoch).GetInt32(0));
}
OpRequest* GetEdgesRequest::Clone() const { return new GetEdgesRequest(EdgeType(), Strategy(), BatchSize(), Epoch()); }
const std::string& GetEdgesRequest::EdgeType() const { return params_.at(kEdgeType).GetString(0); }
const std::string& GetEdgesRequest::Strategy() const { return params_.at(kStrategy).GetString(0); }
int32_t GetEdgesRequest::BatchSize() const { return params_.at(kBatchSize).GetInt32(0); }
int32_t GetEdgesRequest::Epoch() const { return params_.at(kEpoch).GetInt32(0); }

GetEdgesResponse::GetEdgesResponse() : OpResponse() {}
void GetEdgesResponse::Init(int32_t batch_size) {
  batch_size_ = 0;
  for (const char* key : {kSrcIds, kDstIds, kEdgeIds}) {
    tensors_.erase(key);
    ADD_TENSOR(tensors_, key, kInt64, batch_size);
  }
}
void GetEdgesResponse::Append(int64_t src_id, int64_t dst_id, int64_t edge_id) {
  tensors_[kSrcIds].AddInt64(src_id);
  tensors_[kDstIds].AddInt64(dst_id);
  tensors_[kEdgeIds].AddInt64(edge_id);
  ++batch_size_;
}
const int64_t* GetEdgesResponse::SrcIds() const { return tensors_.at(kSrcIds).GetInt64(); }
const int64_t* GetEdgesResponse::DstIds() const { return tensors_.at(kDstIds).GetInt64(); }
const int64_t* GetEdgesResponse::EdgeIds() const { return tensors_.at(kEdgeIds).GetInt64(); }

REGISTER_REQUEST(GetNodes, GetNodesRequest, GetNodesResponse)
REGISTER_REQUEST(GetEdges, GetEdgesRequest, GetEdgesResponse)

namespace op {
namespace {

// Cursor over [0, size): node_generator.h State + Ordered/Shuffled/Random generators in one.
struct Traversal {
  int64_t at = 0;  // State::cursor_: ids [0, at) have been handed out (by_order) or moved into a shuffle buffer
  int32_t epoch = 0;
  std::vector<int64_t> buffer;  // ShuffleBuffer: the shuffled positions of the current window ...
  size_t buffer_at = 0;         // ... and how many of them have been handed out
  std::mt19937_64 rng;
};

// -> positions of the next batch, or OUT_OF_RANGE at an epoch boundary
Status NextPositions(Traversal* t, const std::string& strategy, int64_t size, int32_t batch_size,
                     int32_t request_epoch, std::vector<int64_t>* pos, const char* what = "nodes") {
  const std::string done = std::string("No more ") + what + " exist.";  // node_getter.cc:72,89 / edge_getter.cc
  pos->clear();
  if (strategy == "random") {
    if (size <= 0) return error::OutOfRange(done);
    for (int32_t i = 0; i < batch_size; ++i) pos->push_back((int64_t)(t->rng() % (uint64_t)size));
    return Status::OK();
  }
  if (request_epoch < t->epoch) return error::OutOfRange(done);  // node_getter.cc:71-73
  const bool shuffle = strategy == "shuffle";
  for (int32_t i = 0; i < batch_size; ++i) {
    if (!shuffle) {
      if (t->at >= size) break;
      pos->push_back(t->at++);
      continue;
    }
    // ShuffledGenerator::Next (node_generator.h:205-216): an epoch is walked in windows of ShuffleBufferSize CONSECUTIVE
    // ids, each shuffled as it is filled (ShuffleBuffer::Fill, :168-190) -- not one permutation of the whole type; a
    // window outlives the request that filled it
    if (t->buffer_at >= t->buffer.size()) {
      const int64_t window = std::min<int64_t>(size - t->at, (int64_t)std::max(GLOBAL_FLAG(ShuffleBufferSize), 1));
      t->buffer.clear();
      t->buffer_at = 0;
      if (window <= 0) break;
      t->buffer.resize((size_t)window);
      std::iota(t->buffer.begin(), t->buffer.end(), t->at);
      std::shuffle(t->buffer.begin(), t->buffer.end(), t->rng);
      t->at += window;
    }
    pos->push_back(t->buffer[t->buffer_at++]);
  }
  if (pos->empty()) {  // begin the next epoch (node_getter.cc:84-90)
    t->at = 0;
    ++t->epoch;
    t->buffer.clear();
    t->buffer_at = 0;
    return error::OutOfRange(done);
  }
  return Status::OK();
}

template <class T>
std::vector<T> FirstAppearance(const std::vector<T>& v) {  // GetAllSrcIds / GetAllDstIds order
  std::vector<T> out;
  std::unordered_map<T, char> seen;
  seen.reserve(v.size());
  for (const T& x : v) {
    if (seen.emplace(x, 1).second) out.push_back(x);
  }
  return out;
}

}  // namespace

class NodeGetter : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const GetNodesRequest* request = static_cast<const GetNodesRequest*>(req);
    GetNodesResponse* response = static_cast<GetNodesResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    response->Init(request->BatchSize());
    std::lock_guard<std::mutex> g(mtx_);
    const std::string key = request->Type() + "#" + std::to_string((int)request->GetNodeFrom()) +
                            (request->Strategy() == "shuffle" ? "#shuffled" : "");  // ordered / shuffled cursors are separate
    Slot& slot = slots_[key];
    if (slot.store != graph_store_->Uid()) {  // a new store: rebuild the id list, restart the cursor
      slot = Slot();
      slot.store = graph_store_->Uid();
      slot.state.rng.seed((uint64_t)GLOBAL_FLAG(SamplingSeed) * 0x9E3779B97F4A7C15ull + std::hash<std::string>()(key));
      if (request->GetNodeFrom() == kNode) slot.ids = graph_store_->GetNoder(request->Type())->Ids();
      else if (request->GetNodeFrom() == kEdgeSrc) slot.ids = FirstAppearance(graph_store_->GetGraph(request->Type())->SrcIds());
      else slot.ids = FirstAppearance(graph_store_->GetGraph(request->Type())->DstIds());
    }
    std::vector<int64_t> pos;
    Status s = NextPositions(&slot.state, request->Strategy(), (int64_t)slot.ids.size(), request->BatchSize(),
                             request->Epoch(), &pos);
    if (!s.ok()) return s;
    for (int64_t p : pos) response->Append(slot.ids[p]);
    return Status::OK();
  }

private:
  struct Slot {
    uint64_t store = 0;
    std::vector<int64_t> ids;
    Traversal state;
  };
  std::mutex mtx_;
  std::map<std::string, Slot> slots_;
};

class EdgeGetter : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const GetEdgesRequest* request = static_cast<const GetEdgesRequest*>(req);
    GetEdgesResponse* response = static_cast<GetEdgesResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    response->Init(request->BatchSize());
    std::lock_guard<std::mutex> g(mtx_);
    Slot& slot = slots_[request->EdgeType() + (request->Strategy() == "shuffle" ? "#shuffled" : "")];
    if (slot.store != graph_store_->Uid()) {
      slot = Slot();
      slot.store = graph_store_->Uid();
      slot.state.rng.seed((uint64_t)GLOBAL_FLAG(SamplingSeed) * 0x9E3779B97F4A7C15ull +
                          std::hash<std::string>()(request->EdgeType()));
    }
    Graph* graph = graph_store_->GetGraph(request->EdgeType());
    std::vector<int64_t> pos;
    Status s = NextPositions(&slot.state, request->Strategy(), graph->GetEdgeCount(), request->BatchSize(),
                             request->Epoch(), &pos, "edges");
    if (!s.ok()) return s;
    for (int64_t p : pos) response->Append(graph->GetSrcId(p), graph->GetDstId(p), p);  // edge id = load order
    return Status::OK();
  }

private:
  struct Slot {
    uint64_t store = 0;
    Traversal state;
  };
  std::mutex mtx_;
  std::map<std::string, Slot> slots_;
};

REGISTER_OPERATOR("GetNodes", NodeGetter)
REGISTER_OPERATOR("GetEdges", EdgeGetter)

}  // namespace op
}  // namespace graphlearn
