// "LookupNodes" / "LookupEdges" / "GetDegree" operators (see graph_request.h).
// Unknown node ids and out-of-range edge ids (the -1 of a default-filled sample)
// yield the Default* flags, like AttributeValue::Default (core/io/element_value.cc:26-50).
#include <atomic>
#include <cfloat>
#include <cmath>

#include "glx.h"
#include "graphlearn/config.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"

namespace graphlearn {

// ------------------------------------------------------------ LookupNodes --
LookupNodesRequest::LookupNodesRequest() : OpRequest(kNodeIds), cursor_(0) {}

LookupNodesRequest::LookupNodesRequest(const std::string& node_type) : OpRequest(kNodeIds), cursor_(0) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("LookupNodes");
  ADD_TENSOR(params_, kNodeType, kString, 1);
  params_[kNodeType].AddString(node_type);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, 64);
}

OpRequest* LookupNodesRequest::Clone() const { return new LookupNodesRequest(NodeType()); }

namespace {
// The ids a DAG edge delivers under `key`: a dense tensor, or the values of a ragged one
// (graph_lookup_request.cc:368-385).
const Tensor* IdsOf(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors, const char* key,
                    const Tensor** segments = nullptr) {
  if (segments) *segments = nullptr;
  auto it = tensors.find(key);
  if (it != tensors.end()) return &it->second;
  auto sp = sparse_tensors.find(key);
  if (sp == sparse_tensors.end()) return nullptr;
  if (segments) *segments = &sp->second.Segments();
  return &sp->second.Values();
}
}  // namespace

// DagNodeRunner-style construction (graph_lookup_request.cc:353-385): the property lookup every traversal node of a
// query gets (python/gsl/dag_node.py:455-461).
void LookupNodesRequest::Init(const Tensor::Map& params) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("LookupNodes");
  ADD_TENSOR(params_, kNodeType, kString, 1);
  params_[kNodeType].AddString(params.at(kNodeType).GetString(0));
  ADD_TENSOR(tensors_, kNodeIds, kInt64, 64);
}

void LookupNodesRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) {
  const Tensor* ids = IdsOf(tensors, sparse_tensors, kNodeIds);
  if (ids) Set(ids->GetInt64(), ids->Size());
}

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

// ------------------------------------------------------------ LookupEdges --
LookupEdgesRequest::LookupEdgesRequest() : OpRequest(kSrcIds), cursor_(0) {}

LookupEdgesRequest::LookupEdgesRequest(const std::string& edge_type) : OpRequest(kSrcIds), cursor_(0) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("LookupEdges");
  ADD_TENSOR(params_, kEdgeType, kString, 1);
  params_[kEdgeType].AddString(edge_type);
  ADD_TENSOR(tensors_, kEdgeIds, kInt64, 64);
  ADD_TENSOR(tensors_, kSrcIds, kInt64, 64);
}

OpRequest* LookupEdgesRequest::Clone() const { return new LookupEdgesRequest(EdgeType()); }

// graph_lookup_request.cc:234-308.  The edge ids come from a sampler ([batch * k] dense, or ragged), the src ids
// from the node the sampler started at ([batch]): each src id is repeated over its row -- by the row's count for
// a ragged input, by the node's neighbour count for a dense one.
void LookupEdgesRequest::Init(const Tensor::Map& params) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("LookupEdges");
  ADD_TENSOR(params_, kEdgeType, kString, 1);
  params_[kEdgeType].AddString(params.at(kEdgeType).GetString(0));
  auto nbc = params.find(kNeighborCount);
  if (nbc != params.end()) {
    ADD_TENSOR(params_, kNeighborCount, kInt32, 1);
    params_[kNeighborCount].AddInt32(nbc->second.GetInt32(0));
  }
  ADD_TENSOR(tensors_, kEdgeIds, kInt64, 64);
  ADD_TENSOR(tensors_, kSrcIds, kInt64, 64);
}

void LookupEdgesRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) {
  const Tensor* segments = nullptr;
  const Tensor* edge_ids = IdsOf(tensors, sparse_tensors, kEdgeIds, &segments);
  const Tensor* src_ids = IdsOf(tensors, sparse_tensors, kSrcIds);
  if (!edge_ids || !src_ids) return;  // the operator reports the size mismatch
  Tensor& mine = tensors_[kSrcIds];
  tensors_[kEdgeIds].AddInt64(edge_ids->GetInt64(), edge_ids->GetInt64() + edge_ids->Size());
  const int32_t src_size = src_ids->Size();
  if (edge_ids->Size() == src_size) {
    mine.AddInt64(src_ids->GetInt64(), src_ids->GetInt64() + src_size);
  } else if (segments) {
    for (int32_t i = 0; i < src_size && i < segments->Size(); ++i) {
      for (int32_t j = 0; j < segments->GetInt32(i); ++j) mine.AddInt64(src_ids->GetInt64(i));
    }
  } else if (params_.count(kNeighborCount)) {
    const int32_t k = params_.at(kNeighborCount).GetInt32(0);
    for (int32_t i = 0; i < src_size; ++i) {
      for (int32_t j = 0; j < k; ++j) mine.AddInt64(src_ids->GetInt64(i));
    }
  }
}

void LookupEdgesRequest::Set(const int64_t* edge_ids, const int64_t* src_ids, int32_t batch_size) {
  tensors_[kEdgeIds].AddInt64(edge_ids, edge_ids + batch_size);
  tensors_[kSrcIds].AddInt64(src_ids, src_ids + batch_size);
}

const std::string& LookupEdgesRequest::EdgeType() const { return params_.at(kEdgeType).GetString(0); }
int32_t LookupEdgesRequest::Size() const { return tensors_.at(kEdgeIds).Size(); }
const int64_t* LookupEdgesRequest::EdgeIds() const { return tensors_.at(kEdgeIds).GetInt64(); }
const int64_t* LookupEdgesRequest::SrcIds() const { return tensors_.at(kSrcIds).GetInt64(); }

bool LookupEdgesRequest::Next(int64_t* edge_id, int64_t* src_id) const {
  if (cursor_ >= Size()) return false;
  *edge_id = tensors_.at(kEdgeIds).GetInt64(cursor_);
  *src_id = tensors_.at(kSrcIds).GetInt64(cursor_);
  ++cursor_;
  return true;
}

// --------------------------------------------------------- LookupResponse --
LookupResponse::LookupResponse() : OpResponse() {}

void LookupResponse::SetSideInfo(const io::SideInfo* info, int32_t batch_size) {
  info_ = *info;
  batch_size_ = batch_size;
  strings_.clear();
  strings_.reserve((size_t)batch_size * (info_.s_num > 0 ? info_.s_num : 0));
  tensors_.clear();
  if (info_.IsWeighted()) ADD_TENSOR(tensors_, kWeightKey, kFloat, batch_size);
  if (info_.IsLabeled()) ADD_TENSOR(tensors_, kLabelKey, kInt32, batch_size);
  if (info_.IsTimestamped()) ADD_TENSOR(tensors_, kTimestampKey, kInt64, batch_size);
  if (info_.i_num > 0) ADD_TENSOR(tensors_, kIntAttrKey, kInt64, batch_size * info_.i_num);
  if (info_.f_num > 0) ADD_TENSOR(tensors_, kFloatAttrKey, kFloat, batch_size * info_.f_num);
}

void LookupResponse::AppendWeight(float weight) {
  if (info_.IsWeighted()) tensors_[kWeightKey].AddFloat(weight);
}
void LookupResponse::AppendLabel(int32_t label) {
  if (info_.IsLabeled()) tensors_[kLabelKey].AddInt32(label);
}
void LookupResponse::AppendTimestamp(int64_t timestamp) {
  if (info_.IsTimestamped()) tensors_[kTimestampKey].AddInt64(timestamp);
}

float* LookupResponse::ResizeWeights() {
  if (!info_.IsWeighted()) return nullptr;
  Tensor& t = tensors_[kWeightKey];
  t.Resize(batch_size_);
  return t.MutableFloat();
}
int32_t* LookupResponse::ResizeLabels() {
  if (!info_.IsLabeled()) return nullptr;
  Tensor& t = tensors_[kLabelKey];
  t.Resize(batch_size_);
  return t.MutableInt32();
}
int64_t* LookupResponse::ResizeTimestamps() {
  if (!info_.IsTimestamped()) return nullptr;
  Tensor& t = tensors_[kTimestampKey];
  t.Resize(batch_size_);
  return t.MutableInt64();
}
int64_t* LookupResponse::ResizeIntAttrs() {
  if (info_.i_num <= 0) return nullptr;
  Tensor& t = tensors_[kIntAttrKey];
  t.Resize(batch_size_ * info_.i_num);
  return t.MutableInt64();
}

void LookupResponse::AppendAttribute(const int64_t* ints, const float* floats, const std::string* strings) {
  if (info_.i_num > 0) {
    Tensor& t = tensors_[kIntAttrKey];
    for (int32_t i = 0; i < info_.i_num; ++i) t.AddInt64(ints ? ints[i] : GLOBAL_FLAG(DefaultIntAttribute));
  }
  if (info_.f_num > 0) {
    Tensor& t = tensors_[kFloatAttrKey];
    for (int32_t i = 0; i < info_.f_num; ++i) t.AddFloat(floats ? floats[i] : GLOBAL_FLAG(DefaultFloatAttribute));
  }
  for (int32_t i = 0; i < info_.s_num; ++i) strings_.push_back(strings ? strings[i] : GLOBAL_FLAG(DefaultStringAttribute));
}

namespace {
template <class T>
const T* DataOrNull(const Tensor::Map& m, const char* key, const T* (Tensor::*get)() const) {
  auto it = m.find(key);
  return it == m.end() ? nullptr : (it->second.*get)();
}
}  // namespace

const float* LookupResponse::Weights() const { return DataOrNull<float>(tensors_, kWeightKey, &Tensor::GetFloat); }
const int32_t* LookupResponse::Labels() const { return DataOrNull<int32_t>(tensors_, kLabelKey, &Tensor::GetInt32); }
const int64_t* LookupResponse::Timestamps() const {
  return DataOrNull<int64_t>(tensors_, kTimestampKey, &Tensor::GetInt64);
}
const int64_t* LookupResponse::IntAttrs() const { return DataOrNull<int64_t>(tensors_, kIntAttrKey, &Tensor::GetInt64); }
const float* LookupResponse::FloatAttrs() const { return DataOrNull<float>(tensors_, kFloatAttrKey, &Tensor::GetFloat); }

void LookupResponse::SetShape(int32_t batch_size, int32_t float_attr_num) {
  batch_size_ = batch_size;
  info_.f_num = float_attr_num;
  tensors_.erase(kFloatAttrKey);
  ADD_TENSOR(tensors_, kFloatAttrKey, kFloat, batch_size * float_attr_num);
  tensors_[kFloatAttrKey].Resize(batch_size * float_attr_num);
}

float* LookupResponse::MutableFloatAttrs() { return tensors_[kFloatAttrKey].MutableFloat(); }

REGISTER_REQUEST(LookupNodes, LookupNodesRequest, LookupResponse)
REGISTER_REQUEST(LookupEdges, LookupEdgesRequest, LookupResponse)

// -------------------------------------------------------------- GetDegree --
GetDegreeRequest::GetDegreeRequest() : OpRequest(kNodeIds) {}

GetDegreeRequest::GetDegreeRequest(const std::string& edge_type, NodeFrom node_from) : OpRequest(kNodeIds) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("GetDegree");
  ADD_TENSOR(params_, kNodeFrom, kInt32, 1);
  params_[kNodeFrom].AddInt32((int32_t)node_from);
  ADD_TENSOR(params_, kEdgeType, kString, 1);
  params_[kEdgeType].AddString(edge_type);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, 64);
}

OpRequest* GetDegreeRequest::Clone() const { return new GetDegreeRequest(EdgeType(), GetNodeFrom()); }
// graph_lookup_request.cc:641-671: the degree node a traversal gets per outgoing hop (dag_node.py:66-75).
void GetDegreeRequest::Init(const Tensor::Map& params) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("GetDegree");
  ADD_TENSOR(params_, kNodeFrom, kInt32, 1);
  params_[kNodeFrom].AddInt32(params.at(kNodeFrom).GetInt32(0));
  ADD_TENSOR(params_, kEdgeType, kString, 1);
  params_[kEdgeType].AddString(params.at(kEdgeType).GetString(0));
  ADD_TENSOR(tensors_, kNodeIds, kInt64, 64);
}
void GetDegreeRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) {
  const Tensor* ids = IdsOf(tensors, sparse_tensors, kNodeIds);
  if (ids) Set(ids->GetInt64(), ids->Size());
}
NodeFrom GetDegreeRequest::GetNodeFrom() const {
  auto it = params_.find(kNodeFrom);
  return it == params_.end() ? kEdgeSrc : (NodeFrom)it->second.GetInt32(0);
}
void GetDegreeRequest::Set(const int64_t* node_ids, int32_t batch_size) {
  tensors_[kNodeIds].AddInt64(node_ids, node_ids + batch_size);
}
const std::string& GetDegreeRequest::EdgeType() const { return params_.at(kEdgeType).GetString(0); }
int32_t GetDegreeRequest::Size() const { return tensors_.at(kNodeIds).Size(); }
const int64_t* GetDegreeRequest::NodeIds() const { return tensors_.at(kNodeIds).GetInt64(); }

GetDegreeResponse::GetDegreeResponse() : OpResponse() {}
void GetDegreeResponse::InitDegrees(int32_t batch_size) {
  batch_size_ = batch_size;
  tensors_.erase(kDegrees);
  ADD_TENSOR(tensors_, kDegrees, kInt32, batch_size);
  tensors_[kDegrees].Resize(batch_size);
}
const int32_t* GetDegreeResponse::GetDegrees() const { return tensors_.at(kDegrees).GetInt32(); }
int32_t* GetDegreeResponse::MutableDegrees() { return tensors_[kDegrees].MutableInt32(); }

REGISTER_REQUEST(GetDegree, GetDegreeRequest, GetDegreeResponse)

// ------------------------------------------------------ GetCount / GetStats --
GetCountRequest::GetCountRequest() : OpRequest() {
  DisableShard();  // carries no ids to partition: every server answers for itself (graph_store.cc:278-293)
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("GetCount");
}
GetCountResponse::GetCountResponse() : OpResponse() {}
void GetCountResponse::Init(int32_t type_num) {
  tensors_.erase(kCount);
  ADD_TENSOR(tensors_, kCount, kInt32, type_num);
}
void GetCountResponse::Append(int32_t count) { tensors_[kCount].AddInt32(count); }
const int32_t* GetCountResponse::Count() const { return tensors_.at(kCount).GetInt32(); }
int32_t GetCountResponse::Size() const {
  auto it = tensors_.find(kCount);
  return it == tensors_.end() ? 0 : it->second.Size();
}

GetStatsRequest::GetStatsRequest() : OpRequest() {
  DisableShard();  // answered from the store's statistics, never partitioned (stats_getter.cc:25-48)
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("GetStats");
}
GetStatsResponse::GetStatsResponse() : OpResponse() {}
void GetStatsResponse::SetCounts(const Counts& counts) {
  for (const auto& it : counts) {
    tensors_.erase(it.first);
    ADD_TENSOR(tensors_, it.first, kInt32, (int32_t)it.second.size());
    for (int32_t c : it.second) tensors_[it.first].AddInt32(c);
  }
}
Counts GetStatsResponse::GetCounts() const {
  Counts out;
  for (const auto& it : tensors_) {
    const int32_t* p = it.second.GetInt32();
    out[it.first] = std::vector<int32_t>(p, p + it.second.Size());
  }
  return out;
}

REGISTER_REQUEST(GetCount, GetCountRequest, GetCountResponse)
REGISTER_REQUEST(GetStats, GetStatsRequest, GetStatsResponse)

// -------------------------------------------------------------- RandomWalk --
RandomWalkRequest::RandomWalkRequest() : OpRequest(kSrcIds) {}

RandomWalkRequest::RandomWalkRequest(const std::string& type, float p, float q, int32_t walk_len)
    : OpRequest(kSrcIds) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("RandomWalk");
  ADD_TENSOR(params_, kEdgeType, kString, 1);
  params_[kEdgeType].AddString(type);
  ADD_TENSOR(params_, kSideInfo, kFloat, 2);
  params_[kSideInfo].AddFloat(p);
  params_[kSideInfo].AddFloat(q);
  ADD_TENSOR(params_, kDistances, kInt32, 1);
  params_[kDistances].AddInt32(walk_len);
  ADD_TENSOR(tensors_, kSrcIds, kInt64, 64);
}

OpRequest* RandomWalkRequest::Clone() const {
  RandomWalkRequest* r = new RandomWalkRequest(Type(), P(), Q(), WalkLen());
  if (HasCallCounter()) r->SetCallCounter(CallCounter());
  return r;
}
// random_walk_request.cc:90-132: the node `.random_walk(edge_type, walk_len, p, q)` of a query.  Its second in-edge
// (the parents, kNodeIds) is the reference operator's own bookkeeping between steps; all steps run in one device
// call here.
void RandomWalkRequest::Init(const Tensor::Map& params) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("RandomWalk");
  ADD_TENSOR(params_, kEdgeType, kString, 1);
  params_[kEdgeType].AddString(params.at(kEdgeType).GetString(0));
  ADD_TENSOR(params_, kSideInfo, kFloat, 2);
  params_[kSideInfo].AddFloat(params.at(kSideInfo).GetFloat(0));
  params_[kSideInfo].AddFloat(params.at(kSideInfo).GetFloat(1));
  ADD_TENSOR(params_, kDistances, kInt32, 1);
  params_[kDistances].AddInt32(params.at(kDistances).GetInt32(0));
  ADD_TENSOR(tensors_, kSrcIds, kInt64, 64);
}
void RandomWalkRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map&) {
  const Tensor& src = tensors.at(kSrcIds);
  Set(src.GetInt64(), src.Size());
}
void RandomWalkRequest::Set(const int64_t* src_ids, int32_t batch_size) {
  tensors_[kSrcIds].AddInt64(src_ids, src_ids + batch_size);
}
const std::string& RandomWalkRequest::Type() const { return params_.at(kEdgeType).GetString(0); }
float RandomWalkRequest::P() const { return params_.at(kSideInfo).GetFloat(0); }
float RandomWalkRequest::Q() const { return params_.at(kSideInfo).GetFloat(1); }
int32_t RandomWalkRequest::WalkLen() const { return params_.at(kDistances).GetInt32(0); }
bool RandomWalkRequest::IsDeepWalk() const {
  return std::fabs(P() - 1.0f) < 32 * FLT_EPSILON && std::fabs(Q() - 1.0f) < 32 * FLT_EPSILON;
}
int32_t RandomWalkRequest::BatchSize() const { return tensors_.at(kSrcIds).Size(); }
const int64_t* RandomWalkRequest::GetSrcIds() const { return tensors_.at(kSrcIds).GetInt64(); }
void RandomWalkRequest::SetCallCounter(int64_t call_counter) {
  params_.erase(kCallCounter);
  ADD_TENSOR(params_, kCallCounter, kInt64, 1);
  params_[kCallCounter].AddInt64(call_counter);
}
bool RandomWalkRequest::HasCallCounter() const { return params_.count(kCallCounter) != 0; }
int64_t RandomWalkRequest::CallCounter() const { return params_.at(kCallCounter).GetInt64(0); }

RandomWalkResponse::RandomWalkResponse() : OpResponse() {}
void RandomWalkResponse::InitWalks(int32_t batch_size, int32_t walk_len) {
  batch_size_ = batch_size;
  walk_len_ = walk_len;
  tensors_.erase(kNodeIds);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, batch_size * walk_len);
  tensors_[kNodeIds].Resize(batch_size * walk_len);
}
const int64_t* RandomWalkResponse::GetWalks() const { return tensors_.at(kNodeIds).GetInt64(); }
int64_t* RandomWalkResponse::MutableWalks() { return tensors_[kNodeIds].MutableInt64(); }

REGISTER_REQUEST(RandomWalk, RandomWalkRequest, RandomWalkResponse)

namespace op {

class NodeLookuper : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const LookupNodesRequest* request = static_cast<const LookupNodesRequest*>(req);
    LookupResponse* response = static_cast<LookupResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    Noder* noder = graph_store_->GetNoder(request->NodeType());
    const io::SideInfo* info = noder->GetSideInfo();
    const int32_t n = request->Size();
    const int64_t* ids = request->NodeIds();
    // float attributes: one device gather for the whole batch
    const glx_features* f = noder->Device();
    io::SideInfo host_side = *info;  // everything except the float block is appended per element
    host_side.f_num = 0;
    response->SetSideInfo(&host_side, n);
    // Host-resident columns, column by column: ONE index probe per id, and none at all for a type that only has float
    // attributes (the common GNN case: a query's lookup of 1 M vertices spent 16 ms here, per element, before its
    // 5 ms device gather -- round 5).  Values are NodeStorage::GetWeight / GetLabel / GetTimestamp / GetAttribute's
    // (memory_node_storage.cc:88-138): the Default* flags for an unknown id.
    float* weights = response->ResizeWeights();
    int32_t* labels = response->ResizeLabels();
    int64_t* timestamps = response->ResizeTimestamps();
    int64_t* ints = response->ResizeIntAttrs();
    if (weights || labels || timestamps || ints || info->s_num > 0) {
      std::vector<int32_t> rows((size_t)n);
      for (int32_t i = 0; i < n; ++i) rows[i] = noder->RowOf(ids[i]);
      if (weights) {
        const float dflt = GLOBAL_FLAG(DefaultWeight);
        for (int32_t i = 0; i < n; ++i) weights[i] = rows[i] < 0 ? dflt : noder->WeightAt(rows[i]);
      }
      if (labels) {
        const int32_t dflt = (int32_t)GLOBAL_FLAG(DefaultLabel);
        for (int32_t i = 0; i < n; ++i) labels[i] = rows[i] < 0 ? dflt : noder->LabelAt(rows[i]);
      }
      if (timestamps) {
        const int64_t dflt = GLOBAL_FLAG(DefaultTimestamp);
        for (int32_t i = 0; i < n; ++i) timestamps[i] = rows[i] < 0 ? dflt : noder->TimestampAt(rows[i]);
      }
      if (ints) {
        const int32_t k = info->i_num;
        const int64_t dflt = GLOBAL_FLAG(DefaultIntAttribute);
        for (int32_t i = 0; i < n; ++i) {
          const int64_t* src = rows[i] < 0 ? nullptr : noder->GetIntAttrs(rows[i]);
          for (int32_t j = 0; j < k; ++j) ints[(size_t)i * k + j] = src ? src[j] : dflt;
        }
      }
      if (info->s_num > 0) {
        std::vector<std::string>* strings = response->MutableStringAttrs();
        const std::string& dflt = GLOBAL_FLAG(DefaultStringAttribute);
        for (int32_t i = 0; i < n; ++i) {
          const std::string* src = rows[i] < 0 ? nullptr : noder->GetStringAttrs(rows[i]);
          for (int32_t j = 0; j < info->s_num; ++j) strings->push_back(src ? src[j] : dflt);
        }
      }
    }
    if (info->f_num > 0) {
      if (!f) return error::InvalidArgument("node type '" + request->NodeType() + "' is not built on the device");
      response->SetShape(n, info->f_num);
      int rc = glx_lookup(f, ids, n, GLOBAL_FLAG(DefaultFloatAttribute), response->MutableFloatAttrs(),
                          GLX_PTR_HOST, nullptr);
      return error::FromGlx(rc);
    }
    return Status::OK();
  }
};

class EdgeLookuper : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const LookupEdgesRequest* request = static_cast<const LookupEdgesRequest*>(req);
    LookupResponse* response = static_cast<LookupResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    Graph* graph = graph_store_->GetGraph(request->EdgeType());
    const io::SideInfo* info = graph->GetSideInfo();
    const int32_t n = request->Size();
    const int64_t* eids = request->EdgeIds();
    response->SetSideInfo(info, n);
    // local_graph.cc:72-85, column by column (see NodeLookuper): an edge id outside the storage -- the -1 of a
    // default-filled sample -- reads the Default* flags
    if (float* o = response->ResizeWeights()) {
      for (int32_t i = 0; i < n; ++i) o[i] = graph->GetEdgeWeight(eids[i]);
    }
    if (int32_t* o = response->ResizeLabels()) {
      for (int32_t i = 0; i < n; ++i) o[i] = graph->GetEdgeLabel(eids[i]);
    }
    if (int64_t* o = response->ResizeTimestamps()) {
      for (int32_t i = 0; i < n; ++i) o[i] = graph->GetEdgeTimestamp(eids[i]);
    }
    if (int64_t* o = response->ResizeIntAttrs()) {
      const int32_t k = info->i_num;
      const int64_t dflt = GLOBAL_FLAG(DefaultIntAttribute);
      for (int32_t i = 0; i < n; ++i) {
        const int64_t* src = graph->GetEdgeIntAttrs(eids[i]);
        for (int32_t j = 0; j < k; ++j) o[(size_t)i * k + j] = src ? src[j] : dflt;
      }
    }
    if (info->f_num > 0) {
      const int32_t k = info->f_num;
      const float dflt = GLOBAL_FLAG(DefaultFloatAttribute);
      response->SetShape(n, k);
      float* o = response->MutableFloatAttrs();
      for (int32_t i = 0; i < n; ++i) {
        const float* src = graph->GetEdgeFloatAttrs(eids[i]);
        for (int32_t j = 0; j < k; ++j) o[(size_t)i * k + j] = src ? src[j] : dflt;
      }
    }
    if (info->s_num > 0) {
      std::vector<std::string>* strings = response->MutableStringAttrs();
      const std::string& dflt = GLOBAL_FLAG(DefaultStringAttribute);
      for (int32_t i = 0; i < n; ++i) {
        const std::string* src = graph->GetEdgeStringAttrs(eids[i]);
        for (int32_t j = 0; j < info->s_num; ++j) strings->push_back(src ? src[j] : dflt);
      }
    }
    return Status::OK();
  }
};

class DegreeGetter : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const GetDegreeRequest* request = static_cast<const GetDegreeRequest*>(req);
    GetDegreeResponse* response = static_cast<GetDegreeResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    Graph* graph = graph_store_->GetGraph(request->EdgeType());
    const int32_t n = request->Size();
    response->InitDegrees(n);
    if (!graph->Device()) return error::InvalidArgument("edge type '" + request->EdgeType() + "' is not built on the device");
    std::vector<int64_t> deg((size_t)n);
    int rc;
    if (request->GetNodeFrom() == kEdgeDst) {
      Status s = graph->EnsureInDegree();
      if (!s.ok()) return s;
      rc = glx_graph_in_degrees(graph->Device(), request->NodeIds(), n, deg.data(), GLX_PTR_HOST, nullptr);
    } else {
      rc = glx_graph_degrees(graph->Device(), request->NodeIds(), n, deg.data(), GLX_PTR_HOST, nullptr);
    }
    if (rc != GLX_OK) return error::FromGlx(rc);
    int32_t* out = response->MutableDegrees();
    for (int32_t i = 0; i < n; ++i) out[i] = (int32_t)deg[i];
    return Status::OK();
  }
};

// local_count_getter.cc:25-49
class CountGetter : public Operator {
public:
  Status Process(const OpRequest*, OpResponse* res) override {
    GetCountResponse* response = static_cast<GetCountResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    const std::vector<int32_t> local = graph_store_->GetLocalCount();
    response->Init((int32_t)local.size());
    for (int32_t c : local) response->Append(c);
    return Status::OK();
  }
};

// stats_getter.cc:25-48
class StatsGetter : public Operator {
public:
  Status Process(const OpRequest*, OpResponse* res) override {
    GetStatsResponse* response = static_cast<GetStatsResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    // the statistics are dropped by every GraphStore::Build (the counts change with what was loaded) and gathered
    // again by the first GetStats after it; with several servers that first call is collective (BuildStatistics)
    if (graph_store_->GetStatistics().GetCounts().empty()) {
      Status s = graph_store_->BuildStatistics();
      if (!s.ok()) return s;
    }
    response->SetCounts(graph_store_->GetStatistics().GetCounts());
    return Status::OK();
  }
};

// RandomWalk (core/operator/random_walk/random_walk.cc:30-276): all steps in one device call.
class RandomWalk : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const RandomWalkRequest* request = static_cast<const RandomWalkRequest*>(req);
    RandomWalkResponse* response = static_cast<RandomWalkResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    const int32_t n = request->BatchSize(), len = request->WalkLen();
    response->InitWalks(n, len);
    int64_t* walks = response->MutableWalks();
    const glx_graph* g = graph_store_->GetGraph(request->Type())->Device();
    if (!g) {  // an edge type nobody loaded: every walker is stuck at once
      for (int64_t i = 0; i < (int64_t)n * len; ++i) walks[i] = GLOBAL_FLAG(DefaultNeighborId);
      return Status::OK();
    }
    // every step consumes one call counter value, like one sub-request of the reference
    const uint64_t cc = request->HasCallCounter()
                            ? (uint64_t)request->CallCounter()
                            : call_counter_.fetch_add((uint64_t)(len > 0 ? len : 1), std::memory_order_relaxed);
    int rc = glx_random_walk(g, request->GetSrcIds(), n, len, request->P(), request->Q(),
                             GLOBAL_FLAG(DefaultFullNbrNum), GLOBAL_FLAG(DefaultWeight),
                             GLOBAL_FLAG(DefaultNeighborId), (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, walks,
                             GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }

private:
  std::atomic<uint64_t> call_counter_{0};
};

REGISTER_OPERATOR("LookupNodes", NodeLookuper)
REGISTER_OPERATOR("LookupEdges", EdgeLookuper)
REGISTER_OPERATOR("GetDegree", DegreeGetter)
REGISTER_OPERATOR("GetCount", CountGetter)
REGISTER_OPERATOR("GetStats", StatsGetter)
REGISTER_OPERATOR("RandomWalk", RandomWalk)

}  // namespace op
}  // namespace graphlearn
