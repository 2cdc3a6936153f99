// GraphStore / Graph / Noder: host staging -> CSR -> device (see graph_store.h).
#include "graphlearn/graph_store.h"

#include "glx.h"
#include "graphlearn/config.h"

namespace graphlearn {

UpdateEdgesRequest::UpdateEdgesRequest(const io::SideInfo* info, int32_t batch_size) : info_(*info) {
  values_.reserve(batch_size > 0 ? batch_size : 0);
}
void UpdateEdgesRequest::Append(const io::EdgeValue* value) { values_.push_back(*value); }
UpdateNodesRequest::UpdateNodesRequest(const io::SideInfo* info, int32_t batch_size) : info_(*info) {
  values_.reserve(batch_size > 0 ? batch_size : 0);
}
void UpdateNodesRequest::Append(const io::NodeValue* value) { values_.push_back(*value); }

// ------------------------------------------------------------------ Graph --
Graph::Graph(const std::string& type) : type_(type), dev_(nullptr) {}

Graph::~Graph() { glx_graph_destroy(dev_); }

void Graph::SetSideInfo(const io::SideInfo* info) {
  if (!info_.IsInitialized()) info_ = *info;  // first writer wins (memory_edge_storage.cc:35-39)
}

void Graph::Add(const io::EdgeValue* value) {
  src_.push_back(value->src_id);
  dst_.push_back(value->dst_id);
  if (info_.IsWeighted()) weight_.push_back(value->weight);
}

Status Graph::UpdateEdges(const UpdateEdgesRequest* req, UpdateEdgesResponse*) {
  std::lock_guard<std::mutex> g(mtx_);
  SetSideInfo(&req->GetSideInfo());
  for (const auto& v : req->Values()) Add(&v);
  return Status::OK();
}

Status Graph::Build(const IndexOption& option) {
  std::lock_guard<std::mutex> g(mtx_);
  if (dev_) return Status::OK();
  // The whole build runs on the GPU (glx_graph_build): rows = distinct source ids
  // (AutoIndex, auto_indexing.cc:21-24), edge id = insertion index
  // (memory_edge_storage.cc:53-57), and -- for weighted types built with
  // IndexOption "sort" -- rows ordered by weight descending (MemoryAdjMatrix::Sort,
  // memory_adj_matrix.cc:105-125; ties keep insertion order, which the reference's
  // std::sort leaves unspecified).
  const bool weighted = info_.IsWeighted();
  int rc = glx_graph_build(GLOBAL_FLAG(DeviceId), (int64_t)src_.size(), src_.data(), dst_.data(),
                           weighted ? weight_.data() : nullptr, nullptr,
                           (weighted && option.name == "sort") ? 1 : 0, GLX_PTR_HOST, nullptr, &dev_);
  if (rc != GLX_OK) return error::FromGlx(rc);
  // The reference keeps in/out-degree statistics by default (StorageMode bit 1,
  // config.cc:93, topo_statics.cc); InDegreeSampler needs them as alias tables.
  rc = glx_graph_enable_in_degree(dev_, nullptr);
  return error::FromGlx(rc);
}

// ------------------------------------------------------------------ Noder --
Noder::Noder(const std::string& type) : type_(type), dev_(nullptr) {}

Noder::~Noder() { glx_features_destroy(dev_); }

void Noder::SetSideInfo(const io::SideInfo* info) {
  if (!info_.IsInitialized()) info_ = *info;
}

void Noder::Add(const io::NodeValue* value) {
  if (!index_.emplace(value->id, (int32_t)ids_.size()).second) return;  // duplicate id: ignore
  ids_.push_back(value->id);
  const int32_t dim = info_.f_num;
  for (int32_t i = 0; i < dim; ++i) {
    feats_.push_back(i < (int32_t)value->attrs.size() ? value->attrs[i] : GLOBAL_FLAG(DefaultFloatAttribute));
  }
}

Status Noder::UpdateNodes(const UpdateNodesRequest* req, UpdateNodesResponse*) {
  std::lock_guard<std::mutex> g(mtx_);
  SetSideInfo(&req->GetSideInfo());
  for (const auto& v : req->Values()) Add(&v);
  return Status::OK();
}

Status Noder::Build(const IndexOption&) {
  std::lock_guard<std::mutex> g(mtx_);
  if (dev_) return Status::OK();
  if (info_.f_num <= 0) return Status::OK();  // nothing for the aggregators to read
  int rc = glx_features_create(GLOBAL_FLAG(DeviceId), (int64_t)ids_.size(), info_.f_num, feats_.data(),
                               ids_.data(), GLX_PTR_HOST, nullptr, &dev_);
  return error::FromGlx(rc);
}

// ------------------------------------------------------------- GraphStore --
GraphStore::GraphStore() {}

GraphStore::~GraphStore() {
  for (auto& it : graphs_) delete it.second;
  for (auto& it : noders_) delete it.second;
}

Graph* GraphStore::GetGraph(const std::string& edge_type) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = graphs_.find(edge_type);
  if (it == graphs_.end()) it = graphs_.emplace(edge_type, new Graph(edge_type)).first;
  return it->second;
}

Noder* GraphStore::GetNoder(const std::string& node_type) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = noders_.find(node_type);
  if (it == noders_.end()) it = noders_.emplace(node_type, new Noder(node_type)).first;
  return it->second;
}

}  // namespace graphlearn
