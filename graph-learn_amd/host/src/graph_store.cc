// GraphStore / Graph / Noder: host staging -> CSR -> device (see graph_store.h).
#include "graphlearn/graph_store.h"

#include <algorithm>
#include <numeric>

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
  const int64_t E = (int64_t)src_.size();
  // AutoIndex (auto_indexing.cc:21-24): rows in order of first appearance.
  std::unordered_map<int64_t, int32_t> row_of;
  std::vector<int64_t> ids;
  std::vector<int32_t> row(E);
  for (int64_t e = 0; e < E; ++e) {
    auto it = row_of.find(src_[e]);
    if (it == row_of.end()) {
      it = row_of.emplace(src_[e], (int32_t)ids.size()).first;
      ids.push_back(src_[e]);
    }
    row[e] = it->second;
  }
  const int64_t V = (int64_t)ids.size();
  std::vector<int64_t> row_ptr(V + 1, 0);
  for (int64_t e = 0; e < E; ++e) row_ptr[row[e] + 1]++;
  for (int64_t r = 0; r < V; ++r) row_ptr[r + 1] += row_ptr[r];
  // insertion order inside a row (counting sort is stable)
  std::vector<int64_t> slot_edge(E);
  {
    std::vector<int64_t> fill(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) slot_edge[fill[row[e]]++] = e;
  }
  const bool weighted = info_.IsWeighted();
  if (weighted && option.name == "sort") {
    // MemoryAdjMatrix::Sort (memory_adj_matrix.cc:105-125): weight descending.
    // The reference's std::sort leaves ties unspecified; ties keep insertion order here.
    for (int64_t r = 0; r < V; ++r) {
      std::stable_sort(slot_edge.begin() + row_ptr[r], slot_edge.begin() + row_ptr[r + 1],
                       [&](int64_t a, int64_t b) { return weight_[a] > weight_[b]; });
    }
  }
  std::vector<int64_t> col(E), eid(E);
  std::vector<float> w(weighted ? E : 0);
  for (int64_t s = 0; s < E; ++s) {
    const int64_t e = slot_edge[s];
    col[s] = dst_[e];
    eid[s] = e;  // edge id = insertion index (memory_edge_storage.cc:53-57)
    if (weighted) w[s] = weight_[e];
  }
  int rc = glx_graph_create(GLOBAL_FLAG(DeviceId), V, E, row_ptr.data(), col.data(), eid.data(),
                            weighted ? w.data() : nullptr, ids.data(), GLX_PTR_HOST, nullptr, &dev_);
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
