// GraphStore / Graph / Noder: host staging -> CSR -> device (see graph_store.h).
#include "graphlearn/graph_store.h"

#include <atomic>
#include <iterator>

#include "glx.h"
#include "graphlearn/config.h"

namespace graphlearn {

UpdateEdgesRequest::UpdateEdgesRequest() : OpRequest(kSrcIds) {}
UpdateEdgesRequest::UpdateEdgesRequest(const io::SideInfo* info, int32_t batch_size) : OpRequest(kSrcIds), info_(*info) {
  values_.reserve(batch_size > 0 ? batch_size : 0);
}
OpRequest* UpdateEdgesRequest::Clone() const { return new UpdateEdgesRequest(&info_, Size()); }
void UpdateEdgesRequest::Append(const io::EdgeValue* value) { values_.push_back(*value); }
bool UpdateEdgesRequest::Next(io::EdgeValue* value) {
  if (cursor_ >= Size()) return false;
  *value = values_[cursor_++];
  return true;
}
UpdateNodesRequest::UpdateNodesRequest() : OpRequest(kNodeIds) {}
UpdateNodesRequest::UpdateNodesRequest(const io::SideInfo* info, int32_t batch_size) : OpRequest(kNodeIds), info_(*info) {
  values_.reserve(batch_size > 0 ? batch_size : 0);
}
OpRequest* UpdateNodesRequest::Clone() const { return new UpdateNodesRequest(&info_, Size()); }
void UpdateNodesRequest::Append(const io::NodeValue* value) { values_.push_back(*value); }
bool UpdateNodesRequest::Next(io::NodeValue* value) {
  if (cursor_ >= Size()) return false;
  *value = values_[cursor_++];
  return true;
}

// ------------------------------------------------------------------ Graph --
Graph::Graph(const std::string& type)
    : type_(type), dev_(nullptr), neg_uniform_(nullptr), neg_in_degree_(nullptr), neg_strict_ready_(false),
      in_degree_ready_(false), global_in_degree_ready_(false) {}

Graph::~Graph() {
  glx_negative_destroy(neg_uniform_);
  glx_negative_destroy(neg_in_degree_);
  glx_graph_destroy(dev_);
}

Status Graph::Negative(bool by_in_degree, bool strict, const glx_negative** out) {
  std::lock_guard<std::mutex> g(mtx_);
  *out = nullptr;
  if (!dev_) return error::InvalidArgument("edge type '" + type_ + "' is not built on the device");
  glx_negative*& slot = by_in_degree ? neg_in_degree_ : neg_uniform_;
  if (!slot) {
    int rc = glx_negative_from_graph(dev_, by_in_degree ? 1 : 0, nullptr, &slot);
    if (rc != GLX_OK) return error::FromGlx(rc);
  }
  if (strict && !neg_strict_ready_) {
    int rc = glx_graph_enable_negative(dev_, nullptr);
    if (rc != GLX_OK) return error::FromGlx(rc);
    neg_strict_ready_ = true;
  }
  *out = slot;
  return Status::OK();
}

void Graph::SetSideInfo(const io::SideInfo* info) {
  if (!info_.IsInitialized()) info_ = *info;  // first writer wins (memory_edge_storage.cc:35-39)
}

void Graph::Add(const io::EdgeValue* value) {
  src_.push_back(value->src_id);
  dst_.push_back(value->dst_id);
  if (info_.IsWeighted()) weight_.push_back(value->weight);
  if (info_.IsLabeled()) label_.push_back(value->label);
  if (info_.IsTimestamped()) timestamp_.push_back(value->timestamp);
  if (info_.IsAttributed()) {  // short values are padded with the defaults
    for (int32_t i = 0; i < info_.i_num; ++i) {
      i_attrs_.push_back(i < (int32_t)value->i_attrs.size() ? value->i_attrs[i] : GLOBAL_FLAG(DefaultIntAttribute));
    }
    for (int32_t i = 0; i < info_.f_num; ++i) {
      f_attrs_.push_back(i < (int32_t)value->f_attrs.size() ? value->f_attrs[i] : GLOBAL_FLAG(DefaultFloatAttribute));
    }
    for (int32_t i = 0; i < info_.s_num; ++i) {
      s_attrs_.push_back(i < (int32_t)value->s_attrs.size() ? value->s_attrs[i] : GLOBAL_FLAG(DefaultStringAttribute));
    }
  }
}

// The reference compares `edge_id < Size()` with an unsigned right-hand side, so a
// negative id is "out of range" too (memory_edge_storage.cc:90-125).
#define GLX_EDGE_IN_RANGE(vec, per) ((uint64_t)edge_id < (uint64_t)((vec).size() / (per)))
int64_t Graph::GetSrcId(int64_t edge_id) const { return GLX_EDGE_IN_RANGE(src_, 1) ? src_[edge_id] : -1; }
int64_t Graph::GetDstId(int64_t edge_id) const { return GLX_EDGE_IN_RANGE(dst_, 1) ? dst_[edge_id] : -1; }
float Graph::GetEdgeWeight(int64_t edge_id) const {
  return GLX_EDGE_IN_RANGE(weight_, 1) ? weight_[edge_id] : GLOBAL_FLAG(DefaultWeight);
}
int32_t Graph::GetEdgeLabel(int64_t edge_id) const {
  return GLX_EDGE_IN_RANGE(label_, 1) ? label_[edge_id] : (int32_t)GLOBAL_FLAG(DefaultLabel);
}
int64_t Graph::GetEdgeTimestamp(int64_t edge_id) const {
  return GLX_EDGE_IN_RANGE(timestamp_, 1) ? timestamp_[edge_id] : GLOBAL_FLAG(DefaultTimestamp);
}
const int64_t* Graph::GetEdgeIntAttrs(int64_t edge_id) const {
  return info_.i_num > 0 && GLX_EDGE_IN_RANGE(i_attrs_, info_.i_num) ? i_attrs_.data() + edge_id * info_.i_num : nullptr;
}
const float* Graph::GetEdgeFloatAttrs(int64_t edge_id) const {
  return info_.f_num > 0 && GLX_EDGE_IN_RANGE(f_attrs_, info_.f_num) ? f_attrs_.data() + edge_id * info_.f_num : nullptr;
}
const std::string* Graph::GetEdgeStringAttrs(int64_t edge_id) const {
  return info_.s_num > 0 && GLX_EDGE_IN_RANGE(s_attrs_, info_.s_num) ? s_attrs_.data() + edge_id * info_.s_num : nullptr;
}
#undef GLX_EDGE_IN_RANGE

Status Graph::AppendColumns(const io::SideInfo& info, io::EdgeColumns* c) {
  std::lock_guard<std::mutex> g(mtx_);
  SetSideInfo(&info);
  const size_t n = c->src.size();
  if (c->dst.size() != n || (info_.IsWeighted() && c->weight.size() != n) || (info_.IsLabeled() && c->label.size() != n) ||
      (info_.IsTimestamped() && c->timestamp.size() != n) || c->i_attrs.size() != n * (size_t)info_.i_num ||
      c->f_attrs.size() != n * (size_t)info_.f_num || c->s_attrs.size() != n * (size_t)info_.s_num) {
    return error::InvalidArgument("edge batch does not match the side info of type '" + type_ + "'");
  }
  src_.insert(src_.end(), c->src.begin(), c->src.end());
  dst_.insert(dst_.end(), c->dst.begin(), c->dst.end());
  if (info_.IsWeighted()) weight_.insert(weight_.end(), c->weight.begin(), c->weight.end());
  if (info_.IsLabeled()) label_.insert(label_.end(), c->label.begin(), c->label.end());
  if (info_.IsTimestamped()) timestamp_.insert(timestamp_.end(), c->timestamp.begin(), c->timestamp.end());
  i_attrs_.insert(i_attrs_.end(), c->i_attrs.begin(), c->i_attrs.end());
  f_attrs_.insert(f_attrs_.end(), c->f_attrs.begin(), c->f_attrs.end());
  s_attrs_.insert(s_attrs_.end(), std::make_move_iterator(c->s_attrs.begin()), std::make_move_iterator(c->s_attrs.end()));
  return Status::OK();
}

Status Graph::UpdateEdges(const UpdateEdgesRequest* req, UpdateEdgesResponse*) {
  std::lock_guard<std::mutex> g(mtx_);
  // the device CSR is built once from everything staged (graph_store.h): records that arrive later would be held
  // on the host and never sampled, so they are refused instead
  if (dev_) return error::InvalidArgument("edge type '" + type_ + "' is already built: UpdateEdges must precede Build()");
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
  const bool sorted = option.name == "sort";
  // Build() orders timestamped types by timestamp, else weighted types by weight
  // (memory_adj_matrix.cc:60-66)
  const int order = !sorted ? GLX_ORDER_INSERTION
                            : info_.IsTimestamped() ? GLX_ORDER_TIMESTAMP_ASC
                                                    : weighted ? GLX_ORDER_WEIGHT_DESC : GLX_ORDER_INSERTION;
  int rc = glx_graph_build_ordered(GLOBAL_FLAG(DeviceId), (int64_t)src_.size(), src_.data(), dst_.data(),
                                   weighted ? weight_.data() : nullptr, nullptr,
                                   info_.IsTimestamped() ? timestamp_.data() : nullptr, order, GLX_PTR_HOST, nullptr,
                                   &dev_);
  return error::FromGlx(rc);
}

Status Graph::EnsureInDegree() {
  // The reference keeps in/out-degree statistics by default (StorageMode bit 1,
  // config.cc:93, topo_statics.cc); InDegreeSampler needs them as per-row alias tables,
  // built on the device the first time that sampler is used on this edge type.
  std::lock_guard<std::mutex> g(mtx_);
  if (in_degree_ready_) return Status::OK();
  if (!dev_) return error::InvalidArgument("edge type '" + type_ + "' is not built on the device");
  int rc = glx_graph_enable_in_degree(dev_, nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  in_degree_ready_ = true;
  return Status::OK();
}

Status Graph::EnsureDefaultWeights() {
  // EdgeWeightSampler on an unweighted type: every edge weighs GLOBAL_FLAG(DefaultWeight), as
  // MemoryEdgeStorage::GetWeight answers for it (memory_edge_storage.cc:97-103); see glx_graph_enable_default_weight.
  std::lock_guard<std::mutex> g(mtx_);
  if (info_.IsWeighted() || default_weights_ready_) return Status::OK();
  if (!dev_) return error::InvalidArgument("edge type '" + type_ + "' is not built on the device");
  int rc = glx_graph_enable_default_weight(dev_, GLOBAL_FLAG(DefaultWeight), nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  default_weights_ready_ = true;
  return Status::OK();
}

Status Graph::EnsureGlobalInDegree(glx_dist_store* store) {
  // A shard of a partitioned edge type: InDegreeSampler's weights are in-degrees over ALL shards
  // (glx_dist_enable_in_degree: collective, every server gets here in the same Run()).
  std::lock_guard<std::mutex> g(mtx_);
  if (global_in_degree_ready_) return Status::OK();
  if (!dev_) return error::InvalidArgument("edge type '" + type_ + "' is not built on the device");
  int rc = glx_dist_enable_in_degree(store, dev_, nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  global_in_degree_ready_ = true;
  in_degree_ready_ = true;
  return Status::OK();
}

Status Graph::EnsureIdIndex() {
  // Per-row id-sorted index (shared with strict negative sampling): an id == value filter then finds its
  // hits with a binary search instead of scanning the row.  Built the first time such a request arrives.
  std::lock_guard<std::mutex> g(mtx_);
  if (neg_strict_ready_) return Status::OK();
  if (!dev_) return error::InvalidArgument("edge type '" + type_ + "' is not built on the device");
  int rc = glx_graph_enable_negative(dev_, nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  neg_strict_ready_ = true;
  return Status::OK();
}

// ------------------------------------------------------------------ Noder --
Noder::Noder(const std::string& type) : type_(type), dev_(nullptr), neg_(nullptr) {}

Noder::~Noder() {
  glx_negative_destroy(neg_);
  glx_features_destroy(dev_);
}

Status Noder::Negative(const glx_negative** out) {
  std::lock_guard<std::mutex> g(mtx_);
  *out = nullptr;
  if (!neg_) {
    if (!info_.IsWeighted()) return error::InvalidArgument("node type '" + type_ + "' has no weights");
    int rc = glx_negative_create(GLOBAL_FLAG(DeviceId), (int64_t)ids_.size(), ids_.data(), weights_.data(),
                                 GLX_PTR_HOST, nullptr, &neg_);
    if (rc != GLX_OK) return error::FromGlx(rc);
  }
  *out = neg_;
  return Status::OK();
}

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
  if (info_.IsWeighted()) weights_.push_back(value->weight);
  if (info_.IsLabeled()) labels_.push_back(value->label);
  if (info_.IsTimestamped()) timestamps_.push_back(value->timestamp);
  for (int32_t i = 0; i < info_.i_num; ++i) {
    i_attrs_.push_back(i < (int32_t)value->i_attrs.size() ? value->i_attrs[i] : GLOBAL_FLAG(DefaultIntAttribute));
  }
  for (int32_t i = 0; i < info_.s_num; ++i) {
    s_attrs_.push_back(i < (int32_t)value->s_attrs.size() ? value->s_attrs[i] : GLOBAL_FLAG(DefaultStringAttribute));
  }
}

int32_t Noder::RowOf(int64_t node_id) const {
  auto it = index_.find(node_id);
  return it == index_.end() ? -1 : it->second;
}
float Noder::GetWeight(int64_t node_id) const {  // memory_node_storage.cc:88-99
  if (!info_.IsWeighted()) return 0.0f;
  const int32_t r = RowOf(node_id);
  return r < 0 ? GLOBAL_FLAG(DefaultWeight) : weights_[r];
}
int32_t Noder::GetLabel(int64_t node_id) const {  // memory_node_storage.cc:101-112
  if (!info_.IsLabeled()) return -1;
  const int32_t r = RowOf(node_id);
  return r < 0 ? (int32_t)GLOBAL_FLAG(DefaultLabel) : labels_[r];
}
int64_t Noder::GetTimestamp(int64_t node_id) const {
  if (!info_.IsTimestamped()) return -1;
  const int32_t r = RowOf(node_id);
  return r < 0 ? GLOBAL_FLAG(DefaultTimestamp) : timestamps_[r];
}

Status Noder::AppendColumns(const io::SideInfo& info, io::NodeColumns* c) {
  std::lock_guard<std::mutex> g(mtx_);
  SetSideInfo(&info);
  const size_t n = c->id.size();
  if ((info_.IsWeighted() && c->weight.size() != n) || (info_.IsLabeled() && c->label.size() != n) ||
      (info_.IsTimestamped() && c->timestamp.size() != n) || c->i_attrs.size() != n * (size_t)info_.i_num ||
      c->f_attrs.size() != n * (size_t)info_.f_num || c->s_attrs.size() != n * (size_t)info_.s_num) {
    return error::InvalidArgument("node batch does not match the side info of type '" + type_ + "'");
  }
  index_.reserve(index_.size() + n);
  ids_.reserve(ids_.size() + n);
  feats_.reserve(feats_.size() + n * (size_t)info_.f_num);
  for (size_t r = 0; r < n; ++r) {
    if (!index_.emplace(c->id[r], (int32_t)ids_.size()).second) continue;  // duplicate id: ignore
    ids_.push_back(c->id[r]);
    if (info_.IsWeighted()) weights_.push_back(c->weight[r]);
    if (info_.IsLabeled()) labels_.push_back(c->label[r]);
    if (info_.IsTimestamped()) timestamps_.push_back(c->timestamp[r]);
    feats_.insert(feats_.end(), c->f_attrs.begin() + r * info_.f_num, c->f_attrs.begin() + (r + 1) * info_.f_num);
    i_attrs_.insert(i_attrs_.end(), c->i_attrs.begin() + r * info_.i_num, c->i_attrs.begin() + (r + 1) * info_.i_num);
    for (int32_t j = 0; j < info_.s_num; ++j) s_attrs_.push_back(std::move(c->s_attrs[r * info_.s_num + j]));
  }
  return Status::OK();
}

Status Noder::UpdateNodes(const UpdateNodesRequest* req, UpdateNodesResponse*) {
  std::lock_guard<std::mutex> g(mtx_);
  if (dev_) return error::InvalidArgument("node type '" + type_ + "' is already built: UpdateNodes must precede Build()");
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
namespace {
std::atomic<uint64_t> g_next_store_uid{1};
}

GraphStore::GraphStore() : uid_(g_next_store_uid.fetch_add(1)) {}

void GraphStore::SetShard(int32_t index, int32_t count) {
  shard_count_ = count < 1 ? 1 : count;
  shard_index_ = (index < 0 || index >= shard_count_) ? 0 : index;
}

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

Status GraphStore::Build(const IndexOption& option) {
  std::vector<Graph*> gs;
  std::vector<Noder*> ns;
  {
    std::lock_guard<std::mutex> g(mtx_);
    for (auto& it : graphs_) gs.push_back(it.second);
    for (auto& it : noders_) ns.push_back(it.second);
  }
  for (Graph* g : gs) {
    Status s = g->Build(option);
    if (!s.ok()) return s;
  }
  for (Noder* n : ns) {
    Status s = n->Build(option);
    if (!s.ok()) return s;
  }
  {
    // what GetStats reports describes the data as built: statistics gathered before this Build (a GetStats issued
    // while sources were still loading) are stale now -- the next GetStats gathers them again
    std::lock_guard<std::mutex> g(mtx_);
    stats_ = GraphStatistics();
  }
  return Status::OK();
}

std::unordered_map<std::string, int64_t> GraphStore::NodeCounts() {
  std::lock_guard<std::mutex> g(mtx_);
  std::unordered_map<std::string, int64_t> out;
  for (auto& it : noders_) out[it.first] = it.second->GetNodeCount();
  return out;
}

std::unordered_map<std::string, int64_t> GraphStore::EdgeCounts() {
  std::lock_guard<std::mutex> g(mtx_);
  std::unordered_map<std::string, int64_t> out;
  for (auto& it : graphs_) out[it.first] = it.second->GetEdgeCount();
  return out;
}

void GraphStore::DeclareEdgeType(const std::string& edge_type) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = e_types_.find(edge_type);
  if (it == e_types_.end()) e_types_.insert({edge_type, 1});
  else it->second = 2;  // graph_store.cc:196-201: undirected homogeneous edges
}

void GraphStore::DeclareNodeType(const std::string& node_type) {
  std::lock_guard<std::mutex> g(mtx_);
  n_types_.insert({node_type, 1});
}

std::vector<int32_t> GraphStore::GetLocalCount() {
  std::lock_guard<std::mutex> g(mtx_);
  // types that were never declared by a source (a store filled through GetGraph() / GetNoder()) count once
  for (auto& it : graphs_) e_types_.insert({it.first, 1});
  for (auto& it : noders_) n_types_.insert({it.first, 1});
  std::vector<int32_t> out;
  out.reserve(e_types_.size() + n_types_.size());
  for (auto& it : e_types_) {
    auto f = graphs_.find(it.first);
    out.push_back((int32_t)((f == graphs_.end() ? 0 : f->second->GetEdgeCount()) * it.second));
  }
  for (auto& it : n_types_) {
    auto f = noders_.find(it.first);
    out.push_back((int32_t)((f == noders_.end() ? 0 : f->second->GetNodeCount()) * it.second));
  }
  return out;
}

void GraphStore::SetCountGatherer(CountGatherer gather) {
  std::lock_guard<std::mutex> g(mtx_);
  gather_ = std::move(gather);
}

Status GraphStore::BuildStatistics() {
  const std::vector<int32_t> local = GetLocalCount();
  std::vector<std::vector<int32_t>> all;
  CountGatherer gather;
  {
    std::lock_guard<std::mutex> g(mtx_);
    gather = gather_;
  }
  if (gather) {
    Status s = gather(local, &all);
    if (!s.ok()) return s;
  } else {
    all.push_back(local);
  }
  std::lock_guard<std::mutex> g(mtx_);
  stats_ = GraphStatistics();
  for (const auto& counts : all) {  // FillCounts, graph_store.cc:295-303
    if (counts.size() != e_types_.size() + n_types_.size()) {
      return error::Internal("GetStats: a server reported " + std::to_string(counts.size()) + " counts, " +
                             std::to_string(e_types_.size() + n_types_.size()) + " types are declared here");
    }
    size_t j = 0;
    for (auto& it : e_types_) stats_.AppendCount(it.first, counts[j++]);
    for (auto& it : n_types_) stats_.AppendCount(it.first, counts[j++]);
  }
  return Status::OK();
}

Noder* GraphStore::GetNoder(const std::string& node_type) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = noders_.find(node_type);
  if (it == noders_.end()) it = noders_.emplace(node_type, new Noder(node_type)).first;
  return it->second;
}

}  // namespace graphlearn
