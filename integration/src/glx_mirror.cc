// export every edge type's post-Build() adjacency and every node type's float attributes
// through the reference's EXISTING storage interfaces (graph_storage.h:40-80, node_storage.h:38-78) and hand them to
// the C-ABI.  Compiled against the reference's headers (integration/Makefile).
#include "glx_mirror.h"

#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "core/graph/storage/graph_storage.h"
#include "core/graph/storage/node_storage.h"
#include "include/config.h"

namespace graphlearn {
namespace op {
namespace {

int Device() {
  const char* e = getenv("GLX_DEVICE");
  return e ? atoi(e) : 0;
}

// [glx-mirror-edge-type]
int MirrorEdgeType(Graph* graph, int device, glx_graph** out) {
  io::GraphStorage* st = graph->GetLocalStorage();          // graph_storage.h:40-57
  const io::IdArray srcs = st->GetAllSrcIds();              // distinct source ids = rows
  std::vector<int64_t> row_ptr{0}, col, eid, ids;
  std::vector<float> w;
  std::vector<int64_t> ts;
  const bool weighted = st->GetSideInfo()->IsWeighted();
  const bool timestamped = st->GetSideInfo()->IsTimestamped();
  for (int32_t r = 0; r < srcs.Size(); ++r) {
    auto nbrs = st->GetNeighbors(srcs[r]);                  // post-Build order: part of every sampler's result
    auto eds = st->GetOutEdges(srcs[r]);
    for (int32_t j = 0; j < nbrs.Size(); ++j) {
      col.push_back(nbrs[j]);
      eid.push_back(eds[j]);
      if (weighted) w.push_back(st->GetEdgeWeight(eds[j]));  // weights are per edge id in the reference
      if (timestamped) ts.push_back(st->GetEdgeTimestamp(eds[j]));
    }
    ids.push_back(srcs[r]);
    row_ptr.push_back(static_cast<int64_t>(col.size()));
  }
  glx_graph* g = nullptr;
  int rc = glx_graph_create(device, static_cast<int64_t>(ids.size()), static_cast<int64_t>(col.size()), row_ptr.data(),
                            col.data(), eid.data(), weighted ? w.data() : nullptr, ids.data(), GLX_PTR_HOST, nullptr, &g);
  if (rc == GLX_OK && timestamped) rc = glx_graph_set_timestamps(g, ts.data(), GLX_PTR_HOST, nullptr);  // for filters
  if (rc != GLX_OK && g != nullptr) {
    glx_graph_destroy(g);
    g = nullptr;
  }
  *out = g;
  return rc;                                                  // not GLX_OK: glx_last_error() has the reason
}
// [/glx-mirror-edge-type]

// [glx-mirror-node-type]
int MirrorNodeType(Noder* noder, int device, glx_features** out) {
  io::NodeStorage* st = noder->GetLocalStorage();           // node_storage.h:38-78
  const io::IdArray ids = st->GetIds();
  const int32_t dim = st->GetSideInfo()->f_num;              // element_value.h:30-34
  *out = nullptr;
  if (dim <= 0) return GLX_INVALID_ARGUMENT;                  // no float attributes: nothing to aggregate
  std::vector<float> X(static_cast<size_t>(ids.Size()) * dim);
  std::vector<int64_t> raw(ids.Size());
  for (int32_t r = 0; r < ids.Size(); ++r) {
    raw[r] = ids[r];
    io::Attribute attr = st->GetAttribute(ids[r]);
    int32_t len = 0;
    const float* f = attr->GetFloats(&len);
    for (int32_t c = 0; c < dim; ++c) X[static_cast<size_t>(r) * dim + c] = c < len ? f[c] : GLOBAL_FLAG(DefaultFloatAttribute);
  }
  return glx_features_create(device, ids.Size(), dim, X.data(), raw.data(), GLX_PTR_HOST, nullptr, out);
}
// [/glx-mirror-node-type]

struct Mirror {
  std::mutex mu;
  struct EdgeEntry { glx_graph* g; int64_t edges; };
  struct NodeEntry { glx_features* f; int64_t nodes; };
  std::unordered_map<std::string, EdgeEntry> graphs;
  std::unordered_map<std::string, NodeEntry> feats;
};

std::mutex g_mu;
std::unordered_map<GraphStore*, Mirror*>& Mirrors() {
  static std::unordered_map<GraphStore*, Mirror*> m;
  return m;
}

Mirror* MirrorOf(GraphStore* store) {
  std::lock_guard<std::mutex> lock(g_mu);
  Mirror*& m = Mirrors()[store];
  if (m == nullptr) m = new Mirror;
  return m;
}

}  // namespace

int GlxGraphOf(GraphStore* store, const std::string& edge_type, const glx_graph** out) {
  Mirror* m = MirrorOf(store);
  std::lock_guard<std::mutex> lock(m->mu);
  Graph* graph = store->GetGraph(edge_type);
  const int64_t edges = graph->GetLocalStorage()->GetEdgeCount();
  auto it = m->graphs.find(edge_type);
  if (it != m->graphs.end() && it->second.edges == edges) {
    *out = it->second.g;
    return GLX_OK;
  }
  if (it != m->graphs.end()) {  // the storage grew since: mirror it again
    glx_graph_destroy(it->second.g);
    m->graphs.erase(it);
  }
  glx_graph* g = nullptr;
  const int rc = MirrorEdgeType(graph, Device(), &g);
  if (rc == GLX_OK) m->graphs[edge_type] = {g, edges};
  *out = g;
  return rc;
}

int GlxFeaturesOf(GraphStore* store, const std::string& node_type, const glx_features** out) {
  Mirror* m = MirrorOf(store);
  std::lock_guard<std::mutex> lock(m->mu);
  Noder* noder = store->GetNoder(node_type);
  const int64_t nodes = noder->GetLocalStorage()->Size();
  auto it = m->feats.find(node_type);
  if (it != m->feats.end() && it->second.nodes == nodes) {
    *out = it->second.f;
    return GLX_OK;
  }
  if (it != m->feats.end()) {
    glx_features_destroy(it->second.f);
    m->feats.erase(it);
  }
  glx_features* f = nullptr;
  const int rc = MirrorNodeType(noder, Device(), &f);
  if (rc == GLX_OK) m->feats[node_type] = {f, nodes};
  *out = f;
  return rc;
}

Status GlxStatus(int rc) {
  if (rc == GLX_OK) return Status::OK();
  return Status(static_cast<error::Code>(rc), glx_last_error());
}

}  // namespace op
}  // namespace graphlearn
