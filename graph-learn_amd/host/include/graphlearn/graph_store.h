// GraphStore: the heterogeneous type -> storage map of the reference
// (graphlearn/src/core/graph/graph_store.h:32-66, heter_dispatcher.h:44-56), with
// device-resident storages.  Edges / nodes are staged on the host exactly like
// LocalGraph::UpdateEdges -> storage->Add (core/graph/local_graph.cc:50-64);
// Build() hands the raw edge list to the GPU (glx_graph_build), which orders rows
// the way MemoryAdjMatrix::Build does (memory_adj_matrix.cc:60-66,105-125: weight
// descending for weighted types), converts to CSR (memory_adj_matrix.cc:169-189)
// and builds the alias tables and the id map there; node features go through
// glx_features_create.  After Build() the storage is immutable and served from HBM.
#ifndef GLX_HOST_GRAPH_STORE_H_
#define GLX_HOST_GRAPH_STORE_H_
#include <cstdint>
#include <mutex>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "graphlearn/op_request.h"
#include "graphlearn/status.h"

struct glx_graph;
struct glx_features;
struct glx_negative;
struct glx_dist_store;

namespace graphlearn {
namespace io {

enum DataFormat { kDefault = 1, kWeighted = 2, kLabeled = 4, kTimestamped = 8, kAttributed = 16 };

// graphlearn/src/core/io/element_value.h:30-70 (fields used on this path).
struct SideInfo {
  int32_t i_num = 0, f_num = 0, s_num = 0;
  int32_t format = 0;
  std::string type, src_type, dst_type;
  bool IsInitialized() const { return format != 0; }
  bool IsWeighted() const { return format & kWeighted; }
  bool IsLabeled() const { return format & kLabeled; }
  bool IsTimestamped() const { return format & kTimestamped; }
  bool IsAttributed() const { return format & kAttributed; }
};

struct EdgeValue {  // element_value.h:104-116
  int64_t src_id = 0, dst_id = 0;
  float weight = 0.f;
  int32_t label = 0;
  int64_t timestamp = 0;
  std::vector<int64_t> i_attrs;  // AttributeValue (element_value.h:72-102), by kind
  std::vector<float> f_attrs;
  std::vector<std::string> s_attrs;
};

struct NodeValue {  // element_value.h:118-132; attrs = the float attributes
  int64_t id = 0;
  float weight = 0.f;
  int32_t label = 0;
  int64_t timestamp = 0;
  std::vector<float> attrs;
  std::vector<int64_t> i_attrs;
  std::vector<std::string> s_attrs;
};

// A batch of records in columnar form (what the file loader produces per chunk); the
// attribute columns hold i_num / f_num / s_num values per record, row-major.
struct EdgeColumns {
  std::vector<int64_t> src, dst, timestamp;
  std::vector<float> weight;
  std::vector<int32_t> label;
  std::vector<int64_t> i_attrs;
  std::vector<float> f_attrs;
  std::vector<std::string> s_attrs;
};
struct NodeColumns {
  std::vector<int64_t> id, timestamp;
  std::vector<float> weight;
  std::vector<int32_t> label;
  std::vector<int64_t> i_attrs;
  std::vector<float> f_attrs;
  std::vector<std::string> s_attrs;
};

}  // namespace io

struct IndexOption {  // include/index_option.h:24-48
  std::string name;  // "sort": Build() orders every adjacency row (timestamp, else weight)
  // the KNN index parameters: carried, never read (no KNN operator here)
  std::string index_type;
  int32_t dimension = 0, nlist = 0, nprobe = 0, m = 0;
};

// "UpdateEdges" / "UpdateNodes" (include/graph_request.h:36-124, service/request/graph_update_request.cc): a batch of
// records for ONE edge / node type, routed by source id / node id (shard keys kSrcIds / kNodeIds,
// graph_update_request.cc:151,234) -- what the reference's loader threads send to the servers that own the records
// (core/graph/graph_store.cc:210-250).  Records are held as values (the reference packs them into tensors for the
// wire: shards here exchange device buffers, not messages); Append / Size / Next keep the reference's cursor surface.
class UpdateEdgesRequest : public OpRequest {
public:
  UpdateEdgesRequest();
  UpdateEdgesRequest(const io::SideInfo* info, int32_t batch_size);
  std::string Name() const override { return "UpdateEdges"; }
  OpRequest* Clone() const override;
  int32_t Size() const { return (int32_t)values_.size(); }
  void Append(const io::EdgeValue* value);
  bool Next(io::EdgeValue* value);  // graph_update_request.cc:205-231
  const io::SideInfo& GetSideInfo() const { return info_; }
  const std::vector<io::EdgeValue>& Values() const { return values_; }
private:
  io::SideInfo info_;
  std::vector<io::EdgeValue> values_;
  int32_t cursor_ = 0;
};
class UpdateEdgesResponse : public OpResponse {
public:
  OpResponse* New() const override { return new UpdateEdgesResponse; }
};
class UpdateNodesRequest : public OpRequest {
public:
  UpdateNodesRequest();
  UpdateNodesRequest(const io::SideInfo* info, int32_t batch_size);
  std::string Name() const override { return "UpdateNodes"; }
  OpRequest* Clone() const override;
  int32_t Size() const { return (int32_t)values_.size(); }
  void Append(const io::NodeValue* value);
  bool Next(io::NodeValue* value);
  const io::SideInfo& GetSideInfo() const { return info_; }
  const std::vector<io::NodeValue>& Values() const { return values_; }
private:
  io::SideInfo info_;
  std::vector<io::NodeValue> values_;
  int32_t cursor_ = 0;
};
class UpdateNodesResponse : public OpResponse {
public:
  OpResponse* New() const override { return new UpdateNodesResponse; }
};

// One edge type.  Host staging + device CSR.
class Graph {
public:
  explicit Graph(const std::string& type);
  ~Graph();
  Status UpdateEdges(const UpdateEdgesRequest* req, UpdateEdgesResponse* res);
  void SetSideInfo(const io::SideInfo* info);
  const io::SideInfo* GetSideInfo() const { return &info_; }
  void Add(const io::EdgeValue* value);          // edge id = insertion index
  // Bulk form of Add for a parsed chunk (consumes the strings); same edge-id rule.
  Status AppendColumns(const io::SideInfo& info, io::EdgeColumns* columns);
  Status Build(const IndexOption& option);       // sort (if option.name=="sort") + upload
  int64_t GetEdgeCount() const { return (int64_t)src_.size(); }
  // All edges in edge-id order (EdgeStorage::GetSrcIds/GetDstIds, memory_edge_storage.cc:127-133).
  const std::vector<int64_t>& SrcIds() const { return src_; }
  const std::vector<int64_t>& DstIds() const { return dst_; }
  const glx_graph* Device() const { return dev_; }  // nullptr before Build()
  // Candidate list of the negative samplers for this edge type (destination ids in
  // first-appearance order, uniform or in-degree weighted), built on first use and
  // cached like AliasMethodFactory::LookupOrCreate does; `strict` also prepares the
  // per-row sorted neighbour lists the exclusion test searches.
  Status Negative(bool by_in_degree, bool strict, const glx_negative** out);
  // In-degree alias tables for InDegreeSampler, built on first use.
  Status EnsureInDegree();
  // Constant-weight alias tables for EdgeWeightSampler on an unweighted type, built on first use.
  Status EnsureDefaultWeights();
  // The same for a shard of a partitioned edge type: in-degrees summed over all shards (collective).
  Status EnsureGlobalInDegree(glx_dist_store* store);
  // Per-row id-sorted index for id == value filters (and strict negative sampling), built on first use.
  Status EnsureIdIndex();

  // Per-edge properties by edge id, host resident (they are not read by the samplers):
  // EdgeStorage::GetWeight/GetLabel/GetTimestamp/GetAttribute
  // (memory_edge_storage.cc:90-125).  An id outside [0, E) -- e.g. the -1 of a
  // default-filled sample -- yields the DefaultWeight/DefaultLabel/... flags.
  int64_t GetSrcId(int64_t edge_id) const;
  int64_t GetDstId(int64_t edge_id) const;
  float GetEdgeWeight(int64_t edge_id) const;
  int32_t GetEdgeLabel(int64_t edge_id) const;
  int64_t GetEdgeTimestamp(int64_t edge_id) const;
  const int64_t* GetEdgeIntAttrs(int64_t edge_id) const;        // i_num values or nullptr
  const float* GetEdgeFloatAttrs(int64_t edge_id) const;        // f_num values or nullptr
  const std::string* GetEdgeStringAttrs(int64_t edge_id) const;  // s_num values or nullptr

private:
  std::string type_;
  io::SideInfo info_;
  std::vector<int64_t> src_, dst_;
  std::vector<float> weight_;
  std::vector<int32_t> label_;
  std::vector<int64_t> timestamp_;
  std::vector<int64_t> i_attrs_;
  std::vector<float> f_attrs_;
  std::vector<std::string> s_attrs_;
  glx_graph* dev_;
  glx_negative* neg_uniform_;
  glx_negative* neg_in_degree_;
  bool neg_strict_ready_;
  bool in_degree_ready_;
  bool default_weights_ready_ = false;
  bool global_in_degree_ready_;
  std::mutex mtx_;
};

// One node type.  Host staging + device feature matrix.
class Noder {
public:
  explicit Noder(const std::string& type);
  ~Noder();
  Status UpdateNodes(const UpdateNodesRequest* req, UpdateNodesResponse* res);
  void SetSideInfo(const io::SideInfo* info);
  const io::SideInfo* GetSideInfo() const { return &info_; }
  void Add(const io::NodeValue* value);          // duplicate ids are ignored (node_storage.h:41)
  Status AppendColumns(const io::SideInfo& info, io::NodeColumns* columns);
  Status Build(const IndexOption& option);
  const glx_features* Device() const { return dev_; }
  int64_t GetNodeCount() const { return (int64_t)ids_.size(); }
  const std::vector<int64_t>& Ids() const { return ids_; }  // NodeStorage::GetIds, insertion order
  const std::vector<float>& Weights() const { return weights_; }  // NodeStorage::GetWeights (weighted types)
  // Candidate list of NodeWeightNegativeSampler: this type's ids weighted by node weight.
  Status Negative(const glx_negative** out);

  // NodeStorage::GetWeight/GetLabel/GetTimestamp/GetAttribute
  // (memory_node_storage.cc:88-138) for the host-resident properties; -1 = unknown id.
  int32_t RowOf(int64_t node_id) const;
  float GetWeight(int64_t node_id) const;
  int32_t GetLabel(int64_t node_id) const;
  int64_t GetTimestamp(int64_t node_id) const;
  float WeightAt(int32_t row) const { return weights_[(size_t)row]; }        // by row (RowOf >= 0), for bulk lookups
  int32_t LabelAt(int32_t row) const { return labels_[(size_t)row]; }
  int64_t TimestampAt(int32_t row) const { return timestamps_[(size_t)row]; }
  const int64_t* GetIntAttrs(int32_t row) const { return i_attrs_.data() + (int64_t)row * info_.i_num; }
  const float* GetFloatAttrs(int32_t row) const { return feats_.data() + (int64_t)row * info_.f_num; }
  const std::string* GetStringAttrs(int32_t row) const { return s_attrs_.data() + (int64_t)row * info_.s_num; }

private:
  std::string type_;
  io::SideInfo info_;
  std::vector<int64_t> ids_;
  std::vector<float> feats_;
  std::vector<float> weights_;
  std::vector<int32_t> labels_;
  std::vector<int64_t> timestamps_;
  std::vector<int64_t> i_attrs_;
  std::vector<std::string> s_attrs_;
  std::unordered_map<int64_t, int32_t> index_;
  glx_features* dev_;
  glx_negative* neg_;
  std::mutex mtx_;
};

// include/graph_statistics.h:26-39: type -> one count per server, edge types and node types alike.
using Counts = std::unordered_map<std::string, std::vector<int32_t>>;
class GraphStatistics {
public:
  const Counts& GetCounts() const { return counts_; }
  void AppendCount(const std::string& type, int32_t count) { counts_[type].push_back(count); }

private:
  Counts counts_;
};

class GraphStore {
public:
  GraphStore();
  ~GraphStore();
  Graph* GetGraph(const std::string& edge_type);
  Noder* GetNoder(const std::string& node_type);
  // Distinguishes stores over the process lifetime (an address can be reused): operators that
  // keep per-store state (traversal cursors) key it by this.
  uint64_t Uid() const { return uid_; }
  // type -> number of nodes / edges held (GetStats: core/operator/graph/stats_getter.cc)
  std::unordered_map<std::string, int64_t> NodeCounts();
  std::unordered_map<std::string, int64_t> EdgeCounts();
  // GraphStore::Init's bookkeeping (graph_store.cc:185-208): an edge type declared by two sources (the
  // reverse pass of an undirected homogeneous edge file) counts double; Load() declares every source.  A
  // store filled directly through GetGraph() / GetNoder() counts every type it holds once.
  void DeclareEdgeType(const std::string& edge_type);
  void DeclareNodeType(const std::string& node_type);
  // Edges / nodes held here, one entry per declared edge type then per declared node type, each group in
  // type-name order (BuildLocalCount, graph_store.cc:305-317); what "GetCount" answers with.
  std::vector<int32_t> GetLocalCount();
  // "GetStats": the local counts of every server, per type (BuildStatistics / FillCounts, graph_store.cc:
  // 278-303).  One server: this store's counts.  Several: the gatherer an Env installs collects every
  // server's GetLocalCount() (COLLECTIVE: every server asks for the statistics at the same time).
  typedef std::function<Status(const std::vector<int32_t>& local, std::vector<std::vector<int32_t>>* all)> CountGatherer;
  void SetCountGatherer(CountGatherer gather);
  Status BuildStatistics();
  const GraphStatistics& GetStatistics() const { return stats_; }
  // Build every storage added so far (GraphStore::Build, graph_store.cc:252-276).
  Status Build(const IndexOption& option);
  // Load every source, then Build (GraphStore::Load, graph_store.cc:60-120): declared in
  // data_source.h's terms, defined in loader.cc.
  template <class EdgeSources, class NodeSources>
  Status Load(const EdgeSources& edges, const NodeSources& nodes);
  // One process per GPU, every process reads the same sources: shard `index` of `count` keeps
  // the edges whose source id -- and the nodes whose id -- hash to it, llabs(id) % count, the
  // rule HashPartitioner routes requests by (hash_partitioner.h:88-90); this is where the
  // reference's servers end up after Initializer has exchanged the records they read.  Edge
  // ids are positions in the shard's own load order, i.e. server-local like the reference's.
  // Set before loading; the default (0, 1) keeps everything.
  void SetShard(int32_t index, int32_t count);
  int32_t ShardIndex() const { return shard_index_; }
  int32_t ShardCount() const { return shard_count_; }
  bool Owns(int64_t id) const {
    return shard_count_ <= 1 || (int32_t)((id < 0 ? -(uint64_t)id : (uint64_t)id) % (uint64_t)shard_count_) == shard_index_;
  }

private:
  int32_t shard_index_ = 0;
  int32_t shard_count_ = 1;
  uint64_t uid_;
  std::mutex mtx_;
  std::unordered_map<std::string, Graph*> graphs_;
  std::unordered_map<std::string, Noder*> noders_;
  std::map<std::string, int32_t> e_types_, n_types_;  // declared types -> multiplier, in name order
  CountGatherer gather_;
  GraphStatistics stats_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_GRAPH_STORE_H_
