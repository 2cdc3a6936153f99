// Data sources of a graph: what Graph.node()/Graph.edge() hand to the engine
// (graphlearn/src/include/data_source.h:30-190, python/graph.py:928-977) and the
// local-file TSV loader that reads them (core/io/{node,edge}_loader.cc,
// core/io/parser.cc, common/io/value.h).  SURVEY 8(f) rank 4: on-disk format
// parity, host-side parsing, device-side build.
//
// File format (platform/local file system): first line = tab separated
// `name:type` columns, every other line = one record.
//   nodes: id:int64 [weight:float] [label:int64] [timestamp:int64] [attrs:string]
//   edges: src_id:int64 dst_id:int64 [weight:float] [label:int64] [timestamp:int64] [attrs:string]
// The optional columns must appear in exactly this order and must match the
// decoder's format bits (edge_loader.cc:110-141).  The attribute column packs all
// attributes into one string split by AttributeInfo::delimiter (parser.cc:39-104).
#ifndef GLX_HOST_DATA_SOURCE_H_
#define GLX_HOST_DATA_SOURCE_H_
#include <cstdint>
#include <string>
#include <vector>

#include "graphlearn/graph_store.h"
#include "graphlearn/status.h"
#include "graphlearn/tensor.h"

namespace graphlearn {
namespace io {

enum Direction { kOrigin = 0, kReversed = 1 };  // include/constants.h

struct AttributeInfo {
  std::string delimiter;
  std::vector<DataType> types;        // kInt32/kInt64 -> int, kFloat/kDouble -> float, kString
  std::vector<int64_t> hash_buckets;  // empty, or one per type: > 0 hashes a string attribute into an int
  bool ignore_invalid;
  AttributeInfo();
  void AppendType(DataType type) { types.push_back(type); }
  void AppendHashBucket(int64_t bucket_size) { hash_buckets.push_back(bucket_size); }
};

struct NodeSource {
  std::string path;
  std::string id_type;
  int32_t format = kAttributed;
  AttributeInfo attr_info;
  IndexOption option;
  std::string view_type, use_attrs;  // vineyard views (data_source.h:86-87): carried, never read
  bool IsWeighted() const { return format & kWeighted; }
  bool IsLabeled() const { return format & kLabeled; }
  bool IsTimestamped() const { return format & kTimestamped; }
  bool IsAttributed() const { return format & kAttributed; }
};

struct EdgeSource {
  std::string path;
  std::string edge_type, src_id_type, dst_id_type;
  int32_t format = kWeighted;
  Direction direction = kOrigin;
  AttributeInfo attr_info;
  IndexOption option;
  std::string view_type, use_attrs;  // :142-143, as above
  bool IsWeighted() const { return format & kWeighted; }
  bool IsLabeled() const { return format & kLabeled; }
  bool IsTimestamped() const { return format & kTimestamped; }
  bool IsAttributed() const { return format & kAttributed; }
};

// MurmurHash64A with the reference's seed (common/base/hash.cc:99-150); string
// attributes with a bucket size are stored as Hash64(s) % bucket.
uint64_t Hash64(const char* data, size_t n);

// One packed attribute string -> typed values, with the reference's rules
// (parser.cc:39-104): token count must equal types.size(); ints via strtol, floats
// via strtof (trailing blanks allowed); the delimiter is a SET of characters and
// empty tokens are kept (string_tool.cc:33-49).
Status ParseAttribute(const char* data, size_t len, const AttributeInfo& info, std::vector<int64_t>* ints,
                      std::vector<float>* floats, std::vector<std::string>* strings);

// format + attribute counts of a source (parser.h:35-60)
template <class Source>
void ParseSideInfo(const Source& source, SideInfo* info) {
  info->i_num = info->f_num = info->s_num = 0;
  info->format = source.format;
  const AttributeInfo& a = source.attr_info;
  for (size_t i = 0; i < a.types.size(); ++i) {
    if (a.types[i] == kInt32 || a.types[i] == kInt64) ++info->i_num;
    else if (a.types[i] == kFloat || a.types[i] == kDouble) ++info->f_num;
    else if (!a.hash_buckets.empty() && a.hash_buckets[i] > 0) ++info->i_num;
    else ++info->s_num;
  }
}

// Read one file into the store (appending to the type's storage, like
// Initializer -> LocalGraph::UpdateEdges, graph_store.cc:60-120).  Records are
// appended in file order, so edge id = position in load order.
Status LoadEdges(const EdgeSource& source, GraphStore* store);
Status LoadNodes(const NodeSource& source, GraphStore* store);

}  // namespace io

template <class EdgeSources, class NodeSources>
Status GraphStore::Load(const EdgeSources& edges, const NodeSources& nodes) {
  IndexOption option;
  option.name = "sort";
  for (const auto& e : edges) DeclareEdgeType(e.edge_type);  // GraphStore::Init, graph_store.cc:185-208
  for (const auto& n : nodes) DeclareNodeType(n.id_type);
  for (const auto& e : edges) {
    Status s = io::LoadEdges(e, this);
    if (!s.ok()) return s;
    option = e.option.name.empty() ? option : e.option;
  }
  for (const auto& n : nodes) {
    Status s = io::LoadNodes(n, this);
    if (!s.ok()) return s;
  }
  return Build(option);
}

}  // namespace graphlearn
#endif  // GLX_HOST_DATA_SOURCE_H_
