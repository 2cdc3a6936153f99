// Restates the lookup / degree parts of graphlearn/src/core/operator/graph/test/
// graph_op_unittest.cpp (EdgeLookuper :500-601, NodeLookuper :603-703, DegreeGetter :787-826)
// against the glx host mirror: TSV sources -> GraphStore::Load (host parse, device build) ->
// OpFactory::Create(name)->Process.
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "graphlearn/graphlearn.h"
#include "test_util.h"

using namespace graphlearn;      // NOLINT
using namespace graphlearn::io;  // NOLINT
using namespace graphlearn::op;  // NOLINT

namespace {

void GenFile(const std::string& path, bool edge, int32_t format) {
  std::ofstream out(path);
  std::string title = edge ? "src_id:int64\tdst_id:int64" : "node_id:int64";
  if (format & kWeighted) title += "\tweight:float";
  if (format & kLabeled) title += "\tlabel:int32";
  if (format & kAttributed) title += "\tattribute:string";
  out << title << "\n";
  char buf[96];
  for (int32_t i = 0; i < 100; ++i) {
    std::string line = edge ? std::to_string(i) + "\t" + std::to_string(i) : std::to_string(i);
    if (format & kWeighted) {
      std::snprintf(buf, sizeof(buf), "\t%f", (float)i);
      line += buf;
    }
    if (format & kLabeled) line += "\t" + std::to_string(i);
    if (format & kAttributed) {
      std::snprintf(buf, sizeof(buf), "\t%d:%f:%c", i, (float)i, (char)(i % 26 + 'A'));
      line += buf;
    }
    out << line << "\n";
  }
}

void Attr(AttributeInfo* info, int32_t format) {
  if (format & kAttributed) {
    info->delimiter = ":";
    info->types = {kInt32, kFloat, kString};
    info->hash_buckets = {0, 0, 0};
  }
}

GraphStore* g_store = nullptr;

void SetUpStore() {
  if (g_store) return;
  GenFile("glx_w_edge_file", true, kWeighted);
  GenFile("glx_l_edge_file", true, kLabeled);
  GenFile("glx_a_edge_file", true, kAttributed);
  GenFile("glx_w_node_file", false, kWeighted);
  GenFile("glx_l_node_file", false, kLabeled);
  GenFile("glx_a_node_file", false, kAttributed);
  std::vector<EdgeSource> edges(3);
  const char* efiles[3] = {"glx_w_edge_file", "glx_l_edge_file", "glx_a_edge_file"};
  const char* etypes[3] = {"click", "buy", "watch"};
  const int32_t formats[3] = {kWeighted, kLabeled, kAttributed};
  for (int i = 0; i < 3; ++i) {
    edges[i].path = efiles[i];
    edges[i].edge_type = etypes[i];
    edges[i].src_id_type = "user";
    edges[i].dst_id_type = i == 2 ? "movie" : "item";
    edges[i].format = formats[i];
    Attr(&edges[i].attr_info, formats[i]);
  }
  std::vector<NodeSource> nodes(3);
  const char* nfiles[3] = {"glx_w_node_file", "glx_l_node_file", "glx_a_node_file"};
  const char* ntypes[3] = {"user", "item", "movie"};
  for (int i = 0; i < 3; ++i) {
    nodes[i].path = nfiles[i];
    nodes[i].id_type = ntypes[i];
    nodes[i].format = formats[i];
    Attr(&nodes[i].attr_info, formats[i]);
  }
  g_store = new GraphStore();
  Status s = g_store->Load(edges, nodes);
  for (const char* f : efiles) std::remove(f);
  for (const char* f : nfiles) std::remove(f);
  if (!s.ok()) {
    std::printf("store load failed: %s\n", s.ToString().c_str());
    std::exit(2);
  }
  OpFactory::GetInstance()->Set(g_store);
}

std::vector<int64_t> Iota(int32_t n) {
  std::vector<int64_t> v(n);
  for (int32_t i = 0; i < n; ++i) v[i] = i;
  return v;
}
}  // namespace

TEST(GraphOpTest, EdgeLookuper) {
  SetUpStore();
  const std::vector<int64_t> ids = Iota(100);  // edge ids = load order
  {
    LookupEdgesRequest req("click");
    LookupEdgesResponse res;
    req.Set(ids.data(), ids.data(), 100);
    Operator* op = OpFactory::GetInstance()->Create(req.Name());
    EXPECT_TRUE(op != nullptr);
    EXPECT_TRUE(op->Process(&req, &res).ok());
    EXPECT_EQ(res.Size(), 100);
    EXPECT_EQ(res.Format(), (int)kWeighted);
    EXPECT_EQ(res.IntAttrNum(), 0);
    EXPECT_EQ(res.FloatAttrNum(), 0);
    EXPECT_EQ(res.StringAttrNum(), 0);
    for (int32_t i = 0; i < 100; ++i) EXPECT_TRUE(res.Weights()[i] == (float)i);
  }
  {
    LookupEdgesRequest req("buy");
    LookupEdgesResponse res;
    req.Set(ids.data(), ids.data(), 100);
    EXPECT_TRUE(OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res).ok());
    EXPECT_EQ(res.Format(), (int)kLabeled);
    for (int32_t i = 0; i < 100; ++i) EXPECT_EQ(res.Labels()[i], i);
  }
  {
    LookupEdgesRequest req("watch");
    LookupEdgesResponse res;
    req.Set(ids.data(), ids.data(), 100);
    EXPECT_TRUE(OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res).ok());
    EXPECT_EQ(res.Format(), (int)kAttributed);
    EXPECT_EQ(res.IntAttrNum(), 1);
    EXPECT_EQ(res.FloatAttrNum(), 1);
    EXPECT_EQ(res.StringAttrNum(), 1);
    for (int32_t i = 0; i < 100; ++i) {
      EXPECT_EQ(res.IntAttrs()[i], (int64_t)i);
      EXPECT_TRUE(res.FloatAttrs()[i] == (float)i);
      EXPECT_EQ(res.StringAttrs()[i], std::string(1, (char)('A' + i % 26)));
    }
  }
}

TEST(GraphOpTest, NodeLookuper) {
  SetUpStore();
  const std::vector<int64_t> ids = Iota(100);
  {
    LookupNodesRequest req("user");
    LookupNodesResponse res;
    req.Set(ids.data(), 100);
    Operator* op = OpFactory::GetInstance()->Create(req.Name());
    EXPECT_TRUE(op != nullptr);
    EXPECT_TRUE(op->Process(&req, &res).ok());
    EXPECT_EQ(res.Size(), 100);
    EXPECT_EQ(res.Format(), (int)kWeighted);
    for (int32_t i = 0; i < 100; ++i) EXPECT_TRUE(res.Weights()[i] == (float)i);
  }
  {
    LookupNodesRequest req("item");
    LookupNodesResponse res;
    req.Set(ids.data(), 100);
    EXPECT_TRUE(OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res).ok());
    EXPECT_EQ(res.Format(), (int)kLabeled);
    for (int32_t i = 0; i < 100; ++i) EXPECT_EQ(res.Labels()[i], i);
  }
  {
    LookupNodesRequest req("movie");  // the float attribute comes back from the device (glx_lookup)
    LookupNodesResponse res;
    req.Set(ids.data(), 100);
    EXPECT_TRUE(OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res).ok());
    EXPECT_EQ(res.IntAttrNum(), 1);
    EXPECT_EQ(res.FloatAttrNum(), 1);
    EXPECT_EQ(res.StringAttrNum(), 1);
    for (int32_t i = 0; i < 100; ++i) {
      EXPECT_EQ(res.IntAttrs()[i], (int64_t)i);
      EXPECT_TRUE(res.FloatAttrs()[i] == (float)i);
      EXPECT_EQ(res.StringAttrs()[i], std::string(1, (char)('A' + i % 26)));
    }
  }
}

TEST(GraphOpTest, DegreeGetter) {
  SetUpStore();
  const std::vector<int64_t> ids = Iota(64);
  for (NodeFrom from : {kEdgeSrc, kEdgeDst}) {
    GetDegreeRequest req("click", from);
    req.Set(ids.data(), 64);
    GetDegreeResponse res;
    Operator* op = OpFactory::GetInstance()->Create(req.Name());
    EXPECT_TRUE(op != nullptr);
    EXPECT_TRUE(op->Process(&req, &res).ok());
    for (int32_t i = 0; i < 64; ++i) EXPECT_EQ(res.GetDegrees()[i], 1);  // every i has exactly the edge i -> i
  }
  std::vector<int64_t> unknown = {100, 5000, -1};
  GetDegreeRequest req("click", kEdgeSrc);
  req.Set(unknown.data(), 3);
  GetDegreeResponse res;
  EXPECT_TRUE(OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res).ok());
  for (int32_t i = 0; i < 3; ++i) EXPECT_EQ(res.GetDegrees()[i], 0);
}

// RandomWalk (core/operator/random_walk/random_walk.cc) on the self-loop fixture: a walker stays where it is, an
// unknown id yields the default id and walks on from it; DeepWalk and node2vec requests share the operator.
// graph_op_unittest.cpp:705-789 (NodeCountGetter / EdgeCountGetter): 100 records per type, one count per
// declared type -- edge types first, then node types, each group in type-name order.
TEST(GraphOpTest, CountGetter) {
  SetUpStore();
  for (int32_t i = 0; i < 3; ++i) {
    GetCountRequest req;
    GetCountResponse res;
    Operator* op = OpFactory::GetInstance()->Create(req.Name());
    EXPECT_TRUE(op != nullptr);
    EXPECT_TRUE(op->Process(&req, &res).ok());
    EXPECT_EQ(res.Size(), 6);
    for (int32_t j = 0; j < res.Size(); ++j) EXPECT_EQ(res.Count()[j], 100);
  }
}

// stats_getter.cc:25-48 over GraphStore::BuildStatistics: on one server every type maps to one count.
TEST(GraphOpTest, StatsGetter) {
  SetUpStore();
  GetStatsRequest req;
  GetStatsResponse res;
  Operator* op = OpFactory::GetInstance()->Create(req.Name());
  EXPECT_TRUE(op != nullptr);
  EXPECT_TRUE(op->Process(&req, &res).ok());
  Counts c = res.GetCounts();
  EXPECT_EQ(c.size(), (size_t)6);
  for (const char* t : {"click", "buy", "watch", "user", "item", "movie"}) {
    EXPECT_EQ(c[t].size(), (size_t)1);
    EXPECT_EQ(c[t][0], 100);
  }
}

// GraphStore::Init (graph_store.cc:196-201): an edge type declared by two sources counts double.
TEST(GraphOpTest, UndirectedEdgeTypeCountsTwice) {
  GraphStore store;
  store.DeclareEdgeType("e");
  store.DeclareEdgeType("e");
  store.DeclareNodeType("n");
  io::EdgeValue v;
  v.src_id = 1;
  v.dst_id = 2;
  store.GetGraph("e")->Add(&v);
  store.GetGraph("e")->Add(&v);
  store.GetGraph("e")->Add(&v);
  std::vector<int32_t> c = store.GetLocalCount();
  EXPECT_EQ(c.size(), (size_t)2);
  EXPECT_EQ(c[0], 6);
  EXPECT_EQ(c[1], 0);
}

TEST(GraphOpTest, RandomWalk) {
  SetUpStore();
  for (float p : {1.0f, 0.5f}) {
    RandomWalkRequest req("click", p, p == 1.0f ? 1.0f : 2.0f, 4);
    RandomWalkResponse res;
    int64_t ids[3] = {5, 42, 200};
    req.Set(ids, 3);
    EXPECT_TRUE(req.IsDeepWalk() == (p == 1.0f));
    Operator* op = OpFactory::GetInstance()->Create(req.Name());
    EXPECT_TRUE(op != nullptr);
    EXPECT_TRUE(op->Process(&req, &res).ok());
    EXPECT_EQ(res.WalkLen(), 4);
    for (int t = 0; t < 4; ++t) {
      EXPECT_EQ(res.GetWalks()[t], 5);
      EXPECT_EQ(res.GetWalks()[4 + t], 42);
      EXPECT_EQ(res.GetWalks()[8 + t], 0);  // default id, then vertex 0's self loop
    }
  }
  RandomWalkRequest nobody("no-such-type", 1.0f, 1.0f, 2);
  RandomWalkResponse res;
  int64_t id = 3;
  nobody.Set(&id, 1);
  EXPECT_TRUE(OpFactory::GetInstance()->Create("RandomWalk")->Process(&nobody, &res).ok());
  EXPECT_EQ(res.GetWalks()[0], 0);
  EXPECT_EQ(res.GetWalks()[1], 0);
}

TEST(GraphOpTest, EdgeGetter) {
  // graph_op_unittest.cpp:219-281: batches of 12 over 100 edges: 8 full, one of 4, then OutOfRange
  SetUpStore();
  for (const char* type : {"click", "buy", "watch"}) {
    std::vector<int64_t> seen;
    for (int index = 0; index < 10; ++index) {
      GetEdgesRequest req(type, "by_order", 12);
      GetEdgesResponse res;
      Operator* op = OpFactory::GetInstance()->Create(req.Name());
      EXPECT_TRUE(op != nullptr);
      Status s = op->Process(&req, &res);
      if (index == 9) {
        EXPECT_EQ((int)s.code(), (int)error::OUT_OF_RANGE);
        break;
      }
      EXPECT_TRUE(s.ok());
      EXPECT_EQ(res.Size(), index < 8 ? 12 : 4);
      for (int32_t i = 0; i < res.Size(); ++i) {
        EXPECT_EQ(res.SrcIds()[i], res.DstIds()[i]);
        EXPECT_EQ(res.EdgeIds()[i], res.SrcIds()[i]);  // record i is edge i
        seen.push_back(res.SrcIds()[i]);
      }
    }
    EXPECT_EQ(seen.size(), (size_t)100);
    for (size_t i = 0; i < seen.size(); ++i) EXPECT_EQ(seen[i], (int64_t)i);
    // the next epoch needs a request that knows about it
    GetEdgesRequest stale(type, "by_order", 12, 0);
    GetEdgesResponse r0;
    EXPECT_EQ((int)OpFactory::GetInstance()->Create("GetEdges")->Process(&stale, &r0).code(), (int)error::OUT_OF_RANGE);
    GetEdgesRequest fresh(type, "by_order", 12, 1);
    GetEdgesResponse r1;
    EXPECT_TRUE(OpFactory::GetInstance()->Create("GetEdges")->Process(&fresh, &r1).ok());
    EXPECT_EQ(r1.Size(), 12);
  }
}

TEST(GraphOpTest, NodeGetters) {
  // NodeGetter :283-341, ShuffledNodeGetter :343-401, NodeGetterFromEdgeSrc :403-498
  SetUpStore();
  for (const char* strategy : {"by_order", "shuffle"}) {
    for (const char* type : {"user", "item", "movie"}) {
      std::vector<int64_t> seen;
      for (int index = 0; index < 10; ++index) {
        GetNodesRequest req(type, strategy, kNode, 12);
        GetNodesResponse res;
        Status s = OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res);
        if (index == 9) {
          EXPECT_EQ((int)s.code(), (int)error::OUT_OF_RANGE);
          break;
        }
        EXPECT_TRUE(s.ok());
        EXPECT_EQ(res.Size(), index < 8 ? 12 : 4);
        for (int32_t i = 0; i < res.Size(); ++i) seen.push_back(res.NodeIds()[i]);
      }
      std::vector<int64_t> sorted = seen;
      std::sort(sorted.begin(), sorted.end());
      for (size_t i = 0; i < sorted.size(); ++i) EXPECT_EQ(sorted[i], (int64_t)i);  // every id exactly once
      if (std::string(strategy) == "by_order") EXPECT_TRUE(seen == sorted);
      else EXPECT_TRUE(seen != sorted);
    }
  }
  for (NodeFrom from : {kEdgeSrc, kEdgeDst}) {
    GetNodesRequest req("click", "by_order", from, 64);
    GetNodesResponse res;
    EXPECT_TRUE(OpFactory::GetInstance()->Create(req.Name())->Process(&req, &res).ok());
    EXPECT_EQ(res.Size(), 64);
    for (int32_t i = 0; i < 64; ++i) EXPECT_EQ(res.NodeIds()[i], (int64_t)i);
  }
  GetNodesRequest rnd("user", "random", kNode, 500);
  GetNodesResponse res;
  EXPECT_TRUE(OpFactory::GetInstance()->Create(rnd.Name())->Process(&rnd, &res).ok());
  EXPECT_EQ(res.Size(), 500);
  for (int32_t i = 0; i < 500; ++i) EXPECT_TRUE(res.NodeIds()[i] >= 0 && res.NodeIds()[i] < 100);
}

// A GSL query as the Python layer lowers it (python/gsl/dag_node.py) -- GetNodes -> two Topk hops -> the lookups of
// every traversal node -> Sink -- through Client::RunDag / GetDagValues on the device store: the two hops are ONE
// compiled step (glx_sample_hops), every round's values equal the same operators called one by one, rounds arrive
// in root order, and the root's OUT_OF_RANGE ends the epoch with an invalid response (core/runner/dag_node_runner.cc,
// dag_scheduler.cc, service/executor.cc:46-71).
TEST(GraphOpTest, QueryDagOnTheDeviceStore) {
  SetUpStore();
  auto str = [](const std::string& v) { Tensor t(kString, 1); t.AddString(v); return t; };
  auto i32 = [](int32_t v) { Tensor t(kInt32, 1); t.AddInt32(v); return t; };
  int32_t eid = 9000;
  auto link = [&eid](DagNodeDef* src, DagNodeDef* dst, const char* out, const char* in) {
    DagEdgeDef e;
    e.id = eid++;
    e.src_output = out;
    e.dst_input = in;
    src->out_edges.push_back(e);
    dst->in_edges.push_back(e);
  };
  DagNodeDef root, look0, hop1, look1, hop2, look2, sink;
  root.id = 1; root.op_name = "GetNodes";
  root.params[kNodeType] = str("user"); root.params[kNodeFrom] = i32(kNode); root.params[kStrategy] = str("by_order");
  root.params[kBatchSize] = i32(16); root.params[kEpoch] = i32(0x7fffffff);
  look0.id = 2; look0.op_name = "LookupNodes"; look0.params[kNodeType] = str("user");
  hop1.id = 3; hop1.op_name = "TopkSampler";
  hop1.params[kEdgeType] = str("click"); hop1.params[kStrategy] = str("TopkSampler"); hop1.params[kNeighborCount] = i32(2);
  look1.id = 4; look1.op_name = "LookupNodes"; look1.params[kNodeType] = str("item");
  hop2.id = 5; hop2.op_name = "TopkSampler";
  hop2.params[kEdgeType] = str("buy"); hop2.params[kStrategy] = str("TopkSampler"); hop2.params[kNeighborCount] = i32(3);
  look2.id = 6; look2.op_name = "LookupNodes"; look2.params[kNodeType] = str("item");
  sink.id = 7; sink.op_name = "Sink";
  link(&root, &look0, kNodeIds, kNodeIds);
  link(&root, &hop1, kNodeIds, kSrcIds);
  link(&hop1, &look1, kNodeIds, kNodeIds);
  link(&hop1, &hop2, kNodeIds, kSrcIds);
  link(&hop2, &look2, kNodeIds, kNodeIds);
  for (DagNodeDef* n : {&root, &look0, &hop1, &look1, &hop2, &look2}) link(n, &sink, "fake", "fake");
  DagDef def;
  def.id = 4711;
  def.nodes = {root, look0, hop1, look1, hop2, look2, sink};
  {
    Dag probe(def);
    EXPECT_TRUE(probe.Compile().ok());
    size_t fused = 0;
    for (const Dag::Step& st : probe.Steps()) fused += st.nodes.size() > 1 ? 1 : 0;
    EXPECT_EQ(fused, (size_t)1);  // hop1 + hop2
    EXPECT_EQ(probe.Steps().size(), (size_t)6);
  }
  SetGlobalFlagTapeCapacity(2);
  Client* client = NewInMemoryClient();
  DagRequest req;
  req.ParseFrom(&def, false);
  EXPECT_TRUE(client->RunDag(&req).ok());
  for (int32_t round = 0; round < 8; ++round) {
    GetDagValuesRequest get(4711);
    GetDagValuesResponse res;
    EXPECT_TRUE(client->GetDagValues(&get, &res).ok());
    if (round == 7) {  // 100 ids = 6 x 16 + 4, then the epoch ends
      EXPECT_TRUE(!res.Valid());
      EXPECT_EQ(res.Epoch(), 0);
      break;
    }
    EXPECT_TRUE(res.Valid());
    const int32_t n = round < 6 ? 16 : 4;
    const Tensor* ids = res.GetValue(1, kNodeIds).first;
    EXPECT_TRUE(ids != nullptr && ids->Size() == n);
    if (!ids) break;
    for (int32_t i = 0; i < n; ++i) EXPECT_EQ(ids->GetInt64(i), (int64_t)(round * 16 + i));
    // the same two hops as separate requests (Topk is deterministic)
    SamplingRequest r1("click", "TopkSampler", 2);
    SamplingResponse s1;
    r1.Set(ids->GetInt64(), n);
    EXPECT_TRUE(OpFactory::GetInstance()->Create("TopkSampler")->Process(&r1, &s1).ok());
    SamplingRequest r2("buy", "TopkSampler", 3);
    SamplingResponse s2;
    r2.Set(s1.GetNeighborIds(), n * 2);
    EXPECT_TRUE(OpFactory::GetInstance()->Create("TopkSampler")->Process(&r2, &s2).ok());
    const Tensor* b = res.GetValue(3, kNodeIds).first;
    const Tensor* be = res.GetValue(3, kEdgeIds).first;
    const Tensor* c = res.GetValue(5, kNodeIds).first;
    EXPECT_TRUE(b && be && c && b->Size() == n * 2 && c->Size() == n * 6);
    if (!b || !be || !c) break;
    for (int32_t i = 0; i < n * 2; ++i) {
      EXPECT_EQ(b->GetInt64(i), s1.GetNeighborIds()[i]);
      EXPECT_EQ(be->GetInt64(i), s1.GetEdgeIds()[i]);
      EXPECT_EQ(b->GetInt64(i), ids->GetInt64(i / 2));  // edge i -> i, circular padding
    }
    for (int32_t i = 0; i < n * 6; ++i) EXPECT_EQ(c->GetInt64(i), s2.GetNeighborIds()[i]);
    EXPECT_TRUE(res.GetValue(3, kNodeIds).second == nullptr);  // dense: no segments
    // lookups: user weights (= id), item labels (= id) for both hops
    const Tensor* w0 = res.GetValue(2, kWeightKey).first;
    const Tensor* l1 = res.GetValue(4, kLabelKey).first;
    const Tensor* l2 = res.GetValue(6, kLabelKey).first;
    EXPECT_TRUE(w0 && l1 && l2 && w0->Size() == n && l1->Size() == n * 2 && l2->Size() == n * 6);
    if (!w0 || !l1 || !l2) break;
    for (int32_t i = 0; i < n; ++i) EXPECT_TRUE(w0->GetFloat(i) == (float)ids->GetInt64(i));
    for (int32_t i = 0; i < n * 2; ++i) EXPECT_EQ((int64_t)l1->GetInt32(i), b->GetInt64(i));
    for (int32_t i = 0; i < n * 6; ++i) EXPECT_EQ((int64_t)l2->GetInt32(i), c->GetInt64(i));
    EXPECT_TRUE(res.GetValue(7, kNodeIds).first == nullptr);  // the sink records nothing
  }
  DagScheduler::StopAll();
  delete client;
}

int main() { return RunAllTests(); }
