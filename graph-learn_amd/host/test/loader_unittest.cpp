// Restates graphlearn/src/core/io/test/edge_loader_unittest.cpp and
// node_loader_unittest.cpp against the glx loader (host/src/loader.cc): every combination of
// the optional columns, several files of one type, directory sources, and what the
// reference's loaders reject.  Loading is host work: this test needs no GPU (the stores are
// never built).
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>

#include "graphlearn/graphlearn.h"
#include "test_util.h"

using namespace graphlearn;      // NOLINT
using namespace graphlearn::io;  // NOLINT

namespace {

std::string Title(bool edge, int32_t format) {
  std::string t = edge ? "src_id:int64\tdst_id:int64" : "node_id:int64";
  if (format & kWeighted) t += edge ? "\tedge_weight:float" : "\tnode_weight:float";
  if (format & kLabeled) t += "\tlabel:int32";
  if (format & kAttributed) t += "\tattribute:string";
  return t + "\n";
}

// record i: ids = i, weight = float(i), label = i, attributes "i:float(i):<letter>"
void GenFile(const std::string& path, bool edge, int32_t format, int32_t offset = 0, int32_t count = 100) {
  std::ofstream out(path);
  out << Title(edge, format);
  char buf[128];
  for (int32_t i = offset; i < offset + count; ++i) {
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

void FillAttrInfo(AttributeInfo* info, int32_t format) {
  info->ignore_invalid = false;
  if (format & kAttributed) {
    info->delimiter = ":";
    info->types = {kInt32, kFloat, kString};
    info->hash_buckets = {0, 0, 0};
  }
}

EdgeSource EdgeSrc(const std::string& path, int32_t format, const char* type, const char* src, const char* dst) {
  EdgeSource s;
  s.path = path;
  s.format = format;
  s.edge_type = type;
  s.src_id_type = src;
  s.dst_id_type = dst;
  FillAttrInfo(&s.attr_info, format);
  return s;
}

NodeSource NodeSrc(const std::string& path, int32_t format, const char* type) {
  NodeSource s;
  s.path = path;
  s.format = format;
  s.id_type = type;
  FillAttrInfo(&s.attr_info, format);
  return s;
}

void CheckEdges(Graph* g, int32_t format, int64_t from, int64_t to, int64_t first_edge_id = 0) {
  const SideInfo* info = g->GetSideInfo();
  EXPECT_EQ(info->format, format);
  for (int64_t i = from; i < to; ++i) {
    const int64_t e = first_edge_id + (i - from);
    EXPECT_EQ(g->GetSrcId(e), i);
    EXPECT_EQ(g->GetDstId(e), i);
    if (format & kWeighted) EXPECT_TRUE(g->GetEdgeWeight(e) == (float)i);
    if (format & kLabeled) EXPECT_EQ(g->GetEdgeLabel(e), (int32_t)i);
    if (format & kAttributed) {
      EXPECT_EQ(g->GetEdgeIntAttrs(e)[0], i);
      EXPECT_TRUE(g->GetEdgeFloatAttrs(e)[0] == (float)i);
      EXPECT_EQ(g->GetEdgeStringAttrs(e)[0], std::string(1, (char)('A' + i % 26)));
    }
  }
}

void CheckNodes(Noder* n, int32_t format, int64_t from, int64_t to) {
  for (int64_t i = from; i < to; ++i) {
    const int32_t row = n->RowOf(i);
    EXPECT_TRUE(row >= 0);
    if (format & kWeighted) EXPECT_TRUE(n->GetWeight(i) == (float)i);
    if (format & kLabeled) EXPECT_EQ(n->GetLabel(i), (int32_t)i);
    if (format & kAttributed) {
      EXPECT_EQ(n->GetIntAttrs(row)[0], i);
      EXPECT_TRUE(n->GetFloatAttrs(row)[0] == (float)i);
      EXPECT_EQ(n->GetStringAttrs(row)[0], std::string(1, (char)('A' + i % 26)));
    }
  }
}

const int32_t kFormats[] = {kWeighted, kLabeled, kAttributed, kWeighted | kLabeled, kWeighted | kAttributed,
                            kLabeled | kAttributed, kWeighted | kLabeled | kAttributed};
}  // namespace

TEST(EdgeLoaderTest, EveryColumnCombination) {
  // edge_loader_unittest.cpp: ReadWeightedLabeled / ReadWeightedAttributed / ReadLabeledAttributed /
  // ReadWeightedLabeledAttributed (+ the single-column files of ReadMultiFiles)
  for (int32_t format : kFormats) {
    const std::string file = "glx_efile_" + std::to_string(format);
    GenFile(file, true, format);
    GraphStore store;
    EXPECT_TRUE(LoadEdges(EdgeSrc(file, format, "click", "user", "item"), &store).ok());
    Graph* g = store.GetGraph("click");
    EXPECT_EQ(g->GetEdgeCount(), (int64_t)100);
    EXPECT_EQ(g->GetSideInfo()->type, std::string("click"));
    EXPECT_EQ(g->GetSideInfo()->src_type, std::string("user"));
    EXPECT_EQ(g->GetSideInfo()->dst_type, std::string("item"));
    CheckEdges(g, format, 0, 100);
    std::remove(file.c_str());
  }
}

TEST(EdgeLoaderTest, MultiFilesAndDirectories) {
  // ReadMultiFiles + ReadDirectories: files of one type append in the order given / in sorted
  // name order, edge ids keep counting
  ::mkdir("glx_weighted_efiles", 0755);
  for (int i = 0; i < 3; ++i) GenFile("glx_weighted_efiles/" + std::to_string(i) + "_#100", true, kWeighted, 100 * i);
  GenFile("glx_extra_efile", true, kWeighted, 300);
  GraphStore store;
  EXPECT_TRUE(LoadEdges(EdgeSrc("glx_weighted_efiles/", kWeighted, "click", "user", "item"), &store).ok());
  EXPECT_TRUE(LoadEdges(EdgeSrc("glx_extra_efile", kWeighted, "click", "user", "item"), &store).ok());
  Graph* g = store.GetGraph("click");
  EXPECT_EQ(g->GetEdgeCount(), (int64_t)400);
  CheckEdges(g, kWeighted, 0, 400);
  // a reversed source swaps the end points (edge_loader.cc:66-68)
  EdgeSource rev = EdgeSrc("glx_extra_efile", kWeighted, "click_reverse", "item", "user");
  rev.direction = kReversed;
  GenFile("glx_asym_efile", true, kWeighted, 0, 1);
  {
    std::ofstream out("glx_asym_efile");
    out << Title(true, kWeighted) << "7\t9\t0.5\n";
  }
  rev.path = "glx_asym_efile";
  EXPECT_TRUE(LoadEdges(rev, &store).ok());
  EXPECT_EQ(store.GetGraph("click_reverse")->GetSrcId(0), (int64_t)9);
  EXPECT_EQ(store.GetGraph("click_reverse")->GetDstId(0), (int64_t)7);
  for (int i = 0; i < 3; ++i) std::remove(("glx_weighted_efiles/" + std::to_string(i) + "_#100").c_str());
  ::rmdir("glx_weighted_efiles");
  std::remove("glx_extra_efile");
  std::remove("glx_asym_efile");
}

TEST(NodeLoaderTest, EveryColumnCombinationAndDirectories) {
  for (int32_t format : kFormats) {
    const std::string file = "glx_nfile_" + std::to_string(format);
    GenFile(file, false, format);
    GraphStore store;
    EXPECT_TRUE(LoadNodes(NodeSrc(file, format, "user"), &store).ok());
    Noder* n = store.GetNoder("user");
    EXPECT_EQ(n->GetNodeCount(), (int64_t)100);
    EXPECT_EQ(n->GetSideInfo()->format, format);
    CheckNodes(n, format, 0, 100);
    std::remove(file.c_str());
  }
  ::mkdir("glx_labeled_nfiles", 0755);
  for (int i = 0; i < 2; ++i) GenFile("glx_labeled_nfiles/" + std::to_string(i) + "_#100", false, kLabeled, 100 * i);
  GraphStore store;
  EXPECT_TRUE(LoadNodes(NodeSrc("glx_labeled_nfiles", kLabeled, "item"), &store).ok());
  EXPECT_EQ(store.GetNoder("item")->GetNodeCount(), (int64_t)200);
  CheckNodes(store.GetNoder("item"), kLabeled, 0, 200);
  for (int i = 0; i < 2; ++i) std::remove(("glx_labeled_nfiles/" + std::to_string(i) + "_#100").c_str());
  ::rmdir("glx_labeled_nfiles");
}

TEST(LoaderTest, ShardsKeepWhatHashesToThem) {
  // One process per GPU, all reading the same files: shard r of P keeps the edges whose source id and the nodes
  // whose id satisfy llabs(id) % P == r (HashPartitioner's rule, hash_partitioner.h:88-90); edge ids are the
  // shard's own load order.  The shards are disjoint and together hold every record.
  const int32_t format = kWeighted | kLabeled | kAttributed;
  GenFile("glx_shard_efile", true, format);
  GenFile("glx_shard_nfile", false, format);
  {  // a negative id lands on llabs(id) % P
    std::ofstream extra("glx_shard_efile", std::ios::app);
    extra << "-7\t3\t1.500000\t9\t9:9.000000:J\n";
  }
  int64_t edges_seen = 0, nodes_seen = 0;
  for (int32_t r = 0; r < 3; ++r) {
    GraphStore store;
    store.SetShard(r, 3);
    EXPECT_EQ(store.ShardIndex(), r);
    EXPECT_EQ(store.ShardCount(), 3);
    EXPECT_TRUE(LoadEdges(EdgeSrc("glx_shard_efile", format, "click", "user", "item"), &store).ok());
    EXPECT_TRUE(LoadNodes(NodeSrc("glx_shard_nfile", format, "user"), &store).ok());
    Graph* g = store.GetGraph("click");
    Noder* n = store.GetNoder("user");
    int64_t e = 0;
    for (int64_t i = 0; i < 100; ++i) {
      if (i % 3 != r) {
        EXPECT_TRUE(n->RowOf(i) < 0);
        continue;
      }
      CheckNodes(n, format, i, i + 1);
      CheckEdges(g, format, i, i + 1, e);  // the shard's e-th edge is record i
      ++e;
    }
    if (r == 7 % 3) {
      EXPECT_EQ(g->GetSrcId(e), (int64_t)-7);
      EXPECT_EQ(g->GetDstId(e), (int64_t)3);
      ++e;
    }
    EXPECT_EQ(g->GetEdgeCount(), e);
    edges_seen += g->GetEdgeCount();
    nodes_seen += n->GetNodeCount();
  }
  EXPECT_EQ(edges_seen, (int64_t)101);
  EXPECT_EQ(nodes_seen, (int64_t)100);
  GraphStore whole;  // the default keeps everything
  EXPECT_TRUE(LoadEdges(EdgeSrc("glx_shard_efile", format, "click", "user", "item"), &whole).ok());
  EXPECT_EQ(whole.GetGraph("click")->GetEdgeCount(), (int64_t)101);
  std::remove("glx_shard_efile");
  std::remove("glx_shard_nfile");
}

TEST(LoaderTest, RejectsWhatTheReferenceRejects) {
  GenFile("glx_bad_schema", true, kWeighted);
  GraphStore store;
  // decoder says labeled, the file says weighted (edge_loader.cc:110-160)
  Status s = LoadEdges(EdgeSrc("glx_bad_schema", kLabeled, "e", "a", "b"), &store);
  EXPECT_TRUE(error::IsInvalidArgument(s));
  // missing types (edge_loader.cc:94-104)
  EdgeSource untyped = EdgeSrc("glx_bad_schema", kWeighted, "", "a", "b");
  EXPECT_TRUE(error::IsInvalidArgument(LoadEdges(untyped, &store)));
  EXPECT_EQ((int)LoadEdges(EdgeSrc("glx_no_such_file", kWeighted, "e", "a", "b"), &store).code(), (int)error::NOT_FOUND);
  // wrong attribute count / a non-numeric attribute: fatal unless ignore_invalid
  {
    std::ofstream out("glx_bad_attr");
    out << Title(true, kAttributed) << "1\t2\t1:1.0:A\n3\t4\t1:oops:B\n5\t6\t2:2.0\n7\t8\t3:3.0:C\n";
  }
  EdgeSource strict = EdgeSrc("glx_bad_attr", kAttributed, "strict", "a", "b");
  EXPECT_TRUE(error::IsInvalidArgument(LoadEdges(strict, &store)));
  EdgeSource lenient = EdgeSrc("glx_bad_attr", kAttributed, "lenient", "a", "b");
  lenient.attr_info.ignore_invalid = true;
  EXPECT_TRUE(LoadEdges(lenient, &store).ok());
  Graph* g = store.GetGraph("lenient");
  EXPECT_EQ(g->GetEdgeCount(), (int64_t)2);  // the two malformed records took no edge id
  EXPECT_EQ(g->GetSrcId(1), (int64_t)7);
  EXPECT_EQ(g->GetEdgeStringAttrs(1)[0], std::string("C"));
  // out-of-range edge ids answer with the defaults (memory_edge_storage.cc:90-125)
  EXPECT_EQ(g->GetSrcId(5), (int64_t)-1);
  EXPECT_TRUE(g->GetEdgeIntAttrs(-1) == nullptr);
  std::remove("glx_bad_schema");
  std::remove("glx_bad_attr");
}

TEST(UpdaterTest, RecordsArriveThroughTheRegistry) {
  // The reference's loader hands every batch of records to the operator registered as "UpdateEdges" / "UpdateNodes"
  // (edge_updater.cc:25-55, node_updater.cc; graph_store.cc:210-250).  Same names, same request surface here.
  GraphStore store;
  op::OpFactory::GetInstance()->Set(&store);
  io::SideInfo einfo;
  einfo.type = "click";
  einfo.src_type = "user";
  einfo.dst_type = "item";
  einfo.format = kWeighted | kLabeled;
  UpdateEdgesRequest ereq(&einfo, 5);
  for (int32_t i = 0; i < 5; ++i) {
    io::EdgeValue v;
    v.src_id = i;
    v.dst_id = 100 + i;
    v.weight = 0.5f * i;
    v.label = i % 2;
    ereq.Append(&v);
  }
  EXPECT_EQ(ereq.Name(), std::string("UpdateEdges"));
  EXPECT_EQ(ereq.ShardKey(), std::string(kSrcIds));
  EXPECT_TRUE(ereq.IsShardable());
  EXPECT_EQ(ereq.Size(), 5);
  op::Operator* eop = op::OpFactory::GetInstance()->Create("UpdateEdges");
  EXPECT_TRUE(eop != nullptr);
  std::unique_ptr<OpResponse> eres(RequestFactory::GetInstance()->NewResponse("UpdateEdges"));
  EXPECT_TRUE(dynamic_cast<UpdateEdgesResponse*>(eres.get()) != nullptr);
  EXPECT_TRUE(eop->Process(&ereq, eres.get()).ok());
  EXPECT_TRUE(eop->Process(&ereq, eres.get()).ok());  // a second batch appends: edge id = arrival order
  Graph* g = store.GetGraph("click");
  EXPECT_EQ(g->GetEdgeCount(), (int64_t)10);
  EXPECT_EQ(g->GetSrcId(7), (int64_t)2);
  EXPECT_EQ(g->GetDstId(7), (int64_t)102);
  EXPECT_FLOAT_EQ(g->GetEdgeWeight(3), 1.5f);
  EXPECT_EQ(g->GetEdgeLabel(3), 1);
  io::EdgeValue back;
  int32_t seen = 0;
  while (ereq.Next(&back)) {  // the cursor the reference's storages read a batch with (local_graph.cc:57-62)
    EXPECT_EQ(back.src_id, (int64_t)seen);
    ++seen;
  }
  EXPECT_EQ(seen, 5);

  io::SideInfo ninfo;
  ninfo.type = "user";
  ninfo.format = kWeighted | kAttributed;
  ninfo.f_num = 2;
  std::unique_ptr<OpRequest> made(RequestFactory::GetInstance()->NewRequest("UpdateNodes"));
  EXPECT_TRUE(dynamic_cast<UpdateNodesRequest*>(made.get()) != nullptr);
  UpdateNodesRequest nreq(&ninfo, 3);
  for (int32_t i = 0; i < 3; ++i) {
    io::NodeValue v;
    v.id = 10 + i;
    v.weight = 1.0f + i;
    v.attrs = {0.25f * i, -1.0f};
    nreq.Append(&v);
  }
  EXPECT_EQ(nreq.ShardKey(), std::string(kNodeIds));
  op::Operator* nop = op::OpFactory::GetInstance()->Create("UpdateNodes");
  EXPECT_TRUE(nop != nullptr);
  UpdateNodesResponse nres;
  EXPECT_TRUE(nop->Process(&nreq, &nres).ok());
  EXPECT_TRUE(nop->Process(&nreq, &nres).ok());  // duplicate ids are ignored (node_storage.h:41)
  Noder* n = store.GetNoder("user");
  EXPECT_EQ(n->GetNodeCount(), (int64_t)3);
  EXPECT_FLOAT_EQ(n->GetWeight(12), 3.0f);
  EXPECT_FLOAT_EQ(n->GetFloatAttrs(n->RowOf(11))[0], 0.25f);
  // the wrong request type is an error, not a crash
  EXPECT_TRUE(error::IsInvalidArgument(nop->Process(&ereq, &nres)));
  EXPECT_TRUE(error::IsInvalidArgument(eop->Process(&nreq, eres.get())));
}

int main() { return RunAllTests(); }
