// The query-DAG machinery (host dag.h) on the CPU, with stand-in operators registered under test-only names: no
// kernel is launched here.  Covers what the reference's core/dag/test/{dag,tape,dag_dataset}_unittest.cpp and
// core/runner/test/dag_scheduler_unittest.cpp cover -- node / edge wiring from a definition, tape records reaching
// the consumer in root order, a failed round = a faked tape = the end of an epoch, Dataset::Next's epoch rule --
// plus this design's compile step: topological order and the fusion of dense sampling hops.
#include <atomic>
#include <chrono>
#include <thread>

#include "graphlearn/graphlearn.h"
#include "test_util.h"

using namespace graphlearn;  // NOLINT

namespace {

// "TestCounter": the root.  Emits `bs` consecutive ids per round, 10 ids per epoch; the call after the last batch of
// an epoch answers OUT_OF_RANGE (what GetNodes does at an epoch boundary).
class CounterRequest : public OpRequest {
public:
  void Init(const Tensor::Map& params) override { batch_ = params.at(kBatchSize).GetInt32(0); }
  std::string Name() const override { return "TestCounter"; }
  int32_t batch_ = 0;
};
std::atomic<int64_t> g_counter_at{0};
std::atomic<int32_t> g_double_calls{0};
class CounterOp : public op::Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const int32_t bs = static_cast<const CounterRequest*>(req)->batch_;
    int64_t at = g_counter_at.load();
    if (at >= 10) {
      g_counter_at = 0;
      return error::OutOfRange("No more nodes exist.");
    }
    ADD_TENSOR(res->tensors_, kNodeIds, kInt64, bs);
    for (int32_t i = 0; i < bs && at < 10; ++i, ++at) res->tensors_[kNodeIds].AddInt64(at);
    g_counter_at = at;
    return Status::OK();
  }
};

// "TestDouble": ids in under kSrcIds -> 2 * id out under kNodeIds; fails on request (a parameter names the id).
class DoubleRequest : public OpRequest {
public:
  void Init(const Tensor::Map& params) override { fail_on_ = params.count("fail_on") ? params.at("fail_on").GetInt32(0) : -1; }
  void Set(const Tensor::Map& tensors, const SparseTensor::Map&) override { ids_ = tensors.at(kSrcIds); }
  std::string Name() const override { return "TestDouble"; }
  Tensor ids_;
  int32_t fail_on_ = -1;
};
class DoubleOp : public op::Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const DoubleRequest* r = static_cast<const DoubleRequest*>(req);
    ++g_double_calls;
    ADD_TENSOR(res->tensors_, kNodeIds, kInt64, r->ids_.Size());
    for (int32_t i = 0; i < r->ids_.Size(); ++i) {
      if (r->ids_.GetInt64(i) == r->fail_on_) return error::Internal("asked to fail");
      res->tensors_[kNodeIds].AddInt64(2 * r->ids_.GetInt64(i));
    }
    return Status::OK();
  }
};

REGISTER_REQUEST(TestCounter, CounterRequest, OpResponse)
REGISTER_REQUEST(TestDouble, DoubleRequest, OpResponse)
REGISTER_OPERATOR("TestCounter", CounterOp)
REGISTER_OPERATOR("TestDouble", DoubleOp)

Tensor IntParam(int32_t v) {
  Tensor t(kInt32, 1);
  t.AddInt32(v);
  return t;
}
Tensor StrParam(const std::string& v) {
  Tensor t(kString, 1);
  t.AddString(v);
  return t;
}
DagEdgeDef Edge(int32_t id, const std::string& out, const std::string& in) {
  DagEdgeDef e;
  e.id = id;
  e.src_output = out;
  e.dst_input = in;
  return e;
}
DagNodeDef Node(int32_t id, const std::string& op) {
  DagNodeDef n;
  n.id = id;
  n.op_name = op;
  return n;
}
void Link(DagNodeDef* src, DagNodeDef* dst, const DagEdgeDef& e) {
  src->out_edges.push_back(e);
  dst->in_edges.push_back(e);
}

// counter(1) -> double(2) -> double(3); everything -> sink(4)
DagDef ChainDag(int32_t dag_id, int32_t batch, int32_t fail_on = -1) {
  static int32_t next_edge = 1000;  // edge ids are unique across queries, like python/gsl/dag_edge.py's counter
  DagDef def;
  def.id = dag_id;
  DagNodeDef root = Node(1, "TestCounter"), a = Node(2, "TestDouble"), b = Node(3, "TestDouble"), sink = Node(4, "Sink");
  root.params[kBatchSize] = IntParam(batch);
  if (fail_on >= 0) b.params["fail_on"] = IntParam(fail_on);
  Link(&root, &a, Edge(next_edge++, kNodeIds, kSrcIds));
  Link(&a, &b, Edge(next_edge++, kNodeIds, kSrcIds));
  Link(&root, &sink, Edge(next_edge++, "fake", "fake"));
  Link(&a, &sink, Edge(next_edge++, "fake", "fake"));
  Link(&b, &sink, Edge(next_edge++, "fake", "fake"));
  def.nodes = {root, a, b, sink};
  return def;
}

std::vector<int64_t> Values(GetDagValuesResponse* res, int32_t node, const char* key) {
  const Tensor* t = res->GetValue(node, key).first;
  std::vector<int64_t> out;
  for (int32_t i = 0; t && i < t->Size(); ++i) out.push_back(t->GetInt64(i));
  return out;
}

}  // namespace

TEST(Dag, WiresNodesAndEdgesFromTheDefinition) {
  Dag dag(ChainDag(1, 4));
  EXPECT_EQ(dag.Size(), 4);
  EXPECT_TRUE(dag.Root() != nullptr && dag.Root()->Id() == 1);
  EXPECT_TRUE(dag.Compile().ok());
  const DagNode* b = dag.Nodes()[2].get();
  EXPECT_EQ(b->InDegree(), 1);
  EXPECT_TRUE(b->InEdges()[0]->Src() == dag.Nodes()[1].get());
  EXPECT_TRUE(b->InEdges()[0]->Dst() == b);
  EXPECT_EQ(b->InEdges()[0]->SrcOutput(), std::string(kNodeIds));
  EXPECT_TRUE(dag.Nodes()[3]->IsSink());
  EXPECT_EQ(dag.Nodes()[3]->InDegree(), 3);
  // one step per node (the test operators are not hop-fusable), in dependency order, the sink last
  EXPECT_EQ(dag.Steps().size(), (size_t)4);
  EXPECT_EQ(dag.Steps()[0].nodes[0]->Id(), 1);
  EXPECT_EQ(dag.Steps()[1].nodes[0]->Id(), 2);
  EXPECT_EQ(dag.Steps()[2].nodes[0]->Id(), 3);
  EXPECT_TRUE(dag.Steps()[3].nodes[0]->IsSink());
  EXPECT_TRUE(dag.DebugString().find("op_name: \"TestDouble\"") != std::string::npos);
}

TEST(Dag, CompileRejectsBrokenDefinitions) {
  DagDef loose = ChainDag(2, 4);
  loose.nodes[2].in_edges.clear();  // the edge a -> b now has a source only (and b became a second root)
  Dag dangling(loose);
  EXPECT_TRUE(!dangling.Compile().ok());
  DagDef cyc = ChainDag(3, 4);
  DagEdgeDef back = Edge(5000, kNodeIds, kSrcIds);  // b -> a
  cyc.nodes[2].out_edges.push_back(back);
  cyc.nodes[1].in_edges.push_back(back);
  Dag cycle(cyc);
  EXPECT_TRUE(!cycle.Compile().ok());
  DagDef twice = ChainDag(4, 4);
  twice.nodes[2].id = twice.nodes[1].id;  // two nodes, one tape slot
  Dag dup(twice);
  EXPECT_TRUE(!dup.Compile().ok());
}

TEST(TapeStore, CloseIsStickyForConsumersThatLookLater) {
  // a consumer that wakes up after StopAll has finished (its transient flag is down again) must still
  // learn that the store ended, instead of waiting for a producer that no longer exists
  Dag dag(ChainDag(2, 4));
  EXPECT_TRUE(dag.Compile().ok());
  TapeStore store(2, &dag);
  const std::function<bool()> never = []() { return false; };
  Tape* t = store.New();
  EXPECT_TRUE(store.WaitAndPush(t, never));
  store.Close();
  Tape* got = store.WaitAndPop(0, never);  // what was queued is still handed out ...
  EXPECT_TRUE(got == t);
  delete got;
  EXPECT_TRUE(store.WaitAndPop(0, never) == nullptr);  // ... then the end, without a stop condition saying so
  EXPECT_TRUE(!store.WaitAndPush(store.New(), never));  // and nothing more goes in
}

// `.outV(e1).sample(3).by("random").outV(e2).sample(2).by("random")` with the lookup / degree nodes the Python layer
// adds: the two sampling hops become ONE step; a filtered hop, a different sampler or a second input break the chain.
TEST(Dag, FusesChainsOfDenseSamplingHops) {
  int32_t eid = 2000;
  auto sampler = [](int32_t id, const std::string& op, int32_t k) {
    DagNodeDef n = Node(id, op);
    n.params[kEdgeType] = StrParam("e");
    n.params[kStrategy] = StrParam(op);
    n.params[kNeighborCount] = IntParam(k);
    return n;
  };
  DagDef def;
  def.id = 4;
  DagNodeDef root = Node(1, "GetNodes"), look0 = Node(2, "LookupNodes"), hop1 = sampler(3, "RandomSampler", 3),
             deg = Node(4, "GetDegree"), look1 = Node(5, "LookupNodes"), hop2 = sampler(6, "RandomSampler", 2),
             look2 = Node(7, "LookupNodes"), hop3 = sampler(8, "TopkSampler", 2), sink = Node(9, "Sink");
  Link(&root, &look0, Edge(eid++, kNodeIds, kNodeIds));
  Link(&root, &hop1, Edge(eid++, kNodeIds, kSrcIds));
  Link(&root, &deg, Edge(eid++, kNodeIds, kNodeIds));
  Link(&hop1, &look1, Edge(eid++, kNodeIds, kNodeIds));
  Link(&hop1, &hop2, Edge(eid++, kNodeIds, kSrcIds));
  Link(&hop2, &look2, Edge(eid++, kNodeIds, kNodeIds));
  Link(&hop2, &hop3, Edge(eid++, kNodeIds, kSrcIds));
  for (DagNodeDef* n : {&root, &look0, &hop1, &deg, &look1, &hop2, &look2, &hop3}) Link(n, &sink, Edge(eid++, "fake", "fake"));
  def.nodes = {root, look0, hop1, deg, look1, hop2, look2, hop3, sink};
  Dag dag(def);
  EXPECT_TRUE(dag.Compile().ok());
  int fused_steps = 0;
  size_t nodes_in_steps = 0;
  std::vector<int32_t> position(10, -1);
  for (size_t s = 0; s < dag.Steps().size(); ++s) {
    const Dag::Step& step = dag.Steps()[s];
    nodes_in_steps += step.nodes.size();
    for (const DagNode* n : step.nodes) position[n->Id()] = (int32_t)s;
    if (step.nodes.size() > 1) {
      ++fused_steps;
      EXPECT_EQ(step.nodes.size(), (size_t)2);
      EXPECT_EQ(step.nodes[0]->Id(), 3);  // hop1 + hop2; hop3 is another sampler: its own step
      EXPECT_EQ(step.nodes[1]->Id(), 6);
    }
  }
  EXPECT_EQ(fused_steps, 1);
  EXPECT_EQ(nodes_in_steps, (size_t)9);
  for (const auto& n : dag.Nodes()) {  // every node after everything it reads
    for (const auto& e : n->InEdges()) EXPECT_TRUE(position[e->Src()->Id()] <= position[n->Id()]);
  }
  EXPECT_TRUE(position[9] == (int32_t)dag.Steps().size() - 1);

  // a filter on the second hop (`.filter('dst')`: a second in-edge + kFilterType) keeps the hops apart
  DagDef filtered = def;
  filtered.id = 5;
  filtered.nodes[5].params[kFilterType] = IntParam(1);
  Dag unfused(filtered);
  EXPECT_TRUE(unfused.Compile().ok());
  for (const Dag::Step& step : unfused.Steps()) EXPECT_EQ(step.nodes.size(), (size_t)1);
}

// A query runs round after round in the background; its consumer sees the root's batches in order, then one invalid
// response per epoch end, then the next epoch -- stamped with the epoch number.
TEST(DagScheduler, RoundsReachTheConsumerInOrderAndEpochsEndWithAnInvalidResponse) {
  g_counter_at = 0;
  SetGlobalFlagTapeCapacity(3);
  Client* client = NewInMemoryClient();
  DagRequest req;
  DagDef def = ChainDag(11, 4);
  req.ParseFrom(&def, true);
  EXPECT_TRUE(client->RunDag(&req).ok());
  EXPECT_TRUE(client->RunDag(&req).ok());  // a known id is not an error (executor.cc:55-58)
  for (int epoch = 0; epoch < 2; ++epoch) {
    const int64_t firsts[3] = {0, 4, 8};
    for (int round = 0; round < 3; ++round) {
      GetDagValuesRequest get(11);
      GetDagValuesResponse res;
      EXPECT_TRUE(client->GetDagValues(&get, &res).ok());
      EXPECT_TRUE(res.Valid());
      EXPECT_EQ(res.Epoch(), epoch);
      EXPECT_EQ(res.Index(), epoch * 4 + round);
      std::vector<int64_t> ids = Values(&res, 1, kNodeIds), twice = Values(&res, 2, kNodeIds), four = Values(&res, 3, kNodeIds);
      EXPECT_EQ(ids.size(), (size_t)(round == 2 ? 2 : 4));  // 10 ids: 4 + 4 + 2
      EXPECT_EQ(ids[0], firsts[round]);
      for (size_t i = 0; i < ids.size(); ++i) {
        EXPECT_EQ(twice[i], 2 * ids[i]);
        EXPECT_EQ(four[i], 4 * ids[i]);
      }
      EXPECT_TRUE(res.GetValue(4, kNodeIds).first == nullptr);  // the sink records nothing
      EXPECT_TRUE(res.GetValue(2, "no such key").first == nullptr);
    }
    GetDagValuesRequest get(11);
    GetDagValuesResponse end;
    EXPECT_TRUE(client->GetDagValues(&get, &end).ok());
    EXPECT_TRUE(!end.Valid());  // the faked tape: the root ran out
    EXPECT_EQ(end.Epoch(), epoch);
  }
  DagScheduler::StopAll();
  GetDagValuesRequest get(11);
  GetDagValuesResponse res;
  EXPECT_TRUE(!client->GetDagValues(&get, &res).ok());  // the query went with the stop
  delete client;
}

TEST(DagScheduler, AFailingNodeFakesTheWholeRound) {
  g_counter_at = 0;
  Client* client = NewInMemoryClient();
  DagRequest req;
  DagDef def = ChainDag(12, 4, /*fail_on=*/2 * 5);  // the second TestDouble sees 2 * id: fails in the round holding id 5
  req.ParseFrom(&def, false);
  EXPECT_TRUE(def.nodes.empty());  // moved out unless copy (dag_request.cc:35-43)
  EXPECT_TRUE(client->RunDag(&req).ok());
  GetDagValuesRequest get(12);
  GetDagValuesResponse r0, r1, r2;
  EXPECT_TRUE(client->GetDagValues(&get, &r0).ok() && r0.Valid());
  EXPECT_TRUE(client->GetDagValues(&get, &r1).ok() && !r1.Valid());  // ids 4..7: faked, nothing of it is kept
  EXPECT_TRUE(r1.records_.empty());
  EXPECT_TRUE(client->GetDagValues(&get, &r2).ok() && r2.Valid());
  EXPECT_EQ(Values(&r2, 1, kNodeIds)[0], 8);
  EXPECT_EQ(r2.Epoch(), 1);  // a faked tape counts as an epoch end, whatever faked it (tape.cc:107-110)
  DagScheduler::StopAll();
  delete client;
}

TEST(Dataset, PrefetchesInOrderAndHoldsBackTheNextEpoch) {
  g_counter_at = 0;
  SetGlobalFlagTapeCapacity(2);
  SetGlobalFlagDatasetCapacity(3);
  Client* client = NewInMemoryClient();
  DagRequest req;
  DagDef def = ChainDag(13, 5);
  req.ParseFrom(&def, true);
  EXPECT_TRUE(client->RunDag(&req).ok());
  {
    Dataset dataset(client, 13);
    for (int epoch = 0; epoch < 3; ++epoch) {
      for (int round = 0; round < 2; ++round) {
        std::unique_ptr<GetDagValuesResponse> res(dataset.Next(epoch));
        EXPECT_TRUE(res && res->Valid());
        EXPECT_EQ(Values(res.get(), 1, kNodeIds)[0], round * 5);
        EXPECT_EQ(res->Epoch(), epoch);
      }
      std::unique_ptr<GetDagValuesResponse> end(dataset.Next(epoch));
      EXPECT_TRUE(end && !end->Valid());
      // a caller that has not moved on to the next epoch does not get its data (dag_dataset.cc:78-83) ...
      EXPECT_TRUE(dataset.Next(epoch) == nullptr);
      EXPECT_TRUE(dataset.Next(epoch) == nullptr);
      // ... and loses nothing by asking: the next loop iteration reads round 0 of epoch + 1
    }
    // the query ran ahead of the consumer only by tape capacity + dataset capacity rounds (+ the one in flight)
    std::this_thread::sleep_for(std::chrono::milliseconds(300));  // let it fill up
    const int32_t before = g_double_calls.load();
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    EXPECT_EQ(g_double_calls.load(), before);
    dataset.Close();
    EXPECT_TRUE(dataset.Next(100) == nullptr);
  }
  DagScheduler::StopAll();
  delete client;
}

TEST(Dataset, ClosingWhileTheQueryIsStarvedDoesNotHang) {
  // a query whose root never produces: the dataset's prefetch blocks inside GetDagValues; Close() must get out
  Client* client = NewInMemoryClient();
  DagDef def;
  def.id = 14;
  DagNodeDef root = Node(1, "TestDouble"), sink = Node(2, "Sink");  // TestDouble without its input: every round is faked
  Link(&root, &sink, Edge(3000, "fake", "fake"));
  def.nodes = {root, sink};
  DagRequest req;
  req.ParseFrom(&def, true);
  EXPECT_TRUE(client->RunDag(&req).ok());
  Dataset* dataset = new Dataset(client, 14);
  std::unique_ptr<GetDagValuesResponse> res(dataset->Next(0));
  EXPECT_TRUE(res && !res->Valid());
  dataset->Close();
  delete dataset;
  // and a dataset on a query that does not exist ends at once instead of waiting forever
  Dataset orphan(client, 999);
  EXPECT_TRUE(orphan.Next(0) == nullptr);
  DagScheduler::StopAll();
  delete client;
}

// Queries read the store the operators serve NOW: stopping an older Server beside it leaves them running (ADVICE r05:
// Server::Stop of any one Server ended every query of the process); stopping the serving one ends them.
TEST(DagScheduler, OnlyTheServingServerTakesTheQueriesWithIt) {
  g_counter_at = 0;
  SetGlobalFlagTapeCapacity(2);
  Server* older = NewServer(0, 1, "", "");
  Server* serving = NewServer(0, 1, "", "");
  older->Init({}, {});
  serving->Init({}, {});  // the operators serve the most recently initialised store (client.cc)
  EXPECT_TRUE(older->InitStatus().ok() && serving->InitStatus().ok());
  Client* client = NewInMemoryClient();
  DagRequest req;
  DagDef def = ChainDag(15, 5);
  req.ParseFrom(&def, true);
  EXPECT_TRUE(client->RunDag(&req).ok());
  GetDagValuesRequest get(15);
  GetDagValuesResponse first;
  EXPECT_TRUE(client->GetDagValues(&get, &first).ok() && first.Valid());
  older->Stop();
  for (int i = 0; i < 6; ++i) {  // more rounds than the tape store held when the older server went
    GetDagValuesResponse more;
    EXPECT_TRUE(client->GetDagValues(&get, &more).ok());
  }
  serving->Stop();
  GetDagValuesResponse gone;
  EXPECT_TRUE(!client->GetDagValues(&get, &gone).ok());
  delete older;
  delete serving;
  delete client;
}

int main() { return RunAllTests(); }
