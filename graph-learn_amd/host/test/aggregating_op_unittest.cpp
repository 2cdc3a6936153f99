// Mirrors graphlearn/src/core/operator/aggregator/test/aggregating_op_unittest.cpp
// (:216-364): 100 nodes, node i has one float attribute = i; ids 0..9 in
// segments of sizes {0,1,2,3,4}; the expected embeddings are the reference's
// own EXPECT_FLOAT_EQ tables.
#include <thread>

#include "graphlearn/graphlearn.h"
#include "test_util.h"

using namespace graphlearn;      // NOLINT
using namespace graphlearn::op;  // NOLINT

namespace {
GraphStore* g_store = nullptr;

void SetUpStore() {
  if (g_store) return;
  io::SideInfo info;
  info.format = io::kAttributed;
  info.f_num = 1;
  info.type = "user";
  UpdateNodesRequest req(&info, 100);
  UpdateNodesResponse res;
  for (int i = 0; i < 100; ++i) {
    io::NodeValue v;
    v.id = i;
    v.attrs.push_back(static_cast<float>(i));
    req.Append(&v);
  }
  g_store = new GraphStore();
  Noder* noder = g_store->GetNoder("user");
  noder->UpdateNodes(&req, &res);
  IndexOption option;
  option.name = "sort";
  Status s = noder->Build(option);
  if (!s.ok()) {
    std::printf("noder build failed: %s\n", s.ToString().c_str());
    std::exit(2);
  }
  OpFactory::GetInstance()->Set(g_store);
}

void RunAgg(const char* name, const float* expect) {
  SetUpStore();
  AggregatingRequest* req = new AggregatingRequest("user", name);
  AggregatingResponse* res = new AggregatingResponse();
  int64_t node_ids[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
  int32_t segment_ids[10] = {1, 2, 2, 3, 3, 3, 4, 4, 4, 4};
  int32_t num_segments = 5;
  req->Set(node_ids, segment_ids, 10, num_segments);
  Operator* op = OpFactory::GetInstance()->Create(req->Name());
  EXPECT_TRUE(op != nullptr);
  Status s = op->Process(req, res);
  EXPECT_TRUE(s.ok());
  EXPECT_EQ(res->EmbeddingDim(), 1);
  EXPECT_EQ(res->NumSegments(), num_segments);
  const float* emb = res->Embeddings();
  const int32_t* segs = res->Segments();
  for (int32_t i = 0; i < num_segments; ++i) {
    EXPECT_FLOAT_EQ(emb[i], expect[i]);
    EXPECT_EQ(segs[i], i);
  }
  delete res;
  delete req;
}
}  // namespace

TEST(AggregatingOpTest, SumAggregator) {
  const float e[5] = {0, 0, 3, 12, 30};
  RunAgg("SumAggregator", e);
}
TEST(AggregatingOpTest, MeanAggregator) {
  const float e[5] = {0, 0, 1.5, 4, 7.5};
  RunAgg("MeanAggregator", e);
}
TEST(AggregatingOpTest, MinAggregator) {
  const float e[5] = {0, 0, 1, 3, 6};
  RunAgg("MinAggregator", e);
}
TEST(AggregatingOpTest, MaxAggregator) {
  const float e[5] = {0, 0, 2, 5, 9};
  RunAgg("MaxAggregator", e);
}
TEST(AggregatingOpTest, ProdAggregator) {
  const float e[5] = {0, 0, 2, 60, 3024};
  RunAgg("ProdAggregator", e);
}

TEST(AggregatingOpTest, ConcurrentProcessOnOneInstance) {
  // Process() is called concurrently from pool threads on the single operator
  // instance (in_memory_service.cc:64-71, op_factory.cc:45-61): must be re-entrant.
  SetUpStore();
  Operator* op = OpFactory::GetInstance()->Create("SumAggregator");
  std::vector<std::thread> pool;
  std::vector<int> ok(8, 0);
  for (int t = 0; t < 8; ++t) {
    pool.emplace_back([&, t]() {
      for (int rep = 0; rep < 20; ++rep) {
        AggregatingRequest req("user", "SumAggregator");
        AggregatingResponse res;
        int64_t ids[6] = {t, t + 1, t + 2, 50, 51, 99};
        int32_t seg[6] = {0, 0, 0, 1, 1, 2};
        req.Set(ids, seg, 6, 3);
        if (!op->Process(&req, &res).ok()) return;
        const float* e = res.Embeddings();
        if (e[0] != 3.0f * t + 3 || e[1] != 101.0f || e[2] != 99.0f) return;
      }
      ok[t] = 1;
    });
  }
  for (auto& th : pool) th.join();
  for (int t = 0; t < 8; ++t) EXPECT_TRUE(ok[t] == 1);
}

TEST(NodeLookuperTest, LookupNodesFloatAttributes) {
  // node_lookuper.cc:24-52 / local_noder.cc:85-97: attributes in request order,
  // unknown ids -> the default attribute (here 999.9 as in python/sampler/tests).
  SetUpStore();
  SetGlobalFlagDefaultFloatAttribute(999.9f);
  LookupNodesRequest req("user");
  LookupNodesResponse res;
  int64_t ids[6] = {7, 0, 99, 100, -3, 42};
  req.Set(ids, 6);
  Operator* op = OpFactory::GetInstance()->Create(req.Name());
  EXPECT_TRUE(op != nullptr);
  EXPECT_TRUE(op->Process(&req, &res).ok());
  EXPECT_EQ(res.Size(), 6);
  EXPECT_EQ(res.FloatAttrNum(), 1);
  const float expect[6] = {7.f, 0.f, 99.f, 999.9f, 999.9f, 42.f};
  for (int i = 0; i < 6; ++i) EXPECT_FLOAT_EQ(res.FloatAttrs()[i], expect[i]);
  SetGlobalFlagDefaultFloatAttribute(0.0f);
  OpRequest* rq = RequestFactory::GetInstance()->NewRequest("LookupNodes");
  EXPECT_TRUE(rq != nullptr);
  delete rq;
}

int main() { return RunAllTests(); }
