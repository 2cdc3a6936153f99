// Mirrors graphlearn/src/core/operator/sampler/test/sampler_unittest.cpp: same
// 5-edge fixture (:76-83), same request/operator plumbing (:96-110), same
// assertions -- membership + shape + default fill for the random samplers
// (:112-126,150-166,219-236) and the exact Topk answer {20,10,21,11} (:190-195)
// -- but the operators are the HIP-backed ones behind the same registry names.
#include <thread>
#include <unordered_set>
#include <vector>

#include "graphlearn/graphlearn.h"
#include "test_util.h"

using namespace graphlearn;      // NOLINT
using namespace graphlearn::op;  // NOLINT

namespace {
GraphStore* g_store = nullptr;

void SetUpStore() {
  if (g_store) return;
  io::SideInfo info_edge;
  info_edge.format = io::kWeighted;
  info_edge.type = "u-i";
  info_edge.src_type = "user";
  info_edge.dst_type = "item";
  UpdateEdgesRequest req_edge(&info_edge, 5);
  UpdateEdgesResponse res_edge;
  int64_t src_ids[5] = {0, 0, 0, 1, 1};
  int64_t dst_ids[5] = {10, 20, 30, 11, 21};
  float weights[5] = {0.8f, 1.0f, 0.5f, 0.88f, 1.2f};
  for (int i = 0; i < 5; ++i) {
    io::EdgeValue v;
    v.src_id = src_ids[i];
    v.dst_id = dst_ids[i];
    v.weight = weights[i];
    req_edge.Append(&v);
  }
  g_store = new GraphStore();
  Graph* graph = g_store->GetGraph("u-i");
  graph->UpdateEdges(&req_edge, &res_edge);
  IndexOption option;
  option.name = "sort";
  Status s = graph->Build(option);
  if (!s.ok()) {
    std::printf("graph build failed: %s\n", s.ToString().c_str());
    std::exit(2);
  }
  OpFactory::GetInstance()->Set(g_store);
}

void CheckMembershipAndDefault(const char* strategy) {
  SetUpStore();
  int32_t nbr_count = 2;
  SamplingRequest* req = new SamplingRequest("u-i", strategy, nbr_count);
  SamplingResponse* res = new SamplingResponse();
  // 1 has neighbors {11, 21}, 2 has no neighbors
  int32_t batch_size = 2;
  int64_t ids[2] = {1, 2};
  req->Set(ids, batch_size);

  Operator* op = OpFactory::GetInstance()->Create(req->Name());
  EXPECT_TRUE(op != nullptr);
  Status s = op->Process(req, res);
  EXPECT_TRUE(s.ok());
  EXPECT_EQ(res->GetShape().dim1, (size_t)batch_size);
  EXPECT_EQ(res->GetShape().dim2, (size_t)nbr_count);

  std::unordered_set<int64_t> nbr_set({11, 21});
  const int64_t* neighbor_ids = res->GetNeighborIds();
  for (int32_t i = 0; i < nbr_count; ++i) {
    EXPECT_TRUE(nbr_set.find(neighbor_ids[i]) != nbr_set.end());
  }
  // neighbors of 2: filled with the default id
  for (int32_t i = nbr_count; i < batch_size * nbr_count; ++i) {
    EXPECT_TRUE(neighbor_ids[i] == 0);
  }
  std::unordered_set<int64_t> edge_set({3, 4});
  const int64_t* edge_ids = res->GetEdgeIds();
  for (int32_t i = 0; i < nbr_count; ++i) {
    EXPECT_TRUE(edge_set.find(edge_ids[i]) != edge_set.end());
  }
  for (int32_t i = nbr_count; i < batch_size * nbr_count; ++i) {
    EXPECT_TRUE(edge_ids[i] == -1);
  }
  delete res;
  delete req;
}
}  // namespace

TEST(SamplerTest, Random) { CheckMembershipAndDefault("RandomSampler"); }
TEST(SamplerTest, RandomWithoutReplacement) { CheckMembershipAndDefault("RandomWithoutReplacementSampler"); }
TEST(SamplerTest, EdgeWeight) { CheckMembershipAndDefault("EdgeWeightSampler"); }
TEST(SamplerTest, InDegree) { CheckMembershipAndDefault("InDegreeSampler"); }  // sampler_unittest.cpp:237-273

TEST(SamplerTest, Full) {
  // sampler_unittest.cpp:275-313: ids {0,1,2}, count 2 -> truncated to 2 per row,
  // sparse shape; rows keep the storage (weight-descending) order.
  SetUpStore();
  SamplingRequest req("u-i", "FullSampler", 2);
  SamplingResponse res;
  int64_t ids[3] = {0, 1, 2};
  req.Set(ids, 3);
  Operator* op = OpFactory::GetInstance()->Create(req.Name());
  EXPECT_TRUE(op != nullptr);
  EXPECT_TRUE(op->Process(&req, &res).ok());
  EXPECT_EQ(res.GetShape().dim1, (size_t)3);
  EXPECT_EQ(res.GetShape().dim2, (size_t)2);
  EXPECT_TRUE(res.GetShape().sparse);
  EXPECT_EQ(res.GetShape().size, (size_t)4);
  int32_t seg[3] = {2, 2, 0};
  for (int i = 0; i < 3; ++i) EXPECT_EQ(res.GetShape().segments[i], seg[i]);
  int64_t nbr[4] = {20, 10, 21, 11};
  for (int i = 0; i < 4; ++i) EXPECT_EQ(res.GetNeighborIds()[i], nbr[i]);
  // neighbor_count 0 = no limit
  SamplingRequest req2("u-i", "FullSampler", 0);
  SamplingResponse res2;
  req2.Set(ids, 3);
  EXPECT_TRUE(op->Process(&req2, &res2).ok());
  EXPECT_EQ(res2.GetShape().size, (size_t)5);
  int64_t all[5] = {20, 10, 30, 21, 11};
  for (int i = 0; i < 5; ++i) EXPECT_EQ(res2.GetNeighborIds()[i], all[i]);
}

TEST(SamplerTest, NodeWeightNegative) {
  // sampler_unittest.cpp:315-347 (disabled there: its fixture never loads the "user" nodes).
  // Nodes {0..4} with weights; the request's own ids {0, 1} are never returned.
  SetUpStore();
  static bool nodes_loaded = false;
  if (!nodes_loaded) {
    io::SideInfo info;
    info.format = io::kWeighted;
    info.type = "user";
    UpdateNodesRequest req(&info, 5);
    for (int i = 0; i < 5; ++i) {
      io::NodeValue v;
      v.id = i;
      v.weight = 0.5f + i;
      req.Append(&v);
    }
    UpdateNodesResponse res;
    g_store->GetNoder("user")->UpdateNodes(&req, &res);
    nodes_loaded = true;
  }
  const int32_t nbr_count = 2, batch_size = 2;
  SamplingRequest req("user", "NodeWeightNegativeSampler", nbr_count);
  SamplingResponse res;
  int64_t ids[2] = {0, 1};
  req.Set(ids, batch_size);
  Operator* op = OpFactory::GetInstance()->Create(req.Name());
  EXPECT_TRUE(op != nullptr);
  for (int round = 0; round < 20; ++round) {
    SamplingResponse r;
    EXPECT_TRUE(op->Process(&req, &r).ok());
    EXPECT_EQ(r.GetShape().dim1, (size_t)batch_size);
    EXPECT_EQ(r.GetShape().dim2, (size_t)nbr_count);
    for (int i = 0; i < batch_size * nbr_count; ++i) {
      const int64_t v = r.GetNeighborIds()[i];
      EXPECT_TRUE(v >= 2 && v <= 4);
    }
  }
}

TEST(SamplerTest, EdgeTypeNegatives) {
  // Random / SoftInDegree: any destination id of "u-i" (src 0 -> {10, 20, 30}, src 1 -> {11, 21}).
  // InDegree follows in_degree_negative_sampler.cc:57-92 exactly: with the same pinned random
  // stream, its answer is the Soft sampler's candidate stream taken in blocks of `count`,
  // neighbours of the source dropped in the first three blocks, nothing dropped in the fourth.
  SetUpStore();
  const int32_t count = 4;
  int64_t ids[3] = {0, 1, 99};
  for (const char* name : {"RandomNegativeSampler", "SoftInDegreeNegativeSampler"}) {
    SamplingRequest req("u-i", name, count);
    req.Set(ids, 3);
    Operator* op = OpFactory::GetInstance()->Create(req.Name());
    EXPECT_TRUE(op != nullptr);
    SamplingResponse res;
    EXPECT_TRUE(op->Process(&req, &res).ok());
    for (int i = 0; i < 3 * count; ++i) {
      const int64_t v = res.GetNeighborIds()[i];
      EXPECT_TRUE(v == 10 || v == 20 || v == 30 || v == 11 || v == 21);
    }
  }
  for (int64_t cc = 100; cc < 130; ++cc) {
    SamplingRequest soft("u-i", "SoftInDegreeNegativeSampler", 4 * count);
    soft.Set(ids, 3);
    soft.SetCallCounter(cc);
    SamplingResponse stream;
    EXPECT_TRUE(OpFactory::GetInstance()->Create(soft.Name())->Process(&soft, &stream).ok());
    SamplingRequest strict("u-i", "InDegreeNegativeSampler", count);
    strict.Set(ids, 3);
    strict.SetCallCounter(cc);
    SamplingResponse res;
    EXPECT_TRUE(OpFactory::GetInstance()->Create(strict.Name())->Process(&strict, &res).ok());
    for (int row = 0; row < 3; ++row) {
      std::vector<int64_t> want;
      for (int d = 0; d < 4 * count && (int)want.size() < count; ++d) {
        const int64_t v = stream.GetNeighborIds()[row * 4 * count + d];
        const bool neighbour = row == 0 ? (v == 10 || v == 20 || v == 30) : row == 1 ? (v == 11 || v == 21) : false;
        if (d >= 3 * count || !neighbour) want.push_back(v);
      }
      for (int j = 0; j < count; ++j) EXPECT_EQ(res.GetNeighborIds()[row * count + j], want[j]);
    }
  }
}

TEST(SamplerTest, Topk) {
  SetUpStore();
  int32_t nbr_count = 2;
  SamplingRequest* req = new SamplingRequest("u-i", "TopkSampler", nbr_count);
  SamplingResponse* res = new SamplingResponse();
  int32_t batch_size = 2;
  int64_t ids[2] = {0, 1};
  req->Set(ids, batch_size);
  Operator* op = OpFactory::GetInstance()->Create(req->Name());
  EXPECT_TRUE(op != nullptr);
  Status s = op->Process(req, res);
  EXPECT_TRUE(s.ok());
  EXPECT_EQ(res->GetShape().dim1, (size_t)batch_size);
  EXPECT_EQ(res->GetShape().dim2, (size_t)nbr_count);
  const int64_t* neighbor_ids = res->GetNeighborIds();
  int64_t result[4] = {20, 10, 21, 11};  // sampler_unittest.cpp:190
  for (int32_t i = 0; i < batch_size * nbr_count; ++i) EXPECT_EQ(neighbor_ids[i], result[i]);
  delete res;
  delete req;
}

TEST(SamplerTest, PaddingModes) {
  // circular (config.cc:94 default) repeats the row, replicate default-fills
  // (circular_padder.h:46-63, replicate_padder.h:37-56); python
  // test_topk_neighbor_sampling.py pins both.
  SetUpStore();
  int64_t ids[2] = {1, 0};
  for (int mode = 0; mode < 2; ++mode) {
    SetGlobalFlagPaddingMode(mode);
    SetGlobalFlagDefaultNeighborId(-1);
    SamplingRequest req("u-i", "TopkSampler", 5);
    SamplingResponse res;
    req.Set(ids, 2);
    Status s = OpFactory::GetInstance()->Create("TopkSampler")->Process(&req, &res);
    EXPECT_TRUE(s.ok());
    const int64_t* n = res.GetNeighborIds();
    int64_t circ[10] = {21, 11, 21, 11, 21, 20, 10, 30, 20, 10};
    int64_t repl[10] = {21, 11, -1, -1, -1, 20, 10, 30, -1, -1};
    for (int i = 0; i < 10; ++i) EXPECT_EQ(n[i], mode == kCircular ? circ[i] : repl[i]);
  }
  SetGlobalFlagPaddingMode(kCircular);
  SetGlobalFlagDefaultNeighborId(0);
}

TEST(SamplerTest, SeedingContractIsReproducible) {
  SetUpStore();
  int64_t ids[64];
  for (int i = 0; i < 64; ++i) ids[i] = i % 2;
  // the op's call counter advances per Process() like the reference's RNG state:
  // consecutive calls differ, a fresh process with the same seed repeats (checked
  // through the C-ABI in tests/test_gpu_parity.py); here: valid + varying.
  SamplingRequest r1("u-i", "RandomSampler", 8), r2("u-i", "RandomSampler", 8);
  SamplingResponse s1, s2;
  r1.Set(ids, 64);
  r2.Set(ids, 64);
  Operator* op = OpFactory::GetInstance()->Create("RandomSampler");
  EXPECT_TRUE(op->Process(&r1, &s1).ok());
  EXPECT_TRUE(op->Process(&r2, &s2).ok());
  bool differ = false;
  for (int i = 0; i < 64 * 8; ++i) differ |= s1.GetNeighborIds()[i] != s2.GetNeighborIds()[i];
  EXPECT_TRUE(differ);
}

TEST(SamplerTest, ErrorConventions) {
  SetUpStore();
  // unknown op name -> nullptr from the factory (executor.cc:37-40 turns it into InvalidArgument)
  EXPECT_TRUE(OpFactory::GetInstance()->Create("NoSuchSampler") == nullptr);
  // a request type registered under the same name as the operator (op_request.h:138-152)
  OpRequest* rq = RequestFactory::GetInstance()->NewRequest("TopkSampler");
  OpResponse* rs = RequestFactory::GetInstance()->NewResponse("TopkSampler");
  EXPECT_TRUE(rq != nullptr && rs != nullptr);
  delete rq;
  delete rs;
  int64_t ids[1] = {0};
  // a filter without one value per src id is an InvalidArgument, not a silent no-filter run
  SamplingRequest req("u-i", "RandomSampler", 2, kEqual, kId);
  SamplingResponse res;
  req.Set(ids, 1);
  Status s = OpFactory::GetInstance()->Create("RandomSampler")->Process(&req, &res);
  EXPECT_TRUE(error::IsInvalidArgument(s));
  // an edge type that was never loaded: all rows unknown -> default fill
  SamplingRequest req2("nobody", "TopkSampler", 3);
  SamplingResponse res2;
  req2.Set(ids, 1);
  EXPECT_TRUE(OpFactory::GetInstance()->Create("TopkSampler")->Process(&req2, &res2).ok());
  for (int i = 0; i < 3; ++i) {
    EXPECT_EQ(res2.GetNeighborIds()[i], 0);
    EXPECT_EQ(res2.GetEdgeIds()[i], -1);
  }
}

// op::Filter (sampler/filter.h:30-125) on the 5-edge fixture; rows after Build: 0 -> {20, 10, 30}, 1 -> {21, 11}.
TEST(SamplerTest, Filters) {
  SetUpStore();
  {  // Topk, id == value: ActOn refills the hole from the right end (filter.cc:83-94)
    SamplingRequest req("u-i", "TopkSampler", 3, kEqual, kId);
    SamplingResponse res;
    int64_t ids[2] = {0, 1}, values[2] = {20, 11};
    req.Set(ids, 2);
    req.SetFilterValues(values, 2);
    EXPECT_TRUE(OpFactory::GetInstance()->Create("TopkSampler")->Process(&req, &res).ok());
    const int64_t want[6] = {30, 10, 30, 21, 21, 21};
    for (int i = 0; i < 6; ++i) EXPECT_EQ(res.GetNeighborIds()[i], want[i]);
  }
  {  // values through the tensor map expand like Filter::FillValues (filter.cc:53-67)
    SamplingRequest req("u-i", "TopkSampler", 2, kEqual, kId);
    SamplingResponse res;
    Tensor::Map tensors;
    ADD_TENSOR(tensors, kSrcIds, kInt64, 4);
    ADD_TENSOR(tensors, kFilterValues, kInt64, 2);
    int64_t ids[4] = {0, 0, 1, 1}, values[2] = {10, 21};
    tensors[kSrcIds].AddInt64(ids, ids + 4);
    tensors[kFilterValues].AddInt64(values, values + 2);
    req.Set(tensors);
    EXPECT_TRUE(OpFactory::GetInstance()->Create("TopkSampler")->Process(&req, &res).ok());
    const int64_t want[8] = {20, 30, 20, 30, 11, 11, 11, 11};
    for (int i = 0; i < 8; ++i) EXPECT_EQ(res.GetNeighborIds()[i], want[i]);
  }
  {  // RandomSampler: hits are redrawn; a row whose neighbours all hit is default-filled
    SetGlobalFlagSamplingRetryTimes(60);
    SamplingRequest req("u-i", "RandomSampler", 16, kLargerThan, kId);
    SamplingResponse res;
    int64_t ids[2] = {0, 1}, values[2] = {15, 5};
    req.Set(ids, 2);
    req.SetFilterValues(values, 2);
    EXPECT_TRUE(OpFactory::GetInstance()->Create("RandomSampler")->Process(&req, &res).ok());
    for (int i = 0; i < 16; ++i) EXPECT_EQ(res.GetNeighborIds()[i], 10);
    for (int i = 16; i < 32; ++i) EXPECT_EQ(res.GetNeighborIds()[i], 0);
    SetGlobalFlagSamplingRetryTimes(5);
  }
  for (const char* name : {"RandomWithoutReplacementSampler", "EdgeWeightSampler", "InDegreeSampler"}) {
    SamplingRequest req("u-i", name, 4, kEqual, kId);
    SamplingResponse res;
    int64_t ids[1] = {0}, values[1] = {10};
    req.Set(ids, 1);
    req.SetFilterValues(values, 1);
    EXPECT_TRUE(OpFactory::GetInstance()->Create(name)->Process(&req, &res).ok());
    for (int i = 0; i < 4; ++i) EXPECT_TRUE(res.GetNeighborIds()[i] == 20 || res.GetNeighborIds()[i] == 30);
  }
  {  // FullSampler keeps the unfiltered segment sizes and pads the survivors (full_sampler.cc:55-84)
    SamplingRequest req("u-i", "FullSampler", 0, kEqual, kId);
    SamplingResponse res;
    int64_t ids[2] = {0, 9}, values[2] = {10, 10};
    req.Set(ids, 2);
    req.SetFilterValues(values, 2);
    EXPECT_TRUE(OpFactory::GetInstance()->Create("FullSampler")->Process(&req, &res).ok());
    EXPECT_EQ(res.GetShape().segments[0], 3);
    EXPECT_EQ(res.GetShape().segments[1], 0);
    const int64_t want[3] = {20, 30, 20};
    for (int i = 0; i < 3; ++i) EXPECT_EQ(res.GetNeighborIds()[i], want[i]);
  }
}

TEST(SamplerTest, ConcurrentProcessOnOneInstance) {
  // up to 32 pool threads call Process() on the single operator instance
  // (in_memory_service.cc:64-71): results must stay valid per request.
  SetUpStore();
  Operator* op = OpFactory::GetInstance()->Create("TopkSampler");
  std::vector<std::thread> pool;
  std::vector<int> ok(8, 0);
  for (int t = 0; t < 8; ++t) {
    pool.emplace_back([&, t]() {
      for (int rep = 0; rep < 25; ++rep) {
        SamplingRequest req("u-i", "TopkSampler", 3);
        SamplingResponse res;
        int64_t ids[3] = {t % 2, 1 - t % 2, 7};
        req.Set(ids, 3);
        if (!op->Process(&req, &res).ok()) return;
        const int64_t* n = res.GetNeighborIds();
        const int64_t r0[3] = {20, 10, 30}, r1[3] = {21, 11, 21};
        for (int j = 0; j < 3; ++j) {
          if (n[j] != (t % 2 == 0 ? r0[j] : r1[j])) return;
          if (n[3 + j] != (t % 2 == 0 ? r1[j] : r0[j])) return;
          if (n[6 + j] != 0) return;
        }
      }
      ok[t] = 1;
    });
  }
  for (auto& th : pool) th.join();
  for (int t = 0; t < 8; ++t) EXPECT_TRUE(ok[t] == 1);
}

int main() { return RunAllTests(); }
