// Host-only behaviour of the request classes (no device): what the reference's
// service/request/*.cc do before an operator ever runs -- filter parameters and
// Filter::FillValues (sampling_request.cc:33-136, filter.cc:53-67), partitioning of a
// request's tensors (hash_partitioner.h:33-92), RandomWalkRequest::IsDeepWalk
// (random_walk_request.cc:152-160), Clone() of every request kind.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "glx.h"
#include "graphlearn/graphlearn.h"
#include "test_util.h"

using namespace graphlearn;  // NOLINT

TEST(RequestTest, FilterParametersAndValues) {
  SamplingRequest plain("e", "RandomSampler", 3);
  EXPECT_TRUE(!plain.HasFilter());
  EXPECT_TRUE(plain.GetFilterValues() == nullptr);
  int64_t ids[4] = {7, 8, 9, 10};
  plain.Set(ids, 4);
  int64_t v[4] = {1, 2, 3, 4};
  plain.SetFilterValues(v, 4);  // no filter: ignored, like Filter::FillValues on an unspecified type
  EXPECT_TRUE(plain.GetFilterValues() == nullptr);

  // the reference's enum values (include/constants.h:135-145) are the C-ABI's
  EXPECT_EQ((int)kEqual, GLX_FILTER_EQUAL);
  EXPECT_EQ((int)kLargerThan, GLX_FILTER_LARGER_THAN);
  EXPECT_EQ((int)kId, GLX_FILTER_FIELD_ID);
  EXPECT_EQ((int)kTimestamp, GLX_FILTER_FIELD_TIMESTAMP);

  SamplingRequest req("e", "TopkSampler", 2, kLargerThan, kTimestamp);
  EXPECT_TRUE(req.HasFilter());
  EXPECT_EQ(req.GetFilterType(), kLargerThan);
  EXPECT_EQ(req.GetFilterField(), kTimestamp);
  req.Set(ids, 4);
  EXPECT_TRUE(req.GetFilterValues() == nullptr);  // not one value per src id yet
  req.SetFilterValues(v, 2);
  EXPECT_TRUE(req.GetFilterValues() == nullptr);
  req.SetFilterValues(v + 2, 2);
  EXPECT_TRUE(req.GetFilterValues() != nullptr);
  for (int i = 0; i < 4; ++i) EXPECT_EQ(req.GetFilterValues()[i], v[i]);

  // Clone keeps the filter kind (sampling_request.cc:68-72), not the data
  OpRequest* c = req.Clone();
  SamplingRequest* cs = static_cast<SamplingRequest*>(c);
  EXPECT_TRUE(cs->HasFilter() && cs->GetFilterType() == kLargerThan && cs->GetFilterField() == kTimestamp);
  EXPECT_EQ(cs->BatchSize(), 0);
  delete c;
}

TEST(RequestTest, FillValuesExpandsByFanout) {
  // DAG hand-over: 6 src ids that descend from 2 upstream ids -> every value covers 3 rows
  SamplingRequest req("e", "RandomSampler", 2, kEqual, kId);
  Tensor::Map tensors;
  ADD_TENSOR(tensors, kSrcIds, kInt64, 6);
  ADD_TENSOR(tensors, kFilterValues, kInt64, 2);
  int64_t ids[6] = {1, 2, 3, 4, 5, 6}, values[2] = {100, 200};
  tensors[kSrcIds].AddInt64(ids, ids + 6);
  tensors[kFilterValues].AddInt64(values, values + 2);
  req.Set(tensors);
  EXPECT_EQ(req.BatchSize(), 6);
  EXPECT_TRUE(req.GetFilterValues() != nullptr);
  for (int i = 0; i < 6; ++i) EXPECT_EQ(req.GetFilterValues()[i], i < 3 ? 100 : 200);

  // Init from a parameter map (DagNodeRunner path, sampling_request.cc:87-124)
  Tensor::Map params;
  ADD_TENSOR(params, kEdgeType, kString, 1);
  params[kEdgeType].AddString("e");
  ADD_TENSOR(params, kStrategy, kString, 1);
  params[kStrategy].AddString("TopkSampler");
  ADD_TENSOR(params, kNeighborCount, kInt32, 1);
  params[kNeighborCount].AddInt32(5);
  ADD_TENSOR(params, kFilterType, kInt32, 1);
  params[kFilterType].AddInt32(kEqual);
  ADD_TENSOR(params, kFilterField, kInt32, 1);
  params[kFilterField].AddInt32(kId);
  SamplingRequest from_params;
  from_params.Init(params);
  EXPECT_TRUE(from_params.HasFilter());
  EXPECT_EQ(from_params.NeighborCount(), 5);
  EXPECT_TRUE(from_params.Strategy() == "TopkSampler");
  from_params.Set(tensors);
  EXPECT_EQ(from_params.GetFilterValues()[5], 200);
}

TEST(RequestTest, PartitionCarriesFilterValues) {
  SamplingRequest req("e", "EdgeWeightSampler", 4, kEqual, kId);
  std::vector<int64_t> ids, values;
  for (int i = 0; i < 50; ++i) {
    ids.push_back(i * 3 - 20);
    values.push_back(1000 + i);
  }
  req.Set(ids.data(), 50);
  req.SetFilterValues(values.data(), 50);
  req.SetCallCounter(77);
  HashPartitioner partitioner(3);
  ShardsPtr<OpRequest> parts = partitioner.Partition(&req);
  int seen = 0;
  for (int s = 0; s < 3; ++s) {
    const SamplingRequest* p = static_cast<const SamplingRequest*>(parts->Get(s));
    if (!p) continue;
    EXPECT_TRUE(p->HasFilter() && p->HasCallCounter() && p->CallCounter() == 77);
    EXPECT_TRUE(p->GetFilterValues() != nullptr && p->GetRngRows() != nullptr);
    for (int32_t i = 0; i < p->BatchSize(); ++i) {
      const int64_t row = p->GetRngRows()[i];
      EXPECT_EQ(p->GetSrcIds()[i], ids[(size_t)row]);
      EXPECT_EQ(p->GetFilterValues()[i], values[(size_t)row]);
      EXPECT_EQ((int)(std::llabs(p->GetSrcIds()[i]) % 3), s);
      ++seen;
    }
  }
  EXPECT_EQ(seen, 50);
}

TEST(RequestTest, RandomWalkRequest) {
  RandomWalkRequest deep("e", 1.0f, 1.0f, 5);
  EXPECT_TRUE(deep.IsDeepWalk());
  EXPECT_EQ(deep.WalkLen(), 5);
  RandomWalkRequest nearly("e", 1.0f + 16 * 1.1920929e-07f, 1.0f, 5);
  EXPECT_TRUE(nearly.IsDeepWalk());  // within 32 FLT_EPSILON
  RandomWalkRequest biased("e", 0.5f, 1.0f, 5);
  EXPECT_TRUE(!biased.IsDeepWalk());
  RandomWalkRequest biased_q("e", 1.0f, 2.0f);
  EXPECT_TRUE(!biased_q.IsDeepWalk());
  EXPECT_EQ(biased_q.WalkLen(), 1);
  int64_t ids[2] = {4, 5};
  biased.Set(ids, 2);
  biased.SetCallCounter(9);
  OpRequest* c = biased.Clone();
  RandomWalkRequest* cw = static_cast<RandomWalkRequest*>(c);
  EXPECT_TRUE(cw->P() == 0.5f && cw->Q() == 1.0f && cw->WalkLen() == 5 && cw->CallCounter() == 9);
  EXPECT_TRUE(cw->Type() == "e" && cw->Name() == "RandomWalk");
  delete c;
  // registered under the operator's name (op_request.h:138-152)
  OpRequest* rq = RequestFactory::GetInstance()->NewRequest("RandomWalk");
  OpResponse* rs = RequestFactory::GetInstance()->NewResponse("RandomWalk");
  EXPECT_TRUE(rq != nullptr && rs != nullptr);
  delete rq;
  delete rs;
  EXPECT_TRUE(op::OpFactory::GetInstance()->Create("RandomWalk") != nullptr);
}

// Bootstrap of the shard communicator through the tracker directory (no device involved): what server 0
// publishes is what the other servers read, whichever starts first; a session nobody published times out.
TEST(RequestTest, UniqueIdTravelsThroughTheTrackerDirectory) {
  char tmpl[] = "/tmp/glx_tracker_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  EXPECT_TRUE(dir != nullptr);
  std::string got0(GLX_UNIQUE_ID_BYTES, '\0'), got1;  // server 0 publishes these bytes (no RCCL on a CPU box)
  for (size_t i = 0; i < got0.size(); ++i) got0[i] = (char)(i * 7 + 3);
  Status s1;
  std::thread reader([&] { s1 = ExchangeUniqueId(dir, "job42", 1, 20.0, &got1); });  // starts polling first
  std::this_thread::sleep_for(std::chrono::milliseconds(100));
  Status s0 = ExchangeUniqueId(dir, "job42", 0, 20.0, &got0);
  reader.join();
  EXPECT_TRUE(s0.ok() && s1.ok());
  EXPECT_EQ(got0.size(), (size_t)GLX_UNIQUE_ID_BYTES);
  EXPECT_TRUE(got0 == got1);
  bool nonzero = false;
  for (char c : got0) nonzero |= c != 0;
  EXPECT_TRUE(nonzero);
  std::string late;
  EXPECT_TRUE(ExchangeUniqueId(dir, "job42", 3, 1.0, &late).ok() && late == got0);  // a late joiner
  std::string other;
  Status missing = ExchangeUniqueId(dir, "another-session", 1, 0.2, &other);
  EXPECT_TRUE(error::IsUnavailable(missing));
  EXPECT_TRUE(error::IsInvalidArgument(ExchangeUniqueId(dir, "a/b", 0, 1.0, &other)));
}

// Large tensors sit in the response pool's blocks: anonymous mappings of their own (2 MiB granules), never ranges of the
// malloc heap -- registered heap ranges beside MADV_HUGEPAGE heap memory make later pageable copies fault on ROCm 7.0
// (base.cc, scripts/r06/repro/hostreg_pageable.hip) -- and a released block is parked and handed out again.
TEST(TensorPool, LargeTensorsLiveOutsideTheHeapAndAreRecycled) {
  auto in_heap = [](const void* p) {  // inside the [heap] line of /proc/self/maps?
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return false;
    char line[512];
    bool hit = false;
    while (fgets(line, sizeof(line), f)) {
      unsigned long lo = 0, hi = 0;
      if (sscanf(line, "%lx-%lx", &lo, &hi) == 2 && strstr(line, "[heap]") &&
          reinterpret_cast<uintptr_t>(p) >= lo && reinterpret_cast<uintptr_t>(p) < hi) hit = true;
    }
    fclose(f);
    return hit;
  };
  std::vector<void*> keep_heap_busy;  // make sure the heap exists and has room: a malloc of this size WOULD land there
  for (int i = 0; i < 8; ++i) keep_heap_busy.push_back(malloc(100 << 10));
  const int64_t* first = nullptr;
  for (int32_t n : {40000, 300000, 1500000}) {  // 320 KB (a slab piece), 2.4 MB and 12 MB (mappings of their own)
    Tensor t(kInt64, 0);
    t.Resize(n);
    int64_t* p = t.MutableInt64();
    EXPECT_TRUE(p != nullptr && reinterpret_cast<uintptr_t>(p) % 4096 == 0);
    EXPECT_TRUE(!in_heap(p));
    p[0] = 1;
    p[n - 1] = 2;  // the whole block is ours
    if (n == 40000) first = p;
  }
  Tensor again(kInt64, 0);
  again.Resize(40000);
  EXPECT_TRUE(again.MutableInt64() == first);  // the parked block comes back
  Tensor small(kInt64, 0);
  small.Resize(100);  // below the pool's threshold: the ordinary heap
  EXPECT_TRUE(small.MutableInt64() != nullptr);
  for (void* p : keep_heap_busy) free(p);
}

int main() { return RunAllTests(); }
