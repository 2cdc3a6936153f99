// DistributeRunner over the device-resident distributed store.  Replaces
// graphlearn/src/core/runner/op_runner.h:60-152 (Partition -> RunInParallel -> Stitch) and
// op_runner.cc (GetOpRunner): the sub-requests never exist as host objects -- the C-ABI
// call buckets the request rows by owner on the GPU, exchanges them over the shard
// communicator, runs the owner's kernels, exchanges the results back and stitches.
#include <functional>

#include "graphlearn/op_runner.h"
#include "graphlearn/subgraph_request.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "glx.h"
#include "graphlearn/aggregating_request.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/config.h"
#include "graphlearn/sampling_request.h"

namespace graphlearn {

Status ExchangeUniqueId(const std::string& tracker, const std::string& session, int32_t server_id,
                        double timeout_seconds, std::string* id) {
  if (tracker.empty() || session.empty() || session.find('/') != std::string::npos) {
    return error::InvalidArgument("ExchangeUniqueId needs a tracker directory and a session name without '/'");
  }
  const std::string dir = tracker + (tracker.back() == '/' ? "" : "/") + "glx_comm";
  const std::string path = dir + "/" + session;
  const bool preset = server_id == 0 && id->size() == (size_t)GLX_UNIQUE_ID_BYTES;
  if (!preset) id->assign((size_t)GLX_UNIQUE_ID_BYTES, '\0');
  if (server_id == 0) {
    if (!preset) {
      int rc = glx_comm_unique_id(&(*id)[0]);
      if (rc != GLX_OK) return error::FromGlx(rc);
    }
    if (mkdir(dir.c_str(), 0777) != 0 && errno != EEXIST) {
      return error::InvalidArgument("cannot create " + dir + " (tracker path)");
    }
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return error::InvalidArgument("cannot write " + tmp);
    const bool ok = fwrite(id->data(), 1, id->size(), f) == id->size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) {
      (void)remove(tmp.c_str());
      return error::Internal("cannot publish " + path);
    }
    return Status::OK();
  }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_seconds);
  while (true) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const size_t n = fread(&(*id)[0], 1, id->size(), f);
      fclose(f);
      if (n == id->size()) return Status::OK();
    }
    if (std::chrono::steady_clock::now() > deadline) {
      return error::Unavailable("server 0 did not publish " + path + " within the timeout");
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

Status ConnectServers(const std::string& tracker, const std::string& session, int device, int32_t server_id,
                      int32_t server_count, glx_comm** comm) {
  *comm = nullptr;
  std::string id;
  Status s = ExchangeUniqueId(tracker, session, server_id, 300.0, &id);
  if (!s.ok()) return s;
  return error::FromGlx(glx_comm_init_rccl(device, server_id, server_count, id.data(), comm));
}

Env::Env(glx_comm* comm, GraphStore* store) : comm_(comm), store_(store), server_id_(0), server_count_(1) {
  int rank = 0, world = 1;
  if (comm && glx_comm_info(comm, &rank, &world, nullptr, nullptr) == GLX_OK) {
    server_id_ = rank;
    server_count_ = world;
  }
  if (comm_ && store_ && server_count_ > 1) {
    // GraphStore::BuildStatistics (graph_store.cc:278-293) asks every other server for its local counts with one
    // GetCount RPC each; here all servers contribute theirs to one all-gather over the shard communicator.
    glx_comm* c = comm_;
    const int32_t world_n = server_count_;
    store_->SetCountGatherer([c, world_n](const std::vector<int32_t>& local, std::vector<std::vector<int32_t>>* all) {
      std::vector<int64_t> mine(local.begin(), local.end());
      int64_t n = (int64_t)mine.size();
      std::vector<int64_t> sizes((size_t)world_n);
      int rc = glx_comm_allgather_i64(c, &n, 1, sizes.data(), GLX_PTR_HOST, nullptr);
      if (rc != GLX_OK) return error::FromGlx(rc);
      for (int64_t s : sizes) {
        if (s != n) return error::Internal("GetStats: the servers declare different numbers of types");
      }
      std::vector<int64_t> got((size_t)world_n * (size_t)n);
      if (n > 0) {
        rc = glx_comm_allgather_i64(c, mine.data(), (int32_t)n, got.data(), GLX_PTR_HOST, nullptr);
        if (rc != GLX_OK) return error::FromGlx(rc);
      }
      all->clear();
      for (int32_t r = 0; r < world_n; ++r) {
        all->emplace_back(got.begin() + (size_t)r * n, got.begin() + (size_t)(r + 1) * n);
      }
      return Status::OK();
    });
  }
}

Env::~Env() {
  if (store_ && server_count_ > 1) store_->SetCountGatherer(nullptr);
  for (auto& kv : edge_stores_) glx_dist_store_destroy(kv.second);
  for (auto& kv : node_stores_) glx_dist_store_destroy(kv.second);
  for (auto& kv : graph_replicas_) glx_graph_destroy(kv.second);  // after the stores that borrowed them
  for (auto& kv : negative_tables_) glx_negative_destroy(kv.second);
}

Status Env::EdgeNegativeTable(const std::string& edge_type, bool by_in_degree, const glx_negative** out) {
  glx_dist_store* st = nullptr;
  Status s = EdgeStore(edge_type, &st);
  if (!s.ok()) return s;
  const std::string key = (by_in_degree ? "e/indeg/" : "e/uniform/") + edge_type;
  if (LookupNegativeTable(key, out)) return Status::OK();
  // The collective runs WITHOUT mtx_: a server whose peers are late must not block its own threads' lookups of
  // other tables / stores.  Two threads of one server cannot be in here for the same key: requests that need the
  // table hold RunMutex() (op_runner.h).
  glx_negative* t = nullptr;
  int rc = glx_dist_negative_create(st, by_in_degree ? 1 : 0, nullptr, &t);  // collective
  if (rc != GLX_OK) return error::FromGlx(rc);
  *out = KeepNegativeTable(key, t);
  return Status::OK();
}

bool Env::LookupNegativeTable(const std::string& key, const glx_negative** out) {
  std::lock_guard<std::mutex> lock(mtx_);
  auto it = negative_tables_.find(key);
  if (it == negative_tables_.end()) return false;
  *out = it->second;
  return true;
}

const glx_negative* Env::KeepNegativeTable(const std::string& key, glx_negative* table) {
  std::lock_guard<std::mutex> lock(mtx_);
  auto ins = negative_tables_.emplace(key, table);
  if (!ins.second) glx_negative_destroy(table);  // somebody else's build of the same table got here first
  return ins.first->second;
}

Status Env::NodeNegativeTable(const std::string& node_type, const glx_negative** out) {
  const std::string key = "n/" + node_type;
  if (LookupNegativeTable(key, out)) return Status::OK();
  {  // collectives without mtx_, as in EdgeNegativeTable
    Noder* noder = store_->GetNoder(node_type);
    // A condition that could differ between the servers must not decide who enters the collective: every server
    // takes part in the count exchange and says there whether its shard can contribute (-1: the type has no
    // weights here); all of them then refuse together.
    const bool usable = noder->GetSideInfo()->IsWeighted();
    // every server's (id, weight) list to every server: the counts first, then the lists (each server sends its own
    // list to all), merged by ascending id -- NodeStorage::GetIds() / GetWeights() of the unpartitioned storage up to order
    const std::vector<int64_t>& ids = noder->Ids();
    const std::vector<float>& weights = noder->Weights();
    int64_t n = usable ? (int64_t)ids.size() : -1;
    const size_t P = (size_t)server_count_;
    std::vector<int64_t> counts(P);
    int rc = glx_comm_allgather_i64(comm_, &n, 1, counts.data(), GLX_PTR_HOST, nullptr);
    if (rc != GLX_OK) return error::FromGlx(rc);
    for (size_t p = 0; p < P; ++p) {
      if (counts[p] < 0) {
        return error::InvalidArgument("node type '" + node_type + "' has no weights on server " + std::to_string(p));
      }
    }
    struct Rec { int64_t id; int64_t w; };
    std::vector<Rec> send(P * (size_t)n);
    for (size_t p = 0; p < P; ++p) {
      for (int64_t i = 0; i < n; ++i) {
        int32_t bits = 0;
        memcpy(&bits, &weights[(size_t)i], 4);
        send[p * (size_t)n + (size_t)i] = {ids[(size_t)i], (int64_t)bits};
      }
    }
    int64_t total = 0;
    for (int64_t c : counts) total += c;
    std::vector<Rec> recv((size_t)(total > 0 ? total : 1));
    std::vector<int64_t> send_counts(P, n);
    rc = glx_exchange_v(comm_, send.data(), send_counts.data(), recv.data(), counts.data(), (int64_t)sizeof(Rec), GLX_PTR_HOST,
                        nullptr);
    if (rc != GLX_OK) return error::FromGlx(rc);
    recv.resize((size_t)total);
    // received in source-rank order: a stable sort by id breaks ties by rank, the same on every server
    std::stable_sort(recv.begin(), recv.end(), [](const Rec& a, const Rec& b) { return a.id < b.id; });
    std::vector<int64_t> all_ids((size_t)total);
    std::vector<float> all_w((size_t)total);
    for (size_t i = 0; i < (size_t)total; ++i) {
      all_ids[i] = recv[i].id;
      const int32_t bits = (int32_t)recv[i].w;
      memcpy(&all_w[i], &bits, 4);
    }
    int device = 0;
    glx_comm_info(comm_, nullptr, nullptr, &device, nullptr);
    glx_negative* t = nullptr;
    rc = glx_negative_create(device, total, all_ids.data(), all_w.data(), GLX_PTR_HOST, nullptr, &t);
    if (rc != GLX_OK) return error::FromGlx(rc);
    *out = KeepNegativeTable(key, t);
  }
  return Status::OK();
}

Status Env::EdgeStore(const std::string& edge_type, glx_dist_store** out) {
  std::lock_guard<std::mutex> lock(mtx_);
  auto it = edge_stores_.find(edge_type);
  if (it == edge_stores_.end()) {
    const glx_graph* g = store_->GetGraph(edge_type)->Device();
    if (!g) return error::InvalidArgument("edge type '" + edge_type + "' is not built on server " + std::to_string(server_id_));
    glx_dist_store* st = nullptr;
    int rc = glx_dist_store_create(comm_, g, nullptr, &st);
    if (rc != GLX_OK) return error::FromGlx(rc);
    it = edge_stores_.emplace(edge_type, st).first;
  }
  *out = it->second;
  return Status::OK();
}

Status Env::NodeStore(const std::string& node_type, glx_dist_store** out) {
  std::lock_guard<std::mutex> lock(mtx_);
  auto it = node_stores_.find(node_type);
  if (it == node_stores_.end()) {
    const glx_features* f = store_->GetNoder(node_type)->Device();
    if (!f) return error::InvalidArgument("node type '" + node_type + "' has no float attributes on server " + std::to_string(server_id_));
    glx_dist_store* st = nullptr;
    int rc = glx_dist_store_create(comm_, nullptr, f, &st);
    if (rc != GLX_OK) return error::FromGlx(rc);
    it = node_stores_.emplace(node_type, st).first;
  }
  *out = it->second;
  return Status::OK();
}

Status Env::ReplicateHotNodes(const std::string& node_type, const int64_t* ids, int64_t count) {
  glx_dist_store* st = nullptr;
  Status s = NodeStore(node_type, &st);
  if (!s.ok()) return s;
  return error::FromGlx(glx_dist_store_set_cache(st, ids, count, GLOBAL_FLAG(DefaultFloatAttribute), GLX_PTR_HOST, nullptr));
}

Status Env::ReplicateHotRows(const std::string& edge_type, const int64_t* ids, int64_t count) {
  glx_dist_store* st = nullptr;
  Status s = EdgeStore(edge_type, &st);
  if (!s.ok()) return s;
  glx_graph* built = nullptr;
  int rc = glx_dist_build_graph_replica(st, ids, count, GLX_PTR_HOST, nullptr, &built);
  if (rc != GLX_OK) return error::FromGlx(rc);
  rc = glx_dist_store_set_graph_replica(st, built);
  if (rc != GLX_OK) {
    glx_graph_destroy(built);
    return error::FromGlx(rc);
  }
  std::lock_guard<std::mutex> g(mtx_);
  glx_graph*& slot = graph_replicas_[edge_type];
  if (slot) glx_graph_destroy(slot);
  slot = built;
  return Status::OK();
}

Status Env::AttachGraphReplica(const std::string& edge_type, const Graph* replica) {
  glx_dist_store* st = nullptr;
  Status s = EdgeStore(edge_type, &st);
  if (!s.ok()) return s;
  if (replica != nullptr && replica->Device() == nullptr) return error::InvalidArgument("the replica graph is not built");
  return error::FromGlx(glx_dist_store_set_graph_replica(st, replica ? replica->Device() : nullptr));
}

Status Env::HotNodes(const std::string& edge_type, int64_t want, std::vector<int64_t>* ids) {
  glx_dist_store* st = nullptr;
  Status s = EdgeStore(edge_type, &st);
  if (!s.ok()) return s;
  ids->assign((size_t)(want > 0 ? want : 1), 0);
  int64_t n = 0;
  int rc = glx_dist_hot_ids(st, want, ids->data(), &n, nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  ids->resize((size_t)n);
  return Status::OK();
}

namespace {

int SamplerIdOf(const std::string& name) {
  if (name == "RandomSampler") return GLX_SAMPLER_RANDOM;
  if (name == "RandomWithoutReplacementSampler") return GLX_SAMPLER_RANDOM_WITHOUT_REPLACEMENT;
  if (name == "EdgeWeightSampler") return GLX_SAMPLER_EDGE_WEIGHT;
  if (name == "TopkSampler") return GLX_SAMPLER_TOPK;
  if (name == "InDegreeSampler") return GLX_SAMPLER_IN_DEGREE;
  return -1;
}

int AggregatorIdOf(const std::string& name) {
  if (name == "SumAggregator") return GLX_AGG_SUM;
  if (name == "MeanAggregator") return GLX_AGG_MEAN;
  if (name == "MaxAggregator") return GLX_AGG_MAX;
  if (name == "MinAggregator") return GLX_AGG_MIN;
  if (name == "ProdAggregator") return GLX_AGG_PROD;
  return -1;
}

// FullSampler's sparse response across shards (full_sampler.cc:28-97 behind DistributeRunner): sizes, then values.
Status RunFullSampling(Env* env, const SamplingRequest* req, SamplingResponse* res) {
  const int32_t batch_size = req->BatchSize();
  if (req->HasFilter() && batch_size > 0 && !req->GetFilterValues()) {
    return error::InvalidArgument("the request has a filter but not one filter value per src id");
  }
  const int32_t max_limit = req->NeighborCount();
  glx_dist_store* st = nullptr;
  Status s = env->EdgeStore(req->Type(), &st);
  if (!s.ok()) return s;
  std::vector<int32_t> degrees((size_t)batch_size, 0);
  std::vector<int64_t> offsets((size_t)batch_size + 1, 0);
  int rc = glx_dist_sample_full_sizes(st, req->GetSrcIds(), batch_size, max_limit, degrees.data(), offsets.data(),
                                      GLX_PTR_HOST, nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  res->SetShape(batch_size, max_limit, degrees);
  res->InitNeighborIds();
  res->InitEdgeIds();
  res->ResizeDense();  // sizes both tensors to the sum of the counts
  // collective even when this server's rows are all empty: its peers may have values to fetch from it
  if (req->HasFilter()) {  // full_sampler.cc:66-84 on the owners: the filter values travel with their rows
    glx_filter filter;
    filter.type = (int32_t)req->GetFilterType();
    filter.field = (int32_t)req->GetFilterField();
    filter.values = req->GetFilterValues();
    filter.retry_times = 0;
    filter.default_timestamp = GLOBAL_FLAG(DefaultTimestamp);
    rc = glx_dist_sample_full_filtered(st, req->GetSrcIds(), batch_size, max_limit, degrees.data(), offsets.data(),
                                       GLOBAL_FLAG(PaddingMode), GLOBAL_FLAG(DefaultNeighborId), &filter,
                                       res->GetNeighborIds(), res->GetEdgeIds(), offsets[(size_t)batch_size], GLX_PTR_HOST,
                                       nullptr);
    return error::FromGlx(rc);
  }
  rc = glx_dist_sample_full(st, req->GetSrcIds(), batch_size, max_limit, degrees.data(), offsets.data(),
                            res->GetNeighborIds(), res->GetEdgeIds(), offsets[(size_t)batch_size], GLX_PTR_HOST, nullptr);
  return error::FromGlx(rc);
}

// The negative samplers across shards (random_negative_sampler.cc:30-63, in_degree_negative_sampler.cc:29-135,
// node_weight_negative_sampler.cc:29-110 behind DistributeRunner): candidates = the WHOLE type's list (the same table on
// every server), strict in-degree sampling served by the owners of the source ids.
Status RunNegativeSampling(Env* env, const SamplingRequest* req, SamplingResponse* res) {
  const std::string& name = req->Strategy();
  const int32_t count = req->NeighborCount(), batch_size = req->BatchSize();
  res->SetShape(batch_size, count);
  res->InitEdgeIds();
  res->InitNeighborIds();
  const glx_negative* table = nullptr;
  glx_dist_store* st = nullptr;
  int exclude = GLX_NEG_EXCLUDE_NONE;
  Status s;
  if (name == "NodeWeightNegativeSampler") {
    exclude = GLX_NEG_EXCLUDE_BATCH;
    s = env->NodeNegativeTable(req->Type(), &table);
  } else {
    const bool by_in_degree = name != "RandomNegativeSampler";
    if (name == "InDegreeNegativeSampler") {
      exclude = GLX_NEG_EXCLUDE_NEIGHBORS;
      const glx_negative* local = nullptr;  // builds the shard's sorted neighbour lists (the exclusion test)
      s = env->Store()->GetGraph(req->Type())->Negative(true, true, &local);
      if (!s.ok()) return s;
    }
    s = env->EdgeNegativeTable(req->Type(), by_in_degree, &table);
    if (s.ok()) s = env->EdgeStore(req->Type(), &st);
  }
  if (!s.ok()) return s;
  res->ResizeNeighborIds();
  const uint64_t cc = req->HasCallCounter() ? (uint64_t)req->CallCounter() : env->NextCallCounter();
  int rc;
  if (st != nullptr) {
    rc = glx_dist_negative_sample(st, table, exclude, req->GetSrcIds(), batch_size, count, GLOBAL_FLAG(DefaultNeighborId),
                                  (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, res->GetNeighborIds(), GLX_PTR_HOST, nullptr);
  } else {  // node-weight sampling needs nothing from another shard once the table is global
    rc = glx_negative_sample(table, exclude, nullptr, req->GetSrcIds(), batch_size, count, GLOBAL_FLAG(DefaultNeighborId),
                             (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, res->GetNeighborIds(), GLX_PTR_HOST, nullptr);
  }
  return error::FromGlx(rc);
}

Status RunSampling(Env* env, const SamplingRequest* req, SamplingResponse* res) {
  if (req->Strategy() == "FullSampler") return RunFullSampling(env, req, res);
  if (req->Strategy().find("NegativeSampler") != std::string::npos && req->Strategy() != "ConditionalNegativeSampler") {
    return RunNegativeSampling(env, req, res);
  }
  const int sampler = SamplerIdOf(req->Strategy());
  if (sampler < 0) {
    return error::Unimplemented("'" + req->Strategy() + "' is not served across shards (dense neighbour samplers are)");
  }
  const int32_t count = req->NeighborCount();
  const int32_t batch_size = req->BatchSize();
  res->SetShape(batch_size, count);
  res->InitNeighborIds();
  res->InitEdgeIds();
  if (req->HasFilter() && batch_size > 0 && !req->GetFilterValues()) {
    return error::InvalidArgument("the request has a filter but not one filter value per src id");
  }
  glx_dist_store* st = nullptr;
  Status s = env->EdgeStore(req->Type(), &st);
  if (!s.ok()) return s;
  if (sampler == GLX_SAMPLER_IN_DEGREE && env->ServerCount() > 1) {
    s = env->Store()->GetGraph(req->Type())->EnsureGlobalInDegree(st);
    if (!s.ok()) return s;
  }
  res->ResizeDense();
  glx_filter filter;
  filter.type = req->HasFilter() && req->GetFilterValues() ? (int32_t)req->GetFilterType() : GLX_FILTER_NONE;
  filter.field = (int32_t)req->GetFilterField();
  filter.values = req->GetFilterValues();
  filter.retry_times = GLOBAL_FLAG(SamplingRetryTimes);
  filter.default_timestamp = GLOBAL_FLAG(DefaultTimestamp);
  // the reference's RNG state advances from call to call; here: a per-deployment call counter
  const uint64_t cc = req->HasCallCounter() ? (uint64_t)req->CallCounter() : env->NextCallCounter();
  int rc = glx_dist_sample(st, sampler, req->GetSrcIds(), batch_size, count, GLOBAL_FLAG(PaddingMode),
                           GLOBAL_FLAG(DefaultNeighborId), (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, &filter,
                           res->GetNeighborIds(), res->GetEdgeIds(), GLX_PTR_HOST, nullptr);
  return error::FromGlx(rc);
}

Status RunAggregating(Env* env, const AggregatingRequest* req, AggregatingResponse* res) {
  const int op = AggregatorIdOf(req->Strategy());
  if (op < 0) return error::Unimplemented("'" + req->Strategy() + "' is not an aggregator this engine serves");
  Noder* noder = env->Store()->GetNoder(req->Type());
  res->SetEmbeddingDim(noder->GetSideInfo()->f_num);
  res->SetNumSegments(req->NumSegments());
  res->SetName(req->Name());
  glx_dist_store* st = nullptr;
  Status s = env->NodeStore(req->Type(), &st);
  if (!s.ok()) return s;
  int rc = glx_dist_aggregate(st, op, req->NodeIds(), req->SegmentIds(), req->NumIds(), req->NumSegments(),
                              GLOBAL_FLAG(DefaultFloatAttribute), res->MutableEmbeddings(), res->MutableSegments(),
                              GLX_PTR_HOST, nullptr);
  return error::FromGlx(rc);
}

}  // namespace

Status RunDistributed(Env* env, op::Operator* op, const OpRequest* req, OpResponse* res) {
  if (!env || !env->Comm() || !env->Store()) return error::InvalidArgument("the runner has no communicator / store");
  std::lock_guard<std::mutex> one_request_at_a_time(env->RunMutex());
  if (auto* sreq = dynamic_cast<const SamplingRequest*>(req)) {
    auto* sres = dynamic_cast<SamplingResponse*>(res);
    if (!sres) return error::InvalidArgument("a SamplingRequest needs a SamplingResponse");
    return RunSampling(env, sreq, sres);
  }
  if (auto* areq = dynamic_cast<const AggregatingRequest*>(req)) {
    auto* ares = dynamic_cast<AggregatingResponse*>(res);
    if (!ares) return error::InvalidArgument("an AggregatingRequest needs an AggregatingResponse");
    return RunAggregating(env, areq, ares);
  }
  if (auto* wreq = dynamic_cast<const RandomWalkRequest*>(req)) {
    auto* wres = dynamic_cast<RandomWalkResponse*>(res);
    if (!wres) return error::InvalidArgument("a RandomWalkRequest needs a RandomWalkResponse");
    // across the shards (glx_dist_random_walk_ex): DeepWalk = one partitioned RandomSampler request per step; node2vec =
    // one partitioned FullSampler request per step + the step on the requester; every step consumes one call counter value
    const int32_t n = wreq->BatchSize(), len = wreq->WalkLen();
    wres->InitWalks(n, len);
    glx_dist_store* st = nullptr;
    Status s = env->EdgeStore(wreq->Type(), &st);
    if (!s.ok()) return s;
    const uint64_t cc = wreq->HasCallCounter() ? (uint64_t)wreq->CallCounter()
                                               : env->NextCallCounters((uint64_t)(len > 0 ? len : 1));
    int rc = glx_dist_random_walk_ex(st, wreq->GetSrcIds(), n, len, wreq->P(), wreq->Q(), GLOBAL_FLAG(DefaultFullNbrNum),
                                     GLOBAL_FLAG(DefaultWeight), GLOBAL_FLAG(DefaultNeighborId),
                                     (uint64_t)GLOBAL_FLAG(SamplingSeed), cc, wres->MutableWalks(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }
  if (auto* dreq = dynamic_cast<const GetDegreeRequest*>(req)) {
    auto* dres = dynamic_cast<GetDegreeResponse*>(res);
    if (!dres) return error::InvalidArgument("a GetDegreeRequest needs a GetDegreeResponse");
    const int32_t n = dreq->Size();
    dres->InitDegrees(n);
    glx_dist_store* st = nullptr;
    Status s = env->EdgeStore(dreq->EdgeType(), &st);
    if (!s.ok()) return s;
    if (dreq->GetNodeFrom() != kEdgeSrc) {
      // in-degrees are sums over ALL shards, held by the ids' owners (glx_dist_in_degrees)
      int rc = glx_dist_in_degrees(st, dreq->NodeIds(), n, dres->MutableDegrees(), GLX_PTR_HOST, nullptr);
      return error::FromGlx(rc);
    }
    // out-degrees live with the rows: the sizes half of the partitioned FullSampler, without a limit
    std::vector<int64_t> offsets((size_t)n + 1, 0);
    int rc = glx_dist_sample_full_sizes(st, dreq->NodeIds(), n, 0, dres->MutableDegrees(), offsets.data(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }
  // "UpdateEdges" / "UpdateNodes": the reference partitions the batch by source id / node id and every server adds
  // its part (graph_update_request.cc:151,234 + HashPartitioner).  SPMD: every server is handed the same batch and
  // keeps the records it owns -- the same shards, without the records crossing a link.  Not a collective.
  const auto owns = [env](int64_t id) {
    const uint64_t a = id < 0 ? (uint64_t)0 - (uint64_t)id : (uint64_t)id;
    return (int32_t)(a % (uint64_t)env->ServerCount()) == env->ServerId();
  };
  if (auto* ureq = dynamic_cast<const UpdateEdgesRequest*>(req)) {
    UpdateEdgesRequest mine(&ureq->GetSideInfo(), ureq->Size());
    for (const auto& v : ureq->Values())
      if (owns(v.src_id)) mine.Append(&v);
    return op->Process(&mine, res);
  }
  if (auto* ureq = dynamic_cast<const UpdateNodesRequest*>(req)) {
    UpdateNodesRequest mine(&ureq->GetSideInfo(), ureq->Size());
    for (const auto& v : ureq->Values())
      if (owns(v.id)) mine.Append(&v);
    return op->Process(&mine, res);
  }
  return error::Unimplemented("request '" + req->Name() + "' is shardable but not served across shards");
}

// subgraph.cc
Status RunSubGraph(const SubGraphRequest* request, SubGraphResponse* response, int device,
                   const std::function<Status(const SamplingRequest*, SamplingResponse*)>& full_sampler);

bool RunWithSubRequests(Env* env, const OpRequest* req, OpResponse* res, Status* status) {
  auto* sub = dynamic_cast<const SubGraphRequest*>(req);
  if (!sub) return false;
  auto* sres = dynamic_cast<SubGraphResponse*>(res);
  if (!sres || !env || !env->Comm()) {
    *status = error::InvalidArgument("a SubGraphRequest needs a SubGraphResponse and a communicator");
    return true;
  }
  int device = 0;
  glx_comm_info(env->Comm(), nullptr, nullptr, &device, nullptr);
  std::lock_guard<std::mutex> one_request_at_a_time(env->RunMutex());
  // COLLECTIVE like every partitioned request: each FullSampler sub-request is one glx_dist_sample_full, so every
  // server must be serving a sub-graph request with the same number of hops at the same time
  *status = RunSubGraph(sub, sres, device,
                        [env](const SamplingRequest* q, SamplingResponse* r) { return RunFullSampling(env, q, r); });
  return true;
}

std::unique_ptr<OpRunner> GetOpRunner(Env* env, op::Operator* op) {
  if (env && env->ServerCount() > 1) {
    return std::unique_ptr<OpRunner>(new DistOpRunner(env, env->ServerId(), op));
  }
  return std::unique_ptr<OpRunner>(new OpRunner(env, op));
}

}  // namespace graphlearn
