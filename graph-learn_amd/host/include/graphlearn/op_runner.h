// Operator runners.  Mirrors graphlearn/src/core/runner/op_runner.h:33-159:
//   Runner<Req, Res>::Run             -- local: op->Process(req, res)            (:33-48)
//   DistributeRunner<Req, Res>::Run   -- Partition -> one sub-request per shard, shipped to
//                                        its server -> Process there -> Stitch    (:50-152)
//   GetOpRunner(env, op)              -- the distributed runner when the deployment has
//                                        more than one server                     (:156)
// The reference ships the parts with one gRPC call per remote shard from a thread pool
// (RunInParallel, :86-117).  Here the servers are the GPUs of one node, one process (or
// host thread) each, and `Env` carries the shard communicator (include/glx.h glx_comm: RCCL
// send/recv groups over xGMI) instead of channels and pools.  Run() is one call into the
// device-resident distributed store (glx_dist_sample / glx_dist_aggregate): partition,
// exchange, the owner's kernels, exchange back and stitch all happen on the GPUs; the
// request and response objects are only copied in and out.
//
// SPMD, as the reference's worker loop is not: every rank calls Run() for the same operator
// at the same time, each with its OWN request (a rank with nothing to ask passes an empty
// one).  Results equal the single-store operator's, draw for draw.
#ifndef GLX_HOST_OP_RUNNER_H_
#define GLX_HOST_OP_RUNNER_H_
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

#include "graphlearn/graph_store.h"
#include "graphlearn/op_request.h"
#include "graphlearn/operator.h"
#include "graphlearn/status.h"

struct glx_comm;
struct glx_dist_store;

namespace graphlearn {

// Bootstrap of the shard communicator through a shared directory, the way the reference's servers find each
// other (service/dist/fs_naming_engine.cc:30-106: every server writes `<tracker>/endpoints/<id>`, the others
// poll the directory): server 0 makes the RCCL unique id and publishes it as `<tracker>/glx_comm/<session>`
// (written to a temporary name, then renamed: readers never see a partial file); the others poll for it.
// `session` names this run of the deployment (a job id, a start time): a file left behind by an earlier run
// under another session is never picked up.  Returns the 128 id bytes in `id`; when server 0 passes an `id`
// that already holds 128 bytes, those are published instead of a fresh RCCL id (tests, other transports).
Status ExchangeUniqueId(const std::string& tracker, const std::string& session, int32_t server_id,
                        double timeout_seconds, std::string* id);
// ExchangeUniqueId + glx_comm_init_rccl(device, server_id, server_count): every server calls it once (twice,
// with two session names, for two communicators).  The caller owns *comm (glx_comm_destroy).
Status ConnectServers(const std::string& tracker, const std::string& session, int device, int32_t server_id,
                      int32_t server_count, glx_comm** comm);

// What a runner needs from the deployment (the role of platform/env.h + the naming engine):
// this server's id, the server count, the way to the other servers, and the local shard.
class Env {
public:
  // `comm` stays owned by the caller and must outlive the Env; `store` is this rank's shard
  // (GraphStore::SetShard(comm rank, comm world) before loading).
  Env(glx_comm* comm, GraphStore* store);
  ~Env();
  Env(const Env&) = delete;
  Env& operator=(const Env&) = delete;

  int32_t ServerId() const { return server_id_; }
  int32_t ServerCount() const { return server_count_; }
  glx_comm* Comm() const { return comm_; }
  GraphStore* Store() const { return store_; }

  // Collective.  Every GPU keeps a copy of these nodes' float attributes (the same id list on
  // every rank); aggregation then fetches only the remaining remote rows per request.
  Status ReplicateHotNodes(const std::string& node_type, const int64_t* ids, int64_t count);
  // Collective.  Every GPU keeps the complete adjacency rows of these vertices of `edge_type` (the same id list
  // on every rank): the owners cut them out of their shards -- neighbours, their edge ids, weights, in storage
  // order -- and the pieces are all-gathered once (glx_dist_build_graph_replica); sampling requests for those
  // vertices are then served locally, with the same draws.  The Env owns the replica.
  Status ReplicateHotRows(const std::string& edge_type, const int64_t* ids, int64_t count);
  // Not collective.  `replica` holds the complete adjacency rows of a (hot) vertex set, built like any Graph
  // (the hot rows' edges loaded on every server); sampling requests for those vertices are then served here
  // instead of travelling to their owner (glx_dist_store_set_graph_replica).  nullptr detaches it; the caller
  // keeps `replica` alive.
  Status AttachGraphReplica(const std::string& edge_type, const Graph* replica);
  // Collective.  The `want` vertices with the largest in-degree over all shards of `edge_type`.
  Status HotNodes(const std::string& edge_type, int64_t want, std::vector<int64_t>* ids);

  // Device-resident distributed stores of one edge / node type, created on first use.
  Status EdgeStore(const std::string& edge_type, glx_dist_store** out);
  Status NodeStore(const std::string& node_type, glx_dist_store** out);
  // Collective on first use.  The negative samplers' candidate list over the WHOLE type, the same on every server:
  // an edge type's destination ids of all shards (ascending; uniform or weighted by the in-degree summed over all
  // shards: glx_dist_negative_create), a node type's ids and node weights of all shards (ascending).  Owned by the Env.
  Status EdgeNegativeTable(const std::string& edge_type, bool by_in_degree, const glx_negative** out);
  Status NodeNegativeTable(const std::string& node_type, const glx_negative** out);
  // A distributed store keeps per-call state (request / receive arenas, exchange counters, halo slots) and every
  // partitioned request is a sequence of collectives: two requests of one server must not interleave, and all
  // servers must issue their requests in the same order.  The runners hold this lock for the whole request, so the
  // pool threads of one server (the reference runs operators from up to 32, in_memory_service.cc:64-71) take turns;
  // the order across servers is the caller's contract (SPMD: INTEGRATION.md).
  std::mutex& RunMutex() { return run_mtx_; }
  uint64_t NextCallCounter() { return call_counter_.fetch_add(1, std::memory_order_relaxed); }
  // `count` consecutive values at once (a walk consumes one per step); returns the first
  uint64_t NextCallCounters(uint64_t count) { return call_counter_.fetch_add(count ? count : 1, std::memory_order_relaxed); }

private:
  bool LookupNegativeTable(const std::string& key, const glx_negative** out);
  const glx_negative* KeepNegativeTable(const std::string& key, glx_negative* table);
  glx_comm* comm_;
  GraphStore* store_;
  int32_t server_id_, server_count_;
  std::mutex mtx_;
  std::mutex run_mtx_;
  std::unordered_map<std::string, glx_dist_store*> edge_stores_, node_stores_;
  std::unordered_map<std::string, glx_graph*> graph_replicas_;  // built by ReplicateHotRows
  std::unordered_map<std::string, glx_negative*> negative_tables_;
  std::atomic<uint64_t> call_counter_{0};
};

// op_runner.h:33-48
template <class Request, class Response>
class Runner {
public:
  Runner(Env* env, op::Operator* op) : env_(env), op_(op) {}
  virtual ~Runner() = default;
  virtual Status Run(const Request* req, Response* res) { return op_->Process(req, res); }

protected:
  Env* env_;
  op::Operator* op_;
};

// The body of DistributeRunner::Run for the requests this engine serves across shards:
// SamplingRequest (dense samplers) and AggregatingRequest.  Anything else that is shardable
// is refused with Unimplemented rather than silently answered from the local shard.
Status RunDistributed(Env* env, op::Operator* op, const OpRequest* req, OpResponse* res);

// Unshardable requests whose operator fans out sub-requests (SubGraphSampler): served with the sub-requests going
// through the distributed store.  Returns false when `req` is not of that kind (the caller runs the local operator).
bool RunWithSubRequests(Env* env, const OpRequest* req, OpResponse* res, Status* status);

// op_runner.h:50-152
template <class Request, class Response>
class DistributeRunner : public Runner<Request, Response> {
public:
  DistributeRunner(Env* env, int32_t local_id, op::Operator* op)
      : Runner<Request, Response>(env, op), local_id_(local_id) {}
  Status Run(const Request* req, Response* res) override {
    if (!req->IsShardable()) {  // op_runner.h:61-62: the operator itself runs, here
      // ... but an operator that issues sub-requests through GetOpRunner(Env::Default(), op) (SubGraphSampler's
      // FullSampler calls, subgraph_sampler.cc:27-32) must get THIS deployment's runner for them
      Status s;
      if (RunWithSubRequests(this->env_, req, res, &s)) return s;
      return Runner<Request, Response>::Run(req, res);
    }
    return RunDistributed(this->env_, this->op_, req, res);
  }

private:
  int32_t local_id_;
};

typedef Runner<OpRequest, OpResponse> OpRunner;
typedef DistributeRunner<OpRequest, OpResponse> DistOpRunner;

// op_runner.h:156 / op_runner.cc: distributed when there is more than one server.
std::unique_ptr<OpRunner> GetOpRunner(Env* env, op::Operator* op);

}  // namespace graphlearn
#endif  // GLX_HOST_OP_RUNNER_H_
