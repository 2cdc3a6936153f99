// In-process Client / Server of the local deploy mode -- the two objects the Python
// layer holds (python/graph.py:438-443: Client(...), Server(0, 1, "", "").start(),
// .init(edge_sources, node_sources)).  Mirrors graphlearn/src/include/client.h:33-80
// and include/server.h:30-62 for that mode only: there is no RPC, no queue and no
// tracker here; Client::RunOp is Executor::RunOp (service/executor.cc:34-44), i.e.
// OpFactory::Create(name)->Process(req, res) on the caller's thread.
#ifndef GLX_HOST_CLIENT_H_
#define GLX_HOST_CLIENT_H_
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "graphlearn/aggregating_request.h"
#include "graphlearn/data_source.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/subgraph_request.h"
#include "graphlearn/sampling_request.h"
#include "graphlearn/status.h"

namespace graphlearn {

class DagRequest;
class GetDagValuesRequest;
class GetDagValuesResponse;

class Client {
public:
  ~Client();
  Status Sampling(const SamplingRequest* request, SamplingResponse* response);
  Status Aggregating(const AggregatingRequest* request, AggregatingResponse* response);
  Status LookupNodes(const LookupNodesRequest* request, LookupNodesResponse* response);
  Status LookupEdges(const LookupEdgesRequest* request, LookupEdgesResponse* response);
  Status GetDegree(const GetDegreeRequest* request, GetDegreeResponse* response);
  Status GetCount(const GetCountRequest* request, GetCountResponse* response);
  Status SubGraph(const SubGraphRequest* request, SubGraphResponse* response);
  Status GetStats(const GetStatsRequest* request, GetStatsResponse* response);
  Status RunOp(const OpRequest* request, OpResponse* response);
  // GSL queries (include/client.h:57-58; service/executor.cc:46-71): RunDag registers the query and starts running
  // it in the background (a known id is not an error); GetDagValues takes the next finished round, blocking until
  // there is one.  `cancelled` (optional) lets a caller abandon the wait.  See dag.h.
  Status RunDag(const DagRequest* request);
  Status GetDagValues(const GetDagValuesRequest* request, GetDagValuesResponse* response,
                      const std::function<bool()>* cancelled = nullptr);
  Status Stop();

private:
  Client();
  friend Client* NewInMemoryClient();
};

Client* NewInMemoryClient();

class Server {
public:
  ~Server();
  void Start();
  // Loads every source (files of one type in the order given), builds all storages
  // on the device and binds the operators to the store.  The reference's Init
  // returns void and logs; this one keeps the first error for Status().
  void Init(const std::vector<io::EdgeSource>& edges, const std::vector<io::NodeSource>& nodes);
  void Stop();
  const Status& InitStatus() const { return status_; }
  GraphStore* Store() { return store_; }
  // Device objects of a built type, for callers that keep their data in HBM (torch / DLPack
  // users go straight to the C-ABI with these): 0 when the type is unknown or not built.
  uintptr_t DeviceGraph(const std::string& edge_type);
  uintptr_t DeviceFeatures(const std::string& node_type);

private:
  Server();
  friend Server* NewServer(int32_t, int32_t, const std::string&, const std::string&);
  int32_t shard_index_ = 0, shard_count_ = 1;
  GraphStore* store_;
  Status status_;
  bool bound_;  // the process-wide OpFactory currently serves THIS server's store
};

// server_count > 1: this process holds shard server_id of server_count (GraphStore::SetShard): one
// process per GPU, all reading the same sources; requests are routed between the processes by
// the RCCL layer (graph-learn_amd/dist.py), not by this class.  host/tracker are ignored.
Server* NewServer(int32_t server_id, int32_t server_count, const std::string& server_host,
                  const std::string& tracker);

}  // namespace graphlearn
#endif  // GLX_HOST_CLIENT_H_
