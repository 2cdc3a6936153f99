// Local-mode Client / Server (see client.h).
#include "graphlearn/client.h"

#include <mutex>
#include <vector>

#include "graphlearn/dag.h"
#include "graphlearn/operator.h"

namespace graphlearn {

Client::Client() {}
Client::~Client() {}

Status Client::RunOp(const OpRequest* request, OpResponse* response) {
  // Executor::RunOp (service/executor.cc:34-44)
  op::Operator* op = op::OpFactory::GetInstance()->Create(request->Name());
  if (!op) return error::InvalidArgument("Invalid request name: " + request->Name());
  return op->Process(request, response);
}

Status Client::Sampling(const SamplingRequest* request, SamplingResponse* response) { return RunOp(request, response); }
Status Client::Aggregating(const AggregatingRequest* request, AggregatingResponse* response) {
  return RunOp(request, response);
}
Status Client::LookupNodes(const LookupNodesRequest* request, LookupNodesResponse* response) {
  return RunOp(request, response);
}
Status Client::LookupEdges(const LookupEdgesRequest* request, LookupEdgesResponse* response) {
  return RunOp(request, response);
}
Status Client::GetDegree(const GetDegreeRequest* request, GetDegreeResponse* response) {
  return RunOp(request, response);
}
Status Client::GetCount(const GetCountRequest* request, GetCountResponse* response) { return RunOp(request, response); }
Status Client::GetStats(const GetStatsRequest* request, GetStatsResponse* response) { return RunOp(request, response); }
Status Client::SubGraph(const SubGraphRequest* request, SubGraphResponse* response) { return RunOp(request, response); }
Status Client::Stop() { return Status::OK(); }

Client* NewInMemoryClient() { return new Client(); }

namespace {
// The operators serve ONE store per process (OpFactory::Set, like the reference).  With
// several Server objects alive, the most recently initialised one is served; stopping it
// hands the operators back to the previous one instead of leaving them unbound.
std::mutex g_bound_mtx;
std::vector<Server*> g_bound;
}  // namespace

Server::Server() : store_(nullptr), bound_(false) {}

Server::~Server() { Stop(); }

void Server::Start() {
  if (!store_) {
    store_ = new GraphStore();
    store_->SetShard(shard_index_, shard_count_);
  }
}

void Server::Init(const std::vector<io::EdgeSource>& edges, const std::vector<io::NodeSource>& nodes) {
  Start();
  status_ = Status::OK();
  IndexOption option;
  option.name = "sort";
  for (const io::EdgeSource& e : edges) store_->DeclareEdgeType(e.edge_type);  // GraphStore::Init, graph_store.cc:185-208
  for (const io::NodeSource& n : nodes) store_->DeclareNodeType(n.id_type);
  for (const io::EdgeSource& e : edges) {
    if (!status_.ok()) break;
    status_ = io::LoadEdges(e, store_);
    option = e.option;
  }
  for (const io::NodeSource& n : nodes) {
    if (!status_.ok()) break;
    status_ = io::LoadNodes(n, store_);
  }
  if (status_.ok()) status_ = store_->Build(option);
  if (status_.ok()) {
    std::lock_guard<std::mutex> g(g_bound_mtx);
    op::OpFactory::GetInstance()->Set(store_);
    if (!bound_) g_bound.push_back(this);
    bound_ = true;
  }
}

uintptr_t Server::DeviceGraph(const std::string& edge_type) {
  return store_ ? reinterpret_cast<uintptr_t>(store_->GetGraph(edge_type)->Device()) : 0;
}

uintptr_t Server::DeviceFeatures(const std::string& node_type) {
  return store_ ? reinterpret_cast<uintptr_t>(store_->GetNoder(node_type)->Device()) : 0;
}

void Server::Stop() {
  if (store_) {
    // Queries reach a store only through the process-wide OpFactory, i.e. they read the store that is being served
    // NOW: they end before it does.  A Server whose store the operators do not serve (an older one beside a newer
    // one) takes nobody's queries with it (Server::Stop of any one Server stopped them all).
    bool serving;
    {
      std::lock_guard<std::mutex> g(g_bound_mtx);
      serving = bound_ && !g_bound.empty() && g_bound.back() == this;
    }
    if (serving) DagScheduler::StopAll();
    if (bound_) {
      std::lock_guard<std::mutex> g(g_bound_mtx);
      for (size_t i = 0; i < g_bound.size(); ++i) {
        if (g_bound[i] == this) g_bound.erase(g_bound.begin() + i);
      }
      op::OpFactory::GetInstance()->Set(g_bound.empty() ? nullptr : g_bound.back()->store_);
    }
    bound_ = false;
    delete store_;
    store_ = nullptr;
  }
}

Server* NewServer(int32_t server_id, int32_t server_count, const std::string&, const std::string&) {
  Server* s = new Server();
  s->shard_index_ = server_id;
  s->shard_count_ = server_count;
  return s;
}

}  // namespace graphlearn
