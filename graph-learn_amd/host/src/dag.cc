// Query DAGs: definitions, compilation into steps, tapes, the node runner, the per-query scheduler thread, the
// values request / response and the prefetching Dataset.  See dag.h for the design; each block cites the
// reference code whose behaviour it keeps.
#include "graphlearn/dag.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

#include "graphlearn/client.h"
#include "graphlearn/config.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/op_runner.h"
#include "graphlearn/sampling_request.h"

namespace graphlearn {

// ------------------------------------------------------------------ definitions --
namespace {
void PrintTensor(std::ostringstream& os, const std::string& name, const Tensor& t, const char* indent) {
  os << indent << "params { name: \"" << name << "\" length: " << t.Size() << " values:";
  for (int32_t i = 0; i < t.Size(); ++i) {
    switch (t.DType()) {
      case kInt32: os << ' ' << t.GetInt32(i); break;
      case kInt64: os << ' ' << t.GetInt64(i); break;
      case kFloat: os << ' ' << t.GetFloat(i); break;
      case kDouble: os << ' ' << t.GetDouble(i); break;
      case kString: os << " \"" << t.GetString(i) << '"'; break;
      default: break;
    }
  }
  os << " }\n";
}
void PrintEdge(std::ostringstream& os, const char* what, const DagEdgeDef& e) {
  os << "  " << what << " { id: " << e.id << " src_output: \"" << e.src_output << "\" dst_input: \"" << e.dst_input
     << "\" }\n";
}
}  // namespace

std::string DagDef::DebugString() const {  // the text form of the message, as pywrap.debug_string prints it
  std::ostringstream os;
  os << "id: " << id << "\n";
  for (const DagNodeDef& n : nodes) {
    os << "nodes {\n  id: " << n.id << "\n  op_name: \"" << n.op_name << "\"\n";
    std::vector<std::string> keys;
    for (const auto& it : n.params) keys.push_back(it.first);
    std::sort(keys.begin(), keys.end());
    for (const std::string& k : keys) PrintTensor(os, k, n.params.at(k), "  ");
    for (const DagEdgeDef& e : n.in_edges) PrintEdge(os, "in_edges", e);
    for (const DagEdgeDef& e : n.out_edges) PrintEdge(os, "out_edges", e);
    os << "}\n";
  }
  return os.str();
}

// ----------------------------------------------------------------------- graph --
// core/dag/dag_node.cc:21-47.  The reference resolves edge ids through a process-wide table
// (dag_edge.cc:38-50); the table is per query here -- edge ids are unique across the Python process anyway
// (python/gsl/dag_edge.py:79-96) and a query's edges die with it.
DagNode::DagNode(const DagNodeDef& def, std::unordered_map<int32_t, std::shared_ptr<DagEdge>>* edges)
    : id_(def.id), op_name_(def.op_name), params_(def.params) {
  auto resolve = [edges](const DagEdgeDef& e) {
    std::shared_ptr<DagEdge>& slot = (*edges)[e.id];
    if (!slot) slot = std::make_shared<DagEdge>(e);
    return slot;
  };
  for (const DagEdgeDef& e : def.in_edges) {
    std::shared_ptr<DagEdge> edge = resolve(e);
    edge->SetDst(this);
    in_edges_.push_back(edge);
  }
  for (const DagEdgeDef& e : def.out_edges) {
    std::shared_ptr<DagEdge> edge = resolve(e);
    edge->SetSrc(this);
    out_edges_.push_back(edge);
  }
}

Dag::Dag(const DagDef& def) : id_(def.id), debug_(def.DebugString()) {  // core/dag/dag.cc:24-38
  for (const DagNodeDef& n : def.nodes) {
    nodes_.emplace_back(new DagNode(n, &edges_));
    if (nodes_.back()->InDegree() == 0) root_ = nodes_.back().get();
  }
}

namespace {
// The dense neighbour samplers: fixed [batch, k] answers, so hop h + 1's request shape is known before hop h ran.
bool IsDenseSampler(const std::string& op) {
  return op == "RandomSampler" || op == "RandomWithoutReplacementSampler" || op == "TopkSampler" ||
         op == "EdgeWeightSampler" || op == "InDegreeSampler";
}
bool HasFilter(const DagNode* n) {
  auto it = n->Params().find(kFilterType);
  return it != n->Params().end() && it->second.Size() > 0 && it->second.GetInt32(0) != 0;
}
bool FusableHop(const DagNode* n) { return IsDenseSampler(n->OpName()) && !HasFilter(n); }
// `next` continues the chain at `prev`: the same sampler, fed by prev's neighbour ids and by nothing else.
bool Continues(const DagNode* prev, const DagNode* next) {
  if (!FusableHop(next) || next->OpName() != prev->OpName() || next->InDegree() != 1) return false;
  const DagEdge& e = *next->InEdges()[0];
  return e.Src() == prev && e.SrcOutput() == kNodeIds && e.DstInput() == kSrcIds;
}
}  // namespace

Status Dag::Compile() {
  steps_.clear();
  if (!root_) return error::InvalidArgument("the dag has no root (a node without in-edges)");
  for (const auto& it : edges_) {
    if (!it.second->Src() || !it.second->Dst()) {
      return error::InvalidArgument("dag edge " + std::to_string(it.first) + " is attached at one end only");
    }
  }
  std::vector<bool> seen(nodes_.size() + 1, false);
  for (const auto& n : nodes_) {
    if (n->Id() < 1 || n->Id() > Size()) return error::InvalidArgument("dag node ids must be 1..size");
    // a node's id is its tape slot: two nodes with one id would overwrite each other's record
    if (seen[n->Id()]) return error::InvalidArgument("dag node id " + std::to_string(n->Id()) + " is used twice");
    seen[n->Id()] = true;
  }
  std::vector<int32_t> waiting(nodes_.size() + 1, 0);
  std::vector<bool> emitted(nodes_.size() + 1, false);
  std::deque<const DagNode*> ready;
  for (const auto& n : nodes_) {
    waiting[n->Id()] = n->InDegree();
    if (n->InDegree() == 0) ready.push_back(n.get());
  }
  size_t visited = 0;
  while (!ready.empty()) {
    const DagNode* n = ready.front();
    ready.pop_front();
    ++visited;
    if (!emitted[n->Id()]) {
      Step step;
      step.nodes.push_back(n);
      emitted[n->Id()] = true;
      // a chain of dense hops becomes one step: its later members depend on nothing but the member before them
      const DagNode* tail = n;
      while (FusableHop(tail)) {
        const DagNode* next = nullptr;
        for (const auto& e : tail->OutEdges()) {
          if (!emitted[e->Dst()->Id()] && Continues(tail, e->Dst())) {
            next = e->Dst();
            break;
          }
        }
        if (!next) break;
        step.nodes.push_back(next);
        emitted[next->Id()] = true;
        tail = next;
      }
      steps_.push_back(std::move(step));
    }
    for (const auto& e : n->OutEdges()) {
      if (--waiting[e->Dst()->Id()] == 0) ready.push_back(e->Dst());
    }
  }
  if (visited != nodes_.size()) return error::InvalidArgument("the dag has a cycle or unreachable nodes");
  return Status::OK();
}

DagFactory* DagFactory::GetInstance() {
  static DagFactory* factory = new DagFactory;  // never destroyed: scheduler threads may outlive static teardown
  return factory;
}

Status DagFactory::Create(const DagDef& def, Dag** dag) {
  std::lock_guard<std::mutex> g(mtx_);
  if (map_.count(def.id)) return Status(error::ALREADY_EXISTS, "Dag has already existed.");
  std::unique_ptr<Dag> made(new Dag(def));
  Status s = made->Compile();
  if (!s.ok()) return s;
  *dag = made.get();
  map_[def.id] = std::move(made);
  return Status::OK();
}

Dag* DagFactory::Lookup(int32_t dag_id) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = map_.find(dag_id);
  return it == map_.end() ? nullptr : it->second.get();
}

void DagFactory::Clear() {
  std::lock_guard<std::mutex> g(mtx_);
  map_.clear();
}

// ------------------------------------------------------------------------ tapes --
std::pair<const Tensor*, const Tensor*> TensorMap::Find(const std::string& key) const {  // tensor_map.cc:57-71
  auto it = tensors_.find(key);
  if (it != tensors_.end()) return {&it->second, nullptr};
  auto sp = sparse_tensors_.find(key);
  if (sp == sparse_tensors_.end()) return {nullptr, nullptr};
  return {&sp->second.Values(), &sp->second.Segments()};
}

bool TensorMap::Add(const std::string& key, const Tensor* values, const Tensor* segments) {  // :73-84
  if (!values && !segments) return false;
  if (segments) {
    sparse_tensors_.emplace(key, SparseTensor(*segments, *values));
  } else {
    tensors_.emplace(key, *values);
  }
  return true;
}

namespace {
typedef std::chrono::milliseconds Ms;
const Ms kPoll(50);  // how often a blocked producer / consumer looks at its stop condition
}  // namespace

TapeStore::TapeStore(int32_t capacity, const Dag* dag) : cap_(capacity > 0 ? capacity : 1), dag_(dag) {}

TapeStore::~TapeStore() {
  for (Tape* t : queue_) delete t;
}

bool TapeStore::WaitAndPush(Tape* tape, const std::function<bool()>& stop) {  // tape.cc:105-125
  std::unique_lock<std::mutex> lock(mtx_);
  tape->SetEpoch(epoch_);
  if (tape->IsFaked()) ++epoch_;
  while ((int32_t)queue_.size() >= cap_ || closed_) {
    if (closed_ || stop()) {
      delete tape;
      return false;
    }
    room_.wait_for(lock, kPoll);
  }
  queue_.push_back(tape);
  data_.notify_one();
  return true;
}

Tape* TapeStore::WaitAndPop(int32_t client_id, const std::function<bool()>& stop) {  // :127-148
  std::unique_lock<std::mutex> lock(mtx_);
  while (queue_.empty()) {
    // closed_ is sticky: a consumer that wakes up after StopAll has already finished (its transient flag is down
    // again, no producer is left) still learns that nothing more will come
    if (closed_ || stop()) return nullptr;
    data_.wait_for(lock, kPoll);
  }
  Tape* tape = queue_.front();
  queue_.pop_front();
  auto it = tape_indexes_.emplace(client_id, -1).first;
  tape->SetId(++it->second);  // the index order is the pop order
  room_.notify_one();
  return tape;
}

void TapeStore::Close() {
  std::lock_guard<std::mutex> g(mtx_);
  closed_ = true;
  room_.notify_all();
  data_.notify_all();
}

namespace {
struct Running {
  std::thread thread;
  std::shared_ptr<std::atomic<bool>> stop;
};
struct Registry {
  std::mutex mtx;
  std::unordered_map<int32_t, TapeStorePtr> stores;
  std::vector<Running> running;
  std::atomic<bool> stopping{false};  // between StopAll's start and end: consumers give up instead of waiting
};
Registry& Reg() {
  static Registry* r = new Registry;  // see DagFactory::GetInstance
  return *r;
}
}  // namespace

TapeStorePtr GetTapeStore(int32_t dag_id) {
  Registry& r = Reg();
  std::lock_guard<std::mutex> g(r.mtx);
  auto it = r.stores.find(dag_id);
  if (it != r.stores.end()) return it->second;
  Dag* dag = DagFactory::GetInstance()->Lookup(dag_id);
  if (!dag) return nullptr;
  TapeStorePtr store(new TapeStore(GLOBAL_FLAG(TapeCapacity), dag));
  r.stores[dag_id] = store;
  return store;
}

// ---------------------------------------------------------------------- running --
bool DagNodeRunner::BuildInput(const DagNode* node, Tape* tape, TensorMap* tensors) {  // dag_node_runner.cc:54-70
  for (const auto& edge : node->InEdges()) {
    auto found = tape->Retrieval(edge->Src()->Id()).Find(edge->SrcOutput());
    if (!tensors->Add(edge->DstInput(), found.first, found.second)) return false;
  }
  return true;
}

std::unique_ptr<OpRequest> DagNodeRunner::MakeOpRequest(const DagNode* node, const TensorMap* tensors) {  // :100-109
  std::unique_ptr<OpRequest> req(RequestFactory::GetInstance()->NewRequest(node->OpName()));
  if (!req) return nullptr;
  try {  // a parameter or input the request needs and the query did not provide (Tensor::Map::at)
    req->Init(node->Params());
    if (tensors) req->Set(tensors->tensors_, tensors->sparse_tensors_);
  } catch (const std::out_of_range&) {
    return nullptr;
  }
  return req;
}

namespace {
// What a response leaves on the tape: its tensors under their keys.  Two response kinds keep part of their content
// outside the tensor map in this mirror and are spelled out: a ragged sampling response (FullSampler: per-row counts in
// its Shape) becomes {segments, values} pairs like the reference's sparse_tensors_ (sampling_request.cc:177-195), and a
// lookup's string attributes become a string tensor under "sa".
TensorMap Recorded(OpResponse* res) {
  TensorMap out;
  SamplingResponse* sampled = dynamic_cast<SamplingResponse*>(res);
  const Shape shape = sampled ? sampled->GetShape() : Shape();
  if (sampled && shape.sparse) {
    const std::vector<int32_t>& counts = shape.segments;
    Tensor segments(kInt32, (int32_t)counts.size());
    segments.AddInt32(counts.data(), counts.data() + counts.size());
    for (const char* key : {kNodeIds, kEdgeIds}) {
      auto it = res->tensors_.find(key);
      if (it != res->tensors_.end()) out.sparse_tensors_.emplace(key, SparseTensor(segments, it->second));
    }
    return out;
  }
  out.tensors_ = std::move(res->tensors_);
  if (LookupResponse* looked = dynamic_cast<LookupResponse*>(res)) {
    if (looked->StringAttrNum() > 0) {
      const std::vector<std::string>& strings = looked->StringAttrs();
      Tensor t(kString, (int32_t)strings.size());
      for (const std::string& s : strings) t.AddString(s);
      out.tensors_.emplace(kStringAttrKey, t);
    }
  }
  return out;
}
}  // namespace

void DagNodeRunner::RunNode(const DagNode* node, Tape* tape) {  // dag_node_runner.cc:32-52, 72-98
  if (node->IsSink()) {
    tape->SetReady();
    return;
  }
  TensorMap tensors;
  if (!BuildInput(node, tape, &tensors)) return tape->Fake();
  op::Operator* op = op::OpFactory::GetInstance()->Create(node->OpName());
  std::unique_ptr<OpRequest> req = op ? MakeOpRequest(node, &tensors) : nullptr;
  std::unique_ptr<OpResponse> res(RequestFactory::GetInstance()->NewResponse(node->OpName()));
  if (!op || !req || !res) return tape->Fake();
  Status s = (env_ && env_->ServerCount() > 1) ? GetOpRunner(env_, op)->Run(req.get(), res.get())
                                               : op->Process(req.get(), res.get());
  if (!s.ok()) return tape->Fake();  // OUT_OF_RANGE from the root: the end of an epoch
  tape->Record(node->Id(), Recorded(res.get()));
}

void DagNodeRunner::RunHops(const Dag::Step& step, Tape* tape) {
  const DagNode* first = step.nodes[0];
  op::Operator* op = op::OpFactory::GetInstance()->Create(first->OpName());
  op::HopFusable* fused = dynamic_cast<op::HopFusable*>(op);
  if (!fused || (env_ && env_->ServerCount() > 1)) {  // partitioned requests go hop by hop through the exchange
    for (const DagNode* n : step.nodes) {
      RunNode(n, tape);
      if (tape->IsFaked()) return;
    }
    return;
  }
  TensorMap tensors;
  if (!BuildInput(first, tape, &tensors)) return tape->Fake();
  std::vector<std::unique_ptr<OpRequest>> requests;
  std::vector<std::unique_ptr<OpResponse>> responses;
  std::vector<const OpRequest*> req_ptrs;
  std::vector<OpResponse*> res_ptrs;
  for (size_t h = 0; h < step.nodes.size(); ++h) {
    requests.push_back(MakeOpRequest(step.nodes[h], h == 0 ? &tensors : nullptr));  // later hops: parameters only
    responses.emplace_back(RequestFactory::GetInstance()->NewResponse(step.nodes[h]->OpName()));
    if (!requests.back() || !responses.back()) return tape->Fake();
    req_ptrs.push_back(requests.back().get());
    res_ptrs.push_back(responses.back().get());
  }
  Status s = fused->ProcessHops(req_ptrs, res_ptrs);
  if (!s.ok()) return tape->Fake();
  for (size_t h = 0; h < step.nodes.size(); ++h) tape->Record(step.nodes[h]->Id(), Recorded(res_ptrs[h]));
}

void DagNodeRunner::Run(const Dag::Step& step, Tape* tape) {
  if (step.nodes.size() == 1) return RunNode(step.nodes[0], tape);
  RunHops(step, tape);
}

namespace {
void RunQuery(Env* env, const Dag* dag, TapeStorePtr store, std::shared_ptr<std::atomic<bool>> stop) {
  DagNodeRunner runner(env);
  auto stopped = [stop]() { return stop->load(); };
  // GLX_DAG_TRACE=n (read once per query): the wall time of every step of the first n rounds on stderr
  const char* trace_env = getenv("GLX_DAG_TRACE");
  int trace_rounds = trace_env ? atoi(trace_env) : 0;
  while (!stopped()) {  // dag_scheduler.cc:45-60: round after round until the server stops
    Tape* tape = store->New();
    for (const Dag::Step& step : dag->Steps()) {
      const auto t0 = std::chrono::steady_clock::now();
      try {
        runner.Run(step, tape);
      } catch (const std::exception&) {  // an operator that throws (out of memory, a malformed request) costs this
        tape->Fake();                    // round, not the process: nobody above this thread could catch it
      }
      if (trace_rounds > 0) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[glx dag %d] step %s%s (node %d): %.3f ms\n", dag->Id(), step.nodes[0]->OpName().c_str(),
                step.nodes.size() > 1 ? " x hops" : "", step.nodes[0]->Id(), ms);
      }
      if (tape->IsFaked() || tape->IsReady()) break;
    }
    if (trace_rounds > 0) --trace_rounds;
    if (!tape->IsFaked() && !tape->IsReady()) tape->Fake();  // a query without a sink never completes a round
    if (!store->WaitAndPush(tape, stopped)) break;
  }
}
}  // namespace

void DagScheduler::Take(Env* env, const Dag* dag) {
  TapeStorePtr store = GetTapeStore(dag->Id());
  Registry& r = Reg();
  std::lock_guard<std::mutex> g(r.mtx);
  static bool at_exit = (std::atexit([]() { DagScheduler::StopAll(); }), true);  // never launch kernels into a runtime that is being torn down
  (void)at_exit;
  Running run;
  run.stop = std::make_shared<std::atomic<bool>>(false);
  run.thread = std::thread(RunQuery, env, dag, store, run.stop);
  r.running.push_back(std::move(run));
}

void DagScheduler::StopAll() {
  Registry& r = Reg();
  std::vector<Running> running;
  std::vector<TapeStorePtr> stores;
  {
    std::lock_guard<std::mutex> g(r.mtx);
    r.stopping = true;
    running.swap(r.running);
    for (auto& it : r.stores) stores.push_back(it.second);
  }
  for (Running& run : running) run.stop->store(true);
  for (TapeStorePtr& s : stores) s->Close();
  for (Running& run : running) {
    if (run.thread.joinable()) run.thread.join();
  }
  {
    std::lock_guard<std::mutex> g(r.mtx);
    r.stores.clear();  // consumers blocked on a store keep it alive through their own reference
    r.stopping = false;
  }
  DagFactory::GetInstance()->Clear();
}

// --------------------------------------------------------- requests / responses --
bool DagRequest::ParseFrom(DagDef* def, bool copy) {  // dag_request.cc:35-43
  if (copy) {
    def_ = *def;
  } else {
    def_ = std::move(*def);
    *def = DagDef();
  }
  return true;
}

GetDagValuesRequest::GetDagValuesRequest() : id_(-1), client_id_(GLOBAL_FLAG(ClientId)) {}
GetDagValuesRequest::GetDagValuesRequest(int32_t dag_id) : id_(dag_id), client_id_(GLOBAL_FLAG(ClientId)) {}

void GetDagValuesResponse::MoveFrom(Tape* tape) {  // dag_request.cc:88-95: every node's record but the sink's (the last)
  for (int32_t i = 1; i < tape->Size(); ++i) {
    TensorMap& record = tape->Retrieval(i);
    if (record.Size() > 0) records_.emplace(i, std::move(record));
  }
}

std::pair<const Tensor*, const Tensor*> GetDagValuesResponse::GetValue(int32_t node_id, const std::string& key) const {
  auto it = records_.find(node_id);
  if (it == records_.end()) return {nullptr, nullptr};
  return it->second.Find(key);
}

// Executor::RunDag / GetDagValues (service/executor.cc:46-71), reached through the in-memory client.
Status Client::RunDag(const DagRequest* request) {
  Dag* dag = nullptr;
  Status s = DagFactory::GetInstance()->Create(request->def_, &dag);
  if (s.code() == error::ALREADY_EXISTS) return Status::OK();
  if (s.ok()) DagScheduler::Take(nullptr, dag);
  return s;
}

Status Client::GetDagValues(const GetDagValuesRequest* request, GetDagValuesResponse* response,
                            const std::function<bool()>* cancelled) {
  TapeStorePtr store = GetTapeStore(request->Id());
  if (!store) return error::NotFound("no running dag with id " + std::to_string(request->Id()));
  Registry& r = Reg();
  std::unique_ptr<Tape> tape(store->WaitAndPop(request->ClientId(), [&r, cancelled]() {
    return r.stopping.load() || (cancelled && (*cancelled)());
  }));
  if (!tape) return Status(error::CANCELLED, "the dag was stopped");
  response->SetIndex(tape->Id());
  response->SetEpoch(tape->Epoch());
  if (tape->IsReady()) response->MoveFrom(tape.get());
  return Status::OK();
}

// ---------------------------------------------------------------------- dataset --
Dataset::Dataset(Client* client, int32_t dag_id)
    : client_(client), dag_id_(dag_id), cap_(std::max(1, GLOBAL_FLAG(DatasetCapacity))) {  // dag_dataset.cc:28-48
  worker_ = std::thread(&Dataset::Prefetch, this);
}

Dataset::~Dataset() {
  Close();
  for (GetDagValuesResponse* res : buffer_) delete res;
}

void Dataset::Close() {
  closed_ = true;
  {
    std::lock_guard<std::mutex> g(mtx_);
    room_.notify_all();
    data_.notify_all();
  }
  if (worker_.joinable()) worker_.join();
}

void Dataset::Prefetch() {  // dag_dataset.cc:95-119
  const std::function<bool()> cancelled = [this]() { return closed_.load(); };
  while (!closed_) {
    {
      std::unique_lock<std::mutex> lock(mtx_);
      while (!closed_ && (int32_t)buffer_.size() >= cap_) room_.wait_for(lock, kPoll);
    }
    if (closed_) break;
    GetDagValuesRequest req(dag_id_);
    std::unique_ptr<GetDagValuesResponse> res(new GetDagValuesResponse);
    if (!client_->GetDagValues(&req, res.get(), &cancelled).ok()) break;  // stopped under us: Next() gives nullptr
    std::lock_guard<std::mutex> g(mtx_);
    buffer_.push_back(res.release());
    data_.notify_one();
  }
  std::lock_guard<std::mutex> g(mtx_);
  closed_ = true;
  data_.notify_all();
}

GetDagValuesResponse* Dataset::Next(int32_t epoch) {  // dag_dataset.cc:63-93
  std::unique_lock<std::mutex> lock(mtx_);
  while (!closed_ && buffer_.empty()) {
    data_.wait_for(lock, kPoll);  // the reference logs "Query timeout" every Timeout seconds and keeps waiting
  }
  if (closed_) return nullptr;
  GetDagValuesResponse* res = buffer_.front();
  if (epoch < res->Epoch()) return nullptr;  // another consumer's epoch change: leave it for the next call
  buffer_.pop_front();
  room_.notify_one();
  return res;
}

}  // namespace graphlearn
