// GSL queries as the engine sees them: a DAG of operator nodes that is run round after round in the background,
// each round's node outputs recorded on a tape that a Dataset hands to the caller.  Mirrors, for local deployments,
//   generated/proto/dag.proto (DagDef / DagNodeDef / DagEdgeDef; plain structs here -- nothing is serialised),
//   core/dag/{dag,dag_node,dag_edge,tensor_map,tape}.h, core/runner/dag_node_runner.h:30-60,
//   core/runner/dag_scheduler.h:27-50, include/dag_request.h:30-98, include/dag_dataset.h:28-56
// of graphlearn/src, with the observable behaviour of their .cc files (cited at each definition in src/dag.cc):
// batches come back in the order the root produced them, a round in which any node fails or the root runs out of
// ids is a "faked" tape (an invalid response: the caller's end-of-epoch), epochs are counted by the faked tapes.
//
// What is different, and why.  The reference schedules every node of every round as a task on a 32-thread pool
// and lets up to TapeCapacity rounds overlap, because its operators are CPU loops.  Here an operator is a handful
// of kernel launches on one GPU: node-level host parallelism buys nothing and costs the ordering of the launches.
// So a query is COMPILED once (Dag::Compile) into a topologically ordered list of steps, and one scheduler thread
// per query runs the steps of a round back to back.  A step is one node, or a chain of dense sampling hops
// (`.outV(e1).sample(k1).by(s).outV(e2).sample(k2).by(s)`) lowered to ONE glx_sample_hops call in which the
// frontier never leaves the GPU between hops.  Rounds still run ahead of the consumer, bounded
// by the tape store's capacity, which is what hides the query's latency from the training loop.
#ifndef GLX_HOST_DAG_H_
#define GLX_HOST_DAG_H_
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "graphlearn/op_request.h"
#include "graphlearn/operator.h"
#include "graphlearn/status.h"
#include "graphlearn/tensor.h"

namespace graphlearn {

class Client;
class Env;

// ------------------------------------------------------------------ definitions --
// What the Python layer assembles through pywrap.new_dag / new_dag_node / new_dag_edge / add_dag_node_*
// (python/c/py_wrapper.h:34-130): dag.proto's messages.
struct DagEdgeDef {
  int32_t id = 0;
  std::string src_output;  // the tensor of the source node's response that travels ...
  std::string dst_input;   // ... and the input name it arrives under at the destination
};

struct DagNodeDef {
  int32_t id = 0;  // 1-based position in the query; the sink is last
  std::string op_name;
  Tensor::Map params;
  std::vector<DagEdgeDef> in_edges, out_edges;
};

struct DagDef {
  int32_t id = 0;
  std::vector<DagNodeDef> nodes;
  std::string DebugString() const;
};

// ----------------------------------------------------------------------- graph --
class DagNode;

class DagEdge {
public:
  explicit DagEdge(const DagEdgeDef& def) : id_(def.id), src_output_(def.src_output), dst_input_(def.dst_input) {}
  int32_t Id() const { return id_; }
  const std::string& SrcOutput() const { return src_output_; }
  const std::string& DstInput() const { return dst_input_; }
  DagNode* Src() const { return src_; }
  DagNode* Dst() const { return dst_; }
  void SetSrc(DagNode* n) { src_ = n; }
  void SetDst(DagNode* n) { dst_ = n; }

private:
  int32_t id_;
  std::string src_output_, dst_input_;
  DagNode* src_ = nullptr;
  DagNode* dst_ = nullptr;
};

class DagNode {
public:
  DagNode(const DagNodeDef& def, std::unordered_map<int32_t, std::shared_ptr<DagEdge>>* edges);
  int32_t Id() const { return id_; }
  const std::string& OpName() const { return op_name_; }
  const Tensor::Map& Params() const { return params_; }
  const std::vector<std::shared_ptr<DagEdge>>& InEdges() const { return in_edges_; }
  const std::vector<std::shared_ptr<DagEdge>>& OutEdges() const { return out_edges_; }
  int32_t InDegree() const { return (int32_t)in_edges_.size(); }
  bool IsSink() const { return op_name_ == "Sink"; }

private:
  int32_t id_;
  std::string op_name_;
  Tensor::Map params_;
  std::vector<std::shared_ptr<DagEdge>> in_edges_, out_edges_;
};

class Dag {
public:
  // One unit of a round: the nodes run by one operator call.  More than one node = a chain of dense sampling
  // hops, each fed by the previous one's neighbour ids and by nothing else.
  struct Step {
    std::vector<const DagNode*> nodes;
  };

  explicit Dag(const DagDef& def);
  int32_t Id() const { return id_; }
  int32_t Size() const { return (int32_t)nodes_.size(); }
  const DagNode* Root() const { return root_; }
  const std::vector<std::unique_ptr<DagNode>>& Nodes() const { return nodes_; }
  const std::string& DebugString() const { return debug_; }
  // Kahn order from the root + hop fusion.  Not OK: a cycle, an edge without one of its ends, no root.
  Status Compile();
  const std::vector<Step>& Steps() const { return steps_; }

private:
  int32_t id_;
  std::string debug_;
  std::vector<std::unique_ptr<DagNode>> nodes_;
  std::unordered_map<int32_t, std::shared_ptr<DagEdge>> edges_;
  const DagNode* root_ = nullptr;
  std::vector<Step> steps_;
};

class DagFactory {
public:
  static DagFactory* GetInstance();
  Status Create(const DagDef& def, Dag** dag);  // AlreadyExists for a known id (core/dag/dag.cc:52-62)
  Dag* Lookup(int32_t dag_id);
  void Clear();  // a stopped server's queries go with it

private:
  std::mutex mtx_;
  std::unordered_map<int32_t, std::unique_ptr<Dag>> map_;
};

// ------------------------------------------------------------------------ tapes --
// One node's recorded output: the response's tensors, dense and ragged (core/dag/tensor_map.h).
class TensorMap {
public:
  int32_t Size() const { return (int32_t)(tensors_.size() + sparse_tensors_.size()); }
  // {values, segments}: segments == nullptr for a dense tensor, both nullptr for an unknown key
  std::pair<const Tensor*, const Tensor*> Find(const std::string& key) const;
  bool Add(const std::string& key, const Tensor* values, const Tensor* segments);

  Tensor::Map tensors_;
  SparseTensor::Map sparse_tensors_;
};

class Tape {
public:
  explicit Tape(const Dag* dag) : recordings_(dag->Size()) {}
  int32_t Size() const { return (int32_t)recordings_.size(); }
  void Record(int32_t node_id, TensorMap&& tensors) { recordings_[node_id - 1] = std::move(tensors); }
  TensorMap& Retrieval(int32_t node_id) { return recordings_[node_id - 1]; }
  void SetReady() { ready_ = true; }
  void Fake() {  // tape.cc:77-81: a failed round keeps nothing
    recordings_.clear();
    faked_ = true;
  }
  bool IsReady() const { return ready_; }
  bool IsFaked() const { return faked_; }
  void SetId(int32_t id) { id_ = id; }
  void SetEpoch(int32_t epoch) { epoch_ = epoch; }
  int32_t Id() const { return id_; }
  int32_t Epoch() const { return epoch_; }

private:
  int32_t id_ = -1, epoch_ = -1;
  bool faked_ = false, ready_ = false;
  std::vector<TensorMap> recordings_;  // node i at i - 1
};

// Bounded FIFO between a query's scheduler and its consumers (core/dag/tape.h:104-143).
class TapeStore {
public:
  TapeStore(int32_t capacity, const Dag* dag);
  Tape* New() { return new Tape(dag_); }
  // Stamps the tape with the current epoch -- a faked tape ends it -- and queues it once there is room.
  // false: `stop` became true while waiting (the tape is deleted).
  bool WaitAndPush(Tape* tape, const std::function<bool()>& stop);
  // The oldest tape, numbered in pop order per client.  nullptr: `stop` became true while waiting.
  Tape* WaitAndPop(int32_t client_id, const std::function<bool()>& stop);
  // Ends the store for good: wakes every waiter, a producer's push fails from now on, a consumer drains what is queued
  // and then gets nullptr -- whenever it looks, not only while the stop that closed the store is still in progress.
  void Close();
  ~TapeStore();

private:
  const int32_t cap_;
  const Dag* dag_;
  int32_t epoch_ = 0;
  bool closed_ = false;  // under mtx_
  std::mutex mtx_;
  std::condition_variable room_, data_;
  std::deque<Tape*> queue_;
  std::unordered_map<int32_t, int32_t> tape_indexes_;
};
typedef std::shared_ptr<TapeStore> TapeStorePtr;
TapeStorePtr GetTapeStore(int32_t dag_id);  // nullptr for an unknown query (tape.cc:150-168)

// ---------------------------------------------------------------------- running --
// Operators whose consecutive hops can run as one device call implement this next to Operator::Process.
// requests[0] carries the seeds; requests[h > 0] only their parameters (edge type, neighbour count): hop h's
// source ids are hop h - 1's neighbour ids and never visit the host in between.
namespace op {
class HopFusable {
public:
  virtual ~HopFusable() = default;
  virtual Status ProcessHops(const std::vector<const OpRequest*>& requests, const std::vector<OpResponse*>& responses) = 0;
};
}  // namespace op

class DagNodeRunner {
public:
  explicit DagNodeRunner(Env* env) : env_(env) {}
  // Runs one step of a round: builds each node's request from its parameters and its in-edges' tensors
  // (dag_node_runner.cc:54-70, 100-109), runs the operator, records the response on the tape -- or fakes the
  // tape: an input is missing, the operator is unknown or failed, the root ran out of ids.
  void Run(const Dag::Step& step, Tape* tape);

private:
  bool BuildInput(const DagNode* node, Tape* tape, TensorMap* tensors);
  // tensors == nullptr: the parameters only (a later hop of a fused chain)
  std::unique_ptr<OpRequest> MakeOpRequest(const DagNode* node, const TensorMap* tensors);
  void RunNode(const DagNode* node, Tape* tape);
  void RunHops(const Dag::Step& step, Tape* tape);
  Env* env_;
};

// One background thread per running query (dag_scheduler.cc:27-91).
class DagScheduler {
public:
  static void Take(Env* env, const Dag* dag);
  // Stops and joins every query thread, closes their tape stores and forgets the queries: what Server::Stop does
  // to the reference's (env->IsStopping()).  Safe to call with nothing running.
  static void StopAll();
};

// --------------------------------------------------------- requests / responses --
class DagRequest {  // include/dag_request.h:33-45
public:
  bool ParseFrom(DagDef* def, bool copy = false);  // moves the definition out of *def unless copy
  std::string Name() const { return "DagRequest"; }
  DagDef def_;
};

class GetDagValuesRequest {  // :47-69
public:
  GetDagValuesRequest();
  explicit GetDagValuesRequest(int32_t dag_id);
  GetDagValuesRequest(int32_t dag_id, int32_t client_id) : id_(dag_id), client_id_(client_id) {}
  std::string Name() const { return "GetDagValuesRequest"; }
  int32_t Id() const { return id_; }
  int32_t ClientId() const { return client_id_; }

private:
  int32_t id_, client_id_;
};

class GetDagValuesResponse {  // :71-96
public:
  void MoveFrom(Tape* tape);
  std::pair<const Tensor*, const Tensor*> GetValue(int32_t node_id, const std::string& key) const;
  void SetIndex(int32_t index) { index_ = index; }
  void SetEpoch(int32_t epoch) { epoch_ = epoch; }
  int32_t Index() const { return index_; }
  int32_t Epoch() const { return epoch_; }
  bool Valid() const { return !records_.empty(); }

  std::unordered_map<int32_t, TensorMap> records_;

private:
  int32_t epoch_ = -1, index_ = -1;
};

// The consumer's end (include/dag_dataset.h, core/dag/dag_dataset.cc): keeps DatasetCapacity responses of one
// query prefetched.  The reference prefetches from `capacity` pool threads and re-orders by response index;
// one prefetch thread gets the same order without the bookkeeping.
class Dataset {
public:
  Dataset(Client* client, int32_t dag_id);
  ~Dataset();
  void Close();
  // The next response, handed over to the caller (delete it: pywrap.del_get_dag_value_response).  An invalid one
  // (Valid() == false) is the end of an epoch.  nullptr: the next response belongs to a later epoch than `epoch`
  // (it stays queued, dag_dataset.cc:78-83), or the dataset was closed.
  GetDagValuesResponse* Next(int32_t epoch);

private:
  void Prefetch();
  Client* client_;
  const int32_t dag_id_, cap_;
  std::mutex mtx_;
  std::condition_variable room_, data_;
  std::deque<GetDagValuesResponse*> buffer_;
  std::atomic<bool> closed_{false};
  std::thread worker_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_DAG_H_
