// pywrap_graphlearn: the pybind11 surface the reference's Python layer is written
// against (graphlearn/python/c/py_export.cc:36-277, py_client.cc:36-627), bound to the
// glx host mirror (libglx_host.so -> libglx.so -> HIP).  EVERY name the reference's Python
// tree uses is here (tests/test_refpy_names.py greps them), so `graphlearn/__init__.py` +
// `graphlearn/python/` of the reference run on this module unchanged in local deploy mode:
// Graph.init(), the samplers, GSL queries (the DAG API + Dataset, see host dag.h).  What the
// engine does not have -- RPC clients / servers, KNN, vineyard, the actor engine -- exists by
// name and fails when CALLED (flag setters of those layers just store their value).
// New: set_sampling_seed / set_device_id (the glx seeding contract and GPU placement).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <type_traits>

#include "graphlearn/graphlearn.h"

namespace py = pybind11;
using namespace graphlearn;  // NOLINT

namespace {

template <class T>
py::array_t<T> CopyOut(const T* data, size_t n) {
  py::array_t<T> out(n);
  if (n > 0 && data) std::memcpy(out.mutable_data(), data, n * sizeof(T));
  return out;
}

// Zero-copy hand-off of a response block: the numpy array keeps the tensor's storage
// alive (Tensor::Owner), so it stays valid after del_op_response; the block returns to
// the host layer's pool when the array dies.
template <class T>
py::array_t<T> ViewOf(const OpResponse* res, const char* key, size_t n) {
  auto it = res->tensors_.find(key);
  if (it == res->tensors_.end() || n == 0) return py::array_t<T>(0);
  const Tensor& t = it->second;
  const T* data = nullptr;
  if (std::is_same<T, int64_t>::value) data = reinterpret_cast<const T*>(t.GetInt64());
  else if (std::is_same<T, int32_t>::value) data = reinterpret_cast<const T*>(t.GetInt32());
  else data = reinterpret_cast<const T*>(t.GetFloat());
  auto* keep = new std::shared_ptr<const void>(t.Owner());
  py::capsule owner(keep, [](void* p) { delete static_cast<std::shared_ptr<const void>*>(p); });
  return py::array_t<T>({n}, {sizeof(T)}, data, owner);
}

// Requests / responses cross the boundary as base-class handles; a handle of the wrong kind is a
// Python TypeError, not a crash.
template <class T, class Base>
T* As(Base* p, const char* what) {
  T* t = dynamic_cast<T*>(p);
  if (!t) throw py::type_error(std::string("expected a ") + what);
  return t;
}

// A whole tensor as a 1-d numpy array over its storage (strings: an object array of bytes, py_wrapper.h:668-679).
py::object TensorArray(const Tensor& t) {
  const size_t n = (size_t)t.Size();
  if (t.DType() == kString) {
    py::list out;
    for (size_t i = 0; i < n; ++i) out.append(py::bytes(t.GetString((int32_t)i)));
    py::object arr = py::module_::import("numpy").attr("empty")(n, py::arg("dtype") = "object");
    for (size_t i = 0; i < n; ++i) arr[py::int_(i)] = out[i];
    return arr;
  }
  auto* keep = new std::shared_ptr<const void>(t.Owner());
  py::capsule owner(keep, [](void* p) { delete static_cast<std::shared_ptr<const void>*>(p); });
  switch (t.DType()) {
    case kInt32: return py::array_t<int32_t>({n}, {sizeof(int32_t)}, t.GetInt32(), owner);
    case kInt64: return py::array_t<int64_t>({n}, {sizeof(int64_t)}, t.GetInt64(), owner);
    case kFloat: return py::array_t<float>({n}, {sizeof(float)}, t.GetFloat(), owner);
    case kDouble: return py::array_t<double>({n}, {sizeof(double)}, t.GetDouble(), owner);
    default: return py::none();
  }
}

typedef py::array_t<int64_t, py::array::c_style | py::array::forcecast> I64Array;
typedef py::array_t<int32_t, py::array::c_style | py::array::forcecast> I32Array;

}  // namespace

PYBIND11_MODULE(pywrap_graphlearn, m) {
  m.doc() = "glx: MI355X-native engine behind graph-learn's pywrap_graphlearn interface (hot path only)";

  py::enum_<DeployMode>(m, "DeployMode").value("LOCAL", kLocal).value("SERVER", kServer).value("WORKER", kWorker);

  // ---- global flags (py_export.cc:38-73) ----
  m.def("set_default_neighbor_id", &SetGlobalFlagDefaultNeighborId);
  m.def("set_padding_mode", &SetGlobalFlagPaddingMode);
  m.def("set_default_int_attr", &SetGlobalFlagDefaultIntAttribute);
  m.def("set_default_float_attr", &SetGlobalFlagDefaultFloatAttribute);
  m.def("set_default_string_attr", &SetGlobalFlagDefaultStringAttribute);
  m.def("set_default_weight", &SetGlobalFlagDefaultWeight);
  m.def("set_default_label", &SetGlobalFlagDefaultLabel);
  m.def("set_default_timestamp", &SetGlobalFlagDefaultTimestamp);
  m.def("set_ignore_invalid", &SetGlobalFlagIgnoreInvalid);
  m.def("set_sampler_retry_times", &SetGlobalFlagSamplingRetryTimes);
  m.def("set_default_full_nbr_num", &SetGlobalFlagDefaultFullNbrNum);
  m.def("set_shuffle_buffer_size", &SetGlobalFlagShuffleBufferSize);
  m.def("set_sampling_seed", &SetGlobalFlagSamplingSeed);
  m.def("set_device_id", &SetGlobalFlagDeviceId);
  m.def("set_deploy_mode", [](int32_t mode) { SetGlobalFlagDeployMode(mode); });
  m.def("set_deploy_mode", [](DeployMode mode) { SetGlobalFlagDeployMode((int32_t)mode); });
  m.def("set_client_id", &SetGlobalFlagClientId);
  m.def("set_client_count", &SetGlobalFlagClientCount);
  m.def("set_server_count", &SetGlobalFlagServerCount);
  m.def("set_timeout", &SetGlobalFlagTimeout);
  m.def("set_tape_capacity", &SetGlobalFlagTapeCapacity);
  m.def("set_dataset_capacity", &SetGlobalFlagDatasetCapacity);
  m.def("set_tracker_mode", &SetGlobalFlagTrackerMode);
  m.def("get_tracker_mode", &GetGlobalFlagTrackerMode);
  // flags of layers this engine does not have (thread pools, queues, storage layout, RPC, KNN, vineyard, actors):
  // stored under their name, never read (config.h)
#define GLX_UNUSED_INT_FLAG(pyname) m.def(pyname, [](int64_t v) { SetGlobalFlagUnused(pyname, v); })
#define GLX_UNUSED_STR_FLAG(pyname) m.def(pyname, [](const std::string& v) { SetGlobalFlagUnused(pyname, v); })
  GLX_UNUSED_INT_FLAG("set_inter_threadnum");
  GLX_UNUSED_INT_FLAG("set_inner_threadnum");
  GLX_UNUSED_INT_FLAG("set_intra_threadnum");
  GLX_UNUSED_INT_FLAG("set_datainit_batchsize");
  GLX_UNUSED_INT_FLAG("set_inmemory_queuesize");
  GLX_UNUSED_INT_FLAG("set_storage_mode");
  GLX_UNUSED_INT_FLAG("set_retry_times");
  GLX_UNUSED_INT_FLAG("set_rpc_message_max_size");
  GLX_UNUSED_INT_FLAG("set_knn_metric");
  GLX_UNUSED_INT_FLAG("set_local_node_cache_capacity");
  GLX_UNUSED_INT_FLAG("set_enable_actor");
  GLX_UNUSED_INT_FLAG("set_actor_local_shard_count");
  GLX_UNUSED_INT_FLAG("set_vineyard_graph_id");
  GLX_UNUSED_STR_FLAG("set_tracker");
  GLX_UNUSED_STR_FLAG("set_server_hosts");
  GLX_UNUSED_STR_FLAG("set_field_delimiter");
  GLX_UNUSED_STR_FLAG("set_vineyard_ipc_socket");
#undef GLX_UNUSED_INT_FLAG
#undef GLX_UNUSED_STR_FLAG

  // ---- tensor / parameter keys (py_export.cc:82-131) ----
#define GLX_KEY(name, wire) m.attr(#name) = name;
  GLX_TENSOR_KEYS(GLX_KEY)  // the whole table of constants.h (a superset of what the reference exports)
#undef GLX_KEY

  py::enum_<error::Code>(m, "ErrorCode")
      .value("OK", error::OK)
      .value("CANCELLED", error::CANCELLED)
      .value("UNKNOWN", error::UNKNOWN)
      .value("INVALID_ARGUMENT", error::INVALID_ARGUMENT)
      .value("DEADLINE_EXCEEDED", error::DEADLINE_EXCEEDED)
      .value("NOT_FOUND", error::NOT_FOUND)
      .value("ALREADY_EXISTS", error::ALREADY_EXISTS)
      .value("PERMISSION_DENIED", error::PERMISSION_DENIED)
      .value("UNAUTHENTICATED", error::UNAUTHENTICATED)
      .value("RESOURCE_EXHAUSTED", error::RESOURCE_EXHAUSTED)
      .value("FAILED_PRECONDITION", error::FAILED_PRECONDITION)
      .value("ABORTED", error::ABORTED)
      .value("OUT_OF_RANGE", error::OUT_OF_RANGE)
      .value("UNIMPLEMENTED", error::UNIMPLEMENTED)
      .value("INTERNAL", error::INTERNAL)
      .value("UNAVAILABLE", error::UNAVAILABLE)
      .value("DATA_LOSS", error::DATA_LOSS)
      .value("REQUEST_STOP", error::REQUEST_STOP);

  py::enum_<DataType>(m, "DataType")
      .value("INT32", kInt32)
      .value("INT64", kInt64)
      .value("FLOAT", kFloat)
      .value("DOUBLE", kDouble)
      .value("STRING", kString);

  py::enum_<PaddingMode>(m, "PaddingMode").value("REPLICATE", kReplicate).value("CIRCULAR", kCircular);
  py::enum_<PartitionMode>(m, "PartitionMode").value("NO_PARTITION", kNoPartition).value("BY_SOURCE_ID", kByHash);
  py::enum_<TrackerMode>(m, "TrackerMode").value("RPC", kRpc).value("FILE_SYSTEM", kFileSystem);

  py::enum_<FilterType>(m, "FilterType")
      .value("OPERATOR_UNSPECIFIED", kOperatorUnspecified)
      .value("EQUAL", kEqual)
      .value("LARGER_THAN", kLargerThan);

  py::enum_<FilterField>(m, "FilterField")
      .value("FIELD_UNSPECIFIED", kFieldUnspecified)
      .value("ID", kId)
      .value("TIMESTAMP", kTimestamp);

  py::enum_<io::DataFormat>(m, "DataFormat")
      .value("DEFAULT", io::kDefault)
      .value("WEIGHTED", io::kWeighted)
      .value("LABELED", io::kLabeled)
      .value("ATTRIBUTED", io::kAttributed)
      .value("TIMESTAMPED", io::kTimestamped);

  py::enum_<NodeFrom>(m, "NodeFrom")
      .value("EDGE_SRC", kEdgeSrc)
      .value("EDGE_DST", kEdgeDst)
      .value("NODE", kNode);

  py::enum_<io::Direction>(m, "Direction").value("ORIGIN", io::kOrigin).value("REVERSED", io::kReversed);

  py::class_<IndexOption>(m, "IndexOption")
      .def(py::init<>())
      .def_readwrite("name", &IndexOption::name)
      .def_readwrite("index_type", &IndexOption::index_type)
      .def_readwrite("dimension", &IndexOption::dimension)
      .def_readwrite("nlist", &IndexOption::nlist)
      .def_readwrite("nprobe", &IndexOption::nprobe)
      .def_readwrite("m", &IndexOption::m);

  py::class_<io::AttributeInfo>(m, "AttributeInfo")
      .def(py::init<>())
      .def_readwrite("delimiter", &io::AttributeInfo::delimiter)
      .def_readwrite("ignore_invalid", &io::AttributeInfo::ignore_invalid)
      .def("append_type", &io::AttributeInfo::AppendType)
      .def("append_hash_bucket", &io::AttributeInfo::AppendHashBucket);

  py::class_<io::NodeSource>(m, "NodeSource")
      .def(py::init<>())
      .def_readwrite("path", &io::NodeSource::path)
      .def_readwrite("id_type", &io::NodeSource::id_type)
      .def_readwrite("format", &io::NodeSource::format)
      .def_readwrite("attr_info", &io::NodeSource::attr_info)
      .def_readwrite("option", &io::NodeSource::option)
      .def_readwrite("view_type", &io::NodeSource::view_type)
      .def_readwrite("use_attrs", &io::NodeSource::use_attrs);

  py::class_<io::EdgeSource>(m, "EdgeSource")
      .def(py::init<>())
      .def_readwrite("path", &io::EdgeSource::path)
      .def_readwrite("edge_type", &io::EdgeSource::edge_type)
      .def_readwrite("src_id_type", &io::EdgeSource::src_id_type)
      .def_readwrite("dst_id_type", &io::EdgeSource::dst_id_type)
      .def_readwrite("format", &io::EdgeSource::format)
      .def_readwrite("direction", &io::EdgeSource::direction)
      .def_readwrite("attr_info", &io::EdgeSource::attr_info)
      .def_readwrite("option", &io::EdgeSource::option)
      .def_readwrite("view_type", &io::EdgeSource::view_type)
      .def_readwrite("use_attrs", &io::EdgeSource::use_attrs);

  py::class_<Status>(m, "Status")
      .def("ok", &Status::ok)
      .def("code", &Status::code)
      .def("message", &Status::msg)
      .def("to_string", &Status::ToString);

  py::class_<Server>(m, "Server")
      .def("start", &Server::Start)
      .def("init", &Server::Init, py::call_guard<py::gil_scoped_release>())
      .def("init_status", &Server::InitStatus)
      .def("device_graph", &Server::DeviceGraph)
      .def("node_counts", [](Server& self) { return self.Store()->NodeCounts(); })
      .def("edge_counts", [](Server& self) { return self.Store()->EdgeCounts(); })
      // id lists of the store for the batch-traversal samplers (node_generator.h / edge_generator.h)
      .def("node_ids", [](Server& self, const std::string& node_type) {
        const std::vector<int64_t>& ids = self.Store()->GetNoder(node_type)->Ids();
        return CopyOut(ids.data(), ids.size());
      })
      .def("edge_src_ids", [](Server& self, const std::string& edge_type) {
        const std::vector<int64_t>& ids = self.Store()->GetGraph(edge_type)->SrcIds();
        return CopyOut(ids.data(), ids.size());
      })
      .def("edge_dst_ids", [](Server& self, const std::string& edge_type) {
        const std::vector<int64_t>& ids = self.Store()->GetGraph(edge_type)->DstIds();
        return CopyOut(ids.data(), ids.size());
      })
      .def("device_features", &Server::DeviceFeatures)
      .def("stop", &Server::Stop, py::call_guard<py::gil_scoped_release>())
      .def("stop_sampling", [](Server&) { DagScheduler::StopAll(); }, py::call_guard<py::gil_scoped_release>())
      .def("get_stats", [](Server& self) { return self.Store() ? self.Store()->GetStatistics().GetCounts() : Counts(); });
  m.def("server", &NewServer, py::return_value_policy::take_ownership, py::arg("server_id"),
        py::arg("server_count"), py::arg("server_host"), py::arg("tracker"));

  py::class_<OpRequest>(m, "OpRequest");
  py::class_<OpResponse>(m, "OpResponse");
  m.def("del_op_request", [](OpRequest* req) { delete req; });
  m.def("del_op_response", [](OpResponse* res) { delete res; });

  // ---- client (py_client.cc:40-125) ----
  py::class_<Client>(m, "Client")
      .def("stop", &Client::Stop)
      .def("sample_neighbor",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.Sampling(As<SamplingRequest>(req, "SamplingRequest"), As<SamplingResponse>(res, "SamplingResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("agg_nodes",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.Aggregating(As<AggregatingRequest>(req, "AggregatingRequest"), As<AggregatingResponse>(res, "AggregatingResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("lookup_nodes",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.LookupNodes(As<LookupNodesRequest>(req, "LookupNodesRequest"), As<LookupResponse>(res, "LookupResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("lookup_edges",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.LookupEdges(As<LookupEdgesRequest>(req, "LookupEdgesRequest"), As<LookupResponse>(res, "LookupResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("get_degree",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.GetDegree(As<GetDegreeRequest>(req, "GetDegreeRequest"), As<GetDegreeResponse>(res, "GetDegreeResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("cond_neg_sample",
           [](Client& self, OpRequest* req, OpResponse* res) {
             As<ConditionalSamplingRequest>(req, "ConditionalSamplingRequest");
             return self.RunOp(req, As<SamplingResponse>(res, "SamplingResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("sample_subgraph",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.SubGraph(As<SubGraphRequest>(req, "SubGraphRequest"), As<SubGraphResponse>(res, "SubGraphResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("get_stats",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.GetStats(As<GetStatsRequest>(req, "GetStatsRequest"), As<GetStatsResponse>(res, "GetStatsResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("get_count",
           [](Client& self, OpRequest* req, OpResponse* res) {
             return self.GetCount(As<GetCountRequest>(req, "GetCountRequest"), As<GetCountResponse>(res, "GetCountResponse"));
           },
           py::call_guard<py::gil_scoped_release>())
      .def("get_nodes", [](Client& self, OpRequest* req, OpResponse* res) { return self.RunOp(req, res); })
      .def("get_edges", [](Client& self, OpRequest* req, OpResponse* res) { return self.RunOp(req, res); })
      .def("run_op", &Client::RunOp, py::call_guard<py::gil_scoped_release>())
      // GSL queries (py_client.cc:105-115)
      .def("run_dag",
           [](Client& self, DagDef* dag_def, bool copy) {
             DagRequest req;
             req.ParseFrom(dag_def, copy);
             return self.RunDag(&req);
           },
           py::arg("dag_def"), py::arg("copy") = false)
      .def("get_dag_values",
           [](Client& self, GetDagValuesRequest* req, GetDagValuesResponse* res) { return self.GetDagValues(req, res); },
           py::arg("request"), py::arg("response"), py::call_guard<py::gil_scoped_release>())
      .def("get_own_servers", [](Client&) { return std::vector<int32_t>{0}; });
  m.def("in_memory_client", &NewInMemoryClient, py::return_value_policy::take_ownership);
  m.def("rpc_client",
        [](int32_t, bool) -> Client* {
          throw std::runtime_error(
              "rpc_client: this engine has no RPC service -- its servers are the GPUs of one node, reached through "
              "the in-memory client (local deploy mode) or the RCCL shard communicator (graph-learn_amd/dist.py)");
        },
        py::arg("server_id") = -1, py::arg("client_own") = true);
  // KNN (python/operator/knn_operator.py): no KNN operator here; present by name, fails when called
  for (const char* name : {"new_knn_request", "new_knn_response", "set_knn_request", "get_knn_ids", "get_knn_distances"}) {
    m.def(name, [name](py::args) -> py::object {
      throw std::runtime_error(std::string(name) + ": the KNN operator (faiss) is not part of this engine");
    });
  }

  // ---- GSL: the DAG definition a query is lowered to (py_client.cc:527-627; py_wrapper.h:34-130) ----
  py::class_<DagDef>(m, "DagDef").def(py::init<>());
  py::class_<DagNodeDef>(m, "DagNodeDef").def(py::init<>());
  py::class_<DagEdgeDef>(m, "DagEdgeDef").def(py::init<>());
  m.def("new_dag", []() { return new DagDef(); }, py::return_value_policy::take_ownership);
  m.def("new_dag_node", []() { return new DagNodeDef(); }, py::return_value_policy::take_ownership);
  m.def("new_dag_edge", []() { return new DagEdgeDef(); }, py::return_value_policy::take_ownership);
  m.def("set_dag_id", [](DagDef* dag, int32_t dag_id) { dag->id = dag_id; });
  m.def("debug_string", [](DagDef* dag) { return dag->DebugString(); });
  m.def("add_dag_node", [](DagDef* dag, const DagNodeDef* node) { dag->nodes.push_back(*node); });
  m.def("set_dag_node_id", [](DagNodeDef* node, int32_t node_id) { node->id = node_id; });
  m.def("set_dag_node_op_name", [](DagNodeDef* node, const std::string& op_name) { node->op_name = op_name; });
  m.def("add_dag_node_in_edge", [](DagNodeDef* node, const DagEdgeDef* edge) { node->in_edges.push_back(*edge); });
  m.def("add_dag_node_out_edge", [](DagNodeDef* node, const DagEdgeDef* edge) { node->out_edges.push_back(*edge); });
  m.def("add_dag_node_int_params", [](DagNodeDef* node, const std::string& name, int32_t value) {
    Tensor t(kInt32, 1);
    t.AddInt32(value);
    node->params[name] = t;
  });
  m.def("add_dag_node_string_params", [](DagNodeDef* node, const std::string& name, const std::string& value) {
    Tensor t(kString, 1);
    t.AddString(value);
    node->params[name] = t;
  });
  m.def("add_dag_node_int_vector_params", [](DagNodeDef* node, const std::string& name, const std::vector<int32_t>& values) {
    Tensor t(kInt32, (int32_t)values.size());
    t.AddInt32(values.data(), values.data() + values.size());
    node->params[name] = t;
  });
  m.def("add_dag_node_float_vector_params", [](DagNodeDef* node, const std::string& name, const std::vector<float>& values) {
    Tensor t(kFloat, (int32_t)values.size());
    t.AddFloat(values.data(), values.data() + values.size());
    node->params[name] = t;
  });
  m.def("set_dag_edge_id", [](DagEdgeDef* edge, int32_t id) { edge->id = id; });
  m.def("set_dag_edge_src_output", [](DagEdgeDef* edge, const std::string& v) { edge->src_output = v; });
  m.def("set_dag_edge_dst_input", [](DagEdgeDef* edge, const std::string& v) { edge->dst_input = v; });

  // ---- GSL: a query's values (py_client.cc:492-525; py_wrapper.h:635-720) ----
  py::class_<Dataset>(m, "Dataset")
      .def(py::init<Client*, int32_t>(), py::keep_alive<1, 2>())
      .def("close", &Dataset::Close, py::call_guard<py::gil_scoped_release>())
      .def("next", &Dataset::Next, py::return_value_policy::reference, py::arg("epoch"),
           py::call_guard<py::gil_scoped_release>());
  py::class_<GetDagValuesRequest>(m, "GetDagValuesRequest").def(py::init<int32_t, int32_t>());
  py::class_<GetDagValuesResponse>(m, "GetDagValuesResponse")
      .def(py::init<>())
      .def("valid", &GetDagValuesResponse::Valid)
      .def("epoch", &GetDagValuesResponse::Epoch)
      .def("index", &GetDagValuesResponse::Index);
  m.def("del_get_dag_value_response", [](GetDagValuesResponse* res) { delete res; });
  // One node's tensor as a 1-d array, None when the node recorded nothing under `key`.  Zero-copy: the array keeps
  // the tensor's storage alive, so it also survives del_get_dag_value_response.
  m.def("get_dag_value", [](GetDagValuesResponse* res, int32_t node_id, const std::string& key) -> py::object {
    const Tensor* values = res->GetValue(node_id, key).first;
    if (!values) return py::none();
    return TensorArray(*values);
  });
  // The per-row counts of a ragged value (a FullSampler's output), None for a dense one.
  m.def("get_dag_value_indice", [](GetDagValuesResponse* res, int32_t node_id, const std::string& key) -> py::object {
    const Tensor* segments = res->GetValue(node_id, key).second;
    if (!segments || segments->Size() == 0) return py::none();
    return TensorArray(*segments);
  });

  // ---- sampling (py_client.cc:292-363) ----
  m.def("new_sampling_request",
        [](const std::string& type, const std::string& strategy, int32_t neighbor_count, FilterType filter_type,
           FilterField filter_field) -> OpRequest* {
          return new SamplingRequest(type, strategy, neighbor_count, filter_type, filter_field);
        },
        py::return_value_policy::reference);
  m.def("new_sampling_response", []() -> OpResponse* { return new SamplingResponse(); },
        py::return_value_policy::reference);
  // conditional negative sampling (py_client.cc:301-341; py_wrapper.h:376-419)
  m.def("new_conditional_sampling_request",
        [](const std::string& type, const std::string& strategy, int32_t neighbor_count, const std::string& dst_node_type,
           bool batch_share, bool unique) -> OpRequest* {
          return new ConditionalSamplingRequest(type, strategy, neighbor_count, dst_node_type, batch_share, unique);
        },
        py::return_value_policy::reference);
  m.def("set_conditional_sampling_request_ids", [](OpRequest* req, I64Array src_ids, I64Array dst_ids) {
    if (src_ids.size() != dst_ids.size()) throw std::invalid_argument("src_ids and dst_ids must have the same size");
    As<ConditionalSamplingRequest>(req, "ConditionalSamplingRequest")->SetIds(src_ids.data(), dst_ids.data(), (int32_t)src_ids.size());
  });
  m.def("set_conditional_sampling_request_cols",
        [](OpRequest* req, const std::vector<int32_t>& int_cols, const std::vector<float>& int_props,
           const std::vector<int32_t>& float_cols, const std::vector<float>& float_props,
           const std::vector<int32_t>& str_cols, const std::vector<float>& str_props) {
          As<ConditionalSamplingRequest>(req, "ConditionalSamplingRequest")
              ->SetSelectedCols(int_cols, int_props, float_cols, float_props, str_cols, str_props);
        });
  m.def("set_sampling_request", [](OpRequest* req, I64Array src_ids) {
    As<SamplingRequest>(req, "SamplingRequest")->Set(src_ids.data(), (int32_t)src_ids.size());
  });
  // glx addition: the reference hands filter values over only inside a DAG's tensor map
  // (SamplingRequest::Set(tensors) -> Filter::FillValues); this sets one value per src id.
  m.def("set_sampling_filter_values", [](OpRequest* req, I64Array values) {
    As<SamplingRequest>(req, "SamplingRequest")->SetFilterValues(values.data(), (int32_t)values.size());
  });
  m.def("set_sampling_call_counter", [](OpRequest* req, int64_t call_counter) {
    As<SamplingRequest>(req, "SamplingRequest")->SetCallCounter(call_counter);
  });
  m.def("get_sampling_node_ids", [](OpResponse* res) {
    SamplingResponse* r = As<SamplingResponse>(res, "SamplingResponse");
    return ViewOf<int64_t>(r, kNodeIds, r->GetShape().size);
  });
  m.def("get_sampling_edge_ids", [](OpResponse* res) {
    SamplingResponse* r = As<SamplingResponse>(res, "SamplingResponse");
    return ViewOf<int64_t>(r, kEdgeIds, r->GetShape().size);
  });
  m.def("get_sampling_node_degrees", [](OpResponse* res) {
    SamplingResponse* r = As<SamplingResponse>(res, "SamplingResponse");
    const Shape& shape = r->GetShape();
    return CopyOut(shape.segments.data(), shape.segments.size());
  });

  // ---- random walks (the reference reaches its RandomWalk operator through GSL only) ----
  m.def("new_random_walk_request",
        [](const std::string& type, float p, float q, int32_t walk_len) -> OpRequest* {
          return new RandomWalkRequest(type, p, q, walk_len);
        },
        py::return_value_policy::reference);
  m.def("new_random_walk_response", []() -> OpResponse* { return new RandomWalkResponse(); },
        py::return_value_policy::reference);
  m.def("set_random_walk_request", [](OpRequest* req, I64Array src_ids) {
    As<RandomWalkRequest>(req, "RandomWalkRequest")->Set(src_ids.data(), (int32_t)src_ids.size());
  });
  m.def("set_random_walk_call_counter", [](OpRequest* req, int64_t call_counter) {
    As<RandomWalkRequest>(req, "RandomWalkRequest")->SetCallCounter(call_counter);
  });
  m.def("get_random_walks", [](OpResponse* res) {
    RandomWalkResponse* r = As<RandomWalkResponse>(res, "RandomWalkResponse");
    return ViewOf<int64_t>(r, kNodeIds, (size_t)r->batch_size_ * (size_t)r->WalkLen());
  });

  // ---- aggregation (py_client.cc:365-391) ----
  m.def("new_aggregating_request",
        [](const std::string& node_type, const std::string& strategy) -> OpRequest* {
          return new AggregatingRequest(node_type, strategy);
        },
        py::return_value_policy::reference);
  m.def("new_aggregating_response", []() -> OpResponse* { return new AggregatingResponse(); },
        py::return_value_policy::reference);
  m.def("set_aggregating_request", [](OpRequest* req, I64Array node_ids, I32Array segment_ids, int32_t num_segments) {
    As<AggregatingRequest>(req, "AggregatingRequest")->Set(node_ids.data(), segment_ids.data(), (int32_t)node_ids.size(),
                                               num_segments);
  });
  m.def("get_aggregating_nodes", [](OpResponse* res) {
    AggregatingResponse* r = As<AggregatingResponse>(res, "AggregatingResponse");
    return ViewOf<float>(r, kFloatAttrKey, (size_t)r->NumSegments() * r->EmbeddingDim());
  });

  // ---- lookups (py_client.cc:151-290) ----
  m.def("new_lookup_nodes_request",
        [](const std::string& node_type) -> OpRequest* { return new LookupNodesRequest(node_type); },
        py::return_value_policy::reference);
  m.def("set_lookup_nodes_request", [](OpRequest* req, I64Array node_ids) {
    As<LookupNodesRequest>(req, "LookupNodesRequest")->Set(node_ids.data(), (int32_t)node_ids.size());
  });
  m.def("new_lookup_nodes_response", []() -> OpResponse* { return new LookupResponse(); },
        py::return_value_policy::reference);
  m.def("new_lookup_edges_request",
        [](const std::string& edge_type) -> OpRequest* { return new LookupEdgesRequest(edge_type); },
        py::return_value_policy::reference);
  m.def("set_lookup_edges_request", [](OpRequest* req, I64Array src_ids, I64Array edge_ids) {
    As<LookupEdgesRequest>(req, "LookupEdgesRequest")->Set(edge_ids.data(), src_ids.data(), (int32_t)edge_ids.size());
  });
  m.def("new_lookup_edges_response", []() -> OpResponse* { return new LookupResponse(); },
        py::return_value_policy::reference);
  for (const char* prefix : {"node", "edge"}) {
    const std::string p = std::string("get_") + prefix;
    m.def((p + "_weights").c_str(), [](OpResponse* res) {
      LookupResponse* r = As<LookupResponse>(res, "LookupResponse");
      return CopyOut(r->Weights(), r->Weights() ? (size_t)r->Size() : 0);
    });
    m.def((p + "_labels").c_str(), [](OpResponse* res) {
      LookupResponse* r = As<LookupResponse>(res, "LookupResponse");
      return CopyOut(r->Labels(), r->Labels() ? (size_t)r->Size() : 0);
    });
    m.def((p + "_timestamps").c_str(), [](OpResponse* res) {
      LookupResponse* r = As<LookupResponse>(res, "LookupResponse");
      return CopyOut(r->Timestamps(), r->Timestamps() ? (size_t)r->Size() : 0);
    });
    m.def((p + "_int_attributes").c_str(), [](OpResponse* res) {
      LookupResponse* r = As<LookupResponse>(res, "LookupResponse");
      return CopyOut(r->IntAttrs(), r->IntAttrs() ? (size_t)r->Size() * r->IntAttrNum() : 0);
    });
    m.def((p + "_float_attributes").c_str(), [](OpResponse* res) {
      LookupResponse* r = As<LookupResponse>(res, "LookupResponse");
      return ViewOf<float>(r, kFloatAttrKey, r->FloatAttrs() ? (size_t)r->Size() * r->FloatAttrNum() : 0);
    });
    m.def((p + "_string_attributes").c_str(), [](OpResponse* res) {
      LookupResponse* r = As<LookupResponse>(res, "LookupResponse");
      py::list out;
      for (const std::string& s : r->StringAttrs()) out.append(py::str(s));
      return py::array(py::module_::import("numpy").attr("array")(out, py::arg("dtype") = "object"));
    });
  }

  // ---- batch traversal (py_client.cc:131-149, 208-239) ----
  m.def("new_get_nodes_request",
        [](const std::string& type, const std::string& strategy, NodeFrom node_from, int32_t batch_size,
           int32_t epoch) -> OpRequest* { return new GetNodesRequest(type, strategy, node_from, batch_size, epoch); },
        py::return_value_policy::reference);
  m.def("new_get_nodes_response", []() -> OpResponse* { return new GetNodesResponse(); },
        py::return_value_policy::reference);
  m.def("get_node_ids", [](OpResponse* res) {
    GetNodesResponse* r = As<GetNodesResponse>(res, "GetNodesResponse");
    return CopyOut(r->NodeIds(), (size_t)r->Size());
  });
  m.def("new_get_edges_request",
        [](const std::string& edge_type, const std::string& strategy, int32_t batch_size, int32_t epoch) -> OpRequest* {
          return new GetEdgesRequest(edge_type, strategy, batch_size, epoch);
        },
        py::return_value_policy::reference);
  m.def("new_get_edges_response", []() -> OpResponse* { return new GetEdgesResponse(); },
        py::return_value_policy::reference);
  m.def("get_edge_src_id", [](OpResponse* res) {
    GetEdgesResponse* r = As<GetEdgesResponse>(res, "GetEdgesResponse");
    return CopyOut(r->SrcIds(), (size_t)r->Size());
  });
  m.def("get_edge_dst_id", [](OpResponse* res) {
    GetEdgesResponse* r = As<GetEdgesResponse>(res, "GetEdgesResponse");
    return CopyOut(r->DstIds(), (size_t)r->Size());
  });
  m.def("get_edge_id", [](OpResponse* res) {
    GetEdgesResponse* r = As<GetEdgesResponse>(res, "GetEdgesResponse");
    return CopyOut(r->EdgeIds(), (size_t)r->Size());
  });

  // ---- degrees (py_client.cc:468-491) ----
  m.def("new_get_degree_request",
        [](const std::string& edge_type, NodeFrom node_from) -> OpRequest* {
          return new GetDegreeRequest(edge_type, node_from);
        },
        py::return_value_policy::reference);
  m.def("set_degree_request", [](OpRequest* req, I64Array node_ids) {
    As<GetDegreeRequest>(req, "GetDegreeRequest")->Set(node_ids.data(), (int32_t)node_ids.size());
  });
  m.def("new_get_degree_response", []() -> OpResponse* { return new GetDegreeResponse(); },
        py::return_value_policy::reference);
  m.def("get_degree", [](OpResponse* res) {
    GetDegreeResponse* r = As<GetDegreeResponse>(res, "GetDegreeResponse");
    return CopyOut(r->GetDegrees(), (size_t)r->batch_size_);
  });

  // ---- sub-graph sampling (py_client.cc:393-449; py_wrapper.h:497-580) ----
  m.def("new_subgraph_request",
        [](const std::string& nbr_type, const std::vector<int32_t>& num_nbrs, bool need_dist) -> OpRequest* {
          return new SubGraphRequest(nbr_type, num_nbrs, need_dist);
        },
        py::return_value_policy::reference);
  m.def("new_subgraph_response", []() -> OpResponse* { return new SubGraphResponse(); }, py::return_value_policy::reference);
  m.def("set_subgraph_request", [](OpRequest* req, I64Array src_ids, py::object dst_ids) {
    SubGraphRequest* r = As<SubGraphRequest>(req, "SubGraphRequest");
    if (dst_ids.is_none()) {
      r->Set(src_ids.data(), (int32_t)src_ids.size());
    } else {
      I64Array dst = dst_ids.cast<I64Array>();
      if (dst.size() != src_ids.size()) throw std::invalid_argument("src_ids and dst_ids must have the same size");
      r->Set(src_ids.data(), dst.data(), (int32_t)src_ids.size());
    }
  });
  m.def("get_node_set", [](OpResponse* res) {
    SubGraphResponse* r = As<SubGraphResponse>(res, "SubGraphResponse");
    return CopyOut(r->NodeIds(), (size_t)r->NodeCount());
  });
  m.def("get_row_idx", [](OpResponse* res) {
    SubGraphResponse* r = As<SubGraphResponse>(res, "SubGraphResponse");
    return CopyOut(r->RowIndices(), (size_t)r->EdgeCount());
  });
  m.def("get_col_idx", [](OpResponse* res) {
    SubGraphResponse* r = As<SubGraphResponse>(res, "SubGraphResponse");
    return CopyOut(r->ColIndices(), (size_t)r->EdgeCount());
  });
  m.def("get_edge_set", [](OpResponse* res) {
    SubGraphResponse* r = As<SubGraphResponse>(res, "SubGraphResponse");
    return CopyOut(r->EdgeIds(), (size_t)r->EdgeCount());
  });
  m.def("get_dist_to_src", [](OpResponse* res) {
    SubGraphResponse* r = As<SubGraphResponse>(res, "SubGraphResponse");
    return CopyOut(r->DistToSrc(), (size_t)r->NodeCount());
  });
  m.def("get_dist_to_dst", [](OpResponse* res) {
    SubGraphResponse* r = As<SubGraphResponse>(res, "SubGraphResponse");
    return CopyOut(r->DistToDst(), (size_t)r->NodeCount());
  });

  // ---- statistics (py_client.cc:452-465; py_wrapper.h get_stats) ----
  m.def("new_get_stats_request", []() -> OpRequest* { return new GetStatsRequest(); }, py::return_value_policy::reference);
  m.def("new_get_stats_response", []() -> OpResponse* { return new GetStatsResponse(); }, py::return_value_policy::reference);
  m.def("get_stats", [](OpResponse* res) { return As<GetStatsResponse>(res, "GetStatsResponse")->GetCounts(); });
  m.def("new_get_count_request", []() -> OpRequest* { return new GetCountRequest(); }, py::return_value_policy::reference);
  m.def("new_get_count_response", []() -> OpResponse* { return new GetCountResponse(); }, py::return_value_policy::reference);
  m.def("get_count", [](OpResponse* res) {
    GetCountResponse* r = As<GetCountResponse>(res, "GetCountResponse");
    return CopyOut(r->Count(), (size_t)r->Size());
  });

  // ---- loader primitives exposed for the parity tests ----
  m.def("hash64", [](const py::bytes& b) {
    const std::string s = b;
    return io::Hash64(s.data(), s.size());
  });
  m.def("parse_attribute", [](const py::bytes& b, const io::AttributeInfo& info) {
    const std::string s = b;
    std::vector<int64_t> ints;
    std::vector<float> floats;
    std::vector<std::string> strings;
    Status st = io::ParseAttribute(s.data(), s.size(), info, &ints, &floats, &strings);
    py::list out;
    for (const std::string& x : strings) out.append(py::bytes(x));
    return py::make_tuple((int)st.code(), ints, floats, out);
  });
}
