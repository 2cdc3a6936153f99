// "ConditionalNegativeSampler" (core/operator/sampler/conditional_negative_sampler.cc:37-161) and its request
// (include/sampling_request.h:152-202, service/request/conditional_sampling_request.cc).
//
// The operator's host part is bookkeeping: which ids are candidates (StorageWrapper::GetIds / GetAllInDegrees /
// GetNodeWeights, core/operator/utils/storage_wrapper.cc:35-66), what their condition attributes are
// (GetNodeAttributesWrapper -> LookupNodes, get_node_attributes_wrapper.cc:45-63) and turning attribute values into
// int64 keys.  Grouping, the alias tables and the sampling itself run on the device (glx_cond_table_create /
// glx_cond_negative_sample).  Like the reference's ConditionTableFactory (condition_table.h:112-183) the table is built
// on the first request for a `type` and kept: the selected columns and proportions of THAT request stay in force for the
// type (ConditionTable::Sample reads its own selected_cols_, condition_table.cc:126-148).
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

#include "glx.h"
#include "graphlearn/config.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"
#include "graphlearn/sampling_request.h"

namespace graphlearn {

namespace {

std::vector<int32_t> Ints(const Tensor::Map& m, const char* k) {
  auto it = m.find(k);
  if (it == m.end()) return {};
  return std::vector<int32_t>(it->second.GetInt32(), it->second.GetInt32() + it->second.Size());
}
std::vector<float> Floats(const Tensor::Map& m, const char* k) {
  auto it = m.find(k);
  if (it == m.end()) return {};
  return std::vector<float>(it->second.GetFloat(), it->second.GetFloat() + it->second.Size());
}
}  // namespace

ConditionalSamplingRequest::ConditionalSamplingRequest() : SamplingRequest() {}

ConditionalSamplingRequest::ConditionalSamplingRequest(const std::string& type, const std::string& strategy,
                                                       int32_t neighbor_count, const std::string& dst_node_type,
                                                       bool batch_share, bool unique)
    : SamplingRequest(type, "ConditionalNegativeSampler", neighbor_count) {
  ADD_TENSOR(params_, kStrategy, kString, 1);
  params_[kStrategy].AddString(strategy);
  ADD_TENSOR(params_, kDstType, kString, 1);
  params_[kDstType].AddString(dst_node_type);
  ADD_TENSOR(params_, kBatchShare, kInt32, 1);
  params_[kBatchShare].AddInt32(batch_share ? 1 : 0);
  ADD_TENSOR(params_, kUnique, kInt32, 1);
  params_[kUnique].AddInt32(unique ? 1 : 0);
  ADD_TENSOR(tensors_, kDstIds, kInt64, 64);
}

// DagNodeRunner-style construction (conditional_sampling_request.cc:99-181): the node `.outNeg(t).sample(n)
// .by(strategy).where(target, condition)` of a query; kStrategy is the strategy WORD here ("random"), the operator
// name is fixed.
void ConditionalSamplingRequest::Init(const Tensor::Map& params) {
  Tensor::Map base;
  for (const char* k : {kEdgeType, kNeighborCount}) base.emplace(k, params.at(k));
  ADD_TENSOR(base, kStrategy, kString, 1);
  base[kStrategy].AddString("ConditionalNegativeSampler");
  SamplingRequest::Init(base);
  ADD_TENSOR(params_, kStrategy, kString, 1);
  params_[kStrategy].AddString(params.at(kStrategy).GetString(0));
  ADD_TENSOR(params_, kDstType, kString, 1);
  params_[kDstType].AddString(params.at(kDstType).GetString(0));
  ADD_TENSOR(params_, kBatchShare, kInt32, 1);
  params_[kBatchShare].AddInt32(params.at(kBatchShare).GetInt32(0));
  ADD_TENSOR(params_, kUnique, kInt32, 1);
  params_[kUnique].AddInt32(params.at(kUnique).GetInt32(0));
  ADD_TENSOR(tensors_, kDstIds, kInt64, 64);
  SetSelectedCols(Ints(params, kIntCols), Floats(params, kIntProps), Ints(params, kFloatCols), Floats(params, kFloatProps),
                  Ints(params, kStrCols), Floats(params, kStrProps));
}

void ConditionalSamplingRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map&) {
  const Tensor& src = tensors.at(kSrcIds);
  const Tensor& dst = tensors.at(kDstIds);
  Set(src.GetInt64(), src.Size());
  tensors_[kDstIds].AddInt64(dst.GetInt64(), dst.GetInt64() + dst.Size());
}

OpRequest* ConditionalSamplingRequest::Clone() const {
  ConditionalSamplingRequest* r =
      new ConditionalSamplingRequest(Type(), Strategy(), NeighborCount(), DstNodeType(), BatchShare(), Unique());
  r->SetSelectedCols(IntCols(), IntProps(), FloatCols(), FloatProps(), StrCols(), StrProps());
  if (HasCallCounter()) r->SetCallCounter(CallCounter());
  return r;
}

void ConditionalSamplingRequest::SetIds(const int64_t* src_ids, const int64_t* dst_ids, int32_t batch_size) {
  Set(src_ids, batch_size);
  tensors_[kDstIds].AddInt64(dst_ids, dst_ids + batch_size);
}

void ConditionalSamplingRequest::SetSelectedCols(const std::vector<int32_t>& int_cols, const std::vector<float>& int_props,
                                                 const std::vector<int32_t>& float_cols, const std::vector<float>& float_props,
                                                 const std::vector<int32_t>& str_cols, const std::vector<float>& str_props) {
  auto put_i = [this](const char* k, const std::vector<int32_t>& v) {
    params_.erase(k);
    ADD_TENSOR(params_, k, kInt32, (int32_t)v.size());
    params_[k].AddInt32(v.data(), v.data() + v.size());
  };
  auto put_f = [this](const char* k, const std::vector<float>& v) {
    params_.erase(k);
    ADD_TENSOR(params_, k, kFloat, (int32_t)v.size());
    params_[k].AddFloat(v.data(), v.data() + v.size());
  };
  put_i(kIntCols, int_cols);
  put_f(kIntProps, int_props);
  put_i(kFloatCols, float_cols);
  put_f(kFloatProps, float_props);
  put_i(kStrCols, str_cols);
  put_f(kStrProps, str_props);
}

const std::string& ConditionalSamplingRequest::Strategy() const { return params_.at(kStrategy).GetString(0); }
const std::string& ConditionalSamplingRequest::DstNodeType() const { return params_.at(kDstType).GetString(0); }
bool ConditionalSamplingRequest::BatchShare() const { return params_.at(kBatchShare).GetInt32(0) != 0; }
bool ConditionalSamplingRequest::Unique() const { return params_.at(kUnique).GetInt32(0) != 0; }
const int64_t* ConditionalSamplingRequest::GetDstIds() const {
  auto it = tensors_.find(kDstIds);
  return it == tensors_.end() ? nullptr : it->second.GetInt64();
}
std::vector<int32_t> ConditionalSamplingRequest::IntCols() const { return Ints(params_, kIntCols); }
std::vector<float> ConditionalSamplingRequest::IntProps() const { return Floats(params_, kIntProps); }
std::vector<int32_t> ConditionalSamplingRequest::FloatCols() const { return Ints(params_, kFloatCols); }
std::vector<float> ConditionalSamplingRequest::FloatProps() const { return Floats(params_, kFloatProps); }
std::vector<int32_t> ConditionalSamplingRequest::StrCols() const { return Ints(params_, kStrCols); }
std::vector<float> ConditionalSamplingRequest::StrProps() const { return Floats(params_, kStrProps); }

REGISTER_REQUEST(ConditionalNegativeSampler, ConditionalSamplingRequest, SamplingResponse)

namespace op {

namespace {
const int64_t kNoKey = INT64_MIN;  // a key no group has

int64_t FloatKey(float v) {
  if (std::isnan(v)) return kNoKey;  // unordered_map<float>: NaN equals nothing (attribute_nodes_map.h:88-100)
  v += 0.0f;                         // -0 == +0
  int32_t bits;
  std::memcpy(&bits, &v, 4);
  return (int64_t)bits;
}

// One per `type`, like ConditionTableFactory::map_ + AliasMethodFactory.
struct CondEntry {
  glx_cond_table* table = nullptr;
  const glx_graph* graph = nullptr;  // neighbour exclusion (edge-type strategies)
  std::string dst_node_type;
  std::vector<int32_t> int_cols, float_cols, str_cols;
  std::vector<float> props;  // int, float, str columns in that order
  std::unordered_map<std::string, int64_t> dict;  // string attribute value -> key
  ~CondEntry() {
    if (table) glx_cond_table_destroy(table);
  }
};
}  // namespace

class ConditionalNegativeSampler : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const ConditionalSamplingRequest* request = static_cast<const ConditionalSamplingRequest*>(req);
    SamplingResponse* response = static_cast<SamplingResponse*>(res);
    const int32_t batch_size = request->BatchSize();
    const int32_t count = request->NeighborCount();
    response->SetShape(batch_size, count);
    response->InitEdgeIds();
    response->InitNeighborIds();
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    if (batch_size > 0 && !request->GetDstIds()) return error::InvalidArgument("the request carries no dst ids");
    CondEntry* e = nullptr;
    Status s = Lookup(request, &e);
    if (!s.ok()) return s;
    response->ResizeNeighborIds();
    // the dst ids' condition attributes -> keys
    Noder* attrs = graph_store_->GetNoder(e->dst_node_type);
    const int32_t ncols = (int32_t)e->props.size();
    std::vector<int64_t> dst_keys((size_t)batch_size * (size_t)(ncols > 0 ? ncols : 1), kNoKey);
    for (int32_t i = 0; i < batch_size; ++i) KeysOf(*e, attrs, request->GetDstIds()[i], dst_keys.data() + (size_t)i * ncols, false);
    const uint64_t cc = request->HasCallCounter() ? (uint64_t)request->CallCounter()
                                                   : call_counter_.fetch_add(1, std::memory_order_relaxed);
    int rc = glx_cond_negative_sample(e->table, e->graph, request->GetSrcIds(), request->GetDstIds(), dst_keys.data(),
                                      e->props.data(), batch_size, count, request->BatchShare() ? 1 : 0,
                                      request->Unique() ? 1 : 0, GLOBAL_FLAG(SamplingRetryTimes),
                                      GLOBAL_FLAG(DefaultNeighborId), (uint64_t)GLOBAL_FLAG(SamplingSeed), cc,
                                      response->GetNeighborIds(), GLX_PTR_HOST, nullptr);
    return error::FromGlx(rc);
  }

private:
  // Attribute values of node `id` in the table's selected columns -> keys[ncols].  Unknown ids carry the default
  // attributes, as LookupNodes answers for them.  `build`: new string values get a dictionary entry (the candidates);
  // otherwise an unseen string matches no group.
  static void KeysOf(CondEntry& e, Noder* attrs, int64_t id, int64_t* keys, bool build) {
    const io::SideInfo* info = attrs->GetSideInfo();
    const int32_t row = attrs->RowOf(id);
    int32_t k = 0;
    for (int32_t c : e.int_cols) {
      keys[k++] = (row >= 0 && c >= 0 && c < info->i_num) ? attrs->GetIntAttrs(row)[c] : GLOBAL_FLAG(DefaultIntAttribute);
    }
    for (int32_t c : e.float_cols) {
      keys[k++] = FloatKey((row >= 0 && c >= 0 && c < info->f_num) ? attrs->GetFloatAttrs(row)[c]
                                                                    : GLOBAL_FLAG(DefaultFloatAttribute));
    }
    for (int32_t c : e.str_cols) {
      const std::string& v = (row >= 0 && c >= 0 && c < info->s_num) ? attrs->GetStringAttrs(row)[c]
                                                                     : GLOBAL_FLAG(DefaultStringAttribute);
      auto it = e.dict.find(v);
      if (it == e.dict.end()) {
        if (!build) {
          keys[k++] = kNoKey;
          continue;
        }
        it = e.dict.emplace(v, (int64_t)e.dict.size()).first;
      }
      keys[k++] = it->second;
    }
  }

  Status Lookup(const ConditionalSamplingRequest* request, CondEntry** out) {
    const std::string& type = request->Type();
    std::lock_guard<std::mutex> g(mtx_);
    // The tables belong to ONE store: an entry keeps a pointer to the store's device graph.  The operator is a
    // process-wide singleton (op_factory.cc:26-66) and outlives stores; bound to a new store it starts over (round 6: a
    // second Graph with the same edge type name in one process found the first one's entry -- a dangling graph
    // pointer, seen as a sporadic "graph and condition table live on different devices").
    if (store_uid_ != graph_store_->Uid()) {
      tables_.clear();
      store_uid_ = graph_store_->Uid();
    }
    auto it = tables_.find(type);
    if (it != tables_.end()) {
      *out = it->second.get();
      return Status::OK();
    }
    std::unique_ptr<CondEntry> e(new CondEntry);
    e->dst_node_type = request->DstNodeType();
    e->int_cols = request->IntCols();
    e->float_cols = request->FloatCols();
    e->str_cols = request->StrCols();
    const std::vector<float> ip = request->IntProps(), fp = request->FloatProps(), sp = request->StrProps();
    if (ip.size() != e->int_cols.size() || fp.size() != e->float_cols.size() || sp.size() != e->str_cols.size()) {
      return error::InvalidArgument("selected columns and their proportions differ in length");
    }
    e->props.insert(e->props.end(), ip.begin(), ip.end());
    e->props.insert(e->props.end(), fp.begin(), fp.end());
    e->props.insert(e->props.end(), sp.begin(), sp.end());
    // candidates and weights (StorageWrapper, conditional_negative_sampler.cc:60-93)
    const std::string& strategy = request->Strategy();
    std::vector<int64_t> ids;
    std::vector<float> weights;
    int device = GLOBAL_FLAG(DeviceId);
    if (strategy == "node_weight") {
      Noder* noder = graph_store_->GetNoder(type);
      ids = noder->Ids();
      weights.resize(ids.size());
      for (size_t i = 0; i < ids.size(); ++i) weights[i] = noder->GetWeight(ids[i]);
      if (noder->Device()) glx_features_info(noder->Device(), nullptr, nullptr, nullptr, &device);
    } else {
      Graph* graph = graph_store_->GetGraph(type);
      const glx_negative* cand = nullptr;
      Status s = graph->Negative(false, false, &cand);  // the distinct destination ids in first-appearance order
      if (!s.ok()) return s;
      int64_t n = 0;
      int rc = glx_negative_info(cand, &n, nullptr);
      if (rc != GLX_OK) return error::FromGlx(rc);
      ids.resize((size_t)n);
      if (n > 0) {
        rc = glx_negative_export(cand, ids.data(), nullptr, nullptr, nullptr);
        if (rc != GLX_OK) return error::FromGlx(rc);
      }
      e->graph = graph->Device();
      glx_graph_info(e->graph, nullptr, nullptr, nullptr, nullptr, &device);
      if (strategy == "in_degree") {
        s = graph->EnsureInDegree();
        if (!s.ok()) return s;
        std::vector<int64_t> deg(ids.size());
        if (n > 0) {
          rc = glx_graph_in_degrees(e->graph, ids.data(), n, deg.data(), GLX_PTR_HOST, nullptr);
          if (rc != GLX_OK) return error::FromGlx(rc);
        }
        weights.resize(ids.size());
        for (size_t i = 0; i < ids.size(); ++i) weights[i] = (float)deg[i];
      }
    }
    Noder* attrs = graph_store_->GetNoder(e->dst_node_type);
    const int32_t ncols = (int32_t)e->props.size();
    const size_t U = ids.size();
    std::vector<int64_t> cand_keys((size_t)(ncols > 0 ? ncols : 1) * (U > 0 ? U : 1));
    std::vector<int64_t> row((size_t)(ncols > 0 ? ncols : 1));
    int64_t nan_key = INT64_MIN + 1;  // NaN-valued candidates: one unreachable singleton group each
    for (size_t u = 0; u < U; ++u) {
      KeysOf(*e, attrs, ids[u], row.data(), true);
      for (int32_t c = 0; c < ncols; ++c) cand_keys[(size_t)c * U + u] = row[(size_t)c] == kNoKey ? nan_key++ : row[(size_t)c];
    }
    int rc = glx_cond_table_create(device, (int64_t)U, ids.data(), weights.empty() ? nullptr : weights.data(), ncols,
                                   cand_keys.data(), GLX_PTR_HOST, nullptr, &e->table);
    if (rc != GLX_OK) return error::FromGlx(rc);
    *out = e.get();
    tables_[type] = std::move(e);
    return Status::OK();
  }

  std::mutex mtx_;
  uint64_t store_uid_ = 0;  // GraphStore::Uid() of the store tables_ was built over (uids start at 1)
  std::map<std::string, std::unique_ptr<CondEntry>> tables_;
  std::atomic<uint64_t> call_counter_{0};
};

REGISTER_OPERATOR("ConditionalNegativeSampler", ConditionalNegativeSampler)

}  // namespace op
}  // namespace graphlearn
