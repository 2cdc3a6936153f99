// "SubGraphSampler" (core/operator/subgraph/subgraph_sampler.{h,cc}, subgraph_utils.cc) and its request /
// response (include/subgraph_request.h, service/request/subgraph_request.cc).
//   Process (subgraph_sampler.h:36-78): nodes = the seeds; per hop h with num_nbrs[h] > 0 one FullSampler request
//     (limit num_nbrs[h]) on the previous hop's neighbours; all neighbours go into one sorted set; the node list is
//     the seeds followed by that set (a seed that is also a neighbour appears twice, as in the reference).
//   InduceSubGraph (subgraph_sampler.cc:34-95): FullSampler with limit DefaultFullNbrNum on the node list, then the
//     N x N membership pass -- here one device call on FullSampler's response rows (glx_subgraph_induce).
//   need_dist (:71-93): BFS distances to node 0 ("src") without node 1 ("dst") and vice versa on the induced COO;
//     a graph of a few hundred nodes that is already in host memory for the response: walked here.
// The reference feeds hop h + 1 from the (already destroyed) response of hop h; this operator keeps the response.
#include <algorithm>
#include <climits>
#include <queue>
#include <set>

#include "glx.h"
#include "graphlearn/config.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/op_runner.h"
#include "graphlearn/operator.h"
#include "graphlearn/sampling_request.h"
#include "graphlearn/subgraph_request.h"

namespace graphlearn {

namespace {
const int32_t kReservedSize = 64;
}  // namespace

SubGraphRequest::SubGraphRequest() : OpRequest() { DisableShard(); }

SubGraphRequest::SubGraphRequest(const std::string& nbr_type, const std::vector<int32_t>& num_nbrs, bool need_dist)
    : OpRequest() {
  DisableShard();
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString("SubGraphSampler");
  ADD_TENSOR(params_, kNbrType, kString, 1);
  params_[kNbrType].AddString(nbr_type);
  ADD_TENSOR(params_, kNeighborCount, kInt32, (int32_t)num_nbrs.size());
  params_[kNeighborCount].AddInt32(num_nbrs.data(), num_nbrs.data() + num_nbrs.size());
  ADD_TENSOR(params_, kNeedDist, kInt32, 1);
  params_[kNeedDist].AddInt32(need_dist ? 1 : 0);
  ADD_TENSOR(tensors_, kSrcIds, kInt64, kReservedSize);
}

void SubGraphRequest::Init(const Tensor::Map& params) {
  ADD_TENSOR(params_, kOpName, kString, 1);
  params_[kOpName].AddString(params.at(kOpName).GetString(0));
  ADD_TENSOR(params_, kNbrType, kString, 1);
  params_[kNbrType].AddString(params.at(kNbrType).GetString(0));
  const Tensor& nbc = params.at(kNeighborCount);
  ADD_TENSOR(params_, kNeighborCount, kInt32, nbc.Size());
  params_[kNeighborCount].AddInt32(nbc.GetInt32(), nbc.GetInt32() + nbc.Size());
  ADD_TENSOR(params_, kNeedDist, kInt32, 1);
  params_[kNeedDist].AddInt32(params.at(kNeedDist).GetInt32(0));
  ADD_TENSOR(tensors_, kSrcIds, kInt64, kReservedSize);
}

OpRequest* SubGraphRequest::Clone() const { return new SubGraphRequest(NbrType(), GetNumNbrs(), NeedDist()); }

void SubGraphRequest::Set(const int64_t* src_id, int32_t batch_size) {
  tensors_[kSrcIds].AddInt64(src_id, src_id + batch_size);
}

void SubGraphRequest::Set(const int64_t* src_id, const int64_t* dst_id, int32_t batch_size) {
  tensors_[kSrcIds].AddInt64(src_id, src_id + batch_size);
  tensors_[kSrcIds].AddInt64(dst_id, dst_id + batch_size);
}

void SubGraphRequest::Set(const Tensor::Map& tensors, const SparseTensor::Map&) {
  const Tensor& src = tensors.at(kSrcIds);
  tensors_[kSrcIds].AddInt64(src.GetInt64(), src.GetInt64() + src.Size());
  auto it = tensors.find(kDstIds);  // subgraph_request.cc:92-96
  if (it != tensors.end()) tensors_[kSrcIds].AddInt64(it->second.GetInt64(), it->second.GetInt64() + it->second.Size());
}

const std::string& SubGraphRequest::NbrType() const { return params_.at(kNbrType).GetString(0); }
std::vector<int32_t> SubGraphRequest::GetNumNbrs() const {
  const Tensor& t = params_.at(kNeighborCount);
  return std::vector<int32_t>(t.GetInt32(), t.GetInt32() + t.Size());
}
bool SubGraphRequest::NeedDist() const { return params_.at(kNeedDist).GetInt32(0) == 1; }
const int64_t* SubGraphRequest::GetSrcIds() const {
  auto it = tensors_.find(kSrcIds);
  return it == tensors_.end() ? nullptr : it->second.GetInt64();
}
int32_t SubGraphRequest::BatchSize() const {
  auto it = tensors_.find(kSrcIds);
  return it == tensors_.end() ? 0 : it->second.Size();
}

SubGraphResponse::SubGraphResponse() : OpResponse() {}

void SubGraphResponse::Init(int32_t batch_size) {
  for (const char* k : {kNodeIds, kRowIndices, kColIndices, kEdgeIds, kDistToSrc, kDistToDst}) tensors_.erase(k);
  ADD_TENSOR(tensors_, kNodeIds, kInt64, batch_size);
  ADD_TENSOR(tensors_, kRowIndices, kInt32, batch_size);
  ADD_TENSOR(tensors_, kColIndices, kInt32, batch_size);
  ADD_TENSOR(tensors_, kEdgeIds, kInt64, batch_size);
  ADD_TENSOR(tensors_, kDistToSrc, kInt32, batch_size);
  ADD_TENSOR(tensors_, kDistToDst, kInt32, batch_size);
}
void SubGraphResponse::SetNodeIds(const int64_t* begin, int32_t size) {
  tensors_[kNodeIds].AddInt64(begin, begin + size);
  batch_size_ = size;
}
void SubGraphResponse::AppendEdge(int32_t row_idx, int32_t col_idx, int64_t e_id) {
  tensors_[kRowIndices].AddInt32(row_idx);
  tensors_[kColIndices].AddInt32(col_idx);
  tensors_[kEdgeIds].AddInt64(e_id);
}
void SubGraphResponse::SetDistToSrc(const int32_t* begin, int32_t size) { tensors_[kDistToSrc].AddInt32(begin, begin + size); }
void SubGraphResponse::SetDistToDst(const int32_t* begin, int32_t size) { tensors_[kDistToDst].AddInt32(begin, begin + size); }
void SubGraphResponse::ResizeEdges(int32_t count) {
  tensors_[kRowIndices].Resize(count);
  tensors_[kColIndices].Resize(count);
  tensors_[kEdgeIds].Resize(count);
}
int32_t* SubGraphResponse::MutableRowIndices() { return tensors_[kRowIndices].MutableInt32(); }
int32_t* SubGraphResponse::MutableColIndices() { return tensors_[kColIndices].MutableInt32(); }
int64_t* SubGraphResponse::MutableEdgeIds() { return tensors_[kEdgeIds].MutableInt64(); }
int32_t SubGraphResponse::EdgeCount() const { return tensors_.at(kRowIndices).Size(); }
const int64_t* SubGraphResponse::NodeIds() const { return tensors_.at(kNodeIds).GetInt64(); }
const int32_t* SubGraphResponse::RowIndices() const { return tensors_.at(kRowIndices).GetInt32(); }
const int32_t* SubGraphResponse::ColIndices() const { return tensors_.at(kColIndices).GetInt32(); }
const int64_t* SubGraphResponse::EdgeIds() const { return tensors_.at(kEdgeIds).GetInt64(); }
const int32_t* SubGraphResponse::DistToSrc() const { return tensors_.at(kDistToSrc).GetInt32(); }
const int32_t* SubGraphResponse::DistToDst() const { return tensors_.at(kDistToDst).GetInt32(); }

REGISTER_REQUEST(SubGraphSampler, SubGraphRequest, SubGraphResponse)

namespace {
// BFSShortestPath (subgraph_utils.cc:36-57) on the induced COO without node `skip`; unreachable = INT32_MAX.
std::vector<int32_t> BfsWithout(int32_t n, const std::vector<std::vector<int32_t>>& adj, int32_t skip, int32_t start) {
  std::vector<int32_t> dist((size_t)n, INT32_MAX);
  std::queue<int32_t> q;
  dist[(size_t)start] = 0;
  q.push(start);
  while (!q.empty()) {
    const int32_t s = q.front();
    q.pop();
    if (s == skip) continue;
    for (int32_t nb : adj[(size_t)s]) {
      if (nb == skip || dist[(size_t)nb] != INT32_MAX) continue;
      dist[(size_t)nb] = dist[(size_t)s] + 1;
      q.push(nb);
    }
  }
  return dist;
}
}  // namespace

// The operator's body with the way to run a FullSampler request left open: the local operator, or the
// partitioned FullSampler of the distributed runner (the reference's GetOpRunner(Env::Default(), op), :29-31).
Status RunSubGraph(const SubGraphRequest* request, SubGraphResponse* response, int device,
                   const std::function<Status(const SamplingRequest*, SamplingResponse*)>& full_sampler) {
  const int64_t* seeds = request->GetSrcIds();
  const int32_t n_seeds = request->BatchSize();
  std::vector<int64_t> nodes_vec(seeds, seeds + n_seeds);
  std::vector<int64_t> frontier(nodes_vec);
  std::set<int64_t> nbrs_set;
  for (int32_t num_nbr : request->GetNumNbrs()) {
    if (num_nbr <= 0) continue;
    SamplingRequest req(request->NbrType(), "FullSampler", num_nbr);
    req.Set(frontier.data(), (int32_t)frontier.size());
    SamplingResponse res;
    Status s = full_sampler(&req, &res);
    if (!s.ok()) return s;
    const int32_t total = res.GetShape().size;
    frontier.assign(res.GetNeighborIds(), res.GetNeighborIds() + total);
    nbrs_set.insert(frontier.begin(), frontier.end());
  }
  nodes_vec.insert(nodes_vec.end(), nbrs_set.begin(), nbrs_set.end());
  const int32_t n = (int32_t)nodes_vec.size();

  SamplingRequest req(request->NbrType(), "FullSampler", GLOBAL_FLAG(DefaultFullNbrNum));
  req.Set(nodes_vec.data(), n);
  SamplingResponse res;
  Status s = full_sampler(&req, &res);
  if (!s.ok()) return s;
  const Shape shape = res.GetShape();
  std::vector<int64_t> offsets((size_t)n + 1, 0);
  for (int32_t i = 0; i < n; ++i) offsets[(size_t)i + 1] = offsets[(size_t)i] + shape.segments[(size_t)i];

  response->Init(n);
  response->SetNodeIds(nodes_vec.data(), n);
  int64_t count = 0;
  int rc = glx_subgraph_induce(device, nodes_vec.data(), n, offsets.data(), res.GetNeighborIds(), res.GetEdgeIds(), nullptr,
                               nullptr, nullptr, 0, &count, GLX_PTR_HOST, nullptr);
  if (rc != GLX_OK) return error::FromGlx(rc);
  if (count > INT32_MAX) return error::InvalidArgument("the induced sub-graph exceeds int32 entries (tensor.h:47)");
  response->ResizeEdges((int32_t)count);
  if (count > 0) {
    rc = glx_subgraph_induce(device, nodes_vec.data(), n, offsets.data(), res.GetNeighborIds(), res.GetEdgeIds(),
                             response->MutableRowIndices(), response->MutableColIndices(), response->MutableEdgeIds(), count,
                             &count, GLX_PTR_HOST, nullptr);
    if (rc != GLX_OK) return error::FromGlx(rc);
  }
  if (request->NeedDist()) {
    if (n < 2) return error::InvalidArgument("need_dist needs at least two nodes (src, dst)");
    std::vector<std::vector<int32_t>> adj((size_t)n);
    const int32_t* row = response->RowIndices();
    const int32_t* col = response->ColIndices();
    for (int64_t e = 0; e < count; ++e) adj[(size_t)row[e]].push_back(col[e]);
    std::vector<int32_t> to_dst = BfsWithout(n, adj, /*skip=*/0, /*start=*/1);
    std::vector<int32_t> to_src = BfsWithout(n, adj, /*skip=*/1, /*start=*/0);
    to_dst[0] = 0;
    to_src[1] = 0;
    response->SetDistToSrc(to_src.data(), n);
    response->SetDistToDst(to_dst.data(), n);
  }
  return Status::OK();
}

namespace op {

class SubGraphSampler : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    const SubGraphRequest* request = static_cast<const SubGraphRequest*>(req);
    SubGraphResponse* response = static_cast<SubGraphResponse*>(res);
    if (!graph_store_) return error::InvalidArgument("operator is not bound to a GraphStore");
    const glx_graph* g = graph_store_->GetGraph(request->NbrType())->Device();
    int device = GLOBAL_FLAG(DeviceId);
    if (g) glx_graph_info(g, nullptr, nullptr, nullptr, nullptr, &device);
    Operator* full = OpFactory::GetInstance()->Create("FullSampler");
    if (!full) return error::Internal("FullSampler is not registered");
    return RunSubGraph(request, response, device,
                       [full](const SamplingRequest* q, SamplingResponse* r) { return full->Process(q, r); });
  }
};

REGISTER_OPERATOR("SubGraphSampler", SubGraphSampler)

}  // namespace op
}  // namespace graphlearn
