// "UpdateEdges" / "UpdateNodes": the registry names the reference's loader delivers records by
// (core/operator/graph/edge_updater.cc:25-55, node_updater.cc; Graph::UpdateEdges -> storage->Add,
// core/graph/local_graph.cc:50-64).  Records are staged on the host; GraphStore::Build() then builds the device
// storages from everything staged, so an update after Build() is refused (graph_store.cc).
#include "graphlearn/graph_store.h"
#include "graphlearn/operator.h"

namespace graphlearn {

REGISTER_REQUEST(UpdateEdges, UpdateEdgesRequest, UpdateEdgesResponse)
REGISTER_REQUEST(UpdateNodes, UpdateNodesRequest, UpdateNodesResponse)

namespace op {

class EdgeUpdater : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    auto* request = dynamic_cast<const UpdateEdgesRequest*>(req);
    auto* response = dynamic_cast<UpdateEdgesResponse*>(res);
    if (!request || !response) return error::InvalidArgument("UpdateEdges needs an UpdateEdgesRequest / Response");
    if (!graph_store_) return error::InvalidArgument("UpdateEdges: no graph store is bound");
    const io::SideInfo& info = request->GetSideInfo();
    if (info.type.empty()) return error::InvalidArgument("UpdateEdges: the side info names no edge type");
    return graph_store_->GetGraph(info.type)->UpdateEdges(request, response);
  }
};

class NodeUpdater : public Operator {
public:
  Status Process(const OpRequest* req, OpResponse* res) override {
    auto* request = dynamic_cast<const UpdateNodesRequest*>(req);
    auto* response = dynamic_cast<UpdateNodesResponse*>(res);
    if (!request || !response) return error::InvalidArgument("UpdateNodes needs an UpdateNodesRequest / Response");
    if (!graph_store_) return error::InvalidArgument("UpdateNodes: no graph store is bound");
    const io::SideInfo& info = request->GetSideInfo();
    if (info.type.empty()) return error::InvalidArgument("UpdateNodes: the side info names no node type");
    return graph_store_->GetNoder(info.type)->UpdateNodes(request, response);
  }
};

REGISTER_OPERATOR("UpdateEdges", EdgeUpdater);
REGISTER_OPERATOR("UpdateNodes", NodeUpdater);

}  // namespace op
}  // namespace graphlearn
