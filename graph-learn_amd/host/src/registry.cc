// OpRegistry / OpFactory (op_registry.cc:24-40, op_factory.cc:26-66): name ->
// creator map filled by REGISTER_OPERATOR at static-init time; one operator
// instance per name, created lazily under a mutex, bound to the GraphStore.
#include "graphlearn/operator.h"

namespace graphlearn {
namespace op {

OpRegistry* OpRegistry::GetInstance() {
  static OpRegistry registry;
  return &registry;
}

void OpRegistry::Register(const std::string& name, OpCreator creator) { map_[name] = creator; }

OpRegistry::OpCreator* OpRegistry::Lookup(const std::string& name) {
  // read-only after static initialisation, hence lock free (op_registry.cc:33-35)
  auto it = map_.find(name);
  return it == map_.end() ? nullptr : &it->second;
}

OpFactory::OpFactory() : graph_store_(nullptr) {}

OpFactory::~OpFactory() {
  for (auto& it : map_) delete it.second;
}

OpFactory* OpFactory::GetInstance() {
  static OpFactory factory;
  return &factory;
}

void OpFactory::Set(GraphStore* graph_store) {
  std::lock_guard<std::mutex> g(mtx_);
  graph_store_ = graph_store;
  for (auto& it : map_) it.second->Set(graph_store_);
}

Operator* OpFactory::Create(const std::string& name) {
  std::lock_guard<std::mutex> g(mtx_);
  auto it = map_.find(name);
  if (it != map_.end()) return it->second;
  auto creator = OpRegistry::GetInstance()->Lookup(name);
  if (!creator) return nullptr;
  Operator* op = (*creator)();
  if (graph_store_) op->Set(graph_store_);
  map_[name] = op;
  return op;
}

}  // namespace op
}  // namespace graphlearn
