// Operator plugin API.  Mirrors graphlearn/src/core/operator/operator.h:27-50,
// op_registry.h:28-68 and op_factory.h:27-47: operators are looked up by name,
// created once per name, bound to the GraphStore with Set(), and invoked through
// the virtual Process(const OpRequest*, OpResponse*).
#ifndef GLX_HOST_OPERATOR_H_
#define GLX_HOST_OPERATOR_H_
#include <mutex>
#include <string>
#include <unordered_map>

#include "graphlearn/op_request.h"
#include "graphlearn/status.h"

namespace graphlearn {
class GraphStore;

namespace op {

class Operator {
public:
  Operator() : graph_store_(nullptr) {}
  virtual ~Operator() = default;
  void Set(GraphStore* graph_store) { graph_store_ = graph_store; }
  virtual Status Process(const OpRequest* req, OpResponse* res) = 0;

protected:
  GraphStore* graph_store_;
};

class OpRegistry {
public:
  typedef Operator* (*OpCreator)();
  static OpRegistry* GetInstance();
  void Register(const std::string& name, OpCreator creator);
  OpCreator* Lookup(const std::string& name);

private:
  OpRegistry() = default;
  std::unordered_map<std::string, OpCreator> map_;
};

class OpFactory {
public:
  static OpFactory* GetInstance();
  void Set(GraphStore* graph_store);
  // Returns the (single) instance for `name`, or nullptr for an unknown name --
  // callers turn that into InvalidArgument (service/executor.cc:37-40).
  Operator* Create(const std::string& name);

private:
  OpFactory();
  ~OpFactory();
  std::mutex mtx_;
  GraphStore* graph_store_;
  std::unordered_map<std::string, Operator*> map_;
};

}  // namespace op
}  // namespace graphlearn

#define REGISTER_OPERATOR(OpName, OpClass)                                            \
  inline ::graphlearn::op::Operator* New##OpClass##Operator() { return new OpClass(); } \
  class Register##OpClass {                                                           \
  public:                                                                             \
    Register##OpClass() {                                                             \
      ::graphlearn::op::OpRegistry::GetInstance()->Register(OpName, New##OpClass##Operator); \
    }                                                                                 \
  };                                                                                  \
  static Register##OpClass register_##OpClass;

#endif  // GLX_HOST_OPERATOR_H_
