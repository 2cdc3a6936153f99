// ORACLE BUILD STUB (test infrastructure).  Stands in for the reference's
// core/runner/op_runner.h, whose real version drags in the RPC/closure/protobuf stack.
// The reference's RandomWalk operator only needs "run this operator on a request"
// (random_walk.cc:52,75,123); in a single-process store that is Operator::Process.
#ifndef GLX_REF_STUB_OP_RUNNER_H_
#define GLX_REF_STUB_OP_RUNNER_H_
#include <memory>

#include "core/operator/operator.h"
#include "include/op_request.h"
#include "include/status.h"

namespace graphlearn {

class Env {
public:
  static Env* Default() { return nullptr; }
};

namespace op {

class OpRunner {
public:
  explicit OpRunner(Operator* op) : op_(op) {}
  Status Run(const OpRequest* req, OpResponse* res) { return op_->Process(req, res); }

private:
  Operator* op_;
};

inline std::unique_ptr<OpRunner> GetOpRunner(Env*, Operator* op) {
  return std::unique_ptr<OpRunner>(new OpRunner(op));
}

}  // namespace op
}  // namespace graphlearn
#endif  // GLX_REF_STUB_OP_RUNNER_H_
