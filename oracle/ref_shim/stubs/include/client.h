// ORACLE BUILD STUB (test infrastructure).  Stands in for the reference's include/client.h:
// RandomWalk::Call (random_walk.cc:133-141) names an RPC client that a single-process
// store never creates.
#ifndef GLX_REF_STUB_CLIENT_H_
#define GLX_REF_STUB_CLIENT_H_
#include <cstdint>

#include "include/random_walk_request.h"
#include "include/status.h"

namespace graphlearn {

class Client {
public:
  virtual ~Client() {}
  virtual Status RandomWalk(const RandomWalkRequest*, RandomWalkResponse*) { return Status(); }
};

inline Client* NewRpcClient(int32_t) { return nullptr; }

}  // namespace graphlearn
#endif  // GLX_REF_STUB_CLIENT_H_
