// OpRequest / OpResponse / RequestFactory with the reference's surface
// (graphlearn/src/include/op_request.h:33-152) minus protobuf (de)serialisation:
// in this design shards exchange raw device buffers over RCCL, not messages.
#ifndef GLX_HOST_OP_REQUEST_H_
#define GLX_HOST_OP_REQUEST_H_
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

#include "graphlearn/constants.h"
#include "graphlearn/status.h"
#include "graphlearn/tensor.h"

namespace graphlearn {

class OpRequest {
public:
  explicit OpRequest(const std::string& shard_key = kUnspecified);
  virtual ~OpRequest() = default;

  // The two steps a DAG node's request is made in (core/runner/dag_node_runner.cc:100-109): Init from the node's
  // parameters, Set from the tensors its in-edges deliver (dense ones by input name; ragged ones -- the output of
  // a FullSampler upstream -- in sparse_tensors).
  virtual void Init(const Tensor::Map& params) { (void)params; }
  virtual void Set(const Tensor::Map& tensors, const SparseTensor::Map& sparse_tensors) {
    (void)tensors;
    (void)sparse_tensors;
  }
  void Set(const Tensor::Map& tensors) { Set(tensors, SparseTensor::Map()); }
  virtual std::string Name() const;
  virtual OpRequest* Clone() const;
  const std::string& ShardKey() const { return shard_key_; }
  bool IsShardable() const { return shardable_; }
  void DisableShard() { shardable_ = false; }

public:
  std::unordered_map<std::string, Tensor> params_;
  std::unordered_map<std::string, Tensor> tensors_;

protected:
  std::string shard_key_;
  bool shardable_;
};

class OpResponse {
public:
  OpResponse();
  virtual ~OpResponse() = default;
  virtual OpResponse* New() const { return new OpResponse; }
  virtual void Swap(OpResponse& right);

public:
  int32_t batch_size_;
  std::unordered_map<std::string, Tensor> params_;
  std::unordered_map<std::string, Tensor> tensors_;
};

typedef std::unique_ptr<OpRequest> OpRequestPtr;
typedef std::unique_ptr<OpResponse> OpResponsePtr;

class RequestFactory {
public:
  typedef OpRequest* (*RequestCreator)();
  typedef OpResponse* (*ResponseCreator)();
  static RequestFactory* GetInstance();
  void Register(const std::string& name, RequestCreator req, ResponseCreator res);
  OpRequest* NewRequest(const std::string& name);
  OpResponse* NewResponse(const std::string& name);

private:
  RequestFactory() = default;
  std::mutex mtx_;
  std::unordered_map<std::string, RequestCreator> req_;
  std::unordered_map<std::string, ResponseCreator> res_;
};

}  // namespace graphlearn

#define REGISTER_REQUEST(Name, ReqClass, ResClass)                                          \
  inline ::graphlearn::OpRequest* New##Name##ReqClass() { return new ReqClass(); }          \
  inline ::graphlearn::OpResponse* New##Name##ResClass() { return new ResClass(); }         \
  class Register##Name##ReqClass {                                                          \
  public:                                                                                   \
    Register##Name##ReqClass() {                                                            \
      ::graphlearn::RequestFactory::GetInstance()->Register(#Name, New##Name##ReqClass,     \
                                                            New##Name##ResClass);           \
    }                                                                                       \
  };                                                                                        \
  static Register##Name##ReqClass register_##Name##ReqClass;

#endif  // GLX_HOST_OP_REQUEST_H_
