// Stub of the protoc-generated header (protoc is absent in this image).
// Only forward declarations are needed: the shim's Tensor is vector-backed.
#ifndef GLX_ORACLE_STUB_TENSOR_PB_H_
#define GLX_ORACLE_STUB_TENSOR_PB_H_
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
namespace graphlearn {
class TensorValue;
class SparseTensorValue;
class OpRequestPb;
class OpResponsePb;
}  // namespace graphlearn
#endif
