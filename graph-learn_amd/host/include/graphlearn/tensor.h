// Vector-backed Tensor with the reference's interface (graphlearn/src/include/
// tensor.h:35-86).  The reference backs it by protobuf RepeatedField
// (service/tensor_impl.h:196-200); on this path protobuf is only a container, so
// a std::vector is the whole story -- and it lets the device path write a whole
// response with one copy (MutableInt64/MutableFloat after Resize) instead of
// B*k AddInt64 calls.
#ifndef GLX_HOST_TENSOR_H_
#define GLX_HOST_TENSOR_H_
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace graphlearn {

enum DataType { kInt32, kInt64, kFloat, kDouble, kString, kUnknown };

#define ADD_TENSOR(m, name, type, size) \
  m.emplace(std::piecewise_construct, std::forward_as_tuple(name), std::forward_as_tuple(type, size))

// One block of typed accessors per element type; the method names are the
// reference's (AddInt32 / SetInt32 / GetInt32(i) / GetInt32() ...).
#define GLX_TENSOR_TYPED_API(Name, T)                     \
  void Add##Name(T v);                                    \
  void Add##Name(const T* begin, const T* end);           \
  T Get##Name(int32_t index) const;                       \
  const T* Get##Name() const;

class Tensor {
public:
  Tensor();
  explicit Tensor(DataType dtype);
  Tensor(DataType dtype, int32_t capacity);

  DataType DType() const;
  int32_t Size() const;
  void Resize(int32_t size);
  void Swap(Tensor& right);

  GLX_TENSOR_TYPED_API(Int32, int32_t)
  GLX_TENSOR_TYPED_API(Int64, int64_t)
  GLX_TENSOR_TYPED_API(Float, float)
  GLX_TENSOR_TYPED_API(Double, double)

  void AddString(const std::string& v);
  const std::string& GetString(int32_t index) const;

  void SetInt32(int32_t index, int32_t v);
  void SetInt64(int32_t index, int64_t v);
  void SetFloat(int32_t index, float v);

  // Device-path additions: bulk-writable views (valid after Resize()), so a whole
  // response is one copy out of HBM instead of batch*k Add calls.
  int32_t* MutableInt32();
  int64_t* MutableInt64();
  float* MutableFloat();
  // Keeps the storage alive independently of this Tensor (zero-copy hand-off of a response
  // block to a caller such as numpy).
  std::shared_ptr<const void> Owner() const;

  typedef std::unordered_map<std::string, Tensor> Map;

private:
  struct Impl;
  std::shared_ptr<Impl> impl_;
};

#undef GLX_TENSOR_TYPED_API

// Ragged values: `segments` (int32, one count per row) + `values` (the rows back to back) -- what a
// FullSampler answers with and what a DAG edge hands to the next node (include/sparse_tensor.h:26-60).
// Copies share storage, like Tensor.
class SparseTensor {
public:
  SparseTensor() : segments_(kInt32), values_(kInt64) {}
  SparseTensor(const Tensor& segments, const Tensor& values) : segments_(segments), values_(values) {}
  const Tensor& Segments() const { return segments_; }
  const Tensor& Values() const { return values_; }
  Tensor* MutableSegments() { return &segments_; }
  Tensor* MutableValues() { return &values_; }

  typedef std::unordered_map<std::string, SparseTensor> Map;

private:
  Tensor segments_;
  Tensor values_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_TENSOR_H_
