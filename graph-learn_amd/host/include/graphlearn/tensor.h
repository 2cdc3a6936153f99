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

class Tensor {
public:
  Tensor();
  explicit Tensor(DataType dtype);
  Tensor(DataType dtype, int32_t capacity);

  DataType DType() const;
  int32_t Size() const;
  void Resize(int32_t size);

  void AddInt32(int32_t v);
  void AddInt64(int64_t v);
  void AddFloat(float v);
  void AddDouble(double v);
  void AddString(const std::string& v);
  void AddInt32(const int32_t* begin, const int32_t* end);
  void AddInt64(const int64_t* begin, const int64_t* end);
  void AddFloat(const float* begin, const float* end);
  void AddDouble(const double* begin, const double* end);

  void SetInt32(int32_t index, int32_t v);
  void SetInt64(int32_t index, int64_t v);
  void SetFloat(int32_t index, float v);

  int32_t GetInt32(int32_t index) const;
  int64_t GetInt64(int32_t index) const;
  float GetFloat(int32_t index) const;
  double GetDouble(int32_t index) const;
  const std::string& GetString(int32_t index) const;

  const int32_t* GetInt32() const;
  const int64_t* GetInt64() const;
  const float* GetFloat() const;
  const double* GetDouble() const;

  // Device-path additions: bulk-writable views (valid after Resize()).
  int32_t* MutableInt32();
  int64_t* MutableInt64();
  float* MutableFloat();

  void Swap(Tensor& right);

  typedef std::unordered_map<std::string, Tensor> Map;

private:
  struct Impl;
  std::shared_ptr<Impl> impl_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_TENSOR_H_
