// graphlearn::Status / error::Code for the glx host layer.
// Mirrors graphlearn/src/include/status.h:25-80 and common/base/errors.h (same
// names, same numeric codes -- they are also the return codes of the C-ABI).
#ifndef GLX_HOST_STATUS_H_
#define GLX_HOST_STATUS_H_
#include <string>

namespace graphlearn {
namespace error {
enum Code {
  OK = 0,
  CANCELLED = 1,
  UNKNOWN = 2,
  INVALID_ARGUMENT = 3,
  DEADLINE_EXCEEDED = 4,
  NOT_FOUND = 5,
  ALREADY_EXISTS = 6,
  PERMISSION_DENIED = 7,
  RESOURCE_EXHAUSTED = 8,
  FAILED_PRECONDITION = 9,
  ABORTED = 10,
  OUT_OF_RANGE = 11,
  UNIMPLEMENTED = 12,
  INTERNAL = 13,
  UNAVAILABLE = 14,
  DATA_LOSS = 15,
  UNAUTHENTICATED = 16,
  REQUEST_STOP = 17,
};
}  // namespace error

class Status {
public:
  explicit Status(error::Code code = error::OK, const std::string& msg = std::string())
      : code_(code), msg_(msg) {}
  static Status OK() { return Status(); }
  bool ok() const { return code_ == error::OK; }
  error::Code code() const { return code_; }
  const std::string& msg() const { return msg_; }
  std::string ToString() const;

private:
  error::Code code_;
  std::string msg_;
};

namespace error {
inline Status InvalidArgument(const std::string& m) { return Status(INVALID_ARGUMENT, m); }
inline Status NotFound(const std::string& m) { return Status(NOT_FOUND, m); }
inline Status Unimplemented(const std::string& m) { return Status(UNIMPLEMENTED, m); }
inline Status Internal(const std::string& m) { return Status(INTERNAL, m); }
inline Status Unavailable(const std::string& m) { return Status(UNAVAILABLE, m); }
inline Status OutOfRange(const std::string& m) { return Status(OUT_OF_RANGE, m); }
inline bool IsInvalidArgument(const Status& s) { return s.code() == INVALID_ARGUMENT; }
inline bool IsUnimplemented(const Status& s) { return s.code() == UNIMPLEMENTED; }
inline bool IsUnavailable(const Status& s) { return s.code() == UNAVAILABLE; }
// Wraps a C-ABI return code + glx_last_error() into a Status.
Status FromGlx(int code);
}  // namespace error
}  // namespace graphlearn
#endif  // GLX_HOST_STATUS_H_
