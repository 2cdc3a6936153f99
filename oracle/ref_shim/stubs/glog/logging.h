// Stub of glog for the oracle shim-compile (test infrastructure only).
// The reference's hot path only uses LOG(x) << ... as a sink.
#ifndef GLX_ORACLE_STUB_GLOG_H_
#define GLX_ORACLE_STUB_GLOG_H_
#include <iostream>
#include <sstream>
namespace glx_stub {
struct NullLog {
  template <class T> NullLog& operator<<(const T&) { return *this; }
  NullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
}  // namespace glx_stub
#define LOG(severity) ::glx_stub::NullLog()
#define VLOG(n) ::glx_stub::NullLog()
#define LOG_IF(severity, cond) ::glx_stub::NullLog()
#define CHECK(cond) ::glx_stub::NullLog()
#endif
