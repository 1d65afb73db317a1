// Error reporting for libsagnn.so: a thread-local message behind sagnn_last_error().
#include "common.h"

#include <stdarg.h>
#include <string.h>

namespace sagnn {

std::string& last_error_slot() {
  static thread_local std::string slot;
  return slot;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_slot() = buf;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
  last_error_slot() = buf;
  return (int)e;
}

}  // namespace sagnn

extern "C" int sagnn_version(void) { return SAGNN_VERSION; }

extern "C" size_t sagnn_last_error(char* buf, size_t cap) {
  const std::string& s = sagnn::last_error_slot();
  if (buf && cap > 0) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = '\0';
  }
  return s.size();
}
