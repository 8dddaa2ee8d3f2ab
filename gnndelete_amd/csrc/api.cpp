// ABI bookkeeping: version + thread-local error string.
#include "common.h"

namespace gd {
char* error_buffer() {
  static thread_local char buf[256] = "";
  return buf;
}
}  // namespace gd

extern "C" int gd_abi_version(void) { return GD_ABI_VERSION; }
extern "C" const char* gd_last_error_string(void) { return gd::error_buffer(); }
