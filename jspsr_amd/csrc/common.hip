// Error plumbing of the C ABI (include/jspsr_hip.h).
#include "common.h"

namespace jspsr {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
  return JSPSR_OK;
}

}  // namespace jspsr

extern "C" int jspsr_abi_version(void) { return 12; }
extern "C" const char* jspsr_last_error(void) { return jspsr::err_buf(); }
