// Error plumbing of the C ABI (include/jspsr_hip.h).
#include "common.h"

#include <atomic>
#include <cstring>

namespace jspsr {

// Launch census behind jspsr_launch_count(): every launch passes its name (a string literal) through check_launch();
// a slot per distinct literal, found by pointer.  Diagnostics only (tests assert that a shape really took the kernel it
// is meant to take); ~10 ns per launch.
namespace {
struct Slot {
  std::atomic<const char*> name{nullptr};
  std::atomic<long long> n{0};
};
Slot g_slots[128];

void count_launch(const char* what) {
  for (Slot& s : g_slots) {
    const char* cur = s.name.load(std::memory_order_acquire);
    if (cur == nullptr) {
      const char* expect = nullptr;
      if (s.name.compare_exchange_strong(expect, what, std::memory_order_acq_rel)) cur = what; else cur = expect;
    }
    if (cur == what) { s.n.fetch_add(1, std::memory_order_relaxed); return; }
  }
}
}  // namespace

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
  count_launch(what);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
  return JSPSR_OK;
}

}  // namespace jspsr

namespace jspsr {
std::atomic<int> g_conv_dynq{-1};
int conv_dynq_override() { return g_conv_dynq.load(std::memory_order_relaxed); }
}  // namespace jspsr

extern "C" int jspsr_abi_version(void) { return 15; }
extern "C" int jspsr_conv_dynamic_queue(int on) { return jspsr::g_conv_dynq.exchange(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed); }
extern "C" long long jspsr_launch_count(const char* what) {
  if (!what) return -1;
  long long n = 0;
  for (jspsr::Slot& s : jspsr::g_slots) {
    const char* cur = s.name.load(std::memory_order_acquire);
    if (cur && strcmp(cur, what) == 0) n += s.n.load(std::memory_order_relaxed);
  }
  return n;
}
extern "C" const char* jspsr_last_error(void) { return jspsr::err_buf(); }
