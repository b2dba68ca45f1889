// Shared host-side helpers of libjspsr_hip.so (gfx950 only; no other targets, no shims).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/jspsr_hip.h"

namespace jspsr {

// Per-thread error text behind jspsr_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
int conv_dynq_override();      // jspsr_conv_dynamic_queue(): -1 = the environment's default, 0 / 1 = forced

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// Workgroups b and b+8 share an XCD (private 4 MiB L2) under the observed round-robin dispatch.
// Bijective remap that hands every XCD one contiguous run of logical tile ids, so that
// neighbouring tiles (which share halo rows / operand panels) hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_contiguous(int bid, int nblk) {
  if (nblk < 16) return bid;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// Exact n / d for every 32-bit n by one multiply-high and two shifts (Granlund & Montgomery's round-up form): the per-tile
// index arithmetic of the tile kernels divides by launch constants, and a runtime 32-bit division costs ~30 dependent
// instructions at the head of every workgroup.  The host makes the constants (make_fastdiv), the kernel applies them.
struct FastDiv {
  unsigned m, s1, s2;
};
inline FastDiv make_fastdiv(unsigned d) {      // d >= 1
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                 // ceil(log2 d)
  FastDiv f;
  f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}
__host__ __device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned t = __umulhi(f.m, n);
#else
  const unsigned t = (unsigned)(((unsigned long long)f.m * n) >> 32);
#endif
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

}  // namespace jspsr
