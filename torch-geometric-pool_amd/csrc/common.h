// Shared host/device helpers for the tgp HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tgp_hip.h"

// eps (reference tgp/__init__.py:6, 1e-8) is an ARGUMENT of every entry point that uses it: the reference reads the
// module global at call time (utils/ops.py:72,318,377,395; utils/losses.py:498), so a caller may have changed it.
#define WAVE 64

namespace tgp {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TGP_ERR_LAUNCH;
  }
  return TGP_OK;
}

#define TGP_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      tgp::set_error(__VA_ARGS__);   \
      return (code);                 \
    }                                \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Carves aligned sub-buffers out of the caller's workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off = align_up(off + count * sizeof(T));
    return p;
  }
};

inline int cdiv(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md "Workgroup
// dispatch"); give each XCD a contiguous chunk of the logical grid so neighbouring tiles hit
// the same L2.  Bijective for any nwg (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

}  // namespace tgp
