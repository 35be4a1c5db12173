// Shared by the HIP translation units of libndsm_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "ndsm_kernels.h"

namespace ndsm {

// one library stream; every kernel, copy and RCCL call is ordered on it
hipStream_t stream();
int lane();   // the selected lane (ndsmk_select_lane), -1: none
bool ready();
int cu_count();
// Device-bound process state.  epoch() changes every time the runtime comes up on a device
// (ndsmk_init after ndsmk_shutdown, or a re-target): per-kernel attributes cached in function statics
// are re-issued when their stored epoch differs.  at_reset(fn): fn runs once at the next shutdown /
// re-target, while the old device is still current and its streams are drained - scratch allocations
// that live in a translation unit's globals register their release there.
int epoch();
void at_reset(void (*fn)());
// true once per epoch and call site: `static int ep = 0; if (ndsm::first_in_epoch(ep)) { ... }`
inline bool first_in_epoch(int &seen) {
  const int e = epoch();
  if (seen == e) return false;
  seen = e;
  return true;
}
int fail(int code, const char *what, const char *file, int line);
int not_ready(const char *file, int line);

constexpr int kWave = 64;   // CDNA wavefront

}  // namespace ndsm

#define NDSM_HIP(call)                                                               \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (void)hipGetLastError(); /* do not leave it sticky for the next launch check */ \
      return ndsm::fail((int)e_, hipGetErrorString(e_), __FILE__, __LINE__);         \
    }                                                                                \
  } while (0)

#define NDSM_REQUIRE_READY()                                   \
  do {                                                         \
    if (!ndsm::ready()) return ndsm::not_ready(__FILE__, __LINE__); \
  } while (0)

#define NDSM_CHECK_ARG(cond)                                                     \
  do {                                                                           \
    if (!(cond)) return ndsm::fail(NDSMK_EARG, "argument check failed: " #cond, __FILE__, __LINE__); \
  } while (0)

#define NDSM_LAUNCH_CHECK() NDSM_HIP(hipGetLastError())
