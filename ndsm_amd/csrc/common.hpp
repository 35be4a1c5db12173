// Shared by the HIP translation units of libndsm_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "ndsm_kernels.h"

namespace ndsm {

// one library stream; every kernel, copy and RCCL call is ordered on it
hipStream_t stream();
bool ready();
int cu_count();
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
