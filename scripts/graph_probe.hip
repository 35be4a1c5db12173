// what does replaying a recorded chain of small kernels cost the host and the device? (dev aid)
// build: hipcc -O2 --offload-arch=gfx950 scripts/graph_probe.hip -o scripts/bin/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(double *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int NS = 6;
  hipStream_t st[NS];
  for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  double *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  for (int nk : {10, 70}) {
    hipGraph_t g; hipGraphExec_t ge[NS];
    for (int i = 0; i < NS; ++i) {
      CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, st[i], d + 8 * i);
      CK(hipStreamEndCapture(st[i], &g));
      CK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
    }
    for (int lanes : {1, 6}) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < lanes; ++i) CK(hipGraphLaunch(ge[i], st[i]));
        const double t1 = now();
        CK(hipDeviceSynchronize());
        const double t2 = now();
        // the same as plain launches
        const double t3 = now();
        for (int i = 0; i < lanes; ++i)
          for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, st[i], d + 8 * i);
        const double t4 = now();
        CK(hipDeviceSynchronize());
        const double t5 = now();
        if (rep == 2)
          printf("%2d kernels x %d lanes: graph launch %.0f us host, drained after %.0f us | plain enqueue %.0f us host, drained after %.0f us\n",
                 nk, lanes, t1 - t0, t2 - t0, t4 - t3, t5 - t3);
      }
    }
    for (int i = 0; i < NS; ++i) CK(hipGraphExecDestroy(ge[i]));
  }
  return 0;
}
