// do single-workgroup kernels on different streams overlap? (dev aid)  build: hipcc -O2 --offload-arch=gfx950 scripts/concurrency_probe.hip -o scripts/bin/concurrency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
extern __shared__ double dyn[];
__global__ void spin(double *out, long long cycles, int use_lds) {
  const long long t0 = wall_clock64();
  double a = 0;
  if (use_lds) dyn[threadIdx.x] = 1.0;
  while (wall_clock64() - t0 < cycles) a += 1.0;
  if (use_lds) a += dyn[(threadIdx.x + 1) % blockDim.x];
  if (a < 0) out[0] = a;
}
int main() {
  const int NS = 6;
  hipStream_t st[NS];
  for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  double *d; CK(hipMalloc(&d, 64));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const long long cyc = 100 * 100;  // wall_clock64: 100 MHz -> 100 us
  for (int lds = 0; lds < 2; ++lds)
    for (int ns = 1; ns <= NS; ns += 5) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 10; ++k)
          for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(1024), lds ? 150 * 1024 : 0, st[i], d, cyc, lds);
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep) printf("lds %d, %d streams x 10 kernels of 100 us: %.0f us\n", lds, ns, us);
      }
    }
  return 0;
}
