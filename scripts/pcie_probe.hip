// PCIe probe (dev aid): pageable hipMemcpy with 1..8 threads on disjoint chunks, hipHostRegister cost,
// pinned throughput.  build: hipcc -O2 --offload-arch=gfx950 scripts/pcie_probe.hip -o scripts/bin/pcie_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
  const size_t n = (size_t)4 << 30;
  char *h = (char *)malloc(n);
  memset(h, 1, n);
  char *d;
  CK(hipMalloc((void **)&d, n));
  CK(hipMemset(d, 2, n));
  CK(hipDeviceSynchronize());
  for (int dir = 0; dir < 2; ++dir) {
    for (int nt : {1, 2, 4, 8}) {
      for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
          th.emplace_back([=]() {
            hipStream_t s;
            CK(hipSetDevice(0));
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            size_t c = n / nt, o = c * t;
            if (dir == 0) CK(hipMemcpyAsync(h + o, d + o, c, hipMemcpyDeviceToHost, s));
            else CK(hipMemcpyAsync(d + o, h + o, c, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            CK(hipStreamDestroy(s));
          });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        if (rep) printf("%s pageable %d threads: %.1f GB/s\n", dir ? "H2D" : "D2H", nt, n / dt / 1e9);
      }
    }
  }
  // fresh (untouched) destination pages, as numpy.zeros hands them over
  {
    char *h2 = (char *)malloc(n);
    double t0 = now();
    CK(hipMemcpy(h2, d, n, hipMemcpyDeviceToHost));
    printf("D2H into untouched pages, 1 thread: %.1f GB/s\n", n / (now() - t0) / 1e9);
    free(h2);
  }
  double t0 = now();
  CK(hipHostRegister(h, n, hipHostRegisterDefault));
  printf("hipHostRegister 4 GiB: %.3f s\n", now() - t0);
  for (int dir = 0; dir < 2; ++dir) {
    t0 = now();
    if (dir == 0) CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); else CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    printf("%s pinned: %.1f GB/s\n", dir ? "H2D" : "D2H", n / (now() - t0) / 1e9);
  }
  t0 = now();
  CK(hipHostUnregister(h));
  printf("hipHostUnregister: %.3f s\n", now() - t0);
  return 0;
}
