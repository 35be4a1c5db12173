// host <-> device round-trip probe (dev aid): what does it cost to learn a 16-byte result of a tiny kernel?
//   a) kernel -> hipMemcpyAsync(16 B, D2H, pinned) -> hipStreamSynchronize      (what ndsmk_diff_metrics does)
//   b) kernel writes the pair + a sequence number into mapped pinned memory -> hipStreamSynchronize
//   c) as b), the host spins on the sequence number instead of synchronising the stream
// build: hipcc -O2 --offload-arch=gfx950 scripts/sync_probe.hip -o scripts/bin/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void produce(double *out, double v) { out[0] = v; out[1] = 2 * v; }
__global__ void produce_seq(volatile double *out, volatile unsigned long long *seq, double v, unsigned long long s) {
  out[0] = v; out[1] = 2 * v;
  __threadfence_system();
  *seq = s;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  double *d; CK(hipMalloc(&d, 16));
  double *pin; CK(hipHostMalloc(&pin, 64, hipHostMallocMapped));
  double *dpin; CK(hipHostGetDevicePointer((void **)&dpin, pin, 0));
  volatile unsigned long long *seq = (volatile unsigned long long *)(pin + 4);
  unsigned long long *dseq = (unsigned long long *)(dpin + 4);
  *seq = 0;
  const int N = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 1; i <= N; ++i) {
        if (mode == 0) {
          hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, st, d, (double)i);
          CK(hipMemcpyAsync(pin, d, 16, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
        } else if (mode == 1) {
          hipLaunchKernelGGL(produce_seq, dim3(1), dim3(64), 0, st, dpin, dseq, (double)i, (unsigned long long)(rep * N + i + mode * 100000));
          CK(hipStreamSynchronize(st));
        } else {
          const unsigned long long want = (unsigned long long)(rep * N + i + mode * 100000);
          hipLaunchKernelGGL(produce_seq, dim3(1), dim3(64), 0, st, dpin, dseq, (double)i, want);
          while (*seq != want) { }
        }
        if (pin[0] != (double)i) { printf("mode %d: wrong value\n", mode); return 1; }
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      if (rep) printf("mode %c: %.1f us per round trip\n", "abc"[mode], us);
    }
    CK(hipStreamSynchronize(st));
  }
  return 0;
}
