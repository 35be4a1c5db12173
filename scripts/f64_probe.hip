// fp64 VALU issue-rate probe (dev aid): separate mul+add (as the bit-exact kernels need) vs fma
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double b, double c, int n) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {  // mul then add, not contracted
      a0 = a0 * b; a1 = a1 * b; a2 = a2 * b; a3 = a3 * b; a4 = a4 * b; a5 = a5 * b; a6 = a6 * b; a7 = a7 * b;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    } else {
      a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
      a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  double *o; CK(hipMalloc(&o, 8 * 256 * 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 20000;
  for (int wgs : {256, 1024, 2048}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) k<0><<<wgs, 256>>>(o, 1.0000001, 1e-9, n); else k<1><<<wgs, 256>>>(o, 1.0000001, 1e-9, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double instr = (double)wgs * 4 /*waves*/ * n * (mode == 0 ? 16 : 8);  // wave-instructions
      // per SIMD: waves per SIMD = wgs*4/1024 (256 CUs x 4 SIMDs)
      printf("wgs %4d %s: %.2f ms  %.2f G wave-instr/s  = %.2f cycles/instr/SIMD at 2.4 GHz\n", wgs, mode == 0 ? "mul+add" : "fma    ",
             best, instr / best / 1e6, 2.4e9 * 1024 / (instr / (best * 1e-3)));
    }
  }
  return 0;
}
