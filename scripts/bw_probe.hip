// HBM streaming probe (dev aid): what do plain copy / read / write kernels reach on this device?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void copy_k(const double2 *a, double2 *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_k(const double2 *a, double *out, size_t n) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e-300) out[0] = s;
}
__global__ void write_k(double2 *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_double2(1.0, 2.0);
}
__global__ void copy2_k(const double2 *a, const double2 *c, double2 *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 x = a[i], y = c[i]; b[i] = make_double2(x.x + y.x, x.y + y.y); }
}
int main() {
  const size_t n = (size_t)512 * 512 * 512 / 2;  // double2 elements of one 512^3 field
  double2 *a, *b, *c; double *o;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&o, 8));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) {
          if (mode == 0) copy_k<<<grid, 256>>>(a, b, n);
          if (mode == 1) read_k<<<grid, 256>>>(a, o, n);
          if (mode == 2) write_k<<<grid, 256>>>(b, n);
          if (mode == 3) copy2_k<<<grid, 256>>>(a, c, b, n);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        if (ms < best) best = ms;
      }
      const double bytes = (mode == 0 ? 2.0 : mode == 3 ? 3.0 : 1.0) * n * 16;
      printf("grid %5d %s: %.1f us  %.2f TB/s\n", grid, mode == 0 ? "copy " : mode == 1 ? "read " : mode == 2 ? "write" : "2r+1w", best * 1e3, bytes / best / 1e9);
    }
  }
  return 0;
}
