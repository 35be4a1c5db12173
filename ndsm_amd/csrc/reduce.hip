// Reductions of the V-cycle driver:
//   diff metrics  max|a-b|, sum|a-b| (+ copy)   update_u  ndsm_multigrid_core.f90:1077-1122
//                                               du_metrics               :808-853
//   mean shift    u -= mean(u)                  all-Neumann solves       :1199-1223,
//                                               ndsm_poisson.f90:534-547
//   solve_exact   coarsest-grid iteration       ndsm_multigrid_core.f90:728-800
//
// Two-stage, fixed-shape tree reductions: results are reproducible run to run
// (the reference's OpenMP `+` reductions are not).  max is
// order-independent, so the default convergence metric is exact.
#include "common.hpp"

namespace {

constexpr int kRedBlock = 256;
constexpr int kRedMaxBlocks = 2048;  // 256 CUs x 8 blocks (guide: grid-stride past that)

struct Scratch {
  double *d_part = nullptr;  // [2 * kRedMaxBlocks + 2]
  double *h_pin = nullptr;   // [4] pinned
};
Scratch g_sl[NDSMK_LANES + 1];   // [0] the main stream's, [1 + l] lane l's (ndsmk_select_lane): lanes run concurrently
#define g_s (g_sl[ndsm::lane() + 1])

void scratch_release() {
  for (auto &s : g_sl) {
    if (s.d_part) (void)hipFree(s.d_part);
    if (s.h_pin) (void)hipHostFree(s.h_pin);
    s = Scratch();
  }
}
int ensure_scratch() {
  if (g_s.d_part) return 0;
  ndsm::at_reset(scratch_release);
  NDSM_HIP(hipMalloc((void **)&g_s.d_part, sizeof(double) * (2 * kRedMaxBlocks + 8)));
  NDSM_HIP(hipHostMalloc((void **)&g_s.h_pin, sizeof(double) * 8, hipHostMallocDefault));
  return 0;
}

__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v = v + __shfl_down(v, o, 64);
  return v;
}

// stage 1: per-block (max, sum) of |a-b| over a grid-stride range; optional b <- a
__global__ __launch_bounds__(kRedBlock) void diff_stage1(const double *__restrict__ a, double *__restrict__ b,
                                                         int64_t n, int copy, double *__restrict__ part) {
  __shared__ double smx[kRedBlock / 64], ssm[kRedBlock / 64];
  double mx = 0.0, sm = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double av = a[i];
    const double d = fabs(av - b[i]);
    mx = fmax(mx, d);
    sm = sm + d;
    if (copy) b[i] = av;
  }
  mx = wave_max(mx);
  sm = wave_sum(sm);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    smx[wv] = mx;
    ssm[wv] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smx[0], s = ssm[0];
    for (int w = 1; w < kRedBlock / 64; ++w) {
      m = fmax(m, smx[w]);
      s = s + ssm[w];
    }
    part[2 * blockIdx.x] = m;
    part[2 * blockIdx.x + 1] = s;
  }
}

// stage 2: one block folds the partials in index order
__global__ __launch_bounds__(kRedBlock) void diff_stage2(const double *__restrict__ part, int nblocks,
                                                         double *__restrict__ out2) {
  __shared__ double smx[kRedBlock / 64], ssm[kRedBlock / 64];
  double mx = 0.0, sm = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    mx = fmax(mx, part[2 * i]);
    sm = sm + part[2 * i + 1];
  }
  mx = wave_max(mx);
  sm = wave_sum(sm);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    smx[wv] = mx;
    ssm[wv] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smx[0], s = ssm[0];
    for (int w = 1; w < kRedBlock / 64; ++w) {
      m = fmax(m, smx[w]);
      s = s + ssm[w];
    }
    out2[0] = m;
    out2[1] = s;
  }
}

__global__ __launch_bounds__(kRedBlock) void sum_stage1(const double *__restrict__ a, int64_t n,
                                                        double *__restrict__ part) {
  __shared__ double ssm[kRedBlock / 64];
  double sm = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) sm = sm + a[i];
  sm = wave_sum(sm);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) ssm[wv] = sm;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = ssm[0];
    for (int w = 1; w < kRedBlock / 64; ++w) s = s + ssm[w];
    part[blockIdx.x] = s;
  }
}

// second stage of the mean + the subtraction in one launch: EVERY block folds the partials of sum_stage1 in index
// order with the same strides and the same tree (the same bits in every block, and the bits a single folding
// block followed by a subtraction kernel gave), then subtracts over its grid-stride range.  One launch less per
// sweep of an all-Neumann level; nb <= 2048 partials are L2 hits.
__global__ __launch_bounds__(kRedBlock) void mean_shift_k(double *__restrict__ u, int64_t n, const double *__restrict__ part,
                                                          int nblocks) {
  __shared__ double ssm[kRedBlock / 64];
  __shared__ double s_mean;
  double sm = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) sm = sm + part[i];
  sm = wave_sum(sm);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) ssm[wv] = sm;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = ssm[0];
    for (int w = 1; w < kRedBlock / 64; ++w) s = s + ssm[w];
    s_mean = s / (double)n;
  }
  __syncthreads();
  const double m = s_mean;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) u[i] = u[i] - m;
}

int nblocks_for(int64_t n) {
  int64_t b = (n + kRedBlock - 1) / kRedBlock;
  if (b > kRedMaxBlocks) b = kRedMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

namespace ndsm {

int launch_mean_shift(double *u, int64_t n) {
  if (int rc = ensure_scratch()) return rc;
  const int nb = nblocks_for(n);
  hipStream_t s = stream();
  hipLaunchKernelGGL(sum_stage1, dim3(nb), dim3(kRedBlock), 0, s, u, n, g_s.d_part);
  NDSM_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_shift_k, dim3(nb), dim3(kRedBlock), 0, s, u, n, g_s.d_part, nb);
  NDSM_LAUNCH_CHECK();
  return 0;
}

}  // namespace ndsm

// the two halves of ndsmk_diff_metrics: enqueue (on the selected stream / lane, result on its way to that
// lane's pinned pair) and collect (waits for that stream).  Between the two the caller may enqueue on OTHER lanes.
extern "C" int ndsmk_diff_metrics_begin(const double *a, double *b, int64_t n, int copy) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(n > 0);
  if (int rc = ensure_scratch()) return rc;
  const int nb = nblocks_for(n);
  hipStream_t s = ndsm::stream();
  double *out = g_s.d_part + 2 * kRedMaxBlocks + 2;
  hipLaunchKernelGGL(diff_stage1, dim3(nb), dim3(kRedBlock), 0, s, a, b, n, copy, g_s.d_part);
  NDSM_LAUNCH_CHECK();
  hipLaunchKernelGGL(diff_stage2, dim3(1), dim3(kRedBlock), 0, s, g_s.d_part, nb, out);
  NDSM_LAUNCH_CHECK();
  NDSM_HIP(hipMemcpyAsync(g_s.h_pin, out, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  return 0;
}
extern "C" int ndsmk_diff_metrics_end(double *h_out2) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(g_s.h_pin != nullptr);
  NDSM_HIP(hipStreamSynchronize(ndsm::stream()));
  h_out2[0] = g_s.h_pin[0];
  h_out2[1] = g_s.h_pin[1];
  return 0;
}
extern "C" int ndsmk_diff_metrics(const double *a, double *b, int64_t n, int copy, double *h_out2) {
  if (int rc = ndsmk_diff_metrics_begin(a, b, n, copy)) return rc;
  return ndsmk_diff_metrics_end(h_out2);
}

// Coarsest-grid solve.  Normal case: the single-workgroup LDS kernel in
// coarse.hip (no host round trips).  Grids too large for it fall back to this
// host-driven loop: one relax + one blocking metric per sweep (still all on
// the device - there is no CPU arithmetic path).
namespace ndsm {
int launch_solve_exact_device(const ndsmk_grid &g, double *u, const double *rhs, double ex_tol, int use_max,
                              int nmax, long long *d_info, bool *handled);
}

extern "C" int ndsmk_solve_exact(const ndsmk_grid *gp, double *u, const double *rhs, double *scratch,
                                 double ex_tol, int use_max, int nmax, int64_t *d_info) {
  NDSM_REQUIRE_READY();
  const ndsmk_grid g = *gp;
  const int64_t n = (int64_t)g.n[0] * g.n[1] * g.n[2];
  bool handled = false;
  if (int rc = ndsm::launch_solve_exact_device(g, u, rhs, ex_tol, use_max, nmax, (long long *)d_info, &handled))
    return rc;
  if (handled) return 0;
  NDSM_HIP(hipMemsetAsync(scratch, 0, sizeof(double) * (size_t)n, ndsm::stream()));  // u_sav = 0
  double du = 1.79769313486231570815e308, m[2];
  int64_t sweeps = 0, converged = 0;
  for (int it = 0; it < nmax; ++it) {
    if (du <= ex_tol) {  // test first (ndsm_multigrid_core.f90:771-774)
      converged = 1;
      break;
    }
    if (int rc = ndsmk_relax(gp, u, nullptr, rhs, 1, 1, nullptr)) return rc;
    if (int rc = ndsmk_diff_metrics(u, scratch, n, 1, m)) return rc;  // metrics, then u_sav <- u
    du = use_max ? m[0] : m[1] / (double)n;
    ++sweeps;
  }
  int64_t info[2];
  if (int rc = ndsmk_d2h(info, d_info, sizeof(info))) return rc;
  info[0] += sweeps;
  info[1] += converged ? 0 : 1;
  return ndsmk_h2d(d_info, info, sizeof(info));
}
