// Coarsest-grid "exact" solve as ONE single-workgroup kernel.
//
// Reference: solve_exact, ndsm_multigrid_core.f90:728-800 - u_sav = 0; repeat at
// most nmax_exact times { if du <= ex_tol exit; relax; du = max (or mean) of
// |u_sav - u|; u_sav = u }.  On the CPU that is 12-21 sweeps of a 4^3..8x8x4
// grid per V-cycle; driven from the host it would be as many kernel launches
// and blocking metric read-backs.  Here the grid, rhs and u_sav live in LDS,
// the data-dependent exit test is evaluated by the workgroup itself, and the
// host is not involved: the sweep count / non-convergence flag are accumulated
// in device memory and read back only when the caller asks.
//
// The update expressions are the same source expressions as smooth.hip, so the
// result is bit-identical to running the level kernels sweep by sweep.
#include "common.hpp"

namespace {

constexpr int kMaxPts = 2048;  // 3 x 16 KiB of LDS
constexpr int kThreads = 256;

__device__ __forceinline__ double block_max(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_sum(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v = v + __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kThreads) void solve_exact_k(double *__restrict__ u_g, const double *__restrict__ rhs_g,
                                                         ndsmk_grid g, double ex_tol, int use_max, int nmax,
                                                         long long *__restrict__ info) {
  __shared__ double u[kMaxPts], rhs[kMaxPts], sav[kMaxPts];
  __shared__ double red[4];
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int n = nx * ny * nz;
  for (int p = threadIdx.x; p < n; p += kThreads) {
    u[p] = u_g[p];
    rhs[p] = rhs_g ? rhs_g[p] : 0.0;
    sav[p] = 0.0;
  }
  __syncthreads();
  double du = 1.79769313486231570815e308;
  int sweeps = 0, converged = 0;
  for (int it = 0; it < nmax; ++it) {
    if (du <= ex_tol) {  // uniform: every thread holds the same du
      converged = 1;
      break;
    }
    for (int pass = 0; pass < 2; ++pass) {
      const int par = (g.first_par + pass) & 1;
      for (int p = threadIdx.x; p < n; p += kThreads) {
        const int i = p % nx, j = (p / nx) % ny, k = p / (nx * ny);
        if (((i + j + k) & 1) != par) continue;
        if (i < g.lb[0] || i > g.ub[0] || j < g.lb[1] || j > g.ub[1]) continue;
        if (g.ndim == 3) {
          if (k < g.lb[2] || k > g.ub[2]) continue;
          int xl = i - 1, xh = i + 1, yl = j - 1, yh = j + 1, zl = k - 1, zh = k + 1;
          if (xl < 0) xl = 1;
          if (xh > nx - 1) xh = nx - 2;
          if (yl < 0) yl = 1;
          if (yh > ny - 1) yh = ny - 2;
          if (zl < 0) zl = 1;
          if (zh > nz - 1) zh = nz - 2;
          const double unew = (u[xh + nx * (j + ny * k)] + u[xl + nx * (j + ny * k)]) * g.w[0] +
                              (u[i + nx * (yh + ny * k)] + u[i + nx * (yl + ny * k)]) * g.w[1] +
                              (u[i + nx * (j + ny * zh)] + u[i + nx * (j + ny * zl)]) * g.w[2] - rhs[p];
          u[p] = g.w1 * unew;
        } else {
          const int xl = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i - 1);
          const int xh = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i + 1);
          const int yl = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j - 1);
          const int yh = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j + 1);
          double un = 0.0;
          un = un + u[xl + nx * j] * g.w[0] + u[xh + nx * j] * g.w[0];
          un = un + u[i + nx * yl] * g.w[1] + u[i + nx * yh] * g.w[1];
          u[p] = (un - rhs[p]) * g.w1;
        }
      }
      __syncthreads();
    }
    if (g.all_neumann) {
      double s = 0.0;
      for (int p = threadIdx.x; p < n; p += kThreads) s = s + u[p];
      const double mean = block_sum(s, red) / (double)n;
      for (int p = threadIdx.x; p < n; p += kThreads) u[p] = u[p] - mean;
      __syncthreads();
    }
    double mx = 0.0, sm = 0.0;
    for (int p = threadIdx.x; p < n; p += kThreads) {
      const double d = fabs(sav[p] - u[p]);
      mx = fmax(mx, d);
      sm = sm + d;
      sav[p] = u[p];
    }
    mx = block_max(mx, red);
    sm = block_sum(sm, red);
    du = use_max ? mx : sm / (double)n;
    ++sweeps;
  }
  for (int p = threadIdx.x; p < n; p += kThreads) u_g[p] = u[p];
  if (threadIdx.x == 0) {
    info[0] += sweeps;
    info[1] += converged ? 0 : 1;
  }
}

}  // namespace

namespace ndsm {

int launch_solve_exact_device(const ndsmk_grid &g, double *u, const double *rhs, double ex_tol, int use_max,
                              int nmax, long long *d_info, bool *handled) {
  const int64_t n = (int64_t)g.n[0] * g.n[1] * g.n[2];
  *handled = false;
  // the in-LDS colouring is the plain checkerboard; a z-slab never owns the coarsest grid
  if (n > kMaxPts || g.k0 != 0 || g.nzg != g.n[2]) return 0;
  hipLaunchKernelGGL(solve_exact_k, dim3(1), dim3(kThreads), 0, stream(), u, rhs, g, ex_tol, use_max, nmax, d_info);
  NDSM_LAUNCH_CHECK();
  *handled = true;
  return 0;
}

}  // namespace ndsm

// does the coarsest-grid solve of this level run as ONE device launch (no host loop, no read-backs)?
extern "C" int ndsmk_solve_exact_on_device(const ndsmk_grid *gp) {
  const int64_t n = (int64_t)gp->n[0] * gp->n[1] * gp->n[2];
  return (n <= kMaxPts && gp->k0 == 0 && gp->nzg == gp->n[2]) ? 1 : 0;
}
