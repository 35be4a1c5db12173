// Residual r = rhs - L u of the 7-/5-point Poisson operator, zero on Dirichlet
// faces.  Reference: ndsm_optimized.f90:346-447 (3-D; evaluates
// (..)wx+(..)wy+(..)wz - rhs - u*wc and negates), ndsm_poisson.f90:280-353
// (2-D; rhs - sum_d (u_lo - 2u + u_hi) w_d).  The r = 0 pre-pass and the six
// face-zeroing array sections of the reference are folded into the one kernel.
// Algorithmic traffic: 24 B per point (u, rhs in; r out).
#include "common.hpp"

namespace {

// x-row per wave: lane -> consecutive i (fully coalesced); the y/z neighbours
// are whole-line loads served by L2 after their first touch.
__global__ __launch_bounds__(256) void residual3(const double *__restrict__ u, const double *__restrict__ rhs,
                                                 double *__restrict__ r, ndsmk_grid g) {
  const int nx = g.n[0], ny = g.n[1];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = g.zown0 + blockIdx.z;  // owned planes only; ghosts of r are filled by the halo exchange
  if (i >= nx || j >= ny) return;
  const size_t c = (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * (size_t)k);
  const int kg = k + g.k0;
  const bool inside = i >= g.lb[0] && i <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1] && k >= g.lb[2] && k <= g.ub[2];
  if (!inside) {  // Dirichlet face (or a z-slab ghost plane): enforced exactly
    r[c] = 0.0;
    return;
  }
  const size_t sy = (size_t)nx, sz = (size_t)nx * (size_t)ny;
  const double ul = u[i == 0 ? c + 1 : c - 1];
  const double uh = u[i == nx - 1 ? c - 1 : c + 1];
  const double vl = u[j == 0 ? c + sy : c - sy];
  const double vh = u[j == ny - 1 ? c - sy : c + sy];
  const double wl = u[kg == 0 ? c + sz : c - sz];
  const double wh = u[kg == g.nzg - 1 ? c - sz : c + sz];
  const double v = (ul + uh) * g.w[0] + (vl + vh) * g.w[1] + (wl + wh) * g.w[2] - (rhs ? rhs[c] : 0.0) - u[c] * g.wc;
  r[c] = -v;
}

__global__ __launch_bounds__(256) void residual2(const double *__restrict__ u, const double *__restrict__ rhs,
                                                 double *__restrict__ r, ndsmk_grid g) {
  const int nx = g.n[0], ny = g.n[1];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= nx || j >= ny) return;
  const size_t c = (size_t)i + (size_t)nx * (size_t)j;
  const bool inside = i >= g.lb[0] && i <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1];
  if (!inside) {
    r[c] = 0.0;
    return;
  }
  const size_t sy = (size_t)nx;
  const double uc = u[c];
  const double xl = u[i == 0 ? c + 1 : (i == nx - 1 ? c - 1 : c - 1)];
  const double xh = u[i == 0 ? c + 1 : (i == nx - 1 ? c - 1 : c + 1)];
  const double yl = u[j == 0 ? c + sy : (j == ny - 1 ? c - sy : c - sy)];
  const double yh = u[j == 0 ? c + sy : (j == ny - 1 ? c - sy : c + sy)];
  double lap = 0.0;  // ndsm_poisson.f90:334-345
  lap = lap + (xl - 2 * uc + xh) * g.w[0];
  lap = lap + (yl - 2 * uc + yh) * g.w[1];
  r[c] = (rhs ? rhs[c] : 0.0) - lap;
}

}  // namespace

extern "C" int ndsmk_residual(const ndsmk_grid *gp, const double *u, const double *rhs, double *r) {
  NDSM_REQUIRE_READY();
  const ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 2 || g.ndim == 3);
  NDSM_CHECK_ARG(g.n[0] >= 2 && g.n[1] >= 2 && (g.ndim == 2 ? g.n[2] == 1 : g.n[2] >= 2));
  dim3 block(64, 4, 1);
  dim3 grid((g.n[0] + 63) / 64, (g.n[1] + 3) / 4, g.ndim == 3 ? g.zown1 - g.zown0 : 1);
  if (g.ndim == 3)
    hipLaunchKernelGGL(residual3, grid, block, 0, ndsm::stream(), u, rhs, r, g);
  else
    hipLaunchKernelGGL(residual2, grid, block, 0, ndsm::stream(), u, rhs, r, g);
  NDSM_LAUNCH_CHECK();
  return 0;
}
