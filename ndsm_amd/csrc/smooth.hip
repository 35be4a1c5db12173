// Red-black Gauss-Seidel smoother for the 7-point (3-D) / 5-point (2-D) Poisson
// stencil with mirrored-ghost Neumann faces and fixed Dirichlet faces.
//
// Reference semantics: ndsm_optimized.f90:40-191 (3-D), ndsm_poisson.f90:451-619
// (2-D, incl. the all-Neumann mean shift).  Operand order is kept and the file
// is compiled with -ffp-contract=off, so a sweep is bit-identical to the
// reference built by its own Makefile (no FMA contraction on x86-64).
//
// Bandwidth model: 24 B per lattice update = u read + rhs read +
// u write, once per full red+black sweep.
//   rbgs3_color   : two launches per sweep, half-line utilisation -> baseline
//                   and fallback for tiny / oddly shaped levels
//   rbgs3_fused   : one launch per sweep (see smooth_fused.hip)
#include "common.hpp"

namespace ndsm {
int launch_rbgs3_fused(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, int max_sweeps,
                       bool force, int *sweeps_done, double *rout, int *res_done);
int launch_mean_shift(double *u, int64_t n);
}

namespace {

__device__ __forceinline__ size_t lin3(int i, int j, int k, int nx, int ny) {
  return (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * (size_t)k);
}

// One colour of one sweep.  Thread (tx, j, k) updates i = i0(j,k) + 2 tx, so a
// wave touches every second double of a row.
__global__ __launch_bounds__(256) void rbgs3_color(double *__restrict__ u, const double *__restrict__ rhs,
                                                  ndsmk_grid g, int par) {
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int j = g.lb[1] + blockIdx.y * blockDim.y + threadIdx.y;
  const int k = g.lb[2] + blockIdx.z;
  if (j > g.ub[1]) return;
  const int kg = k + g.k0;  // global plane index keeps the colouring global under z-slabs
  const int i0 = g.lb[0] + ((((g.lb[0] + j + kg) & 1) != par) ? 1 : 0);
  const int i = i0 + 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i > g.ub[0]) return;
  int xl = i - 1, xh = i + 1, yl = j - 1, yh = j + 1, zl = k - 1, zh = k + 1;
  if (xl < 0) xl = 1;
  if (xh > nx - 1) xh = nx - 2;
  if (yl < 0) yl = 1;
  if (yh > ny - 1) yh = ny - 2;
  // physical z faces only: a slab's interior edges read the ghost planes k-1 / k+1
  if (kg - 1 < 0) zl = k + 1;
  if (kg + 1 > g.nzg - 1) zh = k - 1;
  (void)nz;
  const double unew = (u[lin3(xh, j, k, nx, ny)] + u[lin3(xl, j, k, nx, ny)]) * g.w[0] +
                      (u[lin3(i, yh, k, nx, ny)] + u[lin3(i, yl, k, nx, ny)]) * g.w[1] +
                      (u[lin3(i, j, zh, nx, ny)] + u[lin3(i, j, zl, nx, ny)]) * g.w[2] -
                      (rhs ? rhs[lin3(i, j, k, nx, ny)] : 0.0);  // rhs == nullptr: identically zero
  u[lin3(i, j, k, nx, ny)] = g.w1 * unew;
}

// Small 3-D levels (<= 16^3 points): all `nsweeps` sweeps in ONE launch of ONE
// workgroup.  The level is L2-resident and a colour pass is shorter than a kernel
// boundary, so two launches per sweep are pure launch latency; inside a single
// workgroup __syncthreads() orders the colour passes (global writes of a block are
// visible to the block after the barrier).  Same update expression as rbgs3_color.
__global__ __launch_bounds__(1024) void rbgs3_small(double *__restrict__ u, const double *__restrict__ rhs,
                                                    ndsmk_grid g, int nsweeps) {
  const int nx = g.n[0], ny = g.n[1];
  const int mx = g.ub[0] - g.lb[0] + 1, my = g.ub[1] - g.lb[1] + 1, mz = g.ub[2] - g.lb[2] + 1;
  const int half = (mx + 1) / 2;
  const int total = half * my * mz;
  for (int sw = 0; sw < nsweeps; ++sw) {
    for (int pass = 0; pass < 2; ++pass) {
      const int par = (g.first_par + pass) & 1;
      for (int p = threadIdx.x; p < total; p += blockDim.x) {
        const int t = p % half, j = g.lb[1] + (p / half) % my, k = g.lb[2] + p / (half * my);
        const int i0 = g.lb[0] + ((((g.lb[0] + j + k) & 1) != par) ? 1 : 0);
        const int i = i0 + 2 * t;
        if (i > g.ub[0]) continue;
        int xl = i - 1, xh = i + 1, yl = j - 1, yh = j + 1, zl = k - 1, zh = k + 1;
        if (xl < 0) xl = 1;
        if (xh > nx - 1) xh = nx - 2;
        if (yl < 0) yl = 1;
        if (yh > ny - 1) yh = ny - 2;
        if (zl < 0) zl = 1;
        if (zh > g.n[2] - 1) zh = g.n[2] - 2;
        const double unew = (u[lin3(xh, j, k, nx, ny)] + u[lin3(xl, j, k, nx, ny)]) * g.w[0] +
                            (u[lin3(i, yh, k, nx, ny)] + u[lin3(i, yl, k, nx, ny)]) * g.w[1] +
                            (u[lin3(i, j, zh, nx, ny)] + u[lin3(i, j, zl, nx, ny)]) * g.w[2] -
                            (rhs ? rhs[lin3(i, j, k, nx, ny)] : 0.0);
        u[lin3(i, j, k, nx, ny)] = g.w1 * unew;
      }
      __syncthreads();
    }
  }
}

// 2-D colour pass (generic N-D path of the reference specialised to ndim = 2):
// red = (i+j) even in either index base (ndsm_poisson.f90:499-501).
__global__ __launch_bounds__(256) void rbgs2_color(double *__restrict__ u, const double *__restrict__ rhs,
                                                  ndsmk_grid g, int par) {
  const int nx = g.n[0], ny = g.n[1];
  const int j = g.lb[1] + blockIdx.y * blockDim.y + threadIdx.y;
  if (j > g.ub[1]) return;
  const int i0 = g.lb[0] + ((((g.lb[0] + j) & 1) != par) ? 1 : 0);
  const int i = i0 + 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i > g.ub[0]) return;
  // stencil_stride (ndsm_poisson.f90:633-658): both neighbours collapse onto
  // the inner one at a boundary
  const int xl = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i - 1);
  const int xh = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i + 1);
  const int yl = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j - 1);
  const int yh = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j + 1);
  double un = 0.0;  // ndsm_poisson.f90:603-617
  un = un + u[(size_t)xl + (size_t)nx * j] * g.w[0] + u[(size_t)xh + (size_t)nx * j] * g.w[0];
  un = un + u[(size_t)i + (size_t)nx * yl] * g.w[1] + u[(size_t)i + (size_t)nx * yh] * g.w[1];
  u[(size_t)i + (size_t)nx * j] = (un - (rhs ? rhs[(size_t)i + (size_t)nx * j] : 0.0)) * g.w1;
}

}  // namespace

static int relax_impl(const ndsmk_grid *gp, double *u, double *ualt, const double *rhs, int nsweeps, int variant,
                      int *result_in_alt, double *rout, int *res_done) {
  NDSM_REQUIRE_READY();
  if (res_done) *res_done = 0;
  const ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 2 || g.ndim == 3);
  NDSM_CHECK_ARG(g.n[0] >= 2 && g.n[1] >= 2 && (g.ndim == 2 ? g.n[2] == 1 : g.n[2] >= 2));
  for (int d = 0; d < g.ndim; ++d) NDSM_CHECK_ARG(g.lb[d] >= 0 && g.ub[d] <= g.n[d] - 1);
  NDSM_CHECK_ARG(variant >= 0 && variant <= 2);
  NDSM_CHECK_ARG(g.ndim == 2 || (g.zown0 >= 0 && g.zown1 <= g.n[2] && g.zown0 < g.zown1));
  const int64_t npts = (int64_t)g.n[0] * g.n[1] * g.n[2];
  hipStream_t s = ndsm::stream();
  const int mx = g.ub[0] - g.lb[0] + 1, my = g.ub[1] - g.lb[1] + 1, mz = g.ub[2] - g.lb[2] + 1;
  if (mx <= 0 || my <= 0 || (g.ndim == 3 && mz <= 0)) return 0;  // nothing to update
  double *const u_entry = u;
  if (result_in_alt) *result_in_alt = 0;
  // small 3-D level: every sweep in one single-workgroup launch
  if (g.ndim == 3 && variant == 0 && !g.all_neumann && npts <= 4096 && g.k0 == 0 && g.zown0 == 0 &&
      g.zown1 == g.n[2] && nsweeps > 0) {
    hipLaunchKernelGGL(rbgs3_small, dim3(1), dim3(1024), 0, s, u, rhs, g, nsweeps);
    NDSM_LAUNCH_CHECK();
    return 0;
  }
  for (int sw = 0; sw < nsweeps; ++sw) {
    if (g.ndim == 3) {
      bool done = false;
      if (variant != 1) {
        // all-Neumann levels shift the mean after EVERY sweep: one sweep per pass there
        int ndone = 0;
        // the residual rides on the last sweep (not on all-Neumann levels: the mean shift comes in between)
        int rc = ndsm::launch_rbgs3_fused(g, u, ualt, rhs, g.all_neumann ? 1 : nsweeps - sw, variant == 2, &ndone,
                                          g.all_neumann ? nullptr : rout, res_done);
        if (rc) return rc;
        done = ndone > 0;
        if (done) {  // the sweeps landed in the other array
          sw += ndone - 1;
          double *t = u;
          u = ualt;
          ualt = t;
        }
        if (!done && variant == 2)
          return ndsm::fail(NDSMK_EARG, "fused smoother does not support this level shape", __FILE__, __LINE__);
      }
      if (!done) {
        if (g.zown0 != 0 || g.zown1 != g.n[2])
          return ndsm::fail(NDSMK_EARG, "z-slab levels need the fused smoother (nx even, >= 16 x 16 x 8 owned)", __FILE__,
                            __LINE__);
        const int half = (mx + 1) / 2;
        dim3 block(64, 4, 1);
        dim3 grid((half + 63) / 64, (my + 3) / 4, mz);
        for (int pass = 0; pass < 2; ++pass) {
          hipLaunchKernelGGL(rbgs3_color, grid, block, 0, s, u, rhs, g, (g.first_par + pass) & 1);
          NDSM_LAUNCH_CHECK();
        }
      }
    } else {
      const int half = (mx + 1) / 2;
      dim3 block(64, 4, 1);
      dim3 grid((half + 63) / 64, (my + 3) / 4, 1);
      for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(rbgs2_color, grid, block, 0, s, u, rhs, g, (g.first_par + pass) & 1);
        NDSM_LAUNCH_CHECK();
      }
    }
    if (g.all_neumann) {
      int rc = ndsm::launch_mean_shift(u, npts);
      if (rc) return rc;
    }
  }
  if (u != u_entry) {
    if (result_in_alt) {
      *result_in_alt = 1;
    } else {  // caller cannot swap: bring the result home
      NDSM_HIP(hipMemcpyAsync(u_entry, u, sizeof(double) * (size_t)npts, hipMemcpyDeviceToDevice, s));
    }
  }
  return 0;
}

extern "C" int ndsmk_relax(const ndsmk_grid *gp, double *u, double *ualt, const double *rhs, int nsweeps,
                           int variant, int *result_in_alt) {
  return relax_impl(gp, u, ualt, rhs, nsweeps, variant, result_in_alt, nullptr, nullptr);
}

// nsweeps sweeps followed by r = rhs - L u: where the fused kernel covers the level
// the residual is produced by the launch of the last sweep, otherwise by residual.hip.
// variant 2 (tests): the fused sweep+residual launch or an error.
extern "C" int ndsmk_relax_residual(const ndsmk_grid *gp, double *u, double *ualt, const double *rhs, double *r,
                                    int nsweeps, int variant, int *result_in_alt) {
  NDSM_CHECK_ARG(r != nullptr && result_in_alt != nullptr && (variant == 0 || variant == 2));
  int res_done = 0;
  int rc = relax_impl(gp, u, ualt, rhs, nsweeps, variant, result_in_alt, r, &res_done);
  if (rc) return rc;
  if (!res_done) {
    if (variant == 2)
      return ndsm::fail(NDSMK_EARG, "fused sweep + residual does not cover this level", __FILE__, __LINE__);
    rc = ndsmk_residual(gp, *result_in_alt ? ualt : u, rhs, r);
  }
  return rc;
}
