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
void fused_metric_accumulate(bool on);
int launch_rbgs3_fused(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, int max_sweeps,
                       bool force, int *sweeps_done, double *rout, int *res_done, const double *prev, int *met_done,
                       const ndsmk_xfer *px, const double *uc);
int fetch_fused_metric(double *h_out2);
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
__global__ __launch_bounds__(1024) void rbgs3_small(double *__restrict__ u_g, const double *__restrict__ rhs_g,
                                                    ndsmk_grid g, int nsweeps) {
  // the whole level (<= 4096 points) and its right-hand side sit in LDS for the duration
  __shared__ double u[4096], rhs[4096];
  const int nx = g.n[0], ny = g.n[1];
  const int n = nx * ny * g.n[2];
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    u[p] = u_g[p];
    rhs[p] = rhs_g ? rhs_g[p] : 0.0;
  }
  __syncthreads();
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
                            rhs[lin3(i, j, k, nx, ny)];
        u[lin3(i, j, k, nx, ny)] = g.w1 * unew;
      }
      __syncthreads();
    }
  }
  for (int p = threadIdx.x; p < n; p += blockDim.x) u_g[p] = u[p];
}

// Small 2-D level (<= 4096 points): every sweep of a relax call - both colour passes and, on
// all-Neumann problems, the mean shift after each sweep (ndsm_poisson.f90:534-547) - in ONE
// single-workgroup launch.  The six face solves of the vector potential are pure dispatch
// latency on their coarse levels (5 of the 8 levels of a 512^2 face); this takes ~30 launches
// per level and V-cycle down to 3.  Same update expressions as rbgs2_color; the mean is a fixed
// tree over the workgroup (the reference's own sum is an unordered OpenMP reduction).
__global__ __launch_bounds__(1024) void rbgs2_small(double *__restrict__ u_g, const double *__restrict__ rhs_g,
                                                    ndsmk_grid g, int nsweeps) {
  __shared__ double u[4096], rhs[4096];  // the whole level in LDS
  __shared__ double red[16];
  const int nx = g.n[0], ny = g.n[1];
  const int n = nx * ny;
  for (int p = threadIdx.x; p < n; p += blockDim.x) {
    u[p] = u_g[p];
    rhs[p] = rhs_g ? rhs_g[p] : 0.0;
  }
  __syncthreads();
  const int mx = g.ub[0] - g.lb[0] + 1, my = g.ub[1] - g.lb[1] + 1;
  const int half = (mx + 1) / 2;
  const int total = half * my;
  for (int sw = 0; sw < nsweeps; ++sw) {
    for (int pass = 0; pass < 2; ++pass) {
      const int par = (g.first_par + pass) & 1;
      for (int p = threadIdx.x; p < total; p += blockDim.x) {
        const int t = p % half, j = g.lb[1] + p / half;
        const int i0 = g.lb[0] + ((((g.lb[0] + j) & 1) != par) ? 1 : 0);
        const int i = i0 + 2 * t;
        if (i > g.ub[0]) continue;
        const int xl = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i - 1);
        const int xh = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i + 1);
        const int yl = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j - 1);
        const int yh = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j + 1);
        double un = 0.0;  // ndsm_poisson.f90:603-617
        un = un + u[xl + nx * j] * g.w[0] + u[xh + nx * j] * g.w[0];
        un = un + u[i + nx * yl] * g.w[1] + u[i + nx * yh] * g.w[1];
        u[i + nx * j] = (un - rhs[i + nx * j]) * g.w1;
      }
      __syncthreads();
    }
    if (g.all_neumann) {
      double sm = 0.0;
      for (int p = threadIdx.x; p < n; p += blockDim.x) sm = sm + u[p];
      for (int o = 32; o > 0; o >>= 1) sm = sm + __shfl_down(sm, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
      __syncthreads();
      double tot = 0.0;
      for (int q = 0; q < 16; ++q) tot = tot + red[q];
      const double mean = tot / (double)n;
      for (int p = threadIdx.x; p < n; p += blockDim.x) u[p] = u[p] - mean;
      __syncthreads();
    }
  }
  for (int p = threadIdx.x; p < n; p += blockDim.x) u_g[p] = u[p];
}

// Mid-size 2-D level (4096 < points <= kMed2D, e.g. the 128^2 level of a 2-D face solve): the same
// single-workgroup scheme with only u in LDS (up to 152 KB, dynamic) and the right-hand side read
// from global memory (L2-resident: one 8-byte read per update).  A level of 16384 points otherwise
// costs ~5 launches per sweep (two colours, two-stage mean, subtraction), i.e. ~50 dispatch
// latencies per level visit; this is one.
constexpr int kMed2D = 19456;
__global__ __launch_bounds__(1024) void rbgs2_medium(double *__restrict__ u_g, const double *__restrict__ rhs_g,
                                                     ndsmk_grid g, int nsweeps) {
  extern __shared__ __attribute__((aligned(16))) double u[];  // the whole level
  __shared__ double red[16];
  constexpr int KQ = 8;   // points per thread and colour held in registers (128^2: exactly these; more: the loop below)
  const int nx = g.n[0], ny = g.n[1];
  const int n = nx * ny;
  for (int p = threadIdx.x; p < n; p += blockDim.x) u[p] = u_g[p];
  const int mx = g.ub[0] - g.lb[0] + 1, my = g.ub[1] - g.lb[1] + 1;
  const int half = (mx + 1) / 2;
  const int total = half * my;
  // one point of the colour pass `par`: ndsm_poisson.f90:603-617, mirror faces folded into the neighbour index
  auto update = [&](int i, int j, double r) {
    const int xl = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i - 1);
    const int xh = (i == 0) ? 1 : (i == nx - 1 ? nx - 2 : i + 1);
    const int yl = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j - 1);
    const int yh = (j == 0) ? 1 : (j == ny - 1 ? ny - 2 : j + 1);
    double un = 0.0;
    un = un + u[xl + nx * j] * g.w[0] + u[xh + nx * j] * g.w[0];
    un = un + u[i + nx * yl] * g.w[1] + u[i + nx * yh] * g.w[1];
    u[i + nx * j] = (un - r) * g.w1;
  };
  auto point_of = [&](int p, int par, int &i, int &j) {
    const int t = p % half;
    j = g.lb[1] + p / half;
    i = g.lb[0] + ((((g.lb[0] + j) & 1) != par) ? 1 : 0) + 2 * t;
    return p < total && i <= g.ub[0];
  };
  // A thread's first KQ points of either colour are fixed for the whole call: their coordinates (-1: none) and
  // right-hand side sit in registers - one integer division per point HERE instead of one per point and colour
  // pass, and no global load inside the sweeps (each was a dependent ~1 us L2 round trip in front of a pass that
  // has ~0.5 us of work).  Same expression per point: same bits.
  int pij[2][KQ];
  double rr[2][KQ];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      int i, j;
      const bool have = point_of((int)threadIdx.x + 1024 * q, (g.first_par + pass) & 1, i, j);
      pij[pass][q] = have ? (i | (j << 16)) : -1;
      rr[pass][q] = (have && rhs_g) ? rhs_g[i + nx * j] : 0.0;
    }
  }
  __syncthreads();
  for (int sw = 0; sw < nsweeps; ++sw) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int c = pij[pass][q];
        if (c >= 0) update(c & 0xffff, c >> 16, rr[pass][q]);
        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four points' neighbour reads in flight at a time: registers)
      }
      for (int p = (int)threadIdx.x + 1024 * KQ; p < total; p += 1024) {   // (levels of more than 16384 points)
        int i, j;
        if (point_of(p, (g.first_par + pass) & 1, i, j)) update(i, j, rhs_g ? rhs_g[i + nx * j] : 0.0);
      }
      __syncthreads();
    }
    if (g.all_neumann) {
      double sm = 0.0;
#pragma unroll 8
      for (int p = threadIdx.x; p < n; p += 1024) sm = sm + u[p];   // (unrolled: the reads overlap, the additions keep their order)
      for (int o = 32; o > 0; o >>= 1) sm = sm + __shfl_down(sm, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
      __syncthreads();
      double tot = 0.0;
      for (int q = 0; q < 16; ++q) tot = tot + red[q];
      const double mean = tot / (double)n;
#pragma unroll 8
      for (int p = threadIdx.x; p < n; p += 1024) u[p] = u[p] - mean;
      __syncthreads();
    }
  }
  for (int p = threadIdx.x; p < n; p += blockDim.x) u_g[p] = u[p];
}

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

// The sweeps of one relax call.  bufs[0] holds u on entry; the out-of-place fused passes write to
// whichever of bufs[0..2] is neither the current one nor `keep` (bufs[2] may be null: plain
// ping-pong); *where tells which buffer holds the result.  keep: a buffer that must survive (the
// iterate the V-cycle started from, which the convergence metric is taken against); prev: evaluate
// that metric in the launch of the last sweep (*met_done).  rout: residual on the last sweep.
// px / uc: u += P uc comes first (coarse_to_fine); folded into the loads of the first launch
// where that launch is the two-sweep Laplace one, else done by the stand-alone kernel.
static int relax_impl(const ndsmk_grid *gp, double *const bufs[3], const double *keep, const double *rhs, int nsweeps,
                      int variant, int *where, double *rout, int *res_done, const double *prev, int *met_done,
                      const ndsmk_xfer *px = nullptr, const double *uc = nullptr) {
  NDSM_REQUIRE_READY();
  if (res_done) *res_done = 0;
  if (met_done) *met_done = 0;
  const ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 2 || g.ndim == 3);
  NDSM_CHECK_ARG(g.n[0] >= 2 && g.n[1] >= 2 && (g.ndim == 2 ? g.n[2] == 1 : g.n[2] >= 2));
  for (int d = 0; d < g.ndim; ++d) NDSM_CHECK_ARG(g.lb[d] >= 0 && g.ub[d] <= g.n[d] - 1);
  NDSM_CHECK_ARG(variant >= 0 && variant <= 2);
  NDSM_CHECK_ARG(g.ndim == 2 || (g.zown0 >= 0 && g.zown1 <= g.n[2] && g.zown0 < g.zown1));
  const int64_t npts = (int64_t)g.n[0] * g.n[1] * g.n[2];
  hipStream_t s = ndsm::stream();
  const int mx = g.ub[0] - g.lb[0] + 1, my = g.ub[1] - g.lb[1] + 1, mz = g.ub[2] - g.lb[2] + 1;
  int cur = 0;
  if (where) *where = 0;
  bool prol_pending = px != nullptr;
  if (prol_pending && (nsweeps <= 0 || mx <= 0 || my <= 0 || (g.ndim == 3 && mz <= 0))) {
    if (int rc = ndsmk_prolong_add(px, uc, bufs[0])) return rc;
    prol_pending = false;
  }
  if (mx <= 0 || my <= 0 || (g.ndim == 3 && mz <= 0)) return 0;  // nothing to update
  // an in-place kernel may not touch the buffer the caller wants kept
  auto in_place_ok = [&]() { return keep == nullptr || bufs[cur] != keep; };
  // small 3-D level: every sweep in one single-workgroup launch
  if (g.ndim == 3 && variant == 0 && !g.all_neumann && npts <= 4096 && g.k0 == 0 && g.zown0 == 0 &&
      g.zown1 == g.n[2] && nsweeps > 0) {
    NDSM_CHECK_ARG(in_place_ok());
    if (prol_pending) {
      if (int rc = ndsmk_prolong_add(px, uc, bufs[0])) return rc;
    }
    hipLaunchKernelGGL(rbgs3_small, dim3(1), dim3(1024), 0, s, bufs[0], rhs, g, nsweeps);
    NDSM_LAUNCH_CHECK();
    return 0;
  }
  // small 2-D level: likewise (incl. the mean shift of all-Neumann problems)
  if (g.ndim == 2 && variant == 0 && npts <= 4096 && nsweeps > 0) {
    NDSM_CHECK_ARG(in_place_ok());
    if (prol_pending) {
      if (int rc = ndsmk_prolong_add(px, uc, bufs[0])) return rc;
    }
    hipLaunchKernelGGL(rbgs2_small, dim3(1), dim3(1024), 0, s, bufs[0], rhs, g, nsweeps);
    NDSM_LAUNCH_CHECK();
    return 0;
  }
  // mid-size 2-D level: the same single-workgroup scheme, u in (dynamic) LDS only
  if (g.ndim == 2 && variant == 0 && npts <= kMed2D && nsweeps > 0) {
    NDSM_CHECK_ARG(in_place_ok());
    if (prol_pending) {
      if (int rc = ndsmk_prolong_add(px, uc, bufs[0])) return rc;
    }
    static int attr_epoch = 0;
    if (ndsm::first_in_epoch(attr_epoch))
      NDSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rbgs2_medium), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   kMed2D * (int)sizeof(double)));
    hipLaunchKernelGGL(rbgs2_medium, dim3(1), dim3(1024), (size_t)npts * sizeof(double), s, bufs[0], rhs, g, nsweeps);
    NDSM_LAUNCH_CHECK();
    return 0;
  }
  for (int sw = 0; sw < nsweeps; ++sw) {
    double *u = bufs[cur];
    if (g.ndim == 3) {
      bool done = false;
      if (variant != 1) {
        // destination of an out-of-place pass: not the current buffer, not the kept one
        int dst = -1;
        for (int c = 0; c < 3; ++c)
          if (c != cur && bufs[c] && bufs[c] != keep) {
            dst = c;
            break;
          }
        // all-Neumann levels shift the mean after EVERY sweep: one sweep per pass there
        int ndone = 0;
        if (prol_pending) {  // first pass: try the launch that interpolates while it loads
          int rc = ndsm::launch_rbgs3_fused(g, u, dst >= 0 ? bufs[dst] : nullptr, rhs, g.all_neumann ? 1 : nsweeps - sw,
                                            variant == 2, &ndone, nullptr, nullptr, nullptr, nullptr, px, uc);
          if (rc) return rc;
          if (ndone == 0) {  // not that launch: interpolate in place, then sweep as usual
            NDSM_CHECK_ARG(in_place_ok());
            rc = ndsmk_prolong_add(px, uc, u);
            if (rc) return rc;
          }
          prol_pending = false;
        }
        // the residual rides on the last sweep (not on all-Neumann levels: the mean shift comes in between)
        int rc = ndone > 0 ? 0
                           : ndsm::launch_rbgs3_fused(g, u, dst >= 0 ? bufs[dst] : nullptr, rhs,
                                                      g.all_neumann ? 1 : nsweeps - sw, variant == 2, &ndone,
                                                      g.all_neumann ? nullptr : rout, res_done,
                                                      g.all_neumann ? nullptr : prev, met_done, nullptr, nullptr);
        if (rc) return rc;
        done = ndone > 0;
        if (done) {  // the sweeps landed in the other array
          sw += ndone - 1;
          cur = dst;
          u = bufs[cur];
        }
        if (!done && variant == 2)
          return ndsm::fail(NDSMK_EARG, "fused smoother does not support this level shape", __FILE__, __LINE__);
      }
      if (!done && prol_pending) {
        NDSM_CHECK_ARG(in_place_ok());
        if (int rc = ndsmk_prolong_add(px, uc, u)) return rc;
        prol_pending = false;
      }
      if (!done) {
        if (g.zown0 != 0 || g.zown1 != g.n[2])
          return ndsm::fail(NDSMK_EARG, "z-slab levels need the fused smoother (nx even, >= 16 x 16 x 8 owned)", __FILE__,
                            __LINE__);
        NDSM_CHECK_ARG(in_place_ok());
        const int half = (mx + 1) / 2;
        dim3 block(64, 4, 1);
        dim3 grid((half + 63) / 64, (my + 3) / 4, mz);
        for (int pass = 0; pass < 2; ++pass) {
          hipLaunchKernelGGL(rbgs3_color, grid, block, 0, s, u, rhs, g, (g.first_par + pass) & 1);
          NDSM_LAUNCH_CHECK();
        }
      }
    } else {
      NDSM_CHECK_ARG(in_place_ok());
      if (prol_pending) {
        if (int rc = ndsmk_prolong_add(px, uc, u)) return rc;
        prol_pending = false;
      }
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
  if (cur != 0) {
    if (where) {
      *where = cur;
    } else {  // caller cannot swap: bring the result home
      NDSM_HIP(hipMemcpyAsync(bufs[0], bufs[cur], sizeof(double) * (size_t)npts, hipMemcpyDeviceToDevice, s));
    }
  }
  return 0;
}

extern "C" int ndsmk_relax(const ndsmk_grid *gp, double *u, double *ualt, const double *rhs, int nsweeps,
                           int variant, int *result_in_alt) {
  double *const bufs[3] = {u, ualt, nullptr};
  return relax_impl(gp, bufs, nullptr, rhs, nsweeps, variant, result_in_alt, nullptr, nullptr, nullptr, nullptr);
}

// nsweeps sweeps followed by r = rhs - L u: where the fused kernel covers the level
// the residual is produced by the launch of the last sweep, otherwise by residual.hip.
// variant 2 (tests): the fused sweep+residual launch or an error.
extern "C" int ndsmk_relax_residual(const ndsmk_grid *gp, double *u, double *ualt, const double *rhs, double *r,
                                    int nsweeps, int variant, int *result_in_alt) {
  NDSM_CHECK_ARG(r != nullptr && result_in_alt != nullptr && (variant == 0 || variant == 2));
  int res_done = 0;
  double *const bufs[3] = {u, ualt, nullptr};
  int rc = relax_impl(gp, bufs, nullptr, rhs, nsweeps, variant, result_in_alt, r, &res_done, nullptr, nullptr);
  if (rc) return rc;
  if (!res_done) {
    if (variant == 2)
      return ndsm::fail(NDSMK_EARG, "fused sweep + residual does not cover this level", __FILE__, __LINE__);
    rc = ndsmk_residual(gp, *result_in_alt ? ualt : u, rhs, r);
  }
  return rc;
}

// The V-cycle driver's level-1 form of the two calls above: three buffers (u on entry, two
// spares), `keep` is never written (the iterate the cycle started from), *where = 0/1/2 says
// which buffer holds the result.  r != NULL: residual of the result (fused into the last sweep's
// launch where possible).  prev != NULL: the launch of the last sweep also evaluates
// max / sum |u_new - prev| (*met_done = 1; read it with ndsmk_fetch_fused_metric), if it can.
// px != NULL: u += P uc (the coarse-grid correction) comes before the sweeps.
extern "C" int ndsmk_relax3(const ndsmk_grid *gp, double *u, double *a, double *b, const double *keep,
                            const double *rhs, int nsweeps, double *r, const double *prev, int *where,
                            int *met_done, const ndsmk_xfer *px, const double *uc) {
  NDSM_CHECK_ARG(where != nullptr && met_done != nullptr && u && a);
  int res_done = 0;
  double *const bufs[3] = {u, a, b};
  int rc = relax_impl(gp, bufs, keep, rhs, nsweeps, 0, where, r, &res_done, prev, met_done, px, uc);
  if (rc) return rc;
  if (r && !res_done) rc = ndsmk_residual(gp, bufs[*where], rhs, r);
  return rc;
}

extern "C" int ndsmk_fetch_fused_metric(double *h_out2) {
  NDSM_REQUIRE_READY();
  return ndsm::fetch_fused_metric(h_out2);
}

// One out-of-place fused pass (nsweeps = 1 or 2) over the owned planes [z0, z1) of a z-slab only:
// the pieces of a pass whose halo exchange overlaps its interior (ndsmh_world).  u is read
// (z0 - 2 nsweeps .. z1 + 2 nsweeps must be valid planes), uout written on [z0, z1).
// px != NULL: u + P uc is formed while the planes are loaded (the launch that starts the
// post-smoothing; uc holds coarse planes [px->c_k0, px->c_k0 + px->c_cnt), which must cover the
// brackets of the window's planes AND of the ghost planes the launch reads).  Returns NDSMK_EARG
// if that launch does not exist for this level (general rhs, one sweep): the caller then
// interpolates with the stand-alone kernel first.
// prev != NULL: the launch also evaluates max / sum of |u_new - prev| over the planes it stores
// (update_u's metric; read back with ndsmk_fetch_fused_metric).  accumulate != 0: added to what the
// previous such launch left on the device - the windows of one pass, the slabs of one process.
extern "C" int ndsmk_fused_window(const ndsmk_grid *gp, const double *u, double *uout, const double *rhs, int nsweeps,
                                  int z0, int z1, const ndsmk_xfer *px, const double *uc, const double *prev,
                                  int accumulate) {
  NDSM_REQUIRE_READY();
  ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 3 && (nsweeps == 1 || nsweeps == 2) && z0 >= g.zown0 && z1 <= g.zown1 && z0 < z1);
  NDSM_CHECK_ARG((!px || uc) && !(px && prev));
  g.zown0 = z0;
  g.zown1 = z1;
  int ndone = 0, met = 0;
  ndsm::fused_metric_accumulate(accumulate != 0);
  int rc = ndsm::launch_rbgs3_fused(g, u, uout, rhs, nsweeps, true, &ndone, nullptr, nullptr, prev, prev ? &met : nullptr,
                                    px, uc);
  ndsm::fused_metric_accumulate(false);
  if (rc) return rc;
  if (ndone != nsweeps || (prev && !met))
    return ndsm::fail(NDSMK_EARG, "fused smoother: this window / sweep count is not covered", __FILE__, __LINE__);
  return 0;
}

// The sweep + residual pass on a window of owned planes [z0, z1): ONE sweep u -> uout and r = rhs - L uout on the
// window (u must be valid on z0 - 3 .. z1 + 2).  For the pieces of the last pre-smoothing pass whose halo
// exchange overlaps its interior (ndsmh_world).
extern "C" int ndsmk_fused_window_res(const ndsmk_grid *gp, const double *u, double *uout, const double *rhs, double *rout,
                                      int z0, int z1) {
  NDSM_REQUIRE_READY();
  ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 3 && rout && z0 >= g.zown0 && z1 <= g.zown1 && z0 < z1);
  g.zown0 = z0;
  g.zown1 = z1;
  int ndone = 0, res = 0;
  int rc = ndsm::launch_rbgs3_fused(g, u, uout, rhs, 1, true, &ndone, rout, &res, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (ndone != 1 || !res)
    return ndsm::fail(NDSMK_EARG, "fused sweep + residual: this window is not covered", __FILE__, __LINE__);
  return 0;
}

// can ndsmk_fused_window(..., prev) evaluate the metric? (fp64 level off the all-Neumann path)
extern "C" int ndsmk_fused_metric_ok(const ndsmk_grid *gp) { return (gp->ndim == 3 && !gp->all_neumann) ? 1 : 0; }

// can ndsmk_fused_window(..., px, uc) interpolate while it loads? (one sweep, or two without a right-hand side; fp64)
extern "C" int ndsmk_fused_prolong_ok(const ndsmk_grid *gp, const double *rhs, int nsweeps) {
  return (gp->ndim == 3 && (!rhs || nsweeps == 1) && (nsweeps == 1 || nsweeps == 2) && !gp->all_neumann) ? 1 : 0;
}
