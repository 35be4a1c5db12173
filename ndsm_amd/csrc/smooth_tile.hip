// Red-black Gauss-Seidel for the MID-SIZE 3-D levels of a hierarchy (a few thousand to a few million
// points: 32^3 ... 128^3): S full sweeps per launch on LDS-resident 3-D tiles, out of place.
//
// Why a third smoother.  These levels are latency bound, not bandwidth bound: a level of 64^3 is 2 MB.
// The colour-pass kernel (smooth.hip) needs two launches per sweep and a launch that touches the whole
// level cannot finish in less than ~5 us (dispatch + one round trip through L2 / Infinity Cache after
// the previous kernel's write-back), so the 15 sweeps a V-cycle spends on such a level cost 30 x 5 us
// whatever the level's size.  The z-streaming kernel (smooth_fused.hip) is built for planes far wider
// than its 128 x 22 tile and pays ~30 us of pipeline fill per launch here.  This kernel trades
// redundant arithmetic - which is free while 250 of the 256 CUs idle - for launches:
//
//   * a workgroup loads a box of LX x LY x LZ points into LDS: its OWNED box plus a ring of R = 2 S
//     (+1 with the residual stage) points on every side that is not a physical face;
//   * it runs the 2 S colour stages of S sweeps on the box in LDS, one barrier per stage.  Stage t
//     may update a point only if it lies at least t + 1 points inside every open side of the box:
//     such a point's neighbours lie at least t points inside, i.e. they carry the result of stage
//     t - 1 - the ring goes stale from the outside in, one point per stage, and never reaches the
//     owned box.  Physical faces impose nothing (the mirror neighbour is inside);
//   * the owned box is written to the OTHER array (workgroups read each other's ring from the input);
//   * RES: one more ring, and the residual r = rhs - L u of the result is evaluated from LDS on the
//     owned box - the V-cycle's "sweeps, then residual" is one launch less.
//
// Five sweeps are 2 + 2 + 1(+residual): 3 launches instead of 10 (+1).  Same update expression and
// operand order as rbgs3_color (ndsm_optimized.f90:103-167) and residual3 (ndsm_optimized.f90:346-447):
// bit-identical, tested against both and against the oracle.
#include "common.hpp"

#include <cstdlib>

namespace {

struct TilePlan {
  int ntx, nty, ntz;
};

template <int S, bool RES, int LX, int LY, int LZ>
__global__ __launch_bounds__(1024) void rbgs3_tile_k(const double *__restrict__ u, double *__restrict__ uout,
                                                     const double *__restrict__ rhs, double *__restrict__ rout,
                                                     ndsmk_grid g, TilePlan pl) {
  constexpr int NT = 1024;
  constexpr int NST = 2 * S;
  constexpr int R = NST + (RES ? 1 : 0);
  constexpr int OX = LX - 2 * R, OY = LY - 2 * R, OZ = LZ - 2 * R;
  static_assert(OX > 0 && OY > 0 && OZ > 0 && (LX % 2) == 0, "tile");
  constexpr int HX = LX / 2;
  constexpr int NBOX = LX * LY * LZ, NHALF = HX * LY * LZ;
  extern __shared__ __attribute__((aligned(16))) double box[];

  const int w = (int)blockIdx.x;
  const int tx = w % pl.ntx, ty = (w / pl.ntx) % pl.nty, tz = w / (pl.ntx * pl.nty);
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int x0 = tx * OX - R, y0 = ty * OY - R, z0 = tz * OZ - R;   // global index of box point (0,0,0)
  const bool oxl = x0 > 0, oxh = x0 + LX < nx, oyl = y0 > 0, oyh = y0 + LY < ny, ozl = z0 > 0, ozh = z0 + LZ < nz;
  const int tid = (int)threadIdx.x;
  const size_t sy = (size_t)nx, sz = (size_t)nx * (size_t)ny;

  // ---- load the box (zero outside the domain) ----
  for (int p = tid; p < NBOX; p += NT) {
    const int li = p % LX, lj = (p / LX) % LY, lk = p / (LX * LY);
    const int i = x0 + li, j = y0 + lj, k = z0 + lk;
    const bool in = i >= 0 && i < nx && j >= 0 && j < ny && k >= 0 && k < nz;
    const size_t c = in ? (size_t)i + sy * (size_t)j + sz * (size_t)k : 0;
    const double v = u[c];           // unconditional (clamped): keeps the loads of a thread in flight together
    box[p] = in ? v : 0.0;
  }
  __syncthreads();

  const double w0 = g.w[0], w1 = g.w[1], w2 = g.w[2], w1i = g.w1;
  const int par0 = (x0 + y0 + z0) & 1;   // (i + j + k) & 1 = (li + lj + lk + par0) & 1 ; x0.. may be negative: & 1 is still the parity
#pragma unroll
  for (int t = 0; t < NST; ++t) {
    const int par = (g.first_par + t) & 1;   // colour of this stage: (i + j + k) & 1 == par
    const int lo_x = oxl ? t + 1 : 0, hi_x = oxh ? LX - 2 - t : LX - 1;
    const int lo_y = oyl ? t + 1 : 0, hi_y = oyh ? LY - 2 - t : LY - 1;
    const int lo_z = ozl ? t + 1 : 0, hi_z = ozh ? LZ - 2 - t : LZ - 1;
    for (int p = tid; p < NHALF; p += NT) {
      const int h = p % HX, lj = (p / HX) % LY, lk = p / (HX * LY);
      const int li = 2 * h + ((lj + lk + par0 + par) & 1);
      const int i = x0 + li, j = y0 + lj, k = z0 + lk;
      const bool upd = li >= lo_x && li <= hi_x && lj >= lo_y && lj <= hi_y && lk >= lo_z && lk <= hi_z &&
                       i >= g.lb[0] && i <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1] && k >= g.lb[2] && k <= g.ub[2];
      if (!upd) continue;
      const int c = li + LX * (lj + LY * lk);
      // mirrored neighbours at the physical faces (ndsm_optimized.f90:113-120)
      const int cxl = i == 0 ? c + 1 : c - 1, cxh = i == nx - 1 ? c - 1 : c + 1;
      const int cyl = j == 0 ? c + LX : c - LX, cyh = j == ny - 1 ? c - LX : c + LX;
      const int czl = k == 0 ? c + LX * LY : c - LX * LY, czh = k == nz - 1 ? c - LX * LY : c + LX * LY;
      const double rr = rhs ? rhs[(size_t)i + sy * (size_t)j + sz * (size_t)k] : 0.0;
      const double unew = (box[cxh] + box[cxl]) * w0 + (box[cyh] + box[cyl]) * w1 + (box[czh] + box[czl]) * w2 - rr;
      box[c] = w1i * unew;
    }
    __syncthreads();
  }

  // ---- store the owned box; RES: its residual too ----
  const int ox0 = tx * OX, ox1 = min(ox0 + OX, nx), oy0 = ty * OY, oy1 = min(oy0 + OY, ny), oz0 = tz * OZ,
            oz1 = min(oz0 + OZ, nz);
  const int mx = ox1 - ox0, my = oy1 - oy0, mz = oz1 - oz0;
  const int nown = mx * my * mz;
  for (int p = tid; p < nown; p += NT) {
    const int a = p % mx, b = (p / mx) % my, d = p / (mx * my);
    const int i = ox0 + a, j = oy0 + b, k = oz0 + d;
    const int c = (i - x0) + LX * ((j - y0) + LY * (k - z0));
    const size_t gc = (size_t)i + sy * (size_t)j + sz * (size_t)k;
    const double uc = box[c];
    uout[gc] = uc;
    if (RES) {
      const bool inside = i >= g.lb[0] && i <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1] && k >= g.lb[2] && k <= g.ub[2];
      double res = 0.0;
      if (inside) {
        const double ul = box[i == 0 ? c + 1 : c - 1], uh = box[i == nx - 1 ? c - 1 : c + 1];
        const double vl = box[j == 0 ? c + LX : c - LX], vh = box[j == ny - 1 ? c - LX : c + LX];
        const double wl = box[k == 0 ? c + LX * LY : c - LX * LY], wh = box[k == nz - 1 ? c - LX * LY : c + LX * LY];
        const double v = (ul + uh) * w0 + (vl + vh) * w1 + (wl + wh) * w2 - (rhs ? rhs[gc] : 0.0) - uc * g.wc;
        res = -v;
      }
      rout[gc] = res;
    }
  }
}

template <int S, bool RES, int LX, int LY, int LZ>
int launch_tile(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, double *rout) {
  constexpr int R = 2 * S + (RES ? 1 : 0);
  constexpr int OX = LX - 2 * R, OY = LY - 2 * R, OZ = LZ - 2 * R;
  TilePlan pl;
  pl.ntx = (g.n[0] + OX - 1) / OX;
  pl.nty = (g.n[1] + OY - 1) / OY;
  pl.ntz = (g.n[2] + OZ - 1) / OZ;
  constexpr size_t lds_bytes = sizeof(double) * LX * LY * LZ;
  auto kfn = rbgs3_tile_k<S, RES, LX, LY, LZ>;
  static int attr_epoch = 0;
  if (ndsm::first_in_epoch(attr_epoch))
    NDSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_bytes));
  hipLaunchKernelGGL(kfn, dim3(pl.ntx * pl.nty * pl.ntz), dim3(1024), lds_bytes, ndsm::stream(), u, uout, rhs, rout, g, pl);
  NDSM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

namespace ndsm {

// levels this kernel serves: more points than the single-workgroup kernel takes, at most NDSM_TILE_MAX
// (environment, or ndsmk_debug_tile_max at run time)
static long long g_tile_max = -1;
bool tile_smoother_applies(const ndsmk_grid &g) {
  long long &tmax = g_tile_max;
  if (tmax < 0) {
    const char *e = std::getenv("NDSM_TILE_MAX");
    tmax = e ? std::atoll(e) : 0;   // OFF by default: measured slower than the colour passes / the streaming kernel so far
  }
  const long long npts = (long long)g.n[0] * g.n[1] * g.n[2];
  return g.ndim == 3 && !g.all_neumann && g.k0 == 0 && g.zown0 == 0 && g.zown1 == g.n[2] && g.nzg == g.n[2] &&
         npts > 4096 && npts <= tmax && g.n[0] >= 4 && g.n[1] >= 4 && g.n[2] >= 4;
}

// up to max_sweeps (1 or 2 are performed; *done says how many) sweeps u -> uout; rout != nullptr and the
// launch performs the LAST of the caller's sweeps (max_sweeps <= 2 ... see below): residual of the result
// too (*res_done = 1).  Rule: two sweeps per launch while more than two remain or no residual is wanted;
// the final launch carries the residual with as many sweeps as are left (1 or 2).
int launch_rbgs3_tile(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, int max_sweeps,
                      int *done, double *rout, int *res_done) {
  *done = 0;
  if (res_done) *res_done = 0;
  if (max_sweeps <= 0) return 0;
  const bool last = max_sweeps <= 2;
  const bool res = rout && res_done && last;
  int rc;
  if (max_sweeps >= 2) {
    rc = res ? launch_tile<2, true, 40, 24, 16>(g, u, uout, rhs, rout) : launch_tile<2, false, 40, 24, 16>(g, u, uout, rhs, nullptr);
    *done = 2;
  } else {
    rc = res ? launch_tile<1, true, 40, 24, 16>(g, u, uout, rhs, rout) : launch_tile<1, false, 40, 24, 16>(g, u, uout, rhs, nullptr);
    *done = 1;
  }
  if (rc) return rc;
  if (res) *res_done = 1;
  return 0;
}

}  // namespace ndsm

// tests / tuning: levels of up to `max_points` points take the tile smoother (0: none)
extern "C" int ndsmk_debug_tile_max(long long max_points) {
  ndsm::g_tile_max = max_points < 0 ? 0 : max_points;
  return 0;
}
