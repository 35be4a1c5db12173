// Red-black Gauss-Seidel for the MID-SIZE 3-D levels of a hierarchy (a few thousand to a few million
// points: 32^3 ... 128^3): S full sweeps per launch on LDS-resident 3-D tiles, out of place.
//
// Why a third smoother.  These levels are latency bound, not bandwidth bound: a level of 64^3 is 2 MB.
// The colour-pass kernel (smooth.hip) needs two launches per sweep and a launch that touches the whole
// level cannot finish in less than ~3.5-5 us (dispatch + one round trip through L2 / Infinity Cache
// after the previous kernel's write-back), so the 15 sweeps a V-cycle spends on such a level cost 30
// launches whatever the level's size.  The z-streaming kernel (smooth_fused.hip) walks its planes one
// after the other (1.3-2.5 us per plane-step) and pays 8 warm-up planes per chunk.  This kernel trades
// redundant arithmetic - which is free while most of the 256 CUs idle - for launches and has no serial
// dimension at all:
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
// Layout and scheduling: a thread owns NQ x-PAIRS of the box (both colours of a pair - one is updated per
// stage); everything that does not depend on the stage - LDS index, stage budgets, mirror flags, the
// right-hand side of both elements - is set up once in registers (the first version read the right-hand
// side from global memory inside the stage loop: a memory round trip per update, 22 us per launch).  The
// box is stored element-planar (all even-x elements, then all odd-x ones), so the lanes of a wave touch
// consecutive words whatever the colour (interleaved, every access is a 2-way bank conflict).
//
// MEASURED (round 2, 512^3 hierarchy): a 1024-thread workgroup needs ~17 us for its box whatever the level
// (LDS bound: 56 LDS operations per thread and stage), so five sweeps cost 64 / 61 / 135 us at 64^3 / 32^3 /
// 128^3 against 34 / 29 us for ten back-to-back colour launches (~3 us each) and 77 us for the streamed
// kernel: OFF by default (NDSM_TILE_MAX / ndsmk_debug_tile_max switch it on).
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

// owned extent in x: the box origin is rounded down to an even column (a pair never straddles the origin),
// which costs one column of ring on the far side when R is odd - two, to keep the owned extent even
constexpr int tile_ox(int LX, int R) { return LX - 2 * R - ((R & 1) ? 2 : 0); }

template <int S, bool RES, bool RHS0, int LX, int LY, int LZ>
__global__ __launch_bounds__(1024) void rbgs3_tile_k(const double *__restrict__ u, double *__restrict__ uout,
                                                     const double *__restrict__ rhs, double *__restrict__ rout,
                                                     ndsmk_grid g, TilePlan pl) {
  constexpr int NT = 1024;
  constexpr int NST = 2 * S;
  constexpr int R = NST + (RES ? 1 : 0);
  constexpr int OX = tile_ox(LX, R), OY = LY - 2 * R, OZ = LZ - 2 * R;
  static_assert(OX > 0 && OY > 0 && OZ > 0 && (LX % 2) == 0 && (OX % 2) == 0, "tile");
  constexpr int HX = LX / 2;
  constexpr int NPAIR = HX * LY * LZ;      // pairs in the box = words per element plane
  constexpr int NQ = (NPAIR + NT - 1) / NT;
  constexpr int SY = HX, SZ = HX * LY;     // strides inside an element plane
  static_assert(NPAIR % 32 == 0, "element planes must start on the same bank");
  static_assert(NST + 1 <= 15, "stage budget field is 4 bits");
  extern __shared__ __attribute__((aligned(16))) double box[];   // [2][NPAIR]: even-x elements, odd-x elements

  const int w = (int)blockIdx.x;
  const int tx = w % pl.ntx, ty = (w / pl.ntx) % pl.nty, tz = w / (pl.ntx * pl.nty);
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int x0 = (tx * OX - R) & ~1, y0 = ty * OY - R, z0 = tz * OZ - R;   // global index of box point (0,0,0); x0 even
  const bool oxl = x0 > 0, oxh = x0 + LX < nx, oyl = y0 > 0, oyh = y0 + LY < ny, ozl = z0 > 0, ozh = z0 + LZ < nz;
  const int ox0 = tx * OX, ox1 = min(ox0 + OX, nx), oy0 = ty * OY, oy1 = min(oy0 + OY, ny), oz0 = tz * OZ,
            oz1 = min(oz0 + OZ, nz);
  const int tid = (int)threadIdx.x;
  const int fp = g.first_par & 1;

  // ---- per-pair constants --------------------------------------------------------------------
  // fl: bit 0 pair exists, 1/2 element 0/1 inside the domain, 3 stage 0 updates element 1 (else element 0),
  //     4 element 0 at i == 0 (x-low mirror), 5 element 1 at i+1 == nx-1 (x-high mirror), 6/7 j == 0 / ny-1,
  //     8/9 k == 0 / nz-1, 10/11 element 0/1 owned, 12/13 element 0/1 inside the residual bounds,
  //     14 element 0 at i == nx-1 (odd nx: its partner is outside, both x neighbours are column nx-2),
  //     16-19 / 20-23 stage budget of element 0 / 1 (stage t may update it iff budget > t)
  int pidx[NQ], fl[NQ], goff[NQ];
  double r0[RHS0 ? 1 : NQ], r1[RHS0 ? 1 : NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int p = tid + NT * q;
    const int h = p % HX, lj = (p / HX) % LY, lk = p / (HX * LY);
    const int li = 2 * h;
    const int i = x0 + li, j = y0 + lj, k = z0 + lk;
    const bool have = p < NPAIR;
    const bool rowin = have && j >= 0 && j < ny && k >= 0 && k < nz;
    const bool in0 = rowin && i >= 0 && i < nx, in1 = rowin && i + 1 >= 0 && i + 1 < nx;
    auto inb = [&](int ii) {
      return ii >= g.lb[0] && ii <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1] && k >= g.lb[2] && k <= g.ub[2];
    };
    auto budget = [&](int ii, int lii) {
      // distance to the open sides of the box (large where none is open); 0 outside the update bounds
      int d = 15;
      if (oxl) d = min(d, lii);
      if (oxh) d = min(d, LX - 1 - lii);
      if (oyl) d = min(d, lj);
      if (oyh) d = min(d, LY - 1 - lj);
      if (ozl) d = min(d, lk);
      if (ozh) d = min(d, LZ - 1 - lk);
      return inb(ii) ? d : 0;   // stage t needs distance >= t + 1, i.e. budget > t
    };
    const int b0 = in0 ? budget(i, li) : 0, b1 = in1 ? budget(i + 1, li + 1) : 0;
    const bool ownyz = j >= oy0 && j < oy1 && k >= oz0 && k < oz1;
    const bool own0 = in0 && ownyz && i >= ox0 && i < ox1, own1 = in1 && ownyz && i + 1 >= ox0 && i + 1 < ox1;
    int f = (have ? 1 : 0) | (in0 ? 2 : 0) | (in1 ? 4 : 0);
    f |= (((i + j + k + fp) & 1) ? 8 : 0);   // colour of stage 0 is (ii + j + k) & 1 == fp: element 0 iff this bit is 0
    f |= (i == 0 ? 16 : 0) | (i + 1 == nx - 1 ? 32 : 0) | (j == 0 ? 64 : 0) | (j == ny - 1 ? 128 : 0) |
         (k == 0 ? 256 : 0) | (k == nz - 1 ? 512 : 0) | (i == nx - 1 ? 16384 : 0);
    f |= (own0 ? 1024 : 0) | (own1 ? 2048 : 0) | ((in0 && inb(i)) ? 4096 : 0) | ((in1 && inb(i + 1)) ? 8192 : 0);
    f |= (b0 << 16) | (b1 << 20);
    fl[q] = f;
    pidx[q] = have ? p : 0;
    // clamped global offset of element 0 (loads are unconditional; element 1 is at +1 where it exists)
    const int ic = min(max(i, 0), nx - 1), jc = min(max(j, 0), ny - 1), kc = min(max(k, 0), nz - 1);
    goff[q] = ic + nx * (jc + ny * kc);   // these levels have far fewer than 2^31 points (tile_smoother_applies)
  }
  // ---- load the box, and the right-hand side of every pair (all in flight together) ----
  {
    double v0[NQ], v1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      v0[q] = u[goff[q]];
      v1[q] = u[goff[q] + ((fl[q] & 4) ? 1 : 0)];
    }
    if (!RHS0) {
#pragma unroll
      for (int q = 0; q < (RHS0 ? 1 : NQ); ++q) {
        r0[q] = rhs[goff[q]];
        r1[q] = rhs[goff[q] + ((fl[q] & 4) ? 1 : 0)];
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (fl[q] & 1) {
        box[pidx[q]] = (fl[q] & 2) ? v0[q] : 0.0;
        box[NPAIR + pidx[q]] = (fl[q] & 4) ? v1[q] : 0.0;
      }
    }
  }
  __syncthreads();

  const double w0 = g.w[0], w1 = g.w[1], w2 = g.w[2], w1i = g.w1;
  constexpr int QG = 4;   // pairs handled side by side (their LDS reads are in flight together)
#pragma unroll 1   // (unrolled, the per-stage index arithmetic of all stages is hoisted and spills)
  for (int t = 0; t < NST; ++t) {
    // stage t updates the colour (i + j + k) & 1 == (fp + t) & 1: element e = e0 ^ (t & 1) of every pair
#pragma unroll
    for (int qb = 0; qb < NQ; qb += QG) {
      double xo[QG], xi[QG], yl[QG], yh[QG], zl[QG], zh[QG];
      int cc[QG];
      bool upd[QG];
#pragma unroll
      for (int qq = 0; qq < QG; ++qq) {
        const int q = qb + qq;
        if (q >= NQ) continue;
        const int f = fl[q];
        const int e = ((f >> 3) ^ t) & 1;
        upd[qq] = ((f >> (16 + 4 * e)) & 15) > t;
        const int c = pidx[q] + e * NPAIR, o = pidx[q] + (1 - e) * NPAIR;   // the element, its pair partner
        cc[qq] = c;
        // x neighbours: the pair partner, and the previous pair's element 1 (e = 0) / the next pair's element 0
        // (e = 1) - mirrored at a physical face onto the partner; element 0 in the last column of an odd nx has
        // no partner: both neighbours are column nx-2
        const bool mx = e ? (f & 32) : (f & 16);
        const int xoi = mx ? o : (e ? o + 1 : o - 1);
        const int xii = (!e && (f & 16384)) ? o - 1 : o;
        const int yli = (f & 64) ? c + SY : c - SY, yhi = (f & 128) ? c - SY : c + SY;
        const int zli = (f & 256) ? c + SZ : c - SZ, zhi = (f & 512) ? c - SZ : c + SZ;
        // never read outside the box: a point that is not updated reads itself
        xi[qq] = box[upd[qq] ? xii : c];
        xo[qq] = box[upd[qq] ? xoi : c];
        yl[qq] = box[upd[qq] ? yli : c];
        yh[qq] = box[upd[qq] ? yhi : c];
        zl[qq] = box[upd[qq] ? zli : c];
        zh[qq] = box[upd[qq] ? zhi : c];
      }
#pragma unroll
      for (int qq = 0; qq < QG; ++qq) {
        const int q = qb + qq;
        if (q >= NQ) continue;
        const int e = ((fl[q] >> 3) ^ t) & 1;
        const double rr = RHS0 ? 0.0 : (e ? r1[RHS0 ? 0 : q] : r0[RHS0 ? 0 : q]);
        // (u(xh) + u(xl)): for element 0 xh is the partner, for element 1 xl is
        const double xs = e ? (xo[qq] + xi[qq]) : (xi[qq] + xo[qq]);
        const double unew = xs * w0 + (yh[qq] + yl[qq]) * w1 + (zh[qq] + zl[qq]) * w2 - rr;
        if (upd[qq]) box[cc[qq]] = w1i * unew;
      }
    }
    __syncthreads();
  }

  // ---- store the owned points; RES: their residual too ----
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int f = fl[q];
    if (!(f & (1024 | 2048))) continue;
    const int c0 = pidx[q], c1 = pidx[q] + NPAIR;
    const double a0 = box[c0], a1 = box[c1];
    if (f & 1024) uout[goff[q]] = a0;
    if (f & 2048) uout[goff[q] + 1] = a1;
    if (RES) {
      // both elements' stencils from LDS (ring R = 2 S + 1: the neighbours of owned points are final)
      const int ym = (f & 64) ? SY : -SY, yp = (f & 128) ? -SY : SY, zm = (f & 256) ? SZ : -SZ, zp = (f & 512) ? -SZ : SZ;
      if (f & 1024) {
        double res = 0.0;
        if (f & 4096) {
          const double ul = (f & 16) ? a1 : box[c1 - 1];
          const double uh = (f & 16384) ? ul : a1;
          const double v = (ul + uh) * w0 + (box[c0 + ym] + box[c0 + yp]) * w1 + (box[c0 + zm] + box[c0 + zp]) * w2 -
                           (RHS0 ? 0.0 : r0[RHS0 ? 0 : q]) - a0 * g.wc;
          res = -v;
        }
        rout[goff[q]] = res;
      }
      if (f & 2048) {
        double res = 0.0;
        if (f & 8192) {
          const double ul = a0, uh = (f & 32) ? a0 : box[c0 + 1];
          const double v = (ul + uh) * w0 + (box[c1 + ym] + box[c1 + yp]) * w1 + (box[c1 + zm] + box[c1 + zp]) * w2 -
                           (RHS0 ? 0.0 : r1[RHS0 ? 0 : q]) - a1 * g.wc;
          res = -v;
        }
        rout[goff[q] + 1] = res;
      }
    }
  }
}

template <int S, bool RES, int LX, int LY, int LZ>
int launch_tile(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, double *rout) {
  constexpr int R = 2 * S + (RES ? 1 : 0);
  constexpr int OX = tile_ox(LX, R), OY = LY - 2 * R, OZ = LZ - 2 * R;
  TilePlan pl;
  pl.ntx = (g.n[0] + OX - 1) / OX;
  pl.nty = (g.n[1] + OY - 1) / OY;
  pl.ntz = (g.n[2] + OZ - 1) / OZ;
  constexpr size_t lds_bytes = sizeof(double) * LX * LY * LZ;
  const void *kfn = rhs ? reinterpret_cast<const void *>(rbgs3_tile_k<S, RES, false, LX, LY, LZ>)
                        : reinterpret_cast<const void *>(rbgs3_tile_k<S, RES, true, LX, LY, LZ>);
  static int attr_epoch[2] = {0, 0};
  if (ndsm::first_in_epoch(attr_epoch[rhs ? 0 : 1]))
    NDSM_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  void *args[] = {(void *)&u, (void *)&uout, (void *)&rhs, (void *)&rout, (void *)&g, (void *)&pl};
  NDSM_HIP(hipLaunchKernel(kfn, dim3(pl.ntx * pl.nty * pl.ntz), dim3(1024), args, lds_bytes, ndsm::stream()));
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
    tmax = e ? std::atoll(e) : 0;   // OFF by default until it is measured faster
  }
  const long long npts = (long long)g.n[0] * g.n[1] * g.n[2];
  return g.ndim == 3 && !g.all_neumann && g.k0 == 0 && g.zown0 == 0 && g.zown1 == g.n[2] && g.nzg == g.n[2] &&
         npts > 4096 && npts <= tmax && npts < (1ll << 30) && g.n[0] >= 4 && g.n[1] >= 4 && g.n[2] >= 4;
}

// up to max_sweeps (1 or 2 are performed; *done says how many) sweeps u -> uout.  Two sweeps per launch
// while more than two remain; the launch that performs the caller's last sweeps (max_sweeps <= 2) carries
// the residual when rout is given (*res_done = 1).
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
