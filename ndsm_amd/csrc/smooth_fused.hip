// Fused red+black Gauss-Seidel sweep: ONE launch reads u (and rhs) once and
// writes the swept field once (OUT OF PLACE: workgroups read each other's halo
// from the input array, so the input must stay intact for the whole launch;
// the caller ping-pongs two arrays) - the 24 B/LUP the roofline is priced on - where
// the two colour passes of smooth.hip move ~48 B/LUP with half-used lines.
//
// Same arithmetic as rbgs3_color (ndsm_optimized.f90:103-167): a black point
// only reads red neighbours of the SAME sweep and a red point only reads black
// neighbours of the PREVIOUS state, so any schedule that respects those two
// dependencies gives bit-identical results.  Schedule used here (per workgroup):
//
//   * an (x,y) tile of TXH x TYH points, of which the inner TXI x TYI are
//     owned (written back) and a 2-deep ring is halo: ring 1 is red-updated
//     redundantly so that owned black points see updated red neighbours;
//   * the tile is streamed through a z chunk [zs, ze) as a 2-stage pipeline:
//     iteration k does  stage 0: red   update of plane k
//                       stage 1: black update of plane k-1, then stores it.
//     z neighbours live in the owning thread's registers, the in-plane
//     neighbours in two LDS planes (R_k and R_{k-1});
//   * every thread owns x-PAIRS (16-byte aligned double2): each global load /
//     store is 16 B per lane, rows are contiguous, and every pair holds exactly
//     one red and one black point per plane - no divergence between colours;
//   * plane k+2 (and its rhs) is requested before plane k is computed, so one
//     whole plane of loads per workgroup is always in flight;
//   * work items (tile, chunk) are laid out so that the y-neighbouring tiles,
//     which share halo rows, sit on the same XCD (blockIdx % 8) and hit its L2.
//
// Requirements of this path (otherwise smooth.hip runs): nx even, n >= 8 in
// every dimension, no z-slab ghosts inside the chunk logic other than k0/nzg.
#include "common.hpp"

namespace {

struct FusedPlan {
  int ntx, nty, nzc;  // tiles in x, y and chunks in z
  int zc;             // planes per chunk
  int nwork;          // ntx * nty * nzc
};

// an x-pair; elements are picked with selects (a runtime-indexed register
// array would be demoted to scratch memory)
struct d2 {
  double x, y;
};

__device__ __forceinline__ d2 ld2(const double *p) {
  const double2 t = *reinterpret_cast<const double2 *>(p);
  d2 r;
  r.x = t.x;
  r.y = t.y;
  return r;
}
__device__ __forceinline__ void st2(double *p, const d2 &a) {
  double2 t;
  t.x = a.x;
  t.y = a.y;
  *reinterpret_cast<double2 *>(p) = t;
}
// by value: selects on values, never on addresses
__device__ __forceinline__ double pick(const d2 a, int e) { return e ? a.y : a.x; }
__device__ __forceinline__ d2 put(const d2 a, int e, double v) {
  d2 r;
  r.x = e ? a.x : v;
  r.y = e ? v : a.y;
  return r;
}

template <int TXH, int TYH, int NT, bool RHS0>
__global__ __launch_bounds__(NT) void rbgs3_fused_k(const double *__restrict__ u, double *__restrict__ uout,
                                                    const double *__restrict__ rhs, ndsmk_grid g, FusedPlan pl) {
  constexpr int NPX = TXH / 2;
  constexpr int NPAIR = NPX * TYH;
  constexpr int NS = (NPAIR + NT - 1) / NT;
  constexpr int TXI = TXH - 4, TYI = TYH - 4;
  constexpr int PLANE = TXH * TYH;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  // ---- which (tile, chunk): consecutive y tiles share an XCD ---------
  const int nb8 = gridDim.x >> 3;
  const int w = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
  if (w >= pl.nwork) return;
  const int ty = w % pl.nty;
  const int t2 = w / pl.nty;
  const int tx = t2 % pl.ntx;
  const int cz = t2 / pl.ntx;

  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int x0 = tx * TXI - 2, y0 = ty * TYI - 2;
  const int zs = cz * pl.zc;
  const int ze = min(zs + pl.zc, nz);
  const int ks = max(zs - 2, 0);
  const int ke = min(ze + 1, nz - 1);
  const size_t sz = (size_t)nx * (size_t)ny;

  // ---- per-slot geometry ----------------------------------------------
  int goff[NS];   // offset of the pair inside a plane (i + nx*j), -1 if outside the domain
  int loff[NS];   // offset inside an LDS plane (li + TXH*lj)
  int gi[NS], gj[NS];
  bool own[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = (int)threadIdx.x + NT * s;
    const int lj = p / NPX, li = 2 * (p - lj * NPX);
    const int i = x0 + li, j = y0 + lj;
    const bool in = (p < NPAIR) && i >= 0 && i + 1 < nx && j >= 0 && j < ny;
    gi[s] = i;
    gj[s] = j;
    goff[s] = in ? i + nx * j : -1;
    loff[s] = (p < NPAIR) ? li + TXH * lj : 0;
    own[s] = in && li >= 2 && li < TXH - 2 && lj >= 2 && lj < TYH - 2;
  }

#define NDSM_LOAD_PLANE(base, k, dst)                          \
  do {                                                          \
    const double *pk_ = (base) + sz * (size_t)(k);              \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {         \
      d2 t_;                                                    \
      t_.x = 0.0;                                               \
      t_.y = 0.0;                                               \
      if (goff[s_] >= 0) t_ = ld2(pk_ + goff[s_]);              \
      dst[s_] = t_;                                             \
    }                                                           \
  } while (0)

  double *Pc = lds;          // plane k   : O_k, red points updated in place -> R_k
  double *Pp = lds + PLANE;  // plane k-1 : R_{k-1}

  d2 c[NS], m1[NS], nxt[NS], nn[NS];
  d2 rk[NS], rn[NS];
  double m2e[NS], rm1e[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    m1[s].x = m1[s].y = 0.0;
    nxt[s].x = nxt[s].y = 0.0;
    nn[s].x = nn[s].y = 0.0;
    rk[s].x = rk[s].y = 0.0;
    rn[s].x = rn[s].y = 0.0;
    m2e[s] = 0.0;
    rm1e[s] = 0.0;
  }

  // ---- prologue: planes ks and ks+1 ------------------------------------
  NDSM_LOAD_PLANE(u, ks, c);
  if (!RHS0) NDSM_LOAD_PLANE(rhs, ks, rk);
  if (ks + 1 <= ke) {
    NDSM_LOAD_PLANE(u, ks + 1, nxt);
    if (!RHS0) NDSM_LOAD_PLANE(rhs, ks + 1, rn);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if ((int)threadIdx.x + NT * s < NPAIR) st2(Pc + loff[s], c[s]);
  __syncthreads();

  const int red_lo = max(zs - 1, 0), red_hi = min(ze, nz - 1);

  for (int k = ks; k <= ze; ++k) {
    const int kg = k + g.k0;
    // request plane k+2 before touching plane k
    d2 rnn[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) rnn[s].x = rnn[s].y = 0.0;
    if (k + 2 <= ke) {
      NDSM_LOAD_PLANE(u, k + 2, nn);
      if (!RHS0) NDSM_LOAD_PLANE(rhs, k + 2, rnn);
    }

    // ---------------- stage 0: red points of plane k ------------------
    if (k >= red_lo && k <= red_hi && k >= g.lb[2] && k <= g.ub[2]) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (goff[s] < 0) continue;
        const int i = gi[s], j = gj[s];
        const int e = (((i + j + kg) & 1) == ((g.first_par) & 1)) ? 0 : 1;  // which element of the pair is red
        const int ii = i + e;
        const int lo = loff[s];
        const int lj = lo / TXH, li = lo - lj * TXH;
        if (ii < g.lb[0] || ii > g.ub[0] || j < g.lb[1] || j > g.ub[1]) continue;
        // in-plane neighbours must be inside the loaded region (ring 0 is never updated)
        const bool xmir = (e == 0) ? (ii == 0) : (ii == nx - 1);
        const int lxn = (e == 0) ? li - 1 : li + 2;
        if (!xmir && (lxn < 0 || lxn >= TXH)) continue;
        const int ljl = (j == 0) ? lj + 1 : lj - 1;
        const int ljh = (j == ny - 1) ? lj - 1 : lj + 1;
        if (ljl < 0 || ljl >= TYH || ljh < 0 || ljh >= TYH) continue;
        const double other = pick(c[s], 1 - e);
        const double xn = xmir ? other : Pc[lj * TXH + lxn];
        const double xs = (e == 0) ? (other + xn) : (xn + other);  // u(xh) + u(xl)
        const double ys = Pc[ljh * TXH + li + e] + Pc[ljl * TXH + li + e];
        const double zhv = (kg == g.nzg - 1) ? pick(m1[s], e) : pick(nxt[s], e);
        const double zlv = (kg == 0) ? pick(nxt[s], e) : pick(m1[s], e);
        const double zsum = zhv + zlv;
        const double rr = RHS0 ? 0.0 : pick(rk[s], e);
        const double unew = xs * g.w[0] + ys * g.w[1] + zsum * g.w[2] - rr;
        const double res = g.w1 * unew;
        c[s] = put(c[s], e, res);
        Pc[lo + e] = res;
      }
    }

    // ---------------- stage 1: black points of plane k-1 --------------
    const int kb = k - 1;
    if (kb >= zs && kb < ze) {
      const int kbg = kb + g.k0;
      const bool zupd = kb >= g.lb[2] && kb <= g.ub[2];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (!own[s]) continue;
        const int i = gi[s], j = gj[s];
        // black element of plane k-1 sits where the red element of plane k sits
        const int e = (((i + j + kg) & 1) == ((g.first_par) & 1)) ? 0 : 1;
        const int ii = i + e;
        const int lo = loff[s];
        const int lj = lo / TXH, li = lo - lj * TXH;
        if (zupd && ii >= g.lb[0] && ii <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1]) {
          const bool xmir = (e == 0) ? (ii == 0) : (ii == nx - 1);
          const int lxn = (e == 0) ? li - 1 : li + 2;
          const int ljl = (j == 0) ? lj + 1 : lj - 1;
          const int ljh = (j == ny - 1) ? lj - 1 : lj + 1;
          const double other = pick(m1[s], 1 - e);
          const double xn = xmir ? other : Pp[lj * TXH + lxn];
          const double xs = (e == 0) ? (other + xn) : (xn + other);
          const double ys = Pp[ljh * TXH + li + e] + Pp[ljl * TXH + li + e];
          const double zhv = (kbg == g.nzg - 1) ? m2e[s] : pick(c[s], e);
          const double zlv = (kbg == 0) ? pick(c[s], e) : m2e[s];
          const double zsum = zhv + zlv;
          const double rr = RHS0 ? 0.0 : rm1e[s];
          const double unew = xs * g.w[0] + ys * g.w[1] + zsum * g.w[2] - rr;
          m1[s] = put(m1[s], e, g.w1 * unew);
        }
        st2(uout + sz * (size_t)kb + goff[s], m1[s]);
      }
    }

    __syncthreads();  // all reads of Pp (R_{k-1}) and red writes into Pc are done

    // ---------------- rotate the window -------------------------------
    {
      // element the NEXT iteration's black stage needs from plane k-1 / rhs of plane k
      const int kg1 = kg + 1;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int e1 = (((gi[s] + gj[s] + kg1) & 1) == ((g.first_par) & 1)) ? 0 : 1;
        m2e[s] = pick(m1[s], e1);
        rm1e[s] = RHS0 ? 0.0 : pick(rk[s], e1);
        m1[s] = c[s];
        c[s] = nxt[s];
        nxt[s] = nn[s];
        if (!RHS0) {
          rk[s] = rn[s];
          rn[s] = rnn[s];
        }
      }
      double *t = Pp;
      Pp = Pc;
      Pc = t;
      if (k + 1 <= ke) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if ((int)threadIdx.x + NT * s < NPAIR) st2(Pc + loff[s], c[s]);
      }
    }
    __syncthreads();
  }
}

template <int TXH, int TYH, int NT>
int launch_cfg(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, int target_wgs) {
  constexpr int TXI = TXH - 4, TYI = TYH - 4;
  FusedPlan pl;
  pl.ntx = (g.n[0] + TXI - 1) / TXI;
  pl.nty = (g.n[1] + TYI - 1) / TYI;
  const int tiles = pl.ntx * pl.nty;
  // z chunks: enough work items to fill the chip once, but chunks of >= 16 planes
  int nzc = (target_wgs + tiles - 1) / tiles;
  if (nzc < 1) nzc = 1;
  int zc = (g.n[2] + nzc - 1) / nzc;
  if (zc < 16) zc = 16 < g.n[2] ? 16 : g.n[2];
  pl.zc = zc;
  pl.nzc = (g.n[2] + zc - 1) / zc;
  pl.nwork = tiles * pl.nzc;
  const int nblk = ((pl.nwork + 7) / 8) * 8;
  const size_t lds_bytes = sizeof(double) * 2 * TXH * TYH;
  auto kfn = rbgs3_fused_k<TXH, TYH, NT, false>;
  static bool attr_set = false;
  if (!attr_set) {
    NDSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(nblk), dim3(NT), lds_bytes, ndsm::stream(), u, uout, rhs, g, pl);
  NDSM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

namespace ndsm {

int launch_rbgs3_fused(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, bool *handled) {
  *handled = false;
  if (!uout || g.ndim != 3 || (g.n[0] & 1) || g.n[0] < 16 || g.n[1] < 16 || g.n[2] < 16) return 0;
  // tile 132 x 31 (128 x 27 owned), 512 threads, 2 workgroups per CU
  int rc = launch_cfg<132, 31, 512>(g, u, uout, rhs, 512);
  if (rc) return rc;
  *handled = true;
  return 0;
}

}  // namespace ndsm
