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

#include <cstdlib>

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

// geometry of slot s of this thread, recomputed where needed (cheap integer
// ops) instead of being held in registers across the z loop
template <int TXH, int TYH, int NT>
struct Slot {
  int li, lj, i, j, lo;
  bool live, in, own;
  __device__ __forceinline__ Slot(int tid, int s, int x0, int y0, int nx, int ny) {
    constexpr int NPX = TXH / 2;
    const int p = tid + NT * s;
    lj = p / NPX;
    li = 2 * (p - lj * NPX);
    i = x0 + li;
    j = y0 + lj;
    live = p < NPX * TYH;
    in = live && i >= 0 && i + 1 < nx && j >= 0 && j < ny;
    own = in && li >= 2 && li < TXH - 2 && lj >= 2 && lj < TYH - 2;
    lo = live ? li + TXH * lj : 0;
  }
};

template <int TXH, int TYH, int NT, int WPS, bool RHS0>
__global__ __launch_bounds__(NT, WPS) void rbgs3_fused_k(const double *__restrict__ u, double *__restrict__ uout,
                                                         const double *__restrict__ rhs, ndsmk_grid g,
                                                         FusedPlan pl) {
  constexpr int NPX = TXH / 2;
  constexpr int NPAIR = NPX * TYH;
  constexpr int NS = (NPAIR + NT - 1) / NT;
  constexpr int TXI = TXH - 4, TYI = TYH - 4;
  constexpr int PLANE = TXH * TYH;
  using SlotT = Slot<TXH, TYH, NT>;
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
  const int zs = g.zown0 + cz * pl.zc;
  const int ze = min(zs + pl.zc, g.zown1);
  const int ks = max(zs - 2, 0);
  const int ke = min(ze + 1, nz - 1);
  const size_t sz = (size_t)nx * (size_t)ny;
  const int tid0 = (int)threadIdx.x;
  const int tid = tid0;
  const int fp = g.first_par & 1;

#define NDSM_LOAD_PLANE(base, k, dst)                              \
  do {                                                              \
    const double *pk_ = (base) + sz * (size_t)(k);                  \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {             \
      const SlotT q_(tid, s_, x0, y0, nx, ny);                      \
      d2 t_;                                                        \
      t_.x = 0.0;                                                   \
      t_.y = 0.0;                                                   \
      if (q_.in) t_ = ld2(pk_ + (q_.i + nx * q_.j));                \
      dst[s_] = t_;                                                 \
    }                                                               \
  } while (0)

  double *Pc = lds;          // plane k   : O_k, red points updated in place -> R_k
  double *Pp = lds + PLANE;  // plane k-1 : R_{k-1}

  // Register window.  Own values of planes k and k-1 are re-read from LDS; only
  // what LDS does not hold stays in registers:
  //   nxt  = O_{k+1} (arrived)            nn  = O_{k+2} (in flight)
  //   m2e  = the one element of plane k-2 the black stage needs
  //   rk   = rhs of plane k, rn = rhs of plane k+1 (in flight), rm1e = rhs element for the black stage
  d2 nxt[NS], nn[NS], rk[NS], rn[NS];
  double m2e[NS], rm1e[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    nxt[s].x = nxt[s].y = 0.0;
    nn[s].x = nn[s].y = 0.0;
    rk[s].x = rk[s].y = 0.0;
    rn[s].x = rn[s].y = 0.0;
    m2e[s] = 0.0;
    rm1e[s] = 0.0;
  }

  // ---- prologue: plane ks into LDS, plane ks+1 into registers ----------
  {
    d2 c0[NS];
    NDSM_LOAD_PLANE(u, ks, c0);
    if (!RHS0) NDSM_LOAD_PLANE(rhs, ks, rk);
    if (ks + 1 <= ke) NDSM_LOAD_PLANE(u, ks + 1, nxt);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const SlotT q(tid, s, x0, y0, nx, ny);
      if (q.live) st2(Pc + q.lo, c0[s]);
    }
  }
  __syncthreads();

  const int red_lo = max(zs - 1, 0), red_hi = min(ze, nz - 1);

  for (int k = ks; k <= ze; ++k) {
    const int kg = k + g.k0;
    // make the thread index opaque once per iteration: the slot geometry is then
    // recomputed (a few integer ops) instead of being kept live across the loop
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    // request plane k+2 of u and plane k+1 of rhs before touching plane k
    if (k + 2 <= ke) NDSM_LOAD_PLANE(u, k + 2, nn);
    if (!RHS0 && k + 1 <= ke) NDSM_LOAD_PLANE(rhs, k + 1, rn);

    const bool do_red = k >= red_lo && k <= red_hi && k >= g.lb[2] && k <= g.ub[2];
    const int kb = k - 1;
    const bool do_black = kb >= zs && kb < ze;
    const int kbg = kb + g.k0;
    const bool zupd = kb >= g.lb[2] && kb <= g.ub[2];

#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const SlotT q(tid, s, x0, y0, nx, ny);
      if (!q.in) continue;
      // the pair element that is red in plane k is the one that is black in plane k-1
      const int e = (((q.i + q.j + kg) & 1) == fp) ? 0 : 1;
      const int ii = q.i + e;
      const bool xmir = (e == 0) ? (ii == 0) : (ii == nx - 1);
      const int lxn = (e == 0) ? q.li - 1 : q.li + 2;
      const int ljl = (q.j == 0) ? q.lj + 1 : q.lj - 1;
      const int ljh = (q.j == ny - 1) ? q.lj - 1 : q.lj + 1;
      const bool inb = ii >= g.lb[0] && ii <= g.ub[0] && q.j >= g.lb[1] && q.j <= g.ub[1];
      const d2 cc = ld2(Pc + q.lo);  // O_k
      double cnew = pick(cc, e);

      // ---------------- stage 0: red point of plane k -------------------
      // ring 0 of the loaded region has no in-plane neighbours: never updated
      if (do_red && inb && (xmir || (lxn >= 0 && lxn < TXH)) && ljl >= 0 && ljl < TYH && ljh >= 0 && ljh < TYH) {
        const double other = pick(cc, 1 - e);
        const double xn = xmir ? other : Pc[q.lj * TXH + lxn];
        const double xs = (e == 0) ? (other + xn) : (xn + other);  // u(xh) + u(xl)
        const double ys = Pc[ljh * TXH + q.li + e] + Pc[ljl * TXH + q.li + e];
        const double m1v = Pp[q.lo + e];  // plane k-1, black there: still the old value
        const double zhv = (kg == g.nzg - 1) ? m1v : pick(nxt[s], e);
        const double zlv = (kg == 0) ? pick(nxt[s], e) : m1v;
        const double zsum = zhv + zlv;
        const double rr = RHS0 ? 0.0 : pick(rk[s], e);
        const double unew = xs * g.w[0] + ys * g.w[1] + zsum * g.w[2] - rr;
        cnew = g.w1 * unew;
        Pc[q.lo + e] = cnew;
      }

      // ---------------- stage 1: black point of plane k-1 ---------------
      if (do_black && q.own) {
        d2 mm = ld2(Pp + q.lo);  // R_{k-1}
        if (zupd && inb) {
          const double other = pick(mm, 1 - e);
          const double xn = xmir ? other : Pp[q.lj * TXH + lxn];
          const double xs = (e == 0) ? (other + xn) : (xn + other);
          const double ys = Pp[ljh * TXH + q.li + e] + Pp[ljl * TXH + q.li + e];
          const double zhv = (kbg == g.nzg - 1) ? m2e[s] : cnew;
          const double zlv = (kbg == 0) ? cnew : m2e[s];
          const double zsum = zhv + zlv;
          const double rr = RHS0 ? 0.0 : rm1e[s];
          const double unew = xs * g.w[0] + ys * g.w[1] + zsum * g.w[2] - rr;
          mm = put(mm, e, g.w1 * unew);
        }
        st2(uout + sz * (size_t)kb + (q.i + nx * q.j), mm);
      }
      // what the NEXT iteration's black stage needs: plane k-1 at the element
      // that is red there (final since the red stage of the previous iteration)
      m2e[s] = Pp[q.lo + 1 - e];
      rm1e[s] = RHS0 ? 0.0 : pick(rk[s], 1 - e);
    }

    __syncthreads();  // all reads of Pp (R_{k-1}) and all red writes into Pc are done

    // ---------------- rotate: O_{k+1} takes the place of R_{k-1} ---------
    {
      double *t = Pp;
      Pp = Pc;
      Pc = t;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const SlotT q(tid, s, x0, y0, nx, ny);
        if (k + 1 <= ke && q.live) st2(Pc + q.lo, nxt[s]);
        nxt[s] = nn[s];
        if (!RHS0) rk[s] = rn[s];
      }
    }
    __syncthreads();
  }
#undef NDSM_LOAD_PLANE
}

template <int TXH, int TYH, int NT, int WPS>
int launch_cfg(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, int target_wgs) {
  constexpr int TXI = TXH - 4, TYI = TYH - 4;
  FusedPlan pl;
  pl.ntx = (g.n[0] + TXI - 1) / TXI;
  pl.nty = (g.n[1] + TYI - 1) / TYI;
  const int tiles = pl.ntx * pl.nty;
  // z chunks: enough work items to fill the chip once, but chunks of >= 16 planes
  int nzc = (target_wgs + tiles - 1) / tiles;
  if (nzc < 1) nzc = 1;
  const int nzo = g.zown1 - g.zown0;  // owned planes
  int zc = (nzo + nzc - 1) / nzc;
  if (zc < 16) zc = 16 < nzo ? 16 : nzo;
  pl.zc = zc;
  pl.nzc = (nzo + zc - 1) / zc;
  pl.nwork = tiles * pl.nzc;
  const int nblk = ((pl.nwork + 7) / 8) * 8;
  const size_t lds_bytes = sizeof(double) * 2 * TXH * TYH;
  auto kfn = rbgs3_fused_k<TXH, TYH, NT, WPS, false>;
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

// Development knob: NDSM_FUSED_CFG=<n> picks a tile configuration (default 0).
static int fused_cfg() {
  static int cfg = -1;
  if (cfg < 0) {
    const char *e = std::getenv("NDSM_FUSED_CFG");
    cfg = e ? std::atoi(e) : 0;
  }
  return cfg;
}

int launch_rbgs3_fused(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, bool *handled) {
  *handled = false;
  if (!uout || g.ndim != 3 || (g.n[0] & 1) || g.n[0] < 16 || g.n[1] < 16 || g.zown1 - g.zown0 < 8) return 0;
  // a z-streaming workgroup walks >= 16 planes serially: with fewer than ~one
  // workgroup per CU the sweep is latency bound and the two colour passes win
  const int64_t npts = (int64_t)g.n[0] * g.n[1] * (g.zown1 - g.zown0);
  if (npts < (int64_t)6 * 1024 * 1024 && g.zown1 - g.zown0 == g.n[2]) return 0;
  // Tile choice measured on MI355X (scripts/tune_smoother.py): the sweep is bound
  // by fabric traffic (halo rows + chunk warm-up planes, ~1.3x compulsory), and
  // more, shorter chunks beat fewer, longer ones up to ~8 work items per CU.
  int rc;
  const int cfg = fused_cfg();
  if (cfg == 0) {
    if (npts >= (int64_t)64 * 1024 * 1024)
      rc = launch_cfg<132, 31, 512, 4>(g, u, uout, rhs, 2048);
    else
      rc = launch_cfg<68, 30, 256, 4>(g, u, uout, rhs, 1024);
  } else {
    switch (cfg) {
      case 1: rc = launch_cfg<132, 62, 1024, 4>(g, u, uout, rhs, 1024); break;
      case 2: rc = launch_cfg<68, 30, 256, 4>(g, u, uout, rhs, 1024); break;
      case 3: rc = launch_cfg<132, 31, 1024, 8>(g, u, uout, rhs, 2048); break;
      case 4: rc = launch_cfg<68, 60, 512, 4>(g, u, uout, rhs, 1024); break;
      case 5: rc = launch_cfg<132, 31, 512, 4>(g, u, uout, rhs, 1024); break;
      default: rc = launch_cfg<132, 31, 512, 4>(g, u, uout, rhs, 2048); break;
    }
  }
  if (rc) return rc;
  *handled = true;
  return 0;
}

}  // namespace ndsm
