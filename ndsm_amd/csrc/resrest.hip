// Fused residual + restriction: rhs_c = R (rhs - L u), u_c = 0, in ONE pass over
// the fine level - the residual never goes to HBM.
//
// Reference: fine_to_coarse computes r_f = rhs - L u into a fresh array
// (ndsm_multigrid_core.f90:528-539, poisson_residual_3D ndsm_optimized.f90:346-447)
// and then restricts it point by point (mg_restrict :1010-1065, nrestrict
// ndsm_interp.f90:186-292).  Separately those move 24 + 9 B per fine point; fused,
// the algorithmic traffic is u + rhs in, rhs_c (+ u_c) out = 16 + 2 B per fine point.
//
// Schedule per workgroup: a tile of CI x CJ coarse columns and a chunk of coarse
// planes.  The fine planes that feed the chunk are streamed in z:
//   iteration k : residual of fine plane k on the fine footprint of the tile
//                 (z neighbours in registers, in-plane neighbours from an LDS
//                 copy of u's plane k)  ->  LDS plane R
//                 then every thread, now owning ONE coarse column (I,J), adds
//                 plane k's taps to the (at most 3) coarse planes whose z window
//                 contains k.
// A coarse value is the reference's sum  fc = fc + w * r  over its taps in
// (z, y, x) order with w = ((((c2x w2x) c2y) w2y) c2z) w2z  (ndsm_interp.f90:263-290).
// Streaming the planes in ascending z and walking (y, x) inside a plane visits
// the taps in exactly that order, so the result is bit-identical to
// residual3 + restrict_k.
#include "common.hpp"

namespace {

constexpr int MAXT = 6;  // taps per dimension this kernel is instantiated for

struct RRArgs {
  int nf[3], nc[3];
  int lb[3], ub[3];
  double w[3], wc;
  const int32_t *rlo[3], *rcnt[3];
  const double *rw[3];
  int maxt[3];
  double w2[3];
  int nti, ntj, nkc, kc;  // coarse tiles in x, y; chunks in z; coarse planes per chunk
  int nwork;
};

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
__device__ __forceinline__ void st2(double *p, const d2 a) {
  double2 t;
  t.x = a.x;
  t.y = a.y;
  *reinterpret_cast<double2 *>(p) = t;
}

// CI x CJ coarse columns per workgroup = NT threads.  Fine footprint UX x UY
// (u incl. its one-point stencil halo), x-pairs as in the smoother.
template <int CI, int CJ>
__global__ __launch_bounds__(CI *CJ) void resrest_k(const double *__restrict__ u, const double *__restrict__ rhs,
                                                    double *__restrict__ rhs_c, double *__restrict__ u_c, RRArgs a) {
  constexpr int NT = CI * CJ;
  constexpr int UX = 2 * CI + 8, UY = 2 * CJ + 6;  // loaded u tile
  constexpr int NPX = UX / 2, NPAIR = NPX * UY, NS = (NPAIR + NT - 1) / NT;
  constexpr int PLANE = UX * UY;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *U = lds, *R = lds + PLANE;

  const int nb8 = gridDim.x >> 3;
  const int wk = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
  if (wk >= a.nwork) return;
  const int tj = wk % a.ntj;
  const int t2 = wk / a.ntj;
  const int ti = t2 % a.nti;
  const int ck = t2 / a.nti;

  const int nx = a.nf[0], ny = a.nf[1], nz = a.nf[2];
  const size_t sz = (size_t)nx * (size_t)ny;
  const int I0 = ti * CI, J0 = tj * CJ;
  const int Ks = ck * a.kc, Ke = min(Ks + a.kc, a.nc[2]);
  // fine footprint origin: first tap of the first coarse column, minus the stencil halo, even
  const int fx0 = (a.rlo[0][I0] - 1) & ~1;
  const int fy0 = a.rlo[1][J0] - 1;
  const int kA = a.rlo[2][Ks];
  const int kB = a.rlo[2][Ke - 1] + a.rcnt[2][Ke - 1] - 1;  // last fine plane of the chunk
  const int tid = (int)threadIdx.x;

  // ---- this thread's coarse column --------------------------------
  const int I = I0 + tid % CI, J = J0 + tid / CI;
  const bool chave = I < a.nc[0] && J < a.nc[1];
  int ni = 0, nj = 0, li0 = 0, lj0 = 0;
  double cx[MAXT], cy[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) cx[t] = cy[t] = 0.0;
  if (chave) {
    ni = a.rcnt[0][I];
    nj = a.rcnt[1][J];
    li0 = a.rlo[0][I] - fx0;
    lj0 = a.rlo[1][J] - fy0;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < ni) cx[t] = a.rw[0][(size_t)I * a.maxt[0] + t];
      if (t < nj) cy[t] = a.rw[1][(size_t)J * a.maxt[1] + t];
    }
  }
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;  // coarse planes K with K & 3 = 0..3
  int Klo = Ks;                                            // first coarse plane not yet finished

  // ---- fine pairs this thread loads / computes the residual of -----
  d2 um[NS], uc[NS], up[NS], rr[NS], rn[NS];
  auto geom = [&](int s, int &li, int &lj, int &i, int &j, bool &in) {
    const int p = tid + NT * s;
    lj = p / NPX;
    li = 2 * (p - lj * NPX);
    i = fx0 + li;
    j = fy0 + lj;
    in = p < NPAIR && i >= 0 && i + 1 < nx && j >= 0 && j < ny;
  };
#define RR_LOAD(base, k, dst)                                      \
  do {                                                              \
    const int kk_ = (k);                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {             \
      int li_, lj_, i_, j_;                                         \
      bool in_;                                                     \
      geom(s_, li_, lj_, i_, j_, in_);                              \
      d2 t_;                                                        \
      t_.x = 0.0;                                                   \
      t_.y = 0.0;                                                   \
      if (in_ && kk_ >= 0 && kk_ < nz) t_ = ld2((base) + sz * (size_t)kk_ + (i_ + nx * j_)); \
      dst[s_] = t_;                                                 \
    }                                                               \
  } while (0)

  RR_LOAD(u, kA - 1, um);
  RR_LOAD(u, kA, uc);
  RR_LOAD(u, kA + 1, up);
  RR_LOAD(rhs, kA, rr);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    int li, lj, i, j;
    bool in;
    geom(s, li, lj, i, j, in);
    if (tid + NT * s < NPAIR) st2(U + li + UX * lj, uc[s]);
  }
  __syncthreads();

  for (int k = kA; k <= kB; ++k) {
    d2 un[NS];
    RR_LOAD(u, k + 2, un);
    RR_LOAD(rhs, k + 1, rn);

    // ---- residual of fine plane k (ndsm_optimized.f90:399-435 + :439-445) ----
    const bool zin = k >= a.lb[2] && k <= a.ub[2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int li, lj, i, j;
      bool in;
      geom(s, li, lj, i, j, in);
      if (tid + NT * s >= NPAIR) continue;
      d2 res;
      res.x = 0.0;
      res.y = 0.0;
      // the outer ring of the loaded tile only serves as stencil halo
      if (in && lj >= 1 && lj < UY - 1) {
        const int ljl = (j == 0) ? lj + 1 : lj - 1;
        const int ljh = (j == ny - 1) ? lj - 1 : lj + 1;
        const bool yin = zin && j >= a.lb[1] && j <= a.ub[1];
        const double zl0 = (k == 0) ? up[s].x : um[s].x, zh0 = (k == nz - 1) ? um[s].x : up[s].x;
        const double zl1 = (k == 0) ? up[s].y : um[s].y, zh1 = (k == nz - 1) ? um[s].y : up[s].y;
        if (li >= 2 && yin && i >= a.lb[0] && i <= a.ub[0]) {  // element 0 (x index i)
          const double xl = (i == 0) ? uc[s].y : U[li - 1 + UX * lj];
          const double xh = uc[s].y;
          const double v = (xl + xh) * a.w[0] + (U[li + UX * ljl] + U[li + UX * ljh]) * a.w[1] + (zl0 + zh0) * a.w[2] -
                           rr[s].x - uc[s].x * a.wc;
          res.x = -v;
        }
        if (li + 2 < UX && yin && i + 1 >= a.lb[0] && i + 1 <= a.ub[0]) {  // element 1 (x index i+1)
          const double xl = uc[s].x;
          const double xh = (i + 1 == nx - 1) ? uc[s].x : U[li + 2 + UX * lj];
          const double v = (xl + xh) * a.w[0] + (U[li + 1 + UX * ljl] + U[li + 1 + UX * ljh]) * a.w[1] +
                           (zl1 + zh1) * a.w[2] - rr[s].y - uc[s].y * a.wc;
          res.y = -v;
        }
      }
      st2(R + li + UX * lj, res);
    }
    __syncthreads();

    // ---- add plane k's taps to the coarse planes whose z window holds k ----
    if (chave) {
      for (int K = Klo; K < Ke; ++K) {
        const int z0 = a.rlo[2][K];
        if (z0 > k) break;
        const int nk = a.rcnt[2][K];
        if (k >= z0 + nk) continue;
        const double c2z = a.rw[2][(size_t)K * a.maxt[2] + (k - z0)];
        const int slot = K & 3;
        double fc = slot == 0 ? acc0 : (slot == 1 ? acc1 : (slot == 2 ? acc2 : acc3));
#pragma unroll
        for (int jj = 0; jj < MAXT; ++jj) {
          if (jj >= nj) break;
          const double *row = R + li0 + UX * (lj0 + jj);
#pragma unroll
          for (int ii = 0; ii < MAXT; ++ii) {
            if (ii >= ni) break;
            double wv = cx[ii] * a.w2[0];
            wv = wv * cy[jj] * a.w2[1];
            wv = wv * c2z * a.w2[2];
            fc = fc + wv * row[ii];
          }
        }
        if (k == z0 + nk - 1) {  // window complete
          const size_t c = (size_t)I + (size_t)a.nc[0] * ((size_t)J + (size_t)a.nc[1] * (size_t)K);
          rhs_c[c] = fc;
          u_c[c] = 0.0;
          fc = 0.0;
        }
        acc0 = slot == 0 ? fc : acc0;
        acc1 = slot == 1 ? fc : acc1;
        acc2 = slot == 2 ? fc : acc2;
        acc3 = slot == 3 ? fc : acc3;
      }
      while (Klo < Ke && a.rlo[2][Klo] + a.rcnt[2][Klo] - 1 <= k) ++Klo;
    }
    __syncthreads();

    // ---- rotate ----
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      um[s] = uc[s];
      uc[s] = up[s];
      up[s] = un[s];
      rr[s] = rn[s];
      int li, lj, i, j;
      bool in;
      geom(s, li, lj, i, j, in);
      if (tid + NT * s < NPAIR) st2(U + li + UX * lj, uc[s]);
    }
    __syncthreads();
  }
#undef RR_LOAD
}

}  // namespace

// Tile constants for the host-side applicability check (ndsmh_mg.f90): the fine
// taps of every coarse tile must fit the loaded footprint.
extern "C" void ndsmk_resrest_tile(int *ci, int *cj, int *ux, int *uy, int *maxt) {
  *ci = 64;
  *cj = 16;
  *ux = 2 * 64 + 8;
  *uy = 2 * 16 + 6;
  *maxt = MAXT;
}

// Caller guarantees applicability (3-D, not a z-slab window, nx even, taps fit
// the tile: checked once per hierarchy on the host).
extern "C" int ndsmk_residual_restrict(const ndsmk_grid *gp, const ndsmk_xfer *x, const double *u,
                                       const double *rhs, double *rhs_c, double *u_c) {
  NDSM_REQUIRE_READY();
  const ndsmk_grid g = *gp;
  constexpr int CI = 64, CJ = 16;
  NDSM_CHECK_ARG(g.ndim == 3 && u_c && rhs_c);
  NDSM_CHECK_ARG(g.zown0 == 0 && g.zown1 == g.n[2] && g.k0 == 0 && (g.n[0] & 1) == 0);
  for (int d = 0; d < 3; ++d) NDSM_CHECK_ARG(x->maxt[d] <= MAXT && x->nf[d] == g.n[d]);
  RRArgs a;
  for (int d = 0; d < 3; ++d) {
    a.nf[d] = g.n[d];
    a.nc[d] = x->nc[d];
    a.lb[d] = g.lb[d];
    a.ub[d] = g.ub[d];
    a.w[d] = g.w[d];
    a.rlo[d] = x->rlo[d];
    a.rcnt[d] = x->rcnt[d];
    a.rw[d] = x->rw[d];
    a.maxt[d] = x->maxt[d];
    a.w2[d] = x->w2[d];
  }
  a.wc = g.wc;
  a.nti = (x->nc[0] + CI - 1) / CI;
  a.ntj = (x->nc[1] + CJ - 1) / CJ;
  const int tiles = a.nti * a.ntj;
  int nkc = (512 + tiles - 1) / tiles;
  if (nkc < 1) nkc = 1;
  int kc = (x->nc[2] + nkc - 1) / nkc;
  if (kc < 8) kc = 8 < x->nc[2] ? 8 : x->nc[2];
  a.kc = kc;
  a.nkc = (x->nc[2] + kc - 1) / kc;
  a.nwork = tiles * a.nkc;
  const int nblk = ((a.nwork + 7) / 8) * 8;
  constexpr size_t lds_bytes = sizeof(double) * 2 * (2 * CI + 8) * (2 * CJ + 6);
  auto kfn = resrest_k<CI, CJ>;
  static int attr_epoch = 0;
  if (ndsm::first_in_epoch(attr_epoch))
    NDSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_bytes));
  hipLaunchKernelGGL(kfn, dim3(nblk), dim3(CI * CJ), lds_bytes, ndsm::stream(), u, rhs, rhs_c, u_c, a);
  NDSM_LAUNCH_CHECK();
  return 0;
}
