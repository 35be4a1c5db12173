// Mixed-precision mode (BASELINE config[4]: "fp32 smoother / fp64 residual"): level 1 of a 3-D
// solve is run as iterative refinement,
//
//     r  = rhs - L u              fp64 arithmetic, stored as fp32
//     L e = r                     ONE V-cycle on the correction, e = 0 initially; level 1 of it
//                                 (smoother, residual, transfers' fine side) in fp32, levels >= 2
//                                 unchanged in fp64
//     u  = u + e                  fp64
//
// which in exact arithmetic is the reference's V-cycle on u itself (a linear stationary
// iteration started from u equals u + the same iteration on the error equation started from 0;
// ndsm_multigrid_core.f90:341-377), so the iterates agree with the fp64 path up to fp32 rounding
// of a correction that shrinks every cycle, and the attainable accuracy is that of the fp64
// residual.  max|u_new - u_old| of update_u (:1077-1122) is max|e|.
//
// This file: the fp64 side of the refinement as ONE z-streaming pass per cycle
// (u' = u + e ; e' = 0 ; r = rhs - L u' ; max|e|, sum|e|), and the fp32 smoother driver.  The level-1
// arrays cost 8 + 4 + 4 + 4 + 4 B per point instead of 8 x 4.
#include "common.hpp"

namespace ndsm {
int launch_rbgs3_fused_f32(const ndsmk_grid &g, const float *u, float *uout, const float *rhs, int max_sweeps,
                           bool force, int *sweeps_done, float *rout, int *res_done);
}

namespace {

struct UPlan {
  int ntx, nty, nzc, zc, nwork;
};

// Tile of TXH x TYH points (1-ring halo), one x-pair per thread.  Plane k of the UPDATED field
// sits in LDS for its in-plane neighbours; the thread's own pairs of planes k-1, k, k+1 are in
// registers; plane k+2 (u and e) is in flight.
template <int TXH, int TYH, int NT>
__global__ __launch_bounds__(NT) void update_residual_k(const double *__restrict__ u, double *__restrict__ unew,
                                                        const double *__restrict__ rhs,
                                                        const float *__restrict__ e, float *__restrict__ ezero,
                                                        float *__restrict__ r, ndsmk_grid g, UPlan pl,
                                                        double *__restrict__ part) {
  constexpr int NPX = TXH / 2;
  constexpr int NPAIR = NPX * TYH;
  static_assert(NPAIR <= NT, "one pair per thread");
  constexpr int TXI = TXH - 4, TYI = TYH - 2;  // x halo 2 (pairs stay aligned), y halo 1
  constexpr int PLANE = TXH * TYH;
  __shared__ __attribute__((aligned(16))) double lds[2 * PLANE];
  __shared__ double smx[NT / 64], ssm[NT / 64];

  const int nb8 = gridDim.x >> 3;
  const int w = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
  const int tid = (int)threadIdx.x;
  double mx = 0.0, sm = 0.0;
  if (w < pl.nwork) {
    const int ty = w % pl.nty;
    const int t2 = w / pl.nty;
    const int tx = t2 % pl.ntx;
    const int cz = t2 / pl.ntx;
    const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
    const size_t sz = (size_t)nx * (size_t)ny;
    const int x0 = tx * TXI - 2, y0 = ty * TYI - 1;
    const int zs = g.zown0 + cz * pl.zc, ze = min(zs + pl.zc, g.zown1);   // owned planes (z-slab: ghosts either side)

    const int lj = tid / NPX, li = 2 * (tid - lj * NPX);
    const int i = x0 + li, j = y0 + lj;
    const bool live = tid < NPAIR;
    const bool in = live && i >= 0 && i + 1 < nx && j >= 0 && j < ny;
    const bool own = in && li >= 2 && li < TXH - 2 && lj >= 1 && lj < TYH - 1;
    const int lo = live ? li + TXH * lj : 0;
    const size_t go = (size_t)(in ? i + nx * j : 0);
    // in-plane neighbour offsets, mirrored at the physical faces (ndsm_optimized.f90:400-421)
    const int oyl = (j == 0) ? lo + TXH : lo - TXH;
    const int oyh = (j == ny - 1) ? lo - TXH : lo + TXH;
    const bool mir0 = i == 0, mir1 = i + 1 == nx - 1;
    const bool yin = j >= g.lb[1] && j <= g.ub[1];
    const bool in0 = yin && i >= g.lb[0] && i <= g.ub[0];
    const bool in1 = yin && i + 1 >= g.lb[0] && i + 1 <= g.ub[0];

    // updated pair of plane k (u + e), zero outside the domain
    auto load = [&](int k, double &ax, double &ay, float &ex, float &ey) {
      ax = ay = 0.0;
      ex = ey = 0.0f;
      if (in && k >= 0 && k < nz) {
        const double2 uu = *reinterpret_cast<const double2 *>(u + sz * (size_t)k + go);
        ax = uu.x;
        ay = uu.y;
        if (e) {
          const float2 ee = *reinterpret_cast<const float2 *>(e + sz * (size_t)k + go);
          ex = ee.x;
          ey = ee.y;
        }
      }
    };
    double pmx, pmy, pcx, pcy, ppx, ppy, nnx, nny;
    float ecx, ecy, epx, epy, enx, eny, t0, t1;
    load(zs - 1, pmx, pmy, t0, t1);
    pmx = pmx + (double)t0;
    pmy = pmy + (double)t1;
    load(zs, pcx, pcy, ecx, ecy);
    pcx = pcx + (double)ecx;
    pcy = pcy + (double)ecy;
    load(zs + 1, ppx, ppy, epx, epy);
    ppx = ppx + (double)epx;
    ppy = ppy + (double)epy;
    if (live) {
      lds[(zs & 1) * PLANE + lo] = pcx;
      lds[(zs & 1) * PLANE + lo + 1] = pcy;
    }
    __syncthreads();

    for (int k = zs; k < ze; ++k) {
      load(k + 2, nnx, nny, enx, eny);
      const double *R = lds + (k & 1) * PLANE;
      if (own) {
        const int kg = k + g.k0;
        const bool inz = k >= g.lb[2] && k <= g.ub[2];
        const double xl0 = mir0 ? pcy : R[lo - 1];
        const double xh1 = mir1 ? pcx : R[lo + 2];
        const double vlx = R[oyl], vly = R[oyl + 1], vhx = R[oyh], vhy = R[oyh + 1];
        const double wlx = (kg == 0) ? ppx : pmx, wly = (kg == 0) ? ppy : pmy;
        const double whx = (kg == g.nzg - 1) ? pmx : ppx, why = (kg == g.nzg - 1) ? pmy : ppy;
        double rx = 0.0, ry = 0.0;
        if (rhs) {
          const double2 rr = *reinterpret_cast<const double2 *>(rhs + sz * (size_t)k + go);
          rx = rr.x;
          ry = rr.y;
        }
        // same expression as residual.hip (ndsm_optimized.f90:424-430)
        const double v0 = (xl0 + pcy) * g.w[0] + (vlx + vhx) * g.w[1] + (wlx + whx) * g.w[2] - rx - pcx * g.wc;
        const double v1 = (pcx + xh1) * g.w[0] + (vly + vhy) * g.w[1] + (wly + why) * g.w[2] - ry - pcy * g.wc;
        float2 res;
        res.x = (inz && in0) ? (float)(-v0) : 0.0f;
        res.y = (inz && in1) ? (float)(-v1) : 0.0f;
        *reinterpret_cast<float2 *>(r + sz * (size_t)k + go) = res;
        if (e) {
          double2 un;
          un.x = pcx;
          un.y = pcy;
          *reinterpret_cast<double2 *>(unew + sz * (size_t)k + go) = un;
          float2 z;
          z.x = z.y = 0.0f;
          *reinterpret_cast<float2 *>(ezero + sz * (size_t)k + go) = z;
          const double a0 = fabs((double)ecx), a1 = fabs((double)ecy);
          mx = fmax(mx, fmax(a0, a1));
          sm = sm + a0;
          sm = sm + a1;
        }
      }
      // plane k+1 into the other buffer (its readers finished one barrier ago); shift the window
      if (live) {
        lds[((k + 1) & 1) * PLANE + lo] = ppx;
        lds[((k + 1) & 1) * PLANE + lo + 1] = ppy;
      }
      pmx = pcx;
      pmy = pcy;
      pcx = ppx;
      pcy = ppy;
      ecx = epx;
      ecy = epy;
      ppx = nnx + (double)enx;
      ppy = nny + (double)eny;
      epx = enx;
      epy = eny;
      __syncthreads();
    }
  }
  // ---- (max, sum) of |e| over the points this workgroup updated ----
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmax(mx, __shfl_down(mx, o, 64));
    sm = sm + __shfl_down(sm, o, 64);
  }
  if ((tid & 63) == 0) {
    smx[tid >> 6] = mx;
    ssm[tid >> 6] = sm;
  }
  __syncthreads();
  if (tid == 0) {
    double m = smx[0], s = ssm[0];
    for (int q = 1; q < NT / 64; ++q) {
      m = fmax(m, smx[q]);
      s = s + ssm[q];
    }
    part[2 * blockIdx.x] = m;
    part[2 * blockIdx.x + 1] = s;
  }
}

__global__ __launch_bounds__(256) void fold_k(const double *__restrict__ part, int nblocks, double *__restrict__ out2) {
  __shared__ double smx[4], ssm[4];
  double mx = 0.0, sm = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    mx = fmax(mx, part[2 * i]);
    sm = sm + part[2 * i + 1];
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmax(mx, __shfl_down(mx, o, 64));
    sm = sm + __shfl_down(sm, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    smx[threadIdx.x >> 6] = mx;
    ssm[threadIdx.x >> 6] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out2[0] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    out2[1] = ((ssm[0] + ssm[1]) + ssm[2]) + ssm[3];
  }
}

struct MScratch {
  double *d_part = nullptr;
  size_t cap = 0;  // blocks
  double *h_pin = nullptr;
};
MScratch g_m;
void mscratch_release() {
  if (g_m.d_part) (void)hipFree(g_m.d_part);
  if (g_m.h_pin) (void)hipHostFree(g_m.h_pin);
  g_m = MScratch();
}

constexpr int kTX = 132, kTY = 15, kNT = 1024;

}  // namespace

// unew = u + e ; ezero = 0 ; r = (float)(rhs - L unew) ; h_out2 = (max|e|, sum|e|) [blocking
// read-back].  Out of place on purpose: workgroups read each other's halo of u and e, so neither
// may change under them (the caller swaps u/unew and e/ezero afterwards).
// e == NULL: residual of u only (first cycle), nothing else is written, h_out2 = (0, 0).
// rhs == NULL: zero right-hand side.
extern "C" int ndsmk_update_residual_f32(const ndsmk_grid *gp, const double *u, double *unew, const double *rhs,
                                         const float *e, float *ezero, float *r, double *h_out2) {
  NDSM_REQUIRE_READY();
  const ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 3 && (g.n[0] & 1) == 0 && g.n[0] >= 4 && g.n[1] >= 2 && g.n[2] >= 2);
  // z-slabs: owned planes only; the caller keeps one ghost plane of u and e per neighbour current
  NDSM_CHECK_ARG(g.zown0 >= 0 && g.zown1 <= g.n[2] && g.zown1 > g.zown0);
  const int nzo = g.zown1 - g.zown0;
  NDSM_CHECK_ARG(u && r && (!e || (unew && ezero && unew != u && ezero != e)));
  constexpr int TXI = kTX - 4, TYI = kTY - 2;
  UPlan pl;
  pl.ntx = (g.n[0] + TXI - 1) / TXI;
  pl.nty = (g.n[1] + TYI - 1) / TYI;
  const int tiles = pl.ntx * pl.nty;
  // ~3 workgroups per CU resident (LDS 31 KB, 1024 threads -> 2): chunks of >= 16 planes
  int nzc = (2 * ndsm::cu_count() * 4 + tiles - 1) / tiles;
  if (nzc < 1) nzc = 1;
  int zc = (nzo + nzc - 1) / nzc;
  if (zc < 16) zc = 16 < nzo ? 16 : nzo;
  pl.zc = zc;
  pl.nzc = (nzo + zc - 1) / zc;
  pl.nwork = tiles * pl.nzc;
  const int nblk = ((pl.nwork + 7) / 8) * 8;
  ndsm::at_reset(mscratch_release);
  if ((size_t)nblk > g_m.cap) {
    if (g_m.d_part) (void)hipFree(g_m.d_part);
    g_m.d_part = nullptr;
    NDSM_HIP(hipMalloc((void **)&g_m.d_part, sizeof(double) * (2 * (size_t)nblk + 2)));
    g_m.cap = (size_t)nblk;
  }
  if (!g_m.h_pin) NDSM_HIP(hipHostMalloc((void **)&g_m.h_pin, sizeof(double) * 2, hipHostMallocDefault));
  hipStream_t s = ndsm::stream();
  hipLaunchKernelGGL((update_residual_k<kTX, kTY, kNT>), dim3(nblk), dim3(kNT), 0, s, u, unew, rhs, e, ezero, r, g, pl,
                     g_m.d_part);
  NDSM_LAUNCH_CHECK();
  if (h_out2) {
    double *out = g_m.d_part + 2 * g_m.cap;
    hipLaunchKernelGGL(fold_k, dim3(1), dim3(256), 0, s, g_m.d_part, nblk, out);
    NDSM_LAUNCH_CHECK();
    NDSM_HIP(hipMemcpyAsync(g_m.h_pin, out, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    NDSM_HIP(hipStreamSynchronize(s));
    h_out2[0] = g_m.h_pin[0];
    h_out2[1] = g_m.h_pin[1];
  }
  return 0;
}

// nsweeps fp32 sweeps of L e = r on level 1 (fused kernel only; e / ealt ping-pong).  r_out: the
// residual of the swept e-equation rides on the last sweep (it must: there is no fp32 residual
// kernel besides the pipeline stage).  force: tests on small shapes.
extern "C" int ndsmk_relax_f32(const ndsmk_grid *gp, float *e, float *ealt, const float *r, int nsweeps, int force,
                               float *r_out, int *result_in_alt) {
  NDSM_REQUIRE_READY();
  const ndsmk_grid g = *gp;
  NDSM_CHECK_ARG(g.ndim == 3 && e && ealt && r && result_in_alt && nsweeps >= (r_out ? 1 : 0));
  NDSM_CHECK_ARG(!g.all_neumann);
  *result_in_alt = 0;
  float *cur = e, *oth = ealt;
  int res_done = 0;
  for (int sw = 0; sw < nsweeps;) {
    int ndone = 0, rd = 0;
    int rc = ndsm::launch_rbgs3_fused_f32(g, cur, oth, r, nsweeps - sw, force != 0, &ndone, r_out, &rd);
    if (rc) return rc;
    if (ndone <= 0)
      return ndsm::fail(NDSMK_EARG, "fp32 smoother: the fused kernel does not cover this level", __FILE__, __LINE__);
    res_done |= rd;
    sw += ndone;
    float *t = cur;
    cur = oth;
    oth = t;
  }
  if (r_out && !res_done)
    return ndsm::fail(NDSMK_EARG, "fp32 smoother: no residual stage was run", __FILE__, __LINE__);
  *result_in_alt = (cur != e) ? 1 : 0;
  return 0;
}
