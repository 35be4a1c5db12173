// The bottom of a V-cycle - every level of at most a few thousand points, down to the coarsest grid
// and back up - as ONE launch of ONE workgroup with all of those levels resident in LDS.
//
// What it replaces (ndsm_multigrid_core.f90: fine_to_coarse :482-560, solve_exact :728-800,
// coarse_to_fine :593-684), for tail levels q = 0 .. nlev-1 (q = 0 the finest of them, entered with
// the right-hand side the level above restricted into it):
//
//     for q = 0 .. nlev-2:  ms sweeps;  r = rhs - L u;  rhs(q+1) = R r;  u(q+1) = 0
//     coarsest:             solve_exact;  ms sweeps
//     for q = nlev-1 .. 1:  u(q-1) += P u(q);  2 ms sweeps      (ms after + ms before the next
//                                                                 interpolation: ndsmh_mg.f90, ascent)
//
// Driven level by level this is ~25 launches of 3-17 us for the 16^3 / 8^3 / 4^3 levels of a 512^3
// hierarchy (~120 us of a 5.6 ms cycle), every one of them dispatch latency.  Here: 67 us (measured per phase
// with NDSM_TAIL_DBG: 16^3 sweeps 0.5-0.65 us per colour pass, 8^3 / 4^3 0.2-0.3 us - the barrier and the
// LDS round trip of one wave - restrictions 7-8 us each (a single wave's 64 dependent taps per coarse
// point), the coarsest-grid solve 1.5 us per sweep with its stop test).  One CU is what one workgroup gets:
// the phases are bound by instruction latency, not by anything a wider launch could use at these sizes.
//
// Same expressions, operand order and loop order as the kernels it stands in for - rbgs3_small /
// rbgs3_color (smooth.hip), residual3 (residual.hip), restrict_k<3> / prolong_add_k<3> (transfer.hip),
// solve_exact_k (coarse.hip) - and -ffp-contract=off: bit-identical to the level-by-level path
// (tests/test_gpu_parity.py::test_tail_cycle_bitwise and every V-cycle test: the default path runs it).
// The order-dependent sums - solve_exact's MEAN metric and, on the all-Neumann 2-D hierarchies of the face
// solves, the mean that is subtracted after every sweep - are formed exactly as the kernels they stand in for
// form them: solve_exact_k's by the first 256 threads, strided by 256, four wave partials; rbgs2_small's by all
// 1024 threads, strided by 1024, sixteen wave partials in order.
// 2-D hierarchies (ndim = 2) take the same path with the five-point expressions of rbgs2_small / residual2 /
// restrict_k<2> / prolong_add_k<2>: a 512^2 face has five such levels (64^2 ... 4^2), a 128^2 face all but one.
#include "common.hpp"

#include <cstdlib>

namespace {

constexpr int kT = 1024;       // threads
constexpr int kQ = 3;          // points per thread and colour: up to 3072 updates per colour pass
constexpr int kMaxLev = 6;
constexpr int kMaxTop = 6144;  // points of the finest tail level
constexpr int kMaxExact = 2048;  // points of the coarsest grid (solve_exact_k's limit: same coverage)
constexpr int kRT = 6;         // restriction taps per dimension (any mesh ratio >= 2)
constexpr int kXT = 256;       // threads that carry solve_exact's loops (solve_exact_k's block size)

struct TailXfer {  // the tables of one transfer: one packed device allocation (ndsmh_mg.f90:upload_xfer), copied
  int maxt[3];     // to LDS verbatim when the kernel starts; o_*: byte offsets of the tables from the LDS base
  const double *blob;
  int blob_dbl, toff;   // its length and its place in LDS, in doubles
  int o_plo[3], o_pwl[3], o_pwh[3], o_rlo[3], o_rcnt[3], o_rw[3];
  double w2[3];
};

struct TailArgs {
  int nlev, ms, use_max, nmax;
  double ex_tol;
  ndsmk_grid g[kMaxLev];
  TailXfer x[kMaxLev - 1];
  double *u[kMaxLev], *rhs[kMaxLev];
  int off_u[kMaxLev], off_rhs[kMaxLev], off_r, off_sav;  // LDS offsets in doubles
  long long *info;
  long long *dbg;
};

// the points one thread updates in one colour pass of a level: LDS byte addresses of the centre (-1: none) and of
// its six neighbours (mirrored at Neumann faces), and the point's right-hand side (constant while a level is swept)
struct PtSet {
  int c[kQ];
  int xl[kQ], xh[kQ], yl[kQ], yh[kQ], zl[kQ], zh[kQ];
  double rv[kQ];
};

__device__ __forceinline__ void make_set(const ndsmk_grid &g, int par, int u_off, const double *rhs, PtSet &s) {
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int mx = g.ub[0] - g.lb[0] + 1, my = g.ub[1] - g.lb[1] + 1, mz = g.ub[2] - g.lb[2] + 1;
  const int half = (mx + 1) / 2;
  const int total = (mx > 0 && my > 0 && mz > 0) ? half * my * mz : 0;
  const int sy = 8 * nx, sz = 8 * nx * ny;
#pragma unroll
  for (int q = 0; q < kQ; ++q) {
    const int p = (int)threadIdx.x + kT * q;
    s.c[q] = -1;
    s.xl[q] = s.xh[q] = s.yl[q] = s.yh[q] = s.zl[q] = s.zh[q] = 0;
    s.rv[q] = 0.0;
    if (p < total) {
      const int t = p % half, j = g.lb[1] + (p / half) % my, k = g.lb[2] + p / (half * my);
      const int i0 = g.lb[0] + ((((g.lb[0] + j + k) & 1) != par) ? 1 : 0);
      const int i = i0 + 2 * t;
      if (i <= g.ub[0]) {
        const int e = i + nx * (j + ny * k);
        const int c = 8 * (u_off + e);
        s.c[q] = c;
        s.xl[q] = i - 1 < 0 ? c + 8 : c - 8;
        s.xh[q] = i + 1 > nx - 1 ? c - 8 : c + 8;
        s.yl[q] = j - 1 < 0 ? c + sy : c - sy;
        s.yh[q] = j + 1 > ny - 1 ? c - sy : c + sy;
        s.zl[q] = k - 1 < 0 ? c + sz : c - sz;
        s.zh[q] = k + 1 > nz - 1 ? c - sz : c + sz;
        s.rv[q] = rhs[e];
      }
    }
  }
}

// one colour pass (ndsm_optimized.f90:123-129; the expression of rbgs3_small / rbgs3_color)
// WAVE: the level lives in ONE wave (exact's single-wave form): the LDS operations of a wave execute in program
// order, so what a lane wrote is there for the lane that reads it next - no workgroup barrier
template <bool WAVE = false>
__device__ __forceinline__ void colour_pass(char *ldsb, const ndsmk_grid &g, const PtSet &s) {
#define LD(off) (*reinterpret_cast<const double *>(ldsb + (off)))
#pragma unroll
  for (int q = 0; q < kQ; ++q) {
    if (s.c[q] >= 0) {
      const double unew = (LD(s.xh[q]) + LD(s.xl[q])) * g.w[0] + (LD(s.yh[q]) + LD(s.yl[q])) * g.w[1] +
                          (LD(s.zh[q]) + LD(s.zl[q])) * g.w[2] - s.rv[q];
      *reinterpret_cast<double *>(ldsb + s.c[q]) = g.w1 * unew;
    }
  }
#undef LD
  if (WAVE)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  else
    __syncthreads();
}

// 2-D (ndsm_poisson.f90:603-617; the expression of rbgs2_small / rbgs2_color - the point sets are the 3-D ones
// with one plane: stencil_stride's collapsed boundary neighbours are the mirrored ones)
template <bool WAVE = false>
__device__ __forceinline__ void colour_pass2(char *ldsb, const ndsmk_grid &g, const PtSet &s) {
#define LD(off) (*reinterpret_cast<const double *>(ldsb + (off)))
#pragma unroll
  for (int q = 0; q < kQ; ++q) {
    if (s.c[q] >= 0) {
      double un = 0.0;
      un = un + LD(s.xl[q]) * g.w[0] + LD(s.xh[q]) * g.w[0];
      un = un + LD(s.yl[q]) * g.w[1] + LD(s.yh[q]) * g.w[1];
      *reinterpret_cast<double *>(ldsb + s.c[q]) = (un - s.rv[q]) * g.w1;
    }
  }
#undef LD
  if (WAVE)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  else
    __syncthreads();
}

// u -= mean(u) after a sweep of an all-Neumann problem (ndsm_poisson.f90:534-547), summed as rbgs2_small /
// rbgs2_medium sum it: per thread strided by the workgroup, a shuffle tree per wave, the 16 wave partials in order
__device__ __forceinline__ void mean_shift(double *u, int n, double *red16) {
  double sm = 0.0;
  for (int p = (int)threadIdx.x; p < n; p += kT) sm = sm + u[p];
  for (int o = 32; o > 0; o >>= 1) sm = sm + __shfl_down(sm, o, 64);
  if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = sm;
  __syncthreads();
  double tot = 0.0;
  for (int q = 0; q < kT / 64; ++q) tot = tot + red16[q];
  const double mean = tot / (double)n;
  for (int p = (int)threadIdx.x; p < n; p += kT) u[p] = u[p] - mean;
  __syncthreads();
}

// nsweeps sweeps of the level whose u / rhs start at doubles u_off / rhs_off of the LDS array
__device__ __forceinline__ void relax(double *lds, int u_off, int rhs_off, const ndsmk_grid &g, int nsweeps, double *red16) {
  PtSet a, b;
  make_set(g, g.first_par & 1, u_off, lds + rhs_off, a);
  make_set(g, (g.first_par + 1) & 1, u_off, lds + rhs_off, b);
  char *const ldsb = reinterpret_cast<char *>(lds);
  if (g.n[0] * g.n[1] * g.n[2] <= 64) {
    // a level of at most 64 points (8 x 8, 4 x 4 (x 4)) lives in ONE wave: the same sweeps and - for the mean of an
    // all-Neumann 2-D level - the same sum in the same order (mean_shift's partials of waves 1..15 are exact zeros
    // and are added as such), without the two to four workgroup barriers per sweep
    const int n = g.n[0] * g.n[1] * g.n[2];
    if (threadIdx.x < 64) {
      double *u = lds + u_off;
      const bool mine = (int)threadIdx.x < n;
      for (int sw = 0; sw < nsweeps; ++sw) {
        if (g.ndim == 3) {
          colour_pass<true>(ldsb, g, a);
          colour_pass<true>(ldsb, g, b);
        } else {
          colour_pass2<true>(ldsb, g, a);
          colour_pass2<true>(ldsb, g, b);
          if (g.all_neumann) {
            double sm = mine ? 0.0 + u[threadIdx.x] : 0.0;
            for (int o = 32; o > 0; o >>= 1) sm = sm + __shfl_down(sm, o, 64);
            sm = __shfl(sm, 0, 64);
            double tot = 0.0;
            tot = tot + sm;
#pragma unroll
            for (int q = 1; q < kT / 64; ++q) tot = tot + 0.0;
            const double mean = tot / (double)n;
            if (mine) u[threadIdx.x] = u[threadIdx.x] - mean;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          }
        }
      }
    }
    __syncthreads();
    return;
  }
  if (g.ndim == 3) {
    for (int sw = 0; sw < nsweeps; ++sw) {
      colour_pass(ldsb, g, a);
      colour_pass(ldsb, g, b);
    }
  } else {
    const int n = g.n[0] * g.n[1];
    for (int sw = 0; sw < nsweeps; ++sw) {
      colour_pass2(ldsb, g, a);
      colour_pass2(ldsb, g, b);
      if (g.all_neumann) mean_shift(lds + u_off, n, red16);
    }
  }
}

// r = rhs - L u, zero outside the update bounds (residual3)
__device__ __forceinline__ void residual(const double *u, const double *rhs, double *r, const ndsmk_grid &g) {
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int n = nx * ny * nz;
  const int sy = nx, sz = nx * ny;
  if (g.ndim == 2) {  // residual2 (ndsm_poisson.f90:280-353): rhs - sum_d (u_lo - 2u + u_hi) w_d
    for (int c = (int)threadIdx.x; c < n; c += kT) {
      const int i = c % nx, j = c / nx;
      const bool inside = i >= g.lb[0] && i <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1];
      double out = 0.0;
      if (inside) {
        const double uc = u[c];
        const double xl = u[i == 0 ? c + 1 : (i == nx - 1 ? c - 1 : c - 1)];
        const double xh = u[i == 0 ? c + 1 : (i == nx - 1 ? c - 1 : c + 1)];
        const double yl = u[j == 0 ? c + sy : (j == ny - 1 ? c - sy : c - sy)];
        const double yh = u[j == 0 ? c + sy : (j == ny - 1 ? c - sy : c + sy)];
        double lap = 0.0;  // ndsm_poisson.f90:334-345
        lap = lap + (xl - 2 * uc + xh) * g.w[0];
        lap = lap + (yl - 2 * uc + yh) * g.w[1];
        out = rhs[c] - lap;
      }
      r[c] = out;
    }
    __syncthreads();
    return;
  }
  for (int c = (int)threadIdx.x; c < n; c += kT) {
    const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
    const bool inside = i >= g.lb[0] && i <= g.ub[0] && j >= g.lb[1] && j <= g.ub[1] && k >= g.lb[2] && k <= g.ub[2];
    double out = 0.0;
    if (inside) {
      const double ul = u[i == 0 ? c + 1 : c - 1];
      const double uh = u[i == nx - 1 ? c - 1 : c + 1];
      const double vl = u[j == 0 ? c + sy : c - sy];
      const double vh = u[j == ny - 1 ? c - sy : c + sy];
      const double wl = u[k == 0 ? c + sz : c - sz];
      const double wh = u[k == nz - 1 ? c - sz : c + sz];
      const double v = (ul + uh) * g.w[0] + (vl + vh) * g.w[1] + (wl + wh) * g.w[2] - rhs[c] - u[c] * g.wc;
      out = -v;
    }
    r[c] = out;
  }
  __syncthreads();
}

// the largest value of v over the lanes of the wave, as a wave-uniform number (lanes without a point pass 0)
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

// rhs_c = R r_f, u_c = 0 (restrict_k<3>: weight chain ((((c2x w2x) c2y) w2y) c2z) w2z, taps summed x fastest)
#define TAB(T, off) reinterpret_cast<const T *>(ldsb + (off))
__device__ __forceinline__ void restrict_to(const double *f, double *rhs_c, double *u_c, const ndsmk_grid &gf,
                                            const ndsmk_grid &gc, const TailXfer &x, const char *ldsb) {
  const int ncx = gc.n[0], ncy = gc.n[1], ncz = gc.n[2];
  const int nc = ncx * ncy * ncz;
  const int sy = gf.n[0], sz = gf.n[0] * gf.n[1];
  if (gf.ndim == 2) {  // restrict_k<2>
    // (every wave runs the same number of rounds so that wave_max sees whole waves; the tap loops stop at the
    // wave's largest tap counts - typically 4 or 5 of the kRT = 6 slots per dimension - and stay predicated per lane)
    for (int c0 = 0; c0 < nc; c0 += kT) {
      const int c = c0 + (int)threadIdx.x;
      const bool have = c < nc;
      const int I = have ? c % ncx : 0, J = have ? c / ncx : 0;
      const int i0 = TAB(int32_t, x.o_rlo[0])[I], ni = have ? TAB(int32_t, x.o_rcnt[0])[I] : 0;
      const int j0 = TAB(int32_t, x.o_rlo[1])[J], nj = have ? TAB(int32_t, x.o_rcnt[1])[J] : 0;
      const double *cx = TAB(double, x.o_rw[0]) + I * x.maxt[0];
      const double *cy = TAB(double, x.o_rw[1]) + J * x.maxt[1];
      const int nim = wave_max(ni), njm = wave_max(nj);
      double wx[kRT], wy[kRT];
#pragma unroll
      for (int q = 0; q < kRT; ++q) {
        wx[q] = q < ni ? cx[q] : 0.0;
        wy[q] = q < nj ? cy[q] : 0.0;
      }
      double fc = 0.0;
#pragma unroll
      for (int jj = 0; jj < kRT; ++jj) {
        if (jj < njm) {
          const double *row = f + i0 + sy * (j0 + jj);
#pragma unroll
          for (int ii = 0; ii < kRT; ++ii) {
            if (ii < nim) {
              if (jj < nj && ii < ni) {
                double w = wx[ii] * x.w2[0];
                w = w * wy[jj] * x.w2[1];
                fc = fc + w * row[ii];
              }
            }
          }
        }
      }
      if (have) {
        rhs_c[c] = fc;
        u_c[c] = 0.0;
      }
    }
    __syncthreads();
    return;
  }
  for (int c0 = 0; c0 < nc; c0 += kT) {
    const int c = c0 + (int)threadIdx.x;
    const bool have = c < nc;
    const int I = have ? c % ncx : 0, J = have ? (c / ncx) % ncy : 0, K = have ? c / (ncx * ncy) : 0;
    const int i0 = TAB(int32_t, x.o_rlo[0])[I], ni = have ? TAB(int32_t, x.o_rcnt[0])[I] : 0;
    const int j0 = TAB(int32_t, x.o_rlo[1])[J], nj = have ? TAB(int32_t, x.o_rcnt[1])[J] : 0;
    const int k0 = TAB(int32_t, x.o_rlo[2])[K], nk = have ? TAB(int32_t, x.o_rcnt[2])[K] : 0;
    const double *cx = TAB(double, x.o_rw[0]) + I * x.maxt[0];
    const double *cy = TAB(double, x.o_rw[1]) + J * x.maxt[1];
    const double *cz = TAB(double, x.o_rw[2]) + K * x.maxt[2];
    // Tap loops with the taps predicated per lane and the weights fetched up front (as restrict_k) - measured faster
    // than per-lane trip counts (8.4 against 9.9 us for 16^3 -> 8^3) - but (round 3) stopped at the WAVE's largest
    // tap counts, wave-uniform numbers: 4 or 5 of the kRT = 6 slots per dimension instead of all 216 predicated
    // rounds (the few waves at work here wait for the length of their own instruction stream)
    const int nim = wave_max(ni), njm = wave_max(nj), nkm = wave_max(nk);
    double wx[kRT], wy[kRT], wz[kRT];
#pragma unroll
    for (int q = 0; q < kRT; ++q) {
      wx[q] = q < ni ? cx[q] : 0.0;
      wy[q] = q < nj ? cy[q] : 0.0;
      wz[q] = q < nk ? cz[q] : 0.0;
    }
    double fc = 0.0;
#pragma unroll
    for (int kk = 0; kk < kRT; ++kk) {
      if (kk < nkm) {
        const double c2z = wz[kk];
#pragma unroll
        for (int jj = 0; jj < kRT; ++jj) {
          if (jj < njm) {
            const double *row = f + i0 + sy * (j0 + jj) + sz * (k0 + kk);
#pragma unroll
            for (int ii = 0; ii < kRT; ++ii) {
              if (ii < nim) {
                if (kk < nk && jj < nj && ii < ni) {
                  double w = wx[ii] * x.w2[0];  // 1 * c2 * w2 (ndsm_interp.f90:277-282)
                  w = w * wy[jj] * x.w2[1];
                  w = w * c2z * x.w2[2];
                  fc = fc + w * row[ii];
                }
              }
            }
          }
        }
      }
    }
    if (have) {
      rhs_c[c] = fc;
      u_c[c] = 0.0;  // ndsm_multigrid_core.f90:557-558
    }
  }
  __syncthreads();
}

// u_f += P u_c (prolong_add_k<3>: z first, then y, then x)
__device__ __forceinline__ void prolong_add(const double *uc, double *uf, const ndsmk_grid &gf, const ndsmk_grid &gc,
                                            const TailXfer &x, const char *ldsb) {
  const int nx = gf.n[0], ny = gf.n[1];
  const int n = nx * ny * gf.n[2];
  const int sy = gc.n[0], sz = gc.n[0] * gc.n[1];
  if (gf.ndim == 2) {  // prolong_add_k<2>
    for (int c = (int)threadIdx.x; c < n; c += kT) {
      const int i = c % nx, j = c / nx;
      const int il = TAB(int32_t, x.o_plo[0])[i], jl = TAB(int32_t, x.o_plo[1])[j];
      const double wlx = TAB(double, x.o_pwl[0])[i], whx = TAB(double, x.o_pwh[0])[i];
      const double wly = TAB(double, x.o_pwl[1])[j], why = TAB(double, x.o_pwh[1])[j];
      const double *p = uc + il + sy * jl;
      double f0 = p[0], f1 = p[1], f2 = p[sy], f3 = p[sy + 1];
      f0 = why * f0 + wly * f2;
      f1 = why * f1 + wly * f3;
      const double v = whx * f0 + wlx * f1;
      uf[c] = uf[c] + v;
    }
    __syncthreads();
    return;
  }
  for (int c = (int)threadIdx.x; c < n; c += kT) {
    const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
    const int il = TAB(int32_t, x.o_plo[0])[i], jl = TAB(int32_t, x.o_plo[1])[j], kl = TAB(int32_t, x.o_plo[2])[k];
    const double wlx = TAB(double, x.o_pwl[0])[i], whx = TAB(double, x.o_pwh[0])[i];
    const double wly = TAB(double, x.o_pwl[1])[j], why = TAB(double, x.o_pwh[1])[j];
    const double wlz = TAB(double, x.o_pwl[2])[k], whz = TAB(double, x.o_pwh[2])[k];
    const double *p = uc + il + sy * jl + sz * kl;
    double f0 = p[0], f1 = p[1], f2 = p[sy], f3 = p[sy + 1];
    double f4 = p[sz], f5 = p[sz + 1], f6 = p[sz + sy], f7 = p[sz + sy + 1];
    f0 = whz * f0 + wlz * f4;  // last dimension first (ndsm_interp.f90:128-154)
    f1 = whz * f1 + wlz * f5;
    f2 = whz * f2 + wlz * f6;
    f3 = whz * f3 + wlz * f7;
    f0 = why * f0 + wly * f2;
    f1 = why * f1 + wly * f3;
    const double v = whx * f0 + wlx * f1;
    uf[c] = uf[c] + v;
  }
  __syncthreads();
}

// solve_exact_k's reductions: the partials of its four waves (threads 0..255), combined in its order
// (both at once behind ONE barrier: sh is not written again before the barriers of the next sweep's colour passes)
__device__ __forceinline__ void xmaxsum(double &mx, double &sm, double *sh) {
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmax(mx, __shfl_down(mx, o, 64));
    sm = sm + __shfl_down(sm, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && threadIdx.x < kXT) {
    sh[threadIdx.x >> 6] = mx;
    sh[4 + (threadIdx.x >> 6)] = sm;
  }
  __syncthreads();
  mx = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  sm = ((sh[4] + sh[5]) + sh[6]) + sh[7];
}

// solve_exact (ndsm_multigrid_core.f90:728-800) as solve_exact_k runs it; returns the sweep count, *conv
__device__ __forceinline__ int exact(double *lds, int u_off, int rhs_off, double *sav, const ndsmk_grid &g, double ex_tol,
                                     int use_max, int nmax, double *red, int *conv) {
  const double *u = lds + u_off;
  char *const ldsb = reinterpret_cast<char *>(lds);
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int n = nx * ny * nz;
  const bool act = threadIdx.x < kXT;
  for (int p = (int)threadIdx.x; p < n; p += kT) sav[p] = 0.0;
  __syncthreads();
  PtSet a, b;
  make_set(g, g.first_par & 1, u_off, lds + rhs_off, a);
  make_set(g, (g.first_par + 1) & 1, u_off, lds + rhs_off, b);
  double du = 1.79769313486231570815e308;
  int sweeps = 0;
  *conv = 0;
  if (n <= 64) {
    // The coarsest grid of NDSM's hierarchies - 4 x 4 (x 4) points - fits ONE wave: the same sweeps, the same sums in
    // the same order (solve_exact_k's partials of waves 1..3 are exact zeros here and are added as such), but no
    // workgroup barrier inside the loop - five (three in 3-D) barriers of ~0.45 us per sweep were the whole cost
    // of it: 2.3 -> 0.3 us per sweep (a 128^2 face cycle spends 17 sweeps here).  The other 15 waves wait at the end.
    if (threadIdx.x < 64) {
      double *uw = lds + u_off;
      const bool mine = (int)threadIdx.x < n;
      for (int it = 0; it < nmax; ++it) {
        if (du <= ex_tol) {
          *conv = 1;
          break;
        }
        if (g.ndim == 3) {
          colour_pass<true>(ldsb, g, a);
          colour_pass<true>(ldsb, g, b);
        } else {
          colour_pass2<true>(ldsb, g, a);
          colour_pass2<true>(ldsb, g, b);
        }
        if (g.all_neumann) {
          double sm1 = mine ? 0.0 + uw[threadIdx.x] : 0.0;
          for (int o = 32; o > 0; o >>= 1) sm1 = sm1 + __shfl_down(sm1, o, 64);
          sm1 = __shfl(sm1, 0, 64);
          const double mean = (((sm1 + 0.0) + 0.0) + 0.0) / (double)n;
          if (mine) uw[threadIdx.x] = uw[threadIdx.x] - mean;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        double mx = 0.0, sm = 0.0;
        if (mine) {
          const double d = fabs(sav[threadIdx.x] - uw[threadIdx.x]);
          mx = fmax(mx, d);
          sm = sm + d;
          sav[threadIdx.x] = uw[threadIdx.x];
        }
        for (int o = 32; o > 0; o >>= 1) {
          mx = fmax(mx, __shfl_down(mx, o, 64));
          sm = sm + __shfl_down(sm, o, 64);
        }
        mx = __shfl(mx, 0, 64);
        sm = __shfl(sm, 0, 64);
        mx = fmax(fmax(mx, 0.0), fmax(0.0, 0.0));
        sm = ((sm + 0.0) + 0.0) + 0.0;
        du = use_max ? mx : sm / (double)n;
        ++sweeps;
      }
      if (threadIdx.x == 0) {
        red[8] = (double)sweeps;
        red[9] = (double)*conv;
      }
    }
    __syncthreads();
    sweeps = (int)red[8];
    *conv = (int)red[9];
    __syncthreads();   // (red is written again by the sweeps that follow)
    return sweeps;
  }
  for (int it = 0; it < nmax; ++it) {
    if (du <= ex_tol) {  // uniform: every thread holds the same du
      *conv = 1;
      break;
    }
    if (g.ndim == 3) {
      colour_pass(ldsb, g, a);
      colour_pass(ldsb, g, b);
    } else {
      colour_pass2(ldsb, g, a);
      colour_pass2(ldsb, g, b);
    }
    if (g.all_neumann) {  // solve_exact_k: s strided by its 256 threads, block_sum of four waves
      double s = 0.0, dummy = 0.0;
      if (act)
        for (int p = (int)threadIdx.x; p < n; p += kXT) s = s + u[p];
      xmaxsum(dummy, s, red);
      const double mean = s / (double)n;
      double *uw = lds + u_off;
      if (act)
        for (int p = (int)threadIdx.x; p < n; p += kXT) uw[p] = uw[p] - mean;
      __syncthreads();
    }
    double mx = 0.0, sm = 0.0;
    if (act) {
      for (int p = (int)threadIdx.x; p < n; p += kXT) {
        const double d = fabs(sav[p] - u[p]);
        mx = fmax(mx, d);
        sm = sm + d;
        sav[p] = u[p];
      }
    }
    xmaxsum(mx, sm, red);
    du = use_max ? mx : sm / (double)n;
    ++sweeps;
  }
  return sweeps;
}

__global__ __launch_bounds__(kT) void tail_cycle_k(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[16];
  const int L = a.nlev;
  const char *const ldsb = reinterpret_cast<const char *>(lds);
  for (int q = 0; q + 1 < L; ++q)   // the transfer tables
    for (int p = (int)threadIdx.x; p < a.x[q].blob_dbl; p += kT) lds[a.x[q].toff + p] = a.x[q].blob[p];
  int dbi = 0;
#define TICK() do { if (a.dbg && threadIdx.x == 0) a.dbg[dbi++] = wall_clock64(); } while (0)
  TICK();
  {  // the finest tail level comes in from HBM: u (zeroed by the restriction above it, or whatever the caller left) and rhs
    const int n = a.g[0].n[0] * a.g[0].n[1] * a.g[0].n[2];
    for (int p = (int)threadIdx.x; p < n; p += kT) {
      lds[a.off_u[0] + p] = a.u[0][p];
      lds[a.off_rhs[0] + p] = a.rhs[0][p];
    }
    __syncthreads();
  }
  for (int q = 0; q + 1 < L; ++q) {
    TICK();
    relax(lds, a.off_u[q], a.off_rhs[q], a.g[q], a.ms, red);
    TICK();
    residual(lds + a.off_u[q], lds + a.off_rhs[q], lds + a.off_r, a.g[q]);
    TICK();
    restrict_to(lds + a.off_r, lds + a.off_rhs[q + 1], lds + a.off_u[q + 1], a.g[q], a.g[q + 1], a.x[q], ldsb);
  }
  TICK();
  int conv = 0;
  const int sweeps = exact(lds, a.off_u[L - 1], a.off_rhs[L - 1], lds + a.off_sav, a.g[L - 1], a.ex_tol, a.use_max,
                           a.nmax, red, &conv);
  TICK();
  relax(lds, a.off_u[L - 1], a.off_rhs[L - 1], a.g[L - 1], a.ms, red);
  for (int q = L - 1; q >= 1; --q) {
    TICK();
    prolong_add(lds + a.off_u[q], lds + a.off_u[q - 1], a.g[q - 1], a.g[q], a.x[q - 1], ldsb);
    TICK();
    relax(lds, a.off_u[q - 1], a.off_rhs[q - 1], a.g[q - 1], 2 * a.ms, red);
  }
  TICK();
  // every level goes home (the coarser ones are only ever looked at by tests and tools)
  for (int q = 0; q < L; ++q) {
    const int n = a.g[q].n[0] * a.g[q].n[1] * a.g[q].n[2];
    for (int p = (int)threadIdx.x; p < n; p += kT) {
      a.u[q][p] = lds[a.off_u[q] + p];
      if (q > 0) a.rhs[q][p] = lds[a.off_rhs[q] + p];
    }
  }
  TICK();
  if (threadIdx.x == 0) {
    if (a.dbg) a.dbg[31] = sweeps;
    a.info[0] += sweeps;
    a.info[1] += conv ? 0 : 1;
  }
}

bool tail_enabled() {
  static const bool on = std::getenv("NDSM_HIP_NO_TAIL") == nullptr;
  return on;
}

int g_tail_off = 0;  // ndsmk_debug_tail(0) switches the launch off at run time (tests: A/B in one process)

bool whole(const ndsmk_grid &g) { return g.k0 == 0 && g.nzg == g.n[2] && g.zown0 == 0 && g.zown1 == g.n[2]; }

// the packed tables of one transfer: first byte and length (every table starts on an 8-byte boundary)
void blob_of(const ndsmk_xfer &x, const char **base, size_t *len) {
  const char *lo = nullptr, *hi = nullptr;
  auto span = [&](const void *p, size_t n) {
    const char *b = static_cast<const char *>(p);
    const char *e = b + ((n + 7) & ~(size_t)7);
    if (!lo || b < lo) lo = b;
    if (!hi || e > hi) hi = e;
  };
  for (int d = 0; d < 3; ++d) {
    if (!x.plo[d]) continue;   // (2-D: no z tables)
    span(x.plo[d], sizeof(int32_t) * (size_t)x.nf[d]);
    span(x.pwl[d], sizeof(double) * (size_t)x.nf[d]);
    span(x.pwh[d], sizeof(double) * (size_t)x.nf[d]);
    span(x.rlo[d], sizeof(int32_t) * (size_t)x.nc[d]);
    span(x.rcnt[d], sizeof(int32_t) * (size_t)x.nc[d]);
    span(x.rw[d], sizeof(double) * (size_t)x.nc[d] * (size_t)x.maxt[d]);
  }
  *base = lo;
  *len = (size_t)(hi - lo);
}

// LDS plan; false if the levels do not fit
bool plan(int nlev, const ndsmk_grid *g, const ndsmk_xfer *x, TailArgs *a, size_t *bytes) {
  int off = 0;
  for (int q = 0; q + 1 < nlev; ++q) {
    const char *base;
    size_t len;
    blob_of(x[q], &base, &len);
    if ((reinterpret_cast<uintptr_t>(base) & 7) != 0 || len > (size_t)64 * 1024) return false;
    TailXfer &t = a->x[q];
    t.blob = reinterpret_cast<const double *>(base);
    t.blob_dbl = (int)(len / 8);
    t.toff = off;
    for (int d = 0; d < 3; ++d) {
      auto o = [&](const void *p) { return p ? (int)(8 * (size_t)off + (size_t)(static_cast<const char *>(p) - base)) : 0; };
      t.maxt[d] = x[q].maxt[d];
      t.w2[d] = x[q].w2[d];
      t.o_plo[d] = o(x[q].plo[d]);
      t.o_pwl[d] = o(x[q].pwl[d]);
      t.o_pwh[d] = o(x[q].pwh[d]);
      t.o_rlo[d] = o(x[q].rlo[d]);
      t.o_rcnt[d] = o(x[q].rcnt[d]);
      t.o_rw[d] = o(x[q].rw[d]);
    }
    off += t.blob_dbl;
  }
  for (int q = 0; q < nlev; ++q) {
    const int n = g[q].n[0] * g[q].n[1] * g[q].n[2];
    a->off_u[q] = off;
    off += (n + 1) & ~1;
    a->off_rhs[q] = off;
    off += (n + 1) & ~1;
  }
  const int n0 = g[0].n[0] * g[0].n[1] * g[0].n[2];
  const int nl = g[nlev - 1].n[0] * g[nlev - 1].n[1] * g[nlev - 1].n[2];
  a->off_r = off;
  off += (n0 + 1) & ~1;
  a->off_sav = off;
  off += (nl + 1) & ~1;
  *bytes = sizeof(double) * (size_t)off;
  return *bytes <= (size_t)158 * 1024;
}

bool applies(int nlev, const ndsmk_grid *g, const ndsmk_xfer *x) {
  if (!tail_enabled() || g_tail_off) return false;
  if (nlev < 2 || nlev > kMaxLev) return false;
  const int ndim = g[0].ndim;
  if (ndim != 2 && ndim != 3) return false;
  for (int q = 0; q < nlev; ++q) {
    // (3-D all-Neumann levels shift the mean with the two-stage kernels of reduce.hip: not reproduced here)
    if (g[q].ndim != ndim || (ndim == 3 && g[q].all_neumann) || !whole(g[q])) return false;
    if (ndim == 2 && g[q].n[2] != 1) return false;
    for (int d = 0; d < ndim; ++d)
      if (g[q].n[d] < 2 || g[q].lb[d] < 0 || g[q].ub[d] > g[q].n[d] - 1) return false;
    const int64_t n = (int64_t)g[q].n[0] * g[q].n[1] * g[q].n[2];
    if (n > kMaxTop) return false;
    const int mx = g[q].ub[0] - g[q].lb[0] + 1, my = g[q].ub[1] - g[q].lb[1] + 1, mz = g[q].ub[2] - g[q].lb[2] + 1;
    if (mx > 0 && my > 0 && mz > 0 && (int64_t)((mx + 1) / 2) * my * mz > (int64_t)kQ * kT) return false;
  }
  if ((int64_t)g[nlev - 1].n[0] * g[nlev - 1].n[1] * g[nlev - 1].n[2] > kMaxExact) return false;
  for (int q = 0; q + 1 < nlev; ++q) {
    for (int d = 0; d < ndim; ++d) {
      if (x[q].nf[d] != g[q].n[d] || x[q].nc[d] != g[q + 1].n[d]) return false;
      if (x[q].maxt[d] < 1 || x[q].maxt[d] > kRT) return false;
      if (!x[q].plo[d] || !x[q].pwl[d] || !x[q].pwh[d] || !x[q].rlo[d] || !x[q].rcnt[d] || !x[q].rw[d]) return false;
    }
    if (x[q].f_k0 != 0 || x[q].f_beg != 0 || x[q].f_cnt != x[q].nf[2] || x[q].c_k0 != 0 || x[q].c_beg != 0 ||
        x[q].c_cnt != x[q].nc[2])
      return false;
  }
  TailArgs a;
  size_t bytes;
  return plan(nlev, g, x, &a, &bytes);
}

}  // namespace

extern "C" {

// can levels g[0..nlev-1] (g[nlev-1] the coarsest grid of the hierarchy; x[q]: the transfer g[q] -> g[q+1])
// run as one launch?  1 / 0
int ndsmk_tail_applies(int nlev, const ndsmk_grid *g, const ndsmk_xfer *x) { return applies(nlev, g, x) ? 1 : 0; }

// tests: 0 = the level-by-level path even where the launch applies, 1 = back to the default
int ndsmk_debug_tail(int on) {
  g_tail_off = on ? 0 : 1;
  return 0;
}

// The bottom of one V-cycle (see the header of this file).  u[q], rhs[q]: DEVICE arrays of level q; on entry
// u[0] / rhs[0] hold the finest tail level's iterate (zero after a restriction) and right-hand side, on exit
// every u[q] and rhs[q >= 1] holds what the level-by-level path would have left there.  d_info[0] += sweeps of
// the coarsest-grid solve, d_info[1] += 1 if it did not reach ex_tol in nmax sweeps.
int ndsmk_tail_cycle(int nlev, const ndsmk_grid *g, const ndsmk_xfer *x, double *const *u, double *const *rhs, int ms,
                     double ex_tol, int use_max, int nmax, int64_t *d_info) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(g && x && u && rhs && d_info && ms >= 0 && nmax >= 0);
  if (!applies(nlev, g, x)) return ndsm::fail(NDSMK_EARG, "tail cycle: these levels are not covered", __FILE__, __LINE__);
  TailArgs a;
  size_t bytes = 0;
  plan(nlev, g, x, &a, &bytes);
  a.nlev = nlev;
  a.ms = ms;
  a.use_max = use_max;
  a.nmax = nmax;
  a.ex_tol = ex_tol;
  a.info = reinterpret_cast<long long *>(d_info);
  for (int q = 0; q < nlev; ++q) {
    NDSM_CHECK_ARG(u[q] && rhs[q]);
    a.g[q] = g[q];
    a.u[q] = u[q];
    a.rhs[q] = rhs[q];
  }
  static int attr_epoch = 0;
  if (ndsm::first_in_epoch(attr_epoch))
    NDSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tail_cycle_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 158 * 1024));
  // tuning aid (NDSM_TAIL_DBG=1): the 10th launch of the process prints the duration of its phases - load,
  // then per level sweeps / residual / restriction, coarsest-grid solve, its sweeps, then per level
  // interpolation / sweeps, write-back - from the 100 MHz wall clock
  static const bool want_dbg = std::getenv("NDSM_TAIL_DBG") != nullptr;
  static int dbg_n = 0;
  a.dbg = nullptr;
  if (want_dbg && ++dbg_n == 10) NDSM_HIP(hipMalloc(&a.dbg, 32 * sizeof(long long)));
  if (a.dbg) NDSM_HIP(hipMemsetAsync(a.dbg, 0, 32 * sizeof(long long), ndsm::stream()));
  hipLaunchKernelGGL(tail_cycle_k, dim3(1), dim3(kT), bytes, ndsm::stream(), a);
  NDSM_LAUNCH_CHECK();
  if (a.dbg) {
    long long h[32];
    NDSM_HIP(hipMemcpyAsync(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost, ndsm::stream()));
    NDSM_HIP(hipStreamSynchronize(ndsm::stream()));
    NDSM_HIP(hipFree(a.dbg));
    fprintf(stderr, "tail phases (us):");
    for (int i = 1; i < 31 && h[i] > 0; ++i) fprintf(stderr, " %.2f", (h[i] - h[i - 1]) * 0.01);
    fprintf(stderr, " | coarsest-grid sweeps %lld\n", h[31]);
  }
  return 0;
}

}  // extern "C"
