// Inter-level transfer on NDSM's NON-NESTED hierarchy (coarse meshes span the
// same extent with floor(n/2) points, so h_c/h_f = (n_f-1)/(n_c-1) != 2):
//
//   restriction  rhs_c = R r_f : adjoint of N-linear interpolation, 3-6 taps per
//                dimension (ndsm_interp.f90:186-292, driven by
//                ndsm_multigrid_core.f90:1010-1065)
//   prolongation u_f += P u_c  : N-linear interpolation with per-point weights
//                (ndsm_interp.f90:85-158, ndsm_multigrid_core.f90:865-921, :692-712)
//
// The bracket searches and weights are separable, so the host builds 1-D tables
// per dimension ONCE per hierarchy with the reference's exact arithmetic
// (fsrc/ndsmh_grid.f90) and the kernels only gather + multiply.  The products
// and sums are formed in the reference's order (weight = ((((c2x w2x) c2y) w2y)
// c2z) w2z, taps summed x-fastest; interpolation reduced z, then y, then x) so
// both operators are bit-identical to the reference.
//
// Algorithmic traffic: restriction 9 B / fine point, prolong+correct 17 B / fine point.
#include "common.hpp"

namespace ndsm {
int launch_restrict_stream(const ndsmk_xfer *x, const double *r_f, double *rhs_c, double *u_c);
int launch_restrict_stream_f32(const ndsmk_xfer *x, const float *r_f, double *rhs_c, double *u_c);
}

namespace {

struct XferDev {  // by-value kernel argument (pointers are device pointers)
  int nf[3], nc[3], maxt[3];
  const int32_t *plo[3];
  const double *pwl[3], *pwh[3];
  const int32_t *rlo[3], *rcnt[3];
  const double *rw[3];
  double w2[3];
  int f_k0, f_beg, f_cnt, c_k0, c_beg, c_cnt;
};

template <int NDIM>
__global__ __launch_bounds__(256) void restrict_k(const double *__restrict__ f, double *__restrict__ rhs_c,
                                                  double *__restrict__ u_c, XferDev x) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x;
  const int J = blockIdx.y * blockDim.y + threadIdx.y;
  const int Kl = x.c_beg + (int)blockIdx.z;  // local coarse plane
  const int K = Kl + x.c_k0;                  // its global index (the z tables are global)
  if (I >= x.nc[0] || J >= x.nc[1]) return;
  const int i0 = x.rlo[0][I], ni = x.rcnt[0][I];
  const int j0 = x.rlo[1][J], nj = x.rcnt[1][J];
  const double *cx = x.rw[0] + (size_t)I * x.maxt[0];
  const double *cy = x.rw[1] + (size_t)J * x.maxt[1];
  const size_t sy = (size_t)x.nf[0], sz = (size_t)x.nf[0] * (size_t)x.nf[1];
  double fc = 0.0;
  // The tap loops have per-thread trip counts (3-5 per dimension), so the compiler cannot batch
  // their loads and every tap costs a dependent global-memory latency.  With at most RT taps per
  // dimension (true for every mesh ratio >= 2) the weights are fetched up front and the loops
  // are fixed-bound with the taps predicated - same taps, same order, same expressions.
  constexpr int RT = 6;
  if (x.maxt[0] <= RT && x.maxt[1] <= RT && (NDIM == 2 || x.maxt[2] <= RT)) {
    double wx[RT], wy[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      wx[q] = q < ni ? cx[q] : 0.0;
      wy[q] = q < nj ? cy[q] : 0.0;
    }
    if (NDIM == 3) {
      const int k0 = x.rlo[2][K] - x.f_k0, nk = x.rcnt[2][K];  // first tap as a local fine plane
      const double *cz = x.rw[2] + (size_t)K * x.maxt[2];
      for (int kk = 0; kk < nk; ++kk) {
        const double c2z = cz[kk];
#pragma unroll
        for (int jj = 0; jj < RT; ++jj) {
          if (jj < nj) {
            const double *row = f + (size_t)i0 + sy * (size_t)(j0 + jj) + sz * (size_t)(k0 + kk);
            double fr[RT];
#pragma unroll
            for (int ii = 0; ii < RT; ++ii) fr[ii] = ii < ni ? row[ii] : 0.0;
#pragma unroll
            for (int ii = 0; ii < RT; ++ii) {
              if (ii < ni) {
                double w = wx[ii] * x.w2[0];  // 1 * c2 * w2 (ndsm_interp.f90:277-282)
                w = w * wy[jj] * x.w2[1];
                w = w * c2z * x.w2[2];
                fc = fc + w * fr[ii];
              }
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < RT; ++jj) {
        if (jj < nj) {
          const double *row = f + (size_t)i0 + sy * (size_t)(j0 + jj);
          double fr[RT];
#pragma unroll
          for (int ii = 0; ii < RT; ++ii) fr[ii] = ii < ni ? row[ii] : 0.0;
#pragma unroll
          for (int ii = 0; ii < RT; ++ii) {
            if (ii < ni) {
              double w = wx[ii] * x.w2[0];
              w = w * wy[jj] * x.w2[1];
              fc = fc + w * fr[ii];
            }
          }
        }
      }
    }
  } else if (NDIM == 3) {
    const int k0 = x.rlo[2][K] - x.f_k0, nk = x.rcnt[2][K];  // first tap as a local fine plane
    const double *cz = x.rw[2] + (size_t)K * x.maxt[2];
    for (int kk = 0; kk < nk; ++kk) {
      const double c2z = cz[kk];
      for (int jj = 0; jj < nj; ++jj) {
        const double c2y = cy[jj];
        const double *row = f + (size_t)i0 + sy * (size_t)(j0 + jj) + sz * (size_t)(k0 + kk);
        for (int ii = 0; ii < ni; ++ii) {
          double w = cx[ii] * x.w2[0];  // 1 * c2 * w2 (ndsm_interp.f90:277-282)
          w = w * c2y * x.w2[1];
          w = w * c2z * x.w2[2];
          fc = fc + w * row[ii];
        }
      }
    }
  } else {
    for (int jj = 0; jj < nj; ++jj) {
      const double c2y = cy[jj];
      const double *row = f + (size_t)i0 + sy * (size_t)(j0 + jj);
      for (int ii = 0; ii < ni; ++ii) {
        double w = cx[ii] * x.w2[0];
        w = w * c2y * x.w2[1];
        fc = fc + w * row[ii];
      }
    }
  }
  const size_t c = (size_t)I + (size_t)x.nc[0] * ((size_t)J + (size_t)x.nc[1] * (size_t)(NDIM == 3 ? Kl : 0));
  rhs_c[c] = fc;
  if (u_c) u_c[c] = 0.0;  // coarse correction starts from zero (ndsm_multigrid_core.f90:557-558)
}

template <int NDIM>
__global__ __launch_bounds__(256) void prolong_add_k(const double *__restrict__ uc, double *__restrict__ uf,
                                                     XferDev x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = x.f_beg + (int)blockIdx.z;  // local fine plane
  if (i >= x.nf[0] || j >= x.nf[1]) return;
  const int il = x.plo[0][i], jl = x.plo[1][j];
  const double wlx = x.pwl[0][i], whx = x.pwh[0][i];
  const double wly = x.pwl[1][j], why = x.pwh[1][j];
  const size_t sy = (size_t)x.nc[0], sz = (size_t)x.nc[0] * (size_t)x.nc[1];
  double v;
  if (NDIM == 3) {
    const int kgl = k + x.f_k0;               // global fine plane: index into the z tables
    const int kl = x.plo[2][kgl] - x.c_k0;    // lower bracket as a local coarse plane
    const double wlz = x.pwl[2][kgl], whz = x.pwh[2][kgl];
    const double *p = uc + (size_t)il + sy * (size_t)jl + sz * (size_t)kl;
    // fs(n): bit0 = upper x, bit1 = upper y, bit2 = upper z (ndsm_interp.f90:340-363)
    double f0 = p[0], f1 = p[1], f2 = p[sy], f3 = p[sy + 1];
    double f4 = p[sz], f5 = p[sz + 1], f6 = p[sz + sy], f7 = p[sz + sy + 1];
    f0 = whz * f0 + wlz * f4;  // last dimension first (ndsm_interp.f90:128-154)
    f1 = whz * f1 + wlz * f5;
    f2 = whz * f2 + wlz * f6;
    f3 = whz * f3 + wlz * f7;
    f0 = why * f0 + wly * f2;
    f1 = why * f1 + wly * f3;
    v = whx * f0 + wlx * f1;
  } else {
    const double *p = uc + (size_t)il + sy * (size_t)jl;
    double f0 = p[0], f1 = p[1], f2 = p[sy], f3 = p[sy + 1];
    f0 = why * f0 + wly * f2;
    f1 = why * f1 + wly * f3;
    v = whx * f0 + wlx * f1;
  }
  const size_t c = (size_t)i + (size_t)x.nf[0] * ((size_t)j + (size_t)x.nf[1] * (size_t)k);
  uf[c] = uf[c] + v;
}

// Prolong + correct for the large 3-D levels: the 8 coarse values of a fine point
// are shared with its neighbours, so a block stages the coarse footprint of its
// fine tile (64 x 8 x 8 points -> at most 36 x 7 x 7 coarse values) in LDS once and
// interpolates from there - the coarse level is read ~0.3x instead of 8x per fine
// point, and what remains is the 16 B/pt read-modify-write of u.  Same arithmetic
// and order as prolong_add_k.
constexpr int PT_X = 64, PT_Y = 8, PT_Z = 8;
constexpr int PC_X = 36, PC_Y = 7, PC_Z = 7;

// TF: type of the fine field (float: the correction e of the mixed-precision mode; the
// interpolation and the addition are fp64, the sum is rounded once on the way out)
template <typename TF>
__global__ __launch_bounds__(256) void prolong_tile_k(const double *__restrict__ uc, TF *__restrict__ uf,
                                                      XferDev x) {
  __shared__ double C[PC_Z][PC_Y][PC_X];
  const int i0 = blockIdx.x * PT_X, j0 = blockIdx.y * PT_Y;
  const int kl0 = x.f_beg + (int)blockIdx.z * PT_Z;               // local fine planes [kl0, kl1)
  const int kl1 = min(kl0 + PT_Z, x.f_beg + x.f_cnt);
  const int i1 = min(i0 + PT_X, x.nf[0]) - 1, j1 = min(j0 + PT_Y, x.nf[1]) - 1;
  // coarse footprint (global coarse indices; z made local below)
  const int cx0 = x.plo[0][i0], cy0 = x.plo[1][j0], cz0 = x.plo[2][kl0 + x.f_k0];
  const int ncx = x.plo[0][i1] + 2 - cx0, ncy = x.plo[1][j1] + 2 - cy0;
  const int ncz = x.plo[2][kl1 - 1 + x.f_k0] + 2 - cz0;
  const size_t sy = (size_t)x.nc[0], sz = (size_t)x.nc[0] * (size_t)x.nc[1];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int p = tid; p < PC_X * PC_Y * PC_Z; p += 256) {
    const int a = p % PC_X, b = (p / PC_X) % PC_Y, c = p / (PC_X * PC_Y);
    if (a < ncx && b < ncy && c < ncz)
      C[c][b][a] = uc[(size_t)(cx0 + a) + sy * (size_t)(cy0 + b) + sz * (size_t)(cz0 - x.c_k0 + c)];
  }
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i > i1) return;
  const int il = x.plo[0][i] - cx0;
  const double wlx = x.pwl[0][i], whx = x.pwh[0][i];
  for (int jj = threadIdx.y; jj < PT_Y; jj += 4) {
    const int j = j0 + jj;
    if (j > j1) break;
    const int jl = x.plo[1][j] - cy0;
    const double wly = x.pwl[1][j], why = x.pwh[1][j];
    for (int kl = kl0; kl < kl1; ++kl) {
      const int kg = kl + x.f_k0;
      const int kc = x.plo[2][kg] - cz0;
      const double wlz = x.pwl[2][kg], whz = x.pwh[2][kg];
      double f0 = C[kc][jl][il], f1 = C[kc][jl][il + 1], f2 = C[kc][jl + 1][il], f3 = C[kc][jl + 1][il + 1];
      double f4 = C[kc + 1][jl][il], f5 = C[kc + 1][jl][il + 1], f6 = C[kc + 1][jl + 1][il],
             f7 = C[kc + 1][jl + 1][il + 1];
      f0 = whz * f0 + wlz * f4;  // last dimension first (ndsm_interp.f90:128-154)
      f1 = whz * f1 + wlz * f5;
      f2 = whz * f2 + wlz * f6;
      f3 = whz * f3 + wlz * f7;
      f0 = why * f0 + wly * f2;
      f1 = why * f1 + wly * f3;
      const double v = whx * f0 + wlx * f1;
      const size_t c = (size_t)i + (size_t)x.nf[0] * ((size_t)j + (size_t)x.nf[1] * (size_t)kl);
      uf[c] = (TF)((double)uf[c] + v);
    }
  }
}

int to_dev(const ndsmk_xfer *x, XferDev *d, int *ndim) {
  *ndim = (x->nf[2] == 1 && x->nc[2] == 1) ? 2 : 3;
  for (int a = 0; a < 3; ++a) {
    d->nf[a] = x->nf[a];
    d->nc[a] = x->nc[a];
    d->maxt[a] = x->maxt[a];
    d->plo[a] = x->plo[a];
    d->pwl[a] = x->pwl[a];
    d->pwh[a] = x->pwh[a];
    d->rlo[a] = x->rlo[a];
    d->rcnt[a] = x->rcnt[a];
    d->rw[a] = x->rw[a];
    d->w2[a] = x->w2[a];
  }
  d->f_k0 = x->f_k0;
  d->f_beg = x->f_beg;
  d->f_cnt = x->f_cnt;
  d->c_k0 = x->c_k0;
  d->c_beg = x->c_beg;
  d->c_cnt = x->c_cnt;
  if (*ndim == 2) {
    d->f_k0 = d->f_beg = d->c_k0 = d->c_beg = 0;
    d->f_cnt = d->c_cnt = 1;
  }
  NDSM_CHECK_ARG(d->f_cnt >= 1 && d->c_cnt >= 1 && d->f_beg >= 0 && d->c_beg >= 0);
  for (int a = 0; a < *ndim; ++a) {
    NDSM_CHECK_ARG(x->nf[a] >= 2 && x->nc[a] >= 2 && x->maxt[a] >= 1);
    NDSM_CHECK_ARG(x->plo[a] && x->pwl[a] && x->pwh[a] && x->rlo[a] && x->rcnt[a] && x->rw[a]);
  }
  return 0;
}

}  // namespace

extern "C" int ndsmk_restrict(const ndsmk_xfer *x, const double *r_f, double *rhs_c, double *u_c) {
  NDSM_REQUIRE_READY();
  XferDev d;
  int ndim;
  if (int rc = to_dev(x, &d, &ndim)) return rc;
  if (ndim == 3 && (x->stream_ok & 1)) return ndsm::launch_restrict_stream(x, r_f, rhs_c, u_c);
  dim3 block(32, 8, 1);
  dim3 grid((d.nc[0] + 31) / 32, (d.nc[1] + 7) / 8, d.c_cnt);
  if (ndim == 3)
    hipLaunchKernelGGL(restrict_k<3>, grid, block, 0, ndsm::stream(), r_f, rhs_c, u_c, d);
  else
    hipLaunchKernelGGL(restrict_k<2>, grid, block, 0, ndsm::stream(), r_f, rhs_c, u_c, d);
  NDSM_LAUNCH_CHECK();
  return 0;
}

extern "C" int ndsmk_prolong_add(const ndsmk_xfer *x, const double *u_c, double *u_f) {
  NDSM_REQUIRE_READY();
  XferDev d;
  int ndim;
  if (int rc = to_dev(x, &d, &ndim)) return rc;
  dim3 block(64, 4, 1);
  dim3 grid((d.nf[0] + 63) / 64, (d.nf[1] + 3) / 4, d.f_cnt);
  // the tiled kernel assumes a fine tile spans at most PC_* coarse points: true for every
  // ratio (n_f-1)/(n_c-1) <= 2.2, i.e. all but the tiniest levels - which do not need it
  if (ndim == 3 && (int64_t)d.nf[0] * d.nf[1] * d.f_cnt >= (int64_t)1 << 21 && d.nf[0] >= 64 && d.nc[0] >= 16 &&
      d.nc[1] >= 16 && d.nc[2] >= 16) {
    dim3 g2((d.nf[0] + PT_X - 1) / PT_X, (d.nf[1] + PT_Y - 1) / PT_Y, (d.f_cnt + PT_Z - 1) / PT_Z);
    hipLaunchKernelGGL(prolong_tile_k<double>, g2, block, 0, ndsm::stream(), u_c, u_f, d);
    NDSM_LAUNCH_CHECK();
    return 0;
  }
  if (ndim == 3)
    hipLaunchKernelGGL(prolong_add_k<3>, grid, block, 0, ndsm::stream(), u_c, u_f, d);
  else
    hipLaunchKernelGGL(prolong_add_k<2>, grid, block, 0, ndsm::stream(), u_c, u_f, d);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// ---- mixed-precision mode: level-1 side in fp32 (see mixed.hip) -------------------------
// Only the large-level kernels exist in this flavour; the caller checks stream_ok.
extern "C" int ndsmk_restrict_f32(const ndsmk_xfer *x, const float *r_f, double *rhs_c, double *u_c) {
  NDSM_REQUIRE_READY();
  XferDev d;
  int ndim;
  if (int rc = to_dev(x, &d, &ndim)) return rc;
  if (ndim != 3 || !(x->stream_ok & 2))
    return ndsm::fail(NDSMK_EARG, "fp32 restriction: the streamed kernel does not cover this level pair", __FILE__, __LINE__);
  return ndsm::launch_restrict_stream_f32(x, r_f, rhs_c, u_c);
}

extern "C" int ndsmk_prolong_add_f32(const ndsmk_xfer *x, const double *u_c, float *e_f) {
  NDSM_REQUIRE_READY();
  XferDev d;
  int ndim;
  if (int rc = to_dev(x, &d, &ndim)) return rc;
  if (!(ndim == 3 && d.nf[0] >= 64 && d.nc[0] >= 16 && d.nc[1] >= 16 && d.nc[2] >= 16))
    return ndsm::fail(NDSMK_EARG, "fp32 prolongation: the tiled kernel does not cover this level pair", __FILE__, __LINE__);
  dim3 block(64, 4, 1);
  dim3 g2((d.nf[0] + PT_X - 1) / PT_X, (d.nf[1] + PT_Y - 1) / PT_Y, (d.f_cnt + PT_Z - 1) / PT_Z);
  hipLaunchKernelGGL(prolong_tile_k<float>, g2, block, 0, ndsm::stream(), u_c, e_f, d);
  NDSM_LAUNCH_CHECK();
  return 0;
}
