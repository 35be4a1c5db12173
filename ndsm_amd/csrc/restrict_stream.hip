// Restriction rhs_c = R r_f for the large levels: z-streamed through LDS.
//
// restrict_k (transfer.hip) lets every coarse point gather its 64-125 taps from
// global memory: each fine value is requested ~8 times with a stride-2 lane
// pattern and the kernel runs at ~1/7 of its 9 B/pt roofline.  Here a workgroup
// owns CI x CJ coarse columns and a chunk of coarse planes, streams the fine
// planes of that chunk once (coalesced 16-B loads, double-buffered LDS plane, one
// barrier per plane) and every thread - one coarse column - adds the plane's
// taps to the (at most 3) coarse planes whose z window contains it.
//
// Bit-identical to restrict_k and to the reference (nrestrict, ndsm_interp.f90:
// 263-290): taps are visited in (z, y, x) order and the weight is formed as
// ((((c2x w2x) c2y) w2y) c2z) w2z; the z-independent prefix ((c2x w2x) c2y) w2y is
// the same number for every plane, so it is computed once per thread and kept in
// registers (MT x MT values) - which also takes the multiplications per tap from
// six to three.
#include "common.hpp"

namespace {

struct RSArgs {
  int nf[3], nc[3];
  const int32_t *rlo[3], *rcnt[3];
  const double *rw[3];
  int maxt[3];
  double w2[3];
  int f_k0, c_k0, c_beg, c_cnt;
  int nti, ntj, nkc, kc, nwork;
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

template <int CI, int CJ, int MT>
__global__ __launch_bounds__(CI *CJ) void restrict_stream_k(const double *__restrict__ f, double *__restrict__ rhs_c,
                                                            double *__restrict__ u_c, RSArgs a) {
  constexpr int NT = CI * CJ;
  constexpr int FX = 2 * CI + 6, FY = 2 * CJ + 5;  // fine footprint of the tile (non-nested ratio up to ~2.03)
  constexpr int NPX = FX / 2, NPAIR = NPX * FY, NS = (NPAIR + NT - 1) / NT;
  constexpr int PLANE = FX * FY;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int nb8 = gridDim.x >> 3;
  const int wk = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
  if (wk >= a.nwork) return;
  const int tj = wk % a.ntj;
  const int t2 = wk / a.ntj;
  const int ti = t2 % a.nti;
  const int ck = t2 / a.nti;

  const int nx = a.nf[0], ny = a.nf[1];
  const size_t sz = (size_t)nx * (size_t)ny;
  const int I0 = ti * CI, J0 = tj * CJ;
  // coarse planes of this chunk, GLOBAL numbering (the z tables are global)
  const int Ks = a.c_k0 + a.c_beg + ck * a.kc;
  const int Ke = min(Ks + a.kc, a.c_k0 + a.c_beg + a.c_cnt);
  const int fx0 = a.rlo[0][I0] & ~1;
  const int fy0 = a.rlo[1][J0];
  const int kA = a.rlo[2][Ks];                               // global fine planes [kA, kB]
  const int kB = a.rlo[2][Ke - 1] + a.rcnt[2][Ke - 1] - 1;
  const int tid = (int)threadIdx.x;

  // ---- this thread's coarse column and its z-independent weights ----
  const int I = I0 + tid % CI, J = J0 + tid / CI;
  const bool chave = I < a.nc[0] && J < a.nc[1];
  int ni = 0, nj = 0, li0 = 0, lj0 = 0;
  double wxy[MT][MT];
#pragma unroll
  for (int jj = 0; jj < MT; ++jj)
#pragma unroll
    for (int ii = 0; ii < MT; ++ii) wxy[jj][ii] = 0.0;
  if (chave) {
    ni = a.rcnt[0][I];
    nj = a.rcnt[1][J];
    li0 = a.rlo[0][I] - fx0;
    lj0 = a.rlo[1][J] - fy0;
#pragma unroll
    for (int jj = 0; jj < MT; ++jj) {
      const double c2y = jj < nj ? a.rw[1][(size_t)J * a.maxt[1] + jj] : 0.0;
#pragma unroll
      for (int ii = 0; ii < MT; ++ii) {
        const double c2x = ii < ni ? a.rw[0][(size_t)I * a.maxt[0] + ii] : 0.0;
        double wv = c2x * a.w2[0];  // 1 * c2 * w2 (ndsm_interp.f90:277-282)
        wv = wv * c2y * a.w2[1];
        wxy[jj][ii] = wv;
      }
    }
  }
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;  // coarse planes K with K & 3 = 0..3
  int Klo = Ks;

  d2 nxt[NS], nn[NS];
#define RS_LOAD(kglob, dst)                                                              \
  do {                                                                                   \
    const int kl_ = (kglob) - a.f_k0; /* local fine plane */                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                  \
      const int p_ = tid + NT * s_;                                                      \
      const int lj_ = p_ / NPX, li_ = 2 * (p_ - lj_ * NPX);                              \
      const int i_ = fx0 + li_, j_ = fy0 + lj_;                                          \
      d2 t_;                                                                             \
      t_.x = 0.0;                                                                        \
      t_.y = 0.0;                                                                        \
      if (p_ < NPAIR && i_ + 1 < nx && j_ < ny && (kglob) <= kB) t_ = ld2(f + sz * (size_t)kl_ + (i_ + nx * j_)); \
      dst[s_] = t_;                                                                      \
    }                                                                                    \
  } while (0)
#define RS_STORE(buf, src)                                                               \
  do {                                                                                   \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                  \
      const int p_ = tid + NT * s_;                                                      \
      const int lj_ = p_ / NPX, li_ = 2 * (p_ - lj_ * NPX);                              \
      if (p_ < NPAIR) st2((buf) + li_ + FX * lj_, src[s_]);                              \
    }                                                                                    \
  } while (0)

  RS_LOAD(kA, nxt);
  RS_STORE(lds, nxt);
  RS_LOAD(kA + 1, nxt);
  __syncthreads();

  for (int k = kA; k <= kB; ++k) {
    const double *R = lds + ((k - kA) & 1) * PLANE;
    RS_LOAD(k + 2, nn);

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
        for (int jj = 0; jj < MT; ++jj) {
          if (jj < nj) {
            const double *row = R + li0 + FX * (lj0 + jj);
#pragma unroll
            for (int ii = 0; ii < MT; ++ii) {
              if (ii < ni) {
                const double wv = wxy[jj][ii] * c2z * a.w2[2];
                fc = fc + wv * row[ii];
              }
            }
          }
        }
        if (k == z0 + nk - 1) {  // window complete: the coarse value is final
          const size_t c = (size_t)I + (size_t)a.nc[0] * ((size_t)J + (size_t)a.nc[1] * (size_t)(K - a.c_k0));
          rhs_c[c] = fc;
          if (u_c) u_c[c] = 0.0;  // ndsm_multigrid_core.f90:557-558
          fc = 0.0;
        }
        acc0 = slot == 0 ? fc : acc0;
        acc1 = slot == 1 ? fc : acc1;
        acc2 = slot == 2 ? fc : acc2;
        acc3 = slot == 3 ? fc : acc3;
      }
      while (Klo < Ke && a.rlo[2][Klo] + a.rcnt[2][Klo] - 1 <= k) ++Klo;
    }

    // plane k+1 into the other buffer (its readers finished one barrier ago)
    RS_STORE(lds + ((k + 1 - kA) & 1) * PLANE, nxt);
#pragma unroll
    for (int s = 0; s < NS; ++s) nxt[s] = nn[s];
    __syncthreads();
  }
#undef RS_LOAD
#undef RS_STORE
}

constexpr int kCI = 64, kCJ = 16, kMT = 5;

}  // namespace

namespace ndsm {

// footprint constants for the host-side coverage check (ndsmh_mg.f90)
extern "C" void ndsmk_restrict_stream_tile(int *ci, int *cj, int *fx, int *fy, int *maxt) {
  *ci = kCI;
  *cj = kCJ;
  *fx = 2 * kCI + 6;
  *fy = 2 * kCJ + 5;
  *maxt = kMT;
}

int launch_restrict_stream(const ndsmk_xfer *x, const double *r_f, double *rhs_c, double *u_c) {
  RSArgs a;
  for (int d = 0; d < 3; ++d) {
    a.nf[d] = x->nf[d];
    a.nc[d] = x->nc[d];
    a.rlo[d] = x->rlo[d];
    a.rcnt[d] = x->rcnt[d];
    a.rw[d] = x->rw[d];
    a.maxt[d] = x->maxt[d];
    a.w2[d] = x->w2[d];
  }
  a.f_k0 = x->f_k0;
  a.c_k0 = x->c_k0;
  a.c_beg = x->c_beg;
  a.c_cnt = x->c_cnt;
  a.nti = (x->nc[0] + kCI - 1) / kCI;
  a.ntj = (x->nc[1] + kCJ - 1) / kCJ;
  const int tiles = a.nti * a.ntj;
  int nkc = (768 + tiles - 1) / tiles;
  if (nkc < 1) nkc = 1;
  int kc = (x->c_cnt + nkc - 1) / nkc;
  if (kc < 8) kc = 8 < x->c_cnt ? 8 : x->c_cnt;
  a.kc = kc;
  a.nkc = (x->c_cnt + kc - 1) / kc;
  a.nwork = tiles * a.nkc;
  const int nblk = ((a.nwork + 7) / 8) * 8;
  constexpr size_t lds_bytes = sizeof(double) * 2 * (2 * kCI + 6) * (2 * kCJ + 5);
  auto kfn = restrict_stream_k<kCI, kCJ, kMT>;
  static bool attr_set = false;
  if (!attr_set) {
    NDSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(nblk), dim3(kCI * kCJ), lds_bytes, stream(), r_f, rhs_c, u_c, a);
  NDSM_LAUNCH_CHECK();
  return 0;
}

}  // namespace ndsm
