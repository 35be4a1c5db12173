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

#include <type_traits>

#include <cstdlib>

namespace {

struct RSArgs {
  int nf[3], nc[3];
  const int32_t *rlo[3], *rcnt[3];
  const double *rw[3];
  int maxt[3];
  double w2[3];
  int f_k0, c_k0, c_beg, c_cnt;
  int nti, ntj, nkc, kc, nwork;
  int probe;   // tuning aid (NDSM_RS_PROBE): 1 = stream the planes but skip the tap arithmetic, 2 = arithmetic without the loads
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
// fp32 fine field (mixed-precision mode): converted on load, all arithmetic stays fp64
__device__ __forceinline__ d2 ld2(const float *p) {
  const float2 t = *reinterpret_cast<const float2 *>(p);
  d2 r;
  r.x = (double)t.x;
  r.y = (double)t.y;
  return r;
}
__device__ __forceinline__ void st2(double *p, const d2 a) {
  double2 t;
  t.x = a.x;
  t.y = a.y;
  *reinterpret_cast<double2 *>(p) = t;
}

template <typename TF, int CI, int CJ, int MT, int KCMAX>
__global__ __launch_bounds__(CI *CJ) void restrict_stream_k(const TF *__restrict__ f, double *__restrict__ rhs_c,
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
  const bool oddx = (nx & 1) != 0;
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

  // ---- this thread's coarse column and its x / y tap weights ----
  const int I = I0 + tid % CI, J = J0 + tid / CI;
  const bool chave = I < a.nc[0] && J < a.nc[1];
  int ni = 0, nj = 0, li0 = 0, lj0 = 0;
  double cx[MT], cy[MT];
#pragma unroll
  for (int q = 0; q < MT; ++q) cx[q] = cy[q] = 0.0;
  if (chave) {
    ni = a.rcnt[0][I];
    nj = a.rcnt[1][J];
    li0 = a.rlo[0][I] - fx0;
    lj0 = a.rlo[1][J] - fy0;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      cx[q] = q < ni ? a.rw[0][(size_t)I * a.maxt[0] + q] : 0.0;
      cy[q] = q < nj ? a.rw[1][(size_t)J * a.maxt[1] + q] : 0.0;
    }
  }
  // z tables of this chunk's coarse planes in LDS: the plane loop reads them at uniform
  // addresses (from global memory every lookup is a dependent ~1 us load in the inner loop)
  __shared__ int s_z0[KCMAX], s_nk[KCMAX];
  __shared__ double s_zw[KCMAX * MT];
  for (int t = tid; t < Ke - Ks; t += NT) {
    const int nk = a.rcnt[2][Ks + t];
    s_z0[t] = a.rlo[2][Ks + t];
    s_nk[t] = nk;
    for (int q = 0; q < MT; ++q) s_zw[t * MT + q] = q < nk ? a.rw[2][(size_t)(Ks + t) * a.maxt[2] + q] : 0.0;
  }

#define RS_LOAD(tid, kglob, dst)                                                            \
  do {                                                                                   \
    const int kl_ = (kglob) - a.f_k0; /* local fine plane */                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                  \
      const int p_ = tid + NT * s_;                                                      \
      const int lj_ = p_ / NPX, li_ = 2 * (p_ - lj_ * NPX);                              \
      const int i_ = fx0 + li_, j_ = fy0 + lj_;                                          \
      d2 t_;                                                                             \
      t_.x = 0.0;                                                                        \
      t_.y = 0.0;                                                                        \
      if (p_ < NPAIR && i_ < nx && j_ < ny && (kglob) <= kB) {                           \
        const TF *q_ = f + sz * (size_t)kl_ + (i_ + nx * j_);                            \
        if (!oddx) {                                                                     \
          t_ = ld2(q_);                                                                  \
        } else { /* odd nx: rows are not 16-byte aligned and the last pair is half outside */ \
          const TF *q1_ = q_ + 1;                                                        \
          asm volatile("" : "+v"(q1_));                                                  \
          t_.x = q_[0];                                                                  \
          if (i_ + 1 < nx) t_.y = q1_[0];                                                \
        }                                                                                \
      }                                                                                  \
      dst[s_] = t_;                                                                      \
    }                                                                                    \
  } while (0)
#define RS_STORE(tid, buf, src)                                                               \
  do {                                                                                   \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                  \
      const int p_ = tid + NT * s_;                                                      \
      const int lj_ = p_ / NPX, li_ = 2 * (p_ - lj_ * NPX);                              \
      if (p_ < NPAIR) st2((buf) + li_ + FX * lj_, src[s_]);                              \
    }                                                                                    \
  } while (0)

  // cxw = c2x w2x: first factor of the weight chain ((((c2x w2x) c2y) w2y) c2z) w2z
  double cxw[MT];
#pragma unroll
  for (int q = 0; q < MT; ++q) cxw[q] = cx[q] * a.w2[0];  // 1 * c2 * w2 (ndsm_interp.f90:277-282)
  // a wave is one row of coarse columns (CI = 64 lanes, one J): the y tap count is wave-uniform
  const int njw = __builtin_amdgcn_readfirstlane(nj);

  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;  // coarse planes K with K & 3 = 0..3
  int Klo = Ks;
  auto getacc = [&](int slot) { return slot == 0 ? acc0 : (slot == 1 ? acc1 : (slot == 2 ? acc2 : acc3)); };
  // a coarse plane's sum after this fine plane: store it when its window is complete
  auto finish = [&](int K, double fc, bool complete) {
    if (complete) {
      if (chave) {
        const size_t c = (size_t)I + (size_t)a.nc[0] * ((size_t)J + (size_t)a.nc[1] * (size_t)(K - a.c_k0));
        rhs_c[c] = fc;
        if (u_c) u_c[c] = 0.0;  // ndsm_multigrid_core.f90:557-558
      }
      fc = 0.0;
    }
    const int slot = K & 3;
    acc0 = slot == 0 ? fc : acc0;
    acc1 = slot == 1 ? fc : acc1;
    acc2 = slot == 2 ? fc : acc2;
    acc3 = slot == 3 ? fc : acc3;
  };

  d2 nxt[NS];
  RS_LOAD(tid, kA, nxt);
  RS_STORE(tid, lds, nxt);
  __syncthreads();  // also publishes the z tables

  for (int k = kA; k <= kB; ++k) {
    const double *R = lds + ((k - kA) & 1) * PLANE;
    // the staging geometry is cheap integer work: rebuilt per plane from an opaque copy of the
    // thread index instead of being kept live across the tap loops
    int tidk = tid;
    asm volatile("" : "+v"(tidk));
    RS_LOAD(tidk, k + 1, nxt);  // in flight while plane k is consumed

    // Every thread walks the same coarse planes (threads without a coarse column have no taps
    // and store nothing), so the loop control stays scalar.  A fine plane usually feeds two
    // coarse planes: their accumulation chains - serial, the (z, y, x) tap order is part of the
    // result - are interleaved so that each hides the other's latency, and every patch value
    // read from LDS and every x-y weight prefix serves both.
    const double *P = R + li0 + FX * lj0;  // first tap of this column in the plane
    int K = Klo;
    while (K < Ke) {
      const int z0 = __builtin_amdgcn_readfirstlane(s_z0[K - Ks]);
      if (z0 > k) break;
      const int nk = __builtin_amdgcn_readfirstlane(s_nk[K - Ks]);
      if (k >= z0 + nk) {
        ++K;
        continue;
      }
      int z1 = 0, nk1 = 0;
      bool two = false;
      if (K + 1 < Ke) {
        z1 = __builtin_amdgcn_readfirstlane(s_z0[K + 1 - Ks]);
        nk1 = __builtin_amdgcn_readfirstlane(s_nk[K + 1 - Ks]);
        two = z1 <= k && k < z1 + nk1;
      }
      const double c2za = s_zw[(K - Ks) * MT + (k - z0)];
      double fa = getacc(K & 3);
      if (two) {
        const double c2zb = s_zw[(K + 1 - Ks) * MT + (k - z1)];
        double fb = getacc((K + 1) & 3);
#pragma unroll
        for (int jj = 0; jj < MT; ++jj) {
          if (jj < njw) {
#pragma unroll
            for (int ii = 0; ii < MT; ++ii) {
              if (ii < ni) {
                const double w0 = cxw[ii] * cy[jj] * a.w2[1];
                const double wa = w0 * c2za * a.w2[2];
                const double wb = w0 * c2zb * a.w2[2];
                const double fv = P[ii + FX * jj];
                fa = fa + wa * fv;
                fb = fb + wb * fv;
              }
            }
          }
        }
        finish(K, fa, k == z0 + nk - 1);
        finish(K + 1, fb, k == z1 + nk1 - 1);
        K += 2;
      } else {
#pragma unroll
        for (int jj = 0; jj < MT; ++jj) {
          if (jj < njw) {
#pragma unroll
            for (int ii = 0; ii < MT; ++ii) {
              if (ii < ni) {
                const double w0 = cxw[ii] * cy[jj] * a.w2[1];
                const double wa = w0 * c2za * a.w2[2];
                fa = fa + wa * P[ii + FX * jj];
              }
            }
          }
        }
        finish(K, fa, k == z0 + nk - 1);
        K += 1;
      }
    }
    while (Klo < Ke && __builtin_amdgcn_readfirstlane(s_z0[Klo - Ks] + s_nk[Klo - Ks]) - 1 <= k) ++Klo;

    // plane k+1 into the other buffer (its readers finished one barrier ago)
    RS_STORE(tidk, lds + ((k + 1 - kA) & 1) * PLANE, nxt);
    __syncthreads();
  }
#undef RS_LOAD
#undef RS_STORE
}

// Second form of the same kernel (default; NDSM_RS_VARIANT=0 selects the one above).  What the first
// one loses: a workgroup walks ~130 fine planes with ONE plane of prefetch, so every plane-step waits for
// most of an HBM round trip (counters: 31 % issue utilisation at 3.5 us per plane-step), and its tap
// loops are predicated per lane (ii < ni).  Here
//   * three planes are in flight: plane k+3 is requested while plane k is consumed, each in a register
//     slot of its own (the plane loop is unrolled by three so that the slots are static - a register
//     that is the target of a load in flight cannot be moved without waiting for it);
//   * the x tap count is made wave-uniform (the wave's maximum) and the weights of the taps a column does
//     not have are exact zeros: w * f = +-0 and x + (+-0) = x bit for bit (the running sum starts at +0
//     and can never become -0), so the padded taps change nothing and the loops need no exec masking.
//     LDS is zero-filled once so that a padded tap never meets an uninitialised word.  (Holds for finite
//     data: 0 * Inf is NaN, so a residual that already contains Inf / NaN - a diverged solve - can poison a
//     coarse point one column earlier than in the reference.)
// Same tap order and weight chain: bit-identical (tests/test_gpu_parity.py::test_transfer3d_bitwise,
// test_large_level_kernels_bitwise, the V-cycle tests).
template <typename TF, int CI, int CJ, int MT, int KCMAX, int WPS, bool ODDX, int DEPTH = 3, bool ROWB = false, bool SCHED = false>
__global__ __launch_bounds__(CI *CJ, WPS) void restrict_stream2_k(const TF *__restrict__ f, double *__restrict__ rhs_c,
                                                                  double *__restrict__ u_c, RSArgs a) {
  constexpr int NT = CI * CJ;
  constexpr int FX = 2 * CI + 6, FY = 2 * CJ + 5;
  constexpr int NPX = FX / 2, NPAIR = NPX * FY, NS = (NPAIR + NT - 1) / NT;
  constexpr int PLANE = FX * FY;
  constexpr int LDSN = 2 * PLANE + FX;  // two planes + a pad row for the padded taps of the last rows
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
  const int Ks = a.c_k0 + a.c_beg + ck * a.kc;
  const int Ke = min(Ks + a.kc, a.c_k0 + a.c_beg + a.c_cnt);
  const int fx0 = a.rlo[0][I0] & ~1;
  const int fy0 = a.rlo[1][J0];
  const int kA = a.rlo[2][Ks];
  const int kB = a.rlo[2][Ke - 1] + a.rcnt[2][Ke - 1] - 1;
  const int tid = (int)threadIdx.x;

  for (int t = tid; t < LDSN; t += NT) lds[t] = 0.0;

  const int I = I0 + tid % CI, J = J0 + tid / CI;
  const bool chave = I < a.nc[0] && J < a.nc[1];
  int ni = 0, nj = 0, li0 = 0, lj0 = 0;
  double cxw[MT], cy[MT];
#pragma unroll
  for (int q = 0; q < MT; ++q) cxw[q] = cy[q] = 0.0;
  if (chave) {
    ni = a.rcnt[0][I];
    nj = a.rcnt[1][J];
    li0 = a.rlo[0][I] - fx0;
    lj0 = a.rlo[1][J] - fy0;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      // cxw = c2x w2x: first factor of the weight chain ((((c2x w2x) c2y) w2y) c2z) w2z (ndsm_interp.f90:277-282)
      cxw[q] = q < ni ? a.rw[0][(size_t)I * a.maxt[0] + q] * a.w2[0] : 0.0;
      cy[q] = q < nj ? a.rw[1][(size_t)J * a.maxt[1] + q] : 0.0;
    }
  }
  __shared__ int s_z0[KCMAX], s_nk[KCMAX];
  __shared__ double s_zw[KCMAX * MT];
  for (int t = tid; t < Ke - Ks; t += NT) {
    const int nk = a.rcnt[2][Ks + t];
    s_z0[t] = a.rlo[2][Ks + t];
    s_nk[t] = nk;
    for (int q = 0; q < MT; ++q) s_zw[t * MT + q] = q < nk ? a.rw[2][(size_t)(Ks + t) * a.maxt[2] + q] : 0.0;
  }
  // wave-uniform tap counts: a wave is one row of coarse columns (one J); x: the wave's maximum
  const int njw = __builtin_amdgcn_readfirstlane(nj);
  int nim = ni;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nim = max(nim, __shfl_xor(nim, o, 64));
  nim = __builtin_amdgcn_readfirstlane(nim);

  // staging geometry of this thread's NS pairs.  Loads are UNCONDITIONAL - from a clamped, always valid
  // address - and what must not enter the tile is replaced by zero when the slot is written to LDS: a load
  // under a branch is waited for at the join right behind it (the compiler cannot keep a value that exists
  // on one path only in flight), which turns every prefetch into a synchronous load.
  int goff[NS], loff[NS];
  bool gx0[NS], gx1[NS];   // element 0 / 1 of the pair lies inside the fine grid
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    const int p_ = tid + NT * s_;
    const int lj_ = p_ / NPX, li_ = 2 * (p_ - lj_ * NPX);
    const int i_ = fx0 + li_, j_ = fy0 + lj_;
    loff[s_] = p_ < NPAIR ? li_ + FX * lj_ : -1;
    const bool in = p_ < NPAIR && i_ < nx && j_ < ny;
    gx0[s_] = in;
    gx1[s_] = in && i_ + 1 < nx;
    goff[s_] = in ? i_ + nx * j_ : 0;
  }
  auto load_plane = [&](int kglob, d2(&dst)[NS]) {
    const TF *pk = f + sz * (size_t)(min(kglob, kB) - a.f_k0);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const TF *q_ = pk + goff[s_];
      if (!ODDX) {
        dst[s_] = ld2(q_);
      } else {       // odd nx: rows are not 16-byte aligned (element loads) and the last pair is half outside
        dst[s_].x = (double)q_[0];
        dst[s_].y = (double)q_[gx1[s_] ? 1 : 0];
      }
    }
  };
  auto store_plane = [&](double *buf, const d2(&src)[NS], bool plane_ok) {
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      d2 v = src[s_];
      v.x = (plane_ok && gx0[s_]) ? v.x : 0.0;
      v.y = (plane_ok && gx1[s_]) ? v.y : 0.0;
      if (loff[s_] >= 0) st2(buf + loff[s_], v);
    }
  };

  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;  // coarse planes K with K & 3 = 0..3
  int Klo = Ks;
  auto getacc = [&](int slot) { return slot == 0 ? acc0 : (slot == 1 ? acc1 : (slot == 2 ? acc2 : acc3)); };
  auto finish = [&](int K, double fc, bool complete) {
    if (complete) {
      if (chave) {
        const size_t c = (size_t)I + (size_t)a.nc[0] * ((size_t)J + (size_t)a.nc[1] * (size_t)(K - a.c_k0));
        rhs_c[c] = fc;
        if (u_c) u_c[c] = 0.0;  // ndsm_multigrid_core.f90:557-558
      }
      fc = 0.0;
    }
    const int slot = K & 3;
    acc0 = slot == 0 ? fc : acc0;
    acc1 = slot == 1 ? fc : acc1;
    acc2 = slot == 2 ? fc : acc2;
    acc3 = slot == 3 ? fc : acc3;
  };

  // the taps of fine plane k (in LDS buffer R) for every coarse plane whose z window holds it
  // the taps of one fine plane for two coarse planes at once (they share the plane's values and the x-y weight
  // prefixes) / for one
  auto taps2 = [&](const double *P, double c2za, double c2zb, double &fa, double &fb) {
    if constexpr (ROWB) {
      // a row's taps behind ONE wait: inside each branch the x tap count is a compile-time constant (the wave's
      // 3, 4 or 5), so a row is NI LDS reads issued back to back and then its multiply / add chain - with a
      // uniform branch around every tap each read was waited for on its own (139 s_waitcnt per plane-step)
      auto rows = [&](auto ni_c) {
        constexpr int NI = decltype(ni_c)::value;
#pragma unroll
        for (int jj = 0; jj < MT; ++jj) {
          if (jj < njw) {
            double fv[NI];
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) fv[ii] = P[ii + FX * jj];
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
              const double w0 = cxw[ii] * cy[jj] * a.w2[1];
              const double wa = w0 * c2za * a.w2[2];
              const double wb = w0 * c2zb * a.w2[2];
              fa = fa + wa * fv[ii];
              fb = fb + wb * fv[ii];
            }
          }
        }
      };
      if (nim == 4)
        rows(std::integral_constant<int, 4>());
      else if (nim == 5)
        rows(std::integral_constant<int, 5>());
      else
        rows(std::integral_constant<int, 3>());   // (fewer: the weights of the taps a column lacks are zero)
    } else {
#pragma unroll
      for (int jj = 0; jj < MT; ++jj) {
        if (jj < njw) {
#pragma unroll
          for (int ii = 0; ii < MT; ++ii) {
            if (ii < nim) {
              const double w0 = cxw[ii] * cy[jj] * a.w2[1];
              const double wa = w0 * c2za * a.w2[2];
              const double wb = w0 * c2zb * a.w2[2];
              const double fv = P[ii + FX * jj];
              fa = fa + wa * fv;
              fb = fb + wb * fv;
            }
          }
        }
      }
    }
  };
  auto taps1 = [&](const double *P, double c2za, double &fa) {
#pragma unroll
    for (int jj = 0; jj < MT; ++jj) {
      if (jj < njw) {
#pragma unroll
        for (int ii = 0; ii < MT; ++ii) {
          if (ii < nim) {
            const double w0 = cxw[ii] * cy[jj] * a.w2[1];
            const double wa = w0 * c2za * a.w2[2];
            fa = fa + wa * P[ii + FX * jj];
          }
        }
      }
    }
  };

  // the taps of fine plane k (in LDS buffer R) for every coarse plane K >= Kfrom whose z window holds it: the
  // windows are looked up in the chunk's z tables as the walk goes
  auto consume_from = [&](int k, const double *R, int Kfrom) {
    const double *P = R + li0 + FX * lj0;
    int K = Kfrom;
    while (K < Ke) {
      const int z0 = __builtin_amdgcn_readfirstlane(s_z0[K - Ks]);
      if (z0 > k) break;
      const int nk = __builtin_amdgcn_readfirstlane(s_nk[K - Ks]);
      if (k >= z0 + nk) {
        ++K;
        continue;
      }
      int z1 = 0, nk1 = 0;
      bool two = false;
      if (K + 1 < Ke) {
        z1 = __builtin_amdgcn_readfirstlane(s_z0[K + 1 - Ks]);
        nk1 = __builtin_amdgcn_readfirstlane(s_nk[K + 1 - Ks]);
        two = z1 <= k && k < z1 + nk1;
      }
      const double c2za = s_zw[(K - Ks) * MT + (k - z0)];
      double fa = getacc(K & 3);
      if (two) {
        const double c2zb = s_zw[(K + 1 - Ks) * MT + (k - z1)];
        double fb = getacc((K + 1) & 3);
        taps2(P, c2za, c2zb, fa, fb);
        finish(K, fa, k == z0 + nk - 1);
        finish(K + 1, fb, k == z1 + nk1 - 1);
        K += 2;
      } else {
        taps1(P, c2za, fa);
        finish(K, fa, k == z0 + nk - 1);
        K += 1;
      }
    }
  };
  auto consume = [&](int k, const double *R) {
    consume_from(k, R, Klo);
    while (Klo < Ke && __builtin_amdgcn_readfirstlane(s_z0[Klo - Ks] + s_nk[Klo - Ks]) - 1 <= k) ++Klo;
  };

  // SCHED: that walk costs three to five dependent LDS round trips (window starts, lengths, weights, each read
  // made wave-uniform with a readfirstlane) before the first multiplication of every plane-step.  The answers do
  // not depend on the data: a schedule per fine plane of the chunk - first coarse plane that takes it, how many do
  // (at most 4: launch_rs_t refuses this form for a level pair with more), their z weights, which of them it completes - is built once
  // from the z tables, and the record of plane k+1 is read while plane k's step drains into its barrier.
  constexpr int KPL = SCHED ? 2 * KCMAX + 16 : 1;
  __shared__ int s_sk[KPL], s_sc[KPL];
  __shared__ double s_sw[4 * KPL];
  const int nrec = SCHED ? min(kB - kA + 1, KPL) : 0;
  int rk = 0, rc = 0;
  double rw0 = 0.0, rw1 = 0.0, rw2 = 0.0, rw3 = 0.0;
  auto read_rec = [&](int k) {
    const int t = max(min(k - kA, nrec - 1), 0);
    rk = s_sk[t];
    rc = k - kA < nrec ? s_sc[t] : 0;
    rw0 = s_sw[4 * t];
    rw1 = s_sw[4 * t + 1];
    rw2 = s_sw[4 * t + 2];
    rw3 = s_sw[4 * t + 3];
  };
  // (launch_rs_t only selects this form where a chunk has at most KPL fine planes and no fine plane lies in more
  // than four coarse windows - numbers the host measures per level pair, ndsmh_mg.f90:stream_restrict_applies)
  auto consume_s = [&](const double *R) {
    const int cc = __builtin_amdgcn_readfirstlane(rc);
    const int cnt = cc & 255, done = cc >> 8;
    const double *P = R + li0 + FX * lj0;
    const int K0 = __builtin_amdgcn_readfirstlane(rk);
    int e = 0;
    while (e < cnt) {          // one or two rounds: pairs of coarse planes, then a single one
      const int K = K0 + e;
      const double wa = e == 0 ? rw0 : rw2;
      double fa = getacc(K & 3);
      if (cnt - e >= 2) {
        const double wb = e == 0 ? rw1 : rw3;
        double fb = getacc((K + 1) & 3);
        taps2(P, wa, wb, fa, fb);
        finish(K, fa, ((done >> e) & 1) != 0);
        finish(K + 1, fb, ((done >> (e + 1)) & 1) != 0);
        e += 2;
      } else {
        taps1(P, wa, fa);
        finish(K, fa, ((done >> e) & 1) != 0);
        e += 1;
      }
    }
  };

  // ---- prologue: plane kA into LDS buffer 0, planes kA+1 (, kA+2) on their way ----
  static_assert(DEPTH == 2 || DEPTH == 3, "planes in flight");
  d2 r0[NS], r1[NS], r2[DEPTH == 3 ? NS : 1];
  __syncthreads();  // the zero fill is complete (and the z tables are)
  load_plane(kA, r0);
  load_plane(kA + 1, r1);
  if constexpr (DEPTH == 3) load_plane(kA + 2, r2);
  if constexpr (SCHED) {
    for (int t = tid; t < nrec; t += NT) {
      const int k = kA + t;
      int K0 = Ke, c = 0, done = 0;
      for (int K = Ks; K < Ke; ++K) {
        const int z0 = s_z0[K - Ks], nk = s_nk[K - Ks];
        if (k >= z0 && k < z0 + nk) {
          if (c == 0) K0 = K;
          if (c < 4) {
            s_sw[4 * t + c] = s_zw[(K - Ks) * MT + (k - z0)];
            if (k == z0 + nk - 1) done |= 1 << c;
          }
          ++c;
        }
      }
      for (int q = c; q < 4; ++q) s_sw[4 * t + q] = 0.0;
      s_sk[t] = K0;
      s_sc[t] = (c > 255 ? 255 : c) | (done << 8);
    }
  }
  store_plane(lds, r0, true);
  __syncthreads();  // also publishes the schedule
  if constexpr (SCHED) read_rec(kA);

  // one plane-step: request plane k+DEPTH into the slot plane k came from, consume plane k, move plane k+1
  // from its slot into the other LDS buffer
  auto step = [&](int k, d2(&slot_k)[NS], const d2(&slot_k1)[NS]) {
    if (a.probe != 2) load_plane(k + DEPTH, slot_k);
    if constexpr (SCHED) {
      if (a.probe != 1) consume_s(lds + ((k - kA) & 1) * PLANE);
      read_rec(k + 1);   // (its LDS reads ride the barrier below)
    } else {
      if (a.probe != 1) consume(k, lds + ((k - kA) & 1) * PLANE);
    }
    (void)consume;
    store_plane(lds + ((k + 1 - kA) & 1) * PLANE, slot_k1, k + 1 <= kB);
    __syncthreads();
  };
  if constexpr (DEPTH == 3) {
    for (int k = kA; k <= kB; k += 3) {
      step(k, r0, r1);
      if (k + 1 <= kB) step(k + 1, r1, r2);
      if (k + 2 <= kB) step(k + 2, r2, r0);
    }
  } else {
    for (int k = kA; k <= kB; k += 2) {
      step(k, r0, r1);
      if (k + 1 <= kB) step(k + 1, r1, r0);
    }
  }
}

constexpr int kCI = 64, kMT = 5, kKCMax = 64;

// which form of the kernel runs (NDSM_RS_VARIANT; tuning aid): 0 the first kernel (64 x 8 coarse columns,
// one plane of prefetch: 432 us at 512^3), 1 restrict_stream2_k with 64 x 8 columns and three planes in flight
// (388-393 us, the default until the end of round 2); with two planes in flight: 2 = 64 x 4 columns (256 threads:
// twice as many independent barrier groups per CU; 391 us), 5 = 64 x 8 columns (383 us).
// 8 (the default now) = 5 + the per-chunk SCHEDULE of the z windows (SCHED: see consume_s) - the walk through the
// window tables cost three to five dependent LDS round trips and as many scalar instructions as the plane-step
// has vector ones (counters: 8.8e7 SALU against 1.1e8 VALU instructions); with the record of the next plane read
// behind the barrier: 361-373 us on the box where 1 takes 397-402 (arithmetic alone 334 against 373).
// 6 / 7 / 9 = the x taps of a row read back to back behind one wait (ROWB) on top of 5 / 1 / 8: 383-391 us for 6
// (the per-tap waits were not the limit), 7 and 9 spill (528 / 466 us).  Also tried on top of 8: accumulators kept
// in completion order (entry e of the record = accumulator e, shifted down when planes complete: no selects on
// a slot number) - the compiler parks three of the four accumulators in scratch and the launch takes 421 us;
// a parity-planar LDS plane (even fine columns first, then the odd ones: the tap reads of a wave become
// consecutive words instead of every second one, no 2-way bank conflict) - same bits, 359-368 against 367-380 us:
// inside the noise, as the counters said (LDS issue waits 0.7 % of the wave cycles); not kept.  Forcing three workgroups per CU (<= 85 VGPRs)
// spills and takes 816-1400 us: not built.  Neither the prefetch depth nor the number of barrier groups moves
// the kernel any further, because it is bound by the tap arithmetic, not by memory: with the arithmetic skipped
// the same launch streams its 1.28 GB in 174-190 us (6.7-7.3 TB/s), with the loads skipped the arithmetic alone
// takes 365 us (NDSM_RS_PROBE=1 / 2) - fp64 multiply / add chains at four waves per SIMD (126 VGPRs; the
// compiler keeps a plane's 16-25 tap values and weight prefixes in registers) run the SIMDs at ~50 %.
// Also tried: a straight-line 4 x 4 tap block (no uniform branch around each tap, so that the LDS reads of a
// row overlap) with the y weights in scalar registers and the prefix re-formed per tap to stay inside 128
// VGPRs - 455 us (arithmetic alone 430): the 25 % more multiplications cost more than the branches did.
int rs_variant() {
  static int variant = -1;
  if (variant < 0) {
    const char *e = std::getenv("NDSM_RS_VARIANT");
    variant = e ? std::atoi(e) : 8;
    if (variant != 0 && variant != 1 && variant != 2 && (variant < 5 || variant > 9)) variant = 8;
  }
  return variant;
}
int rs_cj() { return rs_variant() == 2 ? 4 : 8; }

}  // namespace

namespace ndsm {

// footprint constants for the host-side coverage check (ndsmh_mg.f90)
extern "C" void ndsmk_restrict_stream_tile(int *ci, int *cj, int *fx, int *fy, int *maxt) {
  *ci = kCI;
  *cj = rs_cj();
  *fx = 2 * kCI + 6;
  *fy = 2 * rs_cj() + 5;
  *maxt = kMT;
}

template <typename TF, int CJ, int WPS, bool OLD, int DEPTH = 3, bool ROWB = false, bool SCHED = false>
static int launch_rs_v(const ndsmk_xfer *x, const TF *r_f, double *rhs_c, double *u_c) {
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
  a.ntj = (x->nc[1] + CJ - 1) / CJ;
  const int tiles = a.nti * a.ntj;
  constexpr size_t lds_bytes =
      sizeof(double) * (2 * (2 * kCI + 6) * (2 * CJ + 5) + (OLD ? 0 : (2 * kCI + 6)));
  const bool odd = !OLD && (x->nf[0] & 1);
  const void *kfn;
  if constexpr (OLD) {
    kfn = reinterpret_cast<const void *>(restrict_stream_k<TF, kCI, CJ, kMT, kKCMax>);
  } else {
    kfn = odd ? reinterpret_cast<const void *>(restrict_stream2_k<TF, kCI, CJ, kMT, kKCMax, WPS, true, DEPTH, ROWB, SCHED>)
              : reinterpret_cast<const void *>(restrict_stream2_k<TF, kCI, CJ, kMT, kKCMax, WPS, false, DEPTH, ROWB, SCHED>);
  }
  // coarse planes per chunk: a chunk of kc coarse planes walks ~2 kc + 3 fine planes: minimise (rounds of
  // workgroups at the kernel's occupancy) x (planes walked); the chunk's z tables must fit their LDS
  // arrays (kc <= kKCMax)
  static int occ[2] = {0, 0}, epoch[2] = {0, 0};
  if (ndsm::first_in_epoch(epoch[odd])) {
    NDSM_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int o = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kfn, kCI * CJ, lds_bytes) != hipSuccess || o < 1) o = 1;
    occ[odd] = o;
  }
  const int64_t slots = (int64_t)ndsm::cu_count() * occ[odd];
  int kc = x->c_cnt < kKCMax ? x->c_cnt : kKCMax;
  int64_t best = -1;
  for (int c = 1; c <= x->c_cnt; ++c) {
    const int k1 = (x->c_cnt + c - 1) / c;
    if (k1 > kKCMax) continue;
    const int c1 = (x->c_cnt + k1 - 1) / k1;
    const int64_t rounds = ((int64_t)tiles * c1 + slots - 1) / slots;
    const int64_t cost = rounds * (2 * k1 + 3);
    if (best < 0 || cost < best) {
      best = cost;
      kc = k1;
    }
    if (k1 <= 4) break;
  }
  a.kc = kc;
  a.nkc = (x->c_cnt + kc - 1) / kc;
  a.nwork = tiles * a.nkc;
  static int probe = -1;
  if (probe < 0) probe = std::getenv("NDSM_RS_PROBE") ? std::atoi(std::getenv("NDSM_RS_PROBE")) : 0;
  a.probe = probe;
  const int nblk = ((a.nwork + 7) / 8) * 8;
  void *args[] = {(void *)&r_f, (void *)&rhs_c, (void *)&u_c, (void *)&a};
  NDSM_HIP(hipLaunchKernel(kfn, dim3(nblk), dim3(kCI * CJ), args, lds_bytes, stream()));
  return 0;
}

template <typename TF>
static int launch_rs_t(const ndsmk_xfer *x, const TF *r_f, double *rhs_c, double *u_c) {
  int v = rs_variant();
  // The scheduled forms hold a chunk's z windows in a fixed-size table: at most four coarse windows per fine plane
  // (consume_s has four weight slots) and at most 2 * kKCMax + 16 fine planes per chunk (KPL records).  The host
  // measures both for a level pair (ndsmh_mg.f90: stream_restrict_applies); they are checked HERE, against the
  // kernel's limits - a descriptor that does not carry the numbers takes the table walk instead.
  {
    const int wmax = (x->stream_ok >> 8) & 255, smax = (x->stream_ok >> 16) & 32767;
    if ((v == 8 || v == 9) && (!(x->stream_ok & 4) || wmax < 1 || wmax > 4 || smax < 1 || smax > 2 * kKCMax + 16)) v = 5;
  }
  switch (v) {
    case 0: return launch_rs_v<TF, 8, 4, true>(x, r_f, rhs_c, u_c);
    case 2: return launch_rs_v<TF, 4, 4, false, 2>(x, r_f, rhs_c, u_c);
    case 5: return launch_rs_v<TF, 8, 4, false, 2>(x, r_f, rhs_c, u_c);
    case 6: return launch_rs_v<TF, 8, 4, false, 2, true>(x, r_f, rhs_c, u_c);
    case 7: return launch_rs_v<TF, 8, 4, false, 3, true>(x, r_f, rhs_c, u_c);
    case 9: return launch_rs_v<TF, 8, 4, false, 2, true, true>(x, r_f, rhs_c, u_c);
    case 8: return launch_rs_v<TF, 8, 4, false, 2, false, true>(x, r_f, rhs_c, u_c);
    default: return launch_rs_v<TF, 8, 4, false>(x, r_f, rhs_c, u_c);
  }
}

int launch_restrict_stream(const ndsmk_xfer *x, const double *r_f, double *rhs_c, double *u_c) {
  return launch_rs_t<double>(x, r_f, rhs_c, u_c);
}
int launch_restrict_stream_f32(const ndsmk_xfer *x, const float *r_f, double *rhs_c, double *u_c) {
  return launch_rs_t<float>(x, r_f, rhs_c, u_c);
}

}  // namespace ndsm
