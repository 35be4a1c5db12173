// Restriction rhs_c = R r_f for the large levels: z-streamed through LDS.
//
// restrict_k (transfer.hip) lets every coarse point gather its 64-125 taps from
// global memory: each fine value is requested ~8 times with a stride-2 lane
// pattern and the kernel runs at ~1/7 of its 9 B/pt roofline.  Here a workgroup
// owns CI x CJ coarse columns and a chunk of coarse planes, streams the fine
// planes of that chunk once (coalesced 16-B accesses, LDS planes in rotation, one
// barrier per plane) and every thread - one coarse column - adds the plane's
// taps to the coarse planes whose z window contains it.
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
  int probe;   // tuning aid (NDSM_RS_PROBE, register-staged form): 1 = stream the planes but skip the tap arithmetic, 2 = arithmetic without the loads
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

// The kernel.  Two forms of the plane traffic, one tap arithmetic:
//
//   * REGISTER-STAGED (odd nx, fp32 fine fields, level pairs whose z windows do not fit the schedule): two planes in
//     flight, each in a register slot of its own (the plane loop is unrolled by two so that the slots are static - a
//     register that is the target of a load in flight cannot be moved without waiting for it), written to one of two
//     LDS planes when their step comes.  The loads are unconditional from clamped addresses (a load under a branch is
//     waited for at the join right behind it, which turns every prefetch into a synchronous load); what must not enter
//     the tile is zeroed when the slot is written to LDS.
//   * DMA (round 3; fp64, even nx, with the schedule - the default): the fine planes go from HBM STRAIGHT INTO LDS
//     (`buffer_load_dwordx4 ... lds`: the tile is linear in the pair index, so the lane-linear destination of the
//     instruction is the tile itself).  What that removes from every plane-step: the staging registers (24 of 126),
//     the pass that moved a slot into LDS with its four selects per pair (what must not enter the tile - beyond the
//     fine grid, beyond the chunk's last plane - is zero because the buffer descriptor's range check makes it so: a
//     lane outside carries an offset beyond the plane, a plane outside a descriptor of zero records) and its LDS
//     writes.  Three LDS planes: plane k is read while k+1 and k+2 are on their way; a step ends with a COUNTED
//     wait - everything but the plane requested in this step has landed - and a bare s_barrier (__syncthreads()
//     would make the compiler drain every transfer in flight: it treats them as LDS stores the workgroup fence must
//     publish).  With the registers that frees, the accumulators are kept in the ORDER of the open coarse planes
//     (consume_o): no select on a slot number anywhere.  512^3: 362 -> 290-305 us.
//
// In both: the x tap count is wave-uniform (the wave's maximum) and the weights of the taps a column does not have
// are exact zeros: w * f = +-0 and x + (+-0) = x bit for bit (the running sum starts at +0 and can never become
// -0), so the padded taps change nothing and the loops need no exec masking.  LDS is zero-filled once so that a
// padded tap never meets an uninitialised word.  (Holds for finite data: 0 * Inf is NaN, so a residual that already
// contains Inf / NaN - a diverged solve - can poison a coarse point one column earlier than in the reference.)
// Same tap order and weight chain as the reference: bit-identical (tests/test_gpu_parity.py::
// test_transfer3d_bitwise, test_large_level_kernels_bitwise, the V-cycle tests; scripts/fuzz_restrict.py).
template <typename TF, int CI, int CJ, int MT, int KCMAX, int WPS, bool ODDX, bool SCHED, bool DMA>
__global__ __launch_bounds__(CI *CJ, WPS) void restrict_stream_k(const TF *__restrict__ f, double *__restrict__ rhs_c,
                                                                 double *__restrict__ u_c, RSArgs a) {
  constexpr int NT = CI * CJ;
  constexpr int FX = 2 * CI + 6, FY = 2 * CJ + 5;
  constexpr int NPX = FX / 2, NPAIR = NPX * FY, NS = (NPAIR + NT - 1) / NT;
  constexpr int PLANE = FX * FY;
  constexpr int NBUF = DMA ? 3 : 2;
  constexpr int LDSN = NBUF * PLANE + FX;  // the planes + a pad row for the padded taps of the last rows
  static_assert(!DMA || (std::is_same<TF, double>::value && !ODDX && SCHED), "DMA: fp64, 16-byte aligned rows, scheduled");
  static_assert(!DMA || ((NS - 1) * NT < NPAIR && (PLANE % 2) == 0), "DMA: only the last slot may be empty for a wave");
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

  // DMA: byte offset of the slot's pair inside a fine plane (beyond the plane where the pair lies outside the fine
  // grid: the range check returns zeros), and how many of this wave's slots hold a pair at all (NS or NS - 1: the
  // waves at the end of the last slot issue nothing there - the counted wait below must know)
  constexpr unsigned kDead = 0x80000000u;
  unsigned goffb[DMA ? NS : 1];
  int nsw = NS;
  if constexpr (DMA) {
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) goffb[s_] = gx0[s_] ? (unsigned)goff[s_] * 8u : kDead;
    nsw = __builtin_amdgcn_readfirstlane(((tid & ~63) + NT * (NS - 1) < NPAIR) ? NS : NS - 1);
  }
  const unsigned plane_bytes = (unsigned)(sz * sizeof(double));
  auto dma_plane = [&](int kglob, int buf) __attribute__((always_inline)) {
    if constexpr (DMA) {
      const bool ok = kglob <= kB;
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<double *>(reinterpret_cast<const double *>(f)) + sz * (size_t)((ok ? kglob : kB) - a.f_k0), 0,
          ok ? plane_bytes : 0u, 0x00020000);
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const int p0 = __builtin_amdgcn_readfirstlane((tid & ~63) + NT * s_);   // the wave's first pair of the slot
        if (tid + NT * s_ < NPAIR)   // (lanes beyond the tile would land in the next plane)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + buf * PLANE + 2 * p0), 16,
                                                   (int)goffb[s_], 0, 0, 0);   // (the cast matters: with the unsigned lvalue the HOST pass silently drops the kernel stub)
      }
    }
  };
  // end of a DMA step: every transfer but those of the plane requested last has landed (loads return in order; the
  // result stores issued since only make the count stricter), this wave's LDS reads are done, then the barrier
  auto dma_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    if (nsw == NS)
      __builtin_amdgcn_s_waitcnt((NS & 15) | (7 << 4) | ((NS >> 4) << 14));          // vmcnt(NS) lgkmcnt(0)
    else
      __builtin_amdgcn_s_waitcnt(((NS - 1) & 15) | (7 << 4) | (((NS - 1) >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
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
  auto taps2 = [&](const double *P, double c2za, double c2zb, double &fa, double &fb) __attribute__((always_inline)) {
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
  };
  auto taps1 = [&](const double *P, double c2za, double &fa) __attribute__((always_inline)) {
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
  auto read_rec = [&](int k) __attribute__((always_inline)) {
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
  auto consume_s = [&](const double *R) __attribute__((always_inline)) {
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

  // DMA form: the accumulators in the ORDER of the open coarse planes - entry e of a plane's record adds to
  // accumulator e, a completed plane leaves from the front and the others move down (z windows are contiguous and
  // both their starts and their ends grow with K, so the planes that take fine plane k are exactly the open ones,
  // oldest first, and they complete oldest first).  No select on a slot number anywhere: moves under a uniform
  // branch every other step instead of ~30 v_cndmask per step.  (With the staging registers of the first form this
  // spilled - 421 us; the DMA form has 30 registers to spare.)
  auto consume_o = [&](const double *R) __attribute__((always_inline)) {
    const int cc = __builtin_amdgcn_readfirstlane(rc);
    const int cnt = cc & 255;
    int done = cc >> 8;
    const double *P = R + li0 + FX * lj0;
    int K = __builtin_amdgcn_readfirstlane(rk);
    if (cnt >= 2)
      taps2(P, rw0, rw1, acc0, acc1);
    else if (cnt == 1)
      taps1(P, rw0, acc0);
    if (cnt >= 4)
      taps2(P, rw2, rw3, acc2, acc3);
    else if (cnt == 3)
      taps1(P, rw2, acc2);
    while (done & 1) {
      if (chave) {
        const size_t c = (size_t)I + (size_t)a.nc[0] * ((size_t)J + (size_t)a.nc[1] * (size_t)(K - a.c_k0));
        rhs_c[c] = acc0;
        if (u_c) u_c[c] = 0.0;  // ndsm_multigrid_core.f90:557-558
      }
      acc0 = acc1;
      acc1 = acc2;
      acc2 = acc3;
      acc3 = 0.0;
      ++K;
      done >>= 1;
    }
  };

  // ---- prologue: plane kA into LDS buffer 0, plane kA+1 on its way ----
  d2 r0[DMA ? 1 : NS], r1[DMA ? 1 : NS];
  __syncthreads();  // the zero fill is complete (and the z tables are)
  if constexpr (DMA) {
    dma_plane(kA, 0);
    dma_plane(kA + 1, 1);
  } else {
    load_plane(kA, r0);
    load_plane(kA + 1, r1);
  }
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
  if constexpr (DMA) {
    dma_barrier();   // plane kA has landed everywhere; also publishes the schedule
    read_rec(kA);
    // one plane-step: request plane k+2 into the buffer plane k-1 was read from, consume plane k
    auto step_d = [&](int k, auto BT) __attribute__((always_inline)) {
      constexpr int B = decltype(BT)::value;
      dma_plane(k + 2, (B + 2) % 3);
      consume_o(lds + B * PLANE);
      read_rec(k + 1);
      dma_barrier();
    };
    for (int k = kA; k <= kB; k += 3) {
      step_d(k, std::integral_constant<int, 0>());
      if (k + 1 <= kB) step_d(k + 1, std::integral_constant<int, 1>());
      if (k + 2 <= kB) step_d(k + 2, std::integral_constant<int, 2>());
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // vmcnt(0): no transfer outlives the workgroup's LDS
    return;
  } else {
  store_plane(lds, r0, true);
  __syncthreads();  // also publishes the schedule
  if constexpr (SCHED) read_rec(kA);

  // one plane-step: request plane k+2 into the slot plane k came from, consume plane k, move plane k+1
  // from its slot into the other LDS buffer
  auto step = [&](int k, d2(&slot_k)[NS], const d2(&slot_k1)[NS]) {
    if (a.probe != 2) load_plane(k + 2, slot_k);
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
  for (int k = kA; k <= kB; k += 2) {
    step(k, r0, r1);
    if (k + 1 <= kB) step(k + 1, r1, r0);
  }
  }
}

constexpr int kCI = 64, kCJ = 8, kMT = 5, kKCMax = 64;

// Which form runs (NDSM_RS_VARIANT, tuning aid): 10 (default) = DMA + schedule + ordered accumulators where the level
// pair allows it, 8 = register-staged with the schedule, 5 = register-staged with the table walk (what 10 and 8 fall
// back to where the schedule's limits do not hold).  512^3 -> 256^3 on MI355X: 383 / 361-373 / 290-305 us for 5 / 8 / 10.
//
// What was tried on the way and is no longer built (all bit-identical, all measured at 512^3): the first form of
// the kernel with ONE plane of prefetch issued under a branch (432 us: every "prefetch" was a synchronous load);
// three planes in flight instead of two (388-393 us against 383); 64 x 4 coarse columns per workgroup, twice as many
// independent barrier groups per CU (391 us); the x taps of a row read back to back behind one wait (383-391 us, and
// it spills together with the schedule: 466 us - the per-tap waits were not the limit), again on top of the DMA form
// (335-341 against 328-337 us); a straight-line 4 x 4 tap block for the common shape, no uniform branch around a tap
// (455 us with the prefixes re-formed per tap to fit the registers of the staged form, 309-313 against 288-297 us on
// the DMA form: the branches are not the limit either); a parity-planar LDS plane (no 2-way bank conflict on the tap
// reads: inside the noise - LDS issue waits are 0.7 % of the wave cycles); three workgroups per CU (<= 85 VGPRs:
// spills, 816-1400 us); ordered accumulators in the staged form (three of four parked in scratch: 421 us).
// Where the time goes (round 3, probes on the staged form with parts removed): 364 us as built, 329 without the global
// loads, 297 without the pass that writes the staged plane to LDS (-> the DMA form), 356 without the barrier, 347
// without the tap reads from LDS, 270 with all three removed - the tap arithmetic itself: 7.6e7 fp64 wave-instructions
// (3 multiplications + 1 addition per tap and coarse plane: the reference's weight chain) whose issue floor is
// ~125-200 us, in basic blocks of one tap behind uniform tests.
int rs_variant() {
  static int variant = -1;
  if (variant < 0) {
    const char *e = std::getenv("NDSM_RS_VARIANT");
    variant = e ? std::atoi(e) : 10;
    if (variant != 5 && variant != 8 && variant != 10) variant = 10;
  }
  return variant;
}

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

template <typename TF, bool SCHED, bool DMA>
static int launch_rs_v(const ndsmk_xfer *x, const TF *r_f, double *rhs_c, double *u_c) {
  constexpr int WPS = 4;
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
  constexpr size_t lds_bytes = sizeof(double) * ((DMA ? 3 : 2) * (2 * kCI + 6) * (2 * kCJ + 5) + (2 * kCI + 6));
  const bool odd = (x->nf[0] & 1) != 0;
  const void *kfn;
  if constexpr (DMA)   // (launch_rs_t sends only fp64 levels with even nx here)
    kfn = reinterpret_cast<const void *>(restrict_stream_k<TF, kCI, kCJ, kMT, kKCMax, WPS, false, SCHED, true>);
  else
    kfn = odd ? reinterpret_cast<const void *>(restrict_stream_k<TF, kCI, kCJ, kMT, kKCMax, WPS, true, SCHED, false>)
              : reinterpret_cast<const void *>(restrict_stream_k<TF, kCI, kCJ, kMT, kKCMax, WPS, false, SCHED, false>);
  // coarse planes per chunk: a chunk of kc coarse planes walks ~2 kc + 3 fine planes: minimise (rounds of
  // workgroups at the kernel's occupancy) x (planes walked); the chunk's z tables must fit their LDS
  // arrays (kc <= kKCMax)
  static int occ[2] = {0, 0}, epoch[2] = {0, 0};
  if (ndsm::first_in_epoch(epoch[odd])) {
    NDSM_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int o = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kfn, kCI * kCJ, lds_bytes) != hipSuccess || o < 1) o = 1;
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
  NDSM_HIP(hipLaunchKernel(kfn, dim3(nblk), dim3(kCI * kCJ), args, lds_bytes, stream()));
  return 0;
}

template <typename TF>
static int launch_rs_t(const ndsmk_xfer *x, const TF *r_f, double *rhs_c, double *u_c) {
  int v = rs_variant();
  // The scheduled forms hold a chunk's z windows in a fixed-size table: at most four coarse windows per fine plane
  // (four weight slots, four accumulators) and at most 2 * kKCMax + 16 fine planes per chunk (KPL records).  The host
  // measures both for a level pair (ndsmh_mg.f90: stream_restrict_applies); they are checked HERE, against the
  // kernel's limits - a descriptor that does not carry the numbers takes the table walk instead.
  {
    const int wmax = (x->stream_ok >> 8) & 255, smax = (x->stream_ok >> 16) & 32767;
    if (v != 5 && (!(x->stream_ok & 4) || wmax < 1 || wmax > 4 || smax < 1 || smax > 2 * kKCMax + 16)) v = 5;
    // the DMA form: fp64 fine planes whose rows are 16-byte aligned and whose byte offsets fit the descriptor
    if (v == 10 && (!std::is_same<TF, double>::value || (x->nf[0] & 1) ||
                    (int64_t)x->nf[0] * x->nf[1] * (int64_t)sizeof(double) >= (int64_t)0x80000000ll))
      v = 8;
  }
  if (v == 10) {
    if constexpr (std::is_same<TF, double>::value) return launch_rs_v<TF, true, true>(x, r_f, rhs_c, u_c);
  }
  if (v == 8) return launch_rs_v<TF, true, false>(x, r_f, rhs_c, u_c);
  return launch_rs_v<TF, false, false>(x, r_f, rhs_c, u_c);
}

int launch_restrict_stream(const ndsmk_xfer *x, const double *r_f, double *rhs_c, double *u_c) {
  return launch_rs_t<double>(x, r_f, rhs_c, u_c);
}
int launch_restrict_stream_f32(const ndsmk_xfer *x, const float *r_f, double *rhs_c, double *u_c) {
  return launch_rs_t<float>(x, r_f, rhs_c, u_c);
}

}  // namespace ndsm
