// Fused red+black Gauss-Seidel sweeps: ONE launch reads u (and rhs) once, performs S full sweeps
// (S = 1 or 2, temporal blocking) and writes the swept field once - OUT OF PLACE: workgroups read
// each other's halo from the input array, so the input must stay intact for the whole launch and
// the caller ping-pongs (or rotates) arrays.  The two colour passes of smooth.hip move ~48 B/LUP
// with half-used lines; this moves ~24/S B/LUP plus halo.
//
// Same arithmetic as rbgs3_color (ndsm_optimized.f90:103-167): a black point only reads red
// neighbours of the SAME sweep and a red point only reads black neighbours of the PREVIOUS state,
// so any schedule that respects those two dependencies gives bit-identical results.  Schedule
// used here (per workgroup; the comment on rbgs3_fused_k has the pipeline in detail):
//
//   * an (x,y) tile of TXH x TYH points, of which the inner TXI x TYI are owned (written back) and
//     a ring as deep as the pipeline is halo, updated redundantly as far as it stays valid (a halo
//     that reaches the physical face stays valid throughout and is owned as well);
//   * the tile is streamed through a z chunk [zs, ze) as a 2S-stage pipeline skewed along z:
//     iteration k runs stage t (even red, odd black) on plane k - t.  The z+1 neighbour comes
//     from the register the previous stage just produced, everything else from LDS planes kept
//     element-planar (all "element 0" of the x-pairs, then all "element 1") so that a stage's
//     accesses are conflict free;
//   * every thread owns two x-PAIRS (16-byte aligned): each global load / store is 16 B per lane,
//     rows are contiguous, and a pair holds exactly one red and one black point per plane - no
//     divergence between colours;
//   * plane k+2 (and its rhs) is requested before plane k is computed; global stores are issued
//     right after the wait that consumes those loads (a vmcnt wait also waits for anything issued
//     since the load it is meant for);
//   * work items (tile, chunk) are laid out so that the y-neighbouring tiles, which share halo
//     rows, sit on the same XCD (blockIdx % 8) and hit its L2; the number of z chunks minimises
//     rounds of workgroups x planes walked.
//
// Modes (one template parameter): 0 plain, 1 + residual of the result as one more stage,
// 2 + convergence metric against the iterate the V-cycle started from, 3 + coarse-grid
// correction added to every plane as it is loaded.  T = double is the reference's arithmetic;
// T = float serves the correction equation of the mixed-precision mode.
//
// Requirements of this path (otherwise smooth.hip runs): 3-D, >= 16 x 16 points per plane;
// z-slabs enter through k0 / nzg / zown0 / zown1 of the grid descriptor.
//
// Odd nx (template parameter ODD; 2^k + 1 grids, and every other level of NDSM's floor-halving
// hierarchy): the pair grid still starts at an even index, so the last column nx-1 is element 0 of
// a pair whose element 1 lies outside the row.  That element is kept as a GHOST that mirrors
// column nx-2: it is loaded from there, carries that column's flags, stage budget and right-hand
// side, and takes column nx-3 as its outer neighbour - so it receives the same update, from the
// same operands, in the same stage ((a + b) commutes) and stays equal to u(nx-2) bit for bit.
// Column nx-1 thus finds its Neumann mirror value where any interior point finds its x+1
// neighbour, and the inner loop is unchanged; only loads (rows are 8-byte aligned now), stores
// (the ghost is never written back) and the per-slot constants differ.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

// Plane loads and stores go through BUFFER instructions: one descriptor per plane
// (scalar registers), a 32-bit byte offset per lane.  The range check stands in for every predicate - a lane
// outside the domain (or not owned, for stores) carries an offset beyond the plane and a plane outside the
// chunk gets a descriptor of zero records: such loads return 0 and such stores are dropped.  No load or store
// sits under a branch, no 64-bit address arithmetic per lane.  (Round 3 also tried requesting planes TWO steps
// ahead with counted waits, a spare LDS plane that removes one of the two barriers per step, and both together:
// 0 to -4 % - the pass was never waiting for memory latency.  Result stores carry the non-temporal hint: +2 %.)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
constexpr unsigned kDeadLane = 0x80000000u;   // >= any plane's byte count (launch_fused_t checks)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// MODE 3: tables of the interpolation from the next coarser level (ndsmk_xfer's prolongation half)
struct ProlArgs {
  const double *uc;          // coarse u (whole level)
  const int32_t *plo[3];     // per fine index: lower bracket in the coarse axis
  const double *pwl[3], *pwh[3];
  int ncx, ncy, ncz;
  // z windows (z-slabs; all zero / whole levels otherwise): global index of the fine array's plane 0,
  // global fine nz, global index of uc's plane 0 and the number of coarse planes uc holds
  int fk0, nzf, ck0, nczw;
};

struct FusedPlan {
  int ntx, nty, nzc;  // tiles in x, y and chunks in z
  int zc;             // planes per chunk
  int nwork;          // ntx * nty * nzc
};

// an x-pair of the working precision T (double: the reference's arithmetic; float: the
// correction equation of the mixed-precision mode); elements are picked with selects (a
// runtime-indexed register array would be demoted to scratch memory)
template <typename T>
struct P2 {
  T x, y;
};
template <typename T>
struct Vec2;
template <>
struct Vec2<double> {
  using type = double2;
};
template <>
struct Vec2<float> {
  using type = float2;
};

template <typename T>
__device__ __forceinline__ P2<T> ldbuf(__amdgpu_buffer_rsrc_t r, unsigned off) {
  P2<T> v;
  if constexpr (sizeof(T) == 8) {
    const v4u_t a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    __builtin_memcpy(&v, &a, 16);
  } else {
    const v2u_t a = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    __builtin_memcpy(&v, &a, 8);
  }
  return v;
}
template <typename T, int AUX>
__device__ __forceinline__ void stbuf(__amdgpu_buffer_rsrc_t r, unsigned off, const P2<T> &v) {
  if constexpr (sizeof(T) == 8) {
    v4u_t a;
    __builtin_memcpy(&a, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(a, r, off, 0, AUX);
  } else {
    v2u_t a;
    __builtin_memcpy(&a, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(a, r, off, 0, AUX);
  }
}
// one element (the last column of an odd-nx level, stored by itself)
template <typename T, int AUX>
__device__ __forceinline__ void stbuf1(__amdgpu_buffer_rsrc_t r, unsigned off, const T v) {
  if constexpr (sizeof(T) == 8) {
    v2u_t a;
    __builtin_memcpy(&a, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(a, r, off, 0, AUX);
  } else {
    unsigned a;
    __builtin_memcpy(&a, &v, 4);
    __builtin_amdgcn_raw_buffer_store_b32(a, r, off, 0, AUX);
  }
}
// by value: selects on values, never on addresses
template <typename T>
__device__ __forceinline__ T pick(const P2<T> a, int e) {
  return e ? a.y : a.x;
}

// geometry of slot s of this thread, recomputed where needed (cheap integer
// ops) instead of being held in registers across the z loop
template <int TXH, int TYH, int NT, int HX, int HY>
struct Slot {
  int li, lj, i, j, lo;
  bool live, in;
  __device__ __forceinline__ Slot(int tid, int s, int x0, int y0, int nx, int ny) {
    constexpr int NPX = TXH / 2;
    const int p = tid + NT * s;
    lj = p / NPX;
    li = 2 * (p - lj * NPX);
    i = x0 + li;
    j = y0 + lj;
    live = p < NPX * TYH;
    in = live && i >= 0 && i + 1 < nx && j >= 0 && j < ny;
    lo = live ? p : 0;  // pair number = index inside each half of an LDS plane
  }
};

// S full sweeps (2S colour stages) in ONE pass over the data - temporal blocking.
//
// Stage t (t = 0 .. 2S-1; even = red, odd = black) of iteration k updates plane
// k - t, so the 2S stages form a software pipeline skewed along z: a plane is
// loaded once, visited by the 2S stages in 2S consecutive iterations while it
// sits in LDS, and stored once.  HBM traffic per sweep therefore drops to ~1/S
// of the single-sweep kernel's (plus halo: the loaded tile carries a ring of 2S
// points, ring r being valid up to stage r-1, and a chunk warms up over 2S planes).
//
// Dependencies inside one iteration need NO barrier between stages:
//   * stage t reads, in plane k-t, only points of the OTHER colour - last written
//     by stage t-1 one iteration (= one barrier) ago;
//   * its z+1 neighbour (plane k-t+1, same x,y => same thread) is what stage t-1
//     produced a moment ago in this thread (register), its z-1 neighbour (plane
//     k-t-1) is read from LDS before this thread's stage t+1 overwrites it;
//   * every stage of an iteration updates the SAME element of the thread's
//     x-pairs (colour and plane parity flip together), so no two stages of two
//     threads ever touch the same LDS word.
// Plane p lives in LDS buffer p mod 2S; the incoming plane k+1 replaces plane
// k-2S+1, which the last stage has just finished and stored.
//
// RES: one more pipeline stage evaluates the residual r = rhs - L u of the swept
// field (same expression as residual.hip) on plane k - 2S, one plane behind the
// store, and writes it to rout - the sweep + residual pair at the bottom of the
// V-cycle's descent then moves 24 (Laplace) instead of 16 + 16 B per point.  The
// stage needs plane k-2S with its in-plane neighbours (one more LDS buffer) and
// the thread's own pairs of planes k-2S+1 and k-2S-1 (registers); halo and chunk
// warm-up grow by one.
//
// MODE 2 (MET): the launch that ends a V-cycle also evaluates the convergence metric of
// update_u (ndsm_multigrid_core.f90:1077-1122), max and sum of |u_new - u_prev| over the points
// it stores, against the iterate the cycle started from (prev; the V-cycle driver keeps that
// buffer untouched) - per-workgroup partials, folded by fold_metric_k.  Replaces a separate
// 24 B/pt pass (read u, read prev, write prev) by one more 8 B/pt read here.
//
// MODE 3 (PROL): the launch that starts the post-smoothing adds the coarse-grid correction to
// every plane as it is loaded - u + P u_c never makes a round trip through HBM of its own
// (coarse_to_fine's interpolate + add_correction, ndsm_multigrid_core.f90:593-684; same
// expressions and order as prolong_add_k: z first, then y, then x).  Per fine plane the
// workgroup forms the z-interpolated coarse plane of its tile ONCE (one or two coarse points per
// thread, their values of the two bracketing coarse planes rolling through registers) and
// parks it in LDS; each fine point then needs four LDS reads and nine flops.
// LVL1 changes nothing but the kernel's name: launches on levels of >= 64 M points get a symbol of
// their own, so that a rocprofv3 --stats summary does not average the level-1 launches (the
// ones bench.py's roofline line is quoted on) with the much shorter ones of the coarser levels.
template <typename T, int S, int TXH, int TYH, int NT, int WPS, bool RHS0, int MODE, bool LVL1, bool ODD>
__global__ __launch_bounds__(NT, WPS) void rbgs3_fused_k(const T *__restrict__ u, T *__restrict__ uout,
                                                         const T *__restrict__ rhs, T *__restrict__ rout,
                                                         const T *__restrict__ prev, double *__restrict__ part,
                                                         ndsmk_grid g, FusedPlan pl, ProlArgs pa) {
  using d2 = P2<T>;
  constexpr bool RES = MODE == 1, MET = MODE == 2, PROL = MODE == 3;
  constexpr int SZ = (int)sizeof(T);
  constexpr bool DEFER = !RES;
  const T gw0 = (T)g.w[0], gw1 = (T)g.w[1], gw2 = (T)g.w[2], gw1i = (T)g.w1, gwc = (T)g.wc;
  constexpr int NST = 2 * S;                 // smoothing stages
  constexpr int NSTG = RES ? NST + 1 : NST;  // pipeline depth = halo width
  constexpr int NB = NSTG;                   // LDS planes: one per pipeline stage
  constexpr int STAUX = 2;                   // result stores: non-temporal (never re-read by this launch)
  constexpr int NPX = TXH / 2;
  constexpr int NPAIR = NPX * TYH;
  constexpr int NS = (NPAIR + NT - 1) / NT;
  constexpr int HX = (NSTG + 1) & ~1;        // x halo: even, pairs stay 16-byte aligned
  constexpr int TXI = TXH - 2 * HX, TYI = TYH - 2 * NSTG;
  constexpr int PLANE = TXH * TYH;
  // LDS layout.  The points of a plane fall into two CLASSES by the parity of i + j (+ the slab / colour
  // offsets): a pair holds one point of each, and in plane-step k every stage updates the class-(k & 1) point of
  // every pair (colour and plane parity flip together).  The LDS keeps the classes apart - first the class-0
  // points of all NB planes, then the class-1 points ("half-major") - so that WHICH half a stage reads and
  // writes is the same for every lane and known per step: the updated point, its z neighbours and what the
  // next stage needs of it sit in half k & 1, its four in-plane neighbours in the other half.  (Element e of a
  // pair whose parity bit is par lies in half e ^ par.)  Consecutive lanes touch consecutive words: no bank
  // conflicts - interleaved pairs had 2-way conflicts on every access (42 % of the LDS cycles).
  constexpr int HALF = NPAIR * SZ;   // bytes of one class of one plane
  constexpr int HSTR = NB * HALF;    // class h of the plane in buffer b starts at h * HSTR + b * HALF
  using SlotT = Slot<TXH, TYH, NT, HX, NSTG>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *const lds = reinterpret_cast<T *>(lds_raw);

  // ---- which (tile, chunk): consecutive y tiles share an XCD ---------
  const int nb8 = gridDim.x >> 3;
  const int w = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
  if (w >= pl.nwork) {
    if (MET && threadIdx.x == 0) part[2 * blockIdx.x] = part[2 * blockIdx.x + 1] = 0.0;
    return;
  }
  const int ty = w % pl.nty;
  const int t2 = w / pl.nty;
  const int tx = t2 % pl.ntx;
  const int cz = t2 / pl.ntx;

  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const int nxs = ODD ? nx + 1 : nx;  // row length in whole pairs (ODD: the last pair's element 1 is the ghost)
  const int x0 = tx * TXI - HX, y0 = ty * TYI - NSTG;
  const int zs = g.zown0 + cz * pl.zc;
  const int ze = min(zs + pl.zc, g.zown1);
  // first plane walked: NSTG warm-up planes below the chunk.  GROUPS: it is rounded DOWN to a multiple of the number
  // of copies of the plane-step (below) and the loop runs whole groups of copies in a fixed order - no dispatch
  // on k, no register shuffling where the copies meet (18 moves per step in the correction kernel) - at the
  // price of up to NCOPY - 1 more steps at either end, which load nothing.  Not for the sweep + residual kernel:
  // six copies there, and the pass is bound by memory, not by instructions.
  constexpr int NCOPY = (NSTG % 2 == 0) ? NSTG : 2 * NSTG;
  constexpr bool GROUPS = !RES;
  const int kl = max(zs - NSTG, 0);        // first plane ever loaded
  const int ks = GROUPS ? (kl / NCOPY) * NCOPY : kl;
  const int ke = min(ze - 1 + NSTG, nz - 1);  // last plane ever loaded
  const size_t sz = (size_t)nx * (size_t)ny;
  const int tid0 = (int)threadIdx.x;
  const int tid = tid0;
  const int fp = g.first_par & 1;
  // tile edges that coincide with the physical boundary do not shrink the valid region
  const bool openxh = x0 + TXH < nx, openyh = y0 + TYH < ny;
  // descriptor of plane k of a field; ok == false: zero records (every load returns 0, every store is dropped)
  const unsigned plane_bytes = (unsigned)(sz * (size_t)SZ);
  auto rsrc_of = [&](const T *base, int k_, bool ok) {
    return plane_rsrc(base + sz * (size_t)(ok ? k_ : 0), ok ? plane_bytes : 0u);
  };
  // A pair through descriptor r, ONE 16-byte (8-byte) access per lane.  Odd nx: rows are only sizeof(T)-aligned,
  // so half of these accesses straddle a 16-byte boundary - still cheaper than two element-sized ones - and the
  // pair that holds the last column nx-1 (element 0) keeps a GHOST of column nx-2 as element 1 (header): it is
  // LOADED one element to the left and swapped (gh: the lane is that pair), and STORED as element 0 only (og).
  auto ldp = [&](__amdgpu_buffer_rsrc_t r, unsigned off, bool gh) {
    d2 v = ldbuf<T>(r, off);
    if constexpr (ODD) {
      d2 w;
      w.x = gh ? v.y : v.x;
      w.y = gh ? v.x : v.y;
      return w;
    }
    return v;
  };
  auto stp = [&](__amdgpu_buffer_rsrc_t r, unsigned off, unsigned og, const d2 &v) {
    stbuf<T, STAUX>(r, off, v);
    if constexpr (ODD) stbuf1<T, STAUX>(r, og, v.x);
  };

#define NDSM_LOAD_PLANE(base, k, dst)                                                                  \
  do {                                                                                                  \
    const auto r_ = rsrc_of(base, k, (k) >= kl && (k) <= ke);                                           \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) dst[s_] = ldp(r_, scs[s_].ldo, ODD && (scs[s_].fl & 16)); \
  } while (0)

  // Register window per slot (everything else is re-read from LDS):
  //   nxt = plane k+1 (arrived), nn = plane k+2 (in flight)
  //   mLe = the point of plane k-2S (final) the last stage needs as z-1 neighbour
  //   rw[t] = rhs of plane k-t IN CLASS ORDER (.x the class-0 point), rn = rhs of plane k+1 (in flight)
  d2 nxt[NS], nn[NS];
  //   RES: f1 / f2 = the final pairs of planes k-2S and k-2S-1
  d2 rw[RHS0 ? 1 : NS][RHS0 ? 1 : NSTG], rn[RHS0 ? 1 : NS];
  T mLe[RES ? 1 : NS];
  d2 f1[RES ? NS : 1], f2[RES ? NS : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    nxt[s].x = nxt[s].y = 0.0;
    nn[s].x = nn[s].y = 0.0;
    if (RES) {
      f1[RES ? s : 0].x = f1[RES ? s : 0].y = 0.0;
      f2[RES ? s : 0].x = f2[RES ? s : 0].y = 0.0;
    } else {
      mLe[RES ? 0 : s] = 0.0;
    }
  }
  if (!RHS0) {
#pragma unroll
    for (int s = 0; s < (RHS0 ? 1 : NS); ++s) {
      rn[s].x = rn[s].y = 0.0;
#pragma unroll
      for (int t = 0; t < (RHS0 ? 1 : NSTG); ++t) rw[s][t].x = rw[s][t].y = 0.0;
    }
  }

  // ---- per-slot constants of the z loop ----
  // The pass is bound by INSTRUCTION ISSUE - a wave issues one instruction every four cycles whatever its kind -
  // not by HBM, LDS bandwidth or the fp64 rate: round 2's plane-step spent ~70 instructions per point update, 9
  // of them arithmetic, re-deriving per stage which LDS plane it works on, which element of the pair is its
  // own and where that element's neighbours are.  None of that depends on k once the plane loop is unrolled
  // over the LDS buffers (NCOPY copies of the step, below): buffer and class of every access are compile-time
  // constants, every LDS address is one of the per-slot registers set up here plus an immediate offset, every
  // per-lane predicate a lane mask in scalar registers.  Per update that leaves six LDS reads, nine flops
  // and one masked write.
  struct SC {
    int l0, l1;    // where the pair's elements 0 / 1 live in an LDS plane: pair offset + HSTR * (the element's class)
    unsigned ldo;  // byte offset of the pair inside a global plane for loads (kDeadLane outside the domain; the ghost
                   // pair of an odd-nx level: one element to the left, ldp) ...
    unsigned sto;  // ... and for stores (owned pairs only; the ghost pair: kDeadLane, its element 0 goes through stg)
    unsigned stg;
    int fl;        // bit 0 in-domain, 1 owned, 2/3 element 0/1 inside the x-y update bounds, 4 (ODD) the pair
                   // of column nx-1 whose element 1 is the ghost, 6 the pair's parity: class of its element 0
  };
  //   cL[c]  the pair inside class c (byte offset into the plane region, buffer offset not included);
  //   cX[c] / cYH[c] / cYL[c]  the outer x neighbour and the y neighbours of the pair's class-c point (they lie in
  //   class 1-c; mirrored at the physical faces; the pair partner where the neighbour is outside the tile)
  //   mU[c]  lanes whose class-c point lies inside the x-y update bounds and the domain;  mP  lanes whose
  //   element 0 is the class-1 point;  mO  lanes whose pair is owned;  mE[e]  (residual) element e in bounds
  SC scs[NS];
  int cL[NS][2], cX[NS][2], cYH[NS][2], cYL[NS][2];
  unsigned long long mU[NS][2], mP[NS], mO[NS], mE[RES ? NS : 1][2];
  static_assert(NS <= 2, "tile / thread-count combinations with more than two pairs per thread are not built");
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const SlotT q_(tid, s, x0, y0, nxs, ny);
    const int lo = SZ * q_.lo;
    // rows whose y neighbours fall outside the loaded tile (ring 0) read their pair partner instead - any
    // valid address: what such a point becomes is never used
    const bool yok = q_.lj + (q_.j == 0 ? 1 : -1) >= 0 && q_.lj + (q_.j == ny - 1 ? -1 : 1) < TYH;
    const int yl = !yok ? 0 : ((q_.j == 0) ? SZ * NPX : -SZ * NPX);    // mirrored at the physical faces
    const int yh = !yok ? 0 : ((q_.j == ny - 1) ? -SZ * NPX : SZ * NPX);
    const bool yin = q_.j >= g.lb[1] && q_.j <= g.ub[1];
    const bool in0 = yin && q_.i >= g.lb[0] && q_.i <= g.ub[0];
    const bool ghost = ODD && q_.in && q_.i == nx - 1;   // element 1 mirrors column nx-2 (header)
    const int i1 = ghost ? q_.i - 1 : q_.i + 1;           // the column element 1 stands for
    const bool in1 = yin && i1 >= g.lb[0] && i1 <= g.ub[0];
    const bool mir0 = q_.i == 0, mir1 = q_.i + 1 == nx - 1;
    // outer x neighbour of element 0 / 1, as a pair offset in the OTHER class: the pair partner where mirrored
    // or where the neighbour is outside the tile
    const int xlo = (!mir0 && q_.li - 1 >= 0) ? lo - SZ : lo;
    int xhi = (!mir1 && q_.li + 2 < TXH) ? lo + SZ : lo;
    if (ghost) xhi = lo - SZ;                             // column nx-3: element 0 of the previous pair (li >= HX >= 2)
    // owned = written back by this workgroup: the inner TXI x TYI points, and - in the last tile of a
    // row / column of tiles, whose halo reaches the physical face (launch_cfg counts tiles that way) -
    // the halo points up to that face too: they stay valid through every stage, nothing shrinks there
    const bool own = q_.in && q_.li >= HX && q_.lj >= NSTG && (q_.li < TXH - HX || !openxh) && (q_.lj < TYH - NSTG || !openyh);
    const int par = (q_.i + q_.j + g.k0 + fp) & 1;
    SC c;
    const unsigned gob = (unsigned)(q_.i + nx * q_.j) * SZ;
    c.fl = (q_.in ? 1 : 0) | (own ? 2 : 0) | (in0 ? 4 : 0) | (in1 ? 8 : 0) | (ghost ? 16 : 0) | (par ? 64 : 0);
    c.l0 = lo + HSTR * par;
    c.l1 = lo + HSTR - HSTR * par;
    c.ldo = q_.in ? (ghost ? gob - SZ : gob) : kDeadLane;
    c.sto = (own && !ghost) ? gob : kDeadLane;
    c.stg = (own && ghost) ? gob : kDeadLane;
    scs[s] = c;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const int e = cl ^ par;   // the element of class cl
      cL[s][cl] = lo + cl * HSTR;
      cX[s][cl] = (e ? xhi : xlo) + (1 - cl) * HSTR;
      cYH[s][cl] = lo + yh + (1 - cl) * HSTR;
      cYL[s][cl] = lo + yl + (1 - cl) * HSTR;
      // NOT part of the mask: how many stages a halo point stays valid for (ring r: r stages).  A point beyond
      // that takes a meaningless value, but no point still inside ITS count ever reads it - a neighbour one
      // ring further in has one stage more and by then reads what the previous stage left - and halo points are
      // never stored.  The same holds along z for the chunk's warm-up and drain planes.  What must never change
      // is DATA: points outside the update bounds and the domain.
      mU[s][cl] = __builtin_amdgcn_ballot_w64(q_.live && q_.in && (e ? in1 : in0));
    }
    mP[s] = __builtin_amdgcn_ballot_w64(par != 0);
    mO[s] = __builtin_amdgcn_ballot_w64(own);
    if (RES) {
      mE[RES ? s : 0][0] = __builtin_amdgcn_ballot_w64(in0);
      mE[RES ? s : 0][1] = __builtin_amdgcn_ballot_w64(in1);
    }
  }
  char *const ldsb = reinterpret_cast<char *>(lds);
#define LDSD(off) (*reinterpret_cast<T *>(ldsb + (off)))
  // element order <-> class order of a pair (what a lane whose element 0 is the class-1 point swaps)
  auto by_class = [&](const d2 v, int s_) {
    const bool sw = __builtin_amdgcn_inverse_ballot_w64(mP[s_]);
    d2 r;
    r.x = sw ? v.y : v.x;
    r.y = sw ? v.x : v.y;
    return r;
  };


  // ---- PROL: coarse footprint of the tile, per-thread interpolation constants ----
  constexpr int CW = TXH / 2 + 3, CH = TYH / 2 + 3;  // coarse points the tile can touch (ratio >= ~1.97)
  constexpr int NCS = PROL ? (CW * CH + NT - 1) / NT : 1;
  // z-interpolated coarse planes, double buffered: the plane of fine plane kf sits in tile kf & 1,
  // parked one iteration before it is needed so that the correction runs before the barrier
  double *const czt0 = reinterpret_cast<double *>(ldsb + NB * PLANE * SZ);
  // the x / y interpolation weights of the tile's columns and rows sit in LDS behind the parked planes
  // ([NPX][wl0, wh0, wl1, wh1], [TYH][wl, wh]; zero outside the domain): per slot they would take 24 registers
  // the kernel does not have - three 16-byte LDS reads per pair and plane instead
  double *const pwx = czt0 + 2 * CW * CH;
  double *const pwy = pwx + 4 * NPX;
  int p_xa[PROL ? NS : 1], p_ya[PROL ? NS : 1];   // the slot's entries (byte offsets into the LDS)
  int cx0 = 0, cy0 = 0, kcur = 0;
  // where the lower-left corner of element 0's / element 1's bracket sits in the parked coarse tile (tile 0; byte
  // offsets into the LDS): the other three corners and the second tile are immediate offsets from there
  int p_c0[PROL ? NS : 1], p_c1[PROL ? NS : 1];
  double c_lo[NCS], c_hi[NCS], c_nx[NCS];
  unsigned c_off[NCS];   // byte offset of the thread's coarse points inside a coarse plane (kDeadLane: outside the level)
  if (PROL) {
    const int ia = max(x0, 0), ja = max(y0, 0);
    cx0 = pa.plo[0][ia];
    cy0 = pa.plo[1][ja];
#pragma unroll
    for (int s = 0; s < (PROL ? NS : 1); ++s) {
      const SlotT q(tid, s, x0, y0, nxs, ny);
      const int czb = (int)(reinterpret_cast<char *>(czt0) - ldsb);
      p_c0[s] = p_c1[s] = czb;
      p_xa[s] = (int)(reinterpret_cast<char *>(pwx) - ldsb) + 32 * (q.live ? q.li / 2 : 0);
      p_ya[s] = (int)(reinterpret_cast<char *>(pwy) - ldsb) + 16 * (q.live ? q.lj : 0);
      if (q.in) {
        const int jl = pa.plo[1][q.j] - cy0;
        int il[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ih = (ODD && q.i + h >= nx) ? q.i - 1 : q.i + h;   // the ghost gets column nx-2's correction
          il[h] = pa.plo[0][ih] - cx0;
        }
        p_c0[s] = czb + 8 * (jl * CW + il[0]);
        p_c1[s] = czb + 8 * (jl * CW + il[1]);
      }
    }
    for (int t = tid; t < NPX + TYH; t += NT) {
      if (t < NPX) {
        const int i = x0 + 2 * t;                       // the pair of columns i, i+1 (same test as Slot::in)
        const bool ok = i >= 0 && i + 1 < nxs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ih = (ODD && i + h >= nx) ? i - 1 : i + h;
          pwx[4 * t + 2 * h] = ok ? pa.pwl[0][ih] : 0.0;
          pwx[4 * t + 2 * h + 1] = ok ? pa.pwh[0][ih] : 0.0;
        }
      } else {
        const int j = y0 + (t - NPX);
        const bool ok = j >= 0 && j < ny;
        pwy[2 * (t - NPX)] = ok ? pa.pwl[1][j] : 0.0;
        pwy[2 * (t - NPX) + 1] = ok ? pa.pwh[1][j] : 0.0;
      }
    }
#pragma unroll
    for (int c = 0; c < NCS; ++c) {
      const int idx = tid + NT * c;
      const int a = idx % CW, b = idx / CW;
      const bool c_ok = b < CH && cx0 + a < pa.ncx && cy0 + b < pa.ncy;
      c_off[c] = c_ok ? (unsigned)((cx0 + a) + pa.ncx * (cy0 + b)) * 8u : kDeadLane;
      c_lo[c] = c_hi[c] = c_nx[c] = 0.0;
    }
  }
  const size_t csz = PROL ? (size_t)pa.ncx * (size_t)pa.ncy : 0;
  // coarse planes kcur, kcur+1 (and, prefetched, kcur+2) of this thread's coarse points: buffer loads through a
  // descriptor per coarse plane - zero records for a plane outside the coarse window, an offset beyond the plane
  // for a point outside the level, so that what must read as zero does, without a branch or a select (a load under
  // a branch is waited for on the spot: measured 803 us per launch against 531 us without the correction)
  const unsigned cbytes = PROL ? (unsigned)(csz * 8) : 0u;
  auto prol_load = [&](int kc, double *dst) {
    const bool ok = kc >= pa.ck0 && kc < pa.ck0 + pa.nczw;
    const auto rc_ = plane_rsrc(pa.uc + csz * (size_t)(ok ? kc - pa.ck0 : 0), ok ? cbytes : 0u);
#pragma unroll
    for (int c = 0; c < NCS; ++c) {
      const v2u_t a_ = __builtin_amdgcn_raw_buffer_load_b64(rc_, c_off[c], 0, 0);
      __builtin_memcpy(&dst[c], &a_, 8);
    }
  };
  // advance the rolling coarse planes to the bracket of fine plane kf and park its z-interpolated
  // coarse plane in LDS (the caller puts a barrier between this and prol_corr)
  // (the z tables are read one plane ahead: a scalar load issued and consumed in the same
  // iteration costs its whole latency on every plane)
  int zt_kc = 0;
  double zt_wl = 0.0, zt_wh = 0.0;
  // the z tables of the chunk's planes sit in LDS (a lookup is then an LDS read; from global memory the
  // compiler turns the uniform value into a scalar right at the load and waits for it there - one memory
  // round trip per plane); chunks taller than the LDS table keep the global lookups
  constexpr int ZT = PROL ? 160 : 1;
  __shared__ int s_zkc[ZT];
  __shared__ double s_zwl[ZT], s_zwh[ZT];
  const bool zt_lds = PROL && (ke - ks + 4 <= ZT);
  if (PROL && zt_lds) {
    for (int t = tid; t < ke - ks + 4; t += NT) {
      const int kq = min(max(ks + t + pa.fk0, 0), pa.nzf - 1);   // table index = global fine plane
      s_zkc[t] = pa.plo[2][kq];
      s_zwl[t] = pa.pwl[2][kq];
      s_zwh[t] = pa.pwh[2][kq];
    }
    __syncthreads();
  }
  auto prol_ztab = [&](int kf) {
    if (zt_lds) {
      zt_kc = s_zkc[kf - ks];
      zt_wl = s_zwl[kf - ks];
      zt_wh = s_zwh[kf - ks];
    } else {
      const int kq = min(max(kf + pa.fk0, 0), pa.nzf - 1);
      zt_kc = pa.plo[2][kq];
      zt_wl = pa.pwl[2][kq];
      zt_wh = pa.pwh[2][kq];
    }
  };
  auto prol_stage = [&](int kf) {
    double *const czt = czt0 + (kf & 1) * (CW * CH);
    const int kc = zt_kc;
    const double wlz = zt_wl, whz = zt_wh;
    prol_ztab(kf + 1);
    // the bracket moves up by at most one coarse plane per fine plane.  c_nx was requested one stage
    // ago at the latest; it is consumed (masked) here EVERY stage - taken over when the bracket moves,
    // dropped otherwise - and requested again right away, for the plane the next move will need
    const bool adv = kc > kcur;
#pragma unroll
    for (int c = 0; c < NCS; ++c) {
      c_lo[c] = adv ? c_hi[c] : c_lo[c];
      c_hi[c] = adv ? c_nx[c] : c_hi[c];
    }
    kcur = adv ? kc : kcur;
    prol_load(kcur + 2, c_nx);
#pragma unroll
    for (int c = 0; c < NCS; ++c) {
      const int idx = tid + NT * c;
      if (idx < CW * CH) czt[idx] = whz * c_lo[c] + wlz * c_hi[c];  // last dimension first (ndsm_interp.f90:128-154)
    }
  };
  // u + P u_c for this thread's pairs of the fine plane whose coarse plane is parked in tile `tile` (kf & 1: a
  // compile-time constant inside the plane-step).  Every LDS address is a per-slot register plus an immediate; a
  // pair outside the domain has zero weights and holds +0: no predicate.
  auto prol_corr = [&](d2 *pl_, const int tile) {
    const int tb = tile * (CW * CH * 8);
#pragma unroll
    for (int s = 0; s < (PROL ? NS : 1); ++s) {
      const double2 wy = *reinterpret_cast<const double2 *>(ldsb + p_ya[s]);          // wl, wh of the row
      const double2 wx0 = *reinterpret_cast<const double2 *>(ldsb + p_xa[s]);         // wl, wh of element 0
      const double2 wx1 = *reinterpret_cast<const double2 *>(ldsb + p_xa[s] + 16);    // ... of element 1
      double v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const char *cq = ldsb + (h ? p_c1[s] : p_c0[s]) + tb;
        double f0 = *reinterpret_cast<const double *>(cq), f1 = *reinterpret_cast<const double *>(cq + 8);
        const double f2 = *reinterpret_cast<const double *>(cq + 8 * CW), f3 = *reinterpret_cast<const double *>(cq + 8 * CW + 8);
        f0 = wy.y * f0 + wy.x * f2;
        f1 = wy.y * f1 + wy.x * f3;
        v[h] = (h ? wx1.y : wx0.y) * f0 + (h ? wx1.x : wx0.x) * f1;
      }
      pl_[s].x = (T)((double)pl_[s].x + v[0]);
      pl_[s].y = (T)((double)pl_[s].y + v[1]);
    }
  };

  // ---- prologue: plane ks into its LDS buffer, plane ks+1 into registers ----
  {
    d2 c0[NS];
    NDSM_LOAD_PLANE(u, ks, c0);
    if (!RHS0) {
      d2 r0[NS];
      NDSM_LOAD_PLANE(rhs, ks, r0);
#pragma unroll
      for (int s = 0; s < (RHS0 ? 1 : NS); ++s) rw[s][0] = by_class(r0[s], s);
    }
    NDSM_LOAD_PLANE(u, ks + 1, nxt);
    if (PROL) {  // the two planes loaded here get their correction here
      kcur = pa.plo[2][min(max(ks + pa.fk0, 0), pa.nzf - 1)];
      prol_ztab(ks);
      prol_load(kcur, c_lo);
      prol_load(kcur + 1, c_hi);
      prol_load(kcur + 2, c_nx);
      prol_stage(ks);
      if (ks + 1 <= ke) prol_stage(ks + 1);
      __syncthreads();
      prol_corr(c0, ks & 1);
      if (ks + 1 <= ke) prol_corr(nxt, (ks + 1) & 1);
      __syncthreads();
      if (ks + 2 <= ke) prol_stage(ks + 2);  // for the first iteration; published by the barrier below
    }
    char *const B0 = ldsb + (ks % NB) * HALF;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (tid0 + NT * s < NPAIR) {
        *reinterpret_cast<T *>(B0 + scs[s].l0) = c0[s].x;
        *reinterpret_cast<T *>(B0 + scs[s].l1) = c0[s].y;
      }
    }
  }
  __syncthreads();

  const int klast = ze + NSTG - 2;  // iteration in which the last stage reaches plane ze-1
  double met_mx = 0.0, met_sm = 0.0;
  // z range in which a stage plane is an ordinary one: inside the update bounds and off the mirror faces
  const int zin0 = max(g.lb[2], 1 - g.k0), zin1 = min(g.ub[2], g.nzg - 2 - g.k0);
  // One plane-step.  The copy C of the step serves the iterations k = C (mod NCOPY): plane k-d sits in LDS buffer
  // (C - d) mod NB, every stage updates class C & 1.
  static_assert(NB == NSTG, "one LDS plane per pipeline stage");
  // Running addresses of the planes a step touches - plane k+2 of u, k+1 of rhs, pf of uout / prev, pf-1 of rout -
  // advanced by one plane at the end of every step: two scalar additions per array instead of a 64-bit multiply
  // and its carries (the descriptors were a third of the scalar instructions of a step).  An address outside its
  // array belongs to a plane outside its window: the descriptor then has zero records and nothing is accessed.
  const long long pbs = (long long)sz * SZ;
  const char *a_ld = reinterpret_cast<const char *>(u) + pbs * (ks + 2);
  const char *a_rh = RHS0 ? nullptr : reinterpret_cast<const char *>(rhs) + pbs * (ks + 1);
  const char *a_pv = MET ? reinterpret_cast<const char *>(prev) + pbs * (ks - (NST - 1)) : nullptr;
  char *a_st = reinterpret_cast<char *>(uout) + pbs * (ks - (NST - 1));
  char *a_rs = RES ? reinterpret_cast<char *>(rout) + pbs * (ks - NST) : nullptr;
  auto rsrc_at = [&](const void *at, bool ok) { return plane_rsrc(at, ok ? plane_bytes : 0u); };
  auto plane_step = [&](const int k, auto CT) __attribute__((always_inline)) {
    constexpr int C = decltype(CT)::value;
    constexpr int HK = C & 1;
#define NDSM_BO(d) (((((C - (d)) % NB) + NB) % NB) * HALF)
    // PROL: park the z-interpolated coarse plane of fine plane k+3 (the other tile: plane k+2's
    // was parked an iteration ago and is read further down, before the barrier).  BEFORE this
    // iteration's loads are issued: it consumes a prefetched coarse plane, and a vmcnt wait placed
    // after the new loads would wait for those too.
    if (PROL && k + 3 <= ke) prol_stage(k + 3);
    const int pf = k - (NST - 1);          // the plane that leaves the smoothing stages in this step
    const bool pf_st = pf >= zs && pf < ze;
    const int pr = k - NST;                // RES: the plane whose residual is formed
    const bool pr_st = RES && pr >= zs && pr < ze;
    // MET: the previous iterate of plane pf; plane k+1 of rhs; plane k+2 of u - requested before plane k is touched
    d2 pvh[MET ? NS : 1];
    if (MET) {
      const auto rp = rsrc_at(a_pv, pf_st);
#pragma unroll
      for (int s = 0; s < (MET ? NS : 1); ++s) pvh[s] = ldp(rp, __builtin_amdgcn_inverse_ballot_w64(mO[s]) ? scs[s].ldo : kDeadLane, ODD && (scs[s].fl & 16));
    }
    if (!RHS0) {
      const auto rr_ = rsrc_at(a_rh, k + 1 >= kl && k + 1 <= ke);
#pragma unroll
      for (int s = 0; s < (RHS0 ? 1 : NS); ++s) rn[s] = ldp(rr_, scs[s].ldo, ODD && (scs[s].fl & 16));
    }
    {
      const auto ru = rsrc_at(a_ld, k + 2 >= kl && k + 2 <= ke);
#pragma unroll
      for (int s = 0; s < NS; ++s) nn[s] = ldp(ru, scs[s].ldo, ODD && (scs[s].fl & 16));
    }

    // ---- the stages ----
    // z+1 neighbour of the first stage: the class-HK point of plane k+1, still in registers; the later stages
    // read theirs back from LDS - what the stage before left there, or the old value where it did not write
    T zp[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const d2 nc = by_class(nxt[s], s);
      zp[s] = HK ? nc.y : nc.x;
    }
    // ZB: a stage plane of this step lies outside the z update bounds or on a mirror face (a handful of steps
    // per launch): those stages do not write / take the mirrored neighbour.  Stage planes that do not exist
    // (below plane 0, above the last plane) are outside the bounds.
    // (the read-back goes through a copy of the address the compiler cannot see through: knowing that the word is the
    // one the previous stage may have written, it forwards the written value in a register and reads LDS only in
    // the lanes that did not write - a divergent if / else of five scalar instructions per update, in a kernel
    // that is bound by instruction issue, to save one LDS read)
    int zback[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      zback[s] = cL[s][HK];
      if (RHS0) asm volatile("" : "+v"(zback[s]));   // (the kernels with an rhs window have no register to spare for it)
    }
    auto stages = [&](auto ZBT) __attribute__((always_inline)) {
      constexpr bool ZB = decltype(ZBT)::value;
#pragma unroll
      for (int t = 0; t < NST; ++t) {
        const int bB = NDSM_BO(t), bZ = NDSM_BO(t + 1);
        T oth[NS], xn[NS], yhv[NS], ylv[NS], zm[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (t > 0) zp[s] = LDSD((RHS0 ? zback[s] : cL[s][HK]) + NDSM_BO(t - 1));
          oth[s] = LDSD(cL[s][1 - HK] + bB);
          xn[s] = LDSD(cX[s][HK] + bB);
          yhv[s] = LDSD(cYH[s][HK] + bB);
          ylv[s] = LDSD(cYL[s][HK] + bB);
          // plane k-t-1: its LDS copy, or (last stage) the saved final point
          zm[s] = (t < NSTG - 1) ? LDSD(cL[s][HK] + bZ) : mLe[RES ? 0 : s];
        }
        const int pg = k - t + g.k0;
        const bool wr = !ZB || (k - t >= g.lb[2] && k - t <= g.ub[2]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const T xs = oth[s] + xn[s];  // u(xh) + u(xl)
          const T ys = yhv[s] + ylv[s];
          const T zhv = (ZB && pg == g.nzg - 1) ? zm[s] : zp[s];
          const T zlv = (ZB && pg == 0) ? zp[s] : zm[s];
          const T zsum = zhv + zlv;
          const T rr = RHS0 ? (T)0 : (HK ? rw[RHS0 ? 0 : s][RHS0 ? 0 : t].y : rw[RHS0 ? 0 : s][RHS0 ? 0 : t].x);
          const T unew = xs * gw0 + ys * gw1 + zsum * gw2 - rr;
          const T nw = gw1i * unew;
          if (wr && __builtin_amdgcn_inverse_ballot_w64(mU[s][HK])) LDSD(cL[s][HK] + bB) = nw;
        }
      }
    };
    if (pf >= zin0 && k <= zin1)
      stages(std::false_type());
    else
      stages(std::true_type());

    // PROL: the correction of the plane that arrived during the stages (its coarse plane was parked
    // one iteration ago), done here so that it overlaps with other waves' stages
    if (PROL && k + 2 <= ke) prol_corr(nn, C & 1);   // (k + 2 has the parity of k)

    // Plane pf has passed its last stage: it goes to HBM; the next step's last stage needs its class-(1-HK)
    // point.  DEFER: it is STORED only after the window shift below has consumed the loads of plane k+2: vmcnt
    // counts stores as well, and a store issued just before that wait would put its whole round trip on the
    // critical path of every plane (-5 % on the two-sweep launch; the sweep+residual launch stores at once).
    d2 finh[NS], resh[RES ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      finh[s].x = LDSD(scs[s].l0 + NDSM_BO(NST - 1));
      finh[s].y = LDSD(scs[s].l1 + NDSM_BO(NST - 1));
      if (!RES) {
        const d2 fc = by_class(finh[s], s);
        mLe[RES ? 0 : s] = HK ? fc.x : fc.y;
      }
    }
    if (RES) {
      // residual of plane pr: centre f1, below f2, above finh (all final); same expression as residual.hip
      const int bR = NDSM_BO(NST);
      const int prg = pr + g.k0;
      const bool inz = pr >= g.lb[2] && pr <= g.ub[2];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int f = RES ? s : 0;
        const d2 cc = f1[f], below = f2[f], fin = finh[s];
        // (element 0 is the point of class par, element 1 that of class 1 - par)
        const bool pr1 = __builtin_amdgcn_inverse_ballot_w64(mP[s]);
        const T xl0 = LDSD(bR + (pr1 ? cX[s][1] : cX[s][0]));
        const T xh1 = LDSD(bR + (pr1 ? cX[s][0] : cX[s][1]));
        d2 vl, vh;
        vl.x = LDSD(bR + (pr1 ? cYL[s][1] : cYL[s][0]));
        vl.y = LDSD(bR + (pr1 ? cYL[s][0] : cYL[s][1]));
        vh.x = LDSD(bR + (pr1 ? cYH[s][1] : cYH[s][0]));
        vh.y = LDSD(bR + (pr1 ? cYH[s][0] : cYH[s][1]));
        const d2 wl = (prg == 0) ? fin : below;
        const d2 wh = (prg == g.nzg - 1) ? below : fin;
        d2 rr;
        rr.x = rr.y = 0.0;
        if (!RHS0) rr = by_class(rw[RHS0 ? 0 : s][(RHS0 || !RES) ? 0 : NST], s);   // (the swap is its own inverse)
        const T v0 = (xl0 + cc.y) * gw0 + (vl.x + vh.x) * gw1 + (wl.x + wh.x) * gw2 - rr.x - cc.x * gwc;
        const T v1 = (cc.x + xh1) * gw0 + (vl.y + vh.y) * gw1 + (wl.y + wh.y) * gw2 - rr.y - cc.y * gwc;
        d2 res;
        res.x = (inz && __builtin_amdgcn_inverse_ballot_w64(mE[f][0])) ? -v0 : (T)0;
        res.y = (inz && __builtin_amdgcn_inverse_ballot_w64(mE[f][1])) ? -v1 : (T)0;
        resh[f] = res;
        f2[f] = cc;
        f1[f] = fin;
      }
    }
    if (!DEFER) {   // (planes outside the store window: a descriptor of zero records, nothing is written)
      const auto rs_ = rsrc_at(a_st, pf_st);
#pragma unroll
      for (int s = 0; s < NS; ++s) stp(rs_, scs[s].sto, scs[s].stg, finh[s]);
      if (RES) {
        const auto rr_ = rsrc_at(a_rs, pr_st);
#pragma unroll
        for (int s = 0; s < NS; ++s) stp(rr_, scs[s].sto, scs[s].stg, resh[RES ? s : 0]);
      }
    }

    __syncthreads();  // every stage is done with its plane

    // ---- plane k+1 takes the LDS buffer of the plane that has just left; shift the windows ----
    // (past the chunk's last plane the buffer load returned zeros: written, never used)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (tid0 + NT * s < NPAIR) {
        LDSD(scs[s].l0 + NDSM_BO(-1)) = nxt[s].x;
        LDSD(scs[s].l1 + NDSM_BO(-1)) = nxt[s].y;
      }
      nxt[s] = nn[s];
    }
    if (!RHS0) {
#pragma unroll
      for (int s = 0; s < (RHS0 ? 1 : NS); ++s) {
#pragma unroll
        for (int t = (RHS0 ? 1 : NSTG) - 1; t > 0; --t) rw[s][t] = rw[s][t - 1];
        rw[s][0] = by_class(rn[s], s);
      }
    }
    // ---- now the global stores: they have a whole iteration before the next vmcnt wait ----
    // (the empty asm consumes the freshly loaded window HERE, so the wait for those loads is
    // placed before the stores and cannot be sunk to the loop latch behind them)
    if (DEFER) {
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("" ::"v"(nxt[s].x), "v"(nxt[s].y) : "memory");
      if (MET && pf_st) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (__builtin_amdgcn_inverse_ballot_w64(mO[s])) {
            const d2 pv = pvh[MET ? s : 0];
            const double d0 = fabs((double)finh[s].x - (double)pv.x);
            const double d1 = (ODD && (scs[s].fl & 16)) ? 0.0 : fabs((double)finh[s].y - (double)pv.y);
            met_mx = fmax(met_mx, fmax(d0, d1));
            met_sm = met_sm + d0;
            met_sm = met_sm + d1;
          }
        }
      }
      {
        const auto rs_ = rsrc_at(a_st, pf_st);
#pragma unroll
        for (int s = 0; s < NS; ++s) stp(rs_, scs[s].sto, scs[s].stg, finh[s]);
      }
    }
    a_ld += pbs;
    a_st += pbs;
    if (!RHS0) a_rh += pbs;
    if (MET) a_pv += pbs;
    if (RES) a_rs += pbs;
    __syncthreads();
#undef NDSM_BO
  };
  if constexpr (GROUPS) {
    // (ks is a multiple of NCOPY.  The steps of the last group that lie beyond klast walk planes nobody stores and
    // request planes beyond the chunk's last, which a descriptor of zero records answers with zeros.)
#pragma unroll 1
    for (int k = ks; k <= klast; k += NCOPY) {
      plane_step(k, std::integral_constant<int, 0>());
      plane_step(k + 1, std::integral_constant<int, 1>());
      if constexpr (NCOPY > 2) {
        plane_step(k + 2, std::integral_constant<int, 2 % NCOPY>());
        plane_step(k + 3, std::integral_constant<int, 3 % NCOPY>());
      }
    }
  } else {
    int kc = ks % NCOPY;   // which copy serves step k
#pragma unroll 1
    for (int k = ks; k <= klast; ++k) {
      if (kc == 0) {
        plane_step(k, std::integral_constant<int, 0>());
      } else if (kc == 1) {
        plane_step(k, std::integral_constant<int, 1>());
      } else if (kc == 2) {
        plane_step(k, std::integral_constant<int, 2 % NCOPY>());
      } else if (kc == 3) {
        plane_step(k, std::integral_constant<int, 3 % NCOPY>());
      } else if (kc == 4) {
        plane_step(k, std::integral_constant<int, 4 % NCOPY>());
      } else {
        plane_step(k, std::integral_constant<int, 5 % NCOPY>());
      }
      kc = (kc + 1 == NCOPY) ? 0 : kc + 1;
    }
  }
  if (MET) {
    __shared__ double smx[NT / 64], ssm[NT / 64];
    for (int o = 32; o > 0; o >>= 1) {
      met_mx = fmax(met_mx, __shfl_down(met_mx, o, 64));
      met_sm = met_sm + __shfl_down(met_sm, o, 64);
    }
    if ((tid0 & 63) == 0) {
      smx[tid0 >> 6] = met_mx;
      ssm[tid0 >> 6] = met_sm;
    }
    __syncthreads();
    if (tid0 == 0) {
      double m = smx[0], a = ssm[0];
      for (int q = 1; q < NT / 64; ++q) {
        m = fmax(m, smx[q]);
        a = a + ssm[q];
      }
      part[2 * blockIdx.x] = m;
      part[2 * blockIdx.x + 1] = a;
    }
  }
#undef LDSD
#undef NDSM_LOAD_PLANE
}

// folds the per-workgroup (max, sum) partials of a MET launch in index order
// (acc: combine with what out2 holds - the launches of one z-slab pass, window by window, slab by slab)
__global__ __launch_bounds__(256) void fold_metric_k(const double *__restrict__ part, int nblocks,
                                                     double *__restrict__ out2, int acc) {
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
    const double m = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    const double a = ((ssm[0] + ssm[1]) + ssm[2]) + ssm[3];
    out2[0] = acc ? fmax(out2[0], m) : m;
    out2[1] = acc ? out2[1] + a : a;
  }
}

// scratch of the metric partials (2 doubles per workgroup + the folded pair at the end)
struct MetScratch {
  double *d = nullptr;
  size_t cap = 0;
};
MetScratch g_met;
bool g_met_acc = false;  // the next MET launch adds to the folded pair instead of replacing it
void met_release() {
  if (g_met.d) (void)hipFree(g_met.d);
  g_met = MetScratch();
  g_met_acc = false;
}
int met_scratch(size_t nblk, double **part, double **out2) {
  ndsm::at_reset(met_release);
  if (nblk > g_met.cap) {
    double *nd = nullptr;
    NDSM_HIP(hipMalloc((void **)&nd, sizeof(double) * (2 * nblk + 2)));
    if (g_met.d) {  // an accumulation in progress moves with the buffer
      NDSM_HIP(hipMemcpyAsync(nd + 2 * nblk, g_met.d + 2 * g_met.cap, 2 * sizeof(double), hipMemcpyDeviceToDevice,
                              ndsm::stream()));
      NDSM_HIP(hipStreamSynchronize(ndsm::stream()));
      (void)hipFree(g_met.d);
    }
    g_met.d = nd;
    g_met.cap = nblk;
  }
  *part = g_met.d;
  *out2 = g_met.d + 2 * g_met.cap;
  return 0;
}

template <typename T, int S, int TXH, int TYH, int NT, int WPS, int MODE = 0, bool LVL1 = false, bool ODD = false>
int launch_cfg(const ndsmk_grid &g, const T *u, T *uout, const T *rhs, int target_wgs, T *rout = nullptr,
               const T *prev = nullptr, const ProlArgs *prol = nullptr) {
  constexpr bool RES = MODE == 1;
  constexpr int NST = RES ? 2 * S + 1 : 2 * S;
  constexpr int TXI = TXH - 2 * ((NST + 1) & ~1), TYI = TYH - 2 * NST;
  static_assert(TXI > 0 && TYI > 0 && (TXH % 2) == 0, "tile");
  FusedPlan pl;
  // the last tile also owns the part of its halo that lies inside the domain (make_sc): n tiles
  // reach n * TXI + halo points
  constexpr int HXL = (NST + 1) & ~1;
  pl.ntx = g.n[0] > HXL ? (g.n[0] - HXL + TXI - 1) / TXI : 1;
  pl.nty = g.n[1] > NST ? (g.n[1] - NST + TYI - 1) / TYI : 1;
  const int tiles = pl.ntx * pl.nty;
  const int nzo = g.zown1 - g.zown0;  // owned planes
  const size_t lds_bytes = sizeof(T) * NST * TXH * TYH +
                           (MODE == 3 ? sizeof(double) * (2 * (TXH / 2 + 3) * (TYH / 2 + 3) + 2 * TXH + 2 * TYH) : 0);
  static int attr_epoch[2] = {0, 0};   // per instantiation and device epoch (ndsmk_init may re-target)
  static int wgs_per_cu[2] = {1, 1};
  // The TWO-sweep correction launch exists for the declared-zero right-hand side only: with a right-hand-side
  // window on top of the interpolation state it does not fit 128 registers (round 3, weights already in LDS: 124
  // bytes of scratch per lane).  The ONE-sweep one carries a right-hand side too (52 bytes of scratch per lane,
  // outside the plane loop's steady state; 512^3 Poisson cycle 6.83 -> 6.41 ms against the stand-alone
  // interpolation in front of the sweeps).
  constexpr bool GEN = MODE != 3 || S == 1;   // is there an instantiation that reads rhs?
  if (!GEN && rhs) return ndsm::fail(NDSMK_EARG, "the correction launch is built for rhs == 0 only", __FILE__, __LINE__);
  using KF = void (*)(const T *, T *, const T *, T *, const T *, double *, ndsmk_grid, FusedPlan, ProlArgs);
  KF kgen = nullptr;
  if constexpr (GEN) kgen = rbgs3_fused_k<T, S, TXH, TYH, NT, WPS, false, MODE, LVL1, ODD>;
  const KF kfn = rhs ? kgen : rbgs3_fused_k<T, S, TXH, TYH, NT, WPS, true, MODE, LVL1, ODD>;
  const int v = rhs ? 0 : 1;
  const void *kptr = reinterpret_cast<const void *>(kfn);
  if (ndsm::first_in_epoch(attr_epoch[v])) {
    NDSM_HIP(hipFuncSetAttribute(kptr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int occ = 1;
    NDSM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, NT, lds_bytes));
    wgs_per_cu[v] = occ > 0 ? occ : 1;
  }
  // z chunks.  A workgroup walks its chunk plus NST warm-up planes on either side, and
  // the chip runs ncu * wgs_per_cu workgroups at a time: pick the chunk count that
  // minimises (rounds of workgroups) x (planes walked per workgroup).
  int nzc;
  if (target_wgs > 0) {
    nzc = (target_wgs + tiles - 1) / tiles;
    if (nzc < 1) nzc = 1;
    int zc = (nzo + nzc - 1) / nzc;
    if (zc < 16) zc = 16 < nzo ? 16 : nzo;
    nzc = (nzo + zc - 1) / zc;
  } else {
    const int64_t slots = (int64_t)ndsm::cu_count() * wgs_per_cu[v];
    int64_t best = -1;
    nzc = 1;
    for (int c = 1; c <= (nzo + 7) / 8; ++c) {
      const int zc = (nzo + c - 1) / c;
      const int cc = (nzo + zc - 1) / zc;
      const int64_t rounds = ((int64_t)tiles * cc + slots - 1) / slots;
      const int64_t cost = rounds * (zc + 2 * NST + (RES ? 0 : NST - 1));
      if (best < 0 || cost < best) {
        best = cost;
        nzc = cc;
      }
    }
  }
  pl.zc = (nzo + nzc - 1) / nzc;
  pl.nzc = (nzo + pl.zc - 1) / pl.zc;
  pl.nwork = tiles * pl.nzc;
  const int nblk = ((pl.nwork + 7) / 8) * 8;
  ProlArgs pa = {};
  if (prol) pa = *prol;
  double *part = nullptr, *out2 = nullptr;
  if (MODE == 2) {
    if (int rc = met_scratch((size_t)nblk, &part, &out2)) return rc;
  }
  // (rhs == nullptr: the level's rhs is identically zero - level 1 of NDSM's Laplace problems,
  // ndsm_vector_potential.f90:640-641 -; x - 0.0 == x exactly, so the variant that never loads rhs returns the
  // same bits with 8 B/LUP less traffic)
  hipLaunchKernelGGL(kfn, dim3(nblk), dim3(NT), lds_bytes, ndsm::stream(), u, uout, rhs, rout, prev, part, g, pl, pa);
  NDSM_LAUNCH_CHECK();
  if (MODE == 2) {
    hipLaunchKernelGGL(fold_metric_k, dim3(1), dim3(256), 0, ndsm::stream(), part, nblk, out2, g_met_acc ? 1 : 0);
    NDSM_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

namespace ndsm {

// Development knob: NDSM_FUSED_CFG=<two-sweep cfg>,<one-sweep cfg>,<sweep+residual cfg>,<work items>,<big>
// picks tile configurations (default 0,0,0,0,-1; scripts/tune_smoother.py).  <big>: -1 = by level size
// (>= 64 M points), 0 / 1 = never / always take the large-level tiles - the parity tests run them on
// oracle-sized grids that way (ndsmk_debug_fused_cfg sets the same five values at run time).
static int g_fcfg[5] = {-1, 0, 0, 0, -1};
static const int *fused_cfg() {
  int *cfg = g_fcfg;
  if (cfg[0] < 0) {
    cfg[0] = 0;
    const char *e = std::getenv("NDSM_FUSED_CFG");
    for (int i = 0; e && i < 5; ++i) {
      cfg[i] = std::atoi(e);
      e = std::strchr(e, ',');
      if (e) ++e;
    }
  }
  return cfg;
}

// rout != nullptr: the caller wants the residual of the swept field as well.  It is
// produced (and *res_done set) only by the launch that performs the LAST of the
// max_sweeps sweeps, i.e. when this call runs a single sweep with max_sweeps == 1.
template <typename T, bool ODD = false>
static int launch_fused_t(const ndsmk_grid &g, const T *u, T *uout, const T *rhs, int max_sweeps, bool force,
                          int *sweeps_done, T *rout, int *res_done, const T *prev = nullptr, int *met_done = nullptr,
                          const ProlArgs *prol = nullptr) {
  *sweeps_done = 0;
  if (res_done) *res_done = 0;
  if (met_done) *met_done = 0;
  if (!uout || g.ndim != 3 || ((g.n[0] & 1) != 0) != ODD || g.n[0] < 16 || g.n[1] < 16 || g.zown1 - g.zown0 < (force ? 1 : 8))
    return 0;
  // the buffer path addresses a plane with 32-bit byte offsets below kDeadLane (2 GiB: 16384 x 16384 doubles)
  if ((int64_t)g.n[0] * g.n[1] * (int64_t)sizeof(T) >= (int64_t)kDeadLane) return 0;
  // a z-streaming workgroup walks >= 16 planes serially: with fewer than ~one
  // workgroup per CU the sweep is latency bound and the two colour passes win
  const int64_t npts = (int64_t)g.n[0] * g.n[1] * (g.zown1 - g.zown0);
  const bool slab = g.zown1 - g.zown0 != g.n[2];
  if (npts < (int64_t)2 * 1024 * 1024 && !slab && !force) return 0;
  // Tile choices measured on MI355X (scripts/tune_smoother.py); launch_cfg picks the z chunking.
  int rc;
  const int *cfg = fused_cfg();
  const int tgt = cfg[3] > 0 ? cfg[3] : 0;  // > 0: that many work items instead of launch_cfg's own choice
  const bool big = cfg[4] < 0 ? npts >= (int64_t)64 * 1024 * 1024 : cfg[4] != 0;
  // A z-slab carries zown0 ghost planes below and n[2] - zown1 above its owned range; the caller
  // (ndsmh_world) has exchanged as many as the pass it asks for consumes: 2 per sweep, +1 for
  // the residual stage.
  const int ghosts = slab ? (g.zown0 < g.n[2] - g.zown1 ? g.zown0 : g.n[2] - g.zown1) : 1 << 20;
  const bool res = rout && res_done && ghosts >= 3 && cfg[2] != 9;
  // two sweeps per pass - not for the last two sweeps when the residual is wanted (it rides
  // on a one-sweep pass)
  const bool two = max_sweeps >= 2 && ghosts >= 4 && cfg[0] != 9 && !(res && max_sweeps == 2);
  // prev != nullptr: the launch that performs the last of the max_sweeps sweeps also evaluates
  // the convergence metric against prev (fp64, single domain; *met_done says it did)
  const bool met = std::is_same<T, double>::value && prev && met_done;
  // prol != nullptr: u + P u_c is to be formed while the planes are loaded (MODE 3): a one-sweep pass (any
  // right-hand side) or a two-sweep pass (rhs == 0 only).  If this call cannot be such a launch NOTHING is launched
  // (*sweeps_done = 0) and the caller interpolates with the stand-alone kernel first.
  // The interpolation costs ~100 instructions per plane-step and wave.  On top of a TWO-sweep pass - which is bound
  // by instruction issue with it - that is +240 us at 512^3 (720 against 480); a ONE-sweep pass has the issue
  // slots to spare (it is bound by memory: 430 us).  So an odd number of sweeps (NDSM's ms = 5) is taken as
  // 1 + 2 + 2 with the correction on the one-sweep pass - the sweeps that remain are two-sweep passes, the last
  // of which carries the convergence metric - instead of 2 + 2 + 1.
  if (prol) {
    if constexpr (std::is_same<T, double>::value) {
      if (cfg[0] == 0 && (max_sweeps & 1) && (max_sweeps >= 3 || slab)) {   // (a slab window asks for exactly its pass)
        rc = (launch_cfg<T, 1, 132, 31, 1024, 4, 3, false, ODD>(g, u, uout, rhs, tgt, nullptr, nullptr, prol));
        if (rc) return rc;
        *sweeps_done = 1;
      } else if (!rhs && two && !(met && max_sweeps == 2) && cfg[0] == 0) {
        rc = (launch_cfg<T, 2, 136, 30, 1024, 4, 3, false, ODD>(g, u, uout, rhs, tgt, nullptr, nullptr, prol));
        if (rc) return rc;
        *sweeps_done = 2;
      }
    }
    return 0;
  }
  if constexpr (std::is_same<T, double>::value) {
    if (met && two && max_sweeps == 2) {
      rc = (launch_cfg<T, 2, 136, 30, 1024, 4, 2, false, ODD>(g, u, uout, rhs, tgt, nullptr, prev));
      if (rc) return rc;
      *sweeps_done = 2;
      *met_done = 1;
      return 0;
    }
    if (met && !two && max_sweeps == 1 && !res) {
      if (big)
        rc = (launch_cfg<T, 1, 132, 31, 1024, 4, 2, false, ODD>(g, u, uout, rhs, tgt, nullptr, prev));
      else
        rc = (launch_cfg<T, 1, 132, 23, 768, 4, 2, false, ODD>(g, u, uout, rhs, tgt, nullptr, prev));
      if (rc) return rc;
      *sweeps_done = 1;
      *met_done = 1;
      return 0;
    }
  }
  // THREE sweeps per pass were tried and are not built: six LDS planes force a narrower tile (72 x 46 or
  // 88 x 38 owned 60 x 34 / 76 x 26: 1.6-1.7x redundant stage work instead of 1.45x) and the pass turns
  // LDS / issue bound - measured at 512^3 (Laplace): 1075-1120 us per three-sweep pass against 553 us per
  // two-sweep pass, i.e. 358 against 282 us per sweep; same bits (scripts/time_s3.py, round 2).
  if (two) {
    if constexpr (ODD) {   // the tuning alternates and the level-1 symbol exist for even nx only
      rc = (launch_cfg<T, 2, 136, 30, 1024, 4, 0, false, true>(g, u, uout, rhs, tgt));
    } else {
      // levels of a few million points (the 128^3 level of a 512^3 hierarchy) do not fill the chip with
      // 136 x 30 tiles: the 72 x 28 tile (512 threads) gives four times the workgroups - measured 77 against
      // 98 us per five sweeps at 128^3, 293 against 280 us at 256^3 (scripts/time_tail.py)
      const int two_cfg = cfg[0] ? cfg[0] : (npts < (int64_t)6 * 1024 * 1024 && !slab ? 5 : 0);
      switch (two_cfg) {
        case 3: rc = launch_cfg<T, 2, 136, 22, 768, 4>(g, u, uout, rhs, tgt); break;
        case 5: rc = launch_cfg<T, 2, 72, 28, 512, 4>(g, u, uout, rhs, tgt); break;
        default:
          // (fp32: four LDS planes are 65 KB, so two workgroups would fit a CU at <= 64 VGPRs - tried:
          // __launch_bounds__(1024, 8) gives 64 VGPRs + 44 B of scratch and the mixed-precision cycle at 512^3
          // goes from 6.5 to 7.9 ms; not built)
          rc = big ? (launch_cfg<T, 2, 136, 30, 1024, 4, 0, true>(g, u, uout, rhs, tgt))
                   : (launch_cfg<T, 2, 136, 30, 1024, 4>(g, u, uout, rhs, tgt));
          break;
      }
    }
    if (rc) return rc;
    *sweeps_done = 2;
    return 0;
  }
  if (res && max_sweeps == 1) {
    if constexpr (ODD) {
      rc = (launch_cfg<T, 1, 136, 22, 768, 4, 1, false, true>(g, u, uout, rhs, tgt, rout));
    } else {
      rc = (launch_cfg<T, 1, 136, 22, 768, 4, 1>(g, u, uout, rhs, tgt, rout));
    }
    if (rc) return rc;
    *sweeps_done = 1;
    *res_done = 1;
    return 0;
  }
  switch (cfg[1] ? cfg[1] : (big ? 7 : 8)) {
    case 8: rc = (launch_cfg<T, 1, 132, 23, 768, 4, 0, false, ODD>(g, u, uout, rhs, tgt)); break;
    default: rc = (launch_cfg<T, 1, 132, 31, 1024, 4, 0, false, ODD>(g, u, uout, rhs, tgt)); break;
  }
  if (rc) return rc;
  *sweeps_done = 1;
  return 0;
}

int launch_rbgs3_fused(const ndsmk_grid &g, const double *u, double *uout, const double *rhs, int max_sweeps,
                       bool force, int *sweeps_done, double *rout, int *res_done, const double *prev,
                       int *met_done, const ndsmk_xfer *px, const double *uc) {
  if (px) {
    ProlArgs pa;
    pa.uc = uc;
    for (int d = 0; d < 3; ++d) {
      pa.plo[d] = px->plo[d];
      pa.pwl[d] = px->pwl[d];
      pa.pwh[d] = px->pwh[d];
    }
    pa.ncx = px->nc[0];
    pa.ncy = px->nc[1];
    pa.ncz = px->nc[2];
    // z windows: the fine array is the slab described by g, uc holds coarse planes [c_k0, c_k0 + c_cnt)
    // (single domain: c_k0 = 0 and the whole coarse level); the caller guarantees that they cover the
    // brackets of every fine plane the launch loads (owned planes + ghosts)
    pa.fk0 = px->f_k0;
    pa.nzf = px->nf[2];
    pa.ck0 = px->c_k0;
    pa.nczw = (px->f_k0 == 0 && px->c_k0 == 0 && px->nf[2] == g.n[2]) ? px->nc[2] : px->c_cnt;
    if (px->f_k0 != g.k0 || px->nf[0] != g.n[0] || px->nf[1] != g.n[1] || px->nf[2] != g.nzg || pa.nczw < 1 ||
        (int64_t)pa.ncx * pa.ncy * 8 >= (int64_t)kDeadLane) {   // (a coarse plane is addressed with 32-bit byte offsets)
      *sweeps_done = 0;
      return 0;
    }
    if (g.n[0] & 1)
      return launch_fused_t<double, true>(g, u, uout, rhs, max_sweeps, force, sweeps_done, rout, res_done, prev, met_done, &pa);
    return launch_fused_t<double>(g, u, uout, rhs, max_sweeps, force, sweeps_done, rout, res_done, prev, met_done, &pa);
  }
  if (g.n[0] & 1)
    return launch_fused_t<double, true>(g, u, uout, rhs, max_sweeps, force, sweeps_done, rout, res_done, prev, met_done);
  return launch_fused_t<double>(g, u, uout, rhs, max_sweeps, force, sweeps_done, rout, res_done, prev, met_done);
}

// (max, sum) of |u_new - u_prev| left on the device by the last MET launch -> host (blocking)
int fetch_fused_metric(double *h_out2) {
  if (!g_met.d) return fail(NDSMK_EARG, "no fused metric has been computed", __FILE__, __LINE__);
  NDSM_HIP(hipMemcpyAsync(h_out2, g_met.d + 2 * g_met.cap, 2 * sizeof(double), hipMemcpyDeviceToHost, stream()));
  NDSM_HIP(hipStreamSynchronize(stream()));
  return 0;
}

// the MET launches that follow add their (max, sum) to the pair on the device instead of replacing it
void fused_metric_accumulate(bool on) { g_met_acc = on; }

// fp32 instantiation: the correction equation L e = r of the mixed-precision mode (same tiles:
// half the LDS and HBM bytes per point, the same instruction count)
int launch_rbgs3_fused_f32(const ndsmk_grid &g, const float *u, float *uout, const float *rhs, int max_sweeps,
                           bool force, int *sweeps_done, float *rout, int *res_done) {
  return launch_fused_t<float>(g, u, uout, rhs, max_sweeps, force, sweeps_done, rout, res_done);
}

}  // namespace ndsm

// tests / tuning: the five values of NDSM_FUSED_CFG, set at run time
extern "C" int ndsmk_debug_fused_cfg(int two, int one, int res, int work_items, int big) {
  ndsm::g_fcfg[0] = two < 0 ? 0 : two;
  ndsm::g_fcfg[1] = one;
  ndsm::g_fcfg[2] = res;
  ndsm::g_fcfg[3] = work_items;
  ndsm::g_fcfg[4] = big;
  return 0;
}
