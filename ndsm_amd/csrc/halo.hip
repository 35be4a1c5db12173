// Ghost-plane exchange of a smoothing pass: only what the sweeps read.
//
// A red-black sweep updates the RED points of a plane first ((i + j + k + first_par) even, k the GLOBAL plane index -
// rbgs3_fused_k's stage 0, red_black_gauss_3D ndsm_optimized.f90:103-167) from black neighbours only, and the z-slab
// passes recompute the red points of every ghost plane they use before anything reads them (ndsmh_world.f90:
// world_relax: red needs the neighbour plane's black, black needs that red).  So of a ghost plane a pass reads
//   * the BLACK points as the neighbour last left them, and
//   * the points no sweep ever updates - Dirichlet data on the x / y faces, i.e. a subset of the plane's perimeter -
// and nothing else: the interior red points of a ghost plane are dead weight on the link.  These kernels move
// black points + perimeter: (nx ny)/2 + 2 (nx + ny) doubles per plane instead of nx ny.
//
// Packed layout of one plane: ny row slots of W = (nx + 1) / 2 + 2 doubles - the row's black points in ascending
// i, then (at W - 2, W - 1) the row's first and last point - followed by row 0 and row ny - 1 in full.
#include "common.hpp"

namespace {

struct HaloArgs {
  int nx, ny, depth, kg0, fp;
  long long plane, packed;   // doubles per plane / per packed plane
};

__device__ __forceinline__ int row_slot(int nx) { return (nx + 1) / 2 + 2; }

// element e of a packed plane -> (i, j) of the plane, or i < 0: padding
__device__ __forceinline__ void packed_to_ij(const HaloArgs &a, int kg, long long e, int &i, int &j) {
  const int W = row_slot(a.nx);
  const long long nrows = (long long)a.ny * W;
  if (e < nrows) {
    j = (int)(e / W);
    const int t = (int)(e - (long long)j * W);
    if (t == W - 2) {
      i = 0;
    } else if (t == W - 1) {
      i = a.nx - 1;
    } else {
      const int i0 = (j + kg + a.fp + 1) & 1;   // first black point of the row
      i = i0 + 2 * t;
      if (i >= a.nx) i = -1;
    }
  } else {
    const long long r = e - nrows;
    j = r < a.nx ? 0 : a.ny - 1;
    i = (int)(r < a.nx ? r : r - a.nx);
  }
}

// MODE 0: plane -> packed; 1: packed -> plane; 2: plane -> plane (loop-back worlds: the same points, straight across)
template <int MODE>
__global__ __launch_bounds__(256) void halo_k(const double *__restrict__ src, double *__restrict__ dst, HaloArgs a) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;   // plane of the message
  if (e >= a.packed) return;
  int i, j;
  packed_to_ij(a, a.kg0 + p, e, i, j);
  if (i < 0) {
    if (MODE == 0) dst[(long long)p * a.packed + e] = 0.0;   // (padding travels as zeros: the message has no uninitialised word)
    return;
  }
  const long long c = (long long)p * a.plane + (long long)i + (long long)a.nx * j;
  if (MODE == 0)
    dst[(long long)p * a.packed + e] = src[c];
  else if (MODE == 1)
    dst[c] = src[(long long)p * a.packed + e];
  else
    dst[c] = src[c];
}

HaloArgs make_args(int nx, int ny, int depth, int kg0, int fp) {
  HaloArgs a;
  a.nx = nx;
  a.ny = ny;
  a.depth = depth;
  a.kg0 = kg0;
  a.fp = fp & 1;
  a.plane = (long long)nx * ny;
  a.packed = (long long)ny * ((nx + 1) / 2 + 2) + 2LL * nx;
  return a;
}

template <int MODE>
int launch(const double *src, double *dst, int nx, int ny, int depth, int kg0, int fp) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(src && dst && nx >= 2 && ny >= 2 && depth >= 1);
  const HaloArgs a = make_args(nx, ny, depth, kg0, fp);
  const dim3 grid((unsigned)((a.packed + 255) / 256), (unsigned)depth);
  hipLaunchKernelGGL(halo_k<MODE>, grid, dim3(256), 0, ndsm::stream(), src, dst, a);
  NDSM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

// doubles of one packed plane
long long ndsmk_halo_packed_plane(int nx, int ny) { return make_args(nx, ny, 1, 0, 0).packed; }

// src: `depth` consecutive planes (nx ny doubles each) whose first has GLOBAL index kg0; buf: depth packed planes
int ndsmk_halo_pack(const double *src, double *buf, int nx, int ny, int depth, int kg0, int first_par) {
  return launch<0>(src, buf, nx, ny, depth, kg0, first_par);
}
int ndsmk_halo_unpack(double *dst, const double *buf, int nx, int ny, int depth, int kg0, int first_par) {
  return launch<1>(buf, dst, nx, ny, depth, kg0, first_par);
}
// loop-back: the same points from the planes at src into the planes at dst (both nx ny doubles per plane)
int ndsmk_halo_copy(double *dst, const double *src, int nx, int ny, int depth, int kg0, int first_par) {
  return launch<2>(src, dst, nx, ny, depth, kg0, first_par);
}

}  // extern "C"
