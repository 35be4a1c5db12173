// The face phase of the vector-potential driver on the device, so that nothing but six
// fluxes travels between the upload of B.n and the final download:
//
//   extract : B.n on the six faces of a device-resident B        ndsm_vector_potential.f90:283-299, :699-743
//   flux    : trapezoid integral of B.n over each face           :301-306, :1070-1106 (quirk Q4: dq(1)*dq(2))
//   rhs     : bn - phi / area, the right-hand side of the 2-D all-Neumann problem laplace(chi) = ...  :340-352
//   write   : A_t = -grad(chi) x n, central differences of chi with the NORMAL spacing (Q4), zero on the
//             face's own edges, written straight into the boundary plane of a 3-D solver's level-1 array
//             :387-399, :977-1031 (compute_At_bcs) and :647-650, :663-666, :679-682 (extract_bn dir = -1)
//
// Faces are numbered as in ndsmh_vecpot.f90 (0-based here): 0/1 = lower/upper x, 2/3 = y, 4/5 = z; a face
// array is (n1, n2) in Fortran order with (t1, t2) = (y,z), (x,z), (x,y).  The six faces are packed back to
// back in one buffer (ndsmk_face_offsets).  All O(N^(2/3)): every kernel is latency bound and tiny - what
// matters is that the data never leaves HBM, not their speed.
//
// Arithmetic: the reference's expressions and operand order; the one order-dependent step is the flux
// sum, which the reference accumulates serially (:1086-1103) and this file as a fixed-shape tree (the
// same bits on every run; |delta phi| ~ 1e-16 relative, inside the pipeline's stated 1e-12 bound, which
// the unordered OpenMP mean of the reference's own 2-D solves already needs).
#include "common.hpp"

namespace {

struct FaceGeo {
  int n1[6], n2[6];
  size_t off[6];   // element offset of face f in the packed buffer
  size_t total;
};

__host__ __device__ inline int face_axis(int f) { return f >> 1; }

FaceGeo geo_of(const int32_t *n3) {
  FaceGeo g;
  size_t o = 0;
  for (int f = 0; f < 6; ++f) {
    const int ax = face_axis(f);
    g.n1[f] = ax == 0 ? n3[1] : n3[0];
    g.n2[f] = ax == 2 ? n3[1] : n3[2];
    g.off[f] = o;
    o += (size_t)g.n1[f] * (size_t)g.n2[f];
  }
  g.total = o;
  return g;
}

// linear index into a (nx,ny,nz) array of point (a, b) of face f (layer = 0 or n-1 along the normal)
__device__ __forceinline__ size_t face_point(int f, int a, int b, int nx, int ny, int nz) {
  const int ax = f >> 1;
  const bool up = f & 1;
  int i, j, k;
  if (ax == 0) {
    i = up ? nx - 1 : 0; j = a; k = b;
  } else if (ax == 1) {
    i = a; j = up ? ny - 1 : 0; k = b;
  } else {
    i = a; j = b; k = up ? nz - 1 : 0;
  }
  return (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * (size_t)k);
}

// faces[off_f + a + n1 b] = B(face point; component = the face's normal axis)
__global__ __launch_bounds__(256) void face_extract_k(const double *__restrict__ B, double *__restrict__ faces, FaceGeo g,
                                                      int nx, int ny, int nz) {
  const int f = blockIdx.z;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (a >= g.n1[f] || b >= g.n2[f]) return;
  const size_t N = (size_t)nx * ny * nz;
  faces[g.off[f] + (size_t)a + (size_t)g.n1[f] * b] = B[(size_t)(f >> 1) * N + face_point(f, a, b, nx, ny, nz)];
}

// phi[f] = (sum_ab w(a,b) bn(a,b)) * h1h2, w = 1 / 0.5 on edges / 0.25 on corners.  One workgroup per face:
// thread t sums elements t, t + 1024, ... in index order, then a fixed tree over lanes and waves.
__global__ __launch_bounds__(1024) void face_flux_k(const double *__restrict__ faces, FaceGeo g, double h1h2,
                                                    double *__restrict__ phi) {
  __shared__ double part[16];
  const int f = blockIdx.x;
  const int n1 = g.n1[f], n2 = g.n2[f];
  const double *v = faces + g.off[f];
  const int n = n1 * n2;
  double s = 0.0;
  for (int p = threadIdx.x; p < n; p += 1024) {
    const int a = p % n1, b = p / n1;
    const bool ea = a == 0 || a == n1 - 1, eb = b == 0 || b == n2 - 1;
    double w = 1.0;
    if (ea || eb) w = 0.5;
    if (ea && eb) w = 0.25;
    s = s + w * v[p];
  }
  for (int o = 32; o > 0; o >>= 1) s = s + __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t = t + part[q];
    phi[f] = t * h1h2;
  }
}

// rhs2d = bn - phi[f] / area   (the quotient formed once, as the reference's scalar expression)
__global__ __launch_bounds__(256) void face_rhs_k(const double *__restrict__ bn, double *__restrict__ rhs, int n,
                                                  const double *__restrict__ phi, int f, double area) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double q = phi[f] / area;
  rhs[p] = bn[p] - q;
}

// u(face f) = A_t component c of chi_f.  which = 1: at1 = s1 * d(chi)/d(t2), which = 2: at2 = s2 * d(chi)/d(t1)
// with d/dt = fac * (chi(+1) - chi(-1)), zero on the edges of that direction (:1007-1017)
__global__ __launch_bounds__(256) void face_write_k(double *__restrict__ u, const double *__restrict__ chi, int f, int n1,
                                                    int n2, int which, double sgn, double fac, int nx, int ny, int nz) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (a >= n1 || b >= n2) return;
  double d = 0.0;
  if (which == 1) {
    if (b > 0 && b < n2 - 1) d = fac * (chi[a + (size_t)n1 * (b + 1)] - chi[a + (size_t)n1 * (b - 1)]);
  } else {
    if (a > 0 && a < n1 - 1) d = fac * (chi[(a + 1) + (size_t)n1 * b] - chi[(a - 1) + (size_t)n1 * b]);
  }
  u[face_point(f, a, b, nx, ny, nz)] = sgn * d;
}

// u(face f) = vals (a packed face that was computed elsewhere)
__global__ __launch_bounds__(256) void face_put_k(double *__restrict__ u, const double *__restrict__ vals, int f, int n1,
                                                  int n2, int nx, int ny, int nz) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (a >= n1 || b >= n2) return;
  u[face_point(f, a, b, nx, ny, nz)] = vals[a + (size_t)n1 * b];
}

}  // namespace

extern "C" {

// element offsets of the six faces in the packed buffer (off6) and its total length
int ndsmk_face_offsets(const int32_t *n3, int64_t *off6, int64_t *total) {
  const FaceGeo g = geo_of(n3);
  for (int f = 0; f < 6; ++f) off6[f] = (int64_t)g.off[f];
  *total = (int64_t)g.total;
  return 0;
}

// B: DEVICE (nx,ny,nz,3) -> faces: DEVICE packed B.n
int ndsmk_face_extract(const double *B, const int32_t *n3, double *faces) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(B && faces && n3[0] >= 2 && n3[1] >= 2 && n3[2] >= 2);
  const FaceGeo g = geo_of(n3);
  int m1 = 0, m2 = 0;
  for (int f = 0; f < 6; ++f) {
    m1 = g.n1[f] > m1 ? g.n1[f] : m1;
    m2 = g.n2[f] > m2 ? g.n2[f] : m2;
  }
  hipLaunchKernelGGL(face_extract_k, dim3((m1 + 255) / 256, m2, 6), dim3(256), 0, ndsm::stream(), B, faces, g, n3[0], n3[1],
                     n3[2]);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// d_phi6[f] = trapezoid(B.n on face f) * h1h2   (Q4: the same h1h2 = dq(1) dq(2) for every face)
int ndsmk_face_flux(const double *faces, const int32_t *n3, double h1h2, double *d_phi6) {
  NDSM_REQUIRE_READY();
  const FaceGeo g = geo_of(n3);
  hipLaunchKernelGGL(face_flux_k, dim3(6), dim3(1024), 0, ndsm::stream(), faces, g, h1h2, d_phi6);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// rhs (DEVICE, n = n1 n2 values: level 1 of a 2-D solver) = faces_f - d_phi6[f] / area
int ndsmk_face_rhs(const double *faces, const int32_t *n3, int f, const double *d_phi6, double area, double *rhs) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(f >= 0 && f < 6);
  const FaceGeo g = geo_of(n3);
  const int n = g.n1[f] * g.n2[f];
  hipLaunchKernelGGL(face_rhs_k, dim3((n + 255) / 256), dim3(256), 0, ndsm::stream(), faces + g.off[f], rhs, n, d_phi6, f,
                     area);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// boundary plane of face f of the 3-D array u (nx,ny,nz) <- tangential component c (0,1,2 = x,y,z) of
// A_t = -grad(chi_f) x n; chi: DEVICE packed (the six solved faces); fac = 1 / (2 dq(normal axis)) (Q4)
int ndsmk_face_write(double *u, const int32_t *n3, const double *chi, int f, int c, double fac) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(u && chi && f >= 0 && f < 6 && c >= 0 && c < 3 && c != (f >> 1));
  const FaceGeo g = geo_of(n3);
  // tangential axes of the face: (t1, t2) = (y,z), (x,z), (x,y); A_t = (s1 dchi/dt2, s2 dchi/dt1) on (t1, t2)
  const int ax = f >> 1;
  const int t1 = ax == 0 ? 1 : 0;
  static const double s1[3] = {-1.0, +1.0, -1.0}, s2[3] = {+1.0, -1.0, +1.0};
  const int which = (c == t1) ? 1 : 2;
  const double sgn = which == 1 ? s1[ax] : s2[ax];
  hipLaunchKernelGGL(face_write_k, dim3((g.n1[f] + 255) / 256, g.n2[f]), dim3(256), 0, ndsm::stream(), u, chi + g.off[f], f,
                     g.n1[f], g.n2[f], which, sgn, fac, n3[0], n3[1], n3[2]);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// boundary plane of face f of u <- vals (DEVICE, n1 x n2 of that face)
int ndsmk_face_put(double *u, const int32_t *n3, int f, const double *vals) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(u && vals && f >= 0 && f < 6);
  const FaceGeo g = geo_of(n3);
  hipLaunchKernelGGL(face_put_k, dim3((g.n1[f] + 255) / 256, g.n2[f]), dim3(256), 0, ndsm::stream(), u, vals, f, g.n1[f],
                     g.n2[f], n3[0], n3[1], n3[2]);
  NDSM_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
