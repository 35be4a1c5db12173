// Post-processing of the vector-potential driver on the device, so that A and
// B never leave HBM between the last V-cycle and the single final download:
//
//   balance : analytic fields that carry the net flux through opposite faces,
//             added to A and B              ndsm_vector_potential.f90:880-950
//   curl    : B = curl A, second-order centred differences, 3-point one-sided
//             on the end planes             ndsm_vector_potential.f90:759-872
//
// Pure streaming: balance touches 6 N-vectors (48 B/pt in, 48 B/pt out), curl
// reads 3 and writes 3 (48 B/pt with neighbours served by L2).  Operand order
// follows the reference; -ffp-contract=off keeps the roundings identical.
#include "common.hpp"

namespace {

// A and B may be z-slabs of the global field (ndsmh_wvecpot): A holds planes [kg0, kg0 + na) of the
// global nz, B the nb planes that start at A's local plane boff (A carries a ghost plane per
// neighbour for the z differences).  Single domain: kg0 = boff = 0, na = nb = nz.
struct PostArgs {
  int n[3];                 // GLOBAL shape
  int kg0, na, nb, boff;
  const double *x, *y, *z;  // device mesh vectors (global)
  double phi[6];
  double span[3];
  double dq[3];
};

template <bool WITH_B>
__global__ __launch_bounds__(256) void balance_k(double *__restrict__ A, double *__restrict__ B, PostArgs p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;  // local plane of A
  if (i >= p.n[0] || j >= p.n[1]) return;
  const size_t N = (size_t)p.n[0] * p.n[1] * p.na, NB = (size_t)p.n[0] * p.n[1] * p.nb;
  const size_t c = (size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k);
  const double vol = p.span[0] * p.span[1] * p.span[2];
  const double g1 = (p.phi[1] - p.phi[0]) / vol, g2 = (p.phi[3] - p.phi[2]) / vol, g3 = (p.phi[5] - p.phi[4]) / vol;
  const double x = p.x[i], y = p.y[j], z = p.z[p.kg0 + k];
  const double third = 1.0 / 3.0;
  // :927-929
  const double b1 = g1 * x + p.phi[0] * p.span[0] / vol;
  const double b2 = g2 * y + p.phi[2] * p.span[1] / vol;
  const double b3 = g3 * z + p.phi[4] * p.span[2] / vol;
  // :932-939  A1_l, A2_l, A3_l (linear-B part) and A_c (constant-B part)
  const double l1x = -g3 * y * z, l1y = 0.0, l1z = +g1 * x * y;
  const double l2x = +g2 * z * y, l2y = -g1 * x * z, l2z = 0.0;
  const double l3x = 0.0, l3y = +g3 * x * z, l3z = -g2 * x * y;
  const double cx = -(p.phi[4] * p.span[2] * y / vol);
  const double cy = -(p.phi[0] * p.span[0] * z / vol);
  const double cz = -(p.phi[2] * p.span[1] * x / vol);
  // :942-943
  if (WITH_B && k >= p.boff && k < p.boff + p.nb) {
    const size_t cb = c - (size_t)p.n[0] * p.n[1] * (size_t)p.boff;
    B[cb] = B[cb] + b1;
    B[cb + NB] = B[cb + NB] + b2;
    B[cb + 2 * NB] = B[cb + 2 * NB] + b3;
  }
  A[c] = A[c] + cx + third * (l1x + l2x + l3x);
  A[c + N] = A[c + N] + cy + third * (l1y + l2y + l3y);
  A[c + 2 * N] = A[c + 2 * N] + cz + third * (l1z + l2z + l3z);
}

// Component C of the same update of A alone (whole field, single domain): the driver finishes and
// downloads one component while the next one is being solved.  Expressions and order as above.
template <int C>
__global__ __launch_bounds__(256) void balance_comp_k(double *__restrict__ Ac, PostArgs p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= p.n[0] || j >= p.n[1]) return;
  const size_t c = (size_t)i + (size_t)p.n[0] * ((size_t)j + (size_t)p.n[1] * (size_t)k);
  const double vol = p.span[0] * p.span[1] * p.span[2];
  const double g1 = (p.phi[1] - p.phi[0]) / vol, g2 = (p.phi[3] - p.phi[2]) / vol, g3 = (p.phi[5] - p.phi[4]) / vol;
  const double x = p.x[i], y = p.y[j], z = p.z[k];
  const double third = 1.0 / 3.0;
  if (C == 0) {
    const double l1x = -g3 * y * z, l2x = +g2 * z * y, l3x = 0.0;
    const double cx = -(p.phi[4] * p.span[2] * y / vol);
    Ac[c] = Ac[c] + cx + third * (l1x + l2x + l3x);
  } else if (C == 1) {
    const double l1y = 0.0, l2y = -g1 * x * z, l3y = +g3 * x * z;
    const double cy = -(p.phi[0] * p.span[0] * z / vol);
    Ac[c] = Ac[c] + cy + third * (l1y + l2y + l3y);
  } else {
    const double l1z = +g1 * x * y, l2z = 0.0, l3z = -g2 * x * y;
    const double cz = -(p.phi[2] * p.span[1] * x / vol);
    Ac[c] = Ac[c] + cz + third * (l1z + l2z + l3z);
  }
}

// d/dq along one axis at index q of n (stride s), derivq :852-870
__device__ __forceinline__ double ddq(const double *__restrict__ v, size_t c, int q, int n, size_t s, double h) {
  const double half = 0.5;
  double d = 0.0;
  if (q == 0) {
    d = d + v[c] * (-3 * half / h);
    d = d + v[c + s] * (+4 * half / h);
    d = d + v[c + 2 * s] * (-1 * half / h);
  } else if (q == n - 1) {
    d = d + v[c] * (+3 * half / h);
    d = d + v[c - s] * (-4 * half / h);
    d = d + v[c - 2 * s] * (+1 * half / h);
  } else {
    d = d + v[c - s] * (-1 * half / h);
    d = d + v[c + s] * (+1 * half / h);
  }
  return d;
}

__global__ __launch_bounds__(256) void curl_k(const double *__restrict__ A, double *__restrict__ B, PostArgs p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int kb = blockIdx.z;          // local plane of B
  if (i >= p.n[0] || j >= p.n[1]) return;
  const size_t sy = (size_t)p.n[0], sz = (size_t)p.n[0] * p.n[1];
  const size_t N = sz * p.na, NB = sz * p.nb;
  const int k = kb + p.boff;          // local plane of A
  const int kg = p.kg0 + k;           // global plane: decides between centred and one-sided z differences
  const size_t c = (size_t)i + sy * (size_t)j + sz * (size_t)k;
  const size_t cb = (size_t)i + sy * (size_t)j + sz * (size_t)kb;
  const double *Ax = A, *Ay = A + N, *Az = A + 2 * N;
  const double axy = ddq(Ax, c, j, p.n[1], sy, p.dq[1]);
  const double axz = ddq(Ax, c, kg, p.n[2], sz, p.dq[2]);
  const double ayx = ddq(Ay, c, i, p.n[0], 1, p.dq[0]);
  const double ayz = ddq(Ay, c, kg, p.n[2], sz, p.dq[2]);
  const double azx = ddq(Az, c, i, p.n[0], 1, p.dq[0]);
  const double azy = ddq(Az, c, j, p.n[1], sy, p.dq[1]);
  B[cb] = azy - ayz;          // :802-804
  B[cb + NB] = axz - azx;
  B[cb + 2 * NB] = ayx - axy;
}

// one component of the curl (same differences, same expressions as curl_k): component C of B needs only the
// OTHER two components of A, so B_z can be formed - and go home - while A_z is still being solved
template <int C>
__global__ __launch_bounds__(256) void curl_comp_k(const double *__restrict__ A, double *__restrict__ B, PostArgs p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= p.n[0] || j >= p.n[1]) return;
  const size_t sy = (size_t)p.n[0], sz = (size_t)p.n[0] * p.n[1];
  const size_t N = sz * p.na;
  const size_t c = (size_t)i + sy * (size_t)j + sz * (size_t)k;
  const double *Ax = A, *Ay = A + N, *Az = A + 2 * N;
  if (C == 0) {
    const double ayz = ddq(Ay, c, k, p.n[2], sz, p.dq[2]);
    const double azy = ddq(Az, c, j, p.n[1], sy, p.dq[1]);
    B[c] = azy - ayz;          // :802
  } else if (C == 1) {
    const double axz = ddq(Ax, c, k, p.n[2], sz, p.dq[2]);
    const double azx = ddq(Az, c, i, p.n[0], 1, p.dq[0]);
    B[c + N] = axz - azx;      // :803
  } else {
    const double axy = ddq(Ax, c, j, p.n[1], sy, p.dq[1]);
    const double ayx = ddq(Ay, c, i, p.n[0], 1, p.dq[0]);
    B[c + 2 * N] = ayx - axy;  // :804
  }
}

}  // namespace

// A: device array (nx,ny,na,3) holding global planes [kg0, kg0 + na); B: (nx,ny,nb,3), its plane 0
// = A's local plane boff.  x,y,z: DEVICE mesh vectors of the GLOBAL grid n3.  curl_first != 0
// selects the reference's IOPT_FLXCRL == 1 order (:455-465).  The flux-balance fields are added to
// every plane of A (ghosts included: the curl that follows differentiates them).
extern "C" int ndsmk_balance_curl_slab(double *A, double *B, const int32_t *n3, int kg0, int na, int nb, int boff,
                                       const double *x, const double *y, const double *z, const double *h_phi6,
                                       const double *h_span3, const double *h_dq3, int curl_first) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(n3[0] >= 3 && n3[1] >= 3 && n3[2] >= 3);  // one-sided stencils need three points
  NDSM_CHECK_ARG(kg0 >= 0 && na >= 1 && kg0 + na <= n3[2] && nb >= 1 && boff >= 0 && boff + nb <= na);
  // the z stencil of every B plane must lie inside A: centred needs a plane either side,
  // the one-sided ones at the global ends three planes inwards
  NDSM_CHECK_ARG((kg0 + boff == 0 ? na - boff >= 3 : boff >= 1) &&
                 (kg0 + boff + nb == n3[2] ? boff + nb >= 3 : boff + nb < na));
  PostArgs p;
  p.kg0 = kg0;
  p.na = na;
  p.nb = nb;
  p.boff = boff;
  for (int d = 0; d < 3; ++d) {
    p.n[d] = n3[d];
    p.span[d] = h_span3[d];
    p.dq[d] = h_dq3[d];
  }
  for (int f = 0; f < 6; ++f) p.phi[f] = h_phi6[f];
  p.x = x;
  p.y = y;
  p.z = z;
  dim3 block(64, 4, 1);
  dim3 grida((n3[0] + 63) / 64, (n3[1] + 3) / 4, na), gridb((n3[0] + 63) / 64, (n3[1] + 3) / 4, nb);
  hipStream_t s = ndsm::stream();
  if (curl_first) {
    hipLaunchKernelGGL(curl_k, gridb, block, 0, s, A, B, p);
    NDSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(balance_k<true>, grida, block, 0, s, A, B, p);
    NDSM_LAUNCH_CHECK();
  } else {
    // the reference also adds the linear field to B here (:474), but its curl
    // (:475) then overwrites all of B: that dead update is skipped
    hipLaunchKernelGGL(balance_k<false>, grida, block, 0, s, A, B, p);
    NDSM_LAUNCH_CHECK();
    hipLaunchKernelGGL(curl_k, gridb, block, 0, s, A, B, p);
    NDSM_LAUNCH_CHECK();
  }
  return 0;
}

// Ac: DEVICE component c (0,1,2) of A, (nx,ny,nz); x,y,z DEVICE mesh vectors
extern "C" int ndsmk_balance_component(double *Ac, const int32_t *n3, int c, const double *x, const double *y,
                                       const double *z, const double *h_phi6, const double *h_span3) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(Ac && c >= 0 && c < 3);
  PostArgs p;
  p.kg0 = 0;
  p.na = p.nb = n3[2];
  p.boff = 0;
  for (int d = 0; d < 3; ++d) {
    p.n[d] = n3[d];
    p.span[d] = h_span3[d];
    p.dq[d] = 0.0;
  }
  for (int f = 0; f < 6; ++f) p.phi[f] = h_phi6[f];
  p.x = x;
  p.y = y;
  p.z = z;
  dim3 block(64, 4, 1);
  dim3 grid((n3[0] + 63) / 64, (n3[1] + 3) / 4, n3[2]);
  hipStream_t s = ndsm::stream();
  if (c == 0)
    hipLaunchKernelGGL(balance_comp_k<0>, grid, block, 0, s, Ac, p);
  else if (c == 1)
    hipLaunchKernelGGL(balance_comp_k<1>, grid, block, 0, s, Ac, p);
  else
    hipLaunchKernelGGL(balance_comp_k<2>, grid, block, 0, s, Ac, p);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// B = curl A alone, whole field
extern "C" int ndsmk_curl(const double *A, double *B, const int32_t *n3, const double *h_dq3) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(A && B && n3[0] >= 3 && n3[1] >= 3 && n3[2] >= 3);
  PostArgs p;
  p.kg0 = 0;
  p.na = p.nb = n3[2];
  p.boff = 0;
  for (int d = 0; d < 3; ++d) {
    p.n[d] = n3[d];
    p.span[d] = 0.0;
    p.dq[d] = h_dq3[d];
  }
  for (int f = 0; f < 6; ++f) p.phi[f] = 0.0;
  p.x = p.y = p.z = nullptr;
  dim3 block(64, 4, 1);
  dim3 grid((n3[0] + 63) / 64, (n3[1] + 3) / 4, n3[2]);
  hipLaunchKernelGGL(curl_k, grid, block, 0, ndsm::stream(), A, B, p);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// component c (0, 1, 2) of B = curl A alone, whole field: reads the other two components of A only
extern "C" int ndsmk_curl_component(const double *A, double *B, const int32_t *n3, const double *h_dq3, int c) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(A && B && n3[0] >= 3 && n3[1] >= 3 && n3[2] >= 3 && c >= 0 && c < 3);
  PostArgs p;
  p.kg0 = 0;
  p.na = p.nb = n3[2];
  p.boff = 0;
  for (int d = 0; d < 3; ++d) {
    p.n[d] = n3[d];
    p.span[d] = 0.0;
    p.dq[d] = h_dq3[d];
  }
  for (int f = 0; f < 6; ++f) p.phi[f] = 0.0;
  p.x = p.y = p.z = nullptr;
  dim3 block(64, 4, 1);
  dim3 grid((n3[0] + 63) / 64, (n3[1] + 3) / 4, n3[2]);
  if (c == 0)
    hipLaunchKernelGGL(curl_comp_k<0>, grid, block, 0, ndsm::stream(), A, B, p);
  else if (c == 1)
    hipLaunchKernelGGL(curl_comp_k<1>, grid, block, 0, ndsm::stream(), A, B, p);
  else
    hipLaunchKernelGGL(curl_comp_k<2>, grid, block, 0, ndsm::stream(), A, B, p);
  NDSM_LAUNCH_CHECK();
  return 0;
}

// the whole field in one piece
extern "C" int ndsmk_balance_curl(double *A, double *B, const int32_t *n3, const double *x, const double *y,
                                  const double *z, const double *h_phi6, const double *h_span3,
                                  const double *h_dq3, int curl_first) {
  return ndsmk_balance_curl_slab(A, B, n3, 0, n3[2], n3[2], 0, x, y, z, h_phi6, h_span3, h_dq3, curl_first);
}
