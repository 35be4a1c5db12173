/* TEST INFRASTRUCTURE - NOT THE PRODUCT.  See ndsm_oracle.h for the contract.
 *
 * Plain-C restatement of the reference's CPU algorithm, function by function,
 * each citing the reference file:line it follows (paths relative to
 * /root/reference/fortran/).  Floating-point expressions keep the reference's
 * operand order; the file is compiled with -ffp-contract=off.
 */
#include "ndsm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define MAXD 3

static int g_quiet = 0;
void orc_set_quiet(int quiet) { g_quiet = quiet; }

/* ------------------------------------------------------------------ */
/* hierarchy (ndsm_multigrid_core.f90:86-101, 165-270)                  */
/* ------------------------------------------------------------------ */

typedef struct {
  int ndim, ngrids;
  int ms, du_max, nmax_exact;
  double ex_tol;
  char bcs[2 * MAXD];
  int64_t (*nshape)[MAXD]; /* [ngrids][MAXD] */
  int64_t *nsize;          /* [ngrids] */
  double *(*mesh)[MAXD];   /* [ngrids][MAXD] */
  double **u, **rhs;       /* [ngrids] */
  int64_t exact_sweeps;
} mg_t;

/* ndsm_vector_potential.f90:60,631-632 (BASE_GRID = 2) */
int orc_ngrids(int ndim, const int64_t *nshape) {
  int64_t nmin = nshape[0];
  for (int d = 1; d < ndim; ++d)
    if (nshape[d] < nmin) nmin = nshape[d];
  return (int)floor(log((double)nmin / 2.0) / log(2.0));
}

static double vmin(const double *v, int64_t n) {
  double m = v[0];
  for (int64_t i = 1; i < n; ++i)
    if (v[i] < m) m = v[i];
  return m;
}
static double vmax(const double *v, int64_t n) {
  double m = v[0];
  for (int64_t i = 1; i < n; ++i)
    if (v[i] > m) m = v[i];
  return m;
}

static void mg_new(mg_t *h, int ndim, const int64_t *nshape, int ngrids, const double *const *mesh,
                   int du_max, int nmax_exact) {
  memset(h, 0, sizeof(*h));
  h->ndim = ndim;
  h->ngrids = ngrids;
  h->du_max = du_max;
  h->nmax_exact = nmax_exact;
  h->nshape = calloc((size_t)ngrids, sizeof(*h->nshape));
  h->nsize = calloc((size_t)ngrids, sizeof(*h->nsize));
  h->mesh = calloc((size_t)ngrids, sizeof(*h->mesh));
  h->u = calloc((size_t)ngrids, sizeof(double *));
  h->rhs = calloc((size_t)ngrids, sizeof(double *));
  for (int d = 0; d < MAXD; ++d) h->nshape[0][d] = d < ndim ? nshape[d] : 1;
  /* :215-217  nshape(:,i) = MAX(FLOOR(nshape(:,i-1)*0.5),1) */
  for (int l = 1; l < ngrids; ++l)
    for (int d = 0; d < MAXD; ++d) {
      int64_t n = d < ndim ? (int64_t)floor((double)h->nshape[l - 1][d] * 0.5) : 1;
      h->nshape[l][d] = n > 1 ? n : 1;
    }
  for (int l = 0; l < ngrids; ++l) {
    h->nsize[l] = 1;
    for (int d = 0; d < ndim; ++d) h->nsize[l] *= h->nshape[l][d];
  }
  /* :231-238 finest mesh is a copy */
  for (int d = 0; d < ndim; ++d) {
    h->mesh[0][d] = malloc(sizeof(double) * (size_t)nshape[d]);
    memcpy(h->mesh[0][d], mesh[d], sizeof(double) * (size_t)nshape[d]);
  }
  /* :243-263 coarser meshes span the same extent */
  for (int l = 1; l < ngrids; ++l)
    for (int d = 0; d < ndim; ++d) {
      int64_t nq = h->nshape[l][d];
      h->mesh[l][d] = malloc(sizeof(double) * (size_t)nq);
      double qil = vmin(mesh[d], nshape[d]);
      double Lq = vmax(mesh[d], nshape[d]) - qil;
      for (int64_t j = 1; j <= nq; ++j)
        h->mesh[l][d][j - 1] = (double)(j - 1) * Lq / (double)(nq - 1) + qil;
    }
}

static void mg_delete(mg_t *h) {
  for (int l = 0; l < h->ngrids; ++l) {
    free(h->u[l]);
    free(h->rhs[l]);
    for (int d = 0; d < h->ndim; ++d) free(h->mesh[l][d]);
  }
  free(h->u);
  free(h->rhs);
  free(h->mesh);
  free(h->nshape);
  free(h->nsize);
  memset(h, 0, sizeof(*h));
}

void orc_hierarchy(int ndim, const int64_t *nshape, int ngrids, const double *x, const double *y,
                   const double *z, int64_t *shapes_out, double *mesh_out) {
  const double *mesh[MAXD] = {x, y, z};
  mg_t h;
  mg_new(&h, ndim, nshape, ngrids, mesh, 1, 1);
  int64_t p = 0;
  for (int l = 0; l < ngrids; ++l)
    for (int d = 0; d < ndim; ++d) {
      int64_t n = h.nshape[l][d];
      shapes_out[l * ndim + d] = n;
      memcpy(mesh_out + p, h.mesh[l][d], sizeof(double) * (size_t)n);
      p += n;
    }
  mg_delete(&h);
}

/* ------------------------------------------------------------------ */
/* 3-D kernels (ndsm_optimized.f90)                                    */
/* ------------------------------------------------------------------ */

#define IDX3(i, j, k) ((i) + nx * ((j) + ny * (k)))

/* ndsm_optimized.f90:40-191 red_black_gauss_3D */
static void relax3d(const char *bcs, int64_t nx, int64_t ny, int64_t nz, const double *x,
                    const double *y, const double *z, const double *rhs, double *u) {
  /* :68-76 loop bounds shrink by one on Dirichlet faces (0-based here) */
  int64_t lb[3] = {0, 0, 0}, ub[3] = {nx - 1, ny - 1, nz - 1};
  for (int d = 0; d < 3; ++d) {
    if (bcs[d] == 'D') lb[d] += 1;
    if (bcs[3 + d] == 'D') ub[d] -= 1;
  }
  /* :79-94 */
  double hx = x[1] - x[0], hy = y[1] - y[0], hz = z[1] - z[0];
  double wx = 1.0 / (hx * hx), wy = 1.0 / (hy * hy), wz = 1.0 / (hz * hz);
  double w1 = 2 * (wx + wy + wz);
  w1 = 1.0 / w1;
  /* :106 first colour: 1-based i+j+k == lb(1) (mod 2)  <=>  0-based
   * (i+j+k) mod 2 == [x-lower is 'D'];  :139 second colour: the rest. */
  int first = (bcs[0] == 'D') ? 1 : 0;
  for (int pass = 0; pass < 2; ++pass) {
    int par = (first + pass) & 1;
#pragma omp parallel for schedule(static)
    for (int64_t k = lb[2]; k <= ub[2]; ++k)
      for (int64_t j = lb[1]; j <= ub[1]; ++j) {
        int64_t i0 = lb[0] + ((((lb[0] + j + k) & 1) != par) ? 1 : 0);
        for (int64_t i = i0; i <= ub[0]; i += 2) {
          /* :109-120 homogeneous Neumann by mirroring */
          int64_t xl = i - 1, xh = i + 1, yl = j - 1, yh = j + 1, zl = k - 1, zh = k + 1;
          if (xl < 0) xl = 1;
          if (xh > nx - 1) xh = nx - 2;
          if (yl < 0) yl = 1;
          if (yh > ny - 1) yh = ny - 2;
          if (zl < 0) zl = 1;
          if (zh > nz - 1) zh = nz - 2;
          /* :123-129 */
          double unew = (u[IDX3(xh, j, k)] + u[IDX3(xl, j, k)]) * wx +
                        (u[IDX3(i, yh, k)] + u[IDX3(i, yl, k)]) * wy +
                        (u[IDX3(i, j, zh)] + u[IDX3(i, j, zl)]) * wz - rhs[IDX3(i, j, k)];
          u[IDX3(i, j, k)] = w1 * unew;
        }
      }
  }
  /* :173-189 mean subtraction if all six faces are Neumann */
  int alln = 1;
  for (int d = 0; d < 6; ++d) alln = alln && bcs[d] == 'N';
  if (alln) {
    int64_t n = nx * ny * nz;
    double m = 0;
    for (int64_t q = 0; q < n; ++q) m = m + u[q];
    m = m / (double)n;
    for (int64_t q = 0; q < n; ++q) u[q] = u[q] - m;
  }
}

/* ndsm_optimized.f90:346-447 poisson_residual_3D (r = rhs - L u) */
static void residual3d(const char *bcs, int64_t nx, int64_t ny, int64_t nz, const double *x,
                       const double *y, const double *z, const double *rhs, const double *u,
                       double *r) {
  int64_t lb[3] = {0, 0, 0}, ub[3] = {nx - 1, ny - 1, nz - 1};
  for (int d = 0; d < 3; ++d) {
    if (bcs[d] == 'D') lb[d] += 1;
    if (bcs[3 + d] == 'D') ub[d] -= 1;
  }
  double hx = x[1] - x[0], hy = y[1] - y[0], hz = z[1] - z[0];
  double wx = 1.0 / (hx * hx), wy = 1.0 / (hy * hy), wz = 1.0 / (hz * hz);
  double wc = 2 * (wx + wy + wz);
  /* :389-397 r = 0 ; the points the next loop skips are exactly the
   * Dirichlet faces that :439-445 zero again */
  memset(r, 0, sizeof(double) * (size_t)(nx * ny * nz));
#pragma omp parallel for schedule(static)
  for (int64_t k = lb[2]; k <= ub[2]; ++k)
    for (int64_t j = lb[1]; j <= ub[1]; ++j)
      for (int64_t i = lb[0]; i <= ub[0]; ++i) {
        int64_t xl = i - 1, xh = i + 1, yl = j - 1, yh = j + 1, zl = k - 1, zh = k + 1;
        if (xl < 0) xl = 1;
        if (xh > nx - 1) xh = nx - 2;
        if (yl < 0) yl = 1;
        if (yh > ny - 1) yh = ny - 2;
        if (zl < 0) zl = 1;
        if (zh > nz - 1) zh = nz - 2;
        /* :424-430 */
        double v = (u[IDX3(xl, j, k)] + u[IDX3(xh, j, k)]) * wx +
                   (u[IDX3(i, yl, k)] + u[IDX3(i, yh, k)]) * wy +
                   (u[IDX3(i, j, zl)] + u[IDX3(i, j, zh)]) * wz - rhs[IDX3(i, j, k)] -
                   u[IDX3(i, j, k)] * wc;
        r[IDX3(i, j, k)] = -v;
      }
}

void orc_relax3d(const int64_t *n, const double *x, const double *y, const double *z,
                 const char *bcs, const double *rhs, double *u) {
  relax3d(bcs, n[0], n[1], n[2], x, y, z, rhs, u);
}
void orc_residual3d(const int64_t *n, const double *x, const double *y, const double *z,
                    const char *bcs, const double *rhs, const double *u, double *r) {
  residual3d(bcs, n[0], n[1], n[2], x, y, z, rhs, u, r);
}

/* ------------------------------------------------------------------ */
/* generic N-D kernels (ndsm_poisson.f90), reached for ndim != 3        */
/* ------------------------------------------------------------------ */

/* ndsm_root.f90:195 lin2nd (0-based here), x fastest */
static void lin2nd(int ndim, const int64_t *nshape, int64_t n, int64_t *ivec) {
  for (int d = 0; d < ndim; ++d) {
    ivec[d] = n % nshape[d];
    n /= nshape[d];
  }
}

/* ndsm_poisson.f90:409-433 boundary_mask + :361-390 at_dirichlet_boundary */
static int at_dirichlet(int ndim, const int64_t *ivec, const int64_t *nshape, const char *bcs) {
  for (int d = 0; d < ndim; ++d) {
    if (ivec[d] == 0 && bcs[d] == 'D') return 1;
    /* ELSEIF: a one-point dimension counts as "lower" only (:424-427) */
    if (ivec[d] != 0 && ivec[d] == nshape[d] - 1 && bcs[ndim + d] == 'D') return 1;
  }
  return 0;
}

/* ndsm_poisson.f90:633-658 stencil_stride */
static void stencil_stride(const int64_t *ivec, const int64_t *nshape, const int64_t *strides,
                           int d, int64_t *dn) {
  if (ivec[d] == 0) {
    dn[0] = dn[1] = +strides[d];
  } else if (ivec[d] == nshape[d] - 1) {
    dn[0] = dn[1] = -strides[d];
  } else {
    dn[0] = -strides[d];
    dn[1] = +strides[d];
  }
}

/* ndsm_poisson.f90:451-549 relax (with :557-619 relax_stencil_update) */
static void relax_nd(int ndim, const int64_t *nshape, const double *const *q, const char *bcs,
                     double *u, const double *rhs) {
  int64_t strides[MAXD], nsize = 1;
  for (int d = 0; d < ndim; ++d) {
    strides[d] = nsize;
    nsize *= nshape[d];
  }
  /* :483-489 */
  double wc[MAXD + 1];
  wc[0] = 0;
  for (int d = 1; d <= ndim; ++d) {
    double dq = q[d - 1][1] - q[d - 1][0];
    wc[d] = 1.0 / (dq * dq);
    wc[0] = wc[0] + 2.0 * wc[d];
  }
  wc[0] = 1.0 / wc[0];
  /* :494-527 red = all 1-based index parities equal; then the rest */
  for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < nsize; ++n) {
      int64_t ivec[MAXD];
      lin2nd(ndim, nshape, n, ivec);
      int all0 = 1, all1 = 1;
      for (int d = 0; d < ndim; ++d) {
        int par = (int)((ivec[d] + 1) & 1);
        all0 = all0 && par == 0;
        all1 = all1 && par == 1;
      }
      int is_red = all0 || all1;
      if (is_red != (pass == 0)) continue;
      if (at_dirichlet(ndim, ivec, nshape, bcs)) continue; /* :591-594 */
      double un = 0; /* :603-617 */
      for (int d = 0; d < ndim; ++d) {
        int64_t dn[2];
        stencil_stride(ivec, nshape, strides, d, dn);
        un = un + u[n + dn[0]] * wc[d + 1] + u[n + dn[1]] * wc[d + 1];
      }
      u[n] = (un - rhs[n]) * wc[0];
    }
  }
  /* :534-547 */
  int alln = 1;
  for (int d = 0; d < 2 * ndim; ++d) alln = alln && bcs[d] == 'N';
  if (alln) {
    double m = 0; /* ndsm_multigrid_core.f90:1199-1223 mean */
    for (int64_t n = 0; n < nsize; ++n) m = m + u[n];
    m = m / (double)nsize;
    for (int64_t n = 0; n < nsize; ++n) u[n] = u[n] - m;
  }
}

/* ndsm_poisson.f90:280-353 residual */
static void residual_nd(int ndim, const int64_t *nshape, const double *const *q, const char *bcs,
                        const double *u, const double *rhs, double *r) {
  int64_t strides[MAXD], nsize = 1;
  for (int d = 0; d < ndim; ++d) {
    strides[d] = nsize;
    nsize *= nshape[d];
  }
  double wc[MAXD];
  for (int d = 0; d < ndim; ++d) {
    double dq = q[d][1] - q[d][0];
    wc[d] = 1.0 / (dq * dq);
  }
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nsize; ++n) {
    int64_t ivec[MAXD];
    lin2nd(ndim, nshape, n, ivec);
    if (at_dirichlet(ndim, ivec, nshape, bcs)) {
      r[n] = 0;
      continue;
    }
    double lap = 0;
    for (int d = 0; d < ndim; ++d) {
      int64_t dn[2];
      stencil_stride(ivec, nshape, strides, d, dn);
      lap = lap + (u[n + dn[0]] - 2 * u[n] + u[n + dn[1]]) * wc[d]; /* :343 */
    }
    r[n] = rhs[n] - lap; /* :348 */
  }
}

void orc_relax_nd(int ndim, const int64_t *nshape, const double *x, const double *y, const double *z,
                  const char *bcs, const double *rhs, double *u) {
  const double *q[MAXD] = {x, y, z};
  relax_nd(ndim, nshape, q, bcs, u, rhs);
}
void orc_residual_nd(int ndim, const int64_t *nshape, const double *x, const double *y,
                     const double *z, const char *bcs, const double *rhs, const double *u,
                     double *r) {
  const double *q[MAXD] = {x, y, z};
  residual_nd(ndim, nshape, q, bcs, u, rhs, r);
}

/* operator plug-ins: ndsm_poisson.f90:224-272 / :163-216 (3-D -> optimized) */
static void relax_level(mg_t *h, int l) {
  if (h->ndim == 3)
    relax3d(h->bcs, h->nshape[l][0], h->nshape[l][1], h->nshape[l][2], h->mesh[l][0],
            h->mesh[l][1], h->mesh[l][2], h->rhs[l], h->u[l]);
  else
    relax_nd(h->ndim, h->nshape[l], (const double *const *)h->mesh[l], h->bcs, h->u[l], h->rhs[l]);
}
static void residual_level(mg_t *h, int l, double *r) {
  if (h->ndim == 3)
    residual3d(h->bcs, h->nshape[l][0], h->nshape[l][1], h->nshape[l][2], h->mesh[l][0],
               h->mesh[l][1], h->mesh[l][2], h->rhs[l], h->u[l], r);
  else
    residual_nd(h->ndim, h->nshape[l], (const double *const *)h->mesh[l], h->bcs, h->u[l],
                h->rhs[l], r);
}

/* ------------------------------------------------------------------ */
/* transfer operators (ndsm_interp.f90)                                */
/* ------------------------------------------------------------------ */

/* ndsm_interp.f90:373-435 ; returns 1-based qil/qih like the reference */
static void find_bracket(const double *qvec, int64_t nq, double q0, int64_t *qil, int64_t *qih,
                         int *ierr) {
  if (q0 <= qvec[0]) {
    *qil = 1;
    *qih = 2;
    *ierr = -1;
    return;
  }
  if (q0 >= qvec[nq - 1]) {
    *qil = nq - 1;
    *qih = nq;
    *ierr = +1;
    return;
  }
  double dq = qvec[1] - qvec[0];
  *qil = (int64_t)floor((q0 - qvec[0]) / dq) + 1;
  if (*qil >= nq) {
    *qil = nq - 1;
    *qih = nq;
  } else {
    *qih = *qil + 1;
  }
  *ierr = 0;
}

/* ndsm_interp.f90:85-158 ninterp (+ :314-365 get_interpolation_values) */
static double ninterp(int ndim, const int64_t *nshape, const double *const *qv, const double *q0,
                      const double *f) {
  int64_t bp[MAXD][2], strides[MAXD], s = 1;
  for (int d = 0; d < ndim; ++d) {
    int ierr;
    find_bracket(qv[d], nshape[d], q0[d], &bp[d][0], &bp[d][1], &ierr);
    strides[d] = s;
    s *= nshape[d];
  }
  double fs[1 << MAXD];
  int nc = 1 << ndim;
  for (int n = 0; n < nc; ++n) { /* :340-363 bit d of n picks lower/upper in dim d */
    int64_t j = 0;
    for (int d = 0; d < ndim; ++d) j += (bp[d][(n >> d) & 1] - 1) * strides[d];
    fs[n] = f[j];
  }
  for (int d = ndim - 1; d >= 0; --d) { /* :128-154 last dimension first */
    double ql = qv[d][bp[d][0] - 1], qh = qv[d][bp[d][1] - 1];
    double dq = qh - ql;
    double wl = +(q0[d] - ql) / dq;
    double wh = -(q0[d] - qh) / dq;
    int NC = 1 << d;
    for (int j = 0; j < NC; ++j) fs[j] = wh * fs[j] + wl * fs[j + NC];
  }
  return fs[0];
}

/* ndsm_interp.f90:186-292 nrestrict */
static double nrestrict(int ndim, const int64_t *nshape_f, const double *const *qc,
                        const double *const *qf, const double *q0, const double *f) {
  int64_t bp[MAXD][2], ns[MAXD], strides[MAXD], s = 1, nsize_s = 1;
  double dq_c[MAXD], dq_f[MAXD], w2[MAXD];
  for (int d = 0; d < ndim; ++d) {
    dq_c[d] = qc[d][1] - qc[d][0];
    dq_f[d] = qf[d][1] - qf[d][0];
    w2[d] = dq_f[d] / (dq_c[d] * dq_c[d]); /* :229 */
    int64_t qil, qih;
    int ierr;
    find_bracket(qf[d], nshape_f[d], q0[d] - dq_c[d], &qil, &qih, &ierr); /* :234-239 */
    bp[d][0] = ierr < 0 ? qil : qih;
    find_bracket(qf[d], nshape_f[d], q0[d] + dq_c[d], &qil, &qih, &ierr); /* :242-247 */
    bp[d][1] = ierr > 0 ? qih : qil;
    ns[d] = bp[d][1] - bp[d][0] + 1;
    nsize_s *= ns[d];
    strides[d] = s;
    s *= nshape_f[d];
  }
  double fc = 0;
  for (int64_t j = 0; j < nsize_s; ++j) { /* :263-290 stencil index, x fastest */
    int64_t iv[MAXD], n = 0;
    lin2nd(ndim, ns, j, iv);
    double w = 1;
    for (int d = 0; d < ndim; ++d) {
      int64_t qi = bp[d][0] + iv[d]; /* 1-based */
      double qq = qf[d][qi - 1];
      double c1 = fabs(qq - q0[d]);
      double c2 = fabs(dq_c[d] - c1);
      w = w * c2 * w2[d];
      n += (qi - 1) * strides[d];
    }
    fc = fc + w * f[n];
  }
  return fc;
}

/* ndsm_multigrid_core.f90:865-921 mg_interp ; lf = 0-based fine level */
static void mg_interp(const mg_t *h, int lf, const double *u_c, double *u_f) {
  int lc = lf + 1, ndim = h->ndim;
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < h->nsize[lf]; ++n) {
    int64_t iv[MAXD];
    double q0[MAXD];
    lin2nd(ndim, h->nshape[lf], n, iv);
    for (int d = 0; d < ndim; ++d) q0[d] = h->mesh[lf][d][iv[d]];
    u_f[n] = ninterp(ndim, h->nshape[lc], (const double *const *)h->mesh[lc], q0, u_c);
  }
}

/* ndsm_multigrid_core.f90:1010-1065 mg_restrict */
static void mg_restrict(const mg_t *h, int lf, const double *u_f, double *u_c) {
  int lc = lf + 1, ndim = h->ndim;
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < h->nsize[lc]; ++n) {
    int64_t iv[MAXD];
    double q0[MAXD];
    lin2nd(ndim, h->nshape[lc], n, iv);
    for (int d = 0; d < ndim; ++d) q0[d] = h->mesh[lc][d][iv[d]];
    u_c[n] = nrestrict(ndim, h->nshape[lf], (const double *const *)h->mesh[lc],
                       (const double *const *)h->mesh[lf], q0, u_f);
  }
}

void orc_restrict(int ndim, const int64_t *nshape, int ngrids, const double *x, const double *y,
                  const double *z, int id_f, const double *u_f, double *u_c) {
  const double *mesh[MAXD] = {x, y, z};
  mg_t h;
  mg_new(&h, ndim, nshape, ngrids, mesh, 1, 1);
  mg_restrict(&h, id_f - 1, u_f, u_c);
  mg_delete(&h);
}
void orc_interp(int ndim, const int64_t *nshape, int ngrids, const double *x, const double *y,
                const double *z, int id_f, const double *u_c, double *u_f) {
  const double *mesh[MAXD] = {x, y, z};
  mg_t h;
  mg_new(&h, ndim, nshape, ngrids, mesh, 1, 1);
  mg_interp(&h, id_f - 1, u_c, u_f);
  mg_delete(&h);
}

/* ------------------------------------------------------------------ */
/* V-cycle (ndsm_multigrid_core.f90)                                   */
/* ------------------------------------------------------------------ */

/* :808-853 du_metrics */
static void du_metrics(int64_t n, const double *u1, const double *u2, double *m) {
  double mx = 0, sm = 0;
  for (int64_t i = 0; i < n; ++i) {
    double d = fabs(u1[i] - u2[i]);
    mx = d > mx ? d : mx;
    sm = sm + d;
  }
  m[0] = mx;
  m[1] = sm / (double)n;
}

/* :1077-1122 update_u : metrics of |u_new - u_old|, then u_new <- u_old */
void orc_update_u(int64_t nsize, const double *u_old, double *u_new, double *metrics) {
  double mx = 0, sm = 0;
  for (int64_t i = 0; i < nsize; ++i) {
    double d = fabs(u_new[i] - u_old[i]);
    mx = d > mx ? d : mx;
    sm = sm + d;
    u_new[i] = u_old[i];
  }
  metrics[0] = mx;
  metrics[1] = sm / (double)nsize;
}

/* :482-560 fine_to_coarse ; l = 0-based fine level */
static void fine_to_coarse(mg_t *h, int l) {
  for (int i = 0; i < h->ms; ++i) relax_level(h, l); /* :523-525 */
  double *r = calloc((size_t)h->nsize[l], sizeof(double));
  residual_level(h, l, r); /* :539 */
  h->rhs[l + 1] = calloc((size_t)h->nsize[l + 1], sizeof(double));
  mg_restrict(h, l, r, h->rhs[l + 1]); /* :551 */
  free(r);
  h->u[l + 1] = calloc((size_t)h->nsize[l + 1], sizeof(double)); /* :557-558 */
}

/* :593-684 coarse_to_fine ; lc = 0-based coarse level */
static void coarse_to_fine(mg_t *h, int lc) {
  int lf = lc - 1;
  for (int i = 0; i < h->ms; ++i) relax_level(h, lc); /* :642-644 */
  free(h->rhs[lc]);
  h->rhs[lc] = NULL;
  double *cor = calloc((size_t)h->nsize[lf], sizeof(double));
  mg_interp(h, lf, h->u[lc], cor); /* :659 */
  free(h->u[lc]);
  h->u[lc] = NULL;
  for (int64_t i = 0; i < h->nsize[lf]; ++i) h->u[lf][i] = h->u[lf][i] + cor[i]; /* :692-712 */
  free(cor);
  for (int i = 0; i < h->ms; ++i) relax_level(h, lf); /* :680-682 */
}

/* :728-800 solve_exact */
static void solve_exact(mg_t *h, int l) {
  int64_t n = h->nsize[l];
  double *sav = calloc((size_t)n, sizeof(double));
  double du = DBL_MAX, m[2];
  int converged = 0;
  for (int i = 0; i < h->nmax_exact; ++i) {
    if (du <= h->ex_tol) { /* test first (:771-774) */
      converged = 1;
      break;
    }
    relax_level(h, l);
    du_metrics(n, sav, h->u[l], m);
    du = h->du_max ? m[0] : m[1];
    memcpy(sav, h->u[l], sizeof(double) * (size_t)n);
    h->exact_sweeps++;
  }
  if (!converged && !g_quiet)
    printf(" Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged\n");
  free(sav);
}

/* :341-377 v_cycle from the finest grid */
static void v_cycle(mg_t *h) {
  for (int l = 0; l < h->ngrids - 1; ++l) fine_to_coarse(h, l);
  solve_exact(h, h->ngrids - 1);
  for (int l = h->ngrids - 1; l >= 1; --l) coarse_to_fine(h, l);
}

static void setup_bvp(mg_t *h, int ndim, const int64_t *nshape, int ngrids, const double *x,
                      const double *y, const double *z, const char *bcs, int ms, double ex_tol,
                      int du_max, int nmax_exact) {
  const double *mesh[MAXD] = {x, y, z};
  mg_new(h, ndim, nshape, ngrids, mesh, du_max, nmax_exact);
  h->ms = ms;
  h->ex_tol = ex_tol;
  memcpy(h->bcs, bcs, (size_t)(2 * ndim));
}

void orc_vcycle(int ndim, const int64_t *nshape, int ngrids, const double *x, const double *y,
                const double *z, const char *bcs, int ms, double ex_tol, int du_max, int nmax_exact,
                const double *rhs, double *u) {
  mg_t h;
  setup_bvp(&h, ndim, nshape, ngrids, x, y, z, bcs, ms, ex_tol, du_max, nmax_exact);
  size_t bytes = sizeof(double) * (size_t)h.nsize[0];
  h.u[0] = malloc(bytes);
  h.rhs[0] = malloc(bytes);
  memcpy(h.u[0], u, bytes);
  memcpy(h.rhs[0], rhs, bytes);
  v_cycle(&h);
  memcpy(u, h.u[0], bytes);
  mg_delete(&h);
}

/* ndsm_poisson.f90:63-155 solve_poisson_bvp on an existing handle */
static int solve_bvp(mg_t *h, double vc_tol, int nmax, double *u, const double *rhs,
                     double *du_last, double *hist, int hist_len, int *ncycles) {
  size_t bytes = sizeof(double) * (size_t)h->nsize[0];
  if (!h->u[0]) h->u[0] = malloc(bytes);
  if (!h->rhs[0]) h->rhs[0] = malloc(bytes);
  memcpy(h->u[0], u, bytes);
  memcpy(h->rhs[0], rhs, bytes);
  double du = DBL_MAX, m[2];
  int converged = 0, ierr = 0, nc = 0;
  for (int i = 0; i < nmax; ++i) {
    v_cycle(h);
    orc_update_u(h->nsize[0], h->u[0], u, m); /* :122 u keeps the previous iterate */
    du = h->du_max ? m[0] : m[1];
    if (hist && nc < hist_len) hist[nc] = du;
    nc++;
    if (du < vc_tol) { /* strict (:136) */
      converged = 1;
      break;
    }
  }
  *du_last = du;
  if (!converged) {
    ierr = 1;
    if (!g_quiet)
      printf(" Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged\n");
  }
  memcpy(u, h->u[0], bytes); /* :153 */
  if (ncycles) *ncycles = nc;
  return ierr;
}

int orc_solve_bvp(int ndim, const int64_t *nshape, int ngrids, const double *x, const double *y,
                  const double *z, const char *bcs, int ms, double ex_tol, int du_max,
                  int nmax_exact, double vc_tol, int nmax, double *rhs, double *u, double *du_last,
                  double *hist, int hist_len, int *ncycles, int64_t *exact_sweeps) {
  mg_t h;
  setup_bvp(&h, ndim, nshape, ngrids, x, y, z, bcs, ms, ex_tol, du_max, nmax_exact);
  int ierr = solve_bvp(&h, vc_tol, nmax, u, rhs, du_last, hist, hist_len, ncycles);
  if (exact_sweeps) *exact_sweeps += h.exact_sweeps;
  mg_delete(&h);
  return ierr;
}

/* ------------------------------------------------------------------ */
/* physics driver (ndsm_vector_potential.f90)                          */
/* ------------------------------------------------------------------ */

enum { IOPT_LEN = 16, IOPT_MS = 0, IOPT_NCYCLES = 1, IOPT_FACE1 = 2, IOPT_IERR = 3,
       IOPT_FLXCRL = 4, IOPT_DEBUG = 5, IOPT_DUMAX = 6, IOPT_NMAXEX = 7 }; /* :40-48 */
enum { ROPT_VTOL = 0, ROPT_CTOL = 1, ROPT_TIM = 2 };                         /* :54-56 */

/* :81-83 (0-based component / dims) */
static const int imap_ul[6] = {1, 2, 1, 2, 1, 2};
static const int imap_cp[6] = {0, 0, 1, 1, 2, 2};
static const int imap_nc[6][2] = {{1, 2}, {1, 2}, {0, 2}, {0, 2}, {0, 1}, {0, 1}};
/* :94-116 */
static const double tvecs1[6][3] = {{0, 1, 0}, {0, 1, 0}, {1, 0, 0}, {1, 0, 0}, {1, 0, 0}, {1, 0, 0}};
static const double tvecs2[6][3] = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};
static const double nvecs[6][3] = {{1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 1}};

/* :699-743 extract_bn ; cdim 0-based, clay 0-based layer ; dir=+1 b->bc, -1 bc->b */
static void extract_bn(const int64_t *n3, int cdim, int64_t clay, double *b, double *bc, int dir) {
  int64_t lb[3] = {0, 0, 0}, ub[3] = {n3[0] - 1, n3[1] - 1, n3[2] - 1};
  lb[cdim] = ub[cdim] = clay;
  int64_t n = 0, nx = n3[0], ny = n3[1];
  for (int64_t k = lb[2]; k <= ub[2]; ++k)
    for (int64_t j = lb[1]; j <= ub[1]; ++j)
      for (int64_t i = lb[0]; i <= ub[0]; ++i) {
        if (dir == +1) bc[n] = b[IDX3(i, j, k)];
        if (dir == -1) b[IDX3(i, j, k)] = bc[n];
        n++;
      }
}

/* :1070-1106 trapz_2D */
static double trapz_2d(int64_t n1, int64_t n2, double dq1, double dq2, const double *f) {
  double s = 0;
  for (int64_t j = 0; j < n2; ++j)
    for (int64_t i = 0; i < n1; ++i) {
      double w = 1;
      int ei = (i == 0 || i == n1 - 1), ej = (j == 0 || j == n2 - 1);
      if (ei || ej) w = 0.5;
      if (ei && ej) w = 0.25;
      s = s + w * f[i + n1 * j];
    }
  return s * dq1 * dq2;
}

static void cross3(const double *a, const double *b, double *c) { /* :1051-1066 */
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* :977-1031 compute_At_bcs */
static void compute_at_bcs(const int64_t *ns, const double *chi, double dq, const double *t1,
                           const double *t2, const double *nv, double *At1, double *At2) {
  double fac = 1.0 / (2.0 * dq);
  int64_t nq1 = ns[0], nq2 = ns[1];
  double c1[3], c2[3];
  cross3(t1, nv, c1);
  cross3(t2, nv, c2);
  for (int64_t j = 0; j < nq2; ++j)
    for (int64_t i = 0; i < nq1; ++i) {
      double d1 = (i == 0 || i == nq1 - 1) ? 0 : fac * (chi[(i + 1) + nq1 * j] - chi[(i - 1) + nq1 * j]);
      double d2 = (j == 0 || j == nq2 - 1) ? 0 : fac * (chi[i + nq1 * (j + 1)] - chi[i + nq1 * (j - 1)]);
      double g[3];
      for (int c = 0; c < 3; ++c) g[c] = d1 * c1[c] + d2 * c2[c];
      At1[i + nq1 * j] = -(t1[0] * g[0] + t1[1] * g[1] + t1[2] * g[2]);
      At2[i + nq1 * j] = -(t2[0] * g[0] + t2[1] * g[1] + t2[2] * g[2]);
    }
}

/* :825-872 derivq */
static double derivq(const int64_t *n3, const int64_t *iv, const double *dq, int dir,
                     const double *u) {
  int64_t strides[3] = {1, n3[0], n3[0] * n3[1]};
  int64_t n = iv[0] + n3[0] * (iv[1] + n3[1] * iv[2]);
  int64_t st[3];
  double wc[3];
  int nc;
  const double inv2 = 0.5;
  if (iv[dir] == 0) {
    st[0] = 0; st[1] = strides[dir]; st[2] = 2 * strides[dir];
    wc[0] = -3 * inv2 / dq[dir]; wc[1] = +4 * inv2 / dq[dir]; wc[2] = -1 * inv2 / dq[dir];
    nc = 3;
  } else if (iv[dir] == n3[dir] - 1) {
    st[0] = 0; st[1] = -strides[dir]; st[2] = -2 * strides[dir];
    wc[0] = +3 * inv2 / dq[dir]; wc[1] = -4 * inv2 / dq[dir]; wc[2] = +1 * inv2 / dq[dir];
    nc = 3;
  } else {
    st[0] = -strides[dir]; st[1] = +strides[dir];
    wc[0] = -1 * inv2 / dq[dir]; wc[1] = +1 * inv2 / dq[dir];
    nc = 2;
  }
  double s = 0;
  for (int i = 0; i < nc; ++i) s = s + u[n + st[i]] * wc[i];
  return s;
}

/* :759-811 curl */
static void curl(const int64_t *n3, const double *dq, const double *A, double *B) {
  int64_t N = n3[0] * n3[1] * n3[2], nx = n3[0], ny = n3[1];
  const double *Ax = A, *Ay = A + N, *Az = A + 2 * N;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < n3[2]; ++k)
    for (int64_t j = 0; j < n3[1]; ++j)
      for (int64_t i = 0; i < n3[0]; ++i) {
        int64_t iv[3] = {i, j, k};
        double dAx_dy = derivq(n3, iv, dq, 1, Ax), dAx_dz = derivq(n3, iv, dq, 2, Ax);
        double dAy_dx = derivq(n3, iv, dq, 0, Ay), dAy_dz = derivq(n3, iv, dq, 2, Ay);
        double dAz_dx = derivq(n3, iv, dq, 0, Az), dAz_dy = derivq(n3, iv, dq, 1, Az);
        B[IDX3(i, j, k)] = dAz_dy - dAy_dz;
        B[IDX3(i, j, k) + N] = dAx_dz - dAz_dx;
        B[IDX3(i, j, k) + 2 * N] = dAy_dx - dAx_dy;
      }
}

/* :880-950 add_flux_balance_fields */
static void add_flux_balance_fields(const int64_t *n3, const double *const *mesh,
                                    const double *phi, double *B, double *A) {
  double Lq[3];
  for (int d = 0; d < 3; ++d) Lq[d] = vmax(mesh[d], n3[d]) - vmin(mesh[d], n3[d]);
  double Vq = Lq[0] * Lq[1] * Lq[2];
  const double *x = mesh[0], *y = mesh[1], *z = mesh[2];
  double g[3] = {(phi[1] - phi[0]) / Vq, (phi[3] - phi[2]) / Vq, (phi[5] - phi[4]) / Vq};
  const double inv3 = 1.0 / 3.0;
  int64_t N = n3[0] * n3[1] * n3[2], nx = n3[0], ny = n3[1];
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < n3[2]; ++k)
    for (int64_t j = 0; j < n3[1]; ++j)
      for (int64_t i = 0; i < n3[0]; ++i) {
        double bc[3] = {g[0] * x[i] + phi[0] * Lq[0] / Vq, g[1] * y[j] + phi[2] * Lq[1] / Vq,
                        g[2] * z[k] + phi[4] * Lq[2] / Vq};
        double A1[3] = {-g[2] * y[j] * z[k], 0.0, +g[0] * x[i] * y[j]};
        double A2[3] = {+g[1] * z[k] * y[j], -g[0] * x[i] * z[k], 0.0};
        double A3[3] = {0.0, +g[2] * x[i] * z[k], -g[1] * x[i] * y[j]};
        double Ac[3] = {-(phi[4] * Lq[2] * y[j] / Vq), -(phi[0] * Lq[0] * z[k] / Vq),
                        -(phi[2] * Lq[1] * x[i] / Vq)};
        for (int c = 0; c < 3; ++c) {
          int64_t q = IDX3(i, j, k) + c * N;
          B[q] = B[q] + bc[c];
          A[q] = A[q] + Ac[c] + inv3 * (A1[c] + A2[c] + A3[c]);
        }
      }
}

/* :598-691 solve : the three 3-D Laplace problems */
static void solve3(const double *ropt, const int64_t *iopt, const double *const *mesh,
                   const int64_t *n3, double *(*At)[2], double *Ac, int64_t *exact_sweeps) {
  int64_t N = n3[0] * n3[1] * n3[2];
  int ngrids = orc_ngrids(3, n3);
  int du_max = iopt[IOPT_DUMAX] == 1;
  double *rhs = calloc((size_t)N, sizeof(double));
  double du_last;
  /* faces (0-based id): 0,1 = x lower/upper ; 2,3 = y ; 4,5 = z */
  struct { int face, slot; } src[3][4] = {
      {{2, 0}, {3, 0}, {4, 0}, {5, 0}},  /* Ax :647-650 */
      {{0, 0}, {1, 0}, {4, 1}, {5, 1}},  /* Ay :663-666 */
      {{0, 1}, {1, 1}, {2, 1}, {3, 1}}}; /* Az :679-682 */
  const char *bcs[3] = {"NDDNDD", "DNDDND", "DDNDDN"}; /* :655,:671,:687 */
  for (int c = 0; c < 3; ++c) {
    double *u = Ac + c * N;
    for (int s = 0; s < 4; ++s) {
      int f = src[c][s].face, cd = imap_cp[f];
      int64_t lay = imap_ul[f] == 1 ? 0 : n3[cd] - 1;
      extract_bn(n3, cd, lay, u, At[f][src[c][s].slot], -1);
    }
    mg_t h;
    setup_bvp(&h, 3, n3, ngrids, mesh[0], mesh[1], mesh[2], bcs[c],
              c == 2 ? 5 : (int)iopt[IOPT_MS] /* :685 Az uses ms = 5 */, ropt[ROPT_CTOL], du_max,
              (int)iopt[IOPT_NMAXEX]);
    solve_bvp(&h, ropt[ROPT_VTOL], (int)iopt[IOPT_NCYCLES], u, rhs, &du_last, NULL, 0, NULL);
    if (exact_sweeps) *exact_sweeps += h.exact_sweeps;
    mg_delete(&h);
  }
  free(rhs);
}

/* :130-497 compute_vector_potential (default branches; FACE1 is unreachable, :414) */
static void compute_vector_potential(const int64_t *nshape4, int64_t *iopt, double *ropt,
                                     const double *const *mesh, double *Apot, double *B) {
  const int64_t *n3 = nshape4;
  int64_t N = n3[0] * n3[1] * n3[2];
  int du_max = iopt[IOPT_DUMAX] == 1;
  double Lq[3], dq[3];
  for (int d = 0; d < 3; ++d) Lq[d] = vmax(mesh[d], n3[d]) - vmin(mesh[d], n3[d]);
  for (int d = 0; d < 3; ++d) {
    if (n3[d] < 2) { /* :213-216 */
      iopt[IOPT_IERR] = 1;
      return;
    }
    dq[d] = mesh[d][1] - mesh[d][0];
  }
  int64_t ns[6][2], nsz[6];
  double *bn[6], *chi[6], *At[6][2];
  const double *mesh_bn[6][2];
  for (int f = 0; f < 6; ++f) {
    for (int t = 0; t < 2; ++t) {
      ns[f][t] = n3[imap_nc[f][t]];
      mesh_bn[f][t] = mesh[imap_nc[f][t]];
    }
    nsz[f] = ns[f][0] * ns[f][1];
    bn[f] = malloc(sizeof(double) * (size_t)nsz[f]);
    chi[f] = calloc((size_t)nsz[f], sizeof(double));
    At[f][0] = calloc((size_t)nsz[f], sizeof(double));
    At[f][1] = calloc((size_t)nsz[f], sizeof(double));
  }
  /* :283-293 */
  for (int f = 0; f < 6; ++f) {
    int cd = imap_cp[f];
    int64_t lay = imap_ul[f] == 1 ? 0 : n3[cd] - 1;
    extract_bn(n3, cd, lay, B + cd * N, bn[f], +1);
  }
  /* :300-306 fluxes - always dq(1), dq(2) (quirk Q4) */
  double phi[6];
  for (int f = 0; f < 6; ++f) phi[f] = trapz_2d(ns[f][0], ns[f][1], dq[0], dq[1], bn[f]);
  double Aq[6] = {Lq[1] * Lq[2], Lq[1] * Lq[2], Lq[0] * Lq[2], Lq[0] * Lq[2], Lq[0] * Lq[1], Lq[0] * Lq[1]};
  /* :338-365 six 2-D all-Neumann solves */
  int64_t ierr = 0;
  for (int f = 0; f < 6; ++f) {
    int ngrids = orc_ngrids(2, ns[f]);
    double sub = phi[f] / Aq[f];
    for (int64_t q = 0; q < nsz[f]; ++q) bn[f][q] = bn[f][q] - sub;
    mg_t h;
    /* copt(1:6) = "N" (:357): the first 2*ndim = 4 are read */
    setup_bvp(&h, 2, ns[f], ngrids, mesh_bn[f][0], mesh_bn[f][1], NULL, "NNNN", (int)iopt[IOPT_MS],
              ropt[ROPT_CTOL], du_max, (int)iopt[IOPT_NMAXEX]);
    double du_last;
    ierr = solve_bvp(&h, ropt[ROPT_VTOL], (int)iopt[IOPT_NCYCLES], chi[f], bn[f], &du_last, NULL, 0, NULL);
    mg_delete(&h);
  }
  /* :387-399 A_t = -grad(chi) x n, with the NORMAL spacing dq(imap_cp) (quirk Q4) */
  for (int f = 0; f < 6; ++f)
    compute_at_bcs(ns[f], chi[f], dq[imap_cp[f]], tvecs1[f], tvecs2[f], nvecs[f], At[f][0], At[f][1]);
  /* :440 */
  solve3(ropt, iopt, mesh, n3, At, Apot, NULL);
  /* :467-477 default order */
  add_flux_balance_fields(n3, mesh, phi, B, Apot);
  curl(n3, dq, Apot, B);
  /* :480 - `solve` keeps its own ierr, so this is the flag of the LAST 2-D face solve */
  iopt[IOPT_IERR] = ierr;
  for (int f = 0; f < 6; ++f) {
    free(bn[f]);
    free(chi[f]);
    free(At[f][0]);
    free(At[f][1]);
  }
}

/* ndsm_python_wrapper.f90:56-158 */
int orc_vector_solve(size_t nsize, const int *nshape4, int *ioptc, double *ropt, const double *x,
                     const double *y, const double *z, double *A, double *B) {
  (void)nsize;
  int64_t n4[4], iopt[IOPT_LEN];
  for (int i = 0; i < 4; ++i) n4[i] = nshape4[i];
  for (int i = 0; i < IOPT_LEN; ++i) iopt[i] = ioptc[i];
  const double *mesh[3] = {x, y, z};
  compute_vector_potential(n4, iopt, ropt, mesh, A, B);
  ropt[ROPT_TIM] = 0.0;
  for (int i = 0; i < IOPT_LEN; ++i) ioptc[i] = (int)iopt[i];
  return (int)iopt[IOPT_IERR];
}
