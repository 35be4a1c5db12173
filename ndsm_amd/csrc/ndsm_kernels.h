/* Device layer of libndsm_hip: runtime + HIP kernel launchers (gfx950).
 *
 * This is the *internal* C interface between the Fortran 2003 host driver
 * (ndsm_amd/fsrc, bound through ISO_C_BINDING in ndsmh_iface.f90) and the
 * hand-written HIP kernels.  The drop-in boundary a user binds against is
 * include/ndsm_hip.h, not this file.
 *
 * Conventions
 *   - every function returns 0 on success, a non-zero HIP / NDSMK_E* code on
 *     failure; ndsmk_last_error() gives the text.  Nothing falls back to the CPU.
 *   - all kernels are enqueued on ONE library-owned HIP stream and return
 *     immediately unless stated "blocking".
 *   - arrays are Fortran order (x fastest); pointers are device pointers
 *     unless prefixed h_.
 *   - a level's geometry travels in ndsmk_grid (filled by the host driver with
 *     the reference's arithmetic, ndsm_optimized.f90:68-94).
 */
#ifndef NDSM_KERNELS_H
#define NDSM_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  NDSMK_OK = 0,
  NDSMK_ENODEV = 9001,   /* no HIP device visible */
  NDSMK_EARG = 9002,     /* shape/argument check failed on the host */
  NDSMK_ENCCL = 9003     /* RCCL call failed */
};

/* geometry + operator constants of one grid level (interoperable with the
 * Fortran TYPE, BIND(C) :: ndsmk_grid in ndsmh_iface.f90) */
typedef struct {
  int32_t ndim;          /* 2 or 3 */
  int32_t n[3];          /* nx, ny, nz (nz = 1 in 2-D) */
  int32_t lb[3], ub[3];  /* 0-based inclusive update bounds: shrink by one on 'D' faces */
  int32_t first_par;     /* colour of the first half-sweep: (i+j+k)&1 == first_par */
  int32_t all_neumann;   /* 1: subtract the mean after each sweep */
  int32_t k0;            /* global z index of local plane 0 (z-slab runs; may be negative), else 0 */
  int32_t nzg;           /* global nz (== n[2] unless z-slab) */
  int32_t zown0, zown1;  /* local planes [zown0, zown1) are owned (written); the rest are ghosts. 0, n[2] unless z-slab */
  double w[3];           /* 1/h^2 per dimension */
  double w1;             /* 1 / (2 (wx+wy+wz)) */
  double wc;             /* 2 (wx+wy+wz) */
} ndsmk_grid;

/* one inter-level transfer: 1-D tables per dimension, built on the host with
 * the reference's arithmetic (ndsm_interp.f90:373-435, :141-142, :229-281) */
typedef struct {
  int32_t nf[3], nc[3];   /* fine / coarse shapes */
  int32_t maxt[3];        /* taps reserved per coarse index and dim (row length of rw) */
  /* prolongation: per fine index i of dim d */
  const int32_t *plo[3];  /* lower bracket index into the coarse dim (0-based) */
  const double *pwl[3];   /* weight of the UPPER bracket point  wl = (q-ql)/dq */
  const double *pwh[3];   /* weight of the LOWER bracket point  wh = -(q-qh)/dq */
  /* restriction: per coarse index I of dim d */
  const int32_t *rlo[3];  /* first contributing fine index (0-based) */
  const int32_t *rcnt[3]; /* number of contributing fine points */
  const double *rw[3];    /* rw[d][I*maxt[d] + t] = c2 = |h_c - |q_f - q_c||  (ndsm_interp.f90:279-280) */
  double w2[3];           /* h_f / h_c^2 per dimension (ndsm_interp.f90:229) */
  /* z-slab windows (all zero / full range when not distributed).  The z tables are
   * always indexed with GLOBAL plane numbers; the arrays may hold a window. */
  int32_t f_k0;           /* global index of plane 0 of the fine array */
  int32_t f_beg, f_cnt;   /* local fine planes [f_beg, f_beg+f_cnt) are written by prolong_add */
  int32_t c_k0;           /* global index of plane 0 of the coarse array */
  int32_t c_beg, c_cnt;   /* local coarse planes [c_beg, c_beg+c_cnt) are written by restrict */
  int32_t stream_ok;      /* host-checked, restrict_stream.hip: bit 1 its tile covers this pair's taps,
                             bit 0 and the level is large enough for it to be the default; bit 2 + bits 8-31:
                             what the scheduled form needs to know about the z windows (launch_rs_t) */
} ndsmk_xfer;

/* ---- runtime ------------------------------------------------------- */
int ndsmk_device_count(void);
int ndsmk_init(int device);                 /* device < 0: LOCAL_RANK % count (else 0) */
int ndsmk_shutdown(void);
const char *ndsmk_last_error(void);
int ndsmk_note_error(int code, const char *what);   /* records the text, returns code */
int ndsmk_device_name(char *buf, int len);
void *ndsmk_stream(void);                   /* the library's hipStream_t */

int ndsmk_alloc(void **p, size_t bytes);
int ndsmk_free(void *p);
int ndsmk_h2d(void *dst, const void *h_src, size_t bytes);   /* blocking */
int ndsmk_d2h(void *h_dst, const void *src, size_t bytes);   /* blocking */
int ndsmk_d2d(void *dst, const void *src, size_t bytes);
// ghost planes of a smoothing pass: black points + perimeter only (halo.hip)
long long ndsmk_halo_packed_plane(int nx, int ny);
int ndsmk_halo_pack(const double *src, double *buf, int nx, int ny, int depth, int kg0, int first_par);
int ndsmk_halo_unpack(double *dst, const double *buf, int nx, int ny, int depth, int kg0, int first_par);
int ndsmk_halo_copy(double *dst, const double *src, int nx, int ny, int depth, int kg0, int first_par);
int ndsmk_fill0(void *p, size_t bytes);
int ndsmk_sync(void);                                        /* blocking */
int ndsmk_timer_start(void);                /* hipEventRecord on the library stream */
int ndsmk_timer_stop(double *ms);           /* blocking; elapsed between start and now */

/* background transfers: one worker thread + copy stream move the caller's host arrays while the main
 * thread drives the solves.  upload_unless_zero: scan the host array; copy it to d_dst unless every byte
 * is zero (*flag of ndsmk_bg_wait = 1: nothing copied, the caller zero-fills).  download: starts once the
 * main stream has passed the point of the call.  Tickets die in ndsmk_bg_drain (blocking, first error). */
int ndsmk_bg_upload_unless_zero(const void *h_src, void *d_dst, size_t bytes, int *ticket);
int ndsmk_bg_download(void *h_dst, const void *d_src, size_t bytes, int *ticket);
int ndsmk_bg_first_touch(void *h_dst, size_t bytes, int *ticket);   /* h_dst will be overwritten completely */
int ndsmk_bg_wait(int ticket, int *flag);
int ndsmk_bg_drain(void);
int ndsmk_host_alloc(void **p, size_t bytes);               /* pinned */
int ndsmk_host_free(void *p);
int ndsmk_mem_info(size_t *free_bytes, size_t *total_bytes);
int ndsmk_h2d_async(void *dst, const void *h_pinned_src, size_t bytes);
void ndsmk_at_reset(void (*fn)(void));                      /* fn runs at the next shutdown / re-target */
void ndsmk_on_low_memory(void (*fn)(void));                 /* fn runs when a device allocation fails, before one retry */

/* ---- kernels ------------------------------------------------------- */
/* nsweeps full red-black Gauss-Seidel sweeps (ndsm_optimized.f90:40-191 in
 * 3-D, ndsm_poisson.f90:451-549 in 2-D incl. the all-Neumann mean shift).
 * variant: 0 = pick the fastest valid kernel, 1 = two-pass colour kernels (in
 * place), 2 = fused z-streaming kernel (3-D only, out of place).
 * rhs == NULL means rhs is identically zero (never dereferenced; same bits).
 * ualt: a second array of the level's size (may be NULL: colour kernels only).
 * The fused kernel ping-pongs u <-> ualt once per sweep; on return
 * *result_in_alt = 1 means the swept field is in ualt and the caller must swap
 * its two pointers (result_in_alt == NULL: it is copied back into u). */
int ndsmk_relax(const ndsmk_grid *g, double *u, double *ualt, const double *rhs, int nsweeps, int variant,
                int *result_in_alt);
/* nsweeps sweeps, then r = rhs - L u of the result (fused into the last sweep's launch where possible) */
int ndsmk_relax_residual(const ndsmk_grid *g, double *u, double *ualt, const double *rhs, double *r, int nsweeps,
                         int variant, int *result_in_alt);
/* r = rhs - L u, zero on Dirichlet faces (ndsm_optimized.f90:346-447 / ndsm_poisson.f90:280-353) */
int ndsmk_residual(const ndsmk_grid *g, const double *u, const double *rhs, double *r);
/* rhs_c = R r_f ; also u_c = 0 if u_c != NULL (ndsm_multigrid_core.f90:551,557-558) */
int ndsmk_restrict(const ndsmk_xfer *x, const double *r_f, double *rhs_c, double *u_c);
/* footprint of the LDS-streamed restriction, for the host-side coverage check */
void ndsmk_restrict_stream_tile(int *ci, int *cj, int *fx, int *fy, int *maxt);
/* u_f += P u_c (ndsm_multigrid_core.f90:659,672) */
int ndsmk_prolong_add(const ndsmk_xfer *x, const double *u_c, double *u_f);
/* blocking: out[0] = max|a-b|, out[1] = sum|a-b| ; then b <- a if copy != 0
 * (update_u, ndsm_multigrid_core.f90:1077-1122 ; du_metrics :808-853) */
int ndsmk_diff_metrics(const double *a, double *b, int64_t n, int copy, double *h_out2);
/* its two halves: enqueue on the selected stream / lane; wait for that stream and read the pair */
int ndsmk_diff_metrics_begin(const double *a, double *b, int64_t n, int copy);
int ndsmk_diff_metrics_end(double *h_out2);
/* coarsest-grid "exact" solve (ndsm_multigrid_core.f90:728-800): repeat
 * {test du <= ex_tol first; relax; du = max|.| or mean|.| of the change} at most
 * nmax times.  d_info (DEVICE, 2 x int64) accumulates [0] sweeps done and
 * [1] the number of solves that hit nmax unconverged; the call does not block
 * for grids of <= 2048 points.  scratch: n doubles (device). */
int ndsmk_solve_exact(const ndsmk_grid *g, double *u, const double *rhs, double *scratch,
                      double ex_tol, int use_max, int nmax, int64_t *d_info);
int ndsmk_solve_exact_on_device(const ndsmk_grid *g);   /* 1: that solve is one launch, nothing comes back to the host */

/* The bottom of a V-cycle as ONE single-workgroup launch (tail.hip): levels g[0..nlev-1] (g[nlev-1] the
 * coarsest grid, x[q] the transfer g[q] -> g[q+1], u[q] / rhs[q] their DEVICE arrays), all resident in LDS:
 * ms sweeps + residual + restriction down, solve_exact + ms sweeps, interpolation + 2 ms sweeps up.
 * ndsmk_tail_applies: 1 if these levels are covered (3-D, whole levels, <= 6144 points, <= 4 levels).
 * Bit-identical to the level-by-level kernels.  ndsmk_debug_tail(0 / 1): tests switch it off / on. */
int ndsmk_tail_applies(int nlev, const ndsmk_grid *g, const ndsmk_xfer *x);
int ndsmk_tail_cycle(int nlev, const ndsmk_grid *g, const ndsmk_xfer *x, double *const *u, double *const *rhs, int ms,
                     double ex_tol, int use_max, int nmax, int64_t *d_info);
int ndsmk_debug_tail(int on);

/* A += flux-balance fields, B = curl A (+ linear field when curl_first), all on
 * device arrays (nx,ny,nz,3); x,y,z are DEVICE mesh vectors, h_* host scalars
 * (ndsm_vector_potential.f90:453-477, :759-872, :880-950) */
int ndsmk_balance_curl(double *A, double *B, const int32_t *n3, const double *x, const double *y,
                       const double *z, const double *h_phi6, const double *h_span3,
                       const double *h_dq3, int curl_first);
/* the same on a z-slab: A (nx,ny,na,3) holds global planes [kg0, kg0+na) - one ghost plane per
 * neighbour -, B (nx,ny,nb,3) starts at A's local plane boff; n3 and x,y,z stay GLOBAL */
int ndsmk_balance_curl_slab(double *A, double *B, const int32_t *n3, int kg0, int na, int nb, int boff,
                            const double *x, const double *y, const double *z, const double *h_phi6,
                            const double *h_span3, const double *h_dq3, int curl_first);

/* one component (c = 0,1,2) of the flux-balance fields added to that component of A alone (same
 * expressions as ndsmk_balance_curl's first kernel: components can be finished one at a time), and the
 * curl alone: B = curl A on device arrays (nx,ny,nz,3) */
int ndsmk_balance_component(double *Ac, const int32_t *n3, int c, const double *x, const double *y, const double *z,
                            const double *h_phi6, const double *h_span3);
int ndsmk_curl(const double *A, double *B, const int32_t *n3, const double *h_dq3);
/* component c of B alone (needs the other two components of A only) */
int ndsmk_curl_component(const double *A, double *B, const int32_t *n3, const double *h_dq3, int c);

/* the face phase on the device (faces.hip): packed face buffers, six faces back to back */
int ndsmk_face_offsets(const int32_t *n3, int64_t *off6, int64_t *total);
int ndsmk_face_extract(const double *B, const int32_t *n3, double *faces);
int ndsmk_face_flux(const double *faces, const int32_t *n3, double h1h2, double *d_phi6);
int ndsmk_face_rhs(const double *faces, const int32_t *n3, int f, const double *d_phi6, double area, double *rhs);
int ndsmk_face_write(double *u, const int32_t *n3, const double *chi, int f, int c, double fac);
int ndsmk_face_put(double *u, const int32_t *n3, int f, const double *vals);

/* level-1 form for the V-cycle driver: three buffers (u on entry + two spares), `keep` (one of them
 * or NULL) is never written, *where = 0/1/2 names the buffer holding the result.  r: residual of the
 * result (may be NULL).  prev: have the launch of the last sweep evaluate max / sum |u_new - prev|
 * (*met_done = 1 if it did; ndsmk_fetch_fused_metric reads the pair, blocking). */
int ndsmk_relax3(const ndsmk_grid *g, double *u, double *a, double *b, const double *keep, const double *rhs,
                 int nsweeps, double *r, const double *prev, int *where, int *met_done,
                 const ndsmk_xfer *px /* != NULL: u += P uc first */, const double *uc);
int ndsmk_fetch_fused_metric(double *h_out2);

/* pieces of an overlapped z-slab pass: one fused pass over owned planes [z0, z1) only (u -> uout) */
int ndsmk_fused_window(const ndsmk_grid *g, const double *u, double *uout, const double *rhs, int nsweeps, int z0,
                       int z1, const ndsmk_xfer *px, const double *uc, const double *prev, int accumulate);
int ndsmk_fused_window_res(const ndsmk_grid *g, const double *u, double *uout, const double *rhs, double *rout, int z0,
                           int z1);
int ndsmk_fused_metric_ok(const ndsmk_grid *g);
int ndsmk_fused_prolong_ok(const ndsmk_grid *g, const double *rhs, int nsweeps);
/* two streams: 1 = later copies / RCCL calls go to the communication stream, 0 = main stream again;
 * fence(from, to): work enqueued on `from` so far precedes work enqueued on `to` from now on */
#define NDSMK_LANES 6
int ndsmk_select_lane(int lane);           /* -1: main stream */
int ndsmk_lane_fence(int lane, int to_main);
int ndsmk_lane_idle(int lane);   /* 1: the lane's stream has drained, 0: not yet (never blocks), < 0: error */
int ndsmk_lane_sync(int lane);
int ndsmk_capture_begin(void);             /* record what is enqueued on the selected lane ... */
int ndsmk_capture_end(void **exec);        /* ... into an executable graph */
int ndsmk_graph_launch(void *exec);        /* replay it on the selected lane / stream */
int ndsmk_graph_destroy(void *exec);
int ndsmk_select_stream(int which);
int ndsmk_stream_fence(int from, int to);

/* ---- mixed-precision mode (mixed.hip): level 1 as iterative refinement, correction in fp32 ---- */
/* unew = u + e ; ezero = 0 ; r = (float)(rhs - L unew) ; h_out2 = (max|e|, sum|e|), blocking.
 * e == NULL: r = residual of u, nothing else written.  rhs == NULL: zero right-hand side. */
int ndsmk_update_residual_f32(const ndsmk_grid *g, const double *u, double *unew, const double *rhs, const float *e,
                              float *ezero, float *r, double *h_out2);
/* nsweeps fp32 sweeps of L e = r (fused kernel only); r_out != NULL: residual of the swept equation */
int ndsmk_relax_f32(const ndsmk_grid *g, float *e, float *ealt, const float *r, int nsweeps, int force, float *r_out,
                    int *result_in_alt);
int ndsmk_restrict_f32(const ndsmk_xfer *x, const float *r_f, double *rhs_c, double *u_c);
int ndsmk_prolong_add_f32(const ndsmk_xfer *x, const double *u_c, float *e_f);

/* tests / tuning: the five values of NDSM_FUSED_CFG (smooth_fused.hip) at run time */
int ndsmk_debug_fused_cfg(int two, int one, int res, int work_items, int big);

#ifdef __cplusplus
}
#endif
#endif
