/* libndsm_hip - C ABI of the MI355X-native multigrid path for NDSM.
 *
 * PART 1 is the drop-in boundary: exactly the dynamic symbols the reference's
 * shared object ndsmf.so exports (verified with `nm -D` on a build of
 * /root/reference/fortran), i.e. what its Python front-end binds through
 * ctypes (ndsm.py:136-207).  Names, argument lists, option-slot values and
 * return conventions are the reference's; each prototype cites the
 * BIND(C) procedure it replaces.
 *
 * PART 2 is additive (SURVEY.md 8b last row, 8f-4): a scalar Poisson entry, a
 * persistent device-resident solver handle, timing hooks.  Nothing in part 1
 * changes meaning because part 2 exists; all additive option slots are slots
 * the reference leaves unused (value 0 = reference behaviour).
 *
 * All arrays are Fortran order (x fastest): a numpy array of shape (3,nz,ny,nx)
 * in C order IS the (nx,ny,nz,3) array meant here (ndsm.py:161,210).
 * Everything is double precision; sizes in the reference ABI are C int / size_t.
 *
 * Errors: 0 = ok, 1 = V-cycle iteration did not reach vc_tol (reference
 * semantics), >= 9001 = device/runtime failure (text via ndsm_hip_last_error
 * and on stderr).  The library has no CPU fallback: without an MI355X every
 * solve returns 9001.  It never calls exit()/STOP (the reference does on
 * internal asserts, ndsm_root.f90:317-455).
 *
 * Threading: blocking calls, one library-owned HIP stream; not re-entrant (as
 * the reference: module-global DEBUG flag, ndsm_root.f90:64).
 */
#ifndef NDSM_HIP_H
#define NDSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* =====================================================================
 * PART 1 - the reference ABI
 * ===================================================================== */

/* Replaces ndsm_vector_solve, fortran/ndsm_python_wrapper.f90:56-158.
 *   nsize    nx*ny*nz*3                                   (by value, :64)
 *   nshape4  [nx, ny, nz, 3]                              (:65, ndsm.py:161)
 *   ioptc    16 integer options, in/out; slots from the getters below
 *   ropt     16 real options, in/out
 *   x,y,z    mesh vectors of length nx, ny, nz (uniform spacing assumed)
 *   A        in: initial guess of the three Laplace solves (ndsm.py passes
 *            zeros); out: vector potential, (nx,ny,nz,3)
 *   B        in: field whose NORMAL component on the six faces is the
 *            boundary data; out: curl A, (nx,ny,nz,3)
 * returns ioptc[IOPT_IERR]: 0 ok, 1 not converged / bad mesh (see DESIGN.md
 * quirk Q3' for which solve's flag the reference actually returns).
 * After a device / runtime failure (>= 9001) the contents of A and B are unspecified (a worker thread has
 * begun to touch and fill them behind the solves). */
int ndsm_vector_solve(size_t nsize, const int *nshape4, int *ioptc, double *ropt, const double *x,
                      const double *y, const double *z, double *A, double *B);

/* Option-slot getters, fortran/ndsm_python_wrapper.f90:164-234.  ndsm.py never
 * hard-codes a slot; it asks the library (ndsm.py:155-174). */
int get_iopt_len(void);         /* :164  -> 16 */
int get_iopt_ierr(void);        /* :170  -> 16 (sic: the reference returns IOPT_LEN, not IOPT_IERR = 3) */
int get_iopt_ms(void);          /* :176  -> 0  smoothing sweeps */
int get_iopt_ncycles(void);     /* :182  -> 1  max V-cycles */
int get_iopt_debug(void);       /* :188  -> 5 */
int get_iopt_dumax(void);       /* :194  -> 6  1: max|du| metric, 0: mean|du| */
int get_iopt_iopt_nmaxex(void); /* :200  -> 7  max sweeps of the coarsest-grid solve (the doubled "iopt" is the reference's name) */
int get_iopt_true(void);        /* :206  -> 1 */
int get_iopt_false(void);       /* :212  -> 0 */
int get_ropt_tim(void);         /* :218  -> 2  out: wall time of the call, seconds */
int get_ropt_vtol(void);        /* :224  -> 0  V-cycle tolerance */
int get_ropt_ctol(void);        /* :230  -> 1  coarsest-grid tolerance */

/* =====================================================================
 * PART 2 - additive exports
 * ===================================================================== */

/* additive option slots (unused in the reference, ndsm_vector_potential.f90:40-57) */
int get_iopt_fail3d(void);   /* -> 8  out: bit c set if 3-D solve c (0=Ax,1=Ay,2=Az) missed vc_tol */
int get_iopt_ngrids(void);   /* -> 9  in : cap on the number of grid levels, 0 = reference rule
                                          floor(log2(nmin/2)) (ndsm_vector_potential.f90:631-632) */
int get_iopt_ncyc_out(void); /* -> 10 out: V-cycles used by the last solve that iterated */
int get_ropt_dulast(void);   /* -> 3  out: du of its last V-cycle */
int get_iopt_prec(void);     /* -> 11 in : 0 fp64 throughout (reference arithmetic); 1 mixed precision for the 3-D
                                          solves: fp64 residual + fp32 correction V-cycle on level 1 (BASELINE
                                          config[4]); where level 1 is too small for the fp32 kernels the
                                          fp64 path runs */

int ndsm_hip_device_count(void);
int ndsm_hip_init(int device);               /* < 0: LOCAL_RANK % device count; idempotent */
/* Releases what the library itself holds on the device (streams, events, metric scratch, the cached
 * vector-potential hierarchy, a live RCCL communicator).  Destroy solver / world handles first.  Any
 * later call re-initialises (ndsm_hip_init may then name another device). */
int ndsm_hip_shutdown(void);
void ndsm_hip_last_error(char *buf, int len);
int ndsm_hip_sync(void);                     /* wait for the library stream */
int ndsm_hip_timer_start(void);              /* hipEventRecord on the library stream */
int ndsm_hip_timer_stop(double *ms);         /* record + synchronise + elapsed */

/* laplace(u) = rhs with 'D'/'N' faces; the scalar problem the reference only
 * reaches internally (solve_poisson_bvp, ndsm_poisson.f90:63-155).
 *   bcs     2*ndim letters: lower faces of dim 1..ndim, then upper faces
 *           (RESHAPE(copt,[ndim,2]), ndsm_poisson.f90:244-245)
 *   u       in: initial guess INCLUDING the Dirichlet face values; out: solution
 *   rhs     may be NULL (= 0);   hist may be NULL (else du per V-cycle)
 * options in the same slots as ndsm_vector_solve; returns 0 / 1 / >= 9001. */
int ndsm_hip_poisson_solve(int ndim, const int *nshape, const double *x, const double *y,
                           const double *z, const char *bcs, int *ioptc, double *ropt, double *u,
                           const double *rhs, double *hist, int hist_len);

/* ---- persistent solver: hierarchy, tables and level arrays stay in HBM ---- */
int ndsm_hip_mg_create(int ndim, const int *nshape, const double *x, const double *y,
                       const double *z, const char *bcs, int ngrids /* 0 = reference rule */, int ms,
                       double ex_tol, int du_max, int nmax_exact, void **handle);
int ndsm_hip_mg_destroy(void *handle);
int ndsm_hip_mg_levels(void *handle, int ngrids_cap, int *shapes /* [ngrids_cap][3] */); /* returns ngrids */
int ndsm_hip_mg_set_ms(void *handle, int ms);
/* mode 0 fp64, 1 mixed where level 1 is large (>= 6 M points), 2 mixed wherever the fp32 kernels cover
 * level 1.  Returns 1 if ndsm_hip_mg_solve will run in mixed precision, 0 if fp64, < 0 bad mode. */
int ndsm_hip_mg_set_precision(void *handle, int mode);
/* which: 0 = u, 1 = rhs, 2 = residual scratch (level-1 sized) */
int ndsm_hip_mg_upload(void *handle, int level, int which, const double *host);
int ndsm_hip_mg_download(void *handle, int level, int which, double *host);
/* declare rhs(1) == 0 (Laplace problem, the vector-potential case): kernels skip reading it; same bits */
int ndsm_hip_mg_zero_rhs(void *handle);
/* op: 0 relax (count sweeps), 1 residual -> scratch, 2 restrict scratch(level) -> rhs(level+1)
 * and zero u(level+1), 3 u(level) += P u(level+1), 4 coarsest-grid solve, 5/6 relax forced to
 * the two-pass / fused kernel, 8 relax (count sweeps) then residual -> scratch with the last
 * sweep and the residual in one launch where the level allows (9: or fail).  Asynchronous. */
int ndsm_hip_mg_op(void *handle, int op, int level, int count);
int ndsm_hip_mg_vcycle(void *handle, int ncycles);   /* asynchronous, no convergence test */
/* V-cycles to vc_tol: returns 0 converged, 1 not, >= 9001 error */
int ndsm_hip_mg_solve(void *handle, double vc_tol, int nmax, double *du_last, int *ncycles,
                      double *hist, int hist_len);
int ndsm_hip_mg_info(void *handle, int64_t *exact_sweeps, int64_t *unconverged_coarse_solves);

/* device memory on the library's GPU for callers without a HIP binding of their own (blocking copies on
 * the library stream); pointers from the caller's own hipMalloc work just as well */
int ndsm_hip_device_alloc(size_t bytes, void **p);
int ndsm_hip_device_free(void *p);
int ndsm_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int ndsm_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);

/* ---- persistent vector-potential solver ----
 * The reference rebuilds its grid hierarchy per component and per call (ndsm_vector_potential.f90:652-689,
 * ndsm_multigrid_core.f90:165-329).  Here everything that depends on the grid alone - the 3-D hierarchy and
 * its transfer tables, the three 2-D face hierarchies, device arrays for A, B and the six faces - lives in a
 * handle; time-series callers solve again and again on the same mesh.  (ndsm_vector_solve keeps ONE such
 * context internally, keyed by shape, mesh and level cap: a second call on the same mesh skips the set-up
 * without the caller changing anything.  ndsm_hip_shutdown drops it.)
 *   nshape4 = [nx,ny,nz,3], x,y,z as for ndsm_vector_solve; ngrids = level cap (0: reference rule) - a solve
 *   whose ioptc[get_iopt_ngrids()] differs from it fails with 9002.
 *   ndsm_hip_vecpot_solve        A, B HOST arrays: contents, options, return value as ndsm_vector_solve.
 *   ndsm_hip_vecpot_solve_device A, B DEVICE arrays (nx,ny,nz,3) on the library's GPU: B.n is extracted by a
 *                                kernel, nothing crosses PCIe but six fluxes and the 16-byte convergence
 *                                read-backs; results are complete in A, B when the call returns. */
int ndsm_hip_vecpot_create(const int nshape4[4], const double *x, const double *y, const double *z, int ngrids,
                           void **handle);
int ndsm_hip_vecpot_solve(void *handle, int ioptc[16], double ropt[16], double *A, double *B);
int ndsm_hip_vecpot_solve_device(void *handle, int ioptc[16], double ropt[16], double *dA, double *dB);
int ndsm_hip_vecpot_destroy(void *handle);

/* =====================================================================
 * PART 3 - additive exports, multi-GPU
 *
 * One process per GPU.  Level 1 is cut into z-slabs, one per rank; every slab
 * carries `g` ghost planes per side; neighbours exchange 4 planes per two-sweep
 * pass over RCCL (ncclSend / ncclRecv on the library stream); the restricted
 * residual is gathered to rank 0, which runs levels >= 2 and scatters the coarse
 * correction back.  Results are bit-identical to the single-GPU solver (tests:
 * loop-back world on one GPU, 2-rank gloo model on the CPU).
 * ===================================================================== */

/* RCCL bootstrap: rank 0 fills id128 (128 bytes), the caller broadcasts it by its own means
 * (MPI, torch.distributed/gloo, a file), every rank then calls dist_init.  Requires
 * ndsm_hip_init(local device) first. */
int ndsm_hip_dist_unique_id(void *id128);
int ndsm_hip_dist_init(int rank, int nranks, const void *id128);
/* destroys the communicator after draining the library's streams; call on every rank once all
 * worlds are destroyed.  No-op without a communicator. */
int ndsm_hip_dist_finalize(void);
/* Collective transport self-test on the live communicator (meant for the 1-rank bring-up on a one-GPU
 * box, valid at any size): each rank sends nelem doubles to itself and receives them back through the
 * grouped ncclSend / ncclRecv pair a halo exchange uses, once on the main stream and once on the
 * communication stream between two fences (the order an overlapped pass issues them in), then runs the
 * 2-value (max, sum) all-reduce of the convergence metric.  0 = every byte arrived and the reduction is right. */
int ndsm_hip_dist_selftest(int nelem);
/* rank / size as the live RCCL communicator reports them (ncclCommUserRank, ncclCommCount);
 * *nranks = 0 when no communicator is up */
int ndsm_hip_dist_info(int *rank, int *nranks);

/* Slab plan for nranks ranks, 12 ints per rank: rank, z0, z1 (owned fine planes), g (ghost
 * depth), nloc (= z1 - z0 + 2 g), k0 (global index of local plane 0), ck0, ck1 (coarse planes it
 * restricts), pk0, pk1 (coarse planes it needs for prolongation), cb0, cb1 (coarse buffer window).
 * Pure host arithmetic (no GPU needed).  Returns 0, or 9002 when the shape cannot be cut that way. */
int ndsm_hip_slab_plan(const int *nshape, const double *x, const double *y, const double *z, int ngrids,
                       int nranks, int *out /* [nranks][12] */);

/* ndsm_vector_solve on the z-slab decomposition (BASELINE config[4]: 2048 x 2048 x 1024 across the
 * GPUs of a node).  Collective: every rank of the communicator calls it with the GLOBAL nshape4 =
 * [nx,ny,nz,3] and mesh vectors and with ITS planes [z0, z1) of A and B - the split
 * ndsm_hip_slab_plan reports for (nshape, ngrids = ioptc[get_iopt_ngrids()], nranks) - laid out
 * (nx, ny, z1-z0, 3) in Fortran order.  The O(N^(2/3)) face phase (fluxes, six 2-D solves, tangential
 * data) runs on rank 0 exactly as in ndsm_vector_solve, the three 3-D solves on z-slab worlds, flux
 * balance and curl on the slabs.  Options, return value and the contents of A (in: initial guess,
 * out: vector potential) and B (in: boundary normal component, out: curl A + correction) as for
 * ndsm_vector_solve - the same bits on the same input, with ioptc[get_iopt_prec()] != 0 the bits of the
 * single-GPU mixed-precision mode (the 3-D solves then run ndsm_hip_world_set_precision's scheme where
 * the slabs allow it).  nranks == 1 is ndsm_vector_solve.  Reference: none (shared-memory OpenMP only); pipeline of
 * ndsm_vector_potential.f90:130-497. */
int ndsm_hip_world_vector_solve(int rank, int nranks, const int nshape4[4], int ioptc[16], double ropt[16],
                                const double *x, const double *y, const double *z, double *A_slab, double *B_slab);

/* rank >= 0: this process holds slab `rank` (RCCL transport, after ndsm_hip_dist_init);
 * rank <  0: loop-back world - all nranks slabs on this GPU, neighbours reached by device copies
 *            (verification of the slab algebra on one GPU). */
int ndsm_hip_world_create(const int *nshape, const double *x, const double *y, const double *z,
                          const char *bcs, int ngrids, int ms, double ex_tol, int du_max, int nmax_exact,
                          int nranks, int rank, void **handle);
int ndsm_hip_world_destroy(void *handle);
int ndsm_hip_world_nlocal(void *handle);                          /* slabs held by this process */
/* how many levels are distributed (1 = only the finest; more where a rank's share of the next level
 * is still large: its restricted planes then never travel to rank 0; NDSM_HIP_DIST_LEVELS=k forces) */
int ndsm_hip_world_dist_levels(void *handle);
int ndsm_hip_world_slab(void *handle, int ilocal, int *info12);   /* its plan row */
/* which: 0 = u, 1 = rhs, 2 = residual.  host holds nplanes whole x-y planes starting at GLOBAL
 * plane gz0; the planes that fall into slab ilocal's window (ghosts included) are copied. */
int ndsm_hip_world_upload(void *handle, int ilocal, int which, const double *host, int gz0, int nplanes);
int ndsm_hip_world_download(void *handle, int ilocal, int which, double *host /* owned planes */);
/* Mixed precision on the slabs (BASELINE config[4]): as ndsm_hip_mg_set_precision - 0 fp64, != 0 fp64
 * residual + fp32 correction V-cycle on level 1 (halo exchange, restriction and prolongation of the
 * correction in fp32, everything from level 2 down unchanged).  Returns 1 if ndsm_hip_world_solve will
 * run mixed, 0 if the fp64 path stays (a slab out of the fp32 kernels' reach), < 0 bad arguments.
 * Same bits as the single-domain mixed mode. */
int ndsm_hip_world_set_precision(void *world, int mode);
int ndsm_hip_world_zero_rhs(void *handle);                        /* as ndsm_hip_mg_zero_rhs */
int ndsm_hip_world_relax(void *handle, int nsweeps);              /* collective */
int ndsm_hip_world_vcycle(void *handle, int ncycles);             /* collective, asynchronous */
/* collective; every rank returns the same history.  0 converged, 1 not, >= 9001 error */
int ndsm_hip_world_solve(void *handle, double vc_tol, int nmax, double *du_last, int *ncycles, double *hist,
                         int hist_len);

/* "hip=<path>;rccl=<path>": the shared objects this library's HIP / RCCL calls are bound to
 * (a process that also imports PyTorch holds two ROCm stacks) */
int ndsm_hip_bound_libs(char *buf, int len);

/* =====================================================================
 * PART 4 - development hooks (tests and tuning; no reference counterpart)
 * ===================================================================== */

/* Tile configuration of the fused smoother (csrc/smooth_fused.hip), the five values of the
 * NDSM_FUSED_CFG environment variable at run time: alternates for the two-sweep / one-sweep /
 * sweep+residual launches (0 = default), a fixed number of work items (0 = the launch's own choice),
 * and big = -1 by level size / 0 never / 1 always use the tiles of >= 64 M-point levels - so that the
 * parity tests can run the benchmarked 512^3 configurations on grids the oracle handles.
 * Results never depend on these values (bit for bit); only speed does. */
int ndsm_hip_debug_fused_cfg(int two, int one, int res, int work_items, int big);
/* 0: the small levels at the bottom of a V-cycle run kernel by kernel even where the single-launch form
 * (csrc/tail.hip: all of them resident in LDS, one workgroup) covers them; 1 = default (environment
 * NDSM_HIP_NO_TAIL switches it off for a whole process).  Same bits either way. */
int ndsm_hip_debug_tail(int on);

#ifdef __cplusplus
}
#endif
#endif
