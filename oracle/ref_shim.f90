! TEST INFRASTRUCTURE - not part of the product.
!
! C-ABI shim over the *reference's own* Fortran modules, so that the test
! suite (and tests/golden/make_golden.py) can call the reference's kernels
! one at a time.  This file is ours; the modules it USEs are compiled from
! /root/reference/fortran/*.f90 where they lie (see oracle/Makefile) and both
! land in oracle/_ref/ (git-ignored, never committed).
!
! Every entry point below is a thin pass-through to a PUBLIC procedure of
! the reference:
!   refk_relax3d      -> red_black_gauss_3D      (ndsm_optimized.f90:40)
!   refk_residual3d   -> poisson_residual_3D     (ndsm_optimized.f90:346)
!   refk_relax_nd     -> relax                   (ndsm_poisson.f90:451)
!   refk_residual_nd  -> residual                (ndsm_poisson.f90:280)
!   refk_hierarchy    -> new_mg_handle           (ndsm_multigrid_core.f90:165)
!   refk_restrict     -> mg_restrict             (ndsm_multigrid_core.f90:1010)
!   refk_interp       -> mg_interp               (ndsm_multigrid_core.f90:865)
!   refk_vcycle       -> v_cycle                 (ndsm_multigrid_core.f90:341)
!   refk_solve_bvp    -> solve_poisson_bvp       (ndsm_poisson.f90:63)
!   refk_update_u     -> update_u                (ndsm_multigrid_core.f90:1077)
!
MODULE ref_shim

  USE, INTRINSIC :: ISO_C_BINDING
  USE NDSM_ROOT
  USE NDSM_MULTIGRID_CORE
  USE NDSM_POISSON
  USE NDSM_OPTIMIZED, ONLY: red_black_gauss_3D, poisson_residual_3D

  IMPLICIT NONE

CONTAINS

  ! Unpack "NDDNDD"-style C chars (lower x,y,z then upper x,y,z) into bcs(ndim,2)
  SUBROUTINE unpack_bcs(ndim, cb, bcs)
    INTEGER(IT), INTENT(IN) :: ndim
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    CHARACTER(LEN=1), DIMENSION(ndim,2), INTENT(OUT) :: bcs
    INTEGER(IT) :: d
    DO d = 1, ndim
      bcs(d,1) = cb(d)
      bcs(d,2) = cb(ndim+d)
    END DO
  END SUBROUTINE

  SUBROUTINE make_mesh(ndim, nshape, x, y, z, mesh)
    INTEGER(IT), INTENT(IN) :: ndim
    INTEGER(IT), DIMENSION(ndim), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    TYPE(MG_PTR), DIMENSION(ndim), INTENT(OUT) :: mesh
    ALLOCATE(mesh(1)%val(nshape(1)))
    mesh(1)%val = x(1:nshape(1))
    IF (ndim >= 2) THEN
      ALLOCATE(mesh(2)%val(nshape(2)))
      mesh(2)%val = y(1:nshape(2))
    END IF
    IF (ndim >= 3) THEN
      ALLOCATE(mesh(3)%val(nshape(3)))
      mesh(3)%val = z(1:nshape(3))
    END IF
  END SUBROUTINE

  ! -------------------------------------------------------------------

  SUBROUTINE refk_relax3d(nshape3, x, y, z, cb, rhs, u) BIND(C, NAME="refk_relax3d")
    INTEGER(C_INT64_T), DIMENSION(3), INTENT(IN) :: nshape3
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: rhs
    REAL(C_DOUBLE), DIMENSION(*), INTENT(INOUT) :: u
    CHARACTER(LEN=1), DIMENSION(3,2) :: bcs
    CALL unpack_bcs(INT(3,IT), cb, bcs)
    CALL red_black_gauss_3D(bcs, nshape3(1), nshape3(2), nshape3(3), x, y, z, rhs, u)
  END SUBROUTINE

  SUBROUTINE refk_residual3d(nshape3, x, y, z, cb, rhs, u, r) BIND(C, NAME="refk_residual3d")
    INTEGER(C_INT64_T), DIMENSION(3), INTENT(IN) :: nshape3
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: rhs, u
    REAL(C_DOUBLE), DIMENSION(*), INTENT(OUT) :: r
    CHARACTER(LEN=1), DIMENSION(3,2) :: bcs
    CALL unpack_bcs(INT(3,IT), cb, bcs)
    CALL poisson_residual_3D(bcs, nshape3(1), nshape3(2), nshape3(3), x, y, z, rhs, u, r)
  END SUBROUTINE

  SUBROUTINE refk_relax_nd(ndim_c, nshape, x, y, z, cb, rhs, u) BIND(C, NAME="refk_relax_nd")
    INTEGER(C_INT), VALUE :: ndim_c
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: rhs
    REAL(C_DOUBLE), DIMENSION(*), INTENT(INOUT) :: u
    INTEGER(IT) :: ndim, nsize
    CHARACTER(LEN=1), DIMENSION(ndim_c,2) :: bcs
    TYPE(MG_PTR), DIMENSION(ndim_c) :: mesh
    ndim = ndim_c
    nsize = PRODUCT(nshape(1:ndim))
    CALL unpack_bcs(ndim, cb, bcs)
    CALL make_mesh(ndim, nshape(1:ndim), x, y, z, mesh)
    CALL relax(ndim, nsize, nshape(1:ndim), mesh, bcs, u(1:nsize), rhs(1:nsize))
  END SUBROUTINE

  SUBROUTINE refk_residual_nd(ndim_c, nshape, x, y, z, cb, rhs, u, r) BIND(C, NAME="refk_residual_nd")
    INTEGER(C_INT), VALUE :: ndim_c
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: rhs, u
    REAL(C_DOUBLE), DIMENSION(*), INTENT(OUT) :: r
    INTEGER(IT) :: ndim, nsize
    CHARACTER(LEN=1), DIMENSION(ndim_c,2) :: bcs
    TYPE(MG_PTR), DIMENSION(ndim_c) :: mesh
    ndim = ndim_c
    nsize = PRODUCT(nshape(1:ndim))
    CALL unpack_bcs(ndim, cb, bcs)
    CALL make_mesh(ndim, nshape(1:ndim), x, y, z, mesh)
    CALL residual(ndim, nsize, nshape(1:ndim), mesh, bcs, u(1:nsize), rhs(1:nsize), r(1:nsize))
  END SUBROUTINE

  ! Level shapes (ndim*ngrids, level-major) and level meshes, concatenated
  ! level by level, dimension by dimension, into mesh_out.
  SUBROUTINE refk_hierarchy(ndim_c, nshape, ngrids_c, x, y, z, shapes_out, mesh_out) &
      BIND(C, NAME="refk_hierarchy")
    INTEGER(C_INT), VALUE :: ndim_c, ngrids_c
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(OUT) :: shapes_out
    REAL(C_DOUBLE), DIMENSION(*), INTENT(OUT) :: mesh_out
    TYPE(MG_HANDLE) :: h
    TYPE(MG_PTR), DIMENSION(ndim_c) :: mesh
    INTEGER(IT) :: ndim, ngrids, l, d, p, n
    ndim = ndim_c; ngrids = ngrids_c
    CALL make_mesh(ndim, nshape(1:ndim), x, y, z, mesh)
    CALL new_mg_handle(h, ndim, nshape(1:ndim), ngrids, mesh, .TRUE., INT(1,IT))
    p = 0
    DO l = 1, ngrids
      DO d = 1, ndim
        n = h%nshape(d,l)
        shapes_out((l-1)*ndim + d) = n
        mesh_out(p+1:p+n) = h%meshes(d,l)%val(1:n)
        p = p + n
      END DO
    END DO
    CALL delete_mg_handle(h)
  END SUBROUTINE

  SUBROUTINE refk_restrict(ndim_c, nshape, ngrids_c, x, y, z, id_f, u_f, u_c) BIND(C, NAME="refk_restrict")
    INTEGER(C_INT), VALUE :: ndim_c, ngrids_c, id_f
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z, u_f
    REAL(C_DOUBLE), DIMENSION(*), INTENT(OUT) :: u_c
    TYPE(MG_HANDLE) :: h
    TYPE(MG_PTR), DIMENSION(ndim_c) :: mesh
    INTEGER(IT) :: ndim, ngrids
    ndim = ndim_c; ngrids = ngrids_c
    CALL make_mesh(ndim, nshape(1:ndim), x, y, z, mesh)
    CALL new_mg_handle(h, ndim, nshape(1:ndim), ngrids, mesh, .TRUE., INT(1,IT))
    CALL mg_restrict(h, INT(id_f,IT), u_f, INT(id_f+1,IT), u_c)
    CALL delete_mg_handle(h)
  END SUBROUTINE

  SUBROUTINE refk_interp(ndim_c, nshape, ngrids_c, x, y, z, id_f, u_c, u_f) BIND(C, NAME="refk_interp")
    INTEGER(C_INT), VALUE :: ndim_c, ngrids_c, id_f
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z, u_c
    REAL(C_DOUBLE), DIMENSION(*), INTENT(OUT) :: u_f
    TYPE(MG_HANDLE) :: h
    TYPE(MG_PTR), DIMENSION(ndim_c) :: mesh
    INTEGER(IT) :: ndim, ngrids
    ndim = ndim_c; ngrids = ngrids_c
    CALL make_mesh(ndim, nshape(1:ndim), x, y, z, mesh)
    CALL new_mg_handle(h, ndim, nshape(1:ndim), ngrids, mesh, .TRUE., INT(1,IT))
    CALL mg_interp(h, INT(id_f,IT), u_f, INT(id_f+1,IT), u_c)
    CALL delete_mg_handle(h)
  END SUBROUTINE

  SUBROUTINE setup_bvp(h, ndim, nshape, ngrids, x, y, z, cb, ms, ex_tol, du_max, nmax_exact)
    TYPE(MG_HANDLE), INTENT(OUT) :: h
    INTEGER(IT), INTENT(IN) :: ndim, ngrids, ms, nmax_exact
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), INTENT(IN) :: ex_tol
    LOGICAL, INTENT(IN) :: du_max
    TYPE(MG_PTR), DIMENSION(ndim) :: mesh
    INTEGER(IT) :: d
    CALL make_mesh(ndim, nshape(1:ndim), x, y, z, mesh)
    CALL new_mg_handle(h, ndim, nshape(1:ndim), ngrids, mesh, du_max, nmax_exact)
    h%ms = ms
    h%ex_tol = ex_tol
    ! copt(1:2*ndim) = lower faces then upper faces, as ndsm_vector_potential.f90:655
    DO d = 1, 2*ndim
      h%copt(d) = cb(d)
    END DO
  END SUBROUTINE

  ! One V-cycle on (u, rhs), in place.
  SUBROUTINE refk_vcycle(ndim_c, nshape, ngrids_c, x, y, z, cb, ms, ex_tol, du_max_c, nmax_exact, rhs, u) &
      BIND(C, NAME="refk_vcycle")
    INTEGER(C_INT), VALUE :: ndim_c, ngrids_c, ms, du_max_c, nmax_exact
    REAL(C_DOUBLE), VALUE :: ex_tol
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z, rhs
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), DIMENSION(*), INTENT(INOUT) :: u
    TYPE(MG_HANDLE) :: h
    INTEGER(IT) :: ndim, nsize
    ndim = ndim_c
    nsize = PRODUCT(nshape(1:ndim))
    CALL setup_bvp(h, ndim, nshape, INT(ngrids_c,IT), x, y, z, cb, INT(ms,IT), ex_tol, du_max_c == 1, &
                   INT(nmax_exact,IT))
    ALLOCATE(h%u(1)%val(nsize), h%rhs(1)%val(nsize))
    h%u(1)%val = u(1:nsize)
    h%rhs(1)%val = rhs(1:nsize)
    CALL v_cycle(h, INT(1,IT), ndsm_relax_wrapper, ndsm_residual_wrapper)
    u(1:nsize) = h%u(1)%val
    CALL delete_mg_handle(h)
  END SUBROUTINE

  ! Full solve.  out(1) = du_last, returns ierr.
  FUNCTION refk_solve_bvp(ndim_c, nshape, ngrids_c, x, y, z, cb, ms, ex_tol, du_max_c, nmax_exact, &
                          vc_tol, nmax, rhs, u, du_last) BIND(C, NAME="refk_solve_bvp") RESULT(ierr_c)
    INTEGER(C_INT), VALUE :: ndim_c, ngrids_c, ms, du_max_c, nmax_exact, nmax
    REAL(C_DOUBLE), VALUE :: ex_tol, vc_tol
    INTEGER(C_INT64_T), DIMENSION(*), INTENT(IN) :: nshape
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: x, y, z
    CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(IN) :: cb
    REAL(C_DOUBLE), DIMENSION(*), INTENT(INOUT) :: u, rhs
    REAL(C_DOUBLE), INTENT(OUT) :: du_last
    INTEGER(C_INT) :: ierr_c
    TYPE(MG_HANDLE) :: h
    INTEGER(IT) :: ndim, nsize, ierr
    ndim = ndim_c
    nsize = PRODUCT(nshape(1:ndim))
    CALL setup_bvp(h, ndim, nshape, INT(ngrids_c,IT), x, y, z, cb, INT(ms,IT), ex_tol, du_max_c == 1, &
                   INT(nmax_exact,IT))
    CALL solve_poisson_bvp(h, nsize, vc_tol, INT(nmax,IT), u(1:nsize), rhs(1:nsize), du_last, ierr)
    CALL delete_mg_handle(h)
    ierr_c = INT(ierr, C_INT)
  END FUNCTION

  SUBROUTINE refk_update_u(nsize, u_old, u_new, metrics) BIND(C, NAME="refk_update_u")
    INTEGER(C_INT64_T), VALUE :: nsize
    REAL(C_DOUBLE), DIMENSION(*), INTENT(IN) :: u_old
    REAL(C_DOUBLE), DIMENSION(*), INTENT(INOUT) :: u_new
    REAL(C_DOUBLE), DIMENSION(2), INTENT(OUT) :: metrics
    CALL update_u(nsize, u_old(1:nsize), u_new(1:nsize), du_max=metrics(1), du_mean=metrics(2))
  END SUBROUTINE

  SUBROUTINE refk_set_debug(flag) BIND(C, NAME="refk_set_debug")
    INTEGER(C_INT), VALUE :: flag
    DEBUG = (flag == 1)
  END SUBROUTINE

END MODULE
