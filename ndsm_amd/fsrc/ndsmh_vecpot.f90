! libndsm_hip - physics driver around the device multigrid solver: current-free
! B = curl A in a box from B.n on its six faces.
!
! Reference being replaced: compute_vector_potential and its helpers,
! ndsm_vector_potential.f90:130-497 (+ :598-691 solve, :699-743 extract_bn,
! :759-872 curl/derivq, :880-950 add_flux_balance_fields, :977-1031
! compute_At_bcs, :1070-1106 trapz_2D).  Pipeline per call:
!   1. B.n on the six faces, face fluxes (trapezoid)               host, O(N^2/3)
!   2. six 2-D all-Neumann Poisson solves  laplace(chi) = B.n - mean DEVICE (mg_solve, ndim=2)
!   3. tangential Dirichlet data A_t = -grad(chi) x n              host, O(N^2/3)
!   4. three 3-D Laplace solves Ax, Ay, Az (the hot path)          DEVICE (mg_solve, ndim=3)
!   5. analytic flux-balance fields, B = curl A                    DEVICE (post.hip)
! Steps 1 and 3 touch only the six faces (O(N^2/3) points) and stay on the host;
! every Poisson solve and all O(N) work runs on the GPU, and A, B are
! downloaded once at the end.
!
! Quirks of the reference kept on purpose (SURVEY 8a):
!   Q2  the Az solve always uses ms = 5                 (:685)
!   Q3' the returned ierr is the flag of the LAST 2-D face solve: `solve` keeps
!       the 3-D flags in a local (:613) and :480 stores the variable last set
!       at :360.  The 3-D flags are reported additively in iopt slot 8.
!   Q4  face fluxes always use dq(1)*dq(2); grad(chi) uses the NORMAL spacing
!       (:301-305, :394-398) - exact only for dx = dy = dz.
module ndsmh_vecpot

  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: error_unit, int64
  use ndsmh_iface
  use ndsmh_mg
  implicit none
  private

  public :: vecpot_solve, poisson_solve
  ! pieces the distributed driver (ndsmh_wvecpot) shares with vecpot_solve
  public :: face_data, face_axis, face_upper, face_t1, face_t2, face_order, face_copy, vecpot_faces, say
  public :: OPT_LEN, IOPT_MS, IOPT_NCYCLES, IOPT_FACE1, IOPT_IERR, IOPT_FLXCRL, IOPT_DEBUG, IOPT_DUMAX, &
            IOPT_NMAXEX, IOPT_FAIL3D, IOPT_NGRIDS, IOPT_NCYC_OUT, IOPT_PREC, ROPT_VTOL, ROPT_CTOL, ROPT_TIM, ROPT_DULAST
  public :: verbose

  ! option slots, 0-based like the reference (ndsm_vector_potential.f90:40-57)
  integer, parameter :: OPT_LEN = 16
  integer, parameter :: IOPT_MS = 0, IOPT_NCYCLES = 1, IOPT_FACE1 = 2, IOPT_IERR = 3, IOPT_FLXCRL = 4, &
                        IOPT_DEBUG = 5, IOPT_DUMAX = 6, IOPT_NMAXEX = 7
  ! additive slots (unused = 0 in the reference, so 0 keeps its behaviour)
  integer, parameter :: IOPT_FAIL3D = 8     ! out: bit c-1 set if 3-D solve c missed vc_tol
  integer, parameter :: IOPT_NGRIDS = 9     ! in : cap on the number of grid levels (0 = reference rule)
  integer, parameter :: IOPT_NCYC_OUT = 10  ! out: V-cycles used by the last 3-D solve that iterated
  integer, parameter :: IOPT_PREC = 11      ! in : 0 fp64 throughout (reference arithmetic), 1 mixed precision for the
                                            !      3-D solves (fp64 residual, fp32 correction V-cycle on level 1)
  integer, parameter :: ROPT_VTOL = 0, ROPT_CTOL = 1, ROPT_TIM = 2
  integer, parameter :: ROPT_DULAST = 3     ! out: du of the last V-cycle of the last 3-D solve

  logical, save :: verbose = .false.

  ! geometry of the six faces: normal axis, layer side, tangential axes
  integer, parameter :: face_axis(6) = [1, 1, 2, 2, 3, 3]
  logical, parameter :: face_upper(6) = [.false., .true., .false., .true., .false., .true.]
  integer, parameter :: face_t1(6) = [2, 2, 1, 1, 1, 1]
  integer, parameter :: face_t2(6) = [3, 3, 3, 3, 2, 2]
  ! A_t = -grad(chi) x n projected on (t1, t2):  (s1*dchi/dt2, s2*dchi/dt1)
  real(wp), parameter :: at_s1(6) = [-1, -1, +1, +1, -1, -1]
  real(wp), parameter :: at_s2(6) = [+1, +1, -1, -1, +1, +1]

  ! the four faces that carry tangential data of component c, in the order the reference writes them
  ! (:647-650, :663-666, :679-682; later writes win on shared edges)
  integer, parameter :: face_order(4, 3) = reshape([3, 4, 5, 6, 1, 2, 5, 6, 1, 2, 3, 4], [4, 3])

  type :: face_data
    integer :: n1 = 0, n2 = 0
    real(wp), allocatable :: bn(:, :), chi(:, :), at1(:, :), at2(:, :)
  end type

contains

  ! debug trace in the reference's format, plus (NDSM_HIP_TIMING set) the wall time since the
  ! previous message - the device is drained first, so the figure belongs to the finished phase
  subroutine say(where, what)
    character(len=*), intent(in) :: where, what
    integer(int64) :: c, r
    real(wp) :: t
    integer :: st, rc
    logical, save :: first = .true., timing = .false.
    real(wp), save :: t_last = 0
    if (first) then
      call get_environment_variable("NDSM_HIP_TIMING", status=st)
      timing = (st == 0)
      first = .false.
    end if
    if (timing) then
      rc = ndsmk_sync()
      call system_clock(c, r)
      t = real(c, wp) / real(r, wp)
      write (error_unit, '(A,F9.3,A)') "TIMING(+", t - t_last, " s) before: "//what
      t_last = t
    end if
    if (verbose) write (error_unit, '(A)') "DEBUG("//where//"):"//what
  end subroutine

  ! ------------------------------------------------------------------
  ! Scalar Poisson problem on the device (host buffers in and out).
  ! u: initial guess incl. Dirichlet face data on entry, solution on exit.
  ! h_rhs = c_null_ptr means rhs == 0.
  ! ------------------------------------------------------------------
  function poisson_solve(ndim, nshape, qx, qy, qz, bcs, ms, ex_tol, use_max, nmax_exact, ngrids_req, &
                         vc_tol, nmax, h_u, h_rhs, du_last, ncycles, ierr, hist, precision) result(rc)
    integer, intent(in) :: ndim, ms, nmax_exact, ngrids_req, nmax
    integer, intent(in), optional :: precision
    integer(c_int32_t), intent(in) :: nshape(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    character(len=1), intent(in) :: bcs(:)
    real(wp), intent(in) :: ex_tol, vc_tol
    logical, intent(in) :: use_max
    type(c_ptr), intent(in) :: h_u, h_rhs
    real(wp), intent(out) :: du_last
    integer, intent(out) :: ncycles, ierr
    real(wp), intent(inout), optional :: hist(:)
    integer(c_int) :: rc
    type(mg_solver) :: s
    integer(ik) :: sweeps, bad

    ierr = 1; ncycles = 0; du_last = huge(du_last)
    rc = mg_create(s, ndim, nshape, qx, qy, qz, bcs, ngrids_req)
    if (rc == 0) then
      s%ms = ms; s%ex_tol = ex_tol; s%use_max = use_max; s%nmax_exact = nmax_exact
      if (present(precision)) s%precision = precision
      rc = mg_set_u(s, h_u)
    end if
    if (rc == 0) then
      if (c_associated(h_rhs)) then
        rc = mg_set_rhs(s, h_rhs)
      else
        rc = mg_zero_rhs(s)
      end if
    end if
    if (rc == 0) rc = mg_solve(s, vc_tol, nmax, du_last, ncycles, ierr, hist)
    if (rc == 0) rc = mg_get_u(s, h_u)
    if (rc == 0) then
      if (ierr /= 0) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
      rc = mg_read_info(s, sweeps, bad)
      if (rc == 0 .and. bad > 0) &
        print *, "Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged"
    end if
    call mg_destroy(s)
  end function

  ! ------------------------------------------------------------------
  ! Steps 1b-3 of the pipeline, on whole faces: fluxes of fc(:)%bn (:283-306), the six 2-D
  ! all-Neumann solves on the device (:338-365) and the tangential data A_t = -grad(chi) x n
  ! (:387-399, :977-1031).  In: fc(f)%bn (allocated with chi, at1, at2).  Out: phi, fc(f)%at1/at2,
  ! ierr2d = flag of the LAST face solve (Q3').
  ! ------------------------------------------------------------------
  function vecpot_faces(iopt, ropt, qx, qy, qz, dq, span, fc, phi, ierr2d) result(rc)
    integer(ik), intent(in) :: iopt(0:OPT_LEN - 1)
    real(wp), intent(in) :: ropt(0:OPT_LEN - 1)
    real(wp), intent(in), target :: qx(:), qy(:), qz(:)
    real(wp), intent(in) :: dq(3), span(3)
    type(face_data), intent(inout), target :: fc(6)
    real(wp), intent(out) :: phi(6)
    integer, intent(out) :: ierr2d
    integer(c_int) :: rc
    character(len=*), parameter :: me = "compute_vector_potential"
    type(mg_solver) :: s2
    real(wp) :: area(6), du_last, fac
    integer :: f, i, j, ncyc, pair
    integer(ik) :: sweeps, bad
    integer(c_int32_t) :: fshape(3)
    logical :: live2
    character(len=1) :: bc2(4)
    real(wp), pointer :: qa(:), qb(:)

    rc = 0
    live2 = .false.
    do f = 1, 6
      phi(f) = trapezoid(fc(f)%bn, dq(1), dq(2))            ! Q4
    end do
    area = [span(2) * span(3), span(2) * span(3), span(1) * span(3), span(1) * span(3), &
            span(1) * span(2), span(1) * span(2)]

    ! ---- 2. chi on every face: 2-D all-Neumann solves on the device ---
    call say(me, "Solve BVP on each boundary...")
    ierr2d = 0
    bc2 = 'N'
    do pair = 1, 3
      f = 2 * pair - 1
      qa => axis_mesh(face_t1(f)); qb => axis_mesh(face_t2(f))
      fshape = [int(fc(f)%n1, c_int32_t), int(fc(f)%n2, c_int32_t), 1_c_int32_t]
      rc = mg_create(s2, 2, fshape, qa, qb, qb, bc2, int(iopt(IOPT_NGRIDS))); live2 = .true.
      if (rc /= 0) goto 900
      s2%ms = int(iopt(IOPT_MS)); s2%ex_tol = ropt(ROPT_CTOL); s2%use_max = (iopt(IOPT_DUMAX) == 1)
      s2%nmax_exact = int(iopt(IOPT_NMAXEX))
      do f = 2 * pair - 1, 2 * pair
        fc(f)%chi = 0
        fc(f)%bn = fc(f)%bn - phi(f) / area(f)
        rc = mg_set_u(s2, c_loc(fc(f)%chi)); if (rc /= 0) goto 900
        rc = mg_set_rhs(s2, c_loc(fc(f)%bn)); if (rc /= 0) goto 900
        rc = mg_reset_info(s2); if (rc /= 0) goto 900
        rc = mg_solve(s2, ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du_last, ncyc, ierr2d)
        if (rc /= 0) goto 900
        rc = mg_get_u(s2, c_loc(fc(f)%chi)); if (rc /= 0) goto 900
        if (ierr2d /= 0) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
        if (mg_read_info(s2, sweeps, bad) == 0) then
          if (bad > 0) print *, "Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged"
        end if
      end do
      call mg_destroy(s2); live2 = .false.
    end do

    ! ---- 3. A_t = -grad(chi) x n --------------------------------------
    call say(me, "Compute vector potential boundary conditions...")
    do f = 1, 6
      fac = 1.0_wp / (2.0_wp * dq(face_axis(f)))            ! Q4: the normal spacing
      do j = 1, fc(f)%n2
        do i = 1, fc(f)%n1
          call tangential(fc(f), f, i, j, fac)
        end do
      end do
    end do

900 continue
    if (live2) call mg_destroy(s2)

  contains

    function axis_mesh(k) result(q)
      integer, intent(in) :: k
      real(wp), pointer :: q(:)
      select case (k)
      case (1); q => qx
      case (2); q => qy
      case default
        q => qz
      end select
    end function

  end function

  ! ------------------------------------------------------------------
  ! The whole ndsm_vector_solve pipeline.  A, B: host arrays (nx,ny,nz,3).
  ! One 3-D solver (device arrays + transfer tables) serves Ax, Ay and Az; one
  ! 2-D solver serves the two faces of each axis.  A and B live in HBM from the
  ! first upload to the single download at the end.
  ! ------------------------------------------------------------------
  function vecpot_solve(n3, iopt, ropt, qx, qy, qz, A, B) result(rc)
    integer(c_int32_t), intent(in) :: n3(3)
    integer(ik), intent(inout) :: iopt(0:OPT_LEN - 1)
    real(wp), intent(inout) :: ropt(0:OPT_LEN - 1)
    real(wp), intent(in), target :: qx(:), qy(:), qz(:)
    real(wp), intent(inout), target, contiguous :: A(:, :, :, :), B(:, :, :, :)
    integer(c_int) :: rc

    character(len=*), parameter :: me = "compute_vector_potential"
    type(face_data), target :: fc(6)
    type(mg_solver) :: s3
    real(wp) :: dq(3), span(3), phi(6), du_last
    integer :: f, ax, c, i, lay, ierr2d, ierr3d, ncyc, ngr
    integer(ik) :: sweeps, bad, npts
    logical :: use_max, live3
    character(len=1) :: bc3(6)
    real(wp), pointer :: comp(:, :, :)
    type(c_ptr) :: dA, dB, dmesh
    integer(c_size_t) :: nb, off_y, off_z

    rc = 0
    use_max = (iopt(IOPT_DUMAX) == 1)
    ngr = int(iopt(IOPT_NGRIDS))
    iopt(IOPT_FAIL3D) = 0
    live3 = .false.
    dA = c_null_ptr; dB = c_null_ptr; dmesh = c_null_ptr

    ! :201-221 extent and spacing; fewer than two points is the reference's only input check
    if (any(n3 < 2)) then
      iopt(IOPT_IERR) = 1
      return
    end if
    span = [maxval(qx) - minval(qx), maxval(qy) - minval(qy), maxval(qz) - minval(qz)]
    dq = [qx(2) - qx(1), qy(2) - qy(1), qz(2) - qz(1)]
    npts = product(int(n3, ik))
    nb = int(npts, c_size_t) * 8_c_size_t

    ! ---- 1. B.n on the faces and their fluxes ------------------------
    call say(me, "Allocate memory to hold boundary conditions...")
    do f = 1, 6
      ax = face_axis(f)
      fc(f)%n1 = n3(face_t1(f)); fc(f)%n2 = n3(face_t2(f))
      allocate (fc(f)%bn(fc(f)%n1, fc(f)%n2), fc(f)%chi(fc(f)%n1, fc(f)%n2))
      allocate (fc(f)%at1(fc(f)%n1, fc(f)%n2), fc(f)%at2(fc(f)%n1, fc(f)%n2))
      lay = merge(int(n3(ax)), 1, face_upper(f))
      comp => B(:, :, :, ax)
      call face_copy(comp, ax, lay, fc(f)%bn, to_face=.true.)
    end do
    rc = vecpot_faces(iopt, ropt, qx, qy, qz, dq, span, fc, phi, ierr2d)
    if (rc /= 0) goto 900

    ! ---- 4. the three 3-D Laplace problems ----------------------------
    call say(me, "Solve BVP 3D...")
    rc = ndsmk_alloc(dA, 3_c_size_t * nb); if (rc /= 0) goto 900
    bc3 = 'D'; bc3(1) = 'N'; bc3(4) = 'N'
    rc = mg_create(s3, 3, n3, qx, qy, qz, bc3, ngr); live3 = .true.
    if (rc /= 0) goto 900
    s3%ex_tol = ropt(ROPT_CTOL); s3%use_max = use_max; s3%nmax_exact = int(iopt(IOPT_NMAXEX))
    s3%precision = int(iopt(IOPT_PREC))
    rc = mg_zero_rhs(s3); if (rc /= 0) goto 900               ! :640-641 rhs = 0
    do c = 1, 3
      comp => A(:, :, :, c)
      do i = 1, 4
        f = face_order(i, c)
        lay = merge(int(n3(face_axis(f))), 1, face_upper(f))
        ! component c is the t1 direction of face f if t1 == c, else its t2 direction
        if (face_t1(f) == c) then
          call face_copy(comp, face_axis(f), lay, fc(f)%at1, to_face=.false.)
        else
          call face_copy(comp, face_axis(f), lay, fc(f)%at2, to_face=.false.)
        end if
      end do
      bc3 = 'D'
      bc3(c) = 'N'; bc3(3 + c) = 'N'                        ! :655,:671,:687
      rc = mg_set_bcs(s3, bc3); if (rc /= 0) goto 900
      s3%ms = merge(5, int(iopt(IOPT_MS)), c == 3)          ! Q2
      rc = mg_set_u(s3, c_loc(comp)); if (rc /= 0) goto 900
      rc = mg_reset_info(s3); if (rc /= 0) goto 900
      rc = mg_solve(s3, ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du_last, ncyc, ierr3d)
      if (rc /= 0) goto 900
      rc = mg_export_u(s3, dptr_offset(dA, int(c - 1, c_size_t) * nb)); if (rc /= 0) goto 900
      call warn_if_needed(s3, ierr3d)
      if (ierr3d /= 0) iopt(IOPT_FAIL3D) = ior(iopt(IOPT_FAIL3D), ishft(1_ik, c - 1))
      if (ncyc > 1 .or. c == 1) then
        iopt(IOPT_NCYC_OUT) = ncyc
        ropt(ROPT_DULAST) = du_last
      end if
    end do
    call mg_destroy(s3); live3 = .false.
    ! B takes the memory the 3-D hierarchy has just returned: the peak is A + one hierarchy, not A + B + it
    rc = ndsmk_alloc(dB, 3_c_size_t * nb); if (rc /= 0) goto 900

    ! ---- 5. flux balance + curl on the device (default order :467-477) -
    call say(me, "Compute B = curl(B) and flux correction...")
    off_y = int(n3(1), c_size_t) * 8_c_size_t
    off_z = off_y + int(n3(2), c_size_t) * 8_c_size_t
    rc = ndsmk_alloc(dmesh, off_z + int(n3(3), c_size_t) * 8_c_size_t); if (rc /= 0) goto 900
    rc = ndsmk_h2d(dmesh, c_loc(qx), int(n3(1), c_size_t) * 8_c_size_t); if (rc /= 0) goto 900
    rc = ndsmk_h2d(dptr_offset(dmesh, off_y), c_loc(qy), int(n3(2), c_size_t) * 8_c_size_t); if (rc /= 0) goto 900
    rc = ndsmk_h2d(dptr_offset(dmesh, off_z), c_loc(qz), int(n3(3), c_size_t) * 8_c_size_t); if (rc /= 0) goto 900
    if (iopt(IOPT_FLXCRL) == 1) print *, "FLAG SET: FLXCRL"
    rc = ndsmk_balance_curl(dA, dB, n3, dmesh, dptr_offset(dmesh, off_y), dptr_offset(dmesh, off_z), &
                            phi, span, dq, merge(1_c_int, 0_c_int, iopt(IOPT_FLXCRL) == 1))
    if (rc /= 0) goto 900
    rc = ndsmk_d2h(c_loc(A), dA, 3_c_size_t * nb); if (rc /= 0) goto 900
    rc = ndsmk_d2h(c_loc(B), dB, 3_c_size_t * nb); if (rc /= 0) goto 900

    iopt(IOPT_IERR) = ierr2d                                ! Q3'
    call say(me, "Deallocate memory...")

900 continue
    if (live3) call mg_destroy(s3)
    i = ndsmk_free(dA); i = ndsmk_free(dB); i = ndsmk_free(dmesh)

  contains

    subroutine warn_if_needed(sv, ie)
      type(mg_solver), intent(in) :: sv
      integer, intent(in) :: ie
      integer(c_int) :: r
      if (ie /= 0) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
      r = mg_read_info(sv, sweeps, bad)
      if (r == 0 .and. bad > 0) print *, "Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged"
    end subroutine

  end function

  ! central differences of chi, zero on the face's own edges (:1007-1017)
  subroutine tangential(fd, f, i, j, fac)
    type(face_data), intent(inout) :: fd
    integer, intent(in) :: f, i, j
    real(wp), intent(in) :: fac
    real(wp) :: d1, d2
    d1 = 0; d2 = 0
    if (i > 1 .and. i < fd%n1) d1 = fac * (fd%chi(i + 1, j) - fd%chi(i - 1, j))
    if (j > 1 .and. j < fd%n2) d2 = fac * (fd%chi(i, j + 1) - fd%chi(i, j - 1))
    fd%at1(i, j) = at_s1(f) * d2
    fd%at2(i, j) = at_s2(f) * d1
  end subroutine

  ! copy a face layer of a 3-D array to / from a 2-D array (extract_bn, :699-743)
  subroutine face_copy(v, axis, lay, face, to_face)
    real(wp), intent(inout) :: v(:, :, :)
    integer, intent(in) :: axis, lay
    real(wp), intent(inout) :: face(:, :)
    logical, intent(in) :: to_face
    select case (axis)
    case (1)
      if (to_face) then
        face = v(lay, :, :)
      else
        v(lay, :, :) = face
      end if
    case (2)
      if (to_face) then
        face = v(:, lay, :)
      else
        v(:, lay, :) = face
      end if
    case (3)
      if (to_face) then
        face = v(:, :, lay)
      else
        v(:, :, lay) = face
      end if
    end select
  end subroutine

  ! 2-D trapezoid rule, weights 1 / 1/2 (edges) / 1/4 (corners) (:1070-1106)
  function trapezoid(f, h1, h2) result(s)
    real(wp), intent(in) :: f(:, :), h1, h2
    real(wp) :: s, w
    integer :: i, j, n1, n2
    logical :: ei, ej
    n1 = size(f, 1); n2 = size(f, 2)
    s = 0
    do j = 1, n2
      ej = (j == 1 .or. j == n2)
      do i = 1, n1
        ei = (i == 1 .or. i == n1)
        w = 1.0_wp
        if (ei .or. ej) w = 0.5_wp
        if (ei .and. ej) w = 0.25_wp
        s = s + w * f(i, j)
      end do
    end do
    s = s * h1 * h2
  end function

end module ndsmh_vecpot
