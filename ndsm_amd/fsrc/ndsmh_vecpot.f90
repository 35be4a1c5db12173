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
  public :: vecpot_ctx, vecpot_ctx_create, vecpot_ctx_destroy, vecpot_ctx_matches, vecpot_run, vecpot_cache_drop
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

  ! up to this many points the three 3-D component solves get a hierarchy each and run side by side (vecpot_run)
  integer(ik), parameter :: SIDE3D_MAX = 16_ik * 1024_ik * 1024_ik

  type :: face_data
    integer :: n1 = 0, n2 = 0
    real(wp), allocatable :: bn(:, :), chi(:, :), at1(:, :), at2(:, :)
  end type

  ! grid-bound state of the pipeline, reused across calls (vecpot_ctx_create)
  type :: vecpot_ctx
    logical :: live = .false.
    integer(c_int32_t) :: n3(3) = 0
    integer :: ngr = 0
    real(wp), allocatable :: qx(:), qy(:), qz(:)
    type(mg_solver) :: s3v(3), s2(6)        ! one 2-D hierarchy per face: the six face solves run side by side;
                                            ! s3v(1): the 3-D hierarchy; s3v(2:3): small grids only, where the three
                                            ! component solves run side by side as well (live3x)
    logical :: live3 = .false., live3x = .false., live2(6) = .false.
    type(c_ptr) :: dA = c_null_ptr, dB = c_null_ptr, dmesh = c_null_ptr
    type(c_ptr) :: dbn = c_null_ptr, dchi = c_null_ptr, dphi = c_null_ptr   ! packed faces: B.n, chi; six fluxes
    type(c_ptr) :: hbn = c_null_ptr                                         ! pinned staging of the six faces
    integer(ik) :: foff(6) = 0, ftotal = 0
  end type

  type(vecpot_ctx), save, target :: cache
  logical, save :: cache_busy = .false.     ! ndsm_vector_solve is running on the cached context

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
      write (error_unit, '(A,F12.6,A)') "TIMING(+", t - t_last, " s) before: "//what
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
    type(mg_solver) :: s2(6)
    real(wp) :: area(6), du6(6), fac
    integer :: f, i, j, ncyc6(6), ierr6(6), st
    integer(ik) :: sweeps, bad
    integer(c_int32_t) :: fshape(3)
    logical :: live2(6)
    character(len=1) :: bc2(4)
    character(len=8) :: envbuf
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
    ! (one hierarchy per face, the six solves side by side: mg_solve_lanes, as in vecpot_run)
    ierr2d = 0
    bc2 = 'N'
    do f = 1, 6
      qa => axis_mesh(face_t1(f)); qb => axis_mesh(face_t2(f))
      fshape = [int(fc(f)%n1, c_int32_t), int(fc(f)%n2, c_int32_t), 1_c_int32_t]
      rc = mg_create(s2(f), 2, fshape, qa, qb, qb, bc2, int(iopt(IOPT_NGRIDS))); live2(f) = .true.
      if (rc /= 0) goto 900
      s2(f)%ms = int(iopt(IOPT_MS)); s2(f)%ex_tol = ropt(ROPT_CTOL); s2(f)%use_max = (iopt(IOPT_DUMAX) == 1)
      s2(f)%nmax_exact = int(iopt(IOPT_NMAXEX))
      fc(f)%chi = 0
      fc(f)%bn = fc(f)%bn - phi(f) / area(f)
      rc = mg_set_u(s2(f), c_loc(fc(f)%chi)); if (rc /= 0) goto 900
      rc = mg_set_rhs(s2(f), c_loc(fc(f)%bn)); if (rc /= 0) goto 900
      rc = mg_reset_info(s2(f)); if (rc /= 0) goto 900
    end do
    call get_environment_variable("NDSM_HIP_FACE_LANES", envbuf, status=st)
    if (st == 0 .and. envbuf(1:1) == "0") then
      do f = 1, 6
        rc = mg_solve(s2(f), ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du6(f), ncyc6(f), ierr6(f))
        if (rc /= 0) goto 900
      end do
    else
      rc = mg_solve_lanes(s2, ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du6, ncyc6, ierr6)
      if (rc /= 0) goto 900
    end if
    do f = 1, 6
      rc = mg_get_u(s2(f), c_loc(fc(f)%chi)); if (rc /= 0) goto 900
      ierr2d = ierr6(f)
      if (ierr2d /= 0) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
      if (mg_read_info(s2(f), sweeps, bad) == 0) then
        if (bad > 0) print *, "Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged"
      end if
      call mg_destroy(s2(f)); live2(f) = .false.
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
    do f = 1, 6
      if (live2(f)) call mg_destroy(s2(f))
    end do

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
  ! Persistent state of the pipeline (SURVEY 8f-4): everything that depends on the grid only -
  ! the 3-D hierarchy with its transfer tables, the three 2-D face hierarchies, device arrays for
  ! A, B, the mesh and the six faces, a pinned staging buffer - created once and reused by every
  ! later call on the same (shape, mesh, level cap).  The reference builds and frees all of it per
  ! component and per call (ndsm_vector_potential.f90:652-689, ndsm_multigrid_core.f90:165-329).
  ! ------------------------------------------------------------------
  function vecpot_ctx_matches(ctx, n3, qx, qy, qz, ngr) result(same)
    type(vecpot_ctx), intent(in) :: ctx
    integer(c_int32_t), intent(in) :: n3(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    integer, intent(in) :: ngr
    logical :: same
    same = .false.
    if (.not. ctx%live) return
    if (any(ctx%n3 /= n3) .or. ctx%ngr /= ngr) return
    if (size(ctx%qx) /= size(qx) .or. size(ctx%qy) /= size(qy) .or. size(ctx%qz) /= size(qz)) return
    same = all(ctx%qx == qx) .and. all(ctx%qy == qy) .and. all(ctx%qz == qz)
  end function

  subroutine vecpot_ctx_destroy(ctx)
    type(vecpot_ctx), intent(inout) :: ctx
    integer :: p
    integer(c_int) :: rc
    ! (no transfer of this context is in flight here: vecpot_run drains its own before it returns, whatever its
    ! outcome - and draining here would void the tickets of ANOTHER context's running call when the low-memory
    ! hook evicts the cached one from inside it)
    if (ctx%live3) call mg_destroy(ctx%s3v(1))
    if (ctx%live3x) then
      call mg_destroy(ctx%s3v(2)); call mg_destroy(ctx%s3v(3))
    end if
    ctx%live3 = .false.; ctx%live3x = .false.
    do p = 1, 6
      if (ctx%live2(p)) call mg_destroy(ctx%s2(p))
      ctx%live2(p) = .false.
    end do
    rc = ndsmk_free(ctx%dA); ctx%dA = c_null_ptr
    rc = ndsmk_free(ctx%dB); ctx%dB = c_null_ptr
    rc = ndsmk_free(ctx%dmesh); ctx%dmesh = c_null_ptr
    rc = ndsmk_free(ctx%dbn); ctx%dbn = c_null_ptr
    rc = ndsmk_free(ctx%dchi); ctx%dchi = c_null_ptr
    rc = ndsmk_free(ctx%dphi); ctx%dphi = c_null_ptr
    rc = ndsmk_host_free(ctx%hbn); ctx%hbn = c_null_ptr
    if (allocated(ctx%qx)) deallocate (ctx%qx, ctx%qy, ctx%qz)
    ctx%live = .false.
  end subroutine

  ! the grid-only part: mesh copies, face buffers, the three 2-D hierarchies
  function vecpot_ctx_create(ctx, n3, qx, qy, qz, ngr) result(rc)
    type(vecpot_ctx), intent(inout), target :: ctx
    integer(c_int32_t), intent(in) :: n3(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    integer, intent(in) :: ngr
    integer(c_int) :: rc
    integer :: pair, f
    integer(c_int32_t) :: fshape(3)
    integer(c_size_t) :: off_y, off_z
    character(len=1) :: bc2(4)
    real(wp), pointer :: qa(:), qb(:)

    call vecpot_ctx_destroy(ctx)
    rc = ndsmk_init(-1_c_int); if (rc /= 0) return
    ctx%n3 = n3; ctx%ngr = ngr
    allocate (ctx%qx(n3(1)), ctx%qy(n3(2)), ctx%qz(n3(3)))
    ctx%qx = qx(1:n3(1)); ctx%qy = qy(1:n3(2)); ctx%qz = qz(1:n3(3))
    ctx%live = .true.
    rc = ndsmk_face_offsets(n3, ctx%foff, ctx%ftotal); if (rc /= 0) return
    rc = ndsmk_alloc(ctx%dbn, int(ctx%ftotal, c_size_t) * 8_c_size_t); if (rc /= 0) return
    rc = ndsmk_alloc(ctx%dchi, int(ctx%ftotal, c_size_t) * 8_c_size_t); if (rc /= 0) return
    rc = ndsmk_alloc(ctx%dphi, 64_c_size_t); if (rc /= 0) return
    rc = ndsmk_host_alloc(ctx%hbn, int(ctx%ftotal + 8, c_size_t) * 8_c_size_t); if (rc /= 0) return
    off_y = int(n3(1), c_size_t) * 8_c_size_t
    off_z = off_y + int(n3(2), c_size_t) * 8_c_size_t
    rc = ndsmk_alloc(ctx%dmesh, off_z + int(n3(3), c_size_t) * 8_c_size_t); if (rc /= 0) return
    rc = ndsmk_h2d(ctx%dmesh, c_loc(ctx%qx), int(n3(1), c_size_t) * 8_c_size_t); if (rc /= 0) return
    rc = ndsmk_h2d(dptr_offset(ctx%dmesh, off_y), c_loc(ctx%qy), int(n3(2), c_size_t) * 8_c_size_t); if (rc /= 0) return
    rc = ndsmk_h2d(dptr_offset(ctx%dmesh, off_z), c_loc(ctx%qz), int(n3(3), c_size_t) * 8_c_size_t); if (rc /= 0) return
    bc2 = 'N'
    do f = 1, 6
      qa => ctx_axis(ctx, face_t1(f)); qb => ctx_axis(ctx, face_t2(f))
      fshape = [n3(face_t1(f)), n3(face_t2(f)), 1_c_int32_t]
      rc = mg_create(ctx%s2(f), 2, fshape, qa, qb, qb, bc2, ngr); ctx%live2(f) = .true.
      if (rc /= 0) return
    end do
  end function

  function ctx_axis(ctx, k) result(q)
    type(vecpot_ctx), intent(in), target :: ctx
    integer, intent(in) :: k
    real(wp), pointer :: q(:)
    select case (k)
    case (1); q => ctx%qx
    case (2); q => ctx%qy
    case default
      q => ctx%qz
    end select
  end function

  ! the library's own context behind ndsm_vector_solve (the reference ABI has no handle: SURVEY 8b
  ! "Ownership" - an internal cache keyed by shape and mesh is invisible to the caller); dropped by
  ! ndsm_hip_shutdown / a re-target of the runtime
  subroutine vecpot_cache_drop() bind(c)
    call vecpot_ctx_destroy(cache)
  end subroutine

  ! a device allocation somewhere in the library has failed: the cached context (13 GiB at 512^3) is only a
  ! convenience - give it back, unless ndsm_vector_solve is using it right now
  subroutine vecpot_cache_evict() bind(c)
    if (.not. cache%live .or. cache_busy) return
    call vecpot_ctx_destroy(cache)
  end subroutine

  function vecpot_solve(n3, iopt, ropt, qx, qy, qz, A, B) result(rc)
    integer(c_int32_t), intent(in) :: n3(3)
    integer(ik), intent(inout) :: iopt(0:OPT_LEN - 1)
    real(wp), intent(inout) :: ropt(0:OPT_LEN - 1)
    real(wp), intent(in), target :: qx(:), qy(:), qz(:)
    real(wp), intent(inout), target, contiguous :: A(:, :, :, :), B(:, :, :, :)
    integer(c_int) :: rc
    integer :: st
    if (any(n3 < 2)) then              ! :213-216 the reference's only input check
      iopt(IOPT_FAIL3D) = 0
      iopt(IOPT_IERR) = 1
      rc = 0
      return
    end if
    cache_busy = .true.      ! (an allocation failure below must not evict the context that is being built / used)
    call get_environment_variable("NDSM_HIP_NO_CACHE", status=st)      ! A/B testing: rebuild everything per call
    if (st == 0) call vecpot_ctx_destroy(cache)
    if (.not. vecpot_ctx_matches(cache, n3, qx, qy, qz, int(iopt(IOPT_NGRIDS)))) then
      rc = vecpot_ctx_create(cache, n3, qx, qy, qz, int(iopt(IOPT_NGRIDS)))
      if (rc /= 0) then
        call vecpot_ctx_destroy(cache)
        cache_busy = .false.
        return
      end if
      call ndsmk_at_reset(c_funloc(vecpot_cache_drop))
      call ndsmk_on_low_memory(c_funloc(vecpot_cache_evict))
    end if
    rc = vecpot_run(cache, iopt, ropt, c_loc(A), c_loc(B), .false.)
    cache_busy = .false.
    if (rc /= 0 .or. st == 0) call vecpot_ctx_destroy(cache)    ! after an error nothing is assumed about the device state
  end function

  ! ------------------------------------------------------------------
  ! The whole ndsm_vector_solve pipeline on a prepared context.  A, B: (nx,ny,nz,3), on the HOST
  ! (on_device = .false.: the reference ABI) or in HBM (on_device: the additive device-resident entry).
  !
  !   host entry  : B.n of the six faces is gathered on the host into pinned memory (the only part of B
  !                 that is ever read) and uploaded once, 8 N^(2/3) bytes; a worker thread checks the
  !                 initial guess and uploads only components that are not all zero; every finished
  !                 component of A is downloaded behind the next component's solve; B follows the curl.
  !   device entry: B.n is extracted by a kernel; nothing crosses PCIe but the six fluxes and the
  !                 16-byte convergence read-backs.
  ! Between the face upload and the downloads nothing else travels: fluxes, right-hand sides of the 2-D
  ! problems, chi, A_t and the face writes into the 3-D initial guess are device kernels (faces.hip).
  ! ------------------------------------------------------------------
  function vecpot_run(ctx, iopt, ropt, pA, pB, on_device) result(rc)
    type(vecpot_ctx), intent(inout), target :: ctx
    integer(ik), intent(inout) :: iopt(0:OPT_LEN - 1)
    real(wp), intent(inout) :: ropt(0:OPT_LEN - 1)
    type(c_ptr), intent(in) :: pA, pB
    logical, intent(in) :: on_device
    integer(c_int) :: rc

    character(len=*), parameter :: me = "compute_vector_potential"
    real(wp) :: dq(3), span(3), area(6), du_last, fac
    real(wp), target :: phi(6)
    real(wp), pointer, contiguous :: hA(:, :, :, :), hB(:, :, :, :), stage(:)
    integer(c_int32_t) :: n3(3)
    integer :: f, c, i, ierr2d, ierr3d, ncyc, st
    integer :: ncyc6(6), ierr6(6), ncyc3(3), ierr3(3)
    real(wp) :: du6(6), du3(3)
    character(len=8) :: envbuf
    integer(ik) :: sweeps, bad, npts, cnt
    integer(c_int) :: tick_up(3), tick, zero_flag, rcb
    logical :: use_max, resident, host_faces, late_balance, bz_done
    character(len=1) :: bc3(6)
    type(c_ptr) :: dAout, dBout, u3, rhs2, u2
    integer(c_size_t) :: nb, off_y, off_z, fr, tot
    type(face_data), target :: fc(6)

    rc = 0
    n3 = ctx%n3
    use_max = (iopt(IOPT_DUMAX) == 1)
    iopt(IOPT_FAIL3D) = 0
    span = [maxval(ctx%qx) - minval(ctx%qx), maxval(ctx%qy) - minval(ctx%qy), maxval(ctx%qz) - minval(ctx%qz)]
    dq = [ctx%qx(2) - ctx%qx(1), ctx%qy(2) - ctx%qy(1), ctx%qz(2) - ctx%qz(1)]      ! :201-221
    area = [span(2) * span(3), span(2) * span(3), span(1) * span(3), span(1) * span(3), &
            span(1) * span(2), span(1) * span(2)]
    npts = product(int(n3, ik))
    nb = int(npts, c_size_t) * 8_c_size_t
    off_y = int(n3(1), c_size_t) * 8_c_size_t
    off_z = off_y + int(n3(2), c_size_t) * 8_c_size_t
    late_balance = (iopt(IOPT_FLXCRL) == 1)     ! :455-465: curl first, fields added to A AND B afterwards
    bz_done = .false.
    call get_environment_variable("NDSM_HIP_HOST_FACES", status=st)   ! A/B testing: the host face phase (vecpot_faces)
    host_faces = (st == 0) .and. .not. on_device
    tick_up = -1
    if (.not. on_device) then
      call c_f_pointer(pA, hA, [int(n3(1)), int(n3(2)), int(n3(3)), 3])
      call c_f_pointer(pB, hB, [int(n3(1)), int(n3(2)), int(n3(3)), 3])
    end if

    ! ---- device arrays of this call ----------------------------------
    if (.not. ctx%live3) then
      bc3 = 'D'; bc3(1) = 'N'; bc3(4) = 'N'
      rc = mg_create(ctx%s3v(1), 3, n3, ctx%qx, ctx%qy, ctx%qz, bc3, ctx%ngr); ctx%live3 = .true.
      if (rc /= 0) return
      rc = mg_zero_rhs(ctx%s3v(1)); if (rc /= 0) return            ! :640-641 rhs = 0
      ! The three component solves are independent (:643-689).  Up to a few million points a V-cycle is ~150 launches
      ! of a few microseconds each - dispatch latency, not bandwidth - and one host round trip: with a hierarchy per
      ! component they run side by side on three streams (mg_solve_lanes, as the six face solves do), each one the
      ! kernels it would run alone in the same order.  NDSM_HIP_NO_SIDE3D=1: one after the other (same bits).
      call get_environment_variable("NDSM_HIP_NO_SIDE3D", status=st)
      if (npts <= SIDE3D_MAX .and. st /= 0) then
        do c = 2, 3
          rc = mg_create(ctx%s3v(c), 3, n3, ctx%qx, ctx%qy, ctx%qz, bc3, ctx%ngr)
          if (rc == 0) rc = mg_zero_rhs(ctx%s3v(c))
          if (rc /= 0) then                                          ! (no memory for them: one after the other)
            call mg_destroy(ctx%s3v(c))
            if (c == 3) call mg_destroy(ctx%s3v(2))
            rc = 0
            exit
          end if
          if (c == 3) ctx%live3x = .true.
        end do
      end if
    end if
    resident = .true.
    if (on_device) then
      dAout = pA; dBout = pB
    else
      if (.not. c_associated(ctx%dA)) then
        rc = ndsmk_alloc(ctx%dA, 3_c_size_t * nb); if (rc /= 0) return
      end if
      if (.not. c_associated(ctx%dB)) then
        ! B next to the hierarchy if HBM has room for both; otherwise it takes the memory the 3-D
        ! hierarchy returns after the solves (the peak is then A + one hierarchy, as before)
        rc = ndsmk_mem_info(fr, tot); if (rc /= 0) return
        resident = fr >= 3_c_size_t * nb + ishft(1_c_size_t, 31)
        call get_environment_variable("NDSM_HIP_LEAN", status=st)      ! testing: take the small-HBM sequence
        if (st == 0) resident = .false.
        if (resident) then
          rc = ndsmk_alloc(ctx%dB, 3_c_size_t * nb); if (rc /= 0) return
        end if
      end if
      dAout = ctx%dA; dBout = ctx%dB
      ! the worker thread looks at the caller's initial guess meanwhile: components that are not all zero
      ! (the reference's Python passes zeros, ndsm.py:176) go up into their slot of dA
      do c = 1, 3
        rc = ndsmk_bg_upload_unless_zero(c_loc(hA(1, 1, 1, c)), dptr_offset(ctx%dA, int(c - 1, c_size_t) * nb), nb, tick_up(c))
        if (rc /= 0) goto 900
      end do
    end if

    ! ---- 1. B.n on the faces, their fluxes -----------------------------
    call say(me, "Allocate memory to hold boundary conditions...")
    if (host_faces) then
      do f = 1, 6
        fc(f)%n1 = n3(face_t1(f)); fc(f)%n2 = n3(face_t2(f))
        allocate (fc(f)%bn(fc(f)%n1, fc(f)%n2), fc(f)%chi(fc(f)%n1, fc(f)%n2))
        allocate (fc(f)%at1(fc(f)%n1, fc(f)%n2), fc(f)%at2(fc(f)%n1, fc(f)%n2))
        call face_gather(hB, n3, f, fc(f)%bn)
      end do
      rc = vecpot_faces(iopt, ropt, ctx%qx, ctx%qy, ctx%qz, dq, span, fc, phi, ierr2d)
      if (rc /= 0) goto 900
    else
      if (on_device) then
        rc = ndsmk_face_extract(pB, n3, ctx%dbn); if (rc /= 0) goto 900
      else
        call c_f_pointer(ctx%hbn, stage, [ctx%ftotal + 8])
        do f = 1, 6
          call face_gather_flat(hB, n3, f, stage(ctx%foff(f) + 1:ctx%foff(f) + int(n3(face_t1(f)), ik) * int(n3(face_t2(f)), ik)))
        end do
        rc = ndsmk_h2d_async(ctx%dbn, ctx%hbn, int(ctx%ftotal, c_size_t) * 8_c_size_t); if (rc /= 0) goto 900
        ! everything this call reads of the caller's arrays has been read (A: by the upload jobs queued above,
        ! B: its six faces just now); both will be overwritten completely.  Callers like numpy hand over
        ! untouched pages: the worker touches them (4 threads) before the downloads come, which then run at
        ! 50 GB/s instead of 13
        rc = ndsmk_bg_first_touch(pA, 3_c_size_t * nb, tick); if (rc /= 0) goto 900
        rc = ndsmk_bg_first_touch(pB, 3_c_size_t * nb, tick); if (rc /= 0) goto 900
      end if
      rc = ndsmk_face_flux(ctx%dbn, n3, dq(1) * dq(2), ctx%dphi); if (rc /= 0) goto 900     ! Q4
      rc = ndsmk_d2h(c_loc(phi), ctx%dphi, 48_c_size_t); if (rc /= 0) goto 900

      ! ---- 2. chi on every face: 2-D all-Neumann solves, right-hand side and result stay in HBM ----
      call say(me, "Solve BVP on each boundary...")
      ! The six problems are independent (:338-365 solves them one after the other) and each is dispatch
      ! latency plus one host round trip per V-cycle: they run in lockstep on six streams (mg_solve_lanes;
      ! every solve executes the kernels it would execute alone, in the same order - same bits).
      ! NDSM_HIP_FACE_LANES=0: one after the other (A/B testing).
      ierr2d = 0
      do f = 1, 6
        associate (s2 => ctx%s2(f))
          s2%ms = int(iopt(IOPT_MS)); s2%ex_tol = ropt(ROPT_CTOL); s2%use_max = use_max
          s2%nmax_exact = int(iopt(IOPT_NMAXEX))
          rhs2 = mg_level_ptr(s2, 1, MG_BUF_RHS, cnt)
          u2 = mg_level_ptr(s2, 1, MG_BUF_U, cnt)
          rc = ndsmk_face_rhs(ctx%dbn, n3, int(f - 1, c_int), ctx%dphi, area(f), rhs2); if (rc /= 0) goto 900
          call mg_mark_rhs_set(s2)
          rc = ndsmk_fill0(u2, int(cnt, c_size_t) * 8_c_size_t); if (rc /= 0) goto 900
          rc = mg_reset_info(s2); if (rc /= 0) goto 900
        end associate
      end do
      call get_environment_variable("NDSM_HIP_FACE_LANES", envbuf, status=st)
      if (st == 0 .and. envbuf(1:1) == "0") then
        do f = 1, 6
          rc = mg_solve(ctx%s2(f), ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du6(f), ncyc6(f), ierr6(f))
          if (rc /= 0) goto 900
        end do
      else
        rc = mg_solve_lanes(ctx%s2, ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du6, ncyc6, ierr6)
        if (rc /= 0) goto 900
      end if
      do f = 1, 6
        u2 = mg_level_ptr(ctx%s2(f), 1, MG_BUF_U, cnt)                  ! (the solver swaps its buffers)
        rc = ndsmk_d2d(dptr_offset(ctx%dchi, int(ctx%foff(f), c_size_t) * 8_c_size_t), u2, int(cnt, c_size_t) * 8_c_size_t)
        if (rc /= 0) goto 900
        ierr2d = ierr6(f)
        if (ierr2d /= 0) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
        if (mg_read_info(ctx%s2(f), sweeps, bad) == 0) then
          if (bad > 0) print *, "Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged"
        end if
      end do
      call say(me, "Compute vector potential boundary conditions...")
    end if

    ! ---- 4. the three 3-D Laplace problems ----------------------------
    call say(me, "Solve BVP 3D...")
    if (ctx%live3x .and. iopt(IOPT_PREC) == 0) then
      do c = 1, 3
        rc = prep3(ctx%s3v(c), c); if (rc /= 0) goto 900
      end do
      rc = mg_solve_lanes(ctx%s3v, ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du3, ncyc3, ierr3); if (rc /= 0) goto 900
      do c = 1, 3
        rc = finish3(ctx%s3v(c), c, du3(c), ncyc3(c), ierr3(c)); if (rc /= 0) goto 900
      end do
    else
      do c = 1, 3
        rc = prep3(ctx%s3v(1), c); if (rc /= 0) goto 900
        rc = mg_solve(ctx%s3v(1), ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du_last, ncyc, ierr3d)
        if (rc /= 0) goto 900
        rc = finish3(ctx%s3v(1), c, du_last, ncyc, ierr3d); if (rc /= 0) goto 900
      end do
    end if
    if (.not. on_device .and. .not. c_associated(ctx%dB)) then      ! HBM too small for both (see above)
      call mg_destroy(ctx%s3v(1)); ctx%live3 = .false.
      if (ctx%live3x) then
        call mg_destroy(ctx%s3v(2)); call mg_destroy(ctx%s3v(3)); ctx%live3x = .false.
      end if
      rc = ndsmk_alloc(ctx%dB, 3_c_size_t * nb); if (rc /= 0) goto 900
      dBout = ctx%dB
    end if

    ! ---- 5. B = curl A (and, IOPT_FLXCRL == 1, the fields afterwards) ----
    call say(me, "Compute B = curl(B) and flux correction...")
    if (late_balance) then
      print *, "FLAG SET: FLXCRL"
      rc = ndsmk_balance_curl(dAout, dBout, n3, ctx%dmesh, dptr_offset(ctx%dmesh, off_y), dptr_offset(ctx%dmesh, off_z), &
                              phi, span, dq, 1_c_int)
      if (rc /= 0) goto 900
      if (.not. on_device) then
        rc = ndsmk_bg_download(pA, dAout, 3_c_size_t * nb, tick); if (rc /= 0) goto 900
      end if
    else if (bz_done) then
      rc = ndsmk_curl_component(dAout, dBout, n3, dq, 0_c_int); if (rc /= 0) goto 900
      rc = ndsmk_curl_component(dAout, dBout, n3, dq, 1_c_int); if (rc /= 0) goto 900
    else
      rc = ndsmk_curl(dAout, dBout, n3, dq); if (rc /= 0) goto 900
    end if
    if (.not. on_device) then
      rc = ndsmk_bg_download(pB, dBout, merge(2_c_size_t, 3_c_size_t, bz_done) * nb, tick); if (rc /= 0) goto 900
    end if
    iopt(IOPT_IERR) = ierr2d                                ! Q3'
    call say(me, "Deallocate memory...")

900 continue
    rcb = ndsmk_bg_drain()                                  ! every byte of A and B is home (or the first error)
    if (rc == 0) rc = rcb
    if (rc == 0) rc = ndsmk_sync()
    if (.not. resident) then                                ! keep the peak at A + one hierarchy next time too
      rcb = ndsmk_free(ctx%dB); ctx%dB = c_null_ptr
    end if

  contains

    ! component c of the 3-D phase on solver s3: initial guess, Dirichlet data, boundary letters, options
    function prep3(s3, c) result(rc)
      type(mg_solver), intent(inout) :: s3
      integer, intent(in) :: c
      integer(c_int) :: rc
      integer :: i, f
      s3%ex_tol = ropt(ROPT_CTOL); s3%use_max = use_max; s3%nmax_exact = int(iopt(IOPT_NMAXEX))
      s3%precision = int(iopt(IOPT_PREC))
      ! initial guess of component c -> the solver's level-1 array
      u3 = mg_level_ptr(s3, 1, MG_BUF_U, cnt)
      if (on_device) then
        rc = ndsmk_d2d(u3, dptr_offset(pA, int(c - 1, c_size_t) * nb), nb); if (rc /= 0) return
      else
        rc = ndsmk_bg_wait(tick_up(c), zero_flag); if (rc /= 0) return
        if (zero_flag /= 0) then
          rc = ndsmk_fill0(u3, nb)
        else
          rc = ndsmk_d2d(u3, dptr_offset(ctx%dA, int(c - 1, c_size_t) * nb), nb)
        end if
        if (rc /= 0) return
      end if
      ! its Dirichlet data: A_t on the four tangential faces, in the reference's order (later writes
      ! win on shared edges, :647-650, :663-666, :679-682)
      do i = 1, 4
        f = face_order(i, c)
        if (host_faces) then
          rc = face_upload(u3, n3, f, merge(1, 2, face_t1(f) == c), fc(f)); if (rc /= 0) return
        else
          fac = 1.0_wp / (2.0_wp * dq(face_axis(f)))            ! Q4: the normal spacing
          rc = ndsmk_face_write(u3, n3, ctx%dchi, int(f - 1, c_int), int(c - 1, c_int), fac); if (rc /= 0) return
        end if
      end do
      bc3 = 'D'
      bc3(c) = 'N'; bc3(3 + c) = 'N'                        ! :655,:671,:687
      rc = mg_set_bcs(s3, bc3); if (rc /= 0) return
      s3%ms = merge(5, int(iopt(IOPT_MS)), c == 3)          ! Q2
      rc = mg_reset_info(s3)
    end function

    ! ... and what follows its solve: the result into A, the reference's warnings and outputs, its flux-balance
    ! field, its way home
    function finish3(s3, c, du_c, ncyc_c, ierr_c) result(rc)
      type(mg_solver), intent(inout) :: s3
      integer, intent(in) :: c, ncyc_c, ierr_c
      real(wp), intent(in) :: du_c
      integer(c_int) :: rc
      integer :: st3
      rc = mg_export_u(s3, dptr_offset(dAout, int(c - 1, c_size_t) * nb)); if (rc /= 0) return
      if (ierr_c /= 0) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
      if (mg_read_info(s3, sweeps, bad) == 0) then
        if (bad > 0) print *, "Warning: IOPT_NMAXEX exceeded. Coarse-mesh solution may not have converged"
      end if
      if (ierr_c /= 0) iopt(IOPT_FAIL3D) = ior(iopt(IOPT_FAIL3D), ishft(1_ik, c - 1))
      if (ncyc_c > 1 .or. c == 1) then
        iopt(IOPT_NCYC_OUT) = ncyc_c
        ropt(ROPT_DULAST) = du_c
      end if
      ! default order (:467-477): the flux-balance fields come before the curl, so this component is final
      ! once its own field is added - and goes home behind the next component's solve
      if (.not. late_balance) then
        rc = ndsmk_balance_component(dptr_offset(dAout, int(c - 1, c_size_t) * nb), n3, int(c - 1, c_int), ctx%dmesh, &
                                     dptr_offset(ctx%dmesh, off_y), dptr_offset(ctx%dmesh, off_z), phi, span)
        if (rc /= 0) return
        if (.not. on_device) then
          rc = ndsmk_bg_download(c_loc(hA(1, 1, 1, c)), dptr_offset(dAout, int(c - 1, c_size_t) * nb), nb, tick)
          if (rc /= 0) return
        end if
        ! B_z = d(A_y)/dx - d(A_x)/dy needs the two components that are final now: it is formed here and goes
        ! home behind the A_z solve as well (when B's device array exists already: not on the lean path)
        call get_environment_variable("NDSM_HIP_NO_EARLY_BZ", status=st3)      ! A/B testing: one curl at the end
        if (c == 2 .and. c_associated(dBout) .and. all(n3 >= 3) .and. st3 /= 0) then
          rc = ndsmk_curl_component(dAout, dBout, n3, dq, 2_c_int); if (rc /= 0) return
          if (.not. on_device) then
            rc = ndsmk_bg_download(c_loc(hB(1, 1, 1, 3)), dptr_offset(dBout, 2_c_size_t * nb), nb, tick)
            if (rc /= 0) return
          end if
          bz_done = .true.
        end if
      end if
    end function
  end function

  ! B.n of face f (1..6) from the host field (extract_bn, :699-743)
  subroutine face_gather(hB, n3, f, bn)
    real(wp), intent(inout) :: hB(:, :, :, :)
    integer(c_int32_t), intent(in) :: n3(3)
    integer, intent(in) :: f
    real(wp), intent(inout) :: bn(:, :)
    integer :: ax, lay
    ax = face_axis(f)
    lay = merge(int(n3(ax)), 1, face_upper(f))
    call face_copy(hB(:, :, :, ax), ax, lay, bn, to_face=.true.)
  end subroutine

  subroutine face_gather_flat(hB, n3, f, flat)
    real(wp), intent(inout) :: hB(:, :, :, :)
    integer(c_int32_t), intent(in) :: n3(3)
    integer, intent(in) :: f
    real(wp), intent(inout), target, contiguous :: flat(:)
    real(wp), pointer :: bn(:, :)
    bn(1:n3(face_t1(f)), 1:n3(face_t2(f))) => flat
    call face_gather(hB, n3, f, bn)
  end subroutine

  ! host face data (at1 / at2 of vecpot_faces) -> boundary plane f of the device array u: the
  ! NDSM_HIP_HOST_FACES path of vecpot_run (A/B testing of the device face phase)
  function face_upload(u3, n3, f, which, fd) result(rc)
    type(c_ptr), intent(in) :: u3
    integer(c_int32_t), intent(in) :: n3(3)
    integer, intent(in) :: f, which
    type(face_data), intent(in), target :: fd
    integer(c_int) :: rc
    type(c_ptr) :: tmp
    integer(c_size_t) :: nbf
    nbf = int(fd%n1, c_size_t) * int(fd%n2, c_size_t) * 8_c_size_t
    rc = ndsmk_alloc(tmp, nbf); if (rc /= 0) return
    if (which == 1) then
      rc = ndsmk_h2d(tmp, c_loc(fd%at1), nbf)
    else
      rc = ndsmk_h2d(tmp, c_loc(fd%at2), nbf)
    end if
    if (rc == 0) rc = ndsmk_face_put(u3, n3, int(f - 1, c_int), tmp)
    if (ndsmk_free(tmp) /= 0) continue
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

  ! 2-D trapezoid rule, weights 1 / 1/2 (edges) / 1/4 (corners) (:1070-1106).  The reference sums
  ! serially; the device kernel (faces.hip: face_flux_k) sums as a fixed tree - 1024 strided partial sums,
  ! a halving tree over each group of 64, the 16 group sums in order.  This host version walks the SAME
  ! tree, so the host face phase (the distributed driver's rank 0, NDSM_HIP_HOST_FACES) and the device
  ! face phase produce the same bits.
  function trapezoid(f, h1, h2) result(s)
    real(wp), intent(in) :: f(:, :), h1, h2
    real(wp) :: s, w
    real(wp) :: part(0:1023)
    integer :: a, b, n1, n2, p, n, t, o, l, q
    logical :: ea, eb
    n1 = size(f, 1); n2 = size(f, 2)
    n = n1 * n2
    part = 0
    do t = 0, min(1023, n - 1)
      do p = t, n - 1, 1024
        a = mod(p, n1); b = p / n1
        ea = (a == 0 .or. a == n1 - 1); eb = (b == 0 .or. b == n2 - 1)
        w = 1.0_wp
        if (ea .or. eb) w = 0.5_wp
        if (ea .and. eb) w = 0.25_wp
        part(t) = part(t) + w * f(a + 1, b + 1)
      end do
    end do
    do q = 0, 15
      o = 32
      do while (o > 0)
        do l = 0, o - 1
          part(64 * q + l) = part(64 * q + l) + part(64 * q + l + o)
        end do
        o = o / 2
      end do
    end do
    s = 0
    do q = 0, 15
      s = s + part(64 * q)
    end do
    s = s * (h1 * h2)
  end function

end module ndsmh_vecpot
