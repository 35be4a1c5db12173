! libndsm_hip - the C ABI (declared in include/ndsm_hip.h).
!
! Part 1 is the reference's complete dynamic symbol set, same names, same
! argument lists, same option-slot values: ndsm_python_wrapper.f90:56-158
! (ndsm_vector_solve) and :164-234 (twelve getters) - an unmodified ndsm.py
! (ndsm.py:136-207) can load this library in place of ndsmf.so.
! Part 2 is additive: a scalar Poisson entry and a persistent, device-resident
! solver handle (SURVEY 8b "additive exports", 8f-4).  Nothing in part 1 changes
! meaning because of part 2.
!
! Error behaviour: the reference returns 0/1 and STOPs the process on internal
! asserts.  Here nothing STOPs: device/runtime failures come back as return
! codes >= 9001 (ndsm_kernels.h) with the text on stderr and in
! ndsm_hip_last_error().  There is no CPU fallback.
module ndsmh_cabi

  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: error_unit
  use ndsmh_iface
  use ndsmh_grid, only: slab_t
  use ndsmh_mg
  use ndsmh_world
  use ndsmh_vecpot
  use ndsmh_wvecpot
  implicit none
  private

  interface
    function ndsmk_dist_unique_id(out128) bind(c, name="ndsmk_dist_unique_id") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: out128
      integer(c_int) :: rc
    end function
    function ndsmk_dist_init(rank, nranks, id128) bind(c, name="ndsmk_dist_init") result(rc)
      import :: c_ptr, c_int
      integer(c_int), value :: rank, nranks
      type(c_ptr), value :: id128
      integer(c_int) :: rc
    end function
    function ndsmk_dist_finalize() bind(c, name="ndsmk_dist_finalize") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
    function ndsmk_dist_info(rank, nranks) bind(c, name="ndsmk_dist_info") result(rc)
      import :: c_int
      integer(c_int), intent(out) :: rank, nranks
      integer(c_int) :: rc
    end function
    function ndsmk_dist_selftest(nelem) bind(c, name="ndsmk_dist_selftest") result(rc)
      import :: c_int
      integer(c_int), value :: nelem
      integer(c_int) :: rc
    end function
    function ndsmk_shutdown() bind(c, name="ndsmk_shutdown") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
  end interface

  interface
    function c_strlen(s) bind(c, name="strlen") result(n)
      import :: c_ptr, c_size_t
      type(c_ptr), value :: s
      integer(c_size_t) :: n
    end function
  end interface

contains

  ! wall clock in seconds (the reference uses OMP_GET_WTIME, ndsm_root.f90:521-536)
  function wall_seconds() result(t)
    real(c_double) :: t
    integer(c_int64_t) :: cnt, rate
    call system_clock(cnt, rate)
    t = real(cnt, c_double) / real(rate, c_double)
  end function

  subroutine report(where, rc)
    character(len=*), intent(in) :: where
    integer(c_int), intent(in) :: rc
    character(len=512) :: msg
    call fetch_error(msg)
    write (error_unit, '(A,I0)') "ERROR("//where//"):"//trim(msg)//":", rc
  end subroutine

  subroutine fetch_error(msg)
    character(len=*), intent(out) :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: cs(:)
    integer :: i, n
    msg = ""
    p = ndsmk_last_error()
    if (.not. c_associated(p)) return
    n = min(int(c_strlen(p)), len(msg))
    call c_f_pointer(p, cs, [n])
    do i = 1, n
      msg(i:i) = cs(i)
    end do
  end subroutine

  ! =====================================================================
  ! Part 1 - the reference's symbols
  ! =====================================================================

  ! ndsm_python_wrapper.f90:56-79.  nshape4 = [nx,ny,nz,3]; A: in = initial
  ! guess, out = vector potential; B: in = field whose normal boundary
  ! component is read, out = curl A (+ flux balance).
  function ndsm_vector_solve(nsize, nshape4, ioptc, ropt, x, y, z, A, B) bind(c, name="ndsm_vector_solve") &
      result(ierr)
    integer(c_size_t), value :: nsize
    integer(c_int), intent(in) :: nshape4(4)
    integer(c_int), intent(inout) :: ioptc(0:OPT_LEN - 1)
    real(c_double), intent(inout) :: ropt(0:OPT_LEN - 1)
    real(c_double), intent(in) :: x(nshape4(1)), y(nshape4(2)), z(nshape4(3))
    real(c_double), intent(inout), target :: A(nsize), B(nsize)
    integer(c_int) :: ierr
    integer(ik) :: iopt(0:OPT_LEN - 1)
    integer(c_int32_t) :: n3(3)
    real(c_double) :: t0
    real(c_double), pointer, contiguous :: A4(:, :, :, :), B4(:, :, :, :)
    integer(c_int) :: rc

    iopt = ioptc
    verbose = (iopt(IOPT_DEBUG) == 1)
    t0 = wall_seconds()
    n3 = nshape4(1:3)
    if (nshape4(4) /= 3 .or. int(nsize, ik) /= 3_ik * product(int(n3, ik))) then
      ioptc(IOPT_IERR) = 1
      ierr = 1
      return
    end if
    A4(1:n3(1), 1:n3(2), 1:n3(3), 1:3) => A
    B4(1:n3(1), 1:n3(2), 1:n3(3), 1:3) => B

    rc = vecpot_solve(n3, iopt, ropt, x, y, z, A4, B4)

    ropt(ROPT_TIM) = wall_seconds() - t0
    ioptc = int(iopt, c_int)
    if (rc /= 0) then
      call report("ndsm_vector_solve", rc)
      ioptc(IOPT_IERR) = rc
      ierr = rc
    else
      ierr = int(iopt(IOPT_IERR), c_int)
    end if
  end function

  ! The same call on a z-slab decomposition (additive; BASELINE config[4]): one process per GPU,
  ! every rank of the communicator (ndsm_hip_dist_init) calls it with the GLOBAL nshape4 and mesh
  ! and with ITS planes [z0, z1) of A and B - the split ndsm_hip_slab_plan reports - laid out
  ! (nx, ny, z1-z0, 3).  Options, return value and the contents of A and B as for
  ! ndsm_vector_solve; nranks == 1 is ndsm_vector_solve.
  function ndsm_hip_world_vector_solve(rank, nranks, nshape4, ioptc, ropt, x, y, z, A, B) &
      bind(c, name="ndsm_hip_world_vector_solve") result(ierr)
    integer(c_int), value :: rank, nranks
    integer(c_int), intent(in) :: nshape4(4)
    integer(c_int), intent(inout) :: ioptc(0:OPT_LEN - 1)
    real(c_double), intent(inout) :: ropt(0:OPT_LEN - 1)
    real(c_double), intent(in) :: x(nshape4(1)), y(nshape4(2)), z(nshape4(3))
    type(c_ptr), value :: A, B
    integer(c_int) :: ierr
    integer(ik) :: iopt(0:OPT_LEN - 1)
    integer(c_int32_t) :: n3(3)
    real(c_double) :: t0
    real(c_double), pointer, contiguous :: A4(:, :, :, :), B4(:, :, :, :)
    type(slab_t), allocatable :: plan(:)
    integer(c_int) :: rc
    integer :: nzl

    iopt = ioptc
    verbose = (iopt(IOPT_DEBUG) == 1)
    t0 = wall_seconds()
    n3 = nshape4(1:3)
    ierr = NDSMK_EARG
    if (nshape4(4) /= 3 .or. nranks < 1 .or. rank < 0 .or. rank >= nranks) return
    if (.not. c_associated(A) .or. .not. c_associated(B)) return
    rc = ndsmk_init(-1_c_int)
    if (rc == 0) then
      if (nranks == 1) then
        call c_f_pointer(A, A4, [int(n3(1)), int(n3(2)), int(n3(3)), 3])
        call c_f_pointer(B, B4, [int(n3(1)), int(n3(2)), int(n3(3)), 3])
        rc = vecpot_solve(n3, iopt, ropt, x, y, z, A4, B4)
      else if (any(n3 < 2)) then
        iopt(IOPT_IERR) = 1
      else
        rc = world_plan_only(n3, x, y, z, int(iopt(IOPT_NGRIDS)), int(nranks), plan)
        if (rc == 0) then
          nzl = plan(rank)%z1 - plan(rank)%z0
          call c_f_pointer(A, A4, [int(n3(1)), int(n3(2)), nzl, 3])
          call c_f_pointer(B, B4, [int(n3(1)), int(n3(2)), nzl, 3])
          rc = wvecpot_solve(n3, iopt, ropt, x, y, z, int(nranks), int(rank), A4, B4)
        end if
      end if
    end if
    ropt(ROPT_TIM) = wall_seconds() - t0
    ioptc = int(iopt, c_int)
    if (rc /= 0) then
      call report("ndsm_hip_world_vector_solve", rc)
      ioptc(IOPT_IERR) = rc
      ierr = rc
    else
      ierr = int(iopt(IOPT_IERR), c_int)
    end if
  end function

  ! ndsm_python_wrapper.f90:164-234 - slot indices, discovered at run time by ndsm.py:155-174
  function get_iopt_len() bind(c, name="get_iopt_len") result(v)
    integer(c_int) :: v
    v = OPT_LEN
  end function
  function get_iopt_ierr() bind(c, name="get_iopt_ierr") result(v)
    integer(c_int) :: v
    v = OPT_LEN            ! sic: the reference returns IOPT_LEN here (:170-174, quirk Q5)
  end function
  function get_iopt_ms() bind(c, name="get_iopt_ms") result(v)
    integer(c_int) :: v
    v = IOPT_MS
  end function
  function get_iopt_ncycles() bind(c, name="get_iopt_ncycles") result(v)
    integer(c_int) :: v
    v = IOPT_NCYCLES
  end function
  function get_iopt_debug() bind(c, name="get_iopt_debug") result(v)
    integer(c_int) :: v
    v = IOPT_DEBUG
  end function
  function get_iopt_dumax() bind(c, name="get_iopt_dumax") result(v)
    integer(c_int) :: v
    v = IOPT_DUMAX
  end function
  function get_iopt_iopt_nmaxex() bind(c, name="get_iopt_iopt_nmaxex") result(v)
    integer(c_int) :: v
    v = IOPT_NMAXEX
  end function
  function get_iopt_true() bind(c, name="get_iopt_true") result(v)
    integer(c_int) :: v
    v = 1
  end function
  function get_iopt_false() bind(c, name="get_iopt_false") result(v)
    integer(c_int) :: v
    v = 0
  end function
  function get_ropt_tim() bind(c, name="get_ropt_tim") result(v)
    integer(c_int) :: v
    v = ROPT_TIM
  end function
  function get_ropt_vtol() bind(c, name="get_ropt_vtol") result(v)
    integer(c_int) :: v
    v = ROPT_VTOL
  end function
  function get_ropt_ctol() bind(c, name="get_ropt_ctol") result(v)
    integer(c_int) :: v
    v = ROPT_CTOL
  end function

  ! =====================================================================
  ! Part 2 - additive exports
  ! =====================================================================

  function get_iopt_fail3d() bind(c, name="get_iopt_fail3d") result(v)
    integer(c_int) :: v
    v = IOPT_FAIL3D
  end function
  function get_iopt_ngrids() bind(c, name="get_iopt_ngrids") result(v)
    integer(c_int) :: v
    v = IOPT_NGRIDS
  end function
  function get_iopt_prec() bind(c, name="get_iopt_prec") result(v)
    integer(c_int) :: v
    v = IOPT_PREC
  end function
  function get_iopt_ncyc_out() bind(c, name="get_iopt_ncyc_out") result(v)
    integer(c_int) :: v
    v = IOPT_NCYC_OUT
  end function
  function get_ropt_dulast() bind(c, name="get_ropt_dulast") result(v)
    integer(c_int) :: v
    v = ROPT_DULAST
  end function

  function ndsm_hip_device_count() bind(c, name="ndsm_hip_device_count") result(n)
    integer(c_int) :: n
    n = ndsmk_device_count()
  end function

  function ndsm_hip_init(device) bind(c, name="ndsm_hip_init") result(rc)
    integer(c_int), value :: device
    integer(c_int) :: rc
    rc = ndsmk_init(device)
  end function

  ! releases everything the library holds on the device outside the caller's handles (streams, events,
  ! metric scratch, the cached vector-potential hierarchy, a live RCCL communicator); destroy solver /
  ! world handles first.  ndsm_hip_init (or any solve) brings the runtime up again, on any device.
  function ndsm_hip_shutdown() bind(c, name="ndsm_hip_shutdown") result(rc)
    integer(c_int) :: rc
    rc = ndsmk_shutdown()
  end function

  subroutine ndsm_hip_last_error(buf, n) bind(c, name="ndsm_hip_last_error")
    integer(c_int), value :: n
    character(kind=c_char), intent(out) :: buf(n)
    character(len=512) :: msg
    integer :: i, m
    call fetch_error(msg)
    m = min(len_trim(msg), n - 1)
    do i = 1, m
      buf(i) = msg(i:i)
    end do
    buf(m + 1) = c_null_char
  end subroutine

  function ndsm_hip_sync() bind(c, name="ndsm_hip_sync") result(rc)
    integer(c_int) :: rc
    rc = ndsmk_sync()
  end function
  function ndsm_hip_timer_start() bind(c, name="ndsm_hip_timer_start") result(rc)
    integer(c_int) :: rc
    rc = ndsmk_timer_start()
  end function
  function ndsm_hip_timer_stop(ms) bind(c, name="ndsm_hip_timer_stop") result(rc)
    real(c_double), intent(out) :: ms
    integer(c_int) :: rc
    rc = ndsmk_timer_stop(ms)
  end function

  ! Scalar Poisson problem laplace(u) = rhs, host buffers.  bcs = 2*ndim
  ! letters (lower faces, then upper faces).  Options in the same slots as
  ! ndsm_vector_solve; rhs may be NULL (zero); hist may be NULL.
  function ndsm_hip_poisson_solve(ndim, nshape, x, y, z, bcs, ioptc, ropt, u, rhs, hist, hist_len) &
      bind(c, name="ndsm_hip_poisson_solve") result(ierr)
    integer(c_int), value :: ndim, hist_len
    integer(c_int), intent(in) :: nshape(ndim)
    type(c_ptr), value :: x, y, z, u, rhs, hist
    character(kind=c_char), intent(in) :: bcs(2 * ndim)
    integer(c_int), intent(inout) :: ioptc(0:OPT_LEN - 1)
    real(c_double), intent(inout) :: ropt(0:OPT_LEN - 1)
    integer(c_int) :: ierr
    integer(c_int32_t) :: n3(3)
    real(c_double), pointer :: qx(:), qy(:), qz(:), hh(:)
    real(c_double), target :: dummy(2)
    character(len=1) :: bc(6)
    real(c_double) :: du_last, t0
    integer :: ncyc, ie, d
    integer(c_int) :: rc

    ierr = NDSMK_EARG
    if (ndim /= 2 .and. ndim /= 3) return
    t0 = wall_seconds()
    n3 = 1
    n3(1:ndim) = nshape(1:ndim)
    dummy = [0.0_wp, 1.0_wp]
    call c_f_pointer(x, qx, [n3(1)])
    call c_f_pointer(y, qy, [n3(2)])
    if (ndim == 3) then
      call c_f_pointer(z, qz, [n3(3)])
    else
      qz => dummy
    end if
    bc = 'N'
    do d = 1, 2 * ndim
      bc(d) = bcs(d)
    end do
    verbose = (ioptc(IOPT_DEBUG) == 1)
    if (c_associated(hist) .and. hist_len > 0) then
      call c_f_pointer(hist, hh, [hist_len])
      rc = poisson_solve(int(ndim), n3, qx, qy, qz, bc, int(ioptc(IOPT_MS)), ropt(ROPT_CTOL), &
                         ioptc(IOPT_DUMAX) == 1, int(ioptc(IOPT_NMAXEX)), int(ioptc(IOPT_NGRIDS)), &
                         ropt(ROPT_VTOL), int(ioptc(IOPT_NCYCLES)), u, rhs, du_last, ncyc, ie, hh, &
                         precision=int(ioptc(IOPT_PREC)))
    else
      rc = poisson_solve(int(ndim), n3, qx, qy, qz, bc, int(ioptc(IOPT_MS)), ropt(ROPT_CTOL), &
                         ioptc(IOPT_DUMAX) == 1, int(ioptc(IOPT_NMAXEX)), int(ioptc(IOPT_NGRIDS)), &
                         ropt(ROPT_VTOL), int(ioptc(IOPT_NCYCLES)), u, rhs, du_last, ncyc, ie, &
                         precision=int(ioptc(IOPT_PREC)))
    end if
    ropt(ROPT_TIM) = wall_seconds() - t0
    if (rc /= 0) then
      call report("ndsm_hip_poisson_solve", rc)
      ierr = rc
      return
    end if
    ioptc(IOPT_IERR) = ie
    ioptc(IOPT_NCYC_OUT) = ncyc
    ropt(ROPT_DULAST) = du_last
    ierr = ie
  end function

  ! ---- persistent device-resident solver ------------------------------

  function ndsm_hip_mg_create(ndim, nshape, x, y, z, bcs, ngrids, ms, ex_tol, du_max, nmax_exact, handle) &
      bind(c, name="ndsm_hip_mg_create") result(rc)
    integer(c_int), value :: ndim, ngrids, ms, du_max, nmax_exact
    real(c_double), value :: ex_tol
    integer(c_int), intent(in) :: nshape(ndim)
    type(c_ptr), value :: x, y, z
    character(kind=c_char), intent(in) :: bcs(2 * ndim)
    type(c_ptr), intent(out) :: handle
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    integer(c_int32_t) :: n3(3)
    real(c_double), pointer :: qx(:), qy(:), qz(:)
    real(c_double), target :: dummy(2)
    character(len=1) :: bc(6)
    integer :: d

    handle = c_null_ptr
    rc = NDSMK_EARG
    if (ndim /= 2 .and. ndim /= 3) return
    n3 = 1
    n3(1:ndim) = nshape(1:ndim)
    dummy = [0.0_wp, 1.0_wp]
    call c_f_pointer(x, qx, [n3(1)])
    call c_f_pointer(y, qy, [n3(2)])
    if (ndim == 3) then
      call c_f_pointer(z, qz, [n3(3)])
    else
      qz => dummy
    end if
    bc = 'N'
    do d = 1, 2 * ndim
      bc(d) = bcs(d)
    end do
    allocate (s)
    rc = mg_create(s, int(ndim), n3, qx, qy, qz, bc, int(ngrids))
    if (rc /= 0) then
      call mg_destroy(s)
      deallocate (s)
      return
    end if
    s%ms = ms; s%ex_tol = ex_tol; s%use_max = (du_max == 1); s%nmax_exact = nmax_exact
    handle = c_loc(s)
  end function

  function ndsm_hip_mg_destroy(handle) bind(c, name="ndsm_hip_mg_destroy") result(rc)
    type(c_ptr), value :: handle
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    rc = 0
    if (.not. c_associated(handle)) return
    call c_f_pointer(handle, s)
    call mg_destroy(s)
    deallocate (s)
  end function

  ! shapes(3, ngrids) in Fortran order; pass ngrids_cap = room in `shapes`
  function ndsm_hip_mg_levels(handle, ngrids_cap, shapes) bind(c, name="ndsm_hip_mg_levels") result(ng)
    type(c_ptr), value :: handle
    integer(c_int), value :: ngrids_cap
    integer(c_int), intent(out) :: shapes(3, ngrids_cap)
    integer(c_int) :: ng
    type(mg_solver), pointer :: s
    integer :: l
    call c_f_pointer(handle, s)
    ng = s%ngrids
    do l = 1, min(s%ngrids, int(ngrids_cap))
      shapes(:, l) = s%lev(l)%n
    end do
  end function

  function ndsm_hip_mg_set_ms(handle, ms) bind(c, name="ndsm_hip_mg_set_ms") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: ms
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    call c_f_pointer(handle, s)
    s%ms = ms
    rc = 0
  end function

  ! 0: fp64 (reference arithmetic); 1: mixed precision where level 1 is large enough; 2: mixed
  ! wherever the fp32 kernels cover level 1.  Returns 1 if ndsm_hip_mg_solve will run mixed, else 0.
  function ndsm_hip_mg_set_precision(handle, mode) bind(c, name="ndsm_hip_mg_set_precision") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: mode
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    call c_f_pointer(handle, s)
    rc = -1
    if (mode < 0 .or. mode > 2) return
    s%precision = mode
    rc = merge(1_c_int, 0_c_int, mg_mixed_applies(s))
  end function

  ! which: 0 = u, 1 = rhs, 2 = residual scratch (level-1 sized, valid after op RESIDUAL)
  function ndsm_hip_mg_upload(handle, level, which, host) bind(c, name="ndsm_hip_mg_upload") result(rc)
    type(c_ptr), value :: handle, host
    integer(c_int), value :: level, which
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    type(c_ptr) :: d
    integer(ik) :: n
    call c_f_pointer(handle, s)
    d = mg_level_ptr(s, int(level), int(which), n)
    rc = NDSMK_EARG
    if (.not. c_associated(d)) return
    rc = ndsmk_h2d(d, host, int(n, c_size_t) * 8_c_size_t)
    if (level == 1 .and. which == MG_BUF_RHS) call mg_mark_rhs_set(s)
  end function

  ! declare the level-1 right-hand side identically zero (Laplace problem): the
  ! kernels then never read it; results are bit-identical
  function ndsm_hip_mg_zero_rhs(handle) bind(c, name="ndsm_hip_mg_zero_rhs") result(rc)
    type(c_ptr), value :: handle
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    call c_f_pointer(handle, s)
    rc = mg_zero_rhs(s)
  end function

  function ndsm_hip_mg_download(handle, level, which, host) bind(c, name="ndsm_hip_mg_download") result(rc)
    type(c_ptr), value :: handle, host
    integer(c_int), value :: level, which
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    type(c_ptr) :: d
    integer(ik) :: n
    call c_f_pointer(handle, s)
    d = mg_level_ptr(s, int(level), int(which), n)
    rc = NDSMK_EARG
    if (.not. c_associated(d)) return
    rc = ndsmk_d2h(host, d, int(n, c_size_t) * 8_c_size_t)
  end function

  ! op: 0 relax(count sweeps) 1 residual 2 restrict(level->level+1) 3 prolong-add
  ! (level+1->level) 4 coarsest solve 5 relax via colour kernels 6 relax via fused kernel
  function ndsm_hip_mg_op(handle, op, level, count) bind(c, name="ndsm_hip_mg_op") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: op, level, count
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    call c_f_pointer(handle, s)
    rc = mg_op(s, int(op), int(level), int(count))
  end function

  ! enqueue ncycles V-cycles; returns without waiting for the GPU
  function ndsm_hip_mg_vcycle(handle, ncycles) bind(c, name="ndsm_hip_mg_vcycle") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: ncycles
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    integer :: i
    call c_f_pointer(handle, s)
    rc = 0
    do i = 1, ncycles
      rc = mg_vcycle(s)
      if (rc /= 0) return
    end do
  end function

  ! V-cycles to vc_tol on the resident problem: 0 converged, 1 not, >= 9001 error
  function ndsm_hip_mg_solve(handle, vc_tol, nmax, du_last, ncycles, hist, hist_len) &
      bind(c, name="ndsm_hip_mg_solve") result(ierr)
    type(c_ptr), value :: handle, hist
    real(c_double), value :: vc_tol
    integer(c_int), value :: nmax, hist_len
    real(c_double), intent(out) :: du_last
    integer(c_int), intent(out) :: ncycles
    integer(c_int) :: ierr
    type(mg_solver), pointer :: s
    real(c_double), pointer :: hh(:)
    integer :: nc, ie
    integer(c_int) :: rc
    call c_f_pointer(handle, s)
    if (c_associated(hist) .and. hist_len > 0) then
      call c_f_pointer(hist, hh, [hist_len])
      rc = mg_solve(s, vc_tol, int(nmax), du_last, nc, ie, hh)
    else
      rc = mg_solve(s, vc_tol, int(nmax), du_last, nc, ie)
    end if
    ncycles = nc
    ierr = merge(rc, int(ie, c_int), rc /= 0)
  end function

  function ndsm_hip_mg_info(handle, sweeps, unconverged) bind(c, name="ndsm_hip_mg_info") result(rc)
    type(c_ptr), value :: handle
    integer(c_int64_t), intent(out) :: sweeps, unconverged
    integer(c_int) :: rc
    type(mg_solver), pointer :: s
    call c_f_pointer(handle, s)
    rc = mg_read_info(s, sweeps, unconverged)
  end function

  ! ---- device memory for callers without a HIP binding of their own (ctypes) ----
  function ndsm_hip_device_alloc(bytes, p) bind(c, name="ndsm_hip_device_alloc") result(rc)
    integer(c_size_t), value :: bytes
    type(c_ptr), intent(out) :: p
    integer(c_int) :: rc
    p = c_null_ptr
    rc = ndsmk_init(-1_c_int)
    if (rc == 0) rc = ndsmk_alloc(p, bytes)
  end function
  function ndsm_hip_device_free(p) bind(c, name="ndsm_hip_device_free") result(rc)
    type(c_ptr), value :: p
    integer(c_int) :: rc
    rc = ndsmk_free(p)
  end function
  function ndsm_hip_memcpy_h2d(d_dst, h_src, bytes) bind(c, name="ndsm_hip_memcpy_h2d") result(rc)
    type(c_ptr), value :: d_dst, h_src
    integer(c_size_t), value :: bytes
    integer(c_int) :: rc
    rc = ndsmk_h2d(d_dst, h_src, bytes)
  end function
  function ndsm_hip_memcpy_d2h(h_dst, d_src, bytes) bind(c, name="ndsm_hip_memcpy_d2h") result(rc)
    type(c_ptr), value :: h_dst, d_src
    integer(c_size_t), value :: bytes
    integer(c_int) :: rc
    rc = ndsmk_d2h(h_dst, d_src, bytes)
  end function

  ! ---- persistent vector-potential handle (SURVEY 8f-4) ------------------
  ! Everything that depends on the grid alone - the 3-D hierarchy and its transfer tables, the three
  ! 2-D face hierarchies, device arrays for A, B and the faces - lives in the handle; a solve moves
  ! boundary data in and results out.  (ndsm_vector_solve keeps one such context internally, keyed by
  ! shape and mesh.)
  function ndsm_hip_vecpot_create(nshape4, x, y, z, ngrids, handle) bind(c, name="ndsm_hip_vecpot_create") result(rc)
    integer(c_int), intent(in) :: nshape4(4)
    type(c_ptr), value :: x, y, z
    integer(c_int), value :: ngrids
    type(c_ptr), intent(out) :: handle
    integer(c_int) :: rc
    type(vecpot_ctx), pointer :: ctx
    real(c_double), pointer :: qx(:), qy(:), qz(:)
    integer(c_int32_t) :: n3(3)
    handle = c_null_ptr
    rc = NDSMK_EARG
    n3 = nshape4(1:3)
    if (nshape4(4) /= 3 .or. any(n3 < 3)) return
    if (.not. (c_associated(x) .and. c_associated(y) .and. c_associated(z))) return
    call c_f_pointer(x, qx, [n3(1)])
    call c_f_pointer(y, qy, [n3(2)])
    call c_f_pointer(z, qz, [n3(3)])
    allocate (ctx)
    rc = vecpot_ctx_create(ctx, n3, qx, qy, qz, int(ngrids))
    if (rc /= 0) then
      call report("ndsm_hip_vecpot_create", rc)
      call vecpot_ctx_destroy(ctx)
      deallocate (ctx)
      return
    end if
    handle = c_loc(ctx)
  end function

  function ndsm_hip_vecpot_destroy(handle) bind(c, name="ndsm_hip_vecpot_destroy") result(rc)
    type(c_ptr), value :: handle
    integer(c_int) :: rc
    type(vecpot_ctx), pointer :: ctx
    rc = 0
    if (.not. c_associated(handle)) return
    call c_f_pointer(handle, ctx)
    call vecpot_ctx_destroy(ctx)
    deallocate (ctx)
  end function

  ! A, B: HOST arrays (nx,ny,nz,3), contents and options exactly as for ndsm_vector_solve
  function ndsm_hip_vecpot_solve(handle, ioptc, ropt, A, B) bind(c, name="ndsm_hip_vecpot_solve") result(ierr)
    type(c_ptr), value :: handle, A, B
    integer(c_int), intent(inout) :: ioptc(0:OPT_LEN - 1)
    real(c_double), intent(inout) :: ropt(0:OPT_LEN - 1)
    integer(c_int) :: ierr
    ierr = vecpot_handle_solve(handle, ioptc, ropt, A, B, .false., "ndsm_hip_vecpot_solve")
  end function

  ! A, B: DEVICE arrays (nx,ny,nz,3) of the GPU the library runs on (e.g. torch tensors' data_ptr()):
  ! nothing but six fluxes and the per-cycle 16-byte convergence read-backs crosses PCIe.  The call
  ! returns when the results are complete in A and B (the library stream is drained).
  function ndsm_hip_vecpot_solve_device(handle, ioptc, ropt, dA, dB) bind(c, name="ndsm_hip_vecpot_solve_device") &
      result(ierr)
    type(c_ptr), value :: handle, dA, dB
    integer(c_int), intent(inout) :: ioptc(0:OPT_LEN - 1)
    real(c_double), intent(inout) :: ropt(0:OPT_LEN - 1)
    integer(c_int) :: ierr
    ierr = vecpot_handle_solve(handle, ioptc, ropt, dA, dB, .true., "ndsm_hip_vecpot_solve_device")
  end function

  function vecpot_handle_solve(handle, ioptc, ropt, A, B, on_device, who) result(ierr)
    type(c_ptr), intent(in) :: handle, A, B
    integer(c_int), intent(inout) :: ioptc(0:OPT_LEN - 1)
    real(c_double), intent(inout) :: ropt(0:OPT_LEN - 1)
    logical, intent(in) :: on_device
    character(len=*), intent(in) :: who
    integer(c_int) :: ierr
    type(vecpot_ctx), pointer :: ctx
    integer(ik) :: iopt(0:OPT_LEN - 1)
    real(c_double) :: t0
    integer(c_int) :: rc
    ierr = NDSMK_EARG
    if (.not. (c_associated(handle) .and. c_associated(A) .and. c_associated(B))) return
    call c_f_pointer(handle, ctx)
    if (.not. ctx%live) return
    iopt = ioptc
    verbose = (iopt(IOPT_DEBUG) == 1)
    t0 = wall_seconds()
    rc = NDSMK_EARG
    if (int(iopt(IOPT_NGRIDS)) == ctx%ngr) rc = vecpot_run(ctx, iopt, ropt, A, B, on_device)
    ropt(ROPT_TIM) = wall_seconds() - t0
    ioptc = int(iopt, c_int)
    if (rc /= 0) then
      call report(who, rc)
      ioptc(IOPT_IERR) = rc
      ierr = rc
    else
      ierr = int(iopt(IOPT_IERR), c_int)
    end if
  end function

  ! ---- z-slab decomposition over GPUs (SURVEY 8e) -----------------------

  ! rank 0 creates the 128-byte RCCL id; the launcher hands it to every rank
  function ndsm_hip_dist_unique_id(id128) bind(c, name="ndsm_hip_dist_unique_id") result(rc)
    type(c_ptr), value :: id128
    integer(c_int) :: rc
    rc = ndsmk_dist_unique_id(id128)
  end function

  function ndsm_hip_dist_init(rank, nranks, id128) bind(c, name="ndsm_hip_dist_init") result(rc)
    integer(c_int), value :: rank, nranks
    type(c_ptr), value :: id128
    integer(c_int) :: rc
    rc = ndsmk_dist_init(rank, nranks, id128)
  end function

  ! drains the library streams and destroys the communicator (collective: every rank calls it)
  function ndsm_hip_dist_finalize() bind(c, name="ndsm_hip_dist_finalize") result(rc)
    integer(c_int) :: rc
    rc = ndsmk_dist_finalize()
  end function

  ! rank and size as the RCCL communicator reports them (ncclCommUserRank / ncclCommCount);
  ! nranks = 0: no communicator is up
  function ndsm_hip_dist_info(rank, nranks) bind(c, name="ndsm_hip_dist_info") result(rc)
    integer(c_int), intent(out) :: rank, nranks
    integer(c_int) :: rc
    rc = ndsmk_dist_info(rank, nranks)
  end function

  ! transport self-test on the live communicator (collective): self send/recv of nelem doubles through
  ! the grouped ncclSend/ncclRecv pair of a halo exchange, on the main and on the communication stream,
  ! and the 2-value all-reduce; 0 = every byte arrived
  function ndsm_hip_dist_selftest(nelem) bind(c, name="ndsm_hip_dist_selftest") result(rc)
    integer(c_int), value :: nelem
    integer(c_int) :: rc
    rc = ndsmk_dist_selftest(nelem)
  end function

  ! The slab plan every rank derives (pure host arithmetic, no GPU needed).
  ! out(12, nranks): rank, z0, z1, g, nloc, k0, ck0, ck1, pk0, pk1, cb0, cb1
  function ndsm_hip_slab_plan(nshape, x, y, z, ngrids, nranks, out) bind(c, name="ndsm_hip_slab_plan") result(rc)
    integer(c_int), intent(in) :: nshape(3)
    integer(c_int), value :: ngrids, nranks
    type(c_ptr), value :: x, y, z
    integer(c_int), intent(out) :: out(12, nranks)
    integer(c_int) :: rc
    real(c_double), pointer :: qx(:), qy(:), qz(:)
    type(slab_t), allocatable :: plan(:)
    integer(c_int32_t) :: n3(3)
    integer :: r
    n3 = nshape
    call c_f_pointer(x, qx, [n3(1)])
    call c_f_pointer(y, qy, [n3(2)])
    call c_f_pointer(z, qz, [n3(3)])
    rc = world_plan_only(n3, qx, qy, qz, int(ngrids), int(nranks), plan)
    if (rc /= 0) return
    do r = 0, nranks - 1
      out(:, r + 1) = [plan(r)%rank, plan(r)%z0, plan(r)%z1, plan(r)%g, plan(r)%nloc, plan(r)%k0, plan(r)%ck0, &
                       plan(r)%ck1, plan(r)%pk0, plan(r)%pk1, plan(r)%cb0, plan(r)%cb1]
    end do
  end function

  ! rank >= 0: this process holds slab `rank` (RCCL transport, after ndsm_hip_dist_init);
  ! rank < 0 : loop-back world, all nranks slabs on this GPU (verification)
  function ndsm_hip_world_create(nshape, x, y, z, bcs, ngrids, ms, ex_tol, du_max, nmax_exact, nranks, rank, handle) &
      bind(c, name="ndsm_hip_world_create") result(rc)
    integer(c_int), intent(in) :: nshape(3)
    integer(c_int), value :: ngrids, ms, du_max, nmax_exact, nranks, rank
    real(c_double), value :: ex_tol
    type(c_ptr), value :: x, y, z
    character(kind=c_char), intent(in) :: bcs(6)
    type(c_ptr), intent(out) :: handle
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    real(c_double), pointer :: qx(:), qy(:), qz(:)
    integer(c_int32_t) :: n3(3)
    character(len=1) :: bc(6)
    integer :: d, i
    handle = c_null_ptr
    n3 = nshape
    call c_f_pointer(x, qx, [n3(1)])
    call c_f_pointer(y, qy, [n3(2)])
    call c_f_pointer(z, qz, [n3(3)])
    do d = 1, 6
      bc(d) = bcs(d)
    end do
    rc = ndsmk_init(-1_c_int)
    if (rc /= 0) return
    allocate (w)
    rc = world_create(w, n3, qx, qy, qz, bc, int(ngrids), int(nranks), int(rank))
    if (rc /= 0) then
      call world_destroy(w)
      deallocate (w)
      return
    end if
    call world_set_params(w, int(ms), ex_tol, du_max == 1, int(nmax_exact))
    handle = c_loc(w)
  end function

  function ndsm_hip_world_destroy(handle) bind(c, name="ndsm_hip_world_destroy") result(rc)
    type(c_ptr), value :: handle
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    rc = 0
    if (.not. c_associated(handle)) return
    call c_f_pointer(handle, w)
    call world_destroy(w)
    deallocate (w)
  end function

  function ndsm_hip_world_dist_levels(handle) bind(c, name="ndsm_hip_world_dist_levels") result(n)
    type(c_ptr), value :: handle
    integer(c_int) :: n
    type(mg_world), pointer :: w
    call c_f_pointer(handle, w)
    n = world_dist_levels(w)
  end function

  function ndsm_hip_world_nlocal(handle) bind(c, name="ndsm_hip_world_nlocal") result(n)
    type(c_ptr), value :: handle
    integer(c_int) :: n
    type(mg_world), pointer :: w
    call c_f_pointer(handle, w)
    n = w%nlocal
  end function

  ! info(12) of local slab ilocal (1-based): same fields as ndsm_hip_slab_plan
  function ndsm_hip_world_slab(handle, ilocal, info) bind(c, name="ndsm_hip_world_slab") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: ilocal
    integer(c_int), intent(out) :: info(12)
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    call c_f_pointer(handle, w)
    rc = NDSMK_EARG
    if (ilocal < 1 .or. ilocal > w%nlocal) return
    associate (p => w%loc(ilocal)%sl)
      info = [p%rank, p%z0, p%z1, p%g, p%nloc, p%k0, p%ck0, p%ck1, p%pk0, p%pk1, p%cb0, p%cb1]
    end associate
    rc = 0
  end function

  ! host points at global plane gz0 of an (nx,ny,*) array holding nplanes planes
  function ndsm_hip_world_upload(handle, ilocal, which, host, gz0, nplanes) bind(c, name="ndsm_hip_world_upload") &
      result(rc)
    type(c_ptr), value :: handle, host
    integer(c_int), value :: ilocal, which, gz0, nplanes
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    call c_f_pointer(handle, w)
    rc = world_upload(w, int(ilocal), int(which), host, int(gz0), int(nplanes))
  end function

  ! host receives the slab's owned planes [z0, z1)
  function ndsm_hip_world_download(handle, ilocal, which, host) bind(c, name="ndsm_hip_world_download") result(rc)
    type(c_ptr), value :: handle, host
    integer(c_int), value :: ilocal, which
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    call c_f_pointer(handle, w)
    rc = world_download(w, int(ilocal), int(which), host)
  end function

  function ndsm_hip_world_relax(handle, nsweeps) bind(c, name="ndsm_hip_world_relax") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: nsweeps
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    call c_f_pointer(handle, w)
    rc = world_relax(w, int(nsweeps))
  end function

  ! mode as ndsm_hip_mg_set_precision (0 fp64; /= 0 mixed: fp64 residual, fp32 correction V-cycle on the
  ! level-1 slabs).  Returns 1 if world_solve will run mixed, 0 if the fp64 path stays (a slab out of the
  ! fp32 kernels' reach: odd nx, < 64 x 16 points per plane), < 0 on a bad handle.
  function ndsm_hip_world_set_precision(handle, mode) bind(c, name="ndsm_hip_world_set_precision") result(on)
    type(c_ptr), value :: handle
    integer(c_int), value :: mode
    integer(c_int) :: on
    type(mg_world), pointer :: w
    on = -1
    if (.not. c_associated(handle) .or. mode < 0 .or. mode > 2) return
    call c_f_pointer(handle, w)
    on = merge(1_c_int, 0_c_int, world_set_precision(w, int(mode)))
  end function

  function ndsm_hip_world_zero_rhs(handle) bind(c, name="ndsm_hip_world_zero_rhs") result(rc)
    type(c_ptr), value :: handle
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    integer :: i
    call c_f_pointer(handle, w)
    rc = 0
    do i = 1, w%nlocal
      rc = mg_zero_rhs(w%loc(i)); if (rc /= 0) return
    end do
  end function

  function ndsm_hip_bound_libs(buf, len) bind(c, name="ndsm_hip_bound_libs") result(rc)
    type(c_ptr), value :: buf
    integer(c_int), value :: len
    integer(c_int) :: rc
    interface
      function ndsmk_bound_libs(buf, len) bind(c, name="ndsmk_bound_libs") result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: buf
        integer(c_int), value :: len
        integer(c_int) :: rc
      end function
    end interface
    rc = ndsmk_bound_libs(buf, len)
  end function

  ! development hook (tests, tuning): tile configuration of the fused smoother, the five values of
  ! NDSM_FUSED_CFG (smooth_fused.hip) - big = 1 runs the tiles of >= 64 M-point levels on any level
  function ndsm_hip_debug_fused_cfg(two, one, res, work_items, big) bind(c, name="ndsm_hip_debug_fused_cfg") result(rc)
    integer(c_int), value :: two, one, res, work_items, big
    integer(c_int) :: rc
    interface
      function ndsmk_debug_fused_cfg(two, one, res, work_items, big) bind(c, name="ndsmk_debug_fused_cfg") result(rc)
        import :: c_int
        integer(c_int), value :: two, one, res, work_items, big
        integer(c_int) :: rc
      end function
    end interface
    rc = ndsmk_debug_fused_cfg(two, one, res, work_items, big)
  end function

  ! development hook: 0 = the levels of the V-cycle's tail run kernel by kernel even where the single-launch
  ! form (tail.hip) covers them, 1 = default
  function ndsm_hip_debug_tail(on) bind(c, name="ndsm_hip_debug_tail") result(rc)
    integer(c_int), value :: on
    integer(c_int) :: rc
    rc = ndsmk_debug_tail(on)
  end function

  function ndsm_hip_world_vcycle(handle, ncycles) bind(c, name="ndsm_hip_world_vcycle") result(rc)
    type(c_ptr), value :: handle
    integer(c_int), value :: ncycles
    integer(c_int) :: rc
    type(mg_world), pointer :: w
    integer :: i
    call c_f_pointer(handle, w)
    rc = 0
    do i = 1, ncycles
      rc = world_vcycle(w)
      if (rc /= 0) return
    end do
  end function

  function ndsm_hip_world_solve(handle, vc_tol, nmax, du_last, ncycles, hist, hist_len) &
      bind(c, name="ndsm_hip_world_solve") result(ierr)
    type(c_ptr), value :: handle, hist
    real(c_double), value :: vc_tol
    integer(c_int), value :: nmax, hist_len
    real(c_double), intent(out) :: du_last
    integer(c_int), intent(out) :: ncycles
    integer(c_int) :: ierr
    type(mg_world), pointer :: w
    real(c_double), pointer :: hh(:)
    integer :: nc, ie
    integer(c_int) :: rc
    call c_f_pointer(handle, w)
    if (c_associated(hist) .and. hist_len > 0) then
      call c_f_pointer(hist, hh, [hist_len])
      rc = world_solve(w, vc_tol, int(nmax), du_last, nc, ie, hh)
    else
      rc = world_solve(w, vc_tol, int(nmax), du_last, nc, ie)
    end if
    ncycles = nc
    ierr = merge(rc, int(ie, c_int), rc /= 0)
  end function

end module ndsmh_cabi
