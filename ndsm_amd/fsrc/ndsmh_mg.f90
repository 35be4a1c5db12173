! libndsm_hip - device-resident geometric multigrid solver: the V-cycle driver
! loop in Fortran 2003, every grid operation a HIP kernel.
!
! Reference path being replaced (all on the CPU there):
!   solve_poisson_bvp          ndsm_poisson.f90:63-155      -> mg_solve
!   v_cycle                    ndsm_multigrid_core.f90:341-377 -> mg_vcycle
!   fine_to_coarse/coarse_to_fine            :482-560 / :593-684
!   solve_exact                              :728-800
!   update_u                                 :1077-1122
!   new_mg_handle/delete_mg_handle           :165-329      -> mg_create / mg_destroy
!
! Differences in mechanism, not in arithmetic:
!   * all level arrays, the residual scratch and the transfer tables are
!     allocated once per solver in HBM and stay there; the reference allocates,
!     zeroes and frees u(l), rhs(l), r_f and cor_f inside every V-cycle
!     (:528,:542,:557,:647-675).
!   * a V-cycle enqueues kernels on one HIP stream and never synchronises; the
!     only host round trip per cycle is the 16-byte convergence metric.
module ndsmh_mg

  use, intrinsic :: iso_c_binding
  use ndsmh_iface
  use ndsmh_grid
  implicit none
  private

  public :: mg_solver, mg_create, mg_destroy, mg_vcycle, mg_solve
  public :: mg_solve_lanes
  public :: mg_mixed_applies, mg_relax_window, mg_relax_res_window, mg_swap_u, mg_window_prolong_ok, mg_window_metric_ok
  public :: mg_set_u, mg_set_rhs, mg_get_u, mg_zero_rhs, mg_level_ptr, mg_op, mg_read_info
  public :: mg_mark_rhs_set
  public :: mg_set_bcs, mg_export_u, mg_reset_info, mg_vcycle_from, mg_slab_restrict, mg_slab_prolong
  public :: mg_slab_restrict_f32, mg_slab_prolong_f32, mg_mixed_slab_ok, rhs_of
  public :: MG_BUF_U, MG_BUF_RHS, MG_BUF_R
  public :: MG_OP_RELAX_LAST
  public :: MG_OP_RELAX, MG_OP_RESIDUAL, MG_OP_RESTRICT, MG_OP_PROLONG, MG_OP_EXACT, MG_OP_RELAX_COLOR, &
            MG_OP_RELAX_FUSED, MG_OP_RESREST, MG_OP_RELAX_RES, MG_OP_RELAX_RES_FUSED

  integer, parameter :: MG_BUF_U = 0, MG_BUF_RHS = 1, MG_BUF_R = 2
  integer, parameter :: MG_OP_RELAX = 0, MG_OP_RESIDUAL = 1, MG_OP_RESTRICT = 2, MG_OP_PROLONG = 3, &
                        MG_OP_EXACT = 4, MG_OP_RELAX_COLOR = 5, MG_OP_RELAX_FUSED = 6, &
                        MG_OP_RESREST = 7, MG_OP_RELAX_RES = 8, MG_OP_RELAX_RES_FUSED = 9, &
                        MG_OP_RELAX_LAST = 10   ! the sweeps that end a V-cycle (level 1)

  integer(c_size_t), parameter :: R8 = 8_c_size_t, I4 = 4_c_size_t

  type :: dev_level
    type(c_ptr) :: u = c_null_ptr, rhs = c_null_ptr
    type(c_ptr) :: ualt = c_null_ptr   ! ping-pong partner of u for the out-of-place fused smoother
  end type

  type :: dev_xfer
    type(ndsmk_xfer) :: x
    type(c_ptr) :: blob = c_null_ptr
  end type

  type :: mg_solver
    integer :: ndim = 0, ngrids = 0
    integer :: ms = 5, nmax_exact = 10000
    logical :: use_max = .true.
    real(wp) :: ex_tol = 1.0e-13_wp
    character(len=1) :: bcs(6) = 'N'
    type(level_t), allocatable :: lev(:)
    type(dev_level), allocatable :: dl(:)
    type(dev_xfer), allocatable :: xf(:)
    type(c_ptr) :: r = c_null_ptr        ! residual scratch, level-1 sized
    type(c_ptr) :: prev = c_null_ptr     ! previous iterate of level 1 (update_u)
    type(c_ptr) :: scr = c_null_ptr      ! coarsest-level scratch
    type(c_ptr) :: info = c_null_ptr     ! 2 x int64 on the device: exact sweeps, unconverged coarse solves
    integer(ik) :: vcycles_done = 0
    integer(ik) :: npts1 = 0             ! elements of the level-1 device arrays (local window if z-slab)
    logical :: rhs1_zero = .false.        ! level-1 rhs is identically zero: kernels skip reading it
    ! ---- convergence metric without a pass of its own (mg_solve, level 1 on the fused smoother):
    ! the buffer holding the iterate a V-cycle starts from is kept untouched (u, ualt and prev
    ! rotate), and the launch of the cycle's last sweep evaluates max / sum |u_new - u_start|
    logical :: track = .false.           ! the V-cycle in progress runs in that mode
    logical :: met_done = .false.        ! its last sweep did evaluate the metric
    type(c_ptr) :: keep = c_null_ptr     ! the kept buffer
    ! ---- one V-cycle + metric as an executable graph (mg_solve_lanes; small solves that are launch bound) ----
    type(c_ptr) :: graph = c_null_ptr    ! hipGraphExec_t, or null
    type(c_ptr) :: gkey_p(3) = c_null_ptr  ! what it was recorded for: level-1 u, rhs, prev ...
    integer :: gkey_i(6) = 0             ! ... ms, nmax_exact, use_max, first level of the tail launch, lane (its
                                         ! scratch), level-1 rhs declared zero
    real(wp) :: gkey_r = 0               ! ... ex_tol
    integer :: precision = 0             ! 0 fp64 (reference arithmetic), 1 mixed where level 1 is large enough,
                                         ! 2 mixed wherever the fp32 kernels cover level 1 (tests)
    ! ---- z-slab mode (level 1 distributed, SURVEY 8e); unused otherwise
    logical :: slab = .false.
    logical :: has_coarse = .true.       ! levels >= 2 live here (rank 0 only when distributed)
    type(slab_t) :: sl
    type(c_ptr) :: cbuf = c_null_ptr     ! window [cb0, cb1) of level-2 planes: restriction out / prolongation in
    integer(ik) :: plane1 = 0, plane2 = 0  ! elements per z-plane of level 1 / level 2
  end type

contains

  ! ------------------------------------------------------------------
  ! construction
  ! ------------------------------------------------------------------
  function mg_create(s, ndim, nshape, qx, qy, qz, bcs, ngrids_req, slab, coarse_here, ext) result(rc)
    type(mg_solver), intent(out) :: s
    integer, intent(in) :: ndim
    integer(c_int32_t), intent(in) :: nshape(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    character(len=1), intent(in) :: bcs(:)
    integer, intent(in) :: ngrids_req          ! <= 0: the reference's rule
    type(slab_t), intent(in), optional :: slab ! this rank's z-slab of level 1 (distributed runs)
    logical, intent(in), optional :: coarse_here  ! slab runs: do levels >= 2 live in this solver?  (default: on
                                                  ! rank 0; .false. everywhere when level 2 is distributed too)
    real(wp), intent(in), optional :: ext(2, 3)   ! see build_levels
    integer(c_int) :: rc
    integer :: l, d
    integer(c_size_t) :: nbytes

    rc = NDSMK_EARG
    if (ndim /= 2 .and. ndim /= 3) then
      rc = ndsmk_note_error(NDSMK_EARG, "multigrid solver: 2 or 3 dimensions"//c_null_char); return
    end if
    if (any(nshape(1:ndim) < 4)) then          ! the reference needs nmin >= 4 for one grid (its level count
      ! FLOOR(LOG(nmin / 4) / LOG(2)) + 1 is not positive below that and it stops or crashes)
      rc = ndsmk_note_error(NDSMK_EARG, "grid too small: every dimension needs at least 4 points"//c_null_char); return
    end if
    if (size(bcs) < 2 * ndim) return
    do d = 1, 2 * ndim
      if (bcs(d) /= 'D' .and. bcs(d) /= 'N') then
        rc = ndsmk_note_error(NDSMK_EARG, "boundary letters must be D or N"//c_null_char); return
      end if
    end do

    s%ndim = ndim
    s%bcs = 'N'
    s%bcs(1:2 * ndim) = bcs(1:2 * ndim)
    s%ngrids = ndsm_level_count(ndim, nshape)
    if (ngrids_req > 0) s%ngrids = min(ngrids_req, s%ngrids)
    if (s%ngrids < 1) return

    rc = ndsmk_init(-1_c_int)
    if (rc /= 0) return

    call build_levels(ndim, nshape, qx, qy, qz, s%ngrids, s%lev, ext)
    do l = 1, s%ngrids
      call fill_grid_desc(ndim, s%lev(l), s%bcs)
    end do
    s%npts1 = s%lev(1)%npts
    s%plane1 = int(s%lev(1)%n(1), ik) * int(s%lev(1)%n(2), ik)
    if (s%ngrids >= 2) s%plane2 = int(s%lev(2)%n(1), ik) * int(s%lev(2)%n(2), ik)
    if (present(slab)) then
      rc = NDSMK_EARG
      if (ndim /= 3 .or. s%ngrids < 2) return
      s%slab = .true.
      s%sl = slab
      s%has_coarse = (slab%rank == 0)
      if (present(coarse_here)) s%has_coarse = coarse_here
      call apply_slab_window(s%lev(1), s%sl)
      s%npts1 = s%plane1 * int(s%sl%nloc, ik)
    end if

    allocate (s%dl(s%ngrids))
    do l = 1, s%ngrids
      if (l >= 2 .and. .not. s%has_coarse) cycle
      nbytes = int(merge(s%npts1, s%lev(l)%npts, l == 1), c_size_t) * R8
      rc = ndsmk_alloc(s%dl(l)%u, nbytes); if (rc /= 0) return
      rc = ndsmk_alloc(s%dl(l)%rhs, nbytes); if (rc /= 0) return
      rc = ndsmk_alloc(s%dl(l)%ualt, nbytes); if (rc /= 0) return
      rc = ndsmk_fill0(s%dl(l)%u, nbytes); if (rc /= 0) return
      rc = ndsmk_fill0(s%dl(l)%ualt, nbytes); if (rc /= 0) return
      rc = ndsmk_fill0(s%dl(l)%rhs, nbytes); if (rc /= 0) return
    end do
    nbytes = int(s%npts1, c_size_t) * R8
    ! the residual scratch also serves level 2 (rank 0 of a z-slab run: the slab may be smaller)
    if (s%slab .and. s%has_coarse) nbytes = max(nbytes, int(s%lev(2)%npts, c_size_t) * R8)
    rc = ndsmk_alloc(s%r, nbytes); if (rc /= 0) return
    rc = ndsmk_fill0(s%r, nbytes); if (rc /= 0) return
    rc = ndsmk_alloc(s%prev, nbytes); if (rc /= 0) return
    if (s%slab) then
      nbytes = int(s%plane2, c_size_t) * int(max(s%sl%cb1 - s%sl%cb0, 1), c_size_t) * R8
      rc = ndsmk_alloc(s%cbuf, nbytes); if (rc /= 0) return
      rc = ndsmk_fill0(s%cbuf, nbytes); if (rc /= 0) return
    end if
    rc = ndsmk_alloc(s%scr, int(s%lev(s%ngrids)%npts, c_size_t) * R8); if (rc /= 0) return
    rc = ndsmk_alloc(s%info, 16_c_size_t); if (rc /= 0) return
    rc = ndsmk_fill0(s%info, 16_c_size_t); if (rc /= 0) return

    allocate (s%xf(max(s%ngrids - 1, 0)))
    do l = 1, s%ngrids - 1
      if (l >= 2 .and. .not. s%has_coarse) cycle
      rc = upload_xfer(s, l); if (rc /= 0) return
    end do
    if (s%slab) then      ! the level-1 arrays are a window of the global planes
      s%xf(1)%x%f_k0 = s%sl%k0
      s%xf(1)%x%f_beg = s%sl%g
      s%xf(1)%x%f_cnt = s%sl%z1 - s%sl%z0
      s%xf(1)%x%c_k0 = s%sl%cb0
    end if
    rc = 0
  end function

  ! Build the 1-D tables of the l -> l+1 transfer on the host and pack them
  ! into one device allocation.
  function upload_xfer(s, l) result(rc)
    type(mg_solver), intent(inout) :: s
    integer, intent(in) :: l
    integer(c_int) :: rc
    type(axis_xfer_t), target :: t(3)
    integer :: d
    logical :: ok
    integer(c_size_t) :: off, total
    integer(c_size_t) :: o_plo(3), o_pwl(3), o_pwh(3), o_rlo(3), o_rcnt(3), o_rw(3)

    rc = NDSMK_EARG
    total = 0
    do d = 1, s%ndim
      call build_axis_xfer(s%lev(l)%ax(d)%q, int(s%lev(l)%n(d)), s%lev(l + 1)%ax(d)%q, &
                           int(s%lev(l + 1)%n(d)), t(d), ok)
      if (.not. ok) return
      ! doubles first so that every table stays 8-byte aligned
      o_pwl(d) = total; total = total + int(t(d)%nf, c_size_t) * R8
      o_pwh(d) = total; total = total + int(t(d)%nf, c_size_t) * R8
      o_rw(d) = total; total = total + int(t(d)%maxt, c_size_t) * int(t(d)%nc, c_size_t) * R8
      o_plo(d) = total; total = total + pad8(int(t(d)%nf, c_size_t) * I4)
      o_rlo(d) = total; total = total + pad8(int(t(d)%nc, c_size_t) * I4)
      o_rcnt(d) = total; total = total + pad8(int(t(d)%nc, c_size_t) * I4)
    end do
    rc = ndsmk_alloc(s%xf(l)%blob, total); if (rc /= 0) return
    s%xf(l)%x%stream_ok = stream_restrict_applies(s, l, t)

    s%xf(l)%x%nf = s%lev(l)%n
    s%xf(l)%x%nc = s%lev(l + 1)%n
    s%xf(l)%x%maxt = 1
    s%xf(l)%x%w2 = 0
    s%xf(l)%x%f_k0 = 0; s%xf(l)%x%f_beg = 0; s%xf(l)%x%f_cnt = s%lev(l)%n(3)
    s%xf(l)%x%c_k0 = 0; s%xf(l)%x%c_beg = 0; s%xf(l)%x%c_cnt = s%lev(l + 1)%n(3)
    do d = 1, 3
      s%xf(l)%x%plo(d) = c_null_ptr; s%xf(l)%x%pwl(d) = c_null_ptr; s%xf(l)%x%pwh(d) = c_null_ptr
      s%xf(l)%x%rlo(d) = c_null_ptr; s%xf(l)%x%rcnt(d) = c_null_ptr; s%xf(l)%x%rw(d) = c_null_ptr
    end do
    do d = 1, s%ndim
      s%xf(l)%x%maxt(d) = t(d)%maxt
      s%xf(l)%x%w2(d) = t(d)%w2
      s%xf(l)%x%pwl(d) = dptr_offset(s%xf(l)%blob, o_pwl(d))
      s%xf(l)%x%pwh(d) = dptr_offset(s%xf(l)%blob, o_pwh(d))
      s%xf(l)%x%rw(d) = dptr_offset(s%xf(l)%blob, o_rw(d))
      s%xf(l)%x%plo(d) = dptr_offset(s%xf(l)%blob, o_plo(d))
      s%xf(l)%x%rlo(d) = dptr_offset(s%xf(l)%blob, o_rlo(d))
      s%xf(l)%x%rcnt(d) = dptr_offset(s%xf(l)%blob, o_rcnt(d))
      rc = ndsmk_h2d(s%xf(l)%x%pwl(d), c_loc(t(d)%pwl), int(t(d)%nf, c_size_t) * R8); if (rc /= 0) return
      rc = ndsmk_h2d(s%xf(l)%x%pwh(d), c_loc(t(d)%pwh), int(t(d)%nf, c_size_t) * R8); if (rc /= 0) return
      rc = ndsmk_h2d(s%xf(l)%x%rw(d), c_loc(t(d)%rw), &
                     int(t(d)%maxt, c_size_t) * int(t(d)%nc, c_size_t) * R8); if (rc /= 0) return
      rc = ndsmk_h2d(s%xf(l)%x%plo(d), c_loc(t(d)%plo), int(t(d)%nf, c_size_t) * I4); if (rc /= 0) return
      rc = ndsmk_h2d(s%xf(l)%x%rlo(d), c_loc(t(d)%rlo), int(t(d)%nc, c_size_t) * I4); if (rc /= 0) return
      rc = ndsmk_h2d(s%xf(l)%x%rcnt(d), c_loc(t(d)%rcnt), int(t(d)%nc, c_size_t) * I4); if (rc /= 0) return
    end do
    rc = 0
  contains
    pure function pad8(b) result(p)
      integer(c_size_t), intent(in) :: b
      integer(c_size_t) :: p
      p = ((b + 7_c_size_t) / 8_c_size_t) * 8_c_size_t
    end function
  end function

  ! Does restrict_stream.hip cover the transfer l -> l+1?  Same idea as below,
  ! with that kernel's footprint (no stencil halo).
  ! bit 1: the kernel's tile covers every coarse tile's taps; bit 0: and the level is large
  ! enough for it to be the default choice
  function stream_restrict_applies(s, l, t) result(ok)
    type(mg_solver), intent(in) :: s
    integer, intent(in) :: l
    type(axis_xfer_t), intent(in) :: t(3)
    integer(c_int) :: ok
    integer(c_int) :: ci, cj, fx, fy, mt
    integer :: a0, a1, f0, nt, k, st
    ok = 0
    if (s%ndim /= 3) return
    call get_environment_variable("NDSM_HIP_NO_STREAM_RESTRICT", status=st)   ! development switch
    if (st == 0) return
    call ndsmk_restrict_stream_tile(ci, cj, fx, fy, mt)
    if (any([t(1)%maxt, t(2)%maxt, t(3)%maxt] > mt)) return
    nt = (t(1)%nc + ci - 1) / ci
    do k = 0, nt - 1
      a0 = k * ci + 1; a1 = min(a0 + ci - 1, int(t(1)%nc))
      f0 = iand(t(1)%rlo(a0), not(1))
      if (t(1)%rlo(a1) + t(1)%rcnt(a1) - 1 > f0 + fx - 1) return
    end do
    nt = (t(2)%nc + cj - 1) / cj
    do k = 0, nt - 1
      a0 = k * cj + 1; a1 = min(a0 + cj - 1, int(t(2)%nc))
      f0 = t(2)%rlo(a0)
      if (t(2)%rlo(a1) + t(2)%rcnt(a1) - 1 > f0 + fy - 1) return
    end do
    ok = 2
    if (s%lev(l)%npts >= 6_ik * 1024_ik * 1024_ik) ok = 3
    ! What the kernel's per-chunk SCHEDULE of the z windows has to hold, MEASURED here and checked by the launcher
    ! against the kernel's own limits (restrict_stream.hip: launch_rs_t): bits 8-15 the largest number of coarse
    ! windows a fine plane lies in, bits 16-31 the most fine planes 64 consecutive coarse planes (the longest
    ! chunk) span.  Bit 2: measured (a descriptor without these numbers never takes the scheduled form).
    block
      integer :: cntf(0:t(3)%nf), kk, wmax, smax
      cntf = 0
      do k = 1, t(3)%nc
        cntf(t(3)%rlo(k)) = cntf(t(3)%rlo(k)) + 1
        kk = t(3)%rlo(k) + t(3)%rcnt(k)
        if (kk <= t(3)%nf) cntf(kk) = cntf(kk) - 1
      end do
      kk = 0; wmax = 0; smax = 0
      do k = 0, t(3)%nf - 1
        kk = kk + cntf(k)
        wmax = max(wmax, kk)
      end do
      do k = 1, t(3)%nc
        a1 = min(k + 63, int(t(3)%nc))
        smax = max(smax, t(3)%rlo(a1) + t(3)%rcnt(a1) - t(3)%rlo(k))
      end do
      ok = ok + 4 + 256 * min(wmax, 255) + 65536 * min(smax, 32767)
    end block
  end function

  ! Re-target an existing hierarchy at another set of boundary letters: only the
  ! update bounds / first colour / Neumann flag of the level descriptors change,
  ! the device arrays and transfer tables are reused.
  function mg_set_bcs(s, bcs) result(rc)
    type(mg_solver), intent(inout) :: s
    character(len=1), intent(in) :: bcs(:)
    integer(c_int) :: rc
    integer :: l, d
    rc = NDSMK_EARG
    if (size(bcs) < 2 * s%ndim) return
    do d = 1, 2 * s%ndim
      if (bcs(d) /= 'D' .and. bcs(d) /= 'N') return
    end do
    if (all(s%bcs(1:2 * s%ndim) == bcs(1:2 * s%ndim))) then    ! nothing changes (a recorded cycle stays valid)
      rc = 0
      return
    end if
    s%bcs = 'N'
    s%bcs(1:2 * s%ndim) = bcs(1:2 * s%ndim)
    do l = 1, s%ngrids
      call fill_grid_desc(s%ndim, s%lev(l), s%bcs)
    end do
    if (s%slab) call apply_slab_window(s%lev(1), s%sl)
    call drop_graph(s)
    rc = 0
  end function

  subroutine drop_graph(s)
    type(mg_solver), intent(inout) :: s
    integer(c_int) :: rc
    if (c_associated(s%graph)) then
      rc = ndsmk_sync()
      rc = ndsmk_graph_destroy(s%graph)
    end if
    s%graph = c_null_ptr
  end subroutine

  ! May a solve-loop cycle of this solver (mg_vcycle + metric pass, as mg_solve_lanes enqueues it) be RECORDED
  ! and replayed as one graph launch?  Decided before anything is recorded:
  !   * nothing in the cycle may come back to the host: the coarsest grid is solved inside the tail launch or by
  !     the single-workgroup kernel, not by the host-driven fall-back loop (a copy + synchronise on a capturing
  !     stream would end the recording half way);
  !   * every array must be where it was when the next cycle starts: the in-place kernels keep them; the
  !     out-of-place fused smoother swaps a level's array with its partner once per pass, and only level 1
  !     makes an even number of passes per cycle (ceil(ms/2) down and up) - so no level below it may use it;
  !   * it must pay: 2-D levels from ~180^2 points on (below, a cycle is a handful of launches - sweeps of a
  !     whole level and the tail are one each - and a replay costs more than it saves: 64^3 calls measured 12-13
  !     ms without, 14.5 ms with), 3-D solves up to 16 M points (dispatch latency, not bandwidth).
  function graph_eligible(s) result(ok)
    type(mg_solver), intent(in) :: s
    logical :: ok
    ok = .false.
    if (s%slab .or. s%ngrids < 2) return
    if (tail_first(s, 1) > s%ngrids) then
      if (ndsmk_solve_exact_on_device(s%lev(s%ngrids)%g) == 0) return
    end if
    if (s%ndim == 2) then
      ok = s%lev(1)%npts >= 32768_ik
    else if (s%ndim == 3) then
      ok = s%lev(1)%npts <= 16_ik * 1024_ik * 1024_ik .and. s%lev(2)%npts < 2_ik * 1024_ik * 1024_ik
    end if
  end function

  ! is the recorded graph still the sequence mg_vcycle + metric would enqueue now?
  function graph_valid(s, lane) result(ok)
    type(mg_solver), intent(in) :: s
    integer, intent(in) :: lane
    logical :: ok
    ok = c_associated(s%graph)
    if (.not. ok) return
    ok = c_associated(s%gkey_p(1), s%dl(1)%u) .and. c_associated(s%gkey_p(2), s%dl(1)%rhs) .and. &
         c_associated(s%gkey_p(3), s%prev) .and. s%gkey_i(1) == s%ms .and. s%gkey_i(2) == s%nmax_exact .and. &
         s%gkey_i(3) == merge(1, 0, s%use_max) .and. s%gkey_i(4) == tail_first(s, 1) .and. s%gkey_i(5) == lane .and. &
         s%gkey_i(6) == merge(1, 0, s%rhs1_zero) .and. s%gkey_r == s%ex_tol
  end function

  subroutine graph_stamp(s, lane)
    type(mg_solver), intent(inout) :: s
    integer, intent(in) :: lane
    s%gkey_p = [s%dl(1)%u, s%dl(1)%rhs, s%prev]
    s%gkey_i = [s%ms, s%nmax_exact, merge(1, 0, s%use_max), tail_first(s, 1), lane, merge(1, 0, s%rhs1_zero)]
    s%gkey_r = s%ex_tol
  end subroutine

  ! device-to-device copy of the level-1 solution into caller-owned HBM
  function mg_export_u(s, d_dst) result(rc)
    type(mg_solver), intent(in) :: s
    type(c_ptr), intent(in) :: d_dst
    integer(c_int) :: rc
    rc = ndsmk_d2d(d_dst, s%dl(1)%u, int(s%npts1, c_size_t) * R8)
  end function

  subroutine mg_mark_rhs_set(s)
    type(mg_solver), intent(inout) :: s
    s%rhs1_zero = .false.
  end subroutine

  function mg_reset_info(s) result(rc)
    type(mg_solver), intent(inout) :: s
    integer(c_int) :: rc
    rc = ndsmk_fill0(s%info, 16_c_size_t)
  end function

  subroutine mg_destroy(s)
    type(mg_solver), intent(inout) :: s
    integer :: l
    integer(c_int) :: rc
    call drop_graph(s)
    if (allocated(s%dl)) then
      do l = 1, size(s%dl)
        rc = ndsmk_free(s%dl(l)%u)
        rc = ndsmk_free(s%dl(l)%ualt)
        rc = ndsmk_free(s%dl(l)%rhs)
      end do
      deallocate (s%dl)
    end if
    if (allocated(s%xf)) then
      do l = 1, size(s%xf)
        rc = ndsmk_free(s%xf(l)%blob)
      end do
      deallocate (s%xf)
    end if
    rc = ndsmk_free(s%r); s%r = c_null_ptr
    rc = ndsmk_free(s%prev); s%prev = c_null_ptr
    rc = ndsmk_free(s%scr); s%scr = c_null_ptr
    rc = ndsmk_free(s%info); s%info = c_null_ptr
    rc = ndsmk_free(s%cbuf); s%cbuf = c_null_ptr
    if (allocated(s%lev)) deallocate (s%lev)
    s%ngrids = 0
  end subroutine

  ! ------------------------------------------------------------------
  ! data movement (host buffers are the caller's; everything blocking)
  ! ------------------------------------------------------------------
  function mg_set_u(s, h_u) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: h_u
    integer(c_int) :: rc
    rc = ndsmk_h2d(s%dl(1)%u, h_u, int(s%npts1, c_size_t) * R8)
  end function

  function mg_set_rhs(s, h_rhs) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: h_rhs
    integer(c_int) :: rc
    rc = ndsmk_h2d(s%dl(1)%rhs, h_rhs, int(s%npts1, c_size_t) * R8)
    s%rhs1_zero = .false.
  end function

  function mg_zero_rhs(s) result(rc)
    type(mg_solver), intent(inout) :: s
    integer(c_int) :: rc
    rc = ndsmk_fill0(s%dl(1)%rhs, int(s%npts1, c_size_t) * R8)
    s%rhs1_zero = .true.       ! NDSM's 3-D problems are Laplace problems (ndsm_vector_potential.f90:640-641)
  end function

  function mg_get_u(s, h_u) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: h_u
    integer(c_int) :: rc
    rc = ndsmk_d2h(h_u, s%dl(1)%u, int(s%npts1, c_size_t) * R8)
  end function

  ! device pointer + element count of a level buffer (tests, bench)
  function mg_level_ptr(s, level, which, npts) result(p)
    type(mg_solver), intent(in) :: s
    integer, intent(in) :: level, which
    integer(ik), intent(out) :: npts
    type(c_ptr) :: p
    p = c_null_ptr; npts = 0
    if (level < 1 .or. level > s%ngrids) return
    if (level >= 2 .and. .not. s%has_coarse) return
    npts = merge(s%npts1, s%lev(level)%npts, level == 1)
    select case (which)
    case (MG_BUF_U); p = s%dl(level)%u
    case (MG_BUF_RHS); p = s%dl(level)%rhs
    case (MG_BUF_R)
      p = s%r
    end select
  end function

  function mg_read_info(s, sweeps, unconverged) result(rc)
    type(mg_solver), intent(in) :: s
    integer(ik), intent(out) :: sweeps, unconverged
    integer(c_int) :: rc
    integer(ik), target :: buf(2)
    buf = 0
    rc = ndsmk_d2h(c_loc(buf), s%info, 16_c_size_t)
    sweeps = buf(1); unconverged = buf(2)
  end function

  ! rhs pointer handed to the kernels: null when the level's rhs is known to be zero
  function rhs_of(s, level) result(p)
    type(mg_solver), intent(in) :: s
    integer, intent(in) :: level
    type(c_ptr) :: p
    p = s%dl(level)%rhs
    if (level == 1 .and. s%rhs1_zero) p = c_null_ptr
  end function

  ! ------------------------------------------------------------------
  ! single grid operations (also the building blocks of the cycle)
  ! ------------------------------------------------------------------
  function mg_op(s, op, level, count) result(rc)
    type(mg_solver), intent(inout) :: s
    integer, intent(in) :: op, level, count
    integer(c_int) :: rc
    integer(c_int) :: swapped
    integer :: variant
    type(c_ptr) :: tmp
    rc = NDSMK_EARG
    if (level < 1 .or. level > s%ngrids) return
    if (level >= 2 .and. .not. s%has_coarse) return
    if (s%slab .and. level == 1 .and. (op == MG_OP_RESTRICT .or. op == MG_OP_PROLONG)) return  ! mg_slab_* instead
    if (level == 1 .and. s%track .and. (op == MG_OP_RELAX .or. op == MG_OP_RELAX_RES .or. op == MG_OP_RELAX_LAST)) then
      rc = relax_tracked(s, int(count), op == MG_OP_RELAX_RES, op == MG_OP_RELAX_LAST)
      return
    end if
    select case (op)
    case (MG_OP_RELAX, MG_OP_RELAX_COLOR, MG_OP_RELAX_FUSED, MG_OP_RELAX_LAST)
      variant = merge(0, merge(1, 2, op == MG_OP_RELAX_COLOR), op == MG_OP_RELAX .or. op == MG_OP_RELAX_LAST)
      rc = ndsmk_relax(s%lev(level)%g, s%dl(level)%u, s%dl(level)%ualt, rhs_of(s, level), int(count, c_int), &
                       int(variant, c_int), swapped)
      if (rc == 0 .and. swapped /= 0) then      ! the swept field lives in the partner array now
        tmp = s%dl(level)%u
        s%dl(level)%u = s%dl(level)%ualt
        s%dl(level)%ualt = tmp
      end if
    case (MG_OP_RELAX_RES, MG_OP_RELAX_RES_FUSED)  ! count sweeps, then residual -> scratch (last sweep + residual in one launch)
      rc = ndsmk_relax_residual(s%lev(level)%g, s%dl(level)%u, s%dl(level)%ualt, rhs_of(s, level), s%r, &
                                int(count, c_int), merge(0_c_int, 2_c_int, op == MG_OP_RELAX_RES), swapped)
      if (rc == 0 .and. swapped /= 0) then
        tmp = s%dl(level)%u
        s%dl(level)%u = s%dl(level)%ualt
        s%dl(level)%ualt = tmp
      end if
    case (MG_OP_RESIDUAL)
      rc = ndsmk_residual(s%lev(level)%g, s%dl(level)%u, rhs_of(s, level), s%r)
    case (MG_OP_RESTRICT)       ! r(level) -> rhs(level+1), u(level+1) = 0
      if (level >= s%ngrids) return
      rc = ndsmk_restrict(s%xf(level)%x, s%r, s%dl(level + 1)%rhs, s%dl(level + 1)%u)
    case (MG_OP_RESREST)        ! (retired: the fused residual + restriction kernel was slower than sweep + residual
      return                    !  in one launch followed by the streamed restriction; the slot keeps its number)
    case (MG_OP_PROLONG)        ! u(level) += P u(level+1)
      if (level >= s%ngrids) return
      rc = ndsmk_prolong_add(s%xf(level)%x, s%dl(level + 1)%u, s%dl(level)%u)
    case (MG_OP_EXACT)
      rc = ndsmk_solve_exact(s%lev(level)%g, s%dl(level)%u, s%dl(level)%rhs, s%scr, s%ex_tol, &
                             merge(1_c_int, 0_c_int, s%use_max), int(s%nmax_exact, c_int), s%info)
    end select
  end function

  ! level-1 sweeps in track mode: three rotating buffers, s%keep is never written
  ! (prolong: u(1) += P u(2) first - folded into the first launch's loads where possible)
  function relax_tracked(s, nsweeps, with_res, last, prolong) result(rc)
    type(mg_solver), intent(inout), target :: s
    integer, intent(in) :: nsweeps
    logical, intent(in) :: with_res, last
    logical, intent(in), optional :: prolong
    integer(c_int) :: rc
    integer(c_int) :: where, met
    type(c_ptr) :: b(0:2), rr, pv, px, uc
    b(0) = s%dl(1)%u; b(1) = s%dl(1)%ualt; b(2) = s%prev
    rr = c_null_ptr; if (with_res) rr = s%r
    pv = c_null_ptr; if (last) pv = s%keep
    px = c_null_ptr; uc = c_null_ptr
    if (present(prolong)) then
      if (prolong) then
        px = c_loc(s%xf(1)%x); uc = s%dl(2)%u
      end if
    end if
    rc = ndsmk_relax3(s%lev(1)%g, b(0), b(1), b(2), s%keep, rhs_of(s, 1), int(nsweeps, c_int), rr, pv, where, met, &
                      px, uc)
    if (rc /= 0) return
    if (last) s%met_done = (met /= 0)
    s%dl(1)%u = b(where)
    s%dl(1)%ualt = b(mod(where + 1, 3))
    s%prev = b(mod(where + 2, 3))
  end function

  ! ------------------------------------------------------------------
  ! one V-cycle from the finest grid; nothing here waits for the GPU
  ! ------------------------------------------------------------------
  function mg_vcycle(s) result(rc)
    type(mg_solver), intent(inout) :: s
    integer(c_int) :: rc
    rc = mg_vcycle_from(s, 1)
    if (rc == 0) s%vcycles_done = s%vcycles_done + 1
  end function

  ! The V-cycle with level `ltop` as its finest grid (ltop = 1: the whole cycle;
  ! ltop = 2: the part a z-slab run executes on rank 0 between the distributed
  ! restriction and prolongation of level 1).
  function mg_vcycle_from(s, ltop) result(rc)
    type(mg_solver), intent(inout) :: s
    integer, intent(in) :: ltop
    integer(c_int) :: rc
    integer :: l, lt

    ! levels lt .. ngrids run as ONE launch with all of them in LDS (tail.hip); lt = ngrids + 1: none do
    lt = tail_first(s, ltop)

    ! descend: pre-smooth, residual, restrict (fine_to_coarse, :482-560)
    do l = ltop, min(lt, s%ngrids) - 1
      rc = mg_op(s, MG_OP_RELAX_RES, l, s%ms); if (rc /= 0) return
      rc = mg_op(s, MG_OP_RESTRICT, l, 1); if (rc /= 0) return
    end do

    ! coarsest grid: iterate the smoother to ex_tol (solve_exact, :728-800)
    if (lt <= s%ngrids) then
      rc = tail_cycle(s, lt); if (rc /= 0) return
    else
      rc = mg_op(s, MG_OP_EXACT, s%ngrids, 1); if (rc /= 0) return
    end if

    ! ascend: smooth the coarse problem, interpolate + correct, post-smooth
    ! (coarse_to_fine, :593-684)
    ! The reference smooths level l-1 ms times after the correction (:672-675) and, one level up the
    ! loop, ms times again before interpolating it further (:651-654): 2 ms consecutive sweeps of one
    ! array.  They are issued as ONE relax call here (five two-sweep passes instead of 2+2+1 twice on the
    ! streamed levels, one launch instead of two on the single-workgroup levels) - the same sweeps in the
    ! same order.
    if (lt > s%ngrids) then
      rc = mg_op(s, MG_OP_RELAX, s%ngrids, s%ms); if (rc /= 0) return
    end if
    do l = min(lt, s%ngrids), ltop + 1, -1
      if (l - 1 == 1 .and. s%track) then      ! interpolate + correct + post-smooth as one call
        rc = relax_tracked(s, s%ms, .false., .true., prolong=.true.); if (rc /= 0) return
      else
        rc = mg_op(s, MG_OP_PROLONG, l - 1, 1); if (rc /= 0) return
        if (l - 1 == ltop) then
          rc = mg_op(s, merge(MG_OP_RELAX_LAST, MG_OP_RELAX, l - 1 == 1), l - 1, s%ms); if (rc /= 0) return
        else
          rc = mg_op(s, MG_OP_RELAX, l - 1, 2 * s%ms); if (rc /= 0) return
        end if
      end if
    end do
    rc = 0
  end function

  ! First level of the V-cycle's tail: the smallest lt > ltop (lt >= 2) from which every level down to the
  ! coarsest grid fits the single-launch kernel of tail.hip; ngrids + 1 if there is none.  (A few dozen
  ! integer comparisons per V-cycle: not cached, so that boundary letters and the test switch can change.)
  function tail_first(s, ltop) result(lt)
    type(mg_solver), intent(in) :: s
    integer, intent(in) :: ltop
    integer :: lt
    integer :: l, q, nlev
    type(ndsmk_grid) :: gg(6)
    type(ndsmk_xfer) :: xx(5)
    lt = s%ngrids + 1
    if (.not. s%has_coarse) return
    do l = max(2, ltop + 1), s%ngrids - 1           ! (the kernel needs two levels)
      nlev = s%ngrids - l + 1
      if (nlev > 6) cycle
      do q = 1, nlev
        gg(q) = s%lev(l + q - 1)%g
        if (q < nlev) xx(q) = s%xf(l + q - 1)%x
      end do
      if (ndsmk_tail_applies(int(nlev, c_int), gg, xx) /= 0) then
        lt = l
        exit
      end if
    end do
  end function

  function tail_cycle(s, lt) result(rc)
    type(mg_solver), intent(inout) :: s
    integer, intent(in) :: lt
    integer(c_int) :: rc
    integer :: q, nlev
    type(ndsmk_grid) :: gg(6)
    type(ndsmk_xfer) :: xx(5)
    type(c_ptr) :: uu(6), rr(6)
    nlev = s%ngrids - lt + 1
    do q = 1, nlev
      gg(q) = s%lev(lt + q - 1)%g
      if (q < nlev) xx(q) = s%xf(lt + q - 1)%x
      uu(q) = s%dl(lt + q - 1)%u
      rr(q) = s%dl(lt + q - 1)%rhs
    end do
    rc = ndsmk_tail_cycle(int(nlev, c_int), gg, xx, uu, rr, int(s%ms, c_int), s%ex_tol, &
                          merge(1_c_int, 0_c_int, s%use_max), int(s%nmax_exact, c_int), s%info)
  end function

  ! z-slab level 1, pieces of a pass whose halo exchange overlaps its interior: one fused pass
  ! (n = 1 or 2 sweeps) over the owned LOCAL planes [z0, z1) only, u -> ualt; mg_swap_u when all
  ! pieces of the pass are enqueued
  ! src given: the launch also interpolates the coarse-grid correction while it loads (u + P src;
  ! src = src_n whole coarse planes starting at global coarse plane src_k0, which must cover the
  ! brackets of the window's planes and of the ghost planes the launch reads) - only where
  ! mg_window_prolong_ok says so
  ! met: 0 no metric; 1 / 2 the launch also evaluates max / sum of |u_new - s%prev| over the planes it
  ! stores and leaves it on the device (1) or adds it to what is there (2) - mg_window_metric_ok
  function mg_relax_window(s, n, z0, z1, src, src_k0, src_n, met) result(rc)
    type(mg_solver), intent(inout), target :: s
    integer, intent(in) :: n, z0, z1
    type(c_ptr), intent(in), optional :: src
    integer, intent(in), optional :: src_k0, src_n, met
    integer(c_int) :: rc
    type(ndsmk_xfer), target :: x
    type(c_ptr) :: pv
    integer(c_int) :: acc
    pv = c_null_ptr; acc = 0
    if (present(met)) then
      if (met /= 0) pv = s%prev
      if (met == 2) acc = 1
    end if
    if (present(src)) then
      x = s%xf(1)%x
      x%c_k0 = src_k0
      x%c_cnt = src_n
      rc = ndsmk_fused_window(s%lev(1)%g, s%dl(1)%u, s%dl(1)%ualt, rhs_of(s, 1), int(n, c_int), int(z0, c_int), &
                              int(z1, c_int), c_loc(x), src, pv, acc)
    else
      rc = ndsmk_fused_window(s%lev(1)%g, s%dl(1)%u, s%dl(1)%ualt, rhs_of(s, 1), int(n, c_int), int(z0, c_int), &
                              int(z1, c_int), c_null_ptr, c_null_ptr, pv, acc)
    end if
  end function

  ! the sweep + residual pass on owned planes [z0, z1) of a slab: u -> ualt, residual of the result -> r
  function mg_relax_res_window(s, z0, z1) result(rc)
    type(mg_solver), intent(inout), target :: s
    integer, intent(in) :: z0, z1
    integer(c_int) :: rc
    rc = ndsmk_fused_window_res(s%lev(1)%g, s%dl(1)%u, s%dl(1)%ualt, rhs_of(s, 1), s%r, int(z0, c_int), int(z1, c_int))
  end function

  function mg_window_metric_ok(s) result(ok)
    type(mg_solver), intent(in) :: s
    logical :: ok
    ok = ndsmk_fused_metric_ok(s%lev(1)%g) /= 0
  end function

  ! can a pass of n sweeps over this slab fold the prolongation in? (one sweep, or two on a Laplace problem)
  function mg_window_prolong_ok(s, n) result(ok)
    type(mg_solver), intent(in) :: s
    integer, intent(in) :: n
    logical :: ok
    ok = ndsmk_fused_prolong_ok(s%lev(1)%g, rhs_of(s, 1), int(n, c_int)) /= 0
  end function

  subroutine mg_swap_u(s)
    type(mg_solver), intent(inout) :: s
    type(c_ptr) :: tmp
    tmp = s%dl(1)%u; s%dl(1)%u = s%dl(1)%ualt; s%dl(1)%ualt = tmp
  end subroutine

  ! z-slab level 1: restrict the slab's residual into its window of coarse planes
  ! [ck0, ck1) of cbuf (the planes are shipped to rank 0 by the caller)
  ! (dst, dst_k0 given: into that array of whole coarse planes instead, whose plane 0 is global
  ! coarse plane dst_k0 - the rhs slab of the next level when that is distributed as well)
  ! (ka, kb given: only the coarse planes [ka, kb) of that window - the pieces of a restriction whose
  ! residual exchange overlaps the planes that do not need it)
  function mg_slab_restrict(s, dst, dst_k0, ka, kb) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in), optional :: dst
    integer, intent(in), optional :: dst_k0, ka, kb
    integer(c_int) :: rc
    type(ndsmk_xfer) :: x
    integer :: k0, k1
    rc = 0
    k0 = s%sl%ck0; k1 = s%sl%ck1
    if (present(ka)) k0 = max(k0, ka)
    if (present(kb)) k1 = min(k1, kb)
    if (k1 <= k0) return
    x = s%xf(1)%x
    x%c_cnt = k1 - k0
    if (present(dst)) then
      x%c_k0 = dst_k0
      x%c_beg = k0 - dst_k0
      rc = ndsmk_restrict(x, s%r, dst, c_null_ptr)
    else
      x%c_k0 = s%sl%cb0
      x%c_beg = k0 - s%sl%cb0
      rc = ndsmk_restrict(x, s%r, s%cbuf, c_null_ptr)
    end if
  end function

  ! z-slab level 1: u += P (coarse planes [pk0, pk1) received into cbuf; or src, an array of whole
  ! coarse planes starting at global coarse plane src_k0 that holds them)
  function mg_slab_prolong(s, src, src_k0) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in), optional :: src
    integer, intent(in), optional :: src_k0
    integer(c_int) :: rc
    type(ndsmk_xfer) :: x
    if (present(src)) then
      x = s%xf(1)%x
      x%c_k0 = src_k0
      rc = ndsmk_prolong_add(x, src, s%dl(1)%u)
    else
      rc = ndsmk_prolong_add(s%xf(1)%x, s%cbuf, s%dl(1)%u)
    end if
  end function

  ! the same two with the fine side in fp32 (mixed-precision mode on z-slabs, ndsmh_world): r32 = the
  ! e-equation's residual of this slab, e = its correction
  function mg_slab_restrict_f32(s, r32, dst, dst_k0) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: r32
    type(c_ptr), intent(in), optional :: dst
    integer, intent(in), optional :: dst_k0
    integer(c_int) :: rc
    type(ndsmk_xfer) :: x
    rc = 0
    if (s%sl%ck1 <= s%sl%ck0) return
    x = s%xf(1)%x
    x%c_cnt = s%sl%ck1 - s%sl%ck0
    if (present(dst)) then
      x%c_k0 = dst_k0
      x%c_beg = s%sl%ck0 - dst_k0
      rc = ndsmk_restrict_f32(x, r32, dst, c_null_ptr)
    else
      x%c_k0 = s%sl%cb0
      x%c_beg = s%sl%ck0 - s%sl%cb0
      rc = ndsmk_restrict_f32(x, r32, s%cbuf, c_null_ptr)
    end if
  end function

  function mg_slab_prolong_f32(s, e, src, src_k0) result(rc)
    type(mg_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: e
    type(c_ptr), intent(in), optional :: src
    integer, intent(in), optional :: src_k0
    integer(c_int) :: rc
    type(ndsmk_xfer) :: x
    if (present(src)) then
      x = s%xf(1)%x
      x%c_k0 = src_k0
      rc = ndsmk_prolong_add_f32(x, src, e)
    else
      rc = ndsmk_prolong_add_f32(s%xf(1)%x, s%cbuf, e)
    end if
  end function

  ! can this z-slab of level 1 run the fp32 kernels? (same conditions as mg_mixed_applies, on the slab)
  function mg_mixed_slab_ok(s) result(ok)
    type(mg_solver), intent(in) :: s
    logical :: ok
    ok = .false.
    if (s%ndim /= 3 .or. .not. s%slab .or. s%ms < 1) return
    if (s%lev(1)%g%all_neumann /= 0) return
    if (mod(s%lev(1)%n(1), 2) /= 0 .or. s%lev(1)%n(1) < 64 .or. s%lev(1)%n(2) < 16) return
    if (any(s%lev(2)%n(1:3) < 16)) return
    if (iand(int(s%xf(1)%x%stream_ok), 2) == 0) return
    ok = .true.
  end function

  ! ------------------------------------------------------------------
  ! V-cycles to tolerance on the device-resident level-1 problem
  ! (solve_poisson_bvp, ndsm_poisson.f90:104-150).  ierr = 1 if vc_tol was not
  ! reached within nmax cycles.  hist (optional) receives du per cycle.
  ! ------------------------------------------------------------------
  function mg_solve(s, vc_tol, nmax, du_last, ncycles, ierr, hist) result(rc)
    type(mg_solver), intent(inout) :: s
    real(wp), intent(in) :: vc_tol
    integer, intent(in) :: nmax
    real(wp), intent(out) :: du_last
    integer, intent(out) :: ncycles, ierr
    real(wp), intent(inout), optional :: hist(:)
    integer(c_int) :: rc
    real(wp) :: met(2), du
    integer :: it
    logical :: trk

    if (mg_mixed_applies(s)) then
      rc = mg_solve_mixed(s, vc_tol, nmax, du_last, ncycles, ierr, hist)
      return
    end if
    du = huge(du)
    ncycles = 0
    ierr = 1
    trk = mg_track_applies(s)
    ! the caller's array is the "previous iterate" of the first comparison (:122)
    if (.not. trk) then
      rc = ndsmk_d2d(s%prev, s%dl(1)%u, int(s%npts1, c_size_t) * R8); if (rc /= 0) return
    end if
    do it = 1, nmax
      if (trk) then
        ! the buffer u sits in is the previous iterate: kept as it is, no copy, and the cycle's
        ! last sweep takes the metric against it
        s%track = .true.; s%met_done = .false.; s%keep = s%dl(1)%u
        rc = mg_vcycle(s)
        s%track = .false.
        if (rc /= 0) return
        if (s%met_done) then
          rc = ndsmk_fetch_fused_metric(met)
        else
          rc = ndsmk_diff_metrics(s%dl(1)%u, s%keep, s%npts1, 0_c_int, met)
        end if
        s%keep = c_null_ptr
        if (rc /= 0) return
      else
        rc = mg_vcycle(s); if (rc /= 0) return
        rc = ndsmk_diff_metrics(s%dl(1)%u, s%prev, s%npts1, 1_c_int, met); if (rc /= 0) return
      end if
      if (s%use_max) then
        du = met(1)
      else
        du = met(2) / real(s%npts1, wp)
      end if
      ncycles = it
      if (present(hist)) then
        if (it <= size(hist)) hist(it) = du
      end if
      if (du < vc_tol) then         ! strict (:136)
        ierr = 0
        exit
      end if
    end do
    du_last = du
    rc = 0
  end function

  ! Several INDEPENDENT solves in lockstep, solver l on lane l-1 (ndsmk_select_lane): per round every
  ! solve that has not converged yet enqueues one V-cycle + its metric on its own stream, then the metrics
  ! are collected.  Small problems (the six 2-D face solves of the vector potential) are dispatch latency
  ! and one host round trip per cycle when run one after the other; side by side the device overlaps them.
  ! Each solve runs exactly the kernels mg_solve would have run for it, in the same order: same bits, same
  ! cycle counts.  Solvers that would take mg_solve's mixed-precision path are not accepted.
  ! Before the call: whatever the solves read (right-hand sides, initial guesses) was enqueued on the MAIN
  ! stream; after it the main stream has waited for every lane.
  function mg_solve_lanes(ss, vc_tol, nmax, du_last, ncycles, ierr) result(rc)
    type(mg_solver), intent(inout) :: ss(:)
    real(wp), intent(in) :: vc_tol
    integer, intent(in) :: nmax
    real(wp), intent(out) :: du_last(:)
    integer, intent(out) :: ncycles(:), ierr(:)
    integer(c_int) :: rc, rc2
    logical :: active(size(ss))
    real(wp) :: met(2), du
    integer :: l, nl, st
    integer :: itl(size(ss))
    logical :: graphs(size(ss))
    type(dev_level), allocatable :: keep_dl(:)

    nl = size(ss)
    rc = NDSMK_EARG
    if (nl < 1 .or. nl > 6 .or. size(du_last) < nl .or. size(ncycles) < nl .or. size(ierr) < nl) return
    ! (a solver that mg_solve would run in its tracked form - metric inside the last sweep's launch, rotating
    ! buffers - runs the plain form here: V-cycle, then the metric pass; the same bits, and the tracked form's
    ! metric scratch is one per process)
    do l = 1, nl
      if (mg_mixed_applies(ss(l)) .or. ss(l)%slab) return
    end do
    du_last(1:nl) = huge(du); ncycles(1:nl) = 0; ierr(1:nl) = 1
    active = .true.
    ! From the second round on a solve replays its V-cycle + metric as ONE graph launch, recorded in the second
    ! round of the first call (the first round runs plain: every lazily created scratch exists afterwards) and
    ! kept with the solver while its arrays and parameters stay what they were - where graph_eligible says a
    ! cycle can be recorded and is worth it (the 2-D face solves: ~190 launches of a few microseconds per cycle;
    ! the 3-D component solves of small grids).  NDSM_HIP_NO_GRAPHS=1: always enqueue kernel by kernel (same bits).
    call get_environment_variable("NDSM_HIP_NO_GRAPHS", status=st)
    do l = 1, nl
      graphs(l) = (st /= 0) .and. graph_eligible(ss(l))
    end do
    do l = 1, nl
      rc = ndsmk_select_lane(int(l - 1, c_int)); if (rc /= 0) goto 800
      rc = ndsmk_lane_fence(int(l - 1, c_int), 0_c_int); if (rc /= 0) goto 800
      ! the caller's array is the "previous iterate" of the first comparison (:122)
      rc = ndsmk_d2d(ss(l)%prev, ss(l)%dl(1)%u, int(ss(l)%npts1, c_size_t) * R8); if (rc /= 0) goto 800
    end do
    ! Every lane gets its first cycle; from then on a lane's NEXT cycle is enqueued the moment its own metric has
    ! been read and found wanting - not after the metrics of all lanes have been collected: a lane never waits for
    ! the host to finish its round with the others (round 3: six face solves at 128^2 went from ~0.40 to ~0.3 ms per
    ! cycle each).  The lanes are independent, so the order in which the host serves them changes nothing they compute.
    itl = 0
    do l = 1, nl
      rc = enqueue_cycle(l, 1); if (rc /= 0) goto 800
    end do
    do while (any(active(1:nl)))
      do l = 1, nl
        if (.not. active(l)) cycle
        ! (the host serves whichever lane has finished, not the lanes in turn: a lane that is still busy is passed
        ! over - unless it is the only one left, then the host may as well sleep in the wait)
        if (count(active(1:nl)) > 1) then
          rc = ndsmk_lane_idle(int(l - 1, c_int))
          if (rc < 0) then
            rc = NDSMK_EARG; goto 800
          end if
          if (rc == 0) cycle
        end if
        rc = ndsmk_select_lane(int(l - 1, c_int)); if (rc /= 0) goto 800
        rc = ndsmk_diff_metrics_end(met); if (rc /= 0) goto 800
        if (ss(l)%use_max) then
          du = met(1)
        else
          du = met(2) / real(ss(l)%npts1, wp)
        end if
        itl(l) = itl(l) + 1
        ncycles(l) = itl(l)
        du_last(l) = du
        if (du < vc_tol) then         ! strict (:136)
          ierr(l) = 0
          active(l) = .false.
        else if (itl(l) >= nmax) then
          active(l) = .false.
        else
          rc = enqueue_cycle(l, itl(l) + 1); if (rc /= 0) goto 800
        end if
      end do
    end do
    rc = 0
800 continue
    rc2 = ndsmk_select_lane(-1_c_int)
    do l = 1, nl
      rc2 = ndsmk_lane_fence(int(l - 1, c_int), 1_c_int)
    end do

  contains

    ! cycle number `it` of lane l: its V-cycle + the metric pass, enqueued on the lane's stream (as a recorded
    ! graph where one exists or can be recorded now)
    function enqueue_cycle(l, it) result(rc)
      integer, intent(in) :: l, it
      integer(c_int) :: rc
      integer(c_int) :: rc2
      integer(ik) :: vc0
      integer :: q
      rc = ndsmk_select_lane(int(l - 1, c_int)); if (rc /= 0) return
      ! (a cycle recorded by an earlier call and still valid is replayed from the first round on)
      if (graphs(l) .and. (it >= 2 .or. graph_valid(ss(l), l - 1))) then
        if (.not. graph_valid(ss(l), l - 1)) then
          call drop_graph(ss(l))
          rc = ndsmk_select_lane(int(l - 1, c_int)); if (rc /= 0) return   ! (drop_graph drains the device only)
          ! Recording RUNS the host side of the cycle without running the device side: whatever the host changes
          ! on the way - the cycle counter, a level's array swapping places with its partner after an
          ! out-of-place pass - is put back afterwards, and a cycle that does not leave every array where it
          ! found it (an odd number of such passes: even ms on a fused level 1) cannot be replayed at all.
          call graph_stamp(ss(l), l - 1)
          keep_dl = ss(l)%dl
          rc = ndsmk_capture_begin()
          if (rc == 0) then
            vc0 = ss(l)%vcycles_done
            rc = mg_vcycle(ss(l))
            if (rc == 0) rc = ndsmk_diff_metrics_begin(ss(l)%dl(1)%u, ss(l)%prev, ss(l)%npts1, 1_c_int)
            rc2 = ndsmk_capture_end(ss(l)%graph)
            if (rc == 0) rc = rc2
            ss(l)%vcycles_done = vc0                        ! (recorded, not run: the replay below counts)
            if (rc == 0) then
              do q = 1, size(keep_dl)
                if (.not. c_associated(keep_dl(q)%u, ss(l)%dl(q)%u)) rc = NDSMK_EARG
              end do
            end if
            ss(l)%dl = keep_dl
          end if
          if (rc /= 0) then       ! recording is an optimisation: without it the round is enqueued as usual
            call drop_graph(ss(l))
            rc = ndsmk_select_lane(int(l - 1, c_int)); if (rc /= 0) return
            graphs(l) = .false.
          end if
        end if
        if (graphs(l)) then
          rc = ndsmk_graph_launch(ss(l)%graph); if (rc /= 0) return
          ss(l)%vcycles_done = ss(l)%vcycles_done + 1
          return
        end if
      end if
      rc = mg_vcycle(ss(l)); if (rc /= 0) return
      rc = ndsmk_diff_metrics_begin(ss(l)%dl(1)%u, ss(l)%prev, ss(l)%npts1, 1_c_int); if (rc /= 0) return
    end function

  end function

  ! ------------------------------------------------------------------
  ! Mixed-precision mode (BASELINE config[4]; csrc/mixed.hip has the algebra).
  ! ------------------------------------------------------------------
  ! May mg_solve keep the start-of-cycle iterate in place and let the last sweep take the metric?
  ! Level 1 must run on the out-of-place fused smoother for every sweep: 3-D, single domain, even
  ! nx, above the size where relax falls back to the in-place colour passes.  (NDSM_HIP_NO_TRACK
  ! set: never - A/B testing.)
  function mg_track_applies(s) result(ok)
    type(mg_solver), intent(in) :: s
    logical :: ok
    integer :: st
    ok = .false.
    if (s%ndim /= 3 .or. s%slab .or. s%ngrids < 2 .or. s%ms < 1) return
    if (s%lev(1)%g%all_neumann /= 0) return
    if (any(s%lev(1)%n(1:2) < 16) .or. s%lev(1)%n(3) < 8) return
    if (s%lev(1)%npts < 2_ik * 1024_ik * 1024_ik) return
    call get_environment_variable("NDSM_HIP_NO_TRACK", status=st)
    if (st == 0) return
    ok = .true.
  end function

  ! Is the solve run as fp64 residual + fp32 correction V-cycle?  Asked for, 3-D, single domain,
  ! >= 2 grids, ms >= 1, and level 1 within reach of the fp32 kernels (fused smoother, streamed
  ! restriction, tiled prolongation); otherwise the fp64 path runs.
  function mg_mixed_applies(s) result(ok)
    type(mg_solver), intent(in) :: s
    logical :: ok
    ok = .false.
    if (s%precision == 0 .or. s%ndim /= 3 .or. s%slab .or. s%ngrids < 2 .or. s%ms < 1) return
    if (s%lev(1)%g%all_neumann /= 0) return
    if (mod(s%lev(1)%n(1), 2) /= 0 .or. s%lev(1)%n(1) < 64 .or. s%lev(1)%n(2) < 16 .or. s%lev(1)%n(3) < 8) return
    if (any(s%lev(2)%n(1:3) < 16)) return
    if (iand(int(s%xf(1)%x%stream_ok), 2) == 0) return
    if (s%precision == 1 .and. s%lev(1)%npts < 6_ik * 1024_ik * 1024_ik) return
    ok = .true.
  end function

  ! V-cycles to tolerance as iterative refinement: r = rhs - L u in fp64 (stored fp32), one V-cycle
  ! on L e = r from e = 0 with level 1 in fp32 and levels >= 2 as always, u += e in fp64 - the
  ! update, the next residual and max|e| (= the reference's max|u_new - u_old|, update_u
  ! ndsm_multigrid_core.f90:1077-1122) are one pass.  e / its ping-pong partner live in the
  ! memory of the fp64 path's ualt, r and the e-equation's residual in its r, u' in its prev.
  function mg_solve_mixed(s, vc_tol, nmax, du_last, ncycles, ierr, hist) result(rc)
    type(mg_solver), intent(inout) :: s
    real(wp), intent(in) :: vc_tol
    integer, intent(in) :: nmax
    real(wp), intent(out) :: du_last
    integer, intent(out) :: ncycles, ierr
    real(wp), intent(inout), optional :: hist(:)
    integer(c_int) :: rc
    real(wp) :: met(2), du
    integer :: it
    integer(c_int) :: in_alt, force
    integer(c_size_t) :: half
    type(c_ptr) :: e, ealt, r32, rr32, tmp

    du = huge(du); ncycles = 0; ierr = 1
    du_last = du
    half = int(s%npts1, c_size_t) * 4_c_size_t
    e = s%dl(1)%ualt; ealt = dptr_offset(s%dl(1)%ualt, half)
    ! s%r is also the residual scratch of the fp64 levels >= 2 (first 8 npts(2) <= 4 npts(1) bytes): the
    ! e-equation's residual, dead once restricted, takes that half; its right-hand side the other
    rr32 = s%r; r32 = dptr_offset(s%r, half)
    force = merge(1_c_int, 0_c_int, s%precision == 2)
    rc = ndsmk_fill0(e, half); if (rc /= 0) return
    rc = ndsmk_update_residual_f32(s%lev(1)%g, s%dl(1)%u, c_null_ptr, rhs_of(s, 1), c_null_ptr, c_null_ptr, r32, met)
    if (rc /= 0) return
    do it = 1, nmax
      ! ---- one V-cycle on the correction (fine_to_coarse / coarse_to_fine, :482-684) ----
      rc = ndsmk_relax_f32(s%lev(1)%g, e, ealt, r32, int(s%ms, c_int), force, rr32, in_alt); if (rc /= 0) return
      if (in_alt /= 0) then
        tmp = e; e = ealt; ealt = tmp
      end if
      rc = ndsmk_restrict_f32(s%xf(1)%x, rr32, s%dl(2)%rhs, s%dl(2)%u); if (rc /= 0) return
      rc = mg_vcycle_from(s, 2); if (rc /= 0) return
      rc = mg_op(s, MG_OP_RELAX, 2, s%ms); if (rc /= 0) return
      rc = ndsmk_prolong_add_f32(s%xf(1)%x, s%dl(2)%u, e); if (rc /= 0) return
      rc = ndsmk_relax_f32(s%lev(1)%g, e, ealt, r32, int(s%ms, c_int), force, c_null_ptr, in_alt); if (rc /= 0) return
      if (in_alt /= 0) then
        tmp = e; e = ealt; ealt = tmp
      end if
      s%vcycles_done = s%vcycles_done + 1
      ! ---- u' = u + e ; next residual ; max|e| ; the zeroed partner becomes the next e ----
      rc = ndsmk_update_residual_f32(s%lev(1)%g, s%dl(1)%u, s%prev, rhs_of(s, 1), e, ealt, r32, met)
      if (rc /= 0) return
      tmp = s%dl(1)%u; s%dl(1)%u = s%prev; s%prev = tmp
      tmp = e; e = ealt; ealt = tmp
      if (s%use_max) then
        du = met(1)
      else
        du = met(2) / real(s%npts1, wp)
      end if
      ncycles = it
      if (present(hist)) then
        if (it <= size(hist)) hist(it) = du
      end if
      if (du < vc_tol) then         ! strict (:136)
        ierr = 0
        exit
      end if
    end do
    du_last = du
    rc = 0
  end function

end module ndsmh_mg
