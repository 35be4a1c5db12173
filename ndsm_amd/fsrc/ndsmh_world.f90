! libndsm_hip - z-slab decomposition of the finest level across GPUs (SURVEY 8e,
! BASELINE config 4): one process per GPU, level 1 split along z (x-y planes are
! contiguous in Fortran order), levels >= 2 on rank 0.
!
! The reference has no distributed mode; what must be preserved is its
! ARITHMETIC: a distributed V-cycle returns the same bits as the single-GPU
! one (tests/test_gpu_parity.py::test_slab_world_bitwise, loop-back mode).
!
! Per V-cycle, on the library stream, no host waits:
!   level 1 down : ms x { ghost exchange of u (2 planes per neighbour) ; fused
!                  red+black sweep of the owned planes }      - ONE exchange per
!                  full sweep: the kernel's z pipeline recomputes the red update
!                  of the first ghost plane, so the two exchanges per sweep of a
!                  colour-by-colour scheme collapse into one of twice the depth;
!                  exchange u ; residual ; exchange r (restriction taps reach up
!                  to g planes into the neighbour) ; restrict the coarse planes
!                  whose middle tap this rank owns ; ship them to rank 0
!   levels >= 2  : rank 0 runs the rest of the V-cycle (mg_vcycle_from(2)) and the
!                  ms pre-prolongation sweeps of level 2
!   level 1 up   : rank 0 ships each rank the coarse planes its slab interpolates
!                  from ; u += P u_c ; ms x { exchange ; sweep }
! Convergence  : local max / sum of |du| over owned planes, one 2-value
!                all-reduce; every rank takes the same decision.
!
! Transport: a "world" holds the slabs that live in THIS process.  Production:
! one slab per process, neighbours reached with RCCL send/recv (xGMI).  Loop-back
! (rank = -1 at creation): all slabs in one process on one GPU, neighbours
! reached with device-to-device copies - the same driver code, used to verify
! the decomposition bit for bit on a single GPU.
module ndsmh_world

  use, intrinsic :: iso_c_binding
  use ndsmh_iface
  use ndsmh_grid
  use ndsmh_mg
  implicit none
  private

  public :: mg_world, world_create, world_destroy, world_vcycle, world_solve, world_relax
  public :: world_upload, world_download, world_plan_only, world_set_params, world_dist_levels
  public :: ndsmk_dist_group_start, ndsmk_dist_group_end, ndsmk_dist_send, ndsmk_dist_recv   ! for ndsmh_wvecpot
  public :: ndsmk_dist_group_abort
  public :: world_set_precision

  interface
    function ndsmk_dist_size() bind(c, name="ndsmk_dist_size") result(n)
      import :: c_int
      integer(c_int) :: n
    end function
    function ndsmk_dist_rank() bind(c, name="ndsmk_dist_rank") result(n)
      import :: c_int
      integer(c_int) :: n
    end function
    function ndsmk_dist_group_start() bind(c, name="ndsmk_dist_group_start") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
    function ndsmk_dist_group_abort() bind(c, name="ndsmk_dist_group_abort") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
    function ndsmk_dist_group_end() bind(c, name="ndsmk_dist_group_end") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
    function ndsmk_dist_send(p, count, peer) bind(c, name="ndsmk_dist_send") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: p
      integer(c_size_t), value :: count
      integer(c_int), value :: peer
      integer(c_int) :: rc
    end function
    function ndsmk_dist_recv(p, count, peer) bind(c, name="ndsmk_dist_recv") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: p
      integer(c_size_t), value :: count
      integer(c_int), value :: peer
      integer(c_int) :: rc
    end function
    function ndsmk_dist_send_bytes(p, nbytes, peer) bind(c, name="ndsmk_dist_send_bytes") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: p
      integer(c_size_t), value :: nbytes
      integer(c_int), value :: peer
      integer(c_int) :: rc
    end function
    function ndsmk_dist_recv_bytes(p, nbytes, peer) bind(c, name="ndsmk_dist_recv_bytes") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: p
      integer(c_size_t), value :: nbytes
      integer(c_int), value :: peer
      integer(c_int) :: rc
    end function
    function ndsmk_dist_allreduce_max_sum(ms) bind(c, name="ndsmk_dist_allreduce_max_sum") result(rc)
      import :: c_double, c_int
      real(c_double), intent(inout) :: ms(2)
      integer(c_int) :: rc
    end function
  end interface

  integer(c_size_t), parameter :: R8 = 8_c_size_t

  type :: mg_world
    integer :: nranks = 1
    integer :: nlocal = 0            ! slabs held by this process
    integer :: first = 0             ! global rank of loc(1)
    logical :: rccl = .false.
    type(mg_solver), allocatable :: loc(:)
    type(slab_t), allocatable :: plan(:)     ! (0:nranks-1), identical on every rank
    integer :: ghost_depth = 0       ! this many ghost planes of u(1) per side match the neighbours' owned planes
    ! ... and this many match them in what a SMOOTHING pass reads of a ghost plane - its black points and its
    ! perimeter (exchange_sweep, halo.hip); >= ghost_depth
    integer :: ghost_sweep = 0
    type(c_ptr) :: hbuf(4) = c_null_ptr      ! packed planes: send up / receive from above / send down / receive from below
    integer(ik) :: hbuf_planes = 0           ! capacity of each, in packed planes
    integer(ik) :: nzg = 0
    ! The coarse problem.  Null: levels >= 2 live on rank 0 (restricted planes are gathered there,
    ! corrections scattered back).  Associated: level 2 is distributed as well - a world of its
    ! own over the same ranks, split where this level's restriction ownership puts the cuts, so
    ! every rank restricts straight into its own coarse slab and prolongs from it: the plane
    ! traffic to and from rank 0 moves one level down, where it is 8x smaller.
    type(mg_world), pointer :: child => null()
    integer :: level = 1             ! global level index of this world's slabs
    ! mixed-precision mode (world_solve_mixed): 0 = fp64 throughout; per local slab the fp32 arrays of
    ! the correction equation - e and its ping-pong partner (in the memory of the fp64 path's ualt),
    ! the equation's residual rr (in the fp64 residual scratch, dead once restricted) and its
    ! right-hand side r32 (an array of its own: on rank 0 the scratch also serves level 2)
    integer :: precision = 0
    type(c_ptr), allocatable :: e(:), ealt(:), rr32(:), r32(:)
    integer :: eghost = 0            ! ghost planes of e per side that match the neighbours
    ! world_solve: the iterate a V-cycle starts from must survive the cycle for update_u's
    ! max|u_new - u_old|.  Set before the cycle; the first smoothing pass (out of place: u -> ualt)
    ! then hands the untouched input buffer over to `prev` and takes prev's memory as the new
    ! ping-pong partner - three rotating buffers instead of a copy of the iterate per cycle.
    logical :: protect = .false.
    ! where each local slab interpolates its coarse-grid correction from (set by world_vcycle before the
    ! post-smoothing): whole coarse planes starting at global coarse plane psk0, psn of them
    type(c_ptr), allocatable :: psrc(:)
    integer, allocatable :: psk0(:), psn(:)
    ! world_solve asks (want_met) the last smoothing pass of the cycle to evaluate update_u's metric
    ! against prev while it stores the new iterate; met_done says it did (else a pass of its own)
    logical :: want_met = .false., met_done = .false.
  end type

contains

  ! hierarchy + tables on the host only: the plan every rank derives (CPU tests)
  function world_plan_only(nshape, qx, qy, qz, ngrids_req, nranks, plan, part, min_depth, ext) result(rc)
    integer(c_int32_t), intent(in) :: nshape(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    integer, intent(in) :: ngrids_req, nranks
    type(slab_t), allocatable, intent(out) :: plan(:)
    integer, intent(in), optional :: part(0:), min_depth
    real(wp), intent(in), optional :: ext(2, 3)
    integer(c_int) :: rc
    type(level_t), allocatable :: lev(:)
    type(axis_xfer_t) :: tz
    integer :: ng
    logical :: ok
    rc = NDSMK_EARG
    ng = ndsm_level_count(3, nshape)
    if (ngrids_req > 0) ng = min(ng, ngrids_req)
    if (ng < 2) return
    call build_levels(3, nshape, qx, qy, qz, ng, lev, ext)
    call build_axis_xfer(lev(1)%ax(3)%q, int(lev(1)%n(3)), lev(2)%ax(3)%q, int(lev(2)%n(3)), tz, ok)
    if (.not. ok) return
    call plan_slabs(int(lev(1)%n(3)), int(lev(2)%n(3)), tz, nranks, plan, ok, part, min_depth)
    if (ok) rc = 0
  end function

  ! rank < 0: loop-back world holding all nranks slabs in this process.
  ! part / min_depth / level: set by the recursion when this world is the distributed coarse level
  ! of another one.  How many levels get distributed: NDSM_HIP_DIST_LEVELS = k forces (up to) k
  ! where the shapes allow it, unset/0 = while a rank's share of the next level is still a level
  ! the fused kernels like (>= 2 M points and >= 16 planes per rank).
  recursive function world_create(w, nshape, qx, qy, qz, bcs, ngrids_req, nranks, rank, part, min_depth, level, &
                                  ext) result(rc)
    type(mg_world), intent(out) :: w
    integer(c_int32_t), intent(in) :: nshape(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    character(len=1), intent(in) :: bcs(:)
    integer, intent(in) :: ngrids_req, nranks, rank
    integer, intent(in), optional :: part(0:), min_depth, level
    real(wp), intent(in), optional :: ext(2, 3)   ! origin / extent of the finest level of the whole hierarchy
    integer(c_int) :: rc
    real(wp) :: ext1(2, 3)
    integer :: i, r, ng, want, st, reach
    integer, allocatable :: cpart(:)
    type(slab_t), allocatable :: cplan(:)
    integer(c_int32_t) :: n2(3)
    logical :: dist
    character(len=16) :: env
    type(level_t), allocatable :: lev(:)

    if (present(ext)) then
      ext1 = ext
    else
      ext1(1, :) = [minval(qx), minval(qy), minval(qz)]
      ext1(2, :) = [maxval(qx), maxval(qy), maxval(qz)] - ext1(1, :)
    end if
    rc = world_plan_only(nshape, qx, qy, qz, ngrids_req, nranks, w%plan, part, min_depth, ext1)
    if (rc /= 0) return
    w%nranks = nranks
    w%nzg = nshape(3)
    if (present(level)) w%level = level
    if (rank < 0) then
      w%rccl = .false.; w%first = 0; w%nlocal = nranks
    else
      rc = NDSMK_EARG
      if (rank >= nranks) return
      if (nranks > 1 .and. (ndsmk_dist_size() /= nranks .or. ndsmk_dist_rank() /= rank)) return
      w%rccl = (nranks > 1); w%first = rank; w%nlocal = 1
    end if

    ! ---- is the next level distributed too? -----------------------------
    ng = ndsm_level_count(3, nshape)
    if (ngrids_req > 0) ng = min(ng, ngrids_req)
    dist = .false.
    if (nranks > 1 .and. ng >= 3) then
      call build_levels(3, nshape, qx, qy, qz, ng, lev, ext1)
      n2 = lev(2)%n
      want = 0
      call get_environment_variable("NDSM_HIP_DIST_LEVELS", env, status=st)
      if (st == 0) read (env, *, iostat=st) want
      if (st /= 0) want = 0
      if (want > 0) then
        dist = w%level < want
      else
        dist = product(int(n2, ik)) / nranks >= 2_ik * 1024_ik * 1024_ik .and. &
               minval(w%plan(:)%ck1 - w%plan(:)%ck0) >= 16
      end if
      ! the coarse slabs run on the fused smoother, and the coarse hierarchy must be the same
      dist = dist .and. all(n2(1:2) >= 16) .and. ndsm_level_count(3, n2) >= ng - 1
      if (dist) then
        allocate (cpart(0:nranks))
        cpart(0) = 0
        reach = 0
        do r = 0, nranks - 1
          if (w%plan(r)%ck1 <= w%plan(r)%ck0 .or. w%plan(r)%ck0 /= cpart(r)) dist = .false.
          cpart(r + 1) = w%plan(r)%ck1
          reach = max(reach, w%plan(r)%ck0 - w%plan(r)%pk0, w%plan(r)%pk1 - w%plan(r)%ck1)
        end do
        if (cpart(nranks) /= n2(3)) dist = .false.
      end if
      if (dist) then      ! and the split must leave every rank a usable coarse slab
        dist = world_plan_only(n2, lev(2)%ax(1)%q, lev(2)%ax(2)%q, lev(2)%ax(3)%q, ng - 1, nranks, cplan, &
                               cpart, reach, ext1) == 0
      end if
    end if

    allocate (w%loc(w%nlocal))
    do i = 1, w%nlocal
      if (dist) then
        rc = mg_create(w%loc(i), 3, nshape, qx, qy, qz, bcs, ngrids_req, w%plan(w%first + i - 1), &
                       coarse_here=.false., ext=ext1)
      else
        rc = mg_create(w%loc(i), 3, nshape, qx, qy, qz, bcs, ngrids_req, w%plan(w%first + i - 1), ext=ext1)
      end if
      if (rc /= 0) return
    end do
    w%ghost_depth = 0; w%ghost_sweep = 0
    if (dist) then
      allocate (w%child)
      rc = world_create(w%child, n2, lev(2)%ax(1)%q, lev(2)%ax(2)%q, lev(2)%ax(3)%q, bcs, ng - 1, nranks, rank, &
                        cpart, reach, w%level + 1, ext1)
      if (rc /= 0) return
    end if
    rc = 0
  end function

  ! solver parameters of every slab solver, down the chain of distributed levels
  recursive subroutine world_set_params(w, ms, ex_tol, use_max, nmax_exact)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: ms, nmax_exact
    real(wp), intent(in) :: ex_tol
    logical, intent(in) :: use_max
    integer :: i
    do i = 1, w%nlocal
      w%loc(i)%ms = ms; w%loc(i)%ex_tol = ex_tol; w%loc(i)%use_max = use_max
      w%loc(i)%nmax_exact = nmax_exact
    end do
    if (associated(w%child)) call world_set_params(w%child, ms, ex_tol, use_max, nmax_exact)
  end subroutine

  ! number of distributed levels (1 = only the finest)
  recursive function world_dist_levels(w) result(n)
    type(mg_world), intent(in) :: w
    integer :: n
    n = 1
    if (associated(w%child)) n = 1 + world_dist_levels(w%child)
  end function

  recursive subroutine world_destroy(w)
    type(mg_world), intent(inout) :: w
    integer :: i
    integer(c_int) :: rcf
    if (associated(w%child)) then
      call world_destroy(w%child)
      deallocate (w%child)
      w%child => null()
    end if
    if (allocated(w%r32)) then
      do i = 1, size(w%r32)
        if (c_associated(w%r32(i))) rcf = ndsmk_free(w%r32(i))
      end do
      deallocate (w%r32)
    end if
    if (allocated(w%loc)) then
      do i = 1, size(w%loc)
        call mg_destroy(w%loc(i))
      end do
      deallocate (w%loc)
    end if
    if (allocated(w%plan)) deallocate (w%plan)
    do i = 1, 4
      if (c_associated(w%hbuf(i))) rcf = ndsmk_free(w%hbuf(i))
      w%hbuf(i) = c_null_ptr
    end do
    w%hbuf_planes = 0
    w%nlocal = 0
  end subroutine

  ! ------------------------------------------------------------------
  ! host <-> slab.  The host pointer addresses global plane `gz0` of a (nx,ny,*)
  ! array; planes [gz0, gz0+np) are copied where they intersect the slab's window
  ! (ghosts included on upload, owned planes only on download).
  ! ------------------------------------------------------------------
  function world_upload(w, ilocal, which, host, gz0, np) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: ilocal, which, gz0, np
    type(c_ptr), intent(in) :: host
    integer(c_int) :: rc
    integer :: a, b
    integer(ik) :: cnt
    type(c_ptr) :: dev
    rc = NDSMK_EARG
    if (ilocal < 1 .or. ilocal > w%nlocal) return
    associate (s => w%loc(ilocal))
      a = max(gz0, s%sl%k0, 0)
      b = min(gz0 + np, s%sl%k0 + s%sl%nloc, int(w%nzg))
      rc = 0
      if (b <= a) return
      dev = mg_level_ptr(s, 1, which, cnt)
      rc = ndsmk_h2d(dptr_offset(dev, int(a - s%sl%k0, c_size_t) * int(s%plane1, c_size_t) * R8), &
                     dptr_offset(host, int(a - gz0, c_size_t) * int(s%plane1, c_size_t) * R8), &
                     int(b - a, c_size_t) * int(s%plane1, c_size_t) * R8)
    end associate
    if (which == MG_BUF_U) then
      w%ghost_depth = 0; w%ghost_sweep = 0
    end if
    if (which == MG_BUF_RHS) call mg_mark_rhs_set(w%loc(ilocal))
  end function

  ! host receives the slab's OWNED planes [z0, z1), packed from `host` on
  function world_download(w, ilocal, which, host) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: ilocal, which
    type(c_ptr), intent(in) :: host
    integer(c_int) :: rc
    integer(ik) :: cnt
    type(c_ptr) :: dev
    rc = NDSMK_EARG
    if (ilocal < 1 .or. ilocal > w%nlocal) return
    associate (s => w%loc(ilocal))
      dev = mg_level_ptr(s, 1, which, cnt)
      rc = ndsmk_d2h(host, dptr_offset(dev, int(s%sl%g, c_size_t) * int(s%plane1, c_size_t) * R8), &
                     int(s%sl%z1 - s%sl%z0, c_size_t) * int(s%plane1, c_size_t) * R8)
    end associate
  end function

  ! ------------------------------------------------------------------
  ! ghost exchange of a level-1 sized array: `depth` planes per neighbour.
  ! which: MG_BUF_U or MG_BUF_R
  ! ------------------------------------------------------------------
  function exchange(w, which, depth) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: which, depth
    integer(c_int) :: rc, rc2
    integer :: i, r, g, nown
    integer(ik) :: cnt
    integer(c_size_t) :: pl, nb
    type(c_ptr) :: me, nbr

    rc = 0
    if (w%nranks == 1) return
    if (w%rccl) then
      rc = ndsmk_dist_group_start(); if (rc /= 0) return
    end if
    do i = 1, w%nlocal
      associate (s => w%loc(i))
        r = s%sl%rank; g = s%sl%g; nown = s%sl%z1 - s%sl%z0
        pl = int(s%plane1, c_size_t)
        nb = int(depth, c_size_t) * pl
        me = mg_level_ptr(s, 1, which, cnt)
        if (r < w%nranks - 1) then      ! upper neighbour r+1: my last owned planes -> its lower ghosts
          if (w%rccl) then
            rc = ndsmk_dist_send(dptr_offset(me, int(g + nown - depth, c_size_t) * pl * R8), nb, int(r + 1, c_int))
            if (rc /= 0) goto 800
            rc = ndsmk_dist_recv(dptr_offset(me, int(g + nown, c_size_t) * pl * R8), nb, int(r + 1, c_int))
            if (rc /= 0) goto 800
          else
            nbr = mg_level_ptr(w%loc(i + 1), 1, which, cnt)
            rc = ndsmk_d2d(dptr_offset(nbr, int(w%loc(i + 1)%sl%g - depth, c_size_t) * pl * R8), &
                           dptr_offset(me, int(g + nown - depth, c_size_t) * pl * R8), nb * R8)
            if (rc /= 0) goto 800
            rc = ndsmk_d2d(dptr_offset(me, int(g + nown, c_size_t) * pl * R8), &
                           dptr_offset(nbr, int(w%loc(i + 1)%sl%g, c_size_t) * pl * R8), nb * R8)
            if (rc /= 0) goto 800
          end if
        end if
        if (r > 0 .and. w%rccl) then    ! lower neighbour r-1 (loop-back: done from its side)
          rc = ndsmk_dist_send(dptr_offset(me, int(g, c_size_t) * pl * R8), nb, int(r - 1, c_int))
          if (rc /= 0) goto 800
          rc = ndsmk_dist_recv(dptr_offset(me, int(g - depth, c_size_t) * pl * R8), nb, int(r - 1, c_int))
          if (rc /= 0) goto 800
        end if
      end associate
    end do
800 continue   ! errors inside the group land here too: an open ncclGroupStart must be closed
    if (w%rccl) then
      rc2 = ndsmk_dist_group_end()
      if (rc == 0) rc = rc2
    end if
  end function

  ! coarse planes computed by every rank -> rhs(2) on rank 0 ; u(2) = 0 there
  function gather_coarse(w) result(rc)
    type(mg_world), intent(inout) :: w
    integer(c_int) :: rc, rc2
    integer :: i, r
    integer(c_size_t) :: pl2
    integer(ik) :: cnt
    type(c_ptr) :: dst

    rc = 0
    if (w%rccl) then
      rc = ndsmk_dist_group_start(); if (rc /= 0) return
    end if
    do i = 1, w%nlocal
      associate (s => w%loc(i))
        pl2 = int(s%plane2, c_size_t)
        if (s%sl%rank == 0) then
          dst = mg_level_ptr(s, 2, MG_BUF_RHS, cnt)
          do r = 0, w%nranks - 1
            if (w%plan(r)%ck1 <= w%plan(r)%ck0) cycle
            if (r == 0) then
              rc = ndsmk_d2d(dptr_offset(dst, int(w%plan(r)%ck0, c_size_t) * pl2 * R8), &
                             dptr_offset(s%cbuf, int(w%plan(r)%ck0 - s%sl%cb0, c_size_t) * pl2 * R8), &
                             int(w%plan(r)%ck1 - w%plan(r)%ck0, c_size_t) * pl2 * R8)
            else if (w%rccl) then
              rc = ndsmk_dist_recv(dptr_offset(dst, int(w%plan(r)%ck0, c_size_t) * pl2 * R8), &
                                   int(w%plan(r)%ck1 - w%plan(r)%ck0, c_size_t) * pl2, int(r, c_int))
            else
              rc = ndsmk_d2d(dptr_offset(dst, int(w%plan(r)%ck0, c_size_t) * pl2 * R8), &
                             dptr_offset(w%loc(r + 1)%cbuf, int(w%plan(r)%ck0 - w%plan(r)%cb0, c_size_t) * pl2 * R8), &
                             int(w%plan(r)%ck1 - w%plan(r)%ck0, c_size_t) * pl2 * R8)
            end if
            if (rc /= 0) goto 800
          end do
        else if (w%rccl .and. s%sl%ck1 > s%sl%ck0) then
          rc = ndsmk_dist_send(dptr_offset(s%cbuf, int(s%sl%ck0 - s%sl%cb0, c_size_t) * pl2 * R8), &
                               int(s%sl%ck1 - s%sl%ck0, c_size_t) * pl2, 0_c_int)
          if (rc /= 0) goto 800
        end if
      end associate
    end do
800 continue   ! errors inside the group land here too: an open ncclGroupStart must be closed
    if (w%rccl) then
      rc2 = ndsmk_dist_group_end()
      if (rc == 0) rc = rc2
    end if
    if (rc /= 0) return
    do i = 1, w%nlocal
      if (w%loc(i)%sl%rank == 0) then
        dst = mg_level_ptr(w%loc(i), 2, MG_BUF_U, cnt)
        rc = ndsmk_fill0(dst, int(cnt, c_size_t) * R8)      ! ndsm_multigrid_core.f90:557-558
      end if
    end do
  end function

  ! u(2) planes [pk0, pk1) of rank 0 -> every rank's cbuf
  function scatter_coarse(w) result(rc)
    type(mg_world), intent(inout) :: w
    integer(c_int) :: rc, rc2
    integer :: i, r
    integer(c_size_t) :: pl2
    integer(ik) :: cnt
    type(c_ptr) :: src

    rc = 0
    if (w%rccl) then
      rc = ndsmk_dist_group_start(); if (rc /= 0) return
    end if
    do i = 1, w%nlocal
      associate (s => w%loc(i))
        pl2 = int(s%plane2, c_size_t)
        if (s%sl%rank == 0) then
          src = mg_level_ptr(s, 2, MG_BUF_U, cnt)
          do r = 0, w%nranks - 1
            if (r == 0) then
              rc = ndsmk_d2d(dptr_offset(s%cbuf, int(w%plan(r)%pk0 - s%sl%cb0, c_size_t) * pl2 * R8), &
                             dptr_offset(src, int(w%plan(r)%pk0, c_size_t) * pl2 * R8), &
                             int(w%plan(r)%pk1 - w%plan(r)%pk0, c_size_t) * pl2 * R8)
            else if (w%rccl) then
              rc = ndsmk_dist_send(dptr_offset(src, int(w%plan(r)%pk0, c_size_t) * pl2 * R8), &
                                   int(w%plan(r)%pk1 - w%plan(r)%pk0, c_size_t) * pl2, int(r, c_int))
            else
              rc = ndsmk_d2d(dptr_offset(w%loc(r + 1)%cbuf, int(w%plan(r)%pk0 - w%plan(r)%cb0, c_size_t) * pl2 * R8), &
                             dptr_offset(src, int(w%plan(r)%pk0, c_size_t) * pl2 * R8), &
                             int(w%plan(r)%pk1 - w%plan(r)%pk0, c_size_t) * pl2 * R8)
            end if
            if (rc /= 0) goto 800
          end do
        else if (w%rccl) then
          rc = ndsmk_dist_recv(dptr_offset(s%cbuf, int(s%sl%pk0 - s%sl%cb0, c_size_t) * pl2 * R8), &
                               int(s%sl%pk1 - s%sl%pk0, c_size_t) * pl2, 0_c_int)
          if (rc /= 0) goto 800
        end if
      end associate
    end do
800 continue   ! errors inside the group land here too: an open ncclGroupStart must be closed
    if (w%rccl) then
      rc2 = ndsmk_dist_group_end()
      if (rc == 0) rc = rc2
    end if
  end function

  ! make at least `depth` ghost planes of u(1) per side current - sweep: in what a smoothing pass reads of them
  ! (half the bytes: exchange_sweep); otherwise whole planes
  function need_ghosts(w, depth, sweep) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: depth
    logical, intent(in), optional :: sweep
    integer(c_int) :: rc
    logical :: sw
    rc = 0
    sw = .false.
    if (present(sweep)) sw = sweep
    if (sw) then
      if (w%ghost_sweep >= depth) return
      rc = exchange_sweep(w, depth); if (rc /= 0) return
      w%ghost_sweep = max(w%ghost_sweep, depth)
    else
      if (w%ghost_depth >= depth) return
      rc = exchange(w, MG_BUF_U, depth); if (rc /= 0) return
      w%ghost_depth = depth
      w%ghost_sweep = max(w%ghost_sweep, depth)
    end if
  end function

  ! What a red-black sweep reads of a ghost plane is its black points as the neighbour last left them and the
  ! points no sweep updates (Dirichlet data on the x / y faces): the red points are recomputed by the pass before
  ! anything reads them.  So the halo of a smoothing pass travels as black points + perimeter - (nx ny) / 2 +
  ! 2 (nx + ny) doubles per plane - packed by halo.hip (RCCL: pack, one grouped send / receive per neighbour,
  ! unpack; loop-back worlds: the same points straight across).  NDSM_HIP_HALO_FULL=1: whole planes (A/B testing).
  function exchange_sweep(w, depth) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: depth
    integer(c_int) :: rc, rc2
    integer :: i, r, g, nown, st, q
    integer(c_int) :: nx, ny, fp, d
    integer(ik) :: cnt
    integer(c_size_t) :: pl, pk
    type(c_ptr) :: me, nbr
    logical, save :: first = .true., full = .false.
    rc = 0
    if (w%nranks == 1) return
    if (first) then
      call get_environment_variable("NDSM_HIP_HALO_FULL", status=st)
      full = (st == 0)
      first = .false.
    end if
    if (full) then
      rc = exchange(w, MG_BUF_U, depth)
      return
    end if
    d = int(depth, c_int)
    if (w%rccl) then
      associate (s => w%loc(1))
        nx = s%lev(1)%g%n(1); ny = s%lev(1)%g%n(2)
        pk = int(ndsmk_halo_packed_plane(nx, ny), c_size_t)
        if (w%hbuf_planes < depth) then
          do q = 1, 4
            if (c_associated(w%hbuf(q))) rc2 = ndsmk_free(w%hbuf(q))
            w%hbuf(q) = c_null_ptr
          end do
          w%hbuf_planes = max(int(depth, ik), int(w%plan(0)%g, ik))
          do q = 1, 4
            rc = ndsmk_alloc(w%hbuf(q), int(w%hbuf_planes, c_size_t) * pk * R8); if (rc /= 0) return
          end do
        end if
      end associate
    end if
    ! (1) RCCL: pack what goes up / down
    if (w%rccl) then
      associate (s => w%loc(1))
        r = s%sl%rank; g = s%sl%g; nown = s%sl%z1 - s%sl%z0
        pl = int(s%plane1, c_size_t); fp = s%lev(1)%g%first_par
        me = mg_level_ptr(s, 1, MG_BUF_U, cnt)
        if (r < w%nranks - 1) then
          rc = ndsmk_halo_pack(dptr_offset(me, int(g + nown - depth, c_size_t) * pl * R8), w%hbuf(1), nx, ny, d, &
                               int(s%sl%k0 + g + nown - depth, c_int), fp); if (rc /= 0) return
        end if
        if (r > 0) then
          rc = ndsmk_halo_pack(dptr_offset(me, int(g, c_size_t) * pl * R8), w%hbuf(3), nx, ny, d, &
                               int(s%sl%k0 + g, c_int), fp); if (rc /= 0) return
        end if
        rc = ndsmk_dist_group_start(); if (rc /= 0) return
        if (r < w%nranks - 1) then
          rc = ndsmk_dist_send(w%hbuf(1), int(depth, c_size_t) * pk, int(r + 1, c_int))
          if (rc == 0) rc = ndsmk_dist_recv(w%hbuf(2), int(depth, c_size_t) * pk, int(r + 1, c_int))
        end if
        if (rc == 0 .and. r > 0) then
          rc = ndsmk_dist_send(w%hbuf(3), int(depth, c_size_t) * pk, int(r - 1, c_int))
          if (rc == 0) rc = ndsmk_dist_recv(w%hbuf(4), int(depth, c_size_t) * pk, int(r - 1, c_int))
        end if
        rc2 = ndsmk_dist_group_end()       ! (an open group must be closed whatever happened inside)
        if (rc == 0) rc = rc2
        if (rc /= 0) return
        if (r < w%nranks - 1) then
          rc = ndsmk_halo_unpack(dptr_offset(me, int(g + nown, c_size_t) * pl * R8), w%hbuf(2), nx, ny, d, &
                                 int(s%sl%k0 + g + nown, c_int), fp); if (rc /= 0) return
        end if
        if (r > 0) then
          rc = ndsmk_halo_unpack(dptr_offset(me, int(g - depth, c_size_t) * pl * R8), w%hbuf(4), nx, ny, d, &
                                 int(s%sl%k0 + g - depth, c_int), fp); if (rc /= 0) return
        end if
      end associate
      return
    end if
    ! (2) loop-back: slab i and its upper neighbour i + 1 trade straight across
    do i = 1, w%nlocal - 1
      associate (s => w%loc(i), t => w%loc(i + 1))
        g = s%sl%g; nown = s%sl%z1 - s%sl%z0
        pl = int(s%plane1, c_size_t); fp = s%lev(1)%g%first_par
        nx = s%lev(1)%g%n(1); ny = s%lev(1)%g%n(2)
        me = mg_level_ptr(s, 1, MG_BUF_U, cnt)
        nbr = mg_level_ptr(t, 1, MG_BUF_U, cnt)
        ! my last owned planes -> its lower ghosts
        rc = ndsmk_halo_copy(dptr_offset(nbr, int(t%sl%g - depth, c_size_t) * pl * R8), &
                             dptr_offset(me, int(g + nown - depth, c_size_t) * pl * R8), nx, ny, d, &
                             int(s%sl%k0 + g + nown - depth, c_int), fp); if (rc /= 0) return
        ! its first owned planes -> my upper ghosts
        rc = ndsmk_halo_copy(dptr_offset(me, int(g + nown, c_size_t) * pl * R8), &
                             dptr_offset(nbr, int(t%sl%g, c_size_t) * pl * R8), nx, ny, d, &
                             int(s%sl%k0 + g + nown, c_int), fp); if (rc /= 0) return
      end associate
    end do
  end function

  ! May a pass with exchange depth d run its interior while the halo travels?  More than one rank,
  ! every local slab thick enough to have an interior worth a launch, not switched off
  ! (NDSM_HIP_OVERLAP=0).
  function overlap_ok(w, d) result(ok)
    type(mg_world), intent(in) :: w
    integer, intent(in) :: d
    logical :: ok
    integer :: i, st
    character(len=8) :: env
    logical :: forced
    ok = .false.
    if (w%nranks < 2) return
    forced = .false.
    call get_environment_variable("NDSM_HIP_OVERLAP", env, status=st)
    if (st == 0) then
      if (env(1:1) == '0') return
      forced = .true.
    end if
    do i = 1, w%nlocal
      if (w%loc(i)%sl%z1 - w%loc(i)%sl%z0 - 2 * d < 8) return
    end do
    ! Splitting a pass into interior + two edge launches costs ~0.1 ms per slab (the edge launches
    ! walk 8 warm-up planes for d owned ones; measured in loop-back), i.e. what an xGMI link
    ! (~150 GB/s) needs for ~16 MB: below that the exchange is cheaper exposed than hidden.
    if (.not. forced) then
      if (int(d, ik) * w%loc(1)%plane1 * 8_ik < 16_ik * 1024_ik * 1024_ik) return
    end if
    ok = .true.
  end function

  ! nsweeps sweeps of every local slab.  A sweep consumes two ghost planes per side (red needs
  ! black of the neighbour plane, black needs that red), so a halo exchange of depth 4 feeds a
  ! two-sweep pass of the temporally blocked kernel: half the messages, half the passes over HBM.
  ! with_res: the residual rides on the last sweep (depth 3: one more plane for its stencil).
  ! prolong: the coarse-grid correction (w%psrc) is still to be added to u.  Where the first pass can
  ! fold it in (two sweeps, Laplace problem) it reads the UNCORRECTED u - ghosts included, their
  ! correction is formed locally from the same coarse planes the neighbour uses - and adds P u_c to
  ! every plane as it loads it; otherwise the stand-alone kernel corrects the owned planes first.
  function world_relax(w, nsweeps, with_res, prolong, last) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: nsweeps
    logical, intent(in), optional :: with_res, prolong, last
    integer(c_int) :: rc
    integer :: left, n, i, d, mt
    logical :: res, two_ok, pro, pend, fin, domet
    type(c_ptr) :: tmp
    rc = 0
    res = .false.
    if (present(with_res)) res = with_res
    pend = .false.
    if (present(prolong)) pend = prolong
    fin = .false.
    if (present(last)) fin = last
    w%met_done = .false.
    two_ok = w%plan(0)%g >= 4
    left = nsweeps
    if (pend .and. left <= 0) then
      rc = prolong_alone(w); if (rc /= 0) return
    end if
    do while (left > 0)
      n = 1
      if (two_ok .and. left >= 2 .and. .not. (res .and. left == 2)) n = 2
      ! an odd number of sweeps that starts with the correction: 1 + 2 + 2 ..., the correction on the ONE-sweep
      ! pass (bound by memory, it has the instruction slots the interpolation needs; a two-sweep pass does not)
      if (pend .and. .not. res .and. left >= 3 .and. mod(left, 2) == 1) n = 1
      pro = .false.
      if (pend) then
        pro = .not. (res .and. left == 1)
        do i = 1, w%nlocal
          pro = pro .and. mg_window_prolong_ok(w%loc(i), n)
        end do
        if (.not. pro) then
          rc = prolong_alone(w); if (rc /= 0) return
        end if
        pend = .false.
      end if
      ! the pass that ends the cycle carries the convergence metric (mt: 1 = first launch of the
      ! pass in this process, 2 = add to it)
      domet = fin .and. left == n .and. .not. pro .and. .not. res
      if (domet) then
        do i = 1, w%nlocal
          domet = domet .and. mg_window_metric_ok(w%loc(i))
        end do
      end if
      mt = 0
      if (domet) mt = 1
      if (res .and. left == 1 .and. w%ghost_sweep < 3 .and. overlap_ok(w, 3)) then
        ! the sweep + residual pass in three pieces as well: its exchange (3 planes: one more for the
        ! residual's stencil) on the communication stream behind the planes that do not need it
        d = 3
        rc = ndsmk_stream_fence(0_c_int, 1_c_int); if (rc /= 0) return
        rc = ndsmk_select_stream(1_c_int); if (rc /= 0) return
        rc = exchange_sweep(w, d)
        i = ndsmk_select_stream(0_c_int)
        if (rc /= 0) return
        do i = 1, w%nlocal
          associate (s => w%loc(i))
            rc = mg_relax_res_window(s, int(s%lev(1)%g%zown0) + d, int(s%lev(1)%g%zown1) - d); if (rc /= 0) return
          end associate
        end do
        rc = ndsmk_stream_fence(1_c_int, 0_c_int); if (rc /= 0) return
        do i = 1, w%nlocal
          associate (s => w%loc(i))
            rc = mg_relax_res_window(s, int(s%lev(1)%g%zown0), int(s%lev(1)%g%zown0) + d); if (rc /= 0) return
            rc = mg_relax_res_window(s, int(s%lev(1)%g%zown1) - d, int(s%lev(1)%g%zown1)); if (rc /= 0) return
            call mg_swap_u(s)
          end associate
        end do
      else if (res .and. left == 1) then
        rc = need_ghosts(w, 3, sweep=.true.); if (rc /= 0) return
        do i = 1, w%nlocal
          rc = mg_op(w%loc(i), MG_OP_RELAX_RES_FUSED, 1, 1); if (rc /= 0) return
        end do
      else if (w%ghost_sweep < 2 * n .and. overlap_ok(w, 2 * n)) then
        ! the halo exchange of this pass on the communication stream, the planes that do not need
        ! it meanwhile, the 2n planes next to each neighbour once it has arrived (out of place:
        ! all three launches read u and write disjoint planes of its partner)
        d = 2 * n
        rc = ndsmk_stream_fence(0_c_int, 1_c_int); if (rc /= 0) return
        rc = ndsmk_select_stream(1_c_int); if (rc /= 0) return
        rc = exchange_sweep(w, d)
        i = ndsmk_select_stream(0_c_int)
        if (rc /= 0) return
        do i = 1, w%nlocal
          associate (s => w%loc(i))
            rc = window_pass(w, i, n, int(s%lev(1)%g%zown0) + d, int(s%lev(1)%g%zown1) - d, pro, mt); if (rc /= 0) return
            if (mt == 1) mt = 2
          end associate
        end do
        rc = ndsmk_stream_fence(1_c_int, 0_c_int); if (rc /= 0) return
        do i = 1, w%nlocal
          associate (s => w%loc(i))
            rc = window_pass(w, i, n, int(s%lev(1)%g%zown0), int(s%lev(1)%g%zown0) + d, pro, mt); if (rc /= 0) return
            rc = window_pass(w, i, n, int(s%lev(1)%g%zown1) - d, int(s%lev(1)%g%zown1), pro, mt); if (rc /= 0) return
            call mg_swap_u(s)
          end associate
        end do
      else
        rc = need_ghosts(w, 2 * n, sweep=.true.); if (rc /= 0) return
        do i = 1, w%nlocal
          if (pro .or. domet) then
            rc = window_pass(w, i, n, int(w%loc(i)%lev(1)%g%zown0), int(w%loc(i)%lev(1)%g%zown1), pro, mt)
            if (rc /= 0) return
            if (mt == 1) mt = 2
            call mg_swap_u(w%loc(i))
          else
            rc = mg_op(w%loc(i), MG_OP_RELAX_FUSED, 1, n); if (rc /= 0) return
          end if
        end do
      end if
      if (domet) w%met_done = .true.
      if (w%protect) then              ! first pass of a solve-loop cycle: keep its input (see mg_world)
        do i = 1, w%nlocal
          tmp = w%loc(i)%dl(1)%ualt; w%loc(i)%dl(1)%ualt = w%loc(i)%prev; w%loc(i)%prev = tmp
        end do
        w%protect = .false.
      end if
      w%ghost_depth = 0; w%ghost_sweep = 0
      left = left - n
    end do
  end function

  ! one fused pass over local planes [z0, z1) of slab i, u -> ualt, optionally interpolating
  function window_pass(w, i, n, z0, z1, pro, mt) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: i, n, z0, z1, mt
    logical, intent(in) :: pro
    integer(c_int) :: rc
    if (pro) then
      rc = mg_relax_window(w%loc(i), n, z0, z1, w%psrc(i), w%psk0(i), w%psn(i))
    else
      rc = mg_relax_window(w%loc(i), n, z0, z1, met=mt)
    end if
  end function

  ! u += P u_c on the owned planes with the stand-alone kernel; the ghosts are stale afterwards
  function prolong_alone(w) result(rc)
    type(mg_world), intent(inout) :: w
    integer(c_int) :: rc
    integer :: i
    rc = 0
    do i = 1, w%nlocal
      rc = mg_slab_prolong(w%loc(i), w%psrc(i), w%psk0(i)); if (rc /= 0) return
    end do
    w%ghost_depth = 0; w%ghost_sweep = 0
  end function

  recursive function world_vcycle(w) result(rc)
    type(mg_world), intent(inout) :: w
    integer(c_int) :: rc
    integer :: i
    logical :: split
    if (.not. allocated(w%psrc)) allocate (w%psrc(w%nlocal), w%psk0(w%nlocal), w%psn(w%nlocal))

    ! ---- level 1, downwards (fine_to_coarse, ndsm_multigrid_core.f90:482-560)
    if (w%loc(1)%ms >= 1 .and. w%plan(0)%g >= 3) then
      rc = world_relax(w, w%loc(1)%ms, .true.); if (rc /= 0) return
    else
      rc = world_relax(w, w%loc(1)%ms); if (rc /= 0) return
      rc = need_ghosts(w, 2); if (rc /= 0) return   ! the residual reads one ghost plane of u
      do i = 1, w%nlocal
        rc = mg_op(w%loc(i), MG_OP_RESIDUAL, 1, 1); if (rc /= 0) return
      end do
    end if
    ! the residual crosses the slab cuts (the restriction of a rank's first and last coarse planes reads up to g
    ! fine planes of its neighbours) - behind the restriction of the coarse planes that do not need it where the
    ! message is worth it (split = .true.: those planes are restricted below, the others after the fence)
    split = overlap_ok(w, w%plan(0)%g)
    do i = 1, w%nlocal
      split = split .and. w%loc(i)%sl%ci1 - w%loc(i)%sl%ci0 >= 8
    end do
    if (split) then
      rc = ndsmk_stream_fence(0_c_int, 1_c_int); if (rc /= 0) return
      rc = ndsmk_select_stream(1_c_int); if (rc /= 0) return
      rc = exchange(w, MG_BUF_R, w%plan(0)%g)
      i = ndsmk_select_stream(0_c_int)
      if (rc /= 0) return
    else
      rc = exchange(w, MG_BUF_R, w%plan(0)%g); if (rc /= 0) return
    end if
    if (associated(w%child)) then
      ! ---- the next level is distributed as well: every rank restricts its coarse planes straight
      ! into its own slab of the child's right-hand side, the child runs its part of the cycle
      ! (V-cycle from its level + the sweeps that precede an interpolation, :642-644), and the
      ! correction is interpolated from the child's own slab (its ghosts made current first)
      associate (c => w%child)
        if (split) then
          do i = 1, w%nlocal
            rc = mg_slab_restrict(w%loc(i), c%loc(i)%dl(1)%rhs, c%loc(i)%sl%k0, w%loc(i)%sl%ci0, w%loc(i)%sl%ci1)
            if (rc /= 0) return
          end do
          rc = ndsmk_stream_fence(1_c_int, 0_c_int); if (rc /= 0) return
        end if
        do i = 1, w%nlocal
          if (split) then
            rc = mg_slab_restrict(w%loc(i), c%loc(i)%dl(1)%rhs, c%loc(i)%sl%k0, kb=w%loc(i)%sl%ci0); if (rc /= 0) return
            rc = mg_slab_restrict(w%loc(i), c%loc(i)%dl(1)%rhs, c%loc(i)%sl%k0, ka=w%loc(i)%sl%ci1); if (rc /= 0) return
          else
            rc = mg_slab_restrict(w%loc(i), c%loc(i)%dl(1)%rhs, c%loc(i)%sl%k0); if (rc /= 0) return
          end if
          rc = ndsmk_fill0(c%loc(i)%dl(1)%u, int(c%loc(i)%npts1, c_size_t) * R8)   ! :557-558
          if (rc /= 0) return
          call mg_mark_rhs_set(c%loc(i))
          c%loc(i)%ms = w%loc(i)%ms
        end do
        rc = exchange(c, MG_BUF_RHS, c%plan(0)%g); if (rc /= 0) return   ! redundant ghost updates read rhs there
        c%ghost_depth = c%plan(0)%g; c%ghost_sweep = c%plan(0)%g                                        ! u = 0 everywhere: ghosts are current
        rc = world_vcycle(c); if (rc /= 0) return
        rc = world_relax(c, c%loc(1)%ms); if (rc /= 0) return
        rc = need_ghosts(c, c%plan(0)%g); if (rc /= 0) return
        do i = 1, w%nlocal
          w%psrc(i) = c%loc(i)%dl(1)%u; w%psk0(i) = c%loc(i)%sl%k0; w%psn(i) = c%loc(i)%sl%nloc
        end do
      end associate
    else
      if (split) then
        do i = 1, w%nlocal
          rc = mg_slab_restrict(w%loc(i), ka=w%loc(i)%sl%ci0, kb=w%loc(i)%sl%ci1); if (rc /= 0) return
        end do
        rc = ndsmk_stream_fence(1_c_int, 0_c_int); if (rc /= 0) return
        do i = 1, w%nlocal
          rc = mg_slab_restrict(w%loc(i), kb=w%loc(i)%sl%ci0); if (rc /= 0) return
          rc = mg_slab_restrict(w%loc(i), ka=w%loc(i)%sl%ci1); if (rc /= 0) return
        end do
      else
        do i = 1, w%nlocal
          rc = mg_slab_restrict(w%loc(i)); if (rc /= 0) return
        end do
      end if
      rc = gather_coarse(w); if (rc /= 0) return

      ! ---- levels >= 2 on rank 0 ------------------------------------------
      do i = 1, w%nlocal
        if (w%loc(i)%sl%rank /= 0) cycle
        if (w%loc(i)%ngrids == 2) then
          rc = mg_op(w%loc(i), MG_OP_EXACT, 2, 1); if (rc /= 0) return
        else
          rc = mg_vcycle_from(w%loc(i), 2); if (rc /= 0) return
        end if
        rc = mg_op(w%loc(i), MG_OP_RELAX, 2, w%loc(i)%ms); if (rc /= 0) return   ! :642-644
      end do

      ! ---- level 1, upwards (coarse_to_fine, :593-684) --------------------
      rc = scatter_coarse(w); if (rc /= 0) return
      do i = 1, w%nlocal
        w%psrc(i) = w%loc(i)%cbuf; w%psk0(i) = w%loc(i)%sl%cb0; w%psn(i) = w%loc(i)%sl%cb1 - w%loc(i)%sl%cb0
      end do
    end if
    w%ghost_depth = 0; w%ghost_sweep = 0
    rc = world_relax(w, w%loc(1)%ms, prolong=.true., last=w%want_met)
  end function

  ! V-cycles to tolerance; every rank returns the same du history
  function world_solve(w, vc_tol, nmax, du_last, ncycles, ierr, hist) result(rc)
    type(mg_world), intent(inout) :: w
    real(wp), intent(in) :: vc_tol
    integer, intent(in) :: nmax
    real(wp), intent(out) :: du_last
    integer, intent(out) :: ncycles, ierr
    real(wp), intent(inout), optional :: hist(:)
    integer(c_int) :: rc
    real(wp) :: met(2), tot(2), du
    integer :: it, i
    integer(c_size_t) :: off, nb
    integer(ik) :: nown
    logical :: rot

    if (w%precision /= 0) then
      rc = world_solve_mixed(w, vc_tol, nmax, du_last, ncycles, ierr, hist)
      return
    end if
    du = huge(du); ncycles = 0; ierr = 1
    rot = w%loc(1)%ms >= 1           ! a cycle without sweeps has no out-of-place pass to rotate on
    do it = 1, nmax
      if (rot) then
        w%protect = .true.
      else
        do i = 1, w%nlocal
          rc = ndsmk_d2d(w%loc(i)%prev, w%loc(i)%dl(1)%u, int(w%loc(i)%npts1, c_size_t) * R8); if (rc /= 0) return
        end do
      end if
      w%want_met = rot              ! the rotation is what leaves the cycle's starting iterate in prev
      rc = world_vcycle(w)
      w%want_met = .false.
      if (rc /= 0) return
      w%protect = .false.
      tot = 0
      if (w%met_done) then            ! left on the device by the cycle's last smoothing pass
        rc = ndsmk_fetch_fused_metric(tot); if (rc /= 0) return
      end if
      do i = 1, merge(0, w%nlocal, w%met_done)
        associate (s => w%loc(i))
          off = int(s%sl%g, c_size_t) * int(s%plane1, c_size_t) * R8
          nown = int(s%sl%z1 - s%sl%z0, ik) * s%plane1
          nb = int(nown, c_size_t)
          rc = ndsmk_diff_metrics(dptr_offset(s%dl(1)%u, off), dptr_offset(s%prev, off), nown, 0_c_int, met)
          if (rc /= 0) return
          tot(1) = max(tot(1), met(1)); tot(2) = tot(2) + met(2)
        end associate
      end do
      if (w%rccl) then
        rc = ndsmk_dist_allreduce_max_sum(tot); if (rc /= 0) return
      end if
      if (w%loc(1)%use_max) then
        du = tot(1)
      else
        du = tot(2) / (real(w%nzg, wp) * real(w%loc(1)%plane1, wp))
      end if
      ncycles = it
      if (present(hist)) then
        if (it <= size(hist)) hist(it) = du
      end if
      if (du < vc_tol) then
        ierr = 0
        exit
      end if
    end do
    du_last = du
    rc = 0
  end function

  ! ------------------------------------------------------------------
  ! Mixed precision on z-slabs (BASELINE config[4]): the iterative refinement of mg_solve_mixed -
  ! fp64 residual, ONE V-cycle on the correction e with level 1 in fp32, u += e in fp64 - with
  ! level 1 cut into slabs.  mode /= 0 asks for it; returns whether world_solve will run it
  ! (every slab must be in reach of the fp32 kernels: nx even, >= 64 x 16 per plane, streamed
  ! restriction; otherwise the fp64 path stays).  Same arithmetic as the single-domain mode: the
  ! same launches, cut along z - bit-identical to it (tests).
  ! ------------------------------------------------------------------
  function world_set_precision(w, mode) result(on)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: mode
    logical :: on
    integer :: i
    integer(c_int) :: rc
    integer(c_size_t) :: half
    on = .false.
    w%precision = 0
    if (mode == 0 .or. w%nranks < 2 .or. w%plan(0)%g < 4) return
    do i = 1, w%nlocal
      if (.not. mg_mixed_slab_ok(w%loc(i))) return
    end do
    if (.not. allocated(w%e)) then
      allocate (w%e(w%nlocal), w%ealt(w%nlocal), w%rr32(w%nlocal), w%r32(w%nlocal))
      w%r32 = c_null_ptr
      do i = 1, w%nlocal
        half = int(w%loc(i)%npts1, c_size_t) * 4_c_size_t
        rc = ndsmk_alloc(w%r32(i), half); if (rc /= 0) return
      end do
    end if
    w%precision = mode
    on = .true.
  end function

  ! ghost exchange of a level-1 sized fp32 array (p(i): its base on local slab i)
  function exchange_f32(w, p, depth) result(rc)
    type(mg_world), intent(inout) :: w
    type(c_ptr), intent(in) :: p(:)
    integer, intent(in) :: depth
    integer(c_int) :: rc, rc2
    integer :: i, r, g, nown
    integer(c_size_t) :: pl, nb
    integer(c_size_t), parameter :: R4 = 4_c_size_t
    rc = 0
    if (w%nranks == 1) return
    if (w%rccl) then
      rc = ndsmk_dist_group_start(); if (rc /= 0) return
    end if
    do i = 1, w%nlocal
      associate (s => w%loc(i))
        r = s%sl%rank; g = s%sl%g; nown = s%sl%z1 - s%sl%z0
        pl = int(s%plane1, c_size_t) * R4
        nb = int(depth, c_size_t) * pl
        if (r < w%nranks - 1) then
          if (w%rccl) then
            rc = ndsmk_dist_send_bytes(dptr_offset(p(i), int(g + nown - depth, c_size_t) * pl), nb, int(r + 1, c_int))
            if (rc /= 0) goto 800
            rc = ndsmk_dist_recv_bytes(dptr_offset(p(i), int(g + nown, c_size_t) * pl), nb, int(r + 1, c_int))
            if (rc /= 0) goto 800
          else
            rc = ndsmk_d2d(dptr_offset(p(i + 1), int(w%loc(i + 1)%sl%g - depth, c_size_t) * pl), &
                           dptr_offset(p(i), int(g + nown - depth, c_size_t) * pl), nb)
            if (rc /= 0) goto 800
            rc = ndsmk_d2d(dptr_offset(p(i), int(g + nown, c_size_t) * pl), &
                           dptr_offset(p(i + 1), int(w%loc(i + 1)%sl%g, c_size_t) * pl), nb)
            if (rc /= 0) goto 800
          end if
        end if
        if (r > 0 .and. w%rccl) then
          rc = ndsmk_dist_send_bytes(dptr_offset(p(i), int(g, c_size_t) * pl), nb, int(r - 1, c_int))
          if (rc /= 0) goto 800
          rc = ndsmk_dist_recv_bytes(dptr_offset(p(i), int(g - depth, c_size_t) * pl), nb, int(r - 1, c_int))
          if (rc /= 0) goto 800
        end if
      end associate
    end do
800 continue   ! errors inside the group land here too: an open ncclGroupStart must be closed
    if (w%rccl) then
      rc2 = ndsmk_dist_group_end()
      if (rc == 0) rc = rc2
    end if
  end function

  function need_e_ghosts(w, depth) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: depth
    integer(c_int) :: rc
    rc = 0
    if (w%eghost >= depth) return
    rc = exchange_f32(w, w%e, depth); if (rc /= 0) return
    w%eghost = depth
  end function

  ! nsweeps fp32 sweeps of L e = r32 on every local slab, passes of two with a depth-4 exchange as in
  ! world_relax; with_res: the e-equation's residual (rr32) rides on the last sweep
  function world_relax_f32(w, nsweeps, with_res) result(rc)
    type(mg_world), intent(inout) :: w
    integer, intent(in) :: nsweeps
    logical, intent(in) :: with_res
    integer(c_int) :: rc
    integer :: left, n, i
    integer(c_int) :: in_alt
    type(c_ptr) :: rout, tmp
    rc = 0
    left = nsweeps
    do while (left > 0)
      n = 1
      if (left >= 2 .and. .not. (with_res .and. left == 2)) n = 2
      rout = c_null_ptr
      if (with_res .and. left == 1) then
        rc = need_e_ghosts(w, 3); if (rc /= 0) return
      else
        rc = need_e_ghosts(w, 2 * n); if (rc /= 0) return
      end if
      do i = 1, w%nlocal
        if (with_res .and. left == 1) rout = w%rr32(i)
        rc = ndsmk_relax_f32(w%loc(i)%lev(1)%g, w%e(i), w%ealt(i), w%r32(i), int(n, c_int), 1_c_int, rout, in_alt)
        if (rc /= 0) return
        if (in_alt /= 0) then
          tmp = w%e(i); w%e(i) = w%ealt(i); w%ealt(i) = tmp
        end if
      end do
      w%eghost = 0
      left = left - n
    end do
  end function

  function world_solve_mixed(w, vc_tol, nmax, du_last, ncycles, ierr, hist) result(rc)
    type(mg_world), intent(inout) :: w
    real(wp), intent(in) :: vc_tol
    integer, intent(in) :: nmax
    real(wp), intent(out) :: du_last
    integer, intent(out) :: ncycles, ierr
    real(wp), intent(inout), optional :: hist(:)
    integer(c_int) :: rc
    real(wp) :: met(2), tot(2), du
    integer :: it, i, g
    integer(c_size_t) :: half, pl4
    type(c_ptr) :: tmp

    du = huge(du); ncycles = 0; ierr = 1
    du_last = du
    g = w%plan(0)%g
    do i = 1, w%nlocal
      associate (s => w%loc(i))
        half = int(s%npts1, c_size_t) * 4_c_size_t
        w%e(i) = s%dl(1)%ualt; w%ealt(i) = dptr_offset(s%dl(1)%ualt, half)
        w%rr32(i) = s%r
        rc = ndsmk_fill0(s%dl(1)%ualt, 2_c_size_t * half); if (rc /= 0) return     ! e = 0, ghosts included
      end associate
    end do
    w%eghost = g
    ! r32 = rhs - L u (fp64 arithmetic): one ghost plane of u, then the ghosts of r32
    rc = need_ghosts(w, 1); if (rc /= 0) return
    do i = 1, w%nlocal
      associate (s => w%loc(i))
        rc = ndsmk_update_residual_f32(s%lev(1)%g, s%dl(1)%u, c_null_ptr, rhs_of(s, 1), c_null_ptr, c_null_ptr, &
                                       w%r32(i), met)
        if (rc /= 0) return
      end associate
    end do
    rc = exchange_f32(w, w%r32, g); if (rc /= 0) return

    do it = 1, nmax
      ! ---- one V-cycle on the correction (world_vcycle with level 1 in fp32) ----
      rc = world_relax_f32(w, w%loc(1)%ms, .true.); if (rc /= 0) return
      rc = exchange_f32(w, w%rr32, g); if (rc /= 0) return
      if (associated(w%child)) then
        associate (c => w%child)
          do i = 1, w%nlocal
            rc = mg_slab_restrict_f32(w%loc(i), w%rr32(i), c%loc(i)%dl(1)%rhs, c%loc(i)%sl%k0); if (rc /= 0) return
            rc = ndsmk_fill0(c%loc(i)%dl(1)%u, int(c%loc(i)%npts1, c_size_t) * R8); if (rc /= 0) return
            call mg_mark_rhs_set(c%loc(i))
            c%loc(i)%ms = w%loc(i)%ms
          end do
          rc = exchange(c, MG_BUF_RHS, c%plan(0)%g); if (rc /= 0) return
          c%ghost_depth = c%plan(0)%g; c%ghost_sweep = c%plan(0)%g
          rc = world_vcycle(c); if (rc /= 0) return
          rc = world_relax(c, c%loc(1)%ms); if (rc /= 0) return
          rc = need_ghosts(c, c%plan(0)%g); if (rc /= 0) return
          do i = 1, w%nlocal
            rc = mg_slab_prolong_f32(w%loc(i), w%e(i), c%loc(i)%dl(1)%u, c%loc(i)%sl%k0); if (rc /= 0) return
          end do
        end associate
      else
        do i = 1, w%nlocal
          rc = mg_slab_restrict_f32(w%loc(i), w%rr32(i)); if (rc /= 0) return
        end do
        rc = gather_coarse(w); if (rc /= 0) return
        do i = 1, w%nlocal
          if (w%loc(i)%sl%rank /= 0) cycle
          if (w%loc(i)%ngrids == 2) then
            rc = mg_op(w%loc(i), MG_OP_EXACT, 2, 1); if (rc /= 0) return
          else
            rc = mg_vcycle_from(w%loc(i), 2); if (rc /= 0) return
          end if
          rc = mg_op(w%loc(i), MG_OP_RELAX, 2, w%loc(i)%ms); if (rc /= 0) return
        end do
        rc = scatter_coarse(w); if (rc /= 0) return
        do i = 1, w%nlocal
          rc = mg_slab_prolong_f32(w%loc(i), w%e(i)); if (rc /= 0) return
        end do
      end if
      w%eghost = 0
      rc = world_relax_f32(w, w%loc(1)%ms, .false.); if (rc /= 0) return

      ! ---- u' = u + e ; e' = 0 ; next residual ; max|e| ----
      rc = need_e_ghosts(w, 1); if (rc /= 0) return
      rc = need_ghosts(w, 1); if (rc /= 0) return
      tot = 0
      do i = 1, w%nlocal
        associate (s => w%loc(i))
          ! the kernel zeroes the owned planes of the next e; its ghost planes here
          pl4 = int(s%plane1, c_size_t) * 4_c_size_t
          rc = ndsmk_fill0(w%ealt(i), int(s%sl%g, c_size_t) * pl4); if (rc /= 0) return
          rc = ndsmk_fill0(dptr_offset(w%ealt(i), int(s%sl%g + s%sl%z1 - s%sl%z0, c_size_t) * pl4), &
                           int(s%sl%nloc - s%sl%g - (s%sl%z1 - s%sl%z0), c_size_t) * pl4)
          if (rc /= 0) return
          rc = ndsmk_update_residual_f32(s%lev(1)%g, s%dl(1)%u, s%prev, rhs_of(s, 1), w%e(i), w%ealt(i), w%r32(i), met)
          if (rc /= 0) return
          tmp = s%dl(1)%u; s%dl(1)%u = s%prev; s%prev = tmp
          tmp = w%e(i); w%e(i) = w%ealt(i); w%ealt(i) = tmp
          tot(1) = max(tot(1), met(1)); tot(2) = tot(2) + met(2)
        end associate
      end do
      w%ghost_depth = 0; w%ghost_sweep = 0                 ! u' was written on owned planes only
      w%eghost = g                      ! e = 0 everywhere
      rc = exchange_f32(w, w%r32, g); if (rc /= 0) return
      if (w%rccl) then
        rc = ndsmk_dist_allreduce_max_sum(tot); if (rc /= 0) return
      end if
      if (w%loc(1)%use_max) then
        du = tot(1)
      else
        du = tot(2) / (real(w%nzg, wp) * real(w%loc(1)%plane1, wp))
      end if
      ncycles = it
      if (present(hist)) then
        if (it <= size(hist)) hist(it) = du
      end if
      if (du < vc_tol) then
        ierr = 0
        exit
      end if
    end do
    du_last = du
    rc = 0
  end function

end module ndsmh_world
