! libndsm_hip - host-side description of NDSM's grid hierarchy and of the
! inter-level transfer operators, in the form the HIP kernels consume.
!
! What the reference does (and this module must reproduce number for number):
!   * level count  ngrids = FLOOR(LOG(nmin/2)/LOG(2))      ndsm_vector_potential.f90:60,631-632
!   * level shapes n_{l+1} = MAX(FLOOR(n_l * 0.5), 1)       ndsm_multigrid_core.f90:215-217
!   * level meshes: every coarse mesh spans the SAME extent  ndsm_multigrid_core.f90:253-259
!     q(j) = (j-1) L/(nq-1) + qmin, so h_c/h_f = (n_f-1)/(n_c-1) /= 2 - the grids
!     are not nested and the transfer weights depend on position.
!   * prolongation = N-linear interpolation at each fine coordinate
!     (bracket search ndsm_interp.f90:373-435, weights :138-142)
!   * restriction = its adjoint times (h_f/h_c)^ndim: for coarse point q0 all
!     fine points with coordinate in (q0-h_c, q0+h_c]  (ndsm_interp.f90:218-252),
!     weight prod_d |h_c - |q_f - q0|| h_f / h_c^2       (:229, :276-282)
!
! Both operators are tensor products, so instead of a search per point per
! V-cycle (the reference) the host computes 1-D tables per dimension ONCE per
! hierarchy - with the same floating-point expressions, hence the same
! numbers - and the device only gathers and multiplies.
module ndsmh_grid

  use, intrinsic :: iso_c_binding
  use ndsmh_iface, only: wp, ik, ndsmk_grid
  implicit none
  private

  public :: axis_t, level_t, axis_xfer_t, slab_t
  public :: ndsm_level_count, build_levels, build_axis_xfer, fill_grid_desc, locate_uniform
  public :: plan_slabs, apply_slab_window

  ! one coordinate axis of one level
  type :: axis_t
    real(wp), allocatable :: q(:)
  end type

  type :: level_t
    integer(c_int32_t) :: n(3) = 1
    integer(ik) :: npts = 0
    type(axis_t) :: ax(3)
    type(ndsmk_grid) :: g
  end type

  ! 1-D transfer tables between a fine and a coarse axis (0-based indices, ready
  ! for the device)
  type :: axis_xfer_t
    integer(c_int32_t) :: nf = 1, nc = 1, maxt = 1
    integer(c_int32_t), allocatable :: plo(:)        ! (nf) lower bracket in the coarse axis
    real(wp), allocatable :: pwl(:), pwh(:)          ! (nf) weights of upper / lower bracket point
    integer(c_int32_t), allocatable :: rlo(:), rcnt(:) ! (nc) first fine tap, number of taps
    real(wp), allocatable :: rw(:, :)                ! (maxt, nc) c2 = |h_c - |q_f - q0||
    real(wp) :: w2 = 0                               ! h_f / h_c**2
  end type

  ! z-slab of level 1 owned by one rank (all plane indices GLOBAL and 0-based,
  ! ranges half open).  The reference has no distributed mode; this is the
  ! decomposition of SURVEY 8e.
  type :: slab_t
    integer :: rank = 0, nranks = 1
    integer :: z0 = 0, z1 = 0        ! owned fine planes [z0, z1)
    integer :: g = 2                 ! ghost planes on each side of the local arrays
    integer :: nloc = 0              ! z1 - z0 + 2 g : planes of the local level-1 arrays
    integer :: k0 = 0                ! z0 - g : global index of local plane 0 (may be < 0)
    integer :: ck0 = 0, ck1 = 0      ! coarse planes this rank computes in the restriction
    integer :: pk0 = 0, pk1 = 0      ! coarse planes this rank reads in the prolongation
    integer :: cb0 = 0, cb1 = 0      ! window of the local coarse buffer = union of the two
    integer :: ci0 = 0, ci1 = 0      ! the coarse planes of [ck0, ck1) whose taps all lie in owned fine planes
  end type

contains

  ! Split nz fine planes over nranks slabs and derive, from the level 1 -> 2 z
  ! tables, who restricts which coarse plane, which coarse planes each rank needs
  ! back, and the ghost depth that makes both possible:
  !   * owned planes: balanced contiguous split;
  !   * coarse plane K is computed by the owner of its middle tap, so every rank
  !     gets one contiguous range and the ranges tile [0, nzc);
  !   * ghost depth = max(2, deepest tap outside the owner's slab): 2 planes feed
  !     the fused smoother's red/black pipeline, the rest the restriction.
  ! ok = .false. if some slab would own fewer planes than the ghost depth.
  ! part (optional): prescribed slab boundaries, rank r owns [part(r), part(r+1)) - used when this
  ! level is itself the coarse level of a distributed finer one, whose restriction ownership
  ! fixes the split; min_depth (optional): ghost planes the finer level's prolongation reads.
  subroutine plan_slabs(nz, nzc, tz, nranks, plan, ok, part, min_depth)
    integer, intent(in) :: nz, nzc, nranks
    type(axis_xfer_t), intent(in) :: tz
    type(slab_t), allocatable, intent(out) :: plan(:)
    logical, intent(out) :: ok
    integer, intent(in), optional :: part(0:)
    integer, intent(in), optional :: min_depth
    integer :: r, kc, m, own, depth, lo, hi

    ok = .false.
    if (nranks < 1 .or. nz < nranks) return
    allocate (plan(0:nranks - 1))
    do r = 0, nranks - 1
      plan(r)%rank = r; plan(r)%nranks = nranks
      if (present(part)) then
        plan(r)%z0 = part(r); plan(r)%z1 = part(r + 1)
        if (plan(r)%z1 <= plan(r)%z0) return
      else
        plan(r)%z0 = int((int(r, ik) * nz) / nranks)
        plan(r)%z1 = int((int(r + 1, ik) * nz) / nranks)
      end if
      plan(r)%ck0 = nzc; plan(r)%ck1 = 0
      plan(r)%ci0 = nzc; plan(r)%ci1 = 0
    end do
    if (present(part)) then
      if (part(0) /= 0 .or. part(nranks) /= nz) return
    end if
    depth = 4      ! two sweeps per halo exchange: each sweep consumes two ghost planes per side
    if (present(min_depth)) depth = max(depth, min_depth)
    do kc = 0, nzc - 1
      lo = tz%rlo(kc + 1); hi = lo + tz%rcnt(kc + 1)      ! fine taps [lo, hi)
      m = lo + tz%rcnt(kc + 1) / 2
      own = nranks - 1
      do r = 0, nranks - 1
        if (m >= plan(r)%z0 .and. m < plan(r)%z1) own = r
      end do
      plan(own)%ck0 = min(plan(own)%ck0, kc)
      plan(own)%ck1 = max(plan(own)%ck1, kc + 1)
      depth = max(depth, plan(own)%z0 - lo, hi - plan(own)%z1)
      if (lo >= plan(own)%z0 .and. hi <= plan(own)%z1) then   ! (tap ranges are monotone: one contiguous run)
        plan(own)%ci0 = min(plan(own)%ci0, kc)
        plan(own)%ci1 = max(plan(own)%ci1, kc + 1)
      end if
    end do
    do r = 0, nranks - 1
      if (plan(r)%ck1 <= plan(r)%ck0) then      ! owns no coarse plane
        plan(r)%ck0 = 0; plan(r)%ck1 = 0
      end if
      if (plan(r)%ci1 <= plan(r)%ci0) then
        plan(r)%ci0 = plan(r)%ck0; plan(r)%ci1 = plan(r)%ck0
      end if
      plan(r)%g = depth
      plan(r)%nloc = plan(r)%z1 - plan(r)%z0 + 2 * depth
      plan(r)%k0 = plan(r)%z0 - depth
      ! coarse planes the prolongation of this slab reads: the brackets of its owned planes and of its
      ! ghost planes (the smoother launch that folds the prolongation in corrects those as it loads them)
      plan(r)%pk0 = tz%plo(max(plan(r)%z0 - depth, 0) + 1)
      plan(r)%pk1 = tz%plo(min(plan(r)%z1 + depth, nz)) + 2
      plan(r)%cb0 = plan(r)%pk0; plan(r)%cb1 = plan(r)%pk1
      if (plan(r)%ck1 > plan(r)%ck0) then
        plan(r)%cb0 = min(plan(r)%cb0, plan(r)%ck0)
        plan(r)%cb1 = max(plan(r)%cb1, plan(r)%ck1)
      end if
      if (plan(r)%z1 - plan(r)%z0 < max(depth, 8)) return
    end do
    ok = .true.
  end subroutine

  ! Turn the (global) descriptor of level 1 into the descriptor of one slab:
  ! local plane count, global offset for colouring / mirror faces, owned range
  ! and the physical update bounds clipped to the local window.
  subroutine apply_slab_window(lv, sl)
    type(level_t), intent(inout) :: lv
    type(slab_t), intent(in) :: sl
    integer :: lbg, ubg
    lbg = lv%g%lb(3); ubg = lv%g%ub(3)              ! global bounds from fill_grid_desc
    lv%g%n(3) = sl%nloc
    lv%g%k0 = sl%k0
    lv%g%nzg = lv%n(3)
    lv%g%zown0 = sl%g
    lv%g%zown1 = sl%g + (sl%z1 - sl%z0)
    lv%g%lb(3) = max(lbg - sl%k0, 0)
    lv%g%ub(3) = min(ubg - sl%k0, sl%nloc - 1)
  end subroutine

  ! Number of grids the reference would use for this (fine) shape.
  pure function ndsm_level_count(ndim, nshape) result(ng)
    integer, intent(in) :: ndim
    integer(c_int32_t), intent(in) :: nshape(:)
    integer :: ng
    real(wp), parameter :: base_grid = 2
    ng = floor(log(real(minval(nshape(1:ndim)), wp) / base_grid) / log(real(2, wp)))
  end function

  ! Shapes and meshes of all levels; lev(1) is the caller's mesh verbatim.
  ! ext (optional, (2,3)): origin and extent per axis to use for the coarse meshes instead of those
  ! of (qx,qy,qz) - a hierarchy that starts at level l of a larger one must repeat THAT one's
  ! meshes bit for bit, and the end points of a coarse mesh are not exactly those of the finest.
  subroutine build_levels(ndim, nshape, qx, qy, qz, ngrids, lev, ext)
    integer, intent(in) :: ndim, ngrids
    integer(c_int32_t), intent(in) :: nshape(3)
    real(wp), intent(in) :: qx(:), qy(:), qz(:)
    type(level_t), allocatable, intent(out) :: lev(:)
    real(wp), intent(in), optional :: ext(2, 3)
    integer :: l, d, j, nq
    real(wp) :: qmin, span

    allocate (lev(ngrids))
    lev(1)%n = 1
    lev(1)%n(1:ndim) = nshape(1:ndim)
    allocate (lev(1)%ax(1)%q(nshape(1)), source=qx(1:nshape(1)))
    allocate (lev(1)%ax(2)%q(nshape(2)), source=qy(1:nshape(2)))
    if (ndim == 3) allocate (lev(1)%ax(3)%q(nshape(3)), source=qz(1:nshape(3)))

    do l = 2, ngrids
      lev(l)%n = 1
      do d = 1, ndim
        lev(l)%n(d) = max(floor(lev(l - 1)%n(d) * 0.5_wp), 1)
      end do
      do d = 1, ndim
        nq = lev(l)%n(d)
        qmin = minval(lev(1)%ax(d)%q)
        span = maxval(lev(1)%ax(d)%q) - qmin
        if (present(ext)) then
          qmin = ext(1, d); span = ext(2, d)
        end if
        allocate (lev(l)%ax(d)%q(nq))
        do j = 1, nq
          lev(l)%ax(d)%q(j) = (j - 1) * span / real(nq - 1, wp) + qmin
        end do
      end do
    end do

    do l = 1, ngrids
      lev(l)%npts = product(int(lev(l)%n(1:ndim), ik))
    end do
  end subroutine

  ! Operator constants and update bounds of one level.
  ! bcs(1:ndim) = lower faces, bcs(ndim+1:2*ndim) = upper faces, 'D' or 'N'.
  subroutine fill_grid_desc(ndim, lv, bcs)
    integer, intent(in) :: ndim
    type(level_t), intent(inout) :: lv
    character(len=1), intent(in) :: bcs(:)
    integer :: d
    real(wp) :: h, acc

    lv%g%ndim = ndim
    lv%g%n = lv%n
    lv%g%lb = 0
    lv%g%ub = lv%n - 1
    lv%g%w = 0
    do d = 1, ndim
      if (bcs(d) == 'D') lv%g%lb(d) = 1
      if (bcs(ndim + d) == 'D') lv%g%ub(d) = lv%n(d) - 2
      h = lv%ax(d)%q(2) - lv%ax(d)%q(1)
      lv%g%w(d) = 1.0_wp / h**2
    end do
    if (ndim == 3) then
      ! ndsm_optimized.f90:93-94 and :384
      acc = 2 * (lv%g%w(1) + lv%g%w(2) + lv%g%w(3))
      lv%g%wc = acc
      lv%g%w1 = 1.0_wp / acc
      ! colour of the first half sweep (ndsm_optimized.f90:106): 1-based
      ! i+j+k == lb(1) (mod 2)  <=>  0-based (i+j+k) mod 2 == [x-lower is 'D']
      lv%g%first_par = merge(1, 0, bcs(1) == 'D')
    else
      ! ndsm_poisson.f90:483-489 accumulates 2*w_d one dimension at a time
      acc = 0
      do d = 1, ndim
        acc = acc + 2.0_wp * lv%g%w(d)
      end do
      lv%g%wc = acc
      lv%g%w1 = 1.0_wp / acc
      lv%g%first_par = 0      ! red = even i+j (ndsm_poisson.f90:499-501)
    end if
    lv%g%all_neumann = merge(1, 0, all(bcs(1:2 * ndim) == 'N'))
    lv%g%k0 = 0
    lv%g%nzg = lv%n(3)
    lv%g%zown0 = 0
    lv%g%zown1 = lv%n(3)
  end subroutine

  ! Bracket of coordinate `c` in the uniform axis q(1:nq): returns the 1-based
  ! lower index and side = -1 / +1 when c lies at or beyond the first / last
  ! point, 0 otherwise (same decisions as ndsm_interp.f90:399-433).
  pure subroutine locate_uniform(q, nq, c, lo, side)
    real(wp), intent(in) :: q(:)
    integer, intent(in) :: nq
    real(wp), intent(in) :: c
    integer, intent(out) :: lo, side
    if (c <= q(1)) then
      lo = 1; side = -1
    else if (c >= q(nq)) then
      lo = nq - 1; side = +1
    else
      side = 0
      lo = min(floor((c - q(1)) / (q(2) - q(1))) + 1, nq - 1)
    end if
  end subroutine

  subroutine build_axis_xfer(qf, nf, qc, nc, t, ok)
    real(wp), intent(in) :: qf(:), qc(:)
    integer, intent(in) :: nf, nc
    type(axis_xfer_t), intent(out) :: t
    logical, intent(out) :: ok
    integer :: i, lo, hi, side, first, last, k
    real(wp) :: hc, hf, c0, ql, qh, dq, c1
    integer, allocatable :: f0(:), f1(:)

    ok = .false.
    if (nf < 2 .or. nc < 2) return
    t%nf = nf; t%nc = nc
    allocate (t%plo(nf), t%pwl(nf), t%pwh(nf), t%rlo(nc), t%rcnt(nc))

    ! ---- prolongation: bracket every fine coordinate in the coarse axis
    do i = 1, nf
      c0 = qf(i)
      call locate_uniform(qc, nc, c0, lo, side)
      ql = qc(lo); qh = qc(lo + 1)
      dq = qh - ql
      t%plo(i) = lo - 1
      t%pwl(i) = +(c0 - ql) / dq
      t%pwh(i) = -(c0 - qh) / dq
    end do

    ! ---- restriction: fine taps of every coarse coordinate
    hc = qc(2) - qc(1)
    hf = qf(2) - qf(1)
    t%w2 = hf / hc**2
    allocate (f0(nc), f1(nc))
    do i = 1, nc
      c0 = qc(i)
      call locate_uniform(qf, nf, c0 - hc, lo, side)
      first = merge(lo, lo + 1, side < 0)       ! first fine point strictly above c0-hc
      call locate_uniform(qf, nf, c0 + hc, lo, side)
      last = merge(lo + 1, lo, side > 0)        ! last fine point at or below c0+hc
      f0(i) = first; f1(i) = last
    end do
    if (any(f1 < f0) .or. any(f0 < 1) .or. any(f1 > nf)) return
    t%maxt = maxval(f1 - f0 + 1)
    allocate (t%rw(t%maxt, nc))
    t%rw = 0
    do i = 1, nc
      c0 = qc(i)
      t%rlo(i) = f0(i) - 1
      t%rcnt(i) = f1(i) - f0(i) + 1
      do k = f0(i), f1(i)
        c1 = abs(qf(k) - c0)
        t%rw(k - f0(i) + 1, i) = abs(hc - c1)
      end do
    end do
    hi = maxval(t%plo) + 1
    ok = (minval(t%plo) >= 0) .and. (hi <= nc - 1)
  end subroutine

end module ndsmh_grid
