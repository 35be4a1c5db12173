! libndsm_hip - the vector-potential pipeline on a z-slab decomposition (BASELINE config[4]: the
! 2048 x 2048 x 1024 solve across the GPUs of one node).  One process per GPU, rank r holds the
! planes [z0, z1) of A and B that the slab plan (ndsm_hip_slab_plan) gives it; every rank makes
! the same call, collectively.
!
! The reference has no distributed mode; this is the pipeline of ndsmh_vecpot (reference:
! ndsm_vector_potential.f90:130-497) with its O(N) parts cut along z and its O(N^(2/3)) parts kept
! whole on rank 0, so that the ARITHMETIC is the single-GPU pipeline's:
!   1. every rank extracts B.n on its strips of the four side faces (the end ranks also their z
!      face) and ships them to rank 0                       RCCL send/recv, a few MB
!   2. rank 0 runs the face phase exactly as vecpot_solve does (vecpot_faces: fluxes, six 2-D
!      solves, A_t) and ships each rank its strips of the tangential data, the six fluxes and the
!      2-D flag
!   3. per component: initial guess slab + face data -> a z-slab world (ndsmh_world: level 1 in
!      slabs, halo exchange, coarse levels distributed or on rank 0) -> world_solve; the result
!      stays in HBM
!   4. one ghost plane of A per neighbour, flux-balance fields and curl on the slab (post.hip's
!      slab form), one download of the rank's A and B planes
! Same bits as the single-GPU call on identical input (tests/test_gpu_multirank.py), in fp64 and with
! the mixed-precision option (world_set_precision).
module ndsmh_wvecpot

  use, intrinsic :: iso_c_binding
  use ndsmh_iface
  use ndsmh_grid
  use ndsmh_mg
  use ndsmh_world
  use ndsmh_vecpot
  implicit none
  private

  public :: wvecpot_solve

  integer(c_size_t), parameter :: R8 = 8_c_size_t

contains

  ! A, B: host arrays (nx, ny, nzl, 3), nzl = z1 - z0 of plan(rank).  n3, qx, qy, qz: the GLOBAL grid.
  function wvecpot_solve(n3, iopt, ropt, qx, qy, qz, nranks, rank, A, B) result(rc)
    integer(c_int32_t), intent(in) :: n3(3)
    integer(ik), intent(inout) :: iopt(0:OPT_LEN - 1)
    real(wp), intent(inout) :: ropt(0:OPT_LEN - 1)
    real(wp), intent(in), target :: qx(:), qy(:), qz(:)
    integer, intent(in) :: nranks, rank
    real(wp), intent(inout), target, contiguous :: A(:, :, :, :), B(:, :, :, :)
    integer(c_int) :: rc

    character(len=*), parameter :: me = "compute_vector_potential"
    type(slab_t), allocatable :: plan(:)
    type(face_data), target :: fc(6), fl(6)          ! whole faces (rank 0), this rank's strips
    type(mg_world) :: w
    real(wp), allocatable, target :: pack(:)
    real(wp) :: dq(3), span(3), phi(6), du_last, tail(8)
    integer :: f, ax, c, i, r, lay, ierr2d, ierr3d, ncyc, ngr, z0, z1, nzl, glo, ghi, na, q0
    integer(ik) :: plane, cnt, npk, off
    integer(ik), allocatable :: cnt_r(:), off_r(:)
    logical :: livew, first, last
    character(len=1) :: bc3(6)
    real(wp), pointer :: comp(:, :, :)
    type(c_ptr) :: dA, dB, dmesh, dpack, usrc
    integer(c_size_t) :: off_y, off_z, nbA

    rc = 0
    iopt(IOPT_FAIL3D) = 0
    livew = .false.
    dA = c_null_ptr; dB = c_null_ptr; dmesh = c_null_ptr; dpack = c_null_ptr
    ngr = int(iopt(IOPT_NGRIDS))
    if (any(n3 < 2)) then
      iopt(IOPT_IERR) = 1
      return
    end if
    rc = NDSMK_EARG
    if (nranks < 2 .or. rank < 0 .or. rank >= nranks) return
    rc = world_plan_only(n3, qx, qy, qz, ngr, nranks, plan)
    if (rc /= 0) return
    z0 = plan(rank)%z0; z1 = plan(rank)%z1; nzl = z1 - z0
    first = (rank == 0); last = (rank == nranks - 1)
    rc = NDSMK_EARG
    if (size(A, 1) /= n3(1) .or. size(A, 2) /= n3(2) .or. size(A, 3) /= nzl .or. size(A, 4) /= 3) return
    if (any(shape(B) /= shape(A)) .or. nzl < 3) return
    rc = 0
    span = [maxval(qx) - minval(qx), maxval(qy) - minval(qy), maxval(qz) - minval(qz)]
    dq = [qx(2) - qx(1), qy(2) - qy(1), qz(2) - qz(1)]
    plane = int(n3(1), ik) * int(n3(2), ik)

    ! ---- 1. B.n on this rank's part of the faces ---------------------------
    call say(me, "Allocate memory to hold boundary conditions...")
    do f = 1, 6
      ax = face_axis(f)
      fl(f)%n1 = n3(face_t1(f))
      fl(f)%n2 = merge(int(n3(2)), nzl, ax == 3)
      if (ax == 3 .and. .not. merge(last, first, face_upper(f))) fl(f)%n2 = 0     ! not this rank's face
      allocate (fl(f)%bn(fl(f)%n1, fl(f)%n2), fl(f)%at1(fl(f)%n1, fl(f)%n2), fl(f)%at2(fl(f)%n1, fl(f)%n2))
      if (fl(f)%n2 == 0) cycle
      lay = merge(int(n3(ax)), 1, face_upper(f))
      if (ax == 3) lay = merge(nzl, 1, face_upper(f))
      comp => B(:, :, :, ax)
      call face_copy(comp, ax, lay, fl(f)%bn, to_face=.true.)
    end do

    ! what rank q ships to rank 0 (and, doubled + 8, gets back): four side strips, the top face
    allocate (cnt_r(0:nranks - 1), off_r(0:nranks - 1))
    off = 0
    do r = 0, nranks - 1
      cnt_r(r) = 2_ik * (int(n3(1), ik) + int(n3(2), ik)) * int(plan(r)%z1 - plan(r)%z0, ik)
      if (r == nranks - 1) cnt_r(r) = cnt_r(r) + plane
      off_r(r) = off
      off = off + 2_ik * cnt_r(r) + 8_ik
    end do
    npk = off
    allocate (pack(npk))
    rc = ndsmk_alloc(dpack, int(npk, c_size_t) * R8); if (rc /= 0) goto 900

    if (.not. first) then
      ! ---- ship the strips, wait for the tangential data --------------------
      off = 0
      do f = 1, 4
        call put(fl(f)%bn)
      end do
      if (last) call put(fl(6)%bn)
      rc = ndsmk_h2d(dpack, c_loc(pack), int(cnt_r(rank), c_size_t) * R8); if (rc /= 0) goto 900
      rc = ndsmk_dist_group_start(); if (rc /= 0) goto 900
      rc = ndsmk_dist_send(dpack, int(cnt_r(rank), c_size_t), 0_c_int); if (rc /= 0) goto 900
      rc = ndsmk_dist_group_end(); if (rc /= 0) goto 900
      cnt = 2_ik * cnt_r(rank) + 8_ik
      rc = ndsmk_dist_group_start(); if (rc /= 0) goto 900
      rc = ndsmk_dist_recv(dpack, int(cnt, c_size_t), 0_c_int); if (rc /= 0) goto 900
      rc = ndsmk_dist_group_end(); if (rc /= 0) goto 900
      rc = ndsmk_sync(); if (rc /= 0) goto 900
      rc = ndsmk_d2h(c_loc(pack), dpack, int(cnt, c_size_t) * R8); if (rc /= 0) goto 900
      off = 0
      do f = 1, 4
        call get(fl(f)%at1); call get(fl(f)%at2)
      end do
      if (last) then
        call get(fl(6)%at1); call get(fl(6)%at2)
      end if
      tail = pack(off + 1:off + 8)
      phi = tail(1:6); ierr2d = nint(tail(7)); rc = nint(tail(8), c_int)
      if (rc /= 0) goto 900                  ! the face phase failed on rank 0: every rank leaves
    else
      ! ---- rank 0: whole faces, the face phase, the tangential data back ----
      do f = 1, 6
        fc(f)%n1 = n3(face_t1(f)); fc(f)%n2 = n3(face_t2(f))
        allocate (fc(f)%bn(fc(f)%n1, fc(f)%n2), fc(f)%chi(fc(f)%n1, fc(f)%n2))
        allocate (fc(f)%at1(fc(f)%n1, fc(f)%n2), fc(f)%at2(fc(f)%n1, fc(f)%n2))
      end do
      do f = 1, 4
        fc(f)%bn(:, z0 + 1:z1) = fl(f)%bn
      end do
      fc(5)%bn = fl(5)%bn
      rc = ndsmk_dist_group_start(); if (rc /= 0) goto 900
      do r = 1, nranks - 1
        rc = ndsmk_dist_recv(dptr_offset(dpack, int(off_r(r), c_size_t) * R8), int(cnt_r(r), c_size_t), int(r, c_int))
        if (rc /= 0) goto 900
      end do
      rc = ndsmk_dist_group_end(); if (rc /= 0) goto 900
      rc = ndsmk_sync(); if (rc /= 0) goto 900
      rc = ndsmk_d2h(c_loc(pack), dpack, int(npk, c_size_t) * R8); if (rc /= 0) goto 900
      do r = 1, nranks - 1
        off = off_r(r)
        do f = 1, 4
          call get(fc(f)%bn(:, plan(r)%z0 + 1:plan(r)%z1))
        end do
        if (r == nranks - 1) call get(fc(6)%bn)
      end do
      rc = vecpot_faces(iopt, ropt, qx, qy, qz, dq, span, fc, phi, ierr2d)
      ! (a failure here still has to reach the other ranks, which wait in their recv)
      tail = 0
      tail(1:6) = phi; tail(7) = real(ierr2d, wp); tail(8) = real(rc, wp)
      do r = 1, nranks - 1
        off = off_r(r)
        do f = 1, 4
          call put(fc(f)%at1(:, plan(r)%z0 + 1:plan(r)%z1)); call put(fc(f)%at2(:, plan(r)%z0 + 1:plan(r)%z1))
        end do
        if (r == nranks - 1) then
          call put(fc(6)%at1); call put(fc(6)%at2)
        end if
        pack(off + 1:off + 8) = tail
      end do
      i = rc
      rc = ndsmk_h2d(dpack, c_loc(pack), int(npk, c_size_t) * R8); if (rc /= 0) goto 900
      rc = ndsmk_dist_group_start(); if (rc /= 0) goto 900
      do r = 1, nranks - 1
        rc = ndsmk_dist_send(dptr_offset(dpack, int(off_r(r), c_size_t) * R8), int(2_ik * cnt_r(r) + 8_ik, c_size_t), &
                             int(r, c_int))
        if (rc /= 0) goto 900
      end do
      rc = ndsmk_dist_group_end(); if (rc /= 0) goto 900
      rc = int(i, c_int); if (rc /= 0) goto 900
      do f = 1, 4
        fl(f)%at1 = fc(f)%at1(:, z0 + 1:z1); fl(f)%at2 = fc(f)%at2(:, z0 + 1:z1)
      end do
      fl(5)%at1 = fc(5)%at1; fl(5)%at2 = fc(5)%at2
    end if

    ! ---- 3. the three 3-D Laplace problems, level 1 in z-slabs --------------
    call say(me, "Solve BVP 3D...")
    glo = merge(0, 1, first); ghi = merge(0, 1, last)
    na = nzl + glo + ghi
    nbA = int(plane, c_size_t) * int(na, c_size_t) * R8
    rc = ndsmk_alloc(dA, 3_c_size_t * nbA); if (rc /= 0) goto 900
    do c = 1, 3
      comp => A(:, :, :, c)
      do i = 1, 4
        f = face_order(i, c)
        if (fl(f)%n2 == 0) cycle               ! a z face of another rank
        lay = merge(int(n3(face_axis(f))), 1, face_upper(f))
        if (face_axis(f) == 3) lay = merge(nzl, 1, face_upper(f))
        if (face_t1(f) == c) then
          call face_copy(comp, face_axis(f), lay, fl(f)%at1, to_face=.false.)
        else
          call face_copy(comp, face_axis(f), lay, fl(f)%at2, to_face=.false.)
        end if
      end do
      bc3 = 'D'
      bc3(c) = 'N'; bc3(3 + c) = 'N'                        ! :655,:671,:687
      rc = world_create(w, n3, qx, qy, qz, bc3, ngr, nranks, rank); livew = .true.
      if (rc /= 0) goto 900
      call world_set_params(w, merge(5, int(iopt(IOPT_MS)), c == 3), ropt(ROPT_CTOL), iopt(IOPT_DUMAX) == 1, &
                            int(iopt(IOPT_NMAXEX)))          ! Q2
      if (iopt(IOPT_PREC) /= 0) then                       ! fp32 correction cycle where the slabs allow it
        if (.not. world_set_precision(w, int(iopt(IOPT_PREC)))) continue
      end if
      rc = mg_zero_rhs(w%loc(1)); if (rc /= 0) goto 900     ! :640-641 rhs = 0
      rc = world_upload(w, 1, MG_BUF_U, c_loc(comp), z0, nzl); if (rc /= 0) goto 900
      rc = world_solve(w, ropt(ROPT_VTOL), int(iopt(IOPT_NCYCLES)), du_last, ncyc, ierr3d); if (rc /= 0) goto 900
      usrc = mg_level_ptr(w%loc(1), 1, MG_BUF_U, cnt)
      rc = ndsmk_d2d(dptr_offset(dA, int(c - 1, c_size_t) * nbA + int(glo, c_size_t) * int(plane, c_size_t) * R8), &
                     dptr_offset(usrc, int(w%loc(1)%sl%g, c_size_t) * int(plane, c_size_t) * R8), &
                     int(nzl, c_size_t) * int(plane, c_size_t) * R8)
      if (rc /= 0) goto 900
      if (ierr3d /= 0) then
        if (first) print *, "Warning: IOPT_NCYCLES exceeded. V-cycle iteration may not have converged"
        iopt(IOPT_FAIL3D) = ior(iopt(IOPT_FAIL3D), ishft(1_ik, c - 1))
      end if
      if (ncyc > 1 .or. c == 1) then
        iopt(IOPT_NCYC_OUT) = ncyc
        ropt(ROPT_DULAST) = du_last
      end if
      rc = ndsmk_sync(); if (rc /= 0) goto 900
      call world_destroy(w); livew = .false.
    end do

    ! ---- 4. ghost planes of A, flux balance + curl on the slab ---------------
    call say(me, "Compute B = curl(B) and flux correction...")
    rc = ndsmk_dist_group_start(); if (rc /= 0) goto 900
    do c = 1, 3
      q0 = c - 1
      if (.not. first) then
        rc = ndsmk_dist_send(aplane(q0, glo), int(plane, c_size_t), int(rank - 1, c_int)); if (rc /= 0) goto 900
        rc = ndsmk_dist_recv(aplane(q0, 0), int(plane, c_size_t), int(rank - 1, c_int)); if (rc /= 0) goto 900
      end if
      if (.not. last) then
        rc = ndsmk_dist_send(aplane(q0, glo + nzl - 1), int(plane, c_size_t), int(rank + 1, c_int)); if (rc /= 0) goto 900
        rc = ndsmk_dist_recv(aplane(q0, na - 1), int(plane, c_size_t), int(rank + 1, c_int)); if (rc /= 0) goto 900
      end if
    end do
    rc = ndsmk_dist_group_end(); if (rc /= 0) goto 900
    rc = ndsmk_alloc(dB, 3_c_size_t * int(plane, c_size_t) * int(nzl, c_size_t) * R8); if (rc /= 0) goto 900
    off_y = int(n3(1), c_size_t) * R8
    off_z = off_y + int(n3(2), c_size_t) * R8
    rc = ndsmk_alloc(dmesh, off_z + int(n3(3), c_size_t) * R8); if (rc /= 0) goto 900
    rc = ndsmk_h2d(dmesh, c_loc(qx), int(n3(1), c_size_t) * R8); if (rc /= 0) goto 900
    rc = ndsmk_h2d(dptr_offset(dmesh, off_y), c_loc(qy), int(n3(2), c_size_t) * R8); if (rc /= 0) goto 900
    rc = ndsmk_h2d(dptr_offset(dmesh, off_z), c_loc(qz), int(n3(3), c_size_t) * R8); if (rc /= 0) goto 900
    if (iopt(IOPT_FLXCRL) == 1 .and. first) print *, "FLAG SET: FLXCRL"
    rc = ndsmk_balance_curl_slab(dA, dB, n3, int(z0 - glo, c_int), int(na, c_int), int(nzl, c_int), int(glo, c_int), &
                                 dmesh, dptr_offset(dmesh, off_y), dptr_offset(dmesh, off_z), phi, span, dq, &
                                 merge(1_c_int, 0_c_int, iopt(IOPT_FLXCRL) == 1))
    if (rc /= 0) goto 900
    do c = 1, 3
      comp => A(:, :, :, c)
      rc = ndsmk_d2h(c_loc(comp), aplane(c - 1, glo), int(nzl, c_size_t) * int(plane, c_size_t) * R8)
      if (rc /= 0) goto 900
    end do
    rc = ndsmk_d2h(c_loc(B), dB, 3_c_size_t * int(plane, c_size_t) * int(nzl, c_size_t) * R8); if (rc /= 0) goto 900
    iopt(IOPT_IERR) = ierr2d                                ! Q3'
    call say(me, "Deallocate memory...")

900 continue
    if (rc /= 0) i = ndsmk_dist_group_abort()      ! an error inside a send/recv group: close it before leaving
    if (livew) call world_destroy(w)
    i = ndsmk_free(dA); i = ndsmk_free(dB); i = ndsmk_free(dmesh); i = ndsmk_free(dpack)

  contains

    ! device address of local plane k (0-based) of component q (0-based) of the A slab
    function aplane(q, k) result(p)
      integer, intent(in) :: q, k
      type(c_ptr) :: p
      p = dptr_offset(dA, int(q, c_size_t) * nbA + int(k, c_size_t) * int(plane, c_size_t) * R8)
    end function

    subroutine put(v)
      real(wp), intent(in) :: v(:, :)
      integer(ik) :: m
      m = size(v, kind=ik)
      pack(off + 1:off + m) = reshape(v, [m])
      off = off + m
    end subroutine

    subroutine get(v)
      real(wp), intent(out) :: v(:, :)
      integer(ik) :: m
      m = size(v, kind=ik)
      v = reshape(pack(off + 1:off + m), shape(v))
      off = off + m
    end subroutine

  end function

end module ndsmh_wvecpot
