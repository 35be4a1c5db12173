! libndsm_hip - ISO_C_BINDING shim between the Fortran 2003 host driver and the
! hand-written HIP kernels (ndsm_amd/csrc/ndsm_kernels.h).  Every interface
! below is a 1:1 image of a C prototype in that header; the two BIND(C) types
! mirror the C structs member for member.
!
! Kinds follow the reference's conventions (ndsm_root.f90:56-61): reals are
! C doubles, sizes and counters are 64-bit.
module ndsmh_iface

  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer, parameter :: wp = c_double
  integer, parameter :: ik = c_int64_t

  ! error codes shared with ndsm_kernels.h
  integer(c_int), parameter :: NDSMK_OK = 0, NDSMK_ENODEV = 9001, NDSMK_EARG = 9002, NDSMK_ENCCL = 9003

  type, bind(c) :: ndsmk_grid
    integer(c_int32_t) :: ndim = 3
    integer(c_int32_t) :: n(3) = 1
    integer(c_int32_t) :: lb(3) = 0, ub(3) = 0
    integer(c_int32_t) :: first_par = 0
    integer(c_int32_t) :: all_neumann = 0
    integer(c_int32_t) :: k0 = 0
    integer(c_int32_t) :: nzg = 1
    integer(c_int32_t) :: zown0 = 0, zown1 = 1
    real(c_double) :: w(3) = 0
    real(c_double) :: w1 = 0
    real(c_double) :: wc = 0
  end type

  type, bind(c) :: ndsmk_xfer
    integer(c_int32_t) :: nf(3) = 1, nc(3) = 1, maxt(3) = 1
    type(c_ptr) :: plo(3), pwl(3), pwh(3)
    type(c_ptr) :: rlo(3), rcnt(3), rw(3)
    real(c_double) :: w2(3) = 0
    integer(c_int32_t) :: f_k0 = 0, f_beg = 0, f_cnt = 1
    integer(c_int32_t) :: c_k0 = 0, c_beg = 0, c_cnt = 1
    integer(c_int32_t) :: stream_ok = 0
  end type

  interface

    function ndsmk_device_count() bind(c, name="ndsmk_device_count") result(n)
      import :: c_int
      integer(c_int) :: n
    end function

    function ndsmk_init(device) bind(c, name="ndsmk_init") result(rc)
      import :: c_int
      integer(c_int), value :: device
      integer(c_int) :: rc
    end function

    function ndsmk_last_error() bind(c, name="ndsmk_last_error") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function

    function ndsmk_alloc(p, bytes) bind(c, name="ndsmk_alloc") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), intent(out) :: p
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function

    function ndsmk_free(p) bind(c, name="ndsmk_free") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: p
      integer(c_int) :: rc
    end function

    function ndsmk_h2d(dst, src, bytes) bind(c, name="ndsmk_h2d") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function

    function ndsmk_d2h(dst, src, bytes) bind(c, name="ndsmk_d2h") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function

    function ndsmk_halo_packed_plane(nx, ny) bind(c, name="ndsmk_halo_packed_plane") result(n)
      import :: c_int, c_long_long
      integer(c_int), value :: nx, ny
      integer(c_long_long) :: n
    end function
    function ndsmk_halo_pack(src, buf, nx, ny, depth, kg0, first_par) bind(c, name="ndsmk_halo_pack") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: src, buf
      integer(c_int), value :: nx, ny, depth, kg0, first_par
      integer(c_int) :: rc
    end function
    function ndsmk_halo_unpack(dst, buf, nx, ny, depth, kg0, first_par) bind(c, name="ndsmk_halo_unpack") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: dst, buf
      integer(c_int), value :: nx, ny, depth, kg0, first_par
      integer(c_int) :: rc
    end function
    function ndsmk_halo_copy(dst, src, nx, ny, depth, kg0, first_par) bind(c, name="ndsmk_halo_copy") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: dst, src
      integer(c_int), value :: nx, ny, depth, kg0, first_par
      integer(c_int) :: rc
    end function
    function ndsmk_d2d(dst, src, bytes) bind(c, name="ndsmk_d2d") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function

    function ndsmk_fill0(p, bytes) bind(c, name="ndsmk_fill0") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: p
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function

    function ndsmk_sync() bind(c, name="ndsmk_sync") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function

    function ndsmk_timer_start() bind(c, name="ndsmk_timer_start") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function

    function ndsmk_timer_stop(ms) bind(c, name="ndsmk_timer_stop") result(rc)
      import :: c_int, c_double
      real(c_double), intent(out) :: ms
      integer(c_int) :: rc
    end function

    function ndsmk_relax(g, u, ualt, rhs, nsweeps, variant, result_in_alt) bind(c, name="ndsmk_relax") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, ualt, rhs
      integer(c_int), value :: nsweeps, variant
      integer(c_int), intent(out) :: result_in_alt
      integer(c_int) :: rc
    end function

    function ndsmk_relax_residual(g, u, ualt, rhs, r, nsweeps, variant, result_in_alt) &
        bind(c, name="ndsmk_relax_residual") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, ualt, rhs, r
      integer(c_int), value :: nsweeps, variant
      integer(c_int), intent(out) :: result_in_alt
      integer(c_int) :: rc
    end function

    ! px: c_loc of an ndsmk_xfer (u += P uc before the sweeps) or c_null_ptr
    function ndsmk_relax3(g, u, a, b, keep, rhs, nsweeps, r, prev, where, met_done, px, uc) &
        bind(c, name="ndsmk_relax3") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, a, b, keep, rhs, r, prev, px, uc
      integer(c_int), value :: nsweeps
      integer(c_int), intent(out) :: where, met_done
      integer(c_int) :: rc
    end function

    function ndsmk_fetch_fused_metric(h_out2) bind(c, name="ndsmk_fetch_fused_metric") result(rc)
      import :: c_int, c_double
      real(c_double), intent(out) :: h_out2(2)
      integer(c_int) :: rc
    end function

    function ndsmk_fused_window(g, u, uout, rhs, nsweeps, z0, z1, px, uc, prev, accumulate) &
        bind(c, name="ndsmk_fused_window") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, uout, rhs, px, uc, prev
      integer(c_int), value :: nsweeps, z0, z1, accumulate
      integer(c_int) :: rc
    end function

    function ndsmk_fused_window_res(g, u, uout, rhs, rout, z0, z1) bind(c, name="ndsmk_fused_window_res") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, uout, rhs, rout
      integer(c_int), value :: z0, z1
      integer(c_int) :: rc
    end function

    function ndsmk_fused_metric_ok(g) bind(c, name="ndsmk_fused_metric_ok") result(ok)
      import :: ndsmk_grid, c_int
      type(ndsmk_grid), intent(in) :: g
      integer(c_int) :: ok
    end function

    function ndsmk_fused_prolong_ok(g, rhs, nsweeps) bind(c, name="ndsmk_fused_prolong_ok") result(ok)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: rhs
      integer(c_int), value :: nsweeps
      integer(c_int) :: ok
    end function

    function ndsmk_note_error(code, what) bind(c, name="ndsmk_note_error") result(rc)
      import :: c_int, c_char
      integer(c_int), value :: code
      character(kind=c_char), intent(in) :: what(*)
      integer(c_int) :: rc
    end function

    function ndsmk_select_lane(lane) bind(c, name="ndsmk_select_lane") result(rc)
      import :: c_int
      integer(c_int), value :: lane
      integer(c_int) :: rc
    end function

    function ndsmk_lane_idle(lane) bind(c, name="ndsmk_lane_idle") result(rc)
      import :: c_int
      integer(c_int), value :: lane
      integer(c_int) :: rc
    end function
    function ndsmk_lane_fence(lane, to_main) bind(c, name="ndsmk_lane_fence") result(rc)
      import :: c_int
      integer(c_int), value :: lane, to_main
      integer(c_int) :: rc
    end function

    function ndsmk_lane_sync(lane) bind(c, name="ndsmk_lane_sync") result(rc)
      import :: c_int
      integer(c_int), value :: lane
      integer(c_int) :: rc
    end function

    function ndsmk_capture_begin() bind(c, name="ndsmk_capture_begin") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function

    function ndsmk_capture_end(exec) bind(c, name="ndsmk_capture_end") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), intent(out) :: exec
      integer(c_int) :: rc
    end function

    function ndsmk_graph_launch(exec) bind(c, name="ndsmk_graph_launch") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: exec
      integer(c_int) :: rc
    end function

    function ndsmk_graph_destroy(exec) bind(c, name="ndsmk_graph_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: exec
      integer(c_int) :: rc
    end function

    function ndsmk_diff_metrics_begin(a, b, n, copy) bind(c, name="ndsmk_diff_metrics_begin") result(rc)
      import :: c_ptr, c_int64_t, c_int
      type(c_ptr), value :: a, b
      integer(c_int64_t), value :: n
      integer(c_int), value :: copy
      integer(c_int) :: rc
    end function

    function ndsmk_diff_metrics_end(out2) bind(c, name="ndsmk_diff_metrics_end") result(rc)
      import :: c_int, c_double
      real(c_double), intent(out) :: out2(2)
      integer(c_int) :: rc
    end function

    function ndsmk_select_stream(which) bind(c, name="ndsmk_select_stream") result(rc)
      import :: c_int
      integer(c_int), value :: which
      integer(c_int) :: rc
    end function

    function ndsmk_stream_fence(from, to) bind(c, name="ndsmk_stream_fence") result(rc)
      import :: c_int
      integer(c_int), value :: from, to
      integer(c_int) :: rc
    end function

    function ndsmk_update_residual_f32(g, u, unew, rhs, e, ezero, r, h_out2) &
        bind(c, name="ndsmk_update_residual_f32") result(rc)
      import :: ndsmk_grid, c_ptr, c_int, c_double
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, unew, rhs, e, ezero, r
      real(c_double), intent(out) :: h_out2(2)
      integer(c_int) :: rc
    end function

    function ndsmk_relax_f32(g, e, ealt, r, nsweeps, force, r_out, result_in_alt) &
        bind(c, name="ndsmk_relax_f32") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: e, ealt, r, r_out
      integer(c_int), value :: nsweeps, force
      integer(c_int), intent(out) :: result_in_alt
      integer(c_int) :: rc
    end function

    function ndsmk_restrict_f32(x, r_f, rhs_c, u_c) bind(c, name="ndsmk_restrict_f32") result(rc)
      import :: ndsmk_xfer, c_ptr, c_int
      type(ndsmk_xfer), intent(in) :: x
      type(c_ptr), value :: r_f, rhs_c, u_c
      integer(c_int) :: rc
    end function

    function ndsmk_prolong_add_f32(x, u_c, e_f) bind(c, name="ndsmk_prolong_add_f32") result(rc)
      import :: ndsmk_xfer, c_ptr, c_int
      type(ndsmk_xfer), intent(in) :: x
      type(c_ptr), value :: u_c, e_f
      integer(c_int) :: rc
    end function

    function ndsmk_residual(g, u, rhs, r) bind(c, name="ndsmk_residual") result(rc)
      import :: ndsmk_grid, c_ptr, c_int
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, rhs, r
      integer(c_int) :: rc
    end function

    function ndsmk_restrict(x, r_f, rhs_c, u_c) bind(c, name="ndsmk_restrict") result(rc)
      import :: ndsmk_xfer, c_ptr, c_int
      type(ndsmk_xfer), intent(in) :: x
      type(c_ptr), value :: r_f, rhs_c, u_c
      integer(c_int) :: rc
    end function

    subroutine ndsmk_restrict_stream_tile(ci, cj, fx, fy, maxt) bind(c, name="ndsmk_restrict_stream_tile")
      import :: c_int
      integer(c_int), intent(out) :: ci, cj, fx, fy, maxt
    end subroutine

    function ndsmk_prolong_add(x, u_c, u_f) bind(c, name="ndsmk_prolong_add") result(rc)
      import :: ndsmk_xfer, c_ptr, c_int
      type(ndsmk_xfer), intent(in) :: x
      type(c_ptr), value :: u_c, u_f
      integer(c_int) :: rc
    end function

    function ndsmk_diff_metrics(a, b, n, copy, out2) bind(c, name="ndsmk_diff_metrics") result(rc)
      import :: c_ptr, c_int64_t, c_int, c_double
      type(c_ptr), value :: a, b
      integer(c_int64_t), value :: n
      integer(c_int), value :: copy
      real(c_double), intent(out) :: out2(2)
      integer(c_int) :: rc
    end function

    function ndsmk_solve_exact(g, u, rhs, scratch, ex_tol, use_max, nmax, d_info) &
        bind(c, name="ndsmk_solve_exact") result(rc)
      import :: ndsmk_grid, c_ptr, c_int, c_double
      type(ndsmk_grid), intent(in) :: g
      type(c_ptr), value :: u, rhs, scratch, d_info
      real(c_double), value :: ex_tol
      integer(c_int), value :: use_max, nmax
      integer(c_int) :: rc
    end function

    function ndsmk_solve_exact_on_device(g) bind(c, name="ndsmk_solve_exact_on_device") result(ok)
      import :: ndsmk_grid, c_int
      type(ndsmk_grid), intent(in) :: g
      integer(c_int) :: ok
    end function

    function ndsmk_tail_applies(nlev, g, x) bind(c, name="ndsmk_tail_applies") result(ok)
      import :: ndsmk_grid, ndsmk_xfer, c_int
      integer(c_int), value :: nlev
      type(ndsmk_grid), intent(in) :: g(*)
      type(ndsmk_xfer), intent(in) :: x(*)
      integer(c_int) :: ok
    end function

    function ndsmk_tail_cycle(nlev, g, x, u, rhs, ms, ex_tol, use_max, nmax, d_info) &
        bind(c, name="ndsmk_tail_cycle") result(rc)
      import :: ndsmk_grid, ndsmk_xfer, c_ptr, c_int, c_double
      integer(c_int), value :: nlev
      type(ndsmk_grid), intent(in) :: g(*)
      type(ndsmk_xfer), intent(in) :: x(*)
      type(c_ptr), intent(in) :: u(*), rhs(*)
      integer(c_int), value :: ms
      real(c_double), value :: ex_tol
      integer(c_int), value :: use_max, nmax
      type(c_ptr), value :: d_info
      integer(c_int) :: rc
    end function

    function ndsmk_debug_tail(on) bind(c, name="ndsmk_debug_tail") result(rc)
      import :: c_int
      integer(c_int), value :: on
      integer(c_int) :: rc
    end function

    function ndsmk_balance_curl(A, B, n3, x, y, z, phi6, span3, dq3, curl_first) &
        bind(c, name="ndsmk_balance_curl") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: A, B, x, y, z
      integer(c_int32_t), intent(in) :: n3(3)
      real(c_double), intent(in) :: phi6(6), span3(3), dq3(3)
      integer(c_int), value :: curl_first
      integer(c_int) :: rc
    end function

    function ndsmk_balance_curl_slab(A, B, n3, kg0, na, nb, boff, x, y, z, phi6, span3, dq3, curl_first) &
        bind(c, name="ndsmk_balance_curl_slab") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: A, B, x, y, z
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int), value :: kg0, na, nb, boff
      real(c_double), intent(in) :: phi6(6), span3(3), dq3(3)
      integer(c_int), value :: curl_first
      integer(c_int) :: rc
    end function

    ! ---- background transfers, pinned memory (runtime.hip) ----
    function ndsmk_bg_upload_unless_zero(h_src, d_dst, bytes, ticket) bind(c, name="ndsmk_bg_upload_unless_zero") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: h_src, d_dst
      integer(c_size_t), value :: bytes
      integer(c_int), intent(out) :: ticket
      integer(c_int) :: rc
    end function
    function ndsmk_bg_download(h_dst, d_src, bytes, ticket) bind(c, name="ndsmk_bg_download") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: h_dst, d_src
      integer(c_size_t), value :: bytes
      integer(c_int), intent(out) :: ticket
      integer(c_int) :: rc
    end function
    function ndsmk_bg_first_touch(h_dst, bytes, ticket) bind(c, name="ndsmk_bg_first_touch") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: h_dst
      integer(c_size_t), value :: bytes
      integer(c_int), intent(out) :: ticket
      integer(c_int) :: rc
    end function
    function ndsmk_bg_wait(ticket, flag) bind(c, name="ndsmk_bg_wait") result(rc)
      import :: c_int
      integer(c_int), value :: ticket
      integer(c_int), intent(out) :: flag
      integer(c_int) :: rc
    end function
    function ndsmk_bg_drain() bind(c, name="ndsmk_bg_drain") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
    function ndsmk_host_alloc(p, bytes) bind(c, name="ndsmk_host_alloc") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), intent(out) :: p
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function
    function ndsmk_host_free(p) bind(c, name="ndsmk_host_free") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: p
      integer(c_int) :: rc
    end function
    function ndsmk_mem_info(free_bytes, total_bytes) bind(c, name="ndsmk_mem_info") result(rc)
      import :: c_size_t, c_int
      integer(c_size_t), intent(out) :: free_bytes, total_bytes
      integer(c_int) :: rc
    end function
    function ndsmk_h2d_async(dst, src, bytes) bind(c, name="ndsmk_h2d_async") result(rc)
      import :: c_ptr, c_size_t, c_int
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function
    subroutine ndsmk_on_low_memory(fn) bind(c, name="ndsmk_on_low_memory")
      import :: c_funptr
      type(c_funptr), value :: fn
    end subroutine
    subroutine ndsmk_at_reset(fn) bind(c, name="ndsmk_at_reset")
      import :: c_funptr
      type(c_funptr), value :: fn
    end subroutine

    ! ---- one component of the flux balance, the curl alone (post.hip) ----
    function ndsmk_balance_component(Ac, n3, c, x, y, z, phi6, span3) bind(c, name="ndsmk_balance_component") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: Ac, x, y, z
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int), value :: c
      real(c_double), intent(in) :: phi6(6), span3(3)
      integer(c_int) :: rc
    end function
    function ndsmk_curl_component(A, B, n3, dq3, c) bind(c, name="ndsmk_curl_component") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: A, B
      integer(c_int32_t), intent(in) :: n3(3)
      real(c_double), intent(in) :: dq3(3)
      integer(c_int), value :: c
      integer(c_int) :: rc
    end function

    function ndsmk_curl(A, B, n3, dq3) bind(c, name="ndsmk_curl") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: A, B
      integer(c_int32_t), intent(in) :: n3(3)
      real(c_double), intent(in) :: dq3(3)
      integer(c_int) :: rc
    end function

    ! ---- the face phase on the device (faces.hip) ----
    function ndsmk_face_offsets(n3, off6, total) bind(c, name="ndsmk_face_offsets") result(rc)
      import :: c_int, c_int32_t, c_int64_t
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int64_t), intent(out) :: off6(6), total
      integer(c_int) :: rc
    end function
    function ndsmk_face_extract(B, n3, faces) bind(c, name="ndsmk_face_extract") result(rc)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: B, faces
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int) :: rc
    end function
    function ndsmk_face_flux(faces, n3, h1h2, d_phi6) bind(c, name="ndsmk_face_flux") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: faces, d_phi6
      integer(c_int32_t), intent(in) :: n3(3)
      real(c_double), value :: h1h2
      integer(c_int) :: rc
    end function
    function ndsmk_face_rhs(faces, n3, f, d_phi6, area, rhs) bind(c, name="ndsmk_face_rhs") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: faces, d_phi6, rhs
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int), value :: f
      real(c_double), value :: area
      integer(c_int) :: rc
    end function
    function ndsmk_face_put(u, n3, f, vals) bind(c, name="ndsmk_face_put") result(rc)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: u, vals
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int), value :: f
      integer(c_int) :: rc
    end function
    function ndsmk_face_write(u, n3, chi, f, c, fac) bind(c, name="ndsmk_face_write") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: u, chi
      integer(c_int32_t), intent(in) :: n3(3)
      integer(c_int), value :: f, c
      real(c_double), value :: fac
      integer(c_int) :: rc
    end function

  end interface

contains

  ! byte offset into a device allocation
  function dptr_offset(base, bytes) result(p)
    type(c_ptr), intent(in) :: base
    integer(c_size_t), intent(in) :: bytes
    type(c_ptr) :: p
    p = transfer(transfer(base, 0_c_intptr_t) + int(bytes, c_intptr_t), p)
  end function

end module ndsmh_iface
