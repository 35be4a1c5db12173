"""GPU tests of the runtime layer that must run LAST in the session (the file name sorts behind the
parity tests): bring-up of the REAL RCCL library on one rank, and shutdown / re-initialisation of the
device runtime with everything the library caches on the device."""
import ctypes
import os

import numpy as np
import pytest

from golden_inputs import analytic_case, rand_field, uniform_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library()
    rc = L.ndsm_hip_init(-1)
    assert rc == 0, _lib.last_error(L)
    return _lib


def test_real_rccl_one_rank_bringup(hip):
    """the product library against the REAL librccl (the multi-rank tests use a shared-memory stand-in,
    two ranks on one GPU being refused by RCCL): unique id -> ncclCommInitRank(1 rank) -> the library's
    own grouped ncclSend/ncclRecv pair to itself on the main and on the communication stream -> the
    2-value all-reduce -> a World on that communicator -> ncclCommDestroy"""
    L = hip.load_library()
    libs = hip.bound_libs(L)
    assert "/opt/rocm" in libs["rccl"] and "librccl.so" in libs["rccl"], libs
    assert "/opt/rocm" in libs["hip"], libs
    assert hip.dist_info(L) == (0, 0)
    uid = hip.dist_unique_id(L)
    hip.dist_init(0, 1, uid, L)
    try:
        assert hip.dist_info(L) == (0, 1)          # ncclCommUserRank / ncclCommCount of the live communicator
        for nelem in (2, 4096, 1 << 20):            # 8 MiB: the size of one 1024^2 halo plane
            rc = L.ndsm_hip_dist_selftest(nelem)
            assert rc == 0, hip.last_error(L)
        ns = [64, 48, 40]
        mesh = uniform_mesh(ns)
        shp = tuple(ns[::-1])
        u, rhs = rand_field(shp, 1), rand_field(shp, 2)
        W = hip.World(ns, mesh, "NDDNDD", 1, 0)
        W.upload(hip.BUF_U, u)
        W.upload(hip.BUF_RHS, rhs)
        iw, duw, ncw, hw = W.solve(vc_tol=1e-10, nmax=3, hist_len=4)
        got = W.download(hip.BUF_U)
        W.close()
        S = hip.MGSolver(ns, mesh, "NDDNDD")
        S.upload(1, hip.BUF_U, u)
        S.upload(1, hip.BUF_RHS, rhs)
        i_s, dus, ncs, hs = S.solve(vc_tol=1e-10, nmax=3, hist_len=4)
        want = S.download(1, hip.BUF_U)
        S.close()
        assert list(hw) == list(hs) and np.array_equal(got, want)
    finally:
        hip.dist_finalize(L)
    assert hip.dist_info(L) == (0, 0)


def test_shutdown_and_reinit(hip):
    """ndsm_hip_shutdown drops streams, scratch and caches; the next call brings the runtime up again and
    re-issues the per-kernel attributes (dynamic-LDS sizes of the fused smoother / streamed restriction):
    a solve that uses every large-level launch gives the same bits before and after"""
    import ndsm_amd
    L = hip.load_library()
    ns = [161, 120, 115]
    mesh = uniform_mesh(ns)
    shp = tuple(ns[::-1])
    u0 = rand_field(shp, 21)

    def run():
        S = hip.MGSolver(ns, mesh, "NDDNDD", ms=5)
        S.zero_rhs()
        S.upload(1, hip.BUF_U, u0)
        ie, du, nc, h = S.solve(vc_tol=1e-10, nmax=3, hist_len=8)
        out = S.download(1, hip.BUF_U)
        S.close()
        return list(h), out

    x, y, z, _A1, b1 = analytic_case(22)
    h1, a1 = run()
    _i, A1, B1 = ndsm_amd.vector_potential(x, y, z, b1.copy())
    assert L.ndsm_hip_shutdown() == 0
    assert L.ndsm_hip_shutdown() == 0           # idempotent
    assert L.ndsm_hip_sync() == 9001            # nothing is up: an error code, not a crash
    assert L.ndsm_hip_init(0) == 0
    h2, a2 = run()
    assert h1 == h2 and np.array_equal(a1, a2)
    assert L.ndsm_hip_shutdown() == 0
    _i, A2, B2 = ndsm_amd.vector_potential(x, y, z, b1.copy())     # re-initialises by itself
    assert np.array_equal(A1, A2) and np.array_equal(B1, B2)
    h3, a3 = run()
    assert h1 == h3 and np.array_equal(a1, a3)


def test_cached_context_is_evicted_when_memory_runs_out(hip):
    """ndsm_vector_solve keeps its hierarchies and device arrays for the next call on the same mesh; an
    allocation that does not fit next to them must get that memory back instead of failing: a 320^3 call
    (3.4 GiB cached), then device allocations that leave less than that free, then a solver that needs it"""
    import ndsm_amd
    L = hip.load_library()
    L.ndsm_hip_device_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    L.ndsm_hip_device_free.argtypes = [ctypes.c_void_p]
    hipdll = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so", mode=os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))

    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hipdll.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    n = 320
    x, y, z, _A1, b1 = analytic_case(n)
    f0 = free_bytes()
    ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1)
    held = f0 - free_bytes()
    assert ierr == 0 and held > 2 * 2**30, held                     # the cache is alive
    # fill the device until less than the cached amount is left
    blocks = []
    try:
        chunk = 8 * 2**30
        while free_bytes() > chunk + 2**30:
            p = ctypes.c_void_p()
            assert L.ndsm_hip_device_alloc(chunk, ctypes.byref(p)) == 0
            blocks.append(p)
        left = free_bytes()
        # a solver whose level-1 arrays alone need more than what is left, but less than left + cache
        need_pts = int((left + held // 2) / (5.5 * 8))
        m = int(round(need_pts ** (1.0 / 3.0))) & ~1
        mesh = uniform_mesh([m, m, m])
        S = hip.MGSolver([m, m, m], mesh, "NDDNDD")                     # succeeds only if the cache was given back
        S.close()
        assert f0 - free_bytes() - len(blocks) * chunk < held // 4      # ... and it was
    finally:
        for p in blocks:
            L.ndsm_hip_device_free(p)
    ierr2, A2, B2 = ndsm_amd.vector_potential(x, y, z, b1)            # rebuilt on demand, same bits
    assert np.array_equal(A, A2) and np.array_equal(B, B2)


@pytest.mark.parametrize("shape", ([64, 64, 64], [200, 200, 184], [45, 38, 51]), ids=lambda ns: "x".join(str(n) for n in ns))
def test_component_solves_side_by_side_bitwise(hip, shape):
    """small grids: the three 3-D component solves on three streams, a hierarchy each, their cycles replayed as
    recorded graphs where a cycle leaves every array in place (the default up to 16 M points) against one after
    the other on one hierarchy (NDSM_HIP_NO_SIDE3D=1, a fresh context) - the same A and B bit for bit, over a
    sequence of calls with changing options.  Even ms on a grid whose level 1 runs the out-of-place smoother
    swaps that level's arrays an odd number of times per cycle: such a cycle must NOT be replayed."""
    import ndsm_amd
    L = hip.load_library()
    x, y, z, _A1, b1 = analytic_case(shape)
    b = b1 + 0.05 * np.random.default_rng(5).standard_normal(b1.shape)     # all three components iterate
    seq = [dict(), dict(ms=3, mean=True), dict(), dict(ms=4), dict(), dict(ms=2), dict(ms=1), dict()]

    def run():
        L.ndsm_hip_shutdown()                      # drop the cached context: the switch is read when it is built
        assert L.ndsm_hip_init(0) == 0
        return [ndsm_amd.vector_potential(x, y, z, b, **kw) for kw in seq]

    keep = os.environ.get("NDSM_HIP_NO_SIDE3D")
    try:
        os.environ["NDSM_HIP_NO_SIDE3D"] = "1"
        want = run()
        os.environ.pop("NDSM_HIP_NO_SIDE3D")
        got = run()
    finally:
        if keep is None:
            os.environ.pop("NDSM_HIP_NO_SIDE3D", None)
        else:
            os.environ["NDSM_HIP_NO_SIDE3D"] = keep
    for i, (w, g) in enumerate(zip(want, got)):
        assert w[0] == g[0], (i, seq[i])
        assert np.array_equal(w[1], g[1]) and np.array_equal(w[2], g[2]), (i, seq[i])
