"""ctypes binding of libndsm_hip.so (C ABI: include/ndsm_hip.h).

Fails loudly: a missing library raises NdsmHipError at load time and a missing
GPU raises it at the first call that needs the device.  Nothing here computes
on the CPU.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class NdsmHipError(RuntimeError):
    pass


def lib_path():
    # NDSM_HIP_LIB: development override (A/B builds of the same library)
    return os.environ.get("NDSM_HIP_LIB") or os.path.join(HERE, "lib", "libndsm_hip.so")


def load_library(path=None):
    """Load libndsm_hip.so once and declare its prototypes."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or lib_path()
    if not os.path.exists(p):
        raise NdsmHipError(f"{p} not found - build it with `make -C ndsm_amd` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    # RTLD_DEEPBIND: the library and its own dependencies (/opt/rocm's libamdhip64, librccl) are
    # searched BEFORE the global scope.  Without it, a process that imported PyTorch first would
    # bind our hip*/nccl* calls to the second ROCm stack torch bundles (torch loads it RTLD_GLOBAL).
    mode = os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0)
    # multi-process runs (one rank per GPU, RCCL): this pool's host driver does dmabuf IPC only; without the
    # variable RCCL's peer set-up fails with "hipIpcGetMemHandle: invalid argument".  It is read when the HSA
    # runtime comes up, i.e. at the first HIP call below - a caller's own value is left alone.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = ctypes.CDLL(p, mode=mode)
    L.ndsm_vector_solve.restype = ctypes.c_int
    L.ndsm_hip_last_error.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.ndsm_hip_last_error.restype = None
    L.ndsm_hip_timer_stop.argtypes = [_dp]
    L.ndsm_hip_mg_create.argtypes = [ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.ndsm_hip_mg_destroy.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_mg_levels.argtypes = [ctypes.c_void_p, ctypes.c_int, _ip]
    L.ndsm_hip_mg_set_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_mg_set_precision.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_mg_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    L.ndsm_hip_mg_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    L.ndsm_hip_mg_zero_rhs.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_mg_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ndsm_hip_mg_vcycle.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_mg_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, _dp, _ip, _dp, ctypes.c_int]
    L.ndsm_hip_mg_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.ndsm_hip_poisson_solve.argtypes = [ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_char_p, _ip, _dp, _dp, _dp, _dp,
                                         ctypes.c_int]
    L.ndsm_hip_dist_unique_id.argtypes = [ctypes.c_char_p]
    L.ndsm_hip_dist_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    L.ndsm_hip_slab_plan.argtypes = [_ip, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int, _ip]
    L.ndsm_hip_world_create.argtypes = [_ip, _dp, _dp, _dp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_void_p)]
    L.ndsm_hip_world_destroy.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_world_nlocal.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_world_dist_levels.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_world_slab.argtypes = [ctypes.c_void_p, ctypes.c_int, _ip]
    L.ndsm_hip_world_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_int]
    L.ndsm_hip_world_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    L.ndsm_hip_world_relax.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_world_zero_rhs.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_world_vcycle.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_world_set_precision.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_world_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, _dp, _ip, _dp, ctypes.c_int]
    if path is None:
        _LIB = L
    return L


def bound_libs(L=None):
    """paths of the HIP and RCCL shared objects our calls are bound to"""
    L = L or load_library()
    buf = ctypes.create_string_buffer(1024)
    L.ndsm_hip_bound_libs(buf, 1024)
    return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))


def last_error(L=None):
    L = L or load_library()
    buf = ctypes.create_string_buffer(512)
    L.ndsm_hip_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def _check(rc, what, L=None):
    if rc != 0:
        raise NdsmHipError(f"{what} failed with code {rc}: {last_error(L)}")


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ops / buffers of ndsm_hip_mg_op, ndsm_hip_mg_upload (ndsmh_mg.f90)
BUF_U, BUF_RHS, BUF_R = 0, 1, 2
OP_RELAX, OP_RESIDUAL, OP_RESTRICT, OP_PROLONG, OP_EXACT, OP_RELAX_COLOR, OP_RELAX_FUSED, _OP_RETIRED_7, OP_RELAX_RES, OP_RELAX_RES_FUSED = range(10)


class MGSolver:
    """Persistent device-resident multigrid solver (additive API).

    nshape: Fortran order [nx, ny(, nz)]; arrays are numpy C order (nz, ny, nx).
    bcs: 2*ndim letters, lower faces then upper faces, 'D' or 'N'.
    """

    def __init__(self, nshape, mesh, bcs, ngrids=0, ms=5, ex_tol=1e-13, du_max=True, nmax_exact=10000, lib=None):
        self.L = lib or load_library()
        self.ndim = len(nshape)
        ns = np.asarray(nshape, dtype=np.intc)
        m = [_f64(v) for v in mesh]
        while len(m) < 3:
            m.append(np.zeros(2))
        self.h = ctypes.c_void_p()
        rc = self.L.ndsm_hip_mg_create(self.ndim, ns.ctypes.data_as(_ip), _d(m[0]), _d(m[1]), _d(m[2]),
                                       bcs.encode(), int(ngrids), int(ms), float(ex_tol), 1 if du_max else 0,
                                       int(nmax_exact), ctypes.byref(self.h))
        _check(rc, "ndsm_hip_mg_create", self.L)
        shp = np.zeros((32, 3), dtype=np.intc)
        self.ngrids = self.L.ndsm_hip_mg_levels(self.h, 32, shp.ctypes.data_as(_ip))
        self.shapes = [tuple(int(v) for v in shp[l, :self.ndim]) for l in range(self.ngrids)]  # Fortran order

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.ndsm_hip_mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _npshape(self, level):
        return tuple(self.shapes[level - 1][::-1])

    def upload(self, level, which, arr):
        a = _f64(arr)
        assert a.shape == self._npshape(level if which != BUF_R else 1), (a.shape, self._npshape(level))
        _check(self.L.ndsm_hip_mg_upload(self.h, level, which, _d(a)), "upload", self.L)

    def download(self, level, which, shape_level=None):
        out = np.empty(self._npshape(shape_level or (level if which != BUF_R else 1)))
        if which == BUF_R and shape_level:
            full = np.empty(self._npshape(1))
            _check(self.L.ndsm_hip_mg_download(self.h, 1, which, _d(full)), "download", self.L)
            return full.ravel()[:out.size].reshape(out.shape).copy()
        _check(self.L.ndsm_hip_mg_download(self.h, level, which, _d(out)), "download", self.L)
        return out

    def zero_rhs(self):
        """rhs(1) == 0: the Laplace case of the vector potential; kernels then skip the rhs read"""
        _check(self.L.ndsm_hip_mg_zero_rhs(self.h), "zero_rhs", self.L)

    def set_precision(self, mode):
        """0 fp64, 1 mixed where level 1 is large, 2 mixed wherever the fp32 kernels apply;
        returns True if solve() will run in mixed precision"""
        rc = self.L.ndsm_hip_mg_set_precision(self.h, int(mode))
        if rc < 0:
            raise NdsmHipError(f"bad precision mode {mode}")
        return rc == 1

    def op(self, op, level, count=1):
        _check(self.L.ndsm_hip_mg_op(self.h, op, level, count), f"op {op}", self.L)

    def vcycle(self, n=1):
        _check(self.L.ndsm_hip_mg_vcycle(self.h, n), "vcycle", self.L)

    def solve(self, vc_tol=1e-10, nmax=1024, hist_len=0):
        du = ctypes.c_double(0)
        nc = ctypes.c_int(0)
        hist = np.zeros(max(hist_len, 1))
        ierr = self.L.ndsm_hip_mg_solve(self.h, float(vc_tol), int(nmax), ctypes.byref(du), ctypes.byref(nc), _d(hist),
                                        int(hist_len))
        if ierr >= 9000:
            _check(ierr, "ndsm_hip_mg_solve", self.L)
        return ierr, du.value, nc.value, hist[:min(hist_len, nc.value)].copy()

    def info(self):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self.L.ndsm_hip_mg_info(self.h, ctypes.byref(a), ctypes.byref(b)), "info", self.L)
        return a.value, b.value

    def sync(self):
        _check(self.L.ndsm_hip_sync(), "sync", self.L)

    def timed(self, fn):
        """Run fn() between two HIP events on the library stream; returns ms."""
        _check(self.L.ndsm_hip_timer_start(), "timer_start", self.L)
        fn()
        ms = ctypes.c_double(0)
        _check(self.L.ndsm_hip_timer_stop(ctypes.byref(ms)), "timer_stop", self.L)
        return ms.value


class VecPot:
    """Persistent vector-potential solver (additive): the grid hierarchies, transfer tables
    and device arrays are built once per (shape, mesh) and reused by every solve().

    x, y, z: mesh vectors; shape of the fields: numpy (3, nz, ny, nx)."""

    def __init__(self, x, y, z, ngrids=0, lib=None):
        self.L = lib or load_library()
        self.x, self.y, self.z = _f64(x), _f64(y), _f64(z)
        self.nshape4 = np.array([len(self.x), len(self.y), len(self.z), 3], dtype=np.intc)
        self.ngrids = int(ngrids)
        self.h = ctypes.c_void_p()
        self.L.ndsm_hip_vecpot_create.argtypes = [_ip, _dp, _dp, _dp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        self.L.ndsm_hip_vecpot_solve.argtypes = [ctypes.c_void_p, _ip, _dp, _dp, _dp]
        self.L.ndsm_hip_vecpot_solve_device.argtypes = [ctypes.c_void_p, _ip, _dp, ctypes.c_void_p, ctypes.c_void_p]
        self.L.ndsm_hip_vecpot_destroy.argtypes = [ctypes.c_void_p]
        self.L.ndsm_hip_device_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        self.L.ndsm_hip_device_free.argtypes = [ctypes.c_void_p]
        self.L.ndsm_hip_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.L.ndsm_hip_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        rc = self.L.ndsm_hip_vecpot_create(self.nshape4.ctypes.data_as(_ip), _d(self.x), _d(self.y), _d(self.z),
                                           self.ngrids, ctypes.byref(self.h))
        _check(rc, "ndsm_hip_vecpot_create", self.L)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.ndsm_hip_vecpot_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _options(self, niterex_max, ncycles_max, ex_tol, vc_tol, ms, mean, mixed_precision, flxcrl):
        L = self.L
        ioptc = np.zeros(16, dtype=np.intc)
        ropt = np.zeros(16)
        ioptc[L.get_iopt_ms()] = ms
        ioptc[L.get_iopt_ncycles()] = ncycles_max
        ioptc[L.get_iopt_iopt_nmaxex()] = niterex_max
        ioptc[L.get_iopt_dumax()] = 0 if mean else 1
        ioptc[L.get_iopt_ngrids()] = self.ngrids
        ioptc[L.get_iopt_prec()] = int(mixed_precision)
        ioptc[4] = 1 if flxcrl else 0          # IOPT_FLXCRL (ndsm_vector_potential.f90:44; no getter in the reference)
        ropt[L.get_ropt_vtol()] = vc_tol
        ropt[L.get_ropt_ctol()] = ex_tol
        return ioptc, ropt

    def solve(self, b, a_init=None, niterex_max=10000, ncycles_max=1024, ex_tol=1e-13, vc_tol=1e-10, ms=5, mean=False,
              mixed_precision=False, flxcrl=False, device=False):
        """b: (3,nz,ny,nx); returns (ierr, A, B) like ndsm.vector_potential.  device=True: the fields are
        staged in device memory first and the device-resident entry point runs (tests of that path)."""
        ioptc, ropt = self._options(niterex_max, ncycles_max, ex_tol, vc_tol, ms, mean, mixed_precision, flxcrl)
        shape = tuple(int(v) for v in self.nshape4[::-1])
        B = _f64(b).reshape(-1).copy()
        assert B.size == int(np.prod(shape)), (b.shape, shape)
        A = np.zeros(B.size) if a_init is None else _f64(a_init).reshape(-1).copy()
        if not device:
            ierr = self.L.ndsm_hip_vecpot_solve(self.h, ioptc.ctypes.data_as(_ip), _d(ropt), _d(A), _d(B))
        else:
            dA, dB = ctypes.c_void_p(), ctypes.c_void_p()
            _check(self.L.ndsm_hip_device_alloc(A.nbytes, ctypes.byref(dA)), "device_alloc", self.L)
            _check(self.L.ndsm_hip_device_alloc(B.nbytes, ctypes.byref(dB)), "device_alloc", self.L)
            try:
                _check(self.L.ndsm_hip_memcpy_h2d(dA, A.ctypes.data, A.nbytes), "h2d", self.L)
                _check(self.L.ndsm_hip_memcpy_h2d(dB, B.ctypes.data, B.nbytes), "h2d", self.L)
                ierr = self.L.ndsm_hip_vecpot_solve_device(self.h, ioptc.ctypes.data_as(_ip), _d(ropt), dA, dB)
                _check(self.L.ndsm_hip_memcpy_d2h(A.ctypes.data, dA, A.nbytes), "d2h", self.L)
                _check(self.L.ndsm_hip_memcpy_d2h(B.ctypes.data, dB, B.nbytes), "d2h", self.L)
            finally:
                self.L.ndsm_hip_device_free(dA)
                self.L.ndsm_hip_device_free(dB)
        if ierr >= 9000:
            _check(ierr, "ndsm_hip_vecpot_solve", self.L)
        self.last_ioptc, self.last_ropt = ioptc, ropt
        return ierr, A.reshape(shape), B.reshape(shape)


SLAB_FIELDS = ("rank", "z0", "z1", "g", "nloc", "k0", "ck0", "ck1", "pk0", "pk1", "cb0", "cb1")


def slab_plan(nshape, mesh, nranks, ngrids=0, lib=None):
    """The z-slab plan every rank derives (host arithmetic only - works without a GPU)."""
    L = lib or load_library()
    ns = np.asarray(nshape, dtype=np.intc)
    m = [_f64(v) for v in mesh]
    out = np.zeros((nranks, 12), dtype=np.intc)
    rc = L.ndsm_hip_slab_plan(ns.ctypes.data_as(_ip), _d(m[0]), _d(m[1]), _d(m[2]), int(ngrids), int(nranks),
                              out.ctypes.data_as(_ip))
    if rc != 0:
        raise NdsmHipError(f"ndsm_hip_slab_plan failed with code {rc} (slabs thinner than the ghost depth?)")
    return [dict(zip(SLAB_FIELDS, (int(v) for v in row))) for row in out]


class World:
    """Level 1 split into z-slabs.  rank >= 0: this process owns slab `rank` (RCCL);
    rank < 0: loop-back, all slabs on this GPU.  Arrays are numpy (nz, ny, nx)."""

    def __init__(self, nshape, mesh, bcs, nranks, rank=-1, ngrids=0, ms=5, ex_tol=1e-13, du_max=True,
                 nmax_exact=10000, lib=None):
        self.L = lib or load_library()
        self.nshape = [int(v) for v in nshape]
        ns = np.asarray(nshape, dtype=np.intc)
        m = [_f64(v) for v in mesh]
        self.h = ctypes.c_void_p()
        rc = self.L.ndsm_hip_world_create(ns.ctypes.data_as(_ip), _d(m[0]), _d(m[1]), _d(m[2]), bcs.encode(),
                                          int(ngrids), int(ms), float(ex_tol), 1 if du_max else 0, int(nmax_exact),
                                          int(nranks), int(rank), ctypes.byref(self.h))
        _check(rc, "ndsm_hip_world_create", self.L)
        self.nlocal = self.L.ndsm_hip_world_nlocal(self.h)
        self.dist_levels = self.L.ndsm_hip_world_dist_levels(self.h)
        self.slabs = []
        for i in range(1, self.nlocal + 1):
            info = np.zeros(12, dtype=np.intc)
            _check(self.L.ndsm_hip_world_slab(self.h, i, info.ctypes.data_as(_ip)), "world_slab", self.L)
            self.slabs.append(dict(zip(SLAB_FIELDS, (int(v) for v in info))))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.ndsm_hip_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, which, arr):
        """arr: the GLOBAL field (nz, ny, nx); every local slab takes its window incl. ghosts"""
        a = _f64(arr)
        assert a.shape == tuple(self.nshape[::-1])
        for i in range(1, self.nlocal + 1):
            _check(self.L.ndsm_hip_world_upload(self.h, i, which, _d(a), 0, a.shape[0]), "world_upload", self.L)

    def upload_window(self, ilocal, which, window, gz0):
        """window: planes [gz0, gz0 + window.shape[0]) of the global field"""
        a = _f64(window)
        _check(self.L.ndsm_hip_world_upload(self.h, ilocal, which, _d(a), int(gz0), a.shape[0]), "world_upload", self.L)

    def download(self, which, out=None):
        """owned planes of every local slab -> (global-shaped array, filled where owned)"""
        nx, ny, nz = self.nshape
        if out is None:
            out = np.full((nz, ny, nx), np.nan)
        for i, sl in enumerate(self.slabs, start=1):
            buf = np.empty((sl["z1"] - sl["z0"], ny, nx))
            _check(self.L.ndsm_hip_world_download(self.h, i, which, _d(buf)), "world_download", self.L)
            out[sl["z0"]:sl["z1"]] = buf
        return out

    def relax(self, n=1):
        _check(self.L.ndsm_hip_world_relax(self.h, n), "world_relax", self.L)

    def set_precision(self, mode):
        """0 fp64, != 0 mixed (fp64 residual, fp32 correction V-cycle on the level-1 slabs);
        returns True if solve() will run in mixed precision"""
        rc = self.L.ndsm_hip_world_set_precision(self.h, int(mode))
        if rc < 0:
            raise NdsmHipError(f"bad precision mode {mode}")
        return rc == 1

    def zero_rhs(self):
        """Declare rhs == 0 on level 1 (Laplace problem): the kernels stop reading it; same bits."""
        _check(self.L.ndsm_hip_world_zero_rhs(self.h), "world_zero_rhs", self.L)

    def vcycle(self, n=1):
        _check(self.L.ndsm_hip_world_vcycle(self.h, n), "world_vcycle", self.L)

    def solve(self, vc_tol=1e-10, nmax=1024, hist_len=0):
        du = ctypes.c_double(0)
        nc = ctypes.c_int(0)
        hist = np.zeros(max(hist_len, 1))
        ierr = self.L.ndsm_hip_world_solve(self.h, float(vc_tol), int(nmax), ctypes.byref(du), ctypes.byref(nc),
                                           _d(hist), int(hist_len))
        if ierr >= 9000:
            _check(ierr, "ndsm_hip_world_solve", self.L)
        return ierr, du.value, nc.value, hist[:min(hist_len, nc.value)].copy()

    def sync(self):
        _check(self.L.ndsm_hip_sync(), "sync", self.L)

    def timed(self, fn):
        _check(self.L.ndsm_hip_timer_start(), "timer_start", self.L)
        fn()
        ms = ctypes.c_double(0)
        _check(self.L.ndsm_hip_timer_stop(ctypes.byref(ms)), "timer_stop", self.L)
        return ms.value


def dist_unique_id(lib=None):
    L = lib or load_library()
    buf = ctypes.create_string_buffer(128)
    _check(L.ndsm_hip_dist_unique_id(buf), "ndsm_hip_dist_unique_id", L)
    return buf.raw


def dist_init(rank, nranks, uid, lib=None):
    L = lib or load_library()
    _check(L.ndsm_hip_dist_init(int(rank), int(nranks), ctypes.c_char_p(uid)), "ndsm_hip_dist_init", L)


def dist_info(lib=None):
    """(rank, nranks) as the RCCL communicator itself reports them; nranks == 0: none is up"""
    L = lib or load_library()
    r, n = ctypes.c_int(0), ctypes.c_int(0)
    _check(L.ndsm_hip_dist_info(ctypes.byref(r), ctypes.byref(n)), "ndsm_hip_dist_info", L)
    return r.value, n.value


def dist_finalize(lib=None):
    L = lib or load_library()
    _check(L.ndsm_hip_dist_finalize(), "ndsm_hip_dist_finalize", L)


def poisson_solve(u, rhs, mesh, bcs, ms=5, ex_tol=1e-13, du_max=True, nmax_exact=10000, vc_tol=1e-10, nmax=1024,
                  ngrids=0, hist_len=0, lib=None, precision=0):
    """laplace(u) = rhs on the device.  u: initial guess + Dirichlet data,
    numpy order (nz, ny, nx).  precision: 0 fp64 (reference arithmetic), 1 mixed (fp64 residual,
    fp32 correction V-cycle on level 1) where level 1 is large enough, 2 mixed wherever possible.
    Returns (ierr, u, du_last, hist, ncycles)."""
    L = lib or load_library()
    u = _f64(u).copy()
    nd = u.ndim
    ns = np.asarray(u.shape[::-1], dtype=np.intc)
    m = [_f64(v) for v in mesh]
    while len(m) < 3:
        m.append(np.zeros(2))
    iopt = np.zeros(16, dtype=np.intc)
    ropt = np.zeros(16)
    iopt[L.get_iopt_ms()] = ms
    iopt[L.get_iopt_ncycles()] = nmax
    iopt[L.get_iopt_iopt_nmaxex()] = nmax_exact
    iopt[L.get_iopt_dumax()] = 1 if du_max else 0
    iopt[L.get_iopt_ngrids()] = ngrids
    iopt[L.get_iopt_prec()] = precision
    ropt[L.get_ropt_vtol()] = vc_tol
    ropt[L.get_ropt_ctol()] = ex_tol
    hist = np.zeros(max(hist_len, 1))
    rp = _d(_f64(rhs)) if rhs is not None else None
    rhs_keep = _f64(rhs) if rhs is not None else None
    rp = _d(rhs_keep) if rhs_keep is not None else None
    ierr = L.ndsm_hip_poisson_solve(nd, ns.ctypes.data_as(_ip), _d(m[0]), _d(m[1]), _d(m[2]), bcs.encode(),
                                    iopt.ctypes.data_as(_ip), _d(ropt), _d(u), rp, _d(hist), int(hist_len))
    if ierr >= 9000:
        _check(ierr, "ndsm_hip_poisson_solve", L)
    nc = int(iopt[L.get_iopt_ncyc_out()])
    return ierr, u, float(ropt[L.get_ropt_dulast()]), hist[:min(hist_len, nc)].copy(), nc
