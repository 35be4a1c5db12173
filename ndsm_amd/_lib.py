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
    return os.path.join(HERE, "lib", "libndsm_hip.so")


def load_library(path=None):
    """Load libndsm_hip.so once and declare its prototypes."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or lib_path()
    if not os.path.exists(p):
        raise NdsmHipError(f"{p} not found - build it with `make -C ndsm_amd` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(p)
    L.ndsm_vector_solve.restype = ctypes.c_int
    L.ndsm_hip_last_error.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.ndsm_hip_last_error.restype = None
    L.ndsm_hip_timer_stop.argtypes = [_dp]
    L.ndsm_hip_mg_create.argtypes = [ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.ndsm_hip_mg_destroy.argtypes = [ctypes.c_void_p]
    L.ndsm_hip_mg_levels.argtypes = [ctypes.c_void_p, ctypes.c_int, _ip]
    L.ndsm_hip_mg_set_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_mg_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    L.ndsm_hip_mg_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    L.ndsm_hip_mg_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ndsm_hip_mg_vcycle.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndsm_hip_mg_solve.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, _dp, _ip, _dp, ctypes.c_int]
    L.ndsm_hip_mg_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.ndsm_hip_poisson_solve.argtypes = [ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_char_p, _ip, _dp, _dp, _dp, _dp,
                                         ctypes.c_int]
    if path is None:
        _LIB = L
    return L


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
OP_RELAX, OP_RESIDUAL, OP_RESTRICT, OP_PROLONG, OP_EXACT, OP_RELAX_COLOR, OP_RELAX_FUSED = range(7)


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


def poisson_solve(u, rhs, mesh, bcs, ms=5, ex_tol=1e-13, du_max=True, nmax_exact=10000, vc_tol=1e-10, nmax=1024,
                  ngrids=0, hist_len=0, lib=None):
    """laplace(u) = rhs on the device.  u: initial guess + Dirichlet data,
    numpy order (nz, ny, nx).  Returns (ierr, u, du_last, hist, ncycles)."""
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
