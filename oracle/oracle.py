"""TEST INFRASTRUCTURE - not the product.

ctypes front-end shared by tests/, tests/golden/make_golden.py, bench.py's
cpu_baseline leg and __graft_entry__.smoke().  Two back-ends with one Python
API:

  Oracle("port")  -> oracle/liboracle.so       our C restatement (ndsm_oracle.c)
  Oracle("ref")   -> oracle/_ref/libndsm_refk.so + oracle/_ref/ndsmf.so
                     the reference itself (built by oracle/Makefile from
                     /root/reference/fortran; absent unless that was run)

Arrays follow the reference's Python convention (ndsm.py:161,210): numpy C
order with shape (nz, ny, nx) == Fortran (nx, ny, nz).  `nshape` arguments are
Fortran order [nx, ny(, nz)].
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(HERE, "liboracle.so")
REFK_LIB = os.path.join(HERE, "_ref", "libndsm_refk.so")
REF_LIB = os.path.join(HERE, "_ref", "ndsmf.so")



def usable_cpus(cap=16):
    """CPUs this process may really use: affinity mask, cgroup quota, and `cap`.
    A GPU box shows 256 logical CPUs under a 16-CPU quota; an OpenMP team of 256
    spinning threads there does not finish (libgomp spin-wait), so the checkers
    never run wider than this."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def _omp_env():
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("OMP_DYNAMIC", "FALSE")
    return int(os.environ["OMP_NUM_THREADS"])


_dp = ctypes.POINTER(ctypes.c_double)
_ip64 = ctypes.POINTER(ctypes.c_int64)


def have_ref():
    return os.path.exists(REFK_LIB) and os.path.exists(REF_LIB)


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _xyz(mesh):
    m = [_f64(v) for v in mesh]
    while len(m) < 3:
        m.append(np.zeros(2))
    return m


def uniform_mesh(nshape, h=None):
    """x = linspace(0,1,nx); y,z = arange(n)*dx (integration_test1.py:124-127)."""
    x = np.linspace(0.0, 1.0, int(nshape[0]))
    dx = (x[1] - x[0]) if h is None else h
    out = [x if h is None else np.arange(int(nshape[0])) * dx]
    for n in nshape[1:]:
        out.append(np.arange(int(n)) * dx)
    return out


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        self.threads = _omp_env()      # before the OpenMP runtime is loaded
        if kind == "port":
            self.lib = ctypes.CDLL(PORT_LIB)
            self.p = "orc_"
            self.lib.orc_set_quiet(1)
            self.vs = self.lib.orc_vector_solve
        elif kind == "ref":
            self.lib = ctypes.CDLL(REFK_LIB)
            self.p = "refk_"
            self.vs = ctypes.CDLL(REF_LIB).ndsm_vector_solve
        else:
            raise ValueError(kind)
        self.vs.restype = ctypes.c_int

    def _fn(self, name):
        return getattr(self.lib, self.p + name)

    @staticmethod
    def _shape(nshape):
        return np.asarray(nshape, dtype=np.int64)

    def ngrids(self, nshape):
        nmin = min(int(n) for n in nshape)
        return int(np.floor(np.log(nmin / 2.0) / np.log(2.0)))

    def hierarchy(self, nshape, mesh, ngrids=None):
        ns = self._shape(nshape)
        nd = len(ns)
        ng = self.ngrids(nshape) if ngrids is None else ngrids
        x, y, z = _xyz(mesh)
        shapes = np.zeros(nd * ng, dtype=np.int64)
        meshes = np.zeros(int(ns.sum()) * ng + 8)
        self._fn("hierarchy")(ctypes.c_int(nd), ns.ctypes.data_as(_ip64), ctypes.c_int(ng), _d(x),
                              _d(y), _d(z), shapes.ctypes.data_as(_ip64), _d(meshes))
        shapes = shapes.reshape(ng, nd)
        out, p = [], 0
        for l in range(ng):
            lv = []
            for d in range(nd):
                n = int(shapes[l, d])
                lv.append(meshes[p:p + n].copy())
                p += n
            out.append(lv)
        return shapes, out

    def relax3d(self, u, rhs, mesh, bcs, inplace=False):
        """one red+black sweep; inplace=True updates the caller's (float64, C-contiguous) array - timing runs"""
        u = _f64(u) if inplace else _f64(u).copy()
        rhs = _f64(rhs)
        ns = self._shape(u.shape[::-1])
        x, y, z = _xyz(mesh)
        self._fn("relax3d")(ns.ctypes.data_as(_ip64), _d(x), _d(y), _d(z), bcs.encode(), _d(rhs), _d(u))
        return u

    def residual3d(self, u, rhs, mesh, bcs):
        u = _f64(u)
        rhs = _f64(rhs)
        r = np.full_like(u, np.nan)
        ns = self._shape(u.shape[::-1])
        x, y, z = _xyz(mesh)
        self._fn("residual3d")(ns.ctypes.data_as(_ip64), _d(x), _d(y), _d(z), bcs.encode(), _d(rhs), _d(u), _d(r))
        return r

    def relax_nd(self, u, rhs, mesh, bcs):
        u = _f64(u).copy()
        rhs = _f64(rhs)
        ns = self._shape(u.shape[::-1])
        x, y, z = _xyz(mesh)
        self._fn("relax_nd")(ctypes.c_int(u.ndim), ns.ctypes.data_as(_ip64), _d(x), _d(y), _d(z),
                             bcs.encode(), _d(rhs), _d(u))
        return u

    def residual_nd(self, u, rhs, mesh, bcs):
        u = _f64(u)
        rhs = _f64(rhs)
        r = np.full_like(u, np.nan)
        ns = self._shape(u.shape[::-1])
        x, y, z = _xyz(mesh)
        self._fn("residual_nd")(ctypes.c_int(u.ndim), ns.ctypes.data_as(_ip64), _d(x), _d(y), _d(z),
                                bcs.encode(), _d(rhs), _d(u), _d(r))
        return r

    def restrict(self, u_f, nshape, mesh, id_f, ngrids=None):
        """u_f lives on level id_f (1-based) of the hierarchy rooted at nshape."""
        ns = self._shape(nshape)
        nd = len(ns)
        ng = self.ngrids(nshape) if ngrids is None else ngrids
        shapes, _ = self.hierarchy(nshape, mesh, ng)
        x, y, z = _xyz(mesh)
        u_f = _f64(u_f)
        assert u_f.shape == tuple(int(v) for v in shapes[id_f - 1][::-1])
        u_c = np.full(tuple(int(v) for v in shapes[id_f][::-1]), np.nan)
        self._fn("restrict")(ctypes.c_int(nd), ns.ctypes.data_as(_ip64), ctypes.c_int(ng), _d(x), _d(y),
                             _d(z), ctypes.c_int(id_f), _d(u_f), _d(u_c))
        return u_c

    def interp(self, u_c, nshape, mesh, id_f, ngrids=None):
        ns = self._shape(nshape)
        nd = len(ns)
        ng = self.ngrids(nshape) if ngrids is None else ngrids
        shapes, _ = self.hierarchy(nshape, mesh, ng)
        x, y, z = _xyz(mesh)
        u_c = _f64(u_c)
        assert u_c.shape == tuple(int(v) for v in shapes[id_f][::-1])
        u_f = np.full(tuple(int(v) for v in shapes[id_f - 1][::-1]), np.nan)
        if self.kind == "ref":
            self._fn("interp")(ctypes.c_int(nd), ns.ctypes.data_as(_ip64), ctypes.c_int(ng), _d(x), _d(y),
                               _d(z), ctypes.c_int(id_f), _d(u_c), _d(u_f))
        else:
            self._fn("interp")(ctypes.c_int(nd), ns.ctypes.data_as(_ip64), ctypes.c_int(ng), _d(x), _d(y),
                               _d(z), ctypes.c_int(id_f), _d(u_c), _d(u_f))
        return u_f

    def update_u(self, u_old, u_new):
        """Returns (max, mean) of |u_new - u_old| and overwrites u_new <- u_old."""
        u_old = _f64(u_old)
        assert u_new.dtype == np.float64 and u_new.flags.c_contiguous
        m = np.zeros(2)
        self._fn("update_u")(ctypes.c_int64(u_old.size), _d(u_old), _d(u_new), _d(m))
        return float(m[0]), float(m[1])

    def vcycle(self, u, rhs, mesh, bcs, ms=5, ex_tol=1e-13, du_max=True, nmax_exact=10000, ngrids=None):
        u = _f64(u).copy()
        rhs = _f64(rhs)
        ns = self._shape(u.shape[::-1])
        ng = self.ngrids(ns) if ngrids is None else ngrids
        x, y, z = _xyz(mesh)
        self._fn("vcycle")(ctypes.c_int(u.ndim), ns.ctypes.data_as(_ip64), ctypes.c_int(ng), _d(x), _d(y),
                           _d(z), bcs.encode(), ctypes.c_int(ms), ctypes.c_double(ex_tol),
                           ctypes.c_int(1 if du_max else 0), ctypes.c_int(nmax_exact), _d(rhs), _d(u))
        return u

    def solve_bvp(self, u, rhs, mesh, bcs, ms=5, ex_tol=1e-13, du_max=True, nmax_exact=10000,
                  vc_tol=1e-10, nmax=1024, ngrids=None, hist_len=0):
        """Returns (ierr, u, du_last[, hist, ncycles, exact_sweeps] for the port)."""
        u = _f64(u).copy()
        rhs = _f64(rhs).copy()
        ns = self._shape(u.shape[::-1])
        ng = self.ngrids(ns) if ngrids is None else ngrids
        x, y, z = _xyz(mesh)
        du_last = ctypes.c_double(0)
        fn = self._fn("solve_bvp")
        fn.restype = ctypes.c_int
        args = [ctypes.c_int(u.ndim), ns.ctypes.data_as(_ip64), ctypes.c_int(ng), _d(x), _d(y), _d(z),
                bcs.encode(), ctypes.c_int(ms), ctypes.c_double(ex_tol), ctypes.c_int(1 if du_max else 0),
                ctypes.c_int(nmax_exact), ctypes.c_double(vc_tol), ctypes.c_int(nmax), _d(rhs), _d(u),
                ctypes.byref(du_last)]
        if self.kind == "port":
            hist = np.zeros(max(hist_len, 1))
            nc = ctypes.c_int(0)
            sw = ctypes.c_int64(0)
            ierr = fn(*args, _d(hist), ctypes.c_int(hist_len), ctypes.byref(nc), ctypes.byref(sw))
            return ierr, u, du_last.value, hist[:min(hist_len, nc.value)].copy(), nc.value, sw.value
        ierr = fn(*args)
        return ierr, u, du_last.value

    def vector_potential(self, x, y, z, b, niterex_max=10000, ncycles_max=1024, ex_tol=1e-13, vc_tol=1e-10,
                         ms=5, mean=False, Ainit=None):
        """Same call contract as the reference's ndsm.py:66-210 (slots per
        ndsm_vector_potential.f90:40-57)."""
        nshape = np.array(b.shape[::-1], dtype=np.intc)
        ioptc = np.zeros(16, dtype=np.intc)
        ropt = np.zeros(16, dtype=np.float64)
        ioptc[0] = ms
        ioptc[1] = ncycles_max
        ioptc[7] = niterex_max
        ioptc[6] = 0 if mean else 1
        ropt[0] = vc_tol
        ropt[1] = ex_tol
        A = np.zeros(b.size) if Ainit is None else _f64(Ainit).ravel().copy()
        bb = _f64(b).ravel().copy()
        x, y, z = _f64(x), _f64(y), _f64(z)
        ierr = self.vs(ctypes.c_size_t(bb.size), nshape.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                       ioptc.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _d(ropt), _d(x), _d(y), _d(z),
                       _d(A), _d(bb))
        return ierr, A.reshape(b.shape), bb.reshape(b.shape), ioptc, ropt
