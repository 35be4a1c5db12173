"""Python front-end with the call contract of the reference's ndsm.py.

`vector_potential` has the reference's signature, defaults and return triple
(ndsm.py:66-210): it discovers the option slots through the library's getters
(ndsm.py:155-174), hands Fortran-ordered shape [nx,ny,nz,3] (ndsm.py:161) and
flat float64 buffers to `ndsm_vector_solve`, and reshapes the results back to
(3,nz,ny,nx).  The one difference is the default library: ndsm_amd's own
libndsm_hip.so instead of a sys.path search for "ndsmf.so" - pass `libpath`
(or `libname` to search) to load anything else that exports the same C ABI,
e.g. the reference build itself.
"""
import ctypes
import os
import sys

import numpy as np
from numpy import ctypeslib as nct

from ._lib import lib_path, NdsmHipError


def get_lib_path(libname):
    """All files called `libname` below the entries of sys.path (ndsm.py:42-62)."""
    hits = set()
    for base in sys.path:
        for root, _dirs, files in os.walk(base):
            if libname in files:
                hits.add(os.path.join(root, libname))
    return list(hits)


def vector_potential(x, y, z, b, niterex_max=10000, ncycles_max=1024, ex_tol=1e-13, vc_tol=1e-10, ms=5, mean=False,
                     libname=None, libpath=None, debug=False, mixed_precision=False):
    """Vector potential A and B = curl A of the current-free field whose normal
    component on the six box faces is taken from `b` (3,nz,ny,nx).

    mixed_precision (additive; the reference has no such option): run the three 3-D solves as
    fp64 residual + fp32 correction V-cycle (option slot get_iopt_prec(), BASELINE config[4]).

    Returns (ierr, A, B) with A, B shaped (3,nz,ny,nx); ierr != 0 flags a V-cycle
    iteration that did not reach vc_tol.  Device / runtime failures (no MI355X visible, out of
    HBM, ...) come back the way the reference reports everything - as the return code, here
    >= 9001, with A left at zero and the text on stderr and in `_lib.last_error()` - so that
    callers written against the reference's (ierr, A, B) contract keep working; the additive
    entry points of this package (MGSolver, poisson_solve, vector_potential_slab) raise instead.
    """
    if libpath is None:
        if libname is None:
            libpath = lib_path()
        else:
            found = get_lib_path(libname)
            if len(found) == 0:
                raise ValueError("Could not locate {:s}".format(libname))
            if len(found) > 1:
                raise ValueError("More than once instance of {:s} found\n {:s}".format(libname, str(found)))
            libpath = found[0]
    if not os.path.exists(libpath):
        raise NdsmHipError(f"{libpath} not found - build it with `make -C ndsm_amd`; there is no CPU fallback")
    try:
        # same dlopen mode as _lib.load_library: RTLD_DEEPBIND keeps the library's hip*/nccl* calls on the
        # ROCm stack it was linked against even when PyTorch's bundled one is already in the process (the
        # FIRST dlopen of an object fixes its mode for the life of the process, and this may be it).  A CDLL
        # object of its own: the ndpointer prototypes set below stay private to this function.
        lib = ctypes.CDLL(libpath, mode=os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
    except OSError as exc:
        raise ValueError("Could not load library at " + libpath) from exc

    vec_i = nct.ndpointer(np.intc, ndim=1, flags=("C", "A", "W"))
    vec_d = nct.ndpointer(np.float64, ndim=1, flags=("C", "A", "W"))
    lib.ndsm_vector_solve.argtypes = [ctypes.c_size_t, vec_i, vec_i, vec_d, vec_d, vec_d, vec_d, vec_d, vec_d]
    lib.ndsm_vector_solve.restype = ctypes.c_int

    nopt = lib.get_iopt_len()
    nshape = np.array(b.shape[::-1], dtype=np.intc)
    ioptc = np.zeros(nopt, dtype=np.intc)
    ropt = np.zeros(nopt, dtype=np.float64)

    slots = {"ms": lib.get_iopt_ms(), "ncycles": lib.get_iopt_ncycles(), "nmaxex": lib.get_iopt_iopt_nmaxex(),
             "debug": lib.get_iopt_debug(), "dumax": lib.get_iopt_dumax(), "vtol": lib.get_ropt_vtol(),
             "ctol": lib.get_ropt_ctol()}
    if any(v < 0 or v >= nopt for v in slots.values()):
        raise Exception("Option vector (IOPT) index out of bounds. This shouldn't occur.")
    ioptc[slots["ms"]] = ms
    ioptc[slots["ncycles"]] = ncycles_max
    ioptc[slots["nmaxex"]] = niterex_max
    ropt[slots["vtol"]] = vc_tol
    ropt[slots["ctol"]] = ex_tol
    ioptc[slots["debug"]] = lib.get_iopt_true() if debug else lib.get_iopt_false()
    ioptc[slots["dumax"]] = lib.get_iopt_false() if mean else lib.get_iopt_true()
    if mixed_precision:
        if not hasattr(lib, "get_iopt_prec"):
            raise NdsmHipError(f"{libpath} has no mixed-precision option slot (get_iopt_prec): it is not libndsm_hip "
                               "- the reference build only knows fp64")
        ioptc[lib.get_iopt_prec()] = int(mixed_precision)    # True/1: where level 1 is large; 2: wherever the fp32 kernels apply

    apot = np.zeros(b.size, dtype=np.float64)
    if hasattr(lib, "ndsm_hip_init") and b.ndim == 4 and b.shape[0] == 3 and min(b.shape[1:]) >= 2:
        # libndsm_hip reads nothing of B but its normal component on the six faces (as the reference does,
        # ndsm_vector_potential.f90:283-299) and overwrites all of it: hand over a fresh buffer that carries
        # just those faces instead of the reference's 3-array copy b.flatten() (ndsm.py:177; 3 GiB at 512^3)
        bflat = np.empty(b.size, dtype=b.dtype)
        bv = bflat.reshape(b.shape)
        bv[0][:, :, 0], bv[0][:, :, -1] = b[0][:, :, 0], b[0][:, :, -1]
        bv[1][:, 0, :], bv[1][:, -1, :] = b[1][:, 0, :], b[1][:, -1, :]
        bv[2][0], bv[2][-1] = b[2][0], b[2][-1]
    else:
        bflat = b.flatten()
    ierr = lib.ndsm_vector_solve(ctypes.c_size_t(b.size), nshape, ioptc, ropt, x, y, z, apot, bflat)
    if ierr >= 9001:
        bflat = b.flatten()          # device / runtime failure: B comes back as it went in
    return ierr, apot.reshape(nshape[::-1]), bflat.reshape(nshape[::-1])


def vector_potential_slab(x, y, z, b_slab, rank, nranks, niterex_max=10000, ncycles_max=1024, ex_tol=1e-13, vc_tol=1e-10,
                          ms=5, mean=False, debug=False, a_init=None, lib=None, mixed_precision=False):
    """`vector_potential` on a z-slab decomposition (additive; BASELINE config[4]): one process per
    GPU, called collectively by every rank of the communicator set up with `_lib.dist_init`.

    x, y, z: the GLOBAL mesh vectors.  b_slab: this rank's planes of the field, shaped
    (3, z1-z0, ny, nx), [z0, z1) being rank `rank`'s entry of `_lib.slab_plan([nx,ny,nz], mesh, nranks)`.
    a_init: initial guess of the same shape (default zeros, as vector_potential passes).
    Returns (ierr, A_slab, B_slab) - the same bits the single-GPU call returns for those planes."""
    from . import _lib
    L = lib or _lib.load_library()
    b_slab = np.ascontiguousarray(b_slab, dtype=np.float64)
    nzl = b_slab.shape[1]
    nshape = np.array([len(x), len(y), len(z), 3], dtype=np.intc)
    assert b_slab.shape == (3, nzl, len(y), len(x)), b_slab.shape
    nopt = L.get_iopt_len()
    ioptc = np.zeros(nopt, dtype=np.intc)
    ropt = np.zeros(nopt, dtype=np.float64)
    ioptc[L.get_iopt_ms()] = ms
    ioptc[L.get_iopt_ncycles()] = ncycles_max
    ioptc[L.get_iopt_iopt_nmaxex()] = niterex_max
    ropt[L.get_ropt_vtol()] = vc_tol
    ropt[L.get_ropt_ctol()] = ex_tol
    ioptc[L.get_iopt_debug()] = L.get_iopt_true() if debug else L.get_iopt_false()
    ioptc[L.get_iopt_dumax()] = L.get_iopt_false() if mean else L.get_iopt_true()
    if mixed_precision:
        ioptc[L.get_iopt_prec()] = int(mixed_precision)
    apot = np.zeros(b_slab.size) if a_init is None else np.ascontiguousarray(a_init, dtype=np.float64).flatten()
    assert apot.size == b_slab.size
    bflat = b_slab.flatten()
    xs, ys, zs = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, z))
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    L.ndsm_hip_world_vector_solve.restype = ctypes.c_int
    L.ndsm_hip_world_vector_solve.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, dp, dp, dp, dp, dp, dp]
    ierr = L.ndsm_hip_world_vector_solve(int(rank), int(nranks), nshape.ctypes.data_as(ip), ioptc.ctypes.data_as(ip),
                                         ropt.ctypes.data_as(dp), xs.ctypes.data_as(dp), ys.ctypes.data_as(dp),
                                         zs.ctypes.data_as(dp), apot.ctypes.data_as(dp), bflat.ctypes.data_as(dp))
    if ierr >= 9000:
        raise NdsmHipError(f"ndsm_hip_world_vector_solve failed with code {ierr}: {_lib.last_error(L)}")
    return ierr, apot.reshape(b_slab.shape), bflat.reshape(b_slab.shape)
