"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/ndsm_hip.h declares, reproduces the reference's option-slot
getters, and fails loudly (no CPU fallback) when no GPU is visible."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ndsm_hip.h")


@pytest.fixture(scope="module")
def lib():
    import ndsm_amd
    if not os.path.exists(ndsm_amd.lib_path()):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ndsm_amd"), "-j", "8"])
    return ndsm_amd.load_library()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ndsm_[a-z0-9_]+|get_[a-z0-9_]+)\s*\(", src)))


def test_header_is_valid_c():
    """the boundary is a C header: it must compile as C99 (and as C++) on its own"""
    for lang, std in (("c", "-std=c99"), ("c++", "-std=c++11")):
        subprocess.check_call(["gcc", "-fsyntax-only", "-x", lang, std, "-Wall", "-Werror", HEADER])


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert len(names) >= 13 + 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_nothing_else_is_exported():
    """the shared object's dynamic symbol table is exactly the header (linker version script): no
    Fortran runtime, no internal kernel layer - it can share a process with any other flang library"""
    import ndsm_amd
    out = subprocess.check_output(["nm", "-D", "--defined-only", ndsm_amd.lib_path()], text=True)
    live = sorted(l.split()[-1] for l in out.splitlines() if re.search(r" [TDB] ", l))
    assert live == declared_functions()


def test_reference_symbol_set_is_covered(lib):
    """every C symbol of the reference's ndsmf.so (fortran/ndsm_python_wrapper.f90) exists here"""
    ref_symbols = ["ndsm_vector_solve", "get_iopt_len", "get_iopt_ierr", "get_iopt_ms", "get_iopt_ncycles",
                   "get_iopt_debug", "get_iopt_dumax", "get_iopt_iopt_nmaxex", "get_iopt_true", "get_iopt_false",
                   "get_ropt_tim", "get_ropt_vtol", "get_ropt_ctol"]
    ref_so = os.path.join(ROOT, "oracle", "_ref", "ndsmf.so")
    if os.path.exists(ref_so):
        out = subprocess.check_output(["nm", "-D", "--defined-only", ref_so], text=True)
        live = sorted(l.split()[-1] for l in out.splitlines()
                      if re.search(r" T (ndsm_|get_)", l))
        assert live == sorted(ref_symbols)
    for n in ref_symbols:
        assert hasattr(lib, n)


def test_getters_match_reference_values(lib):
    # ndsm_vector_potential.f90:40-57 and the get_iopt_ierr quirk (ndsm_python_wrapper.f90:170-174)
    want = dict(get_iopt_len=16, get_iopt_ierr=16, get_iopt_ms=0, get_iopt_ncycles=1, get_iopt_debug=5,
                get_iopt_dumax=6, get_iopt_iopt_nmaxex=7, get_iopt_true=1, get_iopt_false=0, get_ropt_tim=2,
                get_ropt_vtol=0, get_ropt_ctol=1, get_iopt_fail3d=8, get_iopt_ngrids=9, get_iopt_ncyc_out=10,
                get_iopt_prec=11, get_ropt_dulast=3)
    for name, val in want.items():
        assert getattr(lib, name)() == val, name
    ref_so = os.path.join(ROOT, "oracle", "_ref", "ndsmf.so")
    if os.path.exists(ref_so):
        ref = ctypes.CDLL(ref_so)
        for name in want:
            if hasattr(ref, name):
                assert getattr(ref, name)() == getattr(lib, name)(), name


def test_alias_for_unmodified_ndsm_py():
    import ndsm_amd
    alias = os.path.join(os.path.dirname(ndsm_amd.lib_path()), "ndsmf.so")
    assert os.path.exists(alias)          # the name the reference's ndsm.py searches for (ndsm.py:66)
    L = ctypes.CDLL(alias)
    assert L.get_iopt_len() == 16


def test_no_cpu_fallback(lib):
    """without a GPU the reference entry point must FAIL, not compute"""
    if lib.ndsm_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    import ndsm_amd
    n = 8
    x = np.linspace(0, 1, n)
    b = np.ones((3, n, n, n))
    ierr, A, B = ndsm_amd.vector_potential(x, x.copy(), x.copy(), b)
    assert ierr == 9001
    assert not A.any()
    with pytest.raises(ndsm_amd.NdsmHipError):
        ndsm_amd.MGSolver([n, n, n], [x, x, x], "NDDNDD")
    with pytest.raises(ndsm_amd.NdsmHipError):
        ndsm_amd.poisson_solve(np.zeros((n, n, n)), None, [x, x, x], "NDDNDD")


def test_product_does_not_touch_the_oracle():
    """nothing under ndsm_amd/ may import, link or open the checker"""
    bad = []
    for base, _dirs, files in os.walk(os.path.join(ROOT, "ndsm_amd")):
        if os.sep + "build" in base or os.sep + "lib" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".f90")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"liboracle|ndsm_oracle|oracle/|from oracle|import oracle|_ref/", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    import ndsm_amd
    out = subprocess.check_output(["ldd", ndsm_amd.lib_path()], text=True)
    assert "oracle" not in out


def test_python_loader_signature_matches_reference():
    """ndsm_amd.vector_potential keeps the reference's parameters and defaults (ndsm.py:66)"""
    import inspect
    import ndsm_amd
    sig = inspect.signature(ndsm_amd.vector_potential)
    want = [("x", inspect._empty), ("y", inspect._empty), ("z", inspect._empty), ("b", inspect._empty),
            ("niterex_max", 10000), ("ncycles_max", 1024), ("ex_tol", 1e-13), ("vc_tol", 1e-10), ("ms", 5),
            ("mean", False)]
    got = [(p.name, p.default) for p in sig.parameters.values()]
    assert got[:len(want)] == want
    assert {"libname", "libpath", "debug"} <= set(sig.parameters)


def test_additive_entry_points_fail_cleanly_without_a_gpu(lib):
    """round-2 additions on a box without a GPU: error codes, never a crash, never a computed result"""
    if lib.ndsm_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    n = 8
    x = np.linspace(0, 1, n)
    ns4 = np.array([n, n, n, 3], dtype=np.intc)
    h = ctypes.c_void_p()
    lib.ndsm_hip_vecpot_create.argtypes = [ip, dp, dp, dp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    rc = lib.ndsm_hip_vecpot_create(ns4.ctypes.data_as(ip), x.ctypes.data_as(dp), x.ctypes.data_as(dp),
                                    x.ctypes.data_as(dp), 0, ctypes.byref(h))
    assert rc == 9001 and not h.value
    assert lib.ndsm_hip_vecpot_destroy(None) == 0
    r, m = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib.ndsm_hip_dist_info(ctypes.byref(r), ctypes.byref(m)) == 0 and (r.value, m.value) == (0, 0)
    assert lib.ndsm_hip_dist_selftest(16) != 0                  # no runtime, no communicator
    assert lib.ndsm_hip_shutdown() == 0 and lib.ndsm_hip_shutdown() == 0
    assert lib.ndsm_hip_debug_fused_cfg(0, 0, 0, 0, -1) == 0
    assert lib.ndsm_hip_debug_tail(1) == 0
    p = ctypes.c_void_p()
    lib.ndsm_hip_device_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    assert lib.ndsm_hip_device_alloc(1024, ctypes.byref(p)) == 9001 and not p.value
    import ndsm_amd
    with pytest.raises(ndsm_amd.NdsmHipError):
        ndsm_amd.VecPot(x, x, x)

