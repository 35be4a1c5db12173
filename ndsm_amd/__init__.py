"""ndsm_amd - MI355X-native multigrid V-cycle behind the NDSM C ABI.

The product is the shared library ndsm_amd/lib/libndsm_hip.so (HIP kernels for
gfx950 + a Fortran 2003 host driver).  This package only holds the Python
loader that mirrors the reference's ndsm.py, and thin ctypes wrappers for the
additive C entry points.  There is no CPU fallback: without the library or
without an MI355X every call raises.
"""
from .ndsm import vector_potential, vector_potential_slab, get_lib_path  # noqa: F401
from ._lib import (load_library, lib_path, MGSolver, VecPot, World, slab_plan, poisson_solve,  # noqa: F401
                   NdsmHipError)

__all__ = ["vector_potential", "vector_potential_slab", "get_lib_path", "load_library", "lib_path", "MGSolver", "VecPot", "poisson_solve",
           "NdsmHipError"]
