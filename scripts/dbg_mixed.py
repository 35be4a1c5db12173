"""mixed-precision solve vs fp64 solve (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import manufactured_poisson, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ns = [n, n, n]; mesh = uniform_mesh(ns); bcs = "NDDNDD"
us, rhs = manufactured_poisson(mesh, bcs)
u0 = np.zeros((n, n, n)); u0[:, 0, :], u0[:, -1, :], u0[0], u0[-1] = us[:, 0, :], us[:, -1, :], us[0], us[-1]
S = _lib.MGSolver(ns, mesh, bcs)
S.upload(1, _lib.BUF_RHS, rhs); S.upload(1, _lib.BUF_U, u0)
ie, du, nc, h64 = S.solve(vc_tol=1e-11, nmax=3, hist_len=64); u64 = S.download(1, _lib.BUF_U)
print("fp64 ", ie, nc, h64)
print("mixed applies:", S.set_precision(2))
S.upload(1, _lib.BUF_U, u0)
ie, du, nc, h32 = S.solve(vc_tol=1e-11, nmax=3, hist_len=64); u32 = S.download(1, _lib.BUF_U)
print("mixed", ie, nc, h32)
d = np.abs(u32 - u64); print("max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "max|u|", np.abs(u64).max())
S.close()
