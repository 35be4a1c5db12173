"""loop-back world in mixed precision vs the single-domain mixed solver (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for ns, nr, lev in (([128, 64, 160], 2, 0), ([64, 64, 256], 4, 2), ([128, 96, 192], 3, 0), ([128, 64, 256], 2, 3)):
    if lev:
        os.environ["NDSM_HIP_DIST_LEVELS"] = str(lev)
    else:
        os.environ.pop("NDSM_HIP_DIST_LEVELS", None)
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
    for bcs, lap in (("NDDNDD", True), ("DDNDDN", False)):
        S = _lib.MGSolver(ns, mesh, bcs); assert S.set_precision(2)
        W = _lib.World(ns, mesh, bcs, nr); on = W.set_precision(2)
        for X in (S,):
            X.upload(1, _lib.BUF_U, u)
            X.zero_rhs() if lap else X.upload(1, _lib.BUF_RHS, rhs)
        W.upload(_lib.BUF_U, u)
        W.zero_rhs() if lap else W.upload(_lib.BUF_RHS, rhs)
        a = S.solve(vc_tol=1e-9, nmax=6, hist_len=8); b = W.solve(vc_tol=1e-9, nmax=6, hist_len=8)
        ua, ub = S.download(1, _lib.BUF_U), W.download(_lib.BUF_U)
        print(ns, nr, "levels", W.dist_levels, bcs, "mixed on:", on, "hist equal:", list(a[3]) == list(b[3]), "bits equal:", np.array_equal(ua, ub),
              "max diff", float(np.abs(ua - ub).max()), flush=True)
        if list(a[3]) != list(b[3]):
            print("   ", list(a[3]), list(b[3]))
        S.close(); W.close()
