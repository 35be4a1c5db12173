"""loop-back world vs single domain after n V-cycles (debug aid): dbg_world.py nx ny nz nranks levels [ncyc]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
nx, ny, nz, nr, lv = (int(v) for v in sys.argv[1:6])
nc = int(sys.argv[6]) if len(sys.argv) > 6 else 1
os.environ["NDSM_HIP_DIST_LEVELS"] = str(lv)
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
ns = [nx, ny, nz]; mesh = uniform_mesh(ns); shp = (nz, ny, nx)
u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
for bcs in ("NDDNDD", "DDNDDN"):
    S = _lib.MGSolver(ns, mesh, bcs); W = _lib.World(ns, mesh, bcs, nr)
    S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs); W.upload(_lib.BUF_U, u); W.upload(_lib.BUF_RHS, rhs)
    S.vcycle(nc); W.vcycle(nc)
    a, b = S.download(1, _lib.BUF_U), W.download(_lib.BUF_U)
    d = np.argwhere(a != b)
    print(ns, nr, "levels", W.dist_levels, bcs, "ndiff", len(d), "max", np.abs(a - b).max(), "planes", np.unique(d[:, 0])[:12] if len(d) else "")
    S.close(); W.close()
