"""Where does the fused sweep differ from the two-pass sweep? (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
shapes = [[22, 22, 22], [40, 24, 32], [64, 64, 64], [128, 128, 128], [200, 100, 70]]
if len(sys.argv) > 1:
    shapes = [[int(v) for v in sys.argv[1].split("x")]]
for ns in shapes:
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
    for bcs in ("NDDNDD", "DNDDND", "DDNDDN", "NNNNND", "DDDDDD"):
        S = _lib.MGSolver(ns, mesh, bcs)
        S.upload(1, _lib.BUF_RHS, rhs)
        NSW = int(os.environ.get("NSW", "1"))
        S.upload(1, _lib.BUF_U, u); S.op(_lib.OP_RELAX_COLOR, 1, NSW); a = S.download(1, _lib.BUF_U)
        if os.environ.get("RES"):
            S.op(_lib.OP_RESIDUAL, 1); ra = S.download(1, _lib.BUF_R)
            S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_R, np.full(shp, np.nan))
            S.op(_lib.OP_RELAX_RES_FUSED, 1, NSW); b = S.download(1, _lib.BUF_U); rb = S.download(1, _lib.BUF_R)
            if os.environ.get("RES") == "r":
                a, b = ra, rb
                b = np.where(np.isnan(b), 1e300, b)
        else:
            S.upload(1, _lib.BUF_U, u); S.op(_lib.OP_RELAX_FUSED, 1, NSW); b = S.download(1, _lib.BUF_U)
        S.close()
        d = np.argwhere(a != b)
        print(ns, bcs, "ndiff", len(d), "of", a.size, "max", np.abs(a - b).max() if len(d) else 0.0)
        if len(d):
            k, j, i = d[:, 0], d[:, 1], d[:, 2]
            print("   k:", np.unique(k)[:40], "\n   j:", np.unique(j)[:40], "\n   i:", np.unique(i)[:40])
            par = (i + j + k) & 1
            print("   parity counts", np.bincount(par, minlength=2), "first few", d[:6].tolist())
