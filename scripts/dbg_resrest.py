"""fused residual+restrict vs the two separate kernels (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for ns in ([256, 256, 256], [200, 180, 190], [512, 128, 128], [130, 260, 200]):
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
    for bcs in ("NDDNDD", "DNDDND", "DDNDDN", "NNNNND"):
        S = _lib.MGSolver(ns, mesh, bcs)
        S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs)
        S.op(_lib.OP_RESIDUAL, 1); S.op(_lib.OP_RESTRICT, 1); a = S.download(2, _lib.BUF_RHS)
        if os.environ.get("ORACLE"):
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from oracle import Oracle
            port = Oracle("port")
            want = port.restrict(port.residual3d(u, rhs, mesh, bcs), ns, mesh, 1)
            print("   vs oracle:", np.array_equal(a, want))
        S.upload(2, _lib.BUF_U, np.ones(S._npshape(2)))
        try:
            S.op(_lib.OP_RESREST, 1)
        except _lib.NdsmHipError as e:
            print(ns, bcs, "not covered:", str(e)[:60]); S.close(); continue
        b = S.download(2, _lib.BUF_RHS); z = S.download(2, _lib.BUF_U)
        t1 = S.timed(lambda: [S.op(_lib.OP_RESREST, 1) for _ in range(5)]) / 5
        t2 = S.timed(lambda: [(S.op(_lib.OP_RESIDUAL, 1), S.op(_lib.OP_RESTRICT, 1)) for _ in range(5)]) / 5
        S.close()
        d = np.argwhere(a != b)
        print(ns, bcs, "ndiff", len(d), "of", a.size, "u_c zero:", not z.any(), "fused %.1f us, separate %.1f us" % (t1 * 1e3, t2 * 1e3))
        if len(d):
            print("   K:", np.unique(d[:, 0])[:20], "J:", np.unique(d[:, 1])[:20], "I:", np.unique(d[:, 2])[:20], "max", np.abs(a - b).max())
