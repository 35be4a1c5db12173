"""random shapes: forced fused launches (1..5 sweeps, +residual, zero-rhs variant) vs the two-pass kernels (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for c in range(ncase):
    ns = [int(rng.integers(16, 300)), int(rng.integers(16, 200)), int(rng.integers(8, 120))]
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "DNNNNN"
    nsw = int(rng.integers(1, 6))
    lap = bool(rng.integers(0, 2))
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 100 + c), rand_field(shp, 200 + c)
    S = _lib.MGSolver(ns, mesh, bcs)
    if lap:
        S.zero_rhs()
    else:
        S.upload(1, _lib.BUF_RHS, rhs)
    S.upload(1, _lib.BUF_U, u); S.op(_lib.OP_RELAX_COLOR, 1, nsw); S.op(_lib.OP_RESIDUAL, 1)
    a, ra = S.download(1, _lib.BUF_U), S.download(1, _lib.BUF_R)
    S.upload(1, _lib.BUF_U, u); S.op(_lib.OP_RELAX_FUSED, 1, nsw); b = S.download(1, _lib.BUF_U)
    S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_R, np.full(shp, np.nan)); S.op(_lib.OP_RELAX_RES_FUSED, 1, nsw)
    c2, rc = S.download(1, _lib.BUF_U), S.download(1, _lib.BUF_R)
    S.close()
    ok = np.array_equal(a, b) and np.array_equal(a, c2) and np.array_equal(ra, rc)
    if not ok:
        bad += 1
        print("MISMATCH", ns, bcs, nsw, lap, int((a != b).sum()), int((a != c2).sum()), int((ra != rc).sum()))
print(f"{ncase} cases, {bad} mismatches")
