"""random small shapes / BC sets: GPU V-cycles and solves vs the oracle port, bit for bit (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from oracle import Oracle
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
port = Oracle("port")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for c in range(ncase):
    ns = [int(rng.integers(8, 72)) for _ in range(3)]
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "NNNNND"
    ms = int(rng.integers(1, 6))
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 100 + c), rand_field(shp, 200 + c)
    ierr2, u2, du2, h2, nc2, sw = port.solve_bvp(u.copy(), rhs, mesh, bcs, ms=ms, nmax=4, hist_len=8)
    S = _lib.MGSolver(ns, mesh, bcs, ms=ms)
    S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs)
    ierr, du, nc, h = S.solve(vc_tol=1e-10, nmax=4, hist_len=8)
    got = S.download(1, _lib.BUF_U); S.close()
    ok = np.array_equal(got, u2) and list(h) == list(h2[:len(h)]) and nc == nc2
    if not ok:
        bad += 1
        print("MISMATCH", ns, bcs, ms, "ndiff", int((got != u2).sum()), "hist", list(h), list(h2[:nc2]))
print(f"{ncase} cases, {bad} mismatches")
