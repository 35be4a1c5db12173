"""random shapes / slab counts / distributed-level depths: loop-back world vs single-domain solver, bit for bit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = done = 0
while done < ncase:
    nr = int(rng.integers(2, 9))
    ns = [int(rng.integers(16, 160)), int(rng.integers(16, 120)), int(rng.integers(16, 60)) * nr]
    lv = int(rng.integers(1, 4))
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "DNNNNN"
    ms = int(rng.integers(1, 6))
    os.environ["NDSM_HIP_DIST_LEVELS"] = str(lv)
    os.environ["NDSM_HIP_OVERLAP"] = str(int(rng.integers(0, 2)))
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 100 + done), rand_field(shp, 200 + done)
    try:
        W = _lib.World(ns, mesh, bcs, nr, ms=ms)
    except _lib.NdsmHipError:
        continue          # the shape cannot be cut that way
    S = _lib.MGSolver(ns, mesh, bcs, ms=ms)
    S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs); W.upload(_lib.BUF_U, u); W.upload(_lib.BUF_RHS, rhs)
    r1 = S.solve(vc_tol=1e-10, nmax=3, hist_len=8); r2 = W.solve(vc_tol=1e-10, nmax=3, hist_len=8)
    a, b = S.download(1, _lib.BUF_U), W.download(_lib.BUF_U)
    ok = np.array_equal(a, b) and list(r1[3]) == list(r2[3]) and r1[2] == r2[2]
    if not ok:
        bad += 1
        print("MISMATCH", ns, nr, "levels", lv, W.dist_levels, bcs, ms, os.environ["NDSM_HIP_OVERLAP"], int((a != b).sum()))
    S.close(); W.close(); done += 1
print(f"{ncase} cases, {bad} mismatches")
