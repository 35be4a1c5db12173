"""anisotropic spacings (dx != dy != dz, arbitrary origins): GPU solve vs the oracle port on small shapes,
slab world vs single domain on larger ones, bit for bit (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from oracle import Oracle
from golden_inputs import rand_field
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
port = Oracle("port")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for c in range(ncase):
    big = c % 3 == 2
    ns = [int(rng.integers(64, 150)), int(rng.integers(32, 100)), int(rng.integers(96, 200))] if big else [int(rng.integers(8, 70)) for _ in range(3)]
    mesh = [rng.uniform(-3, 3) + np.linspace(0.0, rng.uniform(0.2, 5.0), n) for n in ns]
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "NNNDNN"
    ms = int(rng.integers(1, 6)); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 100 + c), rand_field(shp, 200 + c)
    S = _lib.MGSolver(ns, mesh, bcs, ms=ms); S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs)
    a = S.solve(vc_tol=1e-10, nmax=3, hist_len=8); ua = S.download(1, _lib.BUF_U); S.close()
    if big:
        nr = int(rng.integers(2, 5))
        try:
            W = _lib.World(ns, mesh, bcs, nr, ms=ms)
        except _lib.NdsmHipError:
            continue
        W.upload(_lib.BUF_U, u); W.upload(_lib.BUF_RHS, rhs)
        b = W.solve(vc_tol=1e-10, nmax=3, hist_len=8); ub = W.download(_lib.BUF_U); W.close()
        ok = list(a[3]) == list(b[3]) and np.array_equal(ua, ub)
    else:
        ie2, u2, du2, h2, nc2, sw = port.solve_bvp(u.copy(), rhs, mesh, bcs, ms=ms, nmax=3, hist_len=8)
        ok = np.array_equal(ua, u2) and list(a[3]) == list(h2[:len(a[3])])
    if not ok:
        bad += 1
        print("MISMATCH", ns, bcs, ms, big)
print(f"{ncase} cases, {bad} mismatches")
