"""random shapes with a random cap on the number of grid levels: GPU solve vs the oracle port (small
shapes) and tracked vs untracked solve (large shapes), bit for bit (dev aid)"""
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
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for c in range(ncase):
    big = c % 4 == 3
    if big:
        ns = [int(rng.integers(120, 200)), int(rng.integers(120, 180)), int(rng.integers(120, 170))]
    else:
        ns = [int(rng.integers(8, 80)) for _ in range(3)]
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "NNNNND"
    ms = int(rng.integers(1, 6))
    full = port.ngrids(np.asarray(ns, dtype=np.int64)) if hasattr(port, "ngrids") else 0
    ng = int(rng.integers(2, max(3, full + 1)))
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 100 + c), rand_field(shp, 200 + c)
    lap = bool(rng.integers(0, 2))
    def gpu(notrack):
        if notrack:
            os.environ["NDSM_HIP_NO_TRACK"] = "1"
        else:
            os.environ.pop("NDSM_HIP_NO_TRACK", None)
        S = _lib.MGSolver(ns, mesh, bcs, ms=ms, ngrids=ng, nmax_exact=200)
        S.upload(1, _lib.BUF_U, u)
        S.zero_rhs() if lap else S.upload(1, _lib.BUF_RHS, rhs)
        r = S.solve(vc_tol=1e-10, nmax=3, hist_len=8)
        out = (r[0], r[2], list(r[3]), S.download(1, _lib.BUF_U)); S.close()
        os.environ.pop("NDSM_HIP_NO_TRACK", None)
        return out
    a = gpu(False)
    if big:
        b = gpu(True)
        ok = a[:3] == b[:3] and np.array_equal(a[3], b[3])
    else:
        r2 = np.zeros(shp) if lap else rhs
        ie2, u2, du2, h2, nc2, sw = port.solve_bvp(u.copy(), r2, mesh, bcs, ms=ms, nmax=3, ngrids=ng, hist_len=8, nmax_exact=200)
        ok = np.array_equal(a[3], u2) and a[2] == list(h2[:len(a[2])]) and a[1] == nc2
    if not ok:
        bad += 1
        print("MISMATCH", ns, bcs, ms, ng, lap, big)
print(f"{ncase} cases, {bad} mismatches")
