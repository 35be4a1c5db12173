"""random large-ish shapes: solve loop with the metric / prolongation folded into the smoother launches
vs the separate passes (NDSM_HIP_NO_TRACK), and vs mixed-precision convergence (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for c in range(ncase):
    while True:
        ns = [int(rng.integers(100, 280)), int(rng.integers(100, 230)), int(rng.integers(100, 210))]
        if ns[0] * ns[1] * ns[2] >= 2.2e6:
            break
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "DNNNNN"
    ms = int(rng.integers(1, 6)); lap = bool(rng.integers(0, 2))
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    u, rhs = rand_field(shp, 100 + c), rand_field(shp, 200 + c) * 20
    out = []
    for notrack in (True, False):
        if notrack:
            os.environ["NDSM_HIP_NO_TRACK"] = "1"
        else:
            os.environ.pop("NDSM_HIP_NO_TRACK", None)
        S = _lib.MGSolver(ns, mesh, bcs, ms=ms)
        if lap:
            S.zero_rhs()
        else:
            S.upload(1, _lib.BUF_RHS, rhs)
        S.upload(1, _lib.BUF_U, u)
        r = S.solve(vc_tol=1e-9, nmax=4, hist_len=8)
        out.append((r[0], r[2], list(r[3]), S.download(1, _lib.BUF_U))); S.close()
    os.environ.pop("NDSM_HIP_NO_TRACK", None)
    a, b = out
    ok = a[:3] == b[:3] and np.array_equal(a[3], b[3])
    if not ok:
        bad += 1
        print("MISMATCH", ns, bcs, ms, lap, a[:3], b[:3], int((a[3] != b[3]).sum()))
print(f"{ncase} cases, {bad} mismatches")
