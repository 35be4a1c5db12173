"""random large-ish shapes (odd and even nx): streamed restriction vs the gather kernel, bit for bit (dev aid)"""
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
        ns = [int(rng.integers(130, 330)), int(rng.integers(130, 260)), int(rng.integers(130, 230))]
        if ns[0] * ns[1] * ns[2] >= 6.4e6:
            break
    bcs = "".join(rng.choice(["D", "N"]) for _ in range(6))
    if bcs == "NNNNNN":
        bcs = "DNNNNN"
    mesh = uniform_mesh(ns); shp = tuple(ns[::-1])
    r = rand_field(shp, 300 + c)
    out = []
    for nostream in (True, False):
        if nostream:
            os.environ["NDSM_HIP_NO_STREAM_RESTRICT"] = "1"
        else:
            os.environ.pop("NDSM_HIP_NO_STREAM_RESTRICT", None)
        S = _lib.MGSolver(ns, mesh, bcs)
        S.upload(1, _lib.BUF_R, r)
        S.upload(2, _lib.BUF_U, np.full(S._npshape(2), 3.0))
        S.op(_lib.OP_RESTRICT, 1)
        out.append((S.download(2, _lib.BUF_RHS), S.download(2, _lib.BUF_U))); S.close()
    os.environ.pop("NDSM_HIP_NO_STREAM_RESTRICT", None)
    ok = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and not out[1][1].any()
    if not ok:
        bad += 1
        print("MISMATCH", ns, bcs, int((out[0][0] != out[1][0]).sum()))
print(f"{ncase} cases, {bad} mismatches")
