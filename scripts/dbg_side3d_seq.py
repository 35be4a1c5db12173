"""sequence of ndsm_vector_solve calls with changing options: graphs on vs NDSM_HIP_NO_GRAPHS=1 (debug aid)"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ndsm_amd
from golden_inputs import analytic_case
shape = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [200, 200, 184]
x, y, z, _A1, b1 = analytic_case(shape)
seq = [dict(), dict(ms=3, mean=True), dict(), dict(), dict(ms=3, mean=True), dict(ms=4), dict(), dict(ms=2), dict()]
def run(tag):
    out = []
    for kw in seq:
        ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1, **kw)
        out.append(hashlib.sha256(A.tobytes() + B.tobytes()).hexdigest()[:10])
    print(tag, out, flush=True)
    return out
os.environ["NDSM_HIP_NO_GRAPHS"] = "1"
ref = run("no graphs")
os.environ.pop("NDSM_HIP_NO_GRAPHS")
got = run("graphs   ")
print("MISMATCH at", [i for i, (a, b) in enumerate(zip(ref, got)) if a != b])
