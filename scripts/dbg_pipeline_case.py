"""re-run one case of fuzz_pipeline.py (seed, index) and take the difference to the oracle apart (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ndsm_amd
from oracle import Oracle
from golden_inputs import analytic_case
port = Oracle("port")
seed, want = int(sys.argv[1]), [int(v) for v in sys.argv[2].split(",")]
rng = np.random.default_rng(seed)
for c in range(1000):
    ns = [int(rng.integers(5, 60)) for _ in range(3)]
    x, y, z, A1, b = analytic_case(ns)
    if rng.integers(0, 2):
        b = b + 0.3 * rng.uniform(-1, 1, b.shape)
    kw = dict(ms=int(rng.integers(1, 6)), mean=bool(rng.integers(0, 2)), ncycles_max=int(rng.choice([1, 3, 1024])),
              vc_tol=float(rng.choice([1e-10, 1e-7])))
    if ns == want:
        break
print(ns, kw)
res = {}
for tag, env in (("device faces", {}), ("host faces", {"NDSM_HIP_HOST_FACES": "1"})):
    os.environ.update(env)
    res[tag] = ndsm_amd.vector_potential(x, y, z, b.copy(), **kw)
    for k in env: os.environ.pop(k)
ie2, A2, B2, _, _ = port.vector_potential(x, y, z, b, **kw)
ref = None
try:
    r = Oracle("ref"); ie3, A3, B3, _, _ = r.vector_potential(x, y, z, b, **kw); ref = (ie3, A3, B3)
except Exception as e:
    print("no reference:", e)
sc = np.abs(A2).max()
print("max|A| oracle", sc, "max|B|", np.abs(B2).max())
for tag, (ie, A, B) in res.items():
    print(tag, "ierr", ie, "dA/sc", np.abs(A - A2).max() / sc, "per component", [np.abs(A[c] - A2[c]).max() for c in range(3)])
print("device == host faces:", np.array_equal(res["device faces"][1], res["host faces"][1]))
if ref:
    print("reference vs port: dA/sc", np.abs(ref[1] - A2).max() / sc, " ours vs reference:", np.abs(res["device faces"][1] - ref[1]).max() / sc)
