"""random mid-size shapes / options: ndsm_vector_solve with its face solves side by side + replayed graphs (default)
against one after the other (NDSM_HIP_FACE_LANES=0), bit for bit, including repeated calls on a cached context
and option changes between calls (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from golden_inputs import analytic_case
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for c in range(ncase):
    ns = [int(rng.integers(20, 280)) for _ in range(3)]
    if rng.random() < 0.3:
        ns[int(rng.integers(0, 3))] = int(rng.integers(8, 24))
    x, y, z, A1, b1 = analytic_case(ns)
    b1 = b1 + 0.05 * rng.standard_normal(b1.shape)          # not the smooth analytic field only
    res = {}
    for rep in range(2):
        kw = dict(ms=int(rng.integers(1, 6)), mean=bool(rng.integers(0, 2)), ncycles_max=int(rng.choice([3, 40, 1024])),
                  vc_tol=float(rng.choice([1e-6, 1e-10])))
        for mode in ("1", "0", "1"):
            os.environ["NDSM_HIP_FACE_LANES"] = mode
            ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1, **kw)
            if (rep, "ref") in res:
                ok = ierr == res[(rep, "ref")][0] and np.array_equal(A, res[(rep, "ref")][1]) and np.array_equal(B, res[(rep, "ref")][2])
                if not ok:
                    bad += 1
                    print("BAD", ns, kw, mode, flush=True)
            else:
                res[(rep, "ref")] = (ierr, A, B)
    print("ok ", ns, flush=True)
print(f"{ncase} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
