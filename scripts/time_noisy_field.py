"""ndsm_vector_solve on a field whose three components all need the full number of V-cycles (analytic field +
noise on the boundary): wall time with B_z formed and sent home behind the A_z solve, and without (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from golden_inputs import analytic_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x, y, z, A1, b1 = analytic_case(n)
rng = np.random.default_rng(3)
for c in range(3):
    b1[c] += 0.2 * rng.standard_normal(b1[c].shape)
out = {}
for mode in ("1", "0", "1", "0"):
    if mode == "0":
        os.environ["NDSM_HIP_NO_EARLY_BZ"] = "1"
    else:
        os.environ.pop("NDSM_HIP_NO_EARLY_BZ", None)
    t = time.perf_counter(); ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1); dt = time.perf_counter() - t
    print(f"{n}^3 early B_z={mode}: ierr {ierr} wall {dt*1e3:.1f} ms", flush=True)
    if mode in out:
        assert np.array_equal(out[mode][0], A) and np.array_equal(out[mode][1], B)
    out[mode] = (A, B)
assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
print("bit-identical")
