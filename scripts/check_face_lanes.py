"""the six 2-D face solves side by side (lanes) against one after the other: same bits, wall time (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from golden_inputs import analytic_case
for n in [int(a) for a in sys.argv[1:]] or (22, 64, 128, 256):
    x, y, z, A1, b1 = analytic_case(n)
    out = {}
    for mode in ("0", "1", "0", "1", "1"):
        os.environ["NDSM_HIP_FACE_LANES"] = mode
        ndsm_amd.vector_potential(x, y, z, b1)
        t = time.perf_counter(); ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1); dt = time.perf_counter() - t
        if mode in out:
            assert np.array_equal(out[mode][0], A) and np.array_equal(out[mode][1], B)
        out[mode] = (A, B)
        print(f"{n}^3 lanes={mode}: ierr {ierr} wall {dt*1e3:.1f} ms", flush=True)
    same = np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
    print(f"{n}^3: lanes vs sequential bit-identical: {same}", flush=True)
    assert same
