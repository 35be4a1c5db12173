"""wall time of the reference-compatible Python front end (ndsm_amd.vector_potential) at n^3 (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from golden_inputs import analytic_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x, y, z, A1, b1 = analytic_case(n)
del A1
for rep in range(3):
    t = time.perf_counter(); ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1); dt = time.perf_counter() - t
    print(f"{n}^3 ndsm_amd.vector_potential call {rep}: ierr {ierr} wall {dt:.3f} s", flush=True)
    del A, B
