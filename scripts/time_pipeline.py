"""Phase timing of ndsm_vector_solve (NDSM_HIP_TIMING=1): usage time_pipeline.py [n]"""
import os, sys, time
os.environ["NDSM_HIP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from golden_inputs import analytic_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x, y, z, A1, b1 = analytic_case(n)
for rep in range(2):
    t = time.time(); ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1); dt = time.time() - t
    print(f"{n}^3 vector potential: ierr {ierr} wall {dt:.3f} s", file=sys.stderr, flush=True)
