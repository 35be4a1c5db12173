"""where the first ndsm_vector_solve of a process spends its time: runtime bring-up, context (hierarchies, device
arrays), first solve, second solve (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t00 = time.perf_counter()
import numpy as np
import ndsm_amd
from golden_inputs import analytic_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x, y, z, A1, b1 = analytic_case(n)
del A1
t0 = time.perf_counter()
L = ndsm_amd.load_library(); t1 = time.perf_counter()
assert L.ndsm_hip_init(0) == 0; t2 = time.perf_counter()
V = ndsm_amd.VecPot(x, y, z); L.ndsm_hip_sync(); t3 = time.perf_counter()
r = V.solve(b1); t4 = time.perf_counter()
r = V.solve(b1); t5 = time.perf_counter()
print(f"{n}^3: load library {1e3*(t1-t0):.0f} ms, runtime init {1e3*(t2-t1):.0f} ms, context {1e3*(t3-t2):.0f} ms, "
      f"first solve {1e3*(t4-t3):.0f} ms, second solve {1e3*(t5-t4):.0f} ms", flush=True)
