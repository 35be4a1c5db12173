"""cost of building and tearing down a solver hierarchy (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for n in (128, 256, 512):
    mesh = [np.linspace(0, 1, n)] * 3
    for rep in range(3):
        t0 = time.perf_counter(); S = _lib.MGSolver([n, n, n], mesh, "NDDNDD"); S.sync(); t1 = time.perf_counter()
        S.close(); L.ndsm_hip_sync(); t2 = time.perf_counter()
    print(f"{n}^3: create {1e3*(t1-t0):.1f} ms, destroy {1e3*(t2-t1):.1f} ms", flush=True)
    m2 = [np.linspace(0, 1, n)] * 2
    t0 = time.perf_counter(); S = _lib.MGSolver([n, n], m2, "NNNN"); S.sync(); t1 = time.perf_counter(); S.close(); t2 = time.perf_counter()
    print(f"{n}^2: create {1e3*(t1-t0):.1f} ms, destroy {1e3*(t2-t1):.1f} ms", flush=True)
