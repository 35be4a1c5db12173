"""host enqueue time vs device time of a V-cycle (dev aid): is a small solve CPU-launch bound?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for n in [int(a) for a in sys.argv[1:]] or (64, 128, 220, 512):
    mesh = [np.linspace(0, 1, n)] * 3
    S = _lib.MGSolver([n, n, n], mesh, "NDDNDD"); S.zero_rhs()
    S.upload(1, _lib.BUF_U, np.random.default_rng(1).uniform(-1, 1, (n, n, n)))
    S.vcycle(3); S.sync()
    t0 = time.perf_counter(); S.vcycle(20); t1 = time.perf_counter(); S.sync(); t2 = time.perf_counter()
    print(f"{n}^3: enqueue {1e3*(t1-t0)/20:.3f} ms per cycle, until drained {1e3*(t2-t0)/20:.3f} ms per cycle", flush=True)
    S.close()
