"""solve-loop cycle time at several sizes (Laplace, fp64) (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for n in [int(a) for a in sys.argv[1:]] or (128, 160, 192, 256, 384, 512):
    mesh = [np.linspace(0, 1, n)] * 3
    S = _lib.MGSolver([n, n, n], mesh, "NDDNDD"); S.zero_rhs()
    S.upload(1, _lib.BUF_U, np.random.default_rng(1).uniform(-1, 1, (n, n, n)))
    S.solve(vc_tol=0.0, nmax=3); S.sync()
    t = time.perf_counter(); S.solve(vc_tol=0.0, nmax=20); S.sync(); dt = (time.perf_counter() - t) / 20
    print(f"{n}^3: {dt*1e3:.3f} ms per solve-loop cycle", flush=True)
    S.close()
