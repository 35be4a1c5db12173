"""solve-loop cycle time with a right-hand side in HBM (Poisson problem, the general kernels) vs without"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mesh = [np.linspace(0, 1, n)] * 3
rng = np.random.default_rng(3)
u0 = rng.uniform(-1, 1, (n, n, n)); rhs = rng.uniform(-1, 1, (n, n, n)) * 100
for lap in (False, True, False, True):
    for prec in (0, 1):
        S = _lib.MGSolver([n, n, n], mesh, "NDDNDD")
        if lap:
            S.zero_rhs()
        else:
            S.upload(1, _lib.BUF_RHS, rhs)
        S.set_precision(prec)
        S.upload(1, _lib.BUF_U, u0); S.solve(vc_tol=0.0, nmax=2); S.sync()
        t = time.perf_counter(); S.solve(vc_tol=0.0, nmax=10); S.sync(); dt = (time.perf_counter() - t) / 10
        print(f"{n}^3 {'Laplace' if lap else 'Poisson'} precision {prec}: {dt*1e3:.2f} ms per solve-loop cycle", flush=True)
        S.close()
