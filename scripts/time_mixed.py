"""fp64 vs mixed-precision solve timing at n^3 (dev aid): usage time_mixed.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sys.path.insert(0, ROOT)
from bench import boundary_problem
mesh, u0 = boundary_problem(n)
S = _lib.MGSolver([n, n, n], mesh, "NDDNDD")
S.zero_rhs()
for mode in (0, 1, 0, 1):
    S.set_precision(mode)
    S.upload(1, _lib.BUF_U, u0); S.sync()
    t = time.perf_counter(); ie, du, nc, h = S.solve(vc_tol=1e-10, nmax=64, hist_len=64); S.sync(); dt = time.perf_counter() - t
    print(f"{n}^3 precision {mode}: ierr {ie} cycles {nc} du {du:.3e}  {dt*1e3:.1f} ms total, {dt*1e3/nc:.2f} ms per V-cycle", flush=True)
S.close()
