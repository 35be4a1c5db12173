"""kernel trace of N solve-loop cycles at 512^3 for gap analysis (dev aid; run under rocprofv3 --kernel-trace)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
mesh = [np.linspace(0, 1, n)] * 3
S = _lib.MGSolver([n, n, n], mesh, "NDDNDD"); S.zero_rhs()
S.upload(1, _lib.BUF_U, np.random.default_rng(1).uniform(-1, 1, (n, n, n)))
S.solve(vc_tol=0.0, nmax=8); S.sync()
S.close()
