"""Run a few level-1 smoother sweeps (for rocprofv3 counter passes). usage: run_sweeps.py n nsweeps [zero]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
n = int(sys.argv[1]); ns = int(sys.argv[2]); zero = len(sys.argv) > 3 and sys.argv[3] == "zero"
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
S = _lib.MGSolver([n, n, n], [np.linspace(0, 1, n)] * 3, "NDDNDD")
rng = np.random.default_rng(1)
S.upload(1, _lib.BUF_U, rng.uniform(-1, 1, (n, n, n)))
if zero:
    S.zero_rhs()
else:
    S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
S.op(_lib.OP_RELAX, 1, ns)
S.sync(); S.close()
