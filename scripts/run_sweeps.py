"""Run a few level-1 smoother sweeps (for rocprofv3 counter passes). usage: run_sweeps.py n nsweeps [op]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
n = int(sys.argv[1]); ns = int(sys.argv[2]); op = int(sys.argv[3]) if len(sys.argv) > 3 else _lib.OP_RELAX
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
S = _lib.MGSolver([n, n, n], [np.linspace(0, 1, n)] * 3, "NDDNDD")
rng = np.random.default_rng(1)
S.upload(1, _lib.BUF_U, rng.uniform(-1, 1, (n, n, n))); S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
for _ in range(ns):
    S.op(op, 1, 1)
S.sync(); S.close()
