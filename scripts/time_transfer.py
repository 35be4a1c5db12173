"""Time restriction / prolongation / residual of level 1 (dev aid): usage time_transfer.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mesh = [np.linspace(0, 1, n)] * 3
S = _lib.MGSolver([n, n, n], mesh, "NDDNDD")
rng = np.random.default_rng(1)
S.upload(1, _lib.BUF_U, rng.uniform(-1, 1, (n, n, n))); S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
S.op(_lib.OP_RESIDUAL, 1); S.sync()
for name, op in (("residual", _lib.OP_RESIDUAL), ("restrict", _lib.OP_RESTRICT), ("prolong", _lib.OP_PROLONG)):
    S.op(op, 1); S.sync()
    t = min(S.timed(lambda: [S.op(op, 1) for _ in range(5)]) / 5 for _ in range(3))
    print(f"{n}^3 {name}: {t*1e3:.0f} us", flush=True)
S.close()
