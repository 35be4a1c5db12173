"""colour passes vs forced fused launches on mid-size levels (dev aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for n in [int(a) for a in sys.argv[1:]] or (128, 160, 192):
    mesh = [np.linspace(0, 1, n)] * 3
    S = _lib.MGSolver([n, n, n], mesh, "NDDNDD")
    rng = np.random.default_rng(1)
    S.upload(1, _lib.BUF_U, rng.uniform(-1, 1, (n, n, n))); S.upload(1, _lib.BUF_RHS, rng.uniform(-1, 1, (n, n, n)))
    for name, op in (("colour", _lib.OP_RELAX_COLOR), ("fused", _lib.OP_RELAX_FUSED)):
        S.op(op, 1, 5); S.sync()
        t = min(S.timed(lambda: [S.op(op, 1, 5) for _ in range(10)]) / 10 for _ in range(3))
        print(f"{n}^3 5 sweeps {name}: {t*1e3:.1f} us", flush=True)
    S.close()
