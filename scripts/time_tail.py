"""Per-level timing of one V-cycle's pieces at n^3 (dev aid): usage time_tail.py [n]"""
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
S.upload(1, _lib.BUF_U, rng.uniform(-1, 1, (n, n, n))); S.zero_rhs()
S.vcycle(2); S.sync()
print("vcycle", min(S.timed(lambda: S.vcycle(5)) / 5 for _ in range(3)) * 1e3, "us", flush=True)
for lvl in range(2, S.ngrids):
    for name, op, cnt in (("relax5", _lib.OP_RELAX, 5), ("relax5+res", _lib.OP_RELAX_RES, 5), ("residual", _lib.OP_RESIDUAL, 1),
                          ("restrict", _lib.OP_RESTRICT, 1), ("prolong", _lib.OP_PROLONG, 1)):
        S.op(op, lvl, cnt); S.sync()
        t = min(S.timed(lambda: [S.op(op, lvl, cnt) for _ in range(10)]) / 10 for _ in range(3))
        print(f"level {lvl} {S.shapes[lvl-1]} {name}: {t*1e3:.1f} us", flush=True)
S.close()
