"""three sweeps per pass (experiment): time and check against the two-sweep + one-sweep passes (dev aid)"""
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
u0 = rng.uniform(-1, 1, (n, n, n)); rhs = rng.uniform(-1, 1, (n, n, n))
for lap in (True, False):
    if lap: S.zero_rhs()
    else: S.upload(1, _lib.BUF_RHS, rhs)
    ref = None
    for cfg in (0, 6, 7, 8):
        L.ndsm_hip_debug_fused_cfg(cfg, 0, 0, 0, -1)
        for nsw in (3, 6):
            S.upload(1, _lib.BUF_U, u0); S.op(_lib.OP_RELAX_FUSED, 1, nsw); S.sync()
            got = S.download(1, _lib.BUF_U)
            if cfg == 0: ref = {**(ref or {}), nsw: got}
            same = np.array_equal(got, ref[nsw])
            t = min(S.timed(lambda: [S.op(_lib.OP_RELAX_FUSED, 1, nsw) for _ in range(5)]) / 5 for _ in range(3))
            print(f"{'laplace' if lap else 'general'} cfg {cfg} {nsw} sweeps: {t*1e3:.0f} us  same bits: {same}", flush=True)
L.ndsm_hip_debug_fused_cfg(0, 0, 0, 0, -1)
S.close()
