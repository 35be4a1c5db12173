"""2-D all-Neumann face solve at n^2: cycles, time per cycle, per-level pieces (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = np.linspace(0, 1, n); mesh = [x, x.copy()]
S = _lib.MGSolver([n, n], mesh, "NNNN")
rng = np.random.default_rng(1)
rhs = rng.uniform(-1, 1, (n, n)); rhs -= rhs.mean()
S.upload(1, _lib.BUF_RHS, rhs); S.upload(1, _lib.BUF_U, np.zeros((n, n)))
t = time.perf_counter(); ie, du, nc, h = S.solve(vc_tol=1e-10, nmax=1024); S.sync(); dt = time.perf_counter() - t
print(f"{n}^2 all-Neumann solve: {nc} cycles, {dt*1e3:.2f} ms ({dt/nc*1e3:.3f} ms per cycle), ierr {ie}")
S.vcycle(2); S.sync()
print("vcycle", min(S.timed(lambda: S.vcycle(5)) / 5 for _ in range(3)) * 1e3, "us")
for lvl in range(1, S.ngrids):
    for name, op, cnt in (("relax5", _lib.OP_RELAX, 5), ("residual", _lib.OP_RESIDUAL, 1), ("restrict", _lib.OP_RESTRICT, 1), ("prolong", _lib.OP_PROLONG, 1)):
        S.op(op, lvl, cnt); S.sync()
        tt = min(S.timed(lambda: [S.op(op, lvl, cnt) for _ in range(10)]) / 10 for _ in range(3))
        print(f"level {lvl} {S.shapes[lvl-1]} {name}: {tt*1e3:.1f} us", flush=True)
S.close()
