"""repeated calls of every pipeline entry: device memory must not creep (dev aid)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ndsm_amd
from ndsm_amd import _lib
from golden_inputs import analytic_case
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so", mode=os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
def free_mb():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20
x, y, z, A1, b = analytic_case([48, 40, 36])
x2, y2, z2, _, b2 = analytic_case([40, 48, 36])
base = None
for it in range(60):
    ndsm_amd.vector_potential(x, y, z, b)
    if it % 7 == 3:
        ndsm_amd.vector_potential(x2, y2, z2, b2)          # another mesh: the cache is replaced
    V = ndsm_amd.VecPot(x, y, z); V.solve(b); V.solve(b, device=True); V.close()
    S = _lib.MGSolver([48, 40, 36], [x, y, z], "NDDNDD"); S.upload(1, _lib.BUF_U, np.zeros((36, 40, 48))); S.solve(nmax=3); S.close()
    if it == 5:
        base = free_mb()
    if it % 10 == 9:
        print(f"iteration {it}: free {free_mb():.1f} MiB (after warm-up: {base:.1f})", flush=True)
drift = base - free_mb()
print("drift MiB:", drift)
assert abs(drift) < 64, drift
L.ndsm_hip_shutdown()
print("after shutdown free", free_mb())
