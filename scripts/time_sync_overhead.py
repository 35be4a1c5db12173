"""what the per-cycle host round trip costs on small grids (dev aid): solve-loop cycles (metric read back and tested on
the host after every cycle) against the same number of V-cycles enqueued without a wait.  usage: time_sync_overhead.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    mesh = [np.linspace(0, 1, n)] * 3
    S = _lib.MGSolver([n, n, n], mesh, "NDDNDD"); S.zero_rhs()
    u0 = np.random.default_rng(1).uniform(-1, 1, (n, n, n))
    S.upload(1, _lib.BUF_U, u0); S.solve(vc_tol=0.0, nmax=3); S.sync()
    k = 20
    t = time.perf_counter(); S.solve(vc_tol=0.0, nmax=k); S.sync(); a = (time.perf_counter() - t) / k
    S.vcycle(3); S.sync()
    t = time.perf_counter(); S.vcycle(k); S.sync(); b = (time.perf_counter() - t) / k
    dev = S.timed(lambda: S.vcycle(k)) / k
    print(f"{n}^3: solve-loop cycle {a*1e6:7.0f} us   V-cycles without a wait {b*1e6:7.0f} us (device time {dev*1e3:7.0f} us)", flush=True)
    S.close()
