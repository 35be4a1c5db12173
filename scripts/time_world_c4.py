"""BASELINE config[3] (1024 x 1024 x 512): solve-loop cycle of the single-domain solver and of the 8-slab loop-back world, Laplace or (argument `rhs`) manufactured Poisson problem (dev aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
ns = [1024, 1024, 512]; nr = int(os.environ.get("NR", "8"))
dx = 1.0 / (ns[0] - 1); mesh = [np.arange(n) * dx for n in ns]
u = np.random.default_rng(11).uniform(-1, 1, (ns[2], 1, 1)) * np.ones((1, ns[1], ns[0]))
for ov in ("1", "0"):
    os.environ["NDSM_HIP_OVERLAP"] = ov
    W = _lib.World(ns, mesh, "NDDNDD", nr)
    W.upload(_lib.BUF_U, u)
    if len(sys.argv) > 1 and sys.argv[1] == "rhs":      # general right-hand side (Poisson problems)
        W.upload(_lib.BUF_RHS, np.random.default_rng(12).uniform(-1, 1, (ns[2], 1, 1)) * np.ones((1, ns[1], ns[0])))
    else:
        W.zero_rhs()
    W.vcycle(2); W.sync()
    t = time.perf_counter(); W.vcycle(5); W.sync(); dt = (time.perf_counter() - t) / 5
    t = time.perf_counter(); W.solve(vc_tol=0.0, nmax=5); W.sync(); dt2 = (time.perf_counter() - t) / 5
    print(f"overlap {ov}: vcycle {dt*1e3:.2f} ms, solve-loop cycle {dt2*1e3:.2f} ms", flush=True)
    W.close()
