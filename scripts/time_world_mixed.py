"""loop-back world (all slabs on one GPU, executed one after the other): fp64 vs mixed solve-loop cycle (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
ns = [1024, 1024, 512]; nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dx = 1.0 / (ns[0] - 1); mesh = [np.arange(n) * dx for n in ns]
rng = np.random.default_rng(11)
az, by, cx = rng.uniform(-1, 1, ns[2]), rng.uniform(-1, 1, ns[1]), rng.uniform(-1, 1, ns[0])
u = az[:, None, None] * by[None, :, None] + cx[None, None, :]
for prec in (0, 1):
    W = _lib.World(ns, mesh, "NDDNDD", nr); on = W.set_precision(prec) if prec else False
    W.upload(_lib.BUF_U, u); W.zero_rhs()
    W.solve(vc_tol=0.0, nmax=2); W.sync()
    t = time.perf_counter(); r = W.solve(vc_tol=0.0, nmax=5); W.sync(); dt = (time.perf_counter() - t) / 5
    print(f"{nr} slabs, precision {prec} (mixed on: {on}): {dt*1e3:.2f} ms per solve-loop cycle, du {r[1]:.3e}", flush=True)
    W.close()
S = _lib.MGSolver(ns, mesh, "NDDNDD")
for prec in (0, 1):
    S.set_precision(prec); S.upload(1, _lib.BUF_U, u); S.zero_rhs()
    S.solve(vc_tol=0.0, nmax=2); S.sync()
    t = time.perf_counter(); r = S.solve(vc_tol=0.0, nmax=5); S.sync(); dt = (time.perf_counter() - t) / 5
    print(f"single domain, precision {prec}: {dt*1e3:.2f} ms per solve-loop cycle, du {r[1]:.3e}", flush=True)
