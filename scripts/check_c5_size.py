"""BASELINE config[4]'s grid (2048 x 2048 x 1024 = 2^32 points, 32 GiB per fp64 array) on ONE GPU:
one V-cycle of the single-domain solver (linear indices beyond 2^31/2^32) against the loop-back world
of 8 z-slabs (every slab well inside 32-bit range) - must be bit-identical.  Needs ~100 GiB of host
memory and ~180 GB of HBM; usage: check_c5_size.py [nx ny nz]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
n3 = [int(a) for a in sys.argv[1:4]] or [2048, 2048, 1024]
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
x = np.linspace(0.0, 1.0, n3[0]); dx = x[1] - x[0]
mesh = [x, np.arange(n3[1]) * dx, np.arange(n3[2]) * dx]
t = time.time()
rng = np.random.default_rng(7)
plane = rng.uniform(-1, 1, (n3[1], n3[0]))
u0 = np.empty((n3[2], n3[1], n3[0]))
zf = np.cos(np.arange(n3[2]) * 0.37) + 0.01 * np.arange(n3[2])
for k in range(n3[2]):
    np.multiply(plane, zf[k], out=u0[k])
    u0[k, (k * 7) % n3[1]] += 0.5          # break the separable structure a little
print(f"host field {u0.nbytes/2**30:.1f} GiB in {time.time()-t:.1f} s", flush=True)
bcs = "NDDNDD"
t = time.time()
S = _lib.MGSolver(n3, mesh, bcs); S.upload(1, _lib.BUF_U, u0); S.zero_rhs()
print(f"single domain up in {time.time()-t:.1f} s, {S.ngrids} grids", flush=True)
t = time.time(); S.vcycle(1); S.sync(); print(f"V-cycle {1e3*(time.time()-t):.1f} ms", flush=True)
ie, du, nc, hist = S.solve(vc_tol=0.0, nmax=1, hist_len=4); print("solve-loop cycle: du", du, flush=True)
t = time.time(); S.solve(vc_tol=0.0, nmax=3); S.sync(); print(f"solve-loop cycle {1e3*(time.time()-t)/3:.1f} ms", flush=True)
S.upload(1, _lib.BUF_U, u0); S.vcycle(1)
a = S.download(1, _lib.BUF_U); S.close()
assert np.isfinite(a).all()
print("single domain done; max|u| =", float(np.abs(a).max()), flush=True)
W = _lib.World(n3, mesh, bcs, 8); W.upload(_lib.BUF_U, u0); W.zero_rhs()
print("world: dist levels", W.dist_levels, flush=True)
t = time.time(); W.vcycle(1); W.sync(); print(f"world V-cycle {1e3*(time.time()-t):.1f} ms", flush=True)
b = W.download(_lib.BUF_U); W.close()
same = np.array_equal(a, b)
print("bit-identical:", same, flush=True)
if not same:
    d = np.argwhere(a != b)
    print("first mismatches (k,j,i):", d[:5].tolist(), "count", len(d))
    sys.exit(1)
