"""Loop-back z-slab world (all slabs on this GPU): V-cycle timing for the C4 shape (dev aid).
usage: world_profile.py nx ny nz nslabs [ncycles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ndsm_amd
from ndsm_amd import _lib
nx, ny, nz, nr = (int(v) for v in sys.argv[1:5])
nc = int(sys.argv[5]) if len(sys.argv) > 5 else 3
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
h = 1.0 / (nx - 1)
mesh = [np.arange(n) * h for n in (nx, ny, nz)]
W = _lib.World([nx, ny, nz], mesh, "NDDNDD", nr)
print("slabs", [(s["z0"], s["z1"], s["g"]) for s in W.slabs][:3], "...", flush=True)
rng = np.random.default_rng(1)
u = rng.uniform(-1, 1, (nz, ny, nx))
W.upload(_lib.BUF_U, u)
W.zero_rhs() if hasattr(W, "zero_rhs") else None
W.vcycle(1); W.sync()
t = W.timed(lambda: W.vcycle(nc)) / nc
print(f"{nx}x{ny}x{nz} in {nr} loop-back slabs: {t:.2f} ms per V-cycle (all slabs serial on one GPU)", flush=True)
W.close()
