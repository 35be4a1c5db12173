"""BASELINE config[3] shape (1024 x 1024 x 512) in loop-back: 8 slabs, distributed levels by the default rule,
exchange overlap on - two V-cycles against the single-domain solver, bit for bit (one-off check)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
ns = [1024, 1024, 512]; nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dx = 1.0 / (ns[0] - 1)
mesh = [np.arange(n) * dx for n in ns]
rng = np.random.default_rng(7)
u = rng.uniform(-1, 1, tuple(ns[::-1])); print("field ready", flush=True)
W = _lib.World(ns, mesh, "NDDNDD", nr); print("world: dist levels", W.dist_levels, flush=True)
W.upload(_lib.BUF_U, u); W.zero_rhs(); W.vcycle(2); b = W.download(_lib.BUF_U); W.close(); print("world done", flush=True)
S = _lib.MGSolver(ns, mesh, "NDDNDD"); S.upload(1, _lib.BUF_U, u); S.zero_rhs(); S.vcycle(2); a = S.download(1, _lib.BUF_U); S.close()
nd = int((a != b).sum())
print(f"1024x1024x512, {nr} slabs: {nd} differing points, max |diff| {np.abs(a - b).max():.3e}", flush=True)
