"""BASELINE.json configs 0-2 through the C ABI: level-count override, large pipeline, timings (dev aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import analytic_case, manufactured_poisson, uniform_mesh
L = ndsm_amd.load_library(); assert L.ndsm_hip_init(0) == 0
# config 0: 64^3, 3-level V-cycle (coarsest 16^3 -> host-driven exact solve), vs oracle
from oracle import Oracle
port = Oracle("port")
ns = [64, 64, 64]; mesh = uniform_mesh(ns); us, rhs = manufactured_poisson(mesh, "NDDNDD")
t = time.time(); ierr, u, du, hist, nc = _lib.poisson_solve(np.zeros_like(us), rhs, mesh, "NDDNDD", ngrids=3, hist_len=64); dt = time.time() - t
ierr2, u2, du2, hist2, nc2, sw = port.solve_bvp(np.zeros_like(us), rhs, mesh, "NDDNDD", ngrids=3, hist_len=64)
print("config0 64^3 3-level: ierr", ierr, "cycles", nc, nc2, "bitwise", np.array_equal(u, u2), "hist eq", list(hist) == list(hist2), "%.2fs" % dt, flush=True)
# config 1: 256^3 6-level
ns = [256, 256, 256]; mesh = uniform_mesh(ns); us, rhs = manufactured_poisson(mesh, "NDDNDD")
t = time.time(); ierr, u, du, hist, nc = _lib.poisson_solve(np.zeros_like(us), rhs, mesh, "NDDNDD", ngrids=6, hist_len=64); dt = time.time() - t
print("config1 256^3 6-level: ierr", ierr, "cycles", nc, "du", du, "err", np.abs(u - us).max(), "%.2fs" % dt, flush=True)
# config 2: 512^3 full vector-potential pipeline through ndsm_vector_solve
n = int(os.environ.get("NBIG", "512"))
x, y, z, A1, b1 = analytic_case(n)
t = time.time(); ierr, A, B = ndsm_amd.vector_potential(x, y, z, b1); dt = time.time() - t
eA = np.linalg.norm(A1 - A, axis=0); eB = np.linalg.norm(b1 - B, axis=0)
print(f"config2 {n}^3 vector potential: ierr {ierr} Ea_max {eA.max():.5e} Ea_avg {eA.mean():.5e} Eb_max {eB.max():.5e} Eb_avg {eB.mean():.5e} wall {dt:.2f}s", flush=True)
