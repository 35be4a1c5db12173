"""Step-by-step bring-up of the device path with a progress log (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "dbg.log"), "w")
def log(*a):
    s = " ".join(str(v) for v in a)
    LOG.write("%8.3f %s\n" % (time.time() - T0, s)); LOG.flush(); os.fsync(LOG.fileno())
    print(s, flush=True)
T0 = time.time()
import numpy as np
log("numpy ok")
import ndsm_amd
from ndsm_amd import _lib
from golden_inputs import rand_field, uniform_mesh
lib = ndsm_amd.load_library(); log("lib loaded")
log("device count", lib.ndsm_hip_device_count())
log("init rc", lib.ndsm_hip_init(0))
ns = [33, 22, 27]; mesh = uniform_mesh(ns)
u, rhs = rand_field(tuple(ns[::-1]), 2112), rand_field(tuple(ns[::-1]), 2113)
S = _lib.MGSolver(ns, mesh, "NDDNDD"); log("solver created", S.shapes)
S.upload(1, _lib.BUF_U, u); S.upload(1, _lib.BUF_RHS, rhs); log("uploaded")
S.op(_lib.OP_RELAX_COLOR, 1, 1); S.sync(); log("relax color ok")
S.op(_lib.OP_RESIDUAL, 1); S.sync(); log("residual ok")
S.op(_lib.OP_RESTRICT, 1); S.sync(); log("restrict ok")
S.op(_lib.OP_PROLONG, 1); S.sync(); log("prolong ok")
S.op(_lib.OP_RELAX, 3, 1); S.sync(); log("relax coarse ok")
S.op(_lib.OP_EXACT, 3, 1); S.sync(); log("exact ok", S.info())
S.vcycle(1); S.sync(); log("vcycle ok")
ierr, du, nc, hist = S.solve(hist_len=32); log("solve", ierr, du, nc)
S.close(); log("closed")
