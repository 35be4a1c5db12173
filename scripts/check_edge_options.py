"""edge values of the options (no sweeps, no cycles, no coarsest-grid sweeps, zero tolerances) against the oracle port (dev aid)"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ndsm_amd
from oracle import Oracle
from golden_inputs import analytic_case
port = Oracle("port")
bad = 0
for ns in ([24, 20, 18], [40, 40, 40]):
    x, y, z, A1, b = analytic_case(ns)
    for ms, ncyc, nex, vtol, etol, mean in itertools.chain(
            itertools.product((0, 1), (0, 1, 2), (0, 1, 3), (1e-10,), (1e-13,), (False,)),
            [(5, 3, 10000, 0.0, 1e-13, False), (5, 1024, 10000, 1e-10, 0.0, True), (2, 5, 2, 1e-3, 1e-2, True)]):
        kw = dict(ms=ms, ncycles_max=ncyc, niterex_max=nex, vc_tol=vtol, ex_tol=etol, mean=mean)
        if etol == 0.0:
            kw["niterex_max"] = 50
        ierr, A, B = ndsm_amd.vector_potential(x, y, z, b.copy(), **kw)
        ierr2, A2, B2, _, _ = port.vector_potential(x, y, z, b, **kw)
        h = x[1] - x[0]; sc = max(np.abs(A2).max(), np.abs(b).max() * h)      # (A may be ~0: no cycles at all)
        ea, eb = np.abs(A - A2).max() / sc, np.abs(B - B2).max() / (sc * 4 / h)
        ok = ierr == ierr2 and ea <= 1e-9 and eb <= 1e-9 and np.isfinite(A).all() == np.isfinite(A2).all()
        if not ok:
            bad += 1
        print(("ok  " if ok else "BAD ") + f"{ns} {kw} ierr {ierr}/{ierr2} dA {ea:.1e} dB {eb:.1e}", flush=True)
print(f"{bad} mismatches")
sys.exit(1 if bad else 0)
