"""tiny grids: ours against the oracle port where the reference's algorithm is defined (every dimension >= 4); below
that the reference itself dies (negative level count -> STOP / crash), this library returns an error code (dev aid)"""
import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ndsm_amd
from oracle import Oracle
from golden_inputs import analytic_case
port = Oracle("port")
for ns in ([4,4,4],[4,5,7],[5,5,5],[6,4,9],[7,7,4],[3,8,8],[2,9,9],[3,3,3],[2,2,2]):
    try:
        x, y, z, A1, b = analytic_case(ns)
    except Exception as e:
        print(ns, "analytic_case failed", e); continue
    if min(ns) >= 4:
        r2 = port.vector_potential(x, y, z, b.copy())
    else:
        r2 = ("(reference undefined)",)
    try:
        r1 = ndsm_amd.vector_potential(x, y, z, b.copy())
    except Exception as e:
        r1 = ("EXC " + str(e)[:60],)
    d = None
    if len(r1) >= 3 and len(r2) >= 3 and isinstance(r1[1], np.ndarray):
        d = (np.abs(r1[1]-r2[1]).max(), np.abs(r1[2]-r2[2]).max())
    print(ns, "ours ierr", r1[0], "port ierr", r2[0], "diff", d, flush=True)
