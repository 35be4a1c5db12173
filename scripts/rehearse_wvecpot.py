"""distributed vector-potential pipeline at a real size, N processes on one GPU over the RCCL test double
(dev aid): usage rehearse_wvecpot.py [n] [world] [mixed_precision mode]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("FAKE_RCCL_SLOT_MB", "64")
import test_gpu_multirank as T
import ndsm_amd
from golden_inputs import analytic_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mixed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cases = [{"ns": [n, n, n], "kw": ({"mixed_precision": mixed} if mixed else {})}]
with tempfile.TemporaryDirectory() as d:
    t = time.time()
    out = T._run_world(d, world, cases, timeout=900, worker="multirank_vecpot_worker.py")
    print(f"{world} ranks, {n}^3: {time.time()-t:.1f} s wall incl. process start", flush=True)
    for r in (0, world - 1):
        print(f"--- rank {r}")
        print("".join(l for l in open(os.path.join(out, f"rank{r}.log")) if "TIMING" in l))
    x, y, z, A1, b = analytic_case([n, n, n])
    ierr, A, B = ndsm_amd.vector_potential(x, y, z, b.copy(), **cases[0]["kw"])
    gA = np.concatenate([np.load(os.path.join(out, f"v0_A_r{r}.npy")) for r in range(world)], axis=1)
    gB = np.concatenate([np.load(os.path.join(out, f"v0_B_r{r}.npy")) for r in range(world)], axis=1)
    print("ierr", ierr, "A identical:", np.array_equal(gA, A), "B identical:", np.array_equal(gB, B))
    h = x[1] - x[0]
    print("Ea_max/h^2", np.linalg.norm(A1 - gA, axis=0).max() / h**2)
