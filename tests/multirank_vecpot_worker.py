"""One rank of the distributed vector-potential rehearsal (see tests/test_gpu_multirank.py):
usage multirank_vecpot_worker.py rank world outdir"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ndsm_amd  # noqa: E402
from ndsm_amd import _lib  # noqa: E402
from golden_inputs import analytic_case  # noqa: E402


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    cases = json.load(open(os.path.join(out, "cases.json")))
    L = ndsm_amd.load_library()
    assert L.ndsm_hip_init(0) == 0, _lib.last_error(L)
    uidf = os.path.join(out, "uid.bin")
    if rank == 0:
        with open(uidf + ".tmp", "wb") as f:
            f.write(_lib.dist_unique_id(L))
        os.rename(uidf + ".tmp", uidf)
    t0 = time.time()
    while not os.path.exists(uidf):
        assert time.time() - t0 < 60, "no unique id from rank 0"
        time.sleep(0.01)
    _lib.dist_init(rank, world, open(uidf, "rb").read(), L)
    assert "fake_rccl" in _lib.bound_libs(L)["rccl"], _lib.bound_libs(L)
    for ci, c in enumerate(cases):
        ns = c["ns"]
        x, y, z, A1, b = analytic_case(ns)
        if c.get("noise"):
            b = b + 0.3 * np.random.default_rng(c["noise"]).uniform(-1, 1, b.shape)
        a0 = None
        if c.get("guess"):
            a0 = np.random.default_rng(c["guess"]).uniform(-1, 1, b.shape)
        plan = _lib.slab_plan(ns, [x, y, z], world)
        z0, z1 = plan[rank]["z0"], plan[rank]["z1"]
        ierr, A, B = ndsm_amd.vector_potential_slab(x, y, z, b[:, z0:z1], rank, world, a_init=None if a0 is None else a0[:, z0:z1],
                                                    lib=L, **c.get("kw", {}))
        np.save(os.path.join(out, f"v{ci}_A_r{rank}.npy"), A)
        np.save(os.path.join(out, f"v{ci}_B_r{rank}.npy"), B)
        json.dump({"ierr": int(ierr), "z0": z0, "z1": z1}, open(os.path.join(out, f"v{ci}_r{rank}.json"), "w"))
    _lib.dist_finalize(L)
    print(f"rank {rank}: done", flush=True)


if __name__ == "__main__":
    main()
